// spmm_panel_v2.h -- the LDS-panel kernel with the N dimension register-blocked: the wide-N form of
// spmm_csr_panel (spmm_csr_kernels.h) for N >= 32.
//
// The reference re-streams its packed non-zero list once per 8-column N tile (read_A, sextans.cpp:57-60,84-87)
// while the B window stays on chip (PEG_Bmtx local_B, sextans.cpp:337,353-381).  The N = 16 kernel inherited that
// shape: one workgroup per (row block, 16-column tile), so at N = 128 every block's meta data, dictionary, A stream
// and per-entry index/broadcast work was done 8 times.  Here
//   * every lane owns 4*H output columns (H "half panels" of 16 columns, H = 2 -> 8 columns per lane): one index
//     unpack, one DPP broadcast pair and one LDS address serve 2x the multiply-adds; the H B rows of an entry are
//     read with H ds_read_b128 at a compile-time offset from the same address register;
//   * a workgroup walks `tpw` such 16*H-column super tiles (tile loop inside the workgroup): block meta, row
//     extents and dictionary indices are fetched once, the A stream comes from HBM once and from L2 afterwards.
// Same packed plan as spmm_csr_panel (panel_plan.h; dictionary-only plans), same 16-bit byte-offset stream, same
// exact-safe padding, same per-row order: bit-identical to cpu_spmm_CSR (sparse_helper.h:262-290).
// LDS: H half panels at a fixed stride (kWideHalfBytes, so the half offset folds into the ds_read immediate) =
// 73.9 KB for H = 2 -> 2 workgroups per CU, up to 256 VGPRs per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spmm_csr_kernels.h"

namespace sx {

constexpr int kWideMaxDict = 576;                                // dictionary capacity of the plan at 4 lanes per row (9 * 64)
constexpr int kWidePadRows = 16;                                 // +1.0f rows behind the dictionary (= kPlanPadRows of plan_device.h)
constexpr int kWideHalfBytes = (kWideMaxDict + kWidePadRows) * 64;   // one half panel: 576 B rows of 64 bytes + the +1.0f rows

template <int G>
__device__ __forceinline__ int quad_bcast(int x) {               // value of lane G of the 4-lane row group
    return __builtin_amdgcn_update_dpp(0, x, G * 0x55, 0xF, 0xF, true);
}

template <bool EXACT>
__device__ __forceinline__ void mac4v(f32x4 &acc, float a, const f32x4 &b) {
    acc.x = mac<EXACT>(acc.x, a, b.x);
    acc.y = mac<EXACT>(acc.y, a, b.y);
    acc.z = mac<EXACT>(acc.z, a, b.z);
    acc.w = mac<EXACT>(acc.w, a, b.w);
}

// Entries 4G .. 4G+3 of the current 16-entry batch (held by lane G of the row group): broadcast, 4*H LDS reads in
// flight, then the multiply-adds in entry order.
template <int G, int H, bool EXACT>
__device__ __forceinline__ void wide_quad(const int (&ix)[4], const float (&vx)[4], const char *pq, f32x4 (&acc)[H]) {
    int i[4];
    float a[4];
    f32x4 b[4][H];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        i[e] = quad_bcast<G>(ix[e]);
        a[e] = __int_as_float(quad_bcast<G>(__float_as_int(vx[e])));
#pragma unroll
        for (int h = 0; h < H; ++h) b[e][h] = *reinterpret_cast<const f32x4 *>(pq + i[e] + h * kWideHalfBytes);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < H; ++h) mac4v<EXACT>(acc[h], a[e], b[e][h]);
}

// Pins the accumulators at this point of the program: the multiply-adds before it cannot sink below it and (being a
// volatile asm with a memory clobber) the LDS reads after it cannot be hoisted above it.  Without it the compiler
// gathers the reads of a whole batch (128 registers at H = 2) in front of the arithmetic and spills.
template <int H>
__device__ __forceinline__ void pin(f32x4 (&acc)[H]) {
#pragma unroll
    for (int h = 0; h < H; ++h) asm volatile("" : "+v"(acc[h]) : : "memory");
}

template <int H, bool EXACT>
__device__ __forceinline__ void wide_batch(const int (&ix)[4], const float (&vx)[4], const char *pq, f32x4 (&acc)[H]) {
    wide_quad<0, H, EXACT>(ix, vx, pq, acc); pin<H>(acc);
    wide_quad<1, H, EXACT>(ix, vx, pq, acc); pin<H>(acc);
    wide_quad<2, H, EXACT>(ix, vx, pq, acc); pin<H>(acc);
    wide_quad<3, H, EXACT>(ix, vx, pq, acc); pin<H>(acc);
}

// Async global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): the LDS destination is the wave-uniform
// `lds_dst` + 16 * lane, the global source is per lane.  No staging registers, no ds_write pass.
__device__ __forceinline__ void glds16(const float *gsrc, char *lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// Vector-memory operations the COMPILER DOES NOT SEE (inline asm: saddr + 32-bit voffset form).  Everything that is in
// flight while a super tile's rows are multiplied -- C_in of the tile, the B rows of the next panel, the C stores of
// the previous tile -- goes through these, and is waited for by ONE explicit `s_waitcnt vmcnt(0)` after the row loop.
// With compiler-visible loads/stores the wait-count insertion pass (which must assume the worst over loop back-edges
// and protects registers it recycles as address temporaries) put full-drain waits -- including waits for the previous
// tile's STORES to be acknowledged -- at the top of every super tile: 12 k of 16 k cycles per tile were spent waiting.
// Rules (cdna_hip_programming.md, inline-asm section): a destination counts as written at the asm statement, so
// nothing may read it before the matching drain; the drain (the s_waitcnt after the row loop and the pins right behind it)
// names every such register as an in/out operand.
__device__ __forceinline__ void aload4(f32x4 &dst, const float *sbase, unsigned voff_bytes) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
__device__ __forceinline__ void aload1(float &dst, const float *sbase, unsigned voff_bytes) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
__device__ __forceinline__ void astore1(float *sbase, unsigned voff_bytes, float val) {
    asm volatile("global_store_dword %0, %1, %2" : : "v"(voff_bytes), "v"(val), "s"(sbase) : "memory");
}
__device__ __forceinline__ void astore4(float *sbase, unsigned voff_bytes, const f32x4 &val) {
    asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(voff_bytes), "v"(val), "s"(sbase) : "memory");
}
// (the same with a 64-bit address per lane: row-major C beyond 4 GB, RM == 2)
__device__ __forceinline__ void aload4p(f32x4 &dst, const float *p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void astore4p(float *p, const f32x4 &val) {
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(val) : "memory");
}

// BCOL: Bp is the caller's column-major B (panel_stride = its leading dimension), staged with 4-byte loads; else Bp
// holds row-major K x 16 panels at stride panel_stride floats.
// Super tile st covers columns [16*H*st, 16*H*(st+1)) of this launch; workgroup (blk, grp) walks super tiles
// [grp*tpw, min(nsuper, (grp+1)*tpw)).
// Phases of one super tile, software-pipelined ACROSS super tiles so that only the first panel of a workgroup is
// waited for: [B rows of super tile st+1 requested into registers] -> row stream of st out of the LDS panel -> C of st
// stored straight from the accumulators (a lane owns 4 consecutive columns of one row: a wavefront's store covers 16
// consecutive rows x 4 columns = 64-byte runs, merged to full lines in L2) -> barrier -> registers -> panel -> barrier.
// TIMED (engine option "phase_timing"; diagnostic instantiations only): one workgroup in 16 adds its wavefront-0 cycle counts
// per phase to dbg[0..3], a launch counter to dbg[4] and the same span in 100 MHz ticks to dbg[5].
// DCAP: dictionary capacity of the launch in units of 64 rows (9 = the plan's maximum; the small-matrix instantiation uses the
// plan's actual maximum, so that a block with 235 dictionary rows does not issue the loads and LDS writes of 576).
// (8 instead of 4 B rows in flight per lane for launches that cannot fill the chip -- one wavefront per SIMD -- was measured on
// nasa4704: row loop 2833 vs 2829 cycles, not kept.)
// CROW (the REORDERED form, reorder_kernels.h): Cin == Cout == the row-major staging buffer Cs[tile][row][16] (ldc_in == ldc ==
// floats per tile = 16 M): C_in of a tile is ONE 16-byte load per lane and C_out one 16-byte store -- the four lanes of a slot cover
// the 64 contiguous bytes of row slot_row[slot], whatever row of the matrix that is; blk_dict then holds RELABELLED columns (rows
// of the permuted B panels).
// BIG: 2 instead of 4 workgroups per CU = up to 256 registers per lane: the column-major staging (BCOL) of a matrix with long rows
// keeps a panel AND 4 .. 6 batches of row entries in registers (it spills at 128), for launches too small to fill the chip anyway.
// SETS (round 4, short rows): a block has SETS * 64 row slots that share ONE dictionary / panel; the workgroup fetches the entries of
// all its sets in the prologue and multiplies them set after set.  Block meta, the two dependent round trips of the prologue and the
// panel are paid once per 128 rows, and a 16 x 4 x 2 brick of a 27-point mesh needs 3.4 dictionary rows per matrix row where a
// 16 x 2 x 2 brick needs 4.5 (the panel copy is more than half of the bytes a short-row block moves through the L1).
// RM (round 5, the row-major entry point sextans_spmm_device_rm): the caller's operands ARE the layouts this kernel wants -- B row-major
// K x N (Bp = B, panel_stride = its leading dimension: dictionary row `col` of tile st is the 64 bytes at B + col * ldb + 16 st; at N = 16
// the caller's B IS panel 0) and C row-major M x N (ldc_in / ldc = row strides: the 16-byte accesses of CROW go straight to the caller's
// rows, C_in + row * ldc_in + 16 st).  No repack launch, no staging passes: the reference lays its operands out for its kernel outside
// the timed call too (sextans-host.cpp:150-195).  The lanes whose 4 columns lie beyond N in the last tile (last_cols = 8) neither copy B
// nor touch C there (a lane only ever reads its own 16-byte column slice of the LDS panel, so what those slices hold does not matter).
// RM == 2: the same with 64-bit lane addresses into C (row * ldc * 4 bytes does not fit 32 bits: 4M rows x 512 columns are 8 GB) -- two
// more registers per row set and direction, a 64-bit add per access; RM == 1 keeps the scalar-base + 32-bit-offset form.
template <int H, int NB, bool EXACT, bool BCOL, bool TIMED = false, int DCAP = 9, bool CROW = false, bool BIG = false, int SETS = 1, int RM = 0>
// (BCOL at the full dictionary capacity keeps a 576-row panel in registers next to the row entries: 132 registers' worth -- at 4
// workgroups per CU it spilled 4 registers to scratch in the prologue of the small-matrix launches it exists for; 3 per CU = 168.)
#ifndef SX_V2_BCOL_WGS
#define SX_V2_BCOL_WGS 3
#endif
__global__ __launch_bounds__(kBlock, (H == 1 && !BIG ? (BCOL && DCAP == 9 ? SX_V2_BCOL_WGS : 4) : 2)) void spmm_csr_panel_v2(
    const int2 *__restrict__ slot_info, const unsigned short *__restrict__ p_idx16, const float *__restrict__ p_val,
    const int *__restrict__ blk_row, const int *__restrict__ dict_cnt, const int *__restrict__ blk_dict, int dict_stride,
    const float *__restrict__ Bp, int64_t panel_stride, const float *Cin, int64_t ldc_in, float *Cout, int64_t ldc,
    int nsuper, int tpw, int nblk, float alpha, float beta, int use_xcd_remap, int pad_row, int blk_begin, int row_base,
    const unsigned char *__restrict__ skip, long long *dbg, const int *__restrict__ slot_row, const int2 *__restrict__ slot_ioff, int last_cols,
    const int *__restrict__ blk_list) {
    // blk_list (mixed plans, split form; may be null): the launch walks the listed blocks only -- those that have a dictionary; the rows of
    // the others belong to the gather kernel (nblk = length of the list, blk_begin unused)
    // last_cols (16-column tiles, column-major C): valid columns of the LAST super tile of the launch -- 8 when N = 16 t + 8 runs as
    // t + 1 tiles (its B panel is zero there): lanes whose 4 columns lie beyond neither load C_in nor store C for that tile
    // slot_ioff (may be null): {where a slot's 16-bit index list starts in p_idx16, shift in bytes to add to every offset of the list}
    // when consecutive rows whose lists are equal up to a constant shift share one copy (plan_device.hip: share_index_lists); null =
    // own list at the slot's first packed entry, like its values.  The shift goes into this lane's LDS base address once.
    static_assert(!CROW || (H == 1 && !BCOL), "the block-major C staging exists for 16-column tiles on repacked panels");
    static_assert(!RM || (CROW && !TIMED && !BIG), "row-major operands use the 16-byte C accesses of the staging form");
    static_assert(SETS == 1 || (H == 1 && !BCOL && !TIMED && !BIG && NB <= 2), "several row sets per block: 16-column tiles on repacked panels, short rows");
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, w0 = 0;
    if constexpr (TIMED) { t0 = clock64(); w0 = wall_clock64(); }
    constexpr int LPR = 4;
    constexpr int RB = kBlock / LPR;          // 64 row slots per set
    constexpr int RBS = RB * SETS;            // row slots per block
    constexpr int NTT = 16 * H;               // columns per super tile
    constexpr int BATCH = 16;
    constexpr int MAXD = DCAP;                // dictionary capacity of this launch = MAXD * RB (<= 9 * RB, the plan's limit)
    extern __shared__ __attribute__((aligned(16))) int smem[];
    char *lds = reinterpret_cast<char *>(smem);

    const int ngrp = (nsuper + tpw - 1) / tpw;
    const unsigned nwg = (unsigned)nblk * (unsigned)ngrp;
    unsigned wg = blockIdx.x;
    if (use_xcd_remap) wg = xcd_remap(wg, nwg);
    const int blk = blk_list ? blk_list[wg / (unsigned)ngrp] : blk_begin + (int)(wg / (unsigned)ngrp);
    const int grp = (int)(wg % (unsigned)ngrp);
    const int st_begin = grp * tpw, st_end = min(nsuper, st_begin + tpw);

    const int tid = threadIdx.x;
    const int slot = tid / LPR;
    const int q = tid % LPR;
    // first round trip: everything at addresses that depend on the block number only
    const int row0 = blk_row[blk];
    const int row1 = blk_row[blk + 1];
    const int nu = dict_cnt[blk];
    int2 si[SETS], io2[SETS];
#pragma unroll
    for (int t = 0; t < SETS; ++t) {
        si[t] = slot_info[(int64_t)blk * RBS + t * RB + slot];
        io2[t] = slot_ioff ? slot_ioff[(int64_t)blk * RBS + t * RB + slot] : make_int2(si[t].x, 0);
    }
    unsigned boff[MAXD];                      // float offset of "my" dictionary rows inside a K x 16 panel (+ my 4 columns)
    {
        const int *bd = blk_dict + (int64_t)blk * dict_stride + slot;
#pragma unroll
        for (int u = 0; u < MAXD; ++u) {
            const int col = bd[min(u * RB, dict_stride - RB)];
            boff[u] = BCOL ? (unsigned)col : RM ? (unsigned)col * (unsigned)panel_stride + 4u * (unsigned)q : (unsigned)col * 16u + 4u * (unsigned)q;
        }
    }
    int len[SETS];
#pragma unroll
    for (int t = 0; t < SETS; ++t) len[t] = si[t].y;
    if constexpr (TIMED) {   // first round trip done: block meta, row extents, dictionary indices
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(boff[0]), "+v"(boff[MAXD - 1]) : : "memory");
        t1 = clock64();
    }
    // Row stream: wave-uniform base (first entry of the wave's first row) + a 32-bit lane offset, so every fetch is one
    // `global_load saddr + voffset + immediate` without 64-bit vector address arithmetic.  Slots past the block's last
    // row ({0, 0}) fetch from the base (never consumed).  The stream is padded: over-reads stay in bounds.
    // (several sets: one base -- the rows of a later set follow those of the first in the stream, and so do their index lists)
    const int wbase = __builtin_amdgcn_readfirstlane(si[0].x);
    const int wbase_i = __builtin_amdgcn_readfirstlane(io2[0].x);
    const float *pv = p_val + wbase;
    const unsigned short *pi = p_idx16 + wbase_i;
    unsigned loff[SETS], loffi[SETS];
    // C: this lane's row (clamped for the loads) and whether it is written
    // (slot_row: the plan walks the rows in clustered order -- row_cluster.hip -- and this table, at an address that depends on the
    // block number only, says which row of the matrix a slot is; without it blocks are runs of consecutive rows)
    int myrow[SETS];
    unsigned coff[SETS];
    bool cwrite[SETS];
    const char *pq[SETS];
#pragma unroll
    for (int t = 0; t < SETS; ++t) {
        loff[t] = (len[t] > 0 ? (unsigned)(si[t].x - wbase) : 0u) + 4u * (unsigned)q;
        loffi[t] = (len[t] > 0 ? (unsigned)(io2[t].x - wbase_i) : 0u) + 4u * (unsigned)q;
        myrow[t] = slot_row ? slot_row[(int64_t)blk * RBS + t * RB + slot] : min(row0 + t * RB + slot, row1 - 1);
        coff[t] = (unsigned)(myrow[t] - row_base);
        cwrite[t] = row0 + t * RB + slot < row1 && !(skip && skip[myrow[t]]);
        pq[t] = lds + 16 * q + io2[t].y;
    }

    // B rows of a panel travel through registers: bv (16-byte loads from the repacked panels) or bs (column-major B:
    // four 4-byte loads per row and lane).  `first`: plain loads the compiler tracks (preheader); otherwise the asm
    // forms above, drained explicitly after the row loop.
    f32x4 bv[BCOL ? 1 : MAXD][H];
    float bs[BCOL ? MAXD : 1][H][4];
    unsigned bvoff[MAXD];                     // byte offset of "my" 16 bytes of dictionary row u from the panel / column base
#pragma unroll
    for (int u = 0; u < MAXD; ++u)
        bvoff[u] = BCOL ? (boff[u] + 4u * (unsigned)q * (unsigned)panel_stride) * 4u : boff[u] * 4u;
    auto load_panel = [&](int st, bool first) {
#pragma unroll
        for (int u = 0; u < MAXD; ++u)
#pragma unroll
            for (int h = 0; h < H; ++h) {
                if constexpr (BCOL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float *sb = Bp + ((int64_t)(st * H + h) * 16 + j) * panel_stride;   // uniform
                        if (first) bs[u][h][j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sb) + bvoff[u]);
                        else aload1(bs[u][h][j], sb, bvoff[u]);
                    }
                } else {
                    const float *sb = Bp + (int64_t)(st * H + h) * panel_stride;                   // uniform
                    if (first) bv[u][h] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(sb) + bvoff[u]);
                    else aload4(bv[u][h], sb, bvoff[u]);
                }
            }
    };
    // H = 1 (16-column tiles, 4 workgroups per CU, 128 registers): no room for a panel in registers -- every panel goes
    // straight from global memory into LDS (LDS-DMA: chunk u = dictionary entries 64u .. 64u+63, entry (64u + slot) is
    // copied by the 4 lanes of `slot`, a wave's 64 lanes write 1 KiB of consecutive LDS; rows past the dictionary stay
    // unwritten and are never read).
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool cvalid = BCOL || H != 1 || 4 * q < last_cols;   // my 4 columns exist in the last tile too (column-major staging never merges a tail)
    // (TIMING-ONLY builds, tools/build_variant.py -DSX_TIMING_DMA_PERCENT=p: only the first p % of a block's dictionary rows are copied --
    // WRONG results -- to bound what a workgroup that kept rows of the previous block in LDS could save: profiles/r06_panel_copy_bound.txt)
#ifndef SX_TIMING_DMA_PERCENT
#define SX_TIMING_DMA_PERCENT 100
#endif
    const int nu_dma = SX_TIMING_DMA_PERCENT == 100 ? nu : nu * SX_TIMING_DMA_PERCENT / 100;
    auto dma_panel = [&](int st) {
#pragma unroll
        for (int u = 0; u < MAXD; ++u)
#pragma unroll
            for (int h = 0; h < H; ++h)
                if (u * RB + slot < nu_dma && (!RM || cvalid || st + 1 < nsuper))
                    glds16(Bp + (RM ? (int64_t)st * 16 : (int64_t)(st * H + h) * panel_stride) + boff[u], lds + h * kWideHalfBytes + (u * RB + wave * 16) * 64);
        // the +1.0f rows the padding entries (value -0.0f) address: kWidePadRows of them, a shifted shared list points further in
#pragma unroll
        for (int h = 0; h < H; ++h) *reinterpret_cast<float *>(lds + h * kWideHalfBytes + pad_row * 64 + 4 * tid) = 1.0f;   // 256 threads = 16 rows
    };
    constexpr bool DMA = (H == 1) && !BCOL;   // (column-major staging, BCOL, goes through registers: small matrices with short rows)
    auto store_panel = [&]() {                // registers -> LDS panel (+ the +1.0f row the padding entries point at)
#pragma unroll
        for (int u = 0; u < MAXD; ++u)
#pragma unroll
            for (int h = 0; h < H; ++h)   // entries past the dictionary rewrite its last row
                *reinterpret_cast<f32x4 *>(lds + h * kWideHalfBytes + max(min(slot + min(u * RB, dict_stride - RB), nu - 1), 0) * 64 +
                                           16 * q) = BCOL ? f32x4{bs[u][h][0], bs[u][h][1], bs[u][h][2], bs[u][h][3]} : bv[u][h];
        // the +1.0f rows the padding entries (value -0.0f) address: kWidePadRows of them, a shifted shared list points further in
#pragma unroll
        for (int h = 0; h < H; ++h) *reinterpret_cast<float *>(lds + h * kWideHalfBytes + pad_row * 64 + 4 * tid) = 1.0f;   // 256 threads = 16 rows
    };

    // ---- the row's entries: the first NB batches (96 entries) are loaded ONCE per block and stay in registers for every
    // super tile (indices already unpacked): the row loop below then contains no memory waits at all, and the loads
    // that ARE in flight during it (C_in, next panel) are only waited for after it.  Longer rows continue from the
    // stream (L2) with plain loads.
    static_assert(NB >= 1 && NB <= 6, "register-resident batches");
    f32x4 av[SETS][NB];
    int ai[SETS][NB][4];
    {
        uint2 aw[SETS][NB];
#pragma unroll
        for (int t = 0; t < SETS; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                // (unconditional, immediate offsets: all 2 * NB loads are in flight together; lanes whose row ends earlier
                // over-read inside the padded stream and never multiply what they read.  NB is chosen per matrix from its
                // mean row length so that short-row matrices do not multiply their A traffic.)
                // (non-temporal loads of this read-once stream -- so that it would not push the B lines neighbouring blocks share out of L2 --
                // were measured on the reordered form, where B is re-fetched 5x: kernel 744 -> 936 us.  Plain loads.)
                av[t][b] = *reinterpret_cast<const f32x4 *>(pv + loff[t] + b * BATCH);
                aw[t][b] = *reinterpret_cast<const uint2 *>(pi + loffi[t] + b * BATCH);
            }
        // All dictionary indices are waited for HERE, in one counted wait behind which the row loads above stay in flight.  Left to
        // itself the compiler waits for boff[1] after the first (conditional) panel request -- and after a control-flow join its
        // wait-count bookkeeping can only say vmcnt(0): the row entries and the first piece had to land before pieces 2..9 were even
        // requested, one extra memory round trip in every workgroup's prologue.  Same-box A/B on the 4M-row FEM matrix (tools/ab.py):
        // N = 16 720 -> 695 us (a first comparison ACROSS boxes had said 687 -> 712: box-to-box spread is +-4 %).
        static_assert(MAXD == 9 || MAXD == 5, "the pin below names every boff register");
        if constexpr (MAXD == 9)
            asm volatile("" : "+v"(boff[0]), "+v"(boff[1]), "+v"(boff[2]), "+v"(boff[3]), "+v"(boff[4]), "+v"(boff[5]), "+v"(boff[6]),
                              "+v"(boff[7]), "+v"(boff[8]));
        else
            asm volatile("" : "+v"(boff[0]), "+v"(boff[1]), "+v"(boff[2]), "+v"(boff[3]), "+v"(boff[4]));
        if constexpr (DMA) dma_panel(st_begin); else load_panel(st_begin, true);
#pragma unroll
        for (int t = 0; t < SETS; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                ai[t][b][0] = (int)(aw[t][b].x & 0xffffu); ai[t][b][1] = (int)(aw[t][b].x >> 16);
                ai[t][b][2] = (int)(aw[t][b].y & 0xffffu); ai[t][b][3] = (int)(aw[t][b].y >> 16);
            }
    }
    // C: column (col0 + 16h + 4q + j) of this lane = uniform column base (col0 + 16h + j) + a per-lane byte offset
    unsigned cvoff_in[SETS], cvoff_out[SETS], cvoff_rin[SETS], cvoff_rout[SETS];
#pragma unroll
    for (int t = 0; t < SETS; ++t) {
        cvoff_in[t] = (4u * (unsigned)q * (unsigned)ldc_in + coff[t]) * 4u;
        cvoff_out[t] = (4u * (unsigned)q * (unsigned)ldc + coff[t]) * 4u;
        // CROW: my 16 bytes of the tile -- row-major staging (rows of 16 floats), or the caller's own row-major C (RM)
        cvoff_rin[t] = RM ? (coff[t] * (unsigned)ldc_in + 4u * (unsigned)q) * 4u : ((unsigned)myrow[t] * 16u + 4u * (unsigned)q) * 4u;
        cvoff_rout[t] = RM ? (coff[t] * (unsigned)ldc + 4u * (unsigned)q) * 4u : cvoff_rin[t];
    }
    const int64_t ct_in = RM ? 16 : ldc_in, ct_out = RM ? 16 : ldc;   // CROW: floats from one tile of C to the next
    // (row-major operands with two row sets: the two byte offsets per set are formed again where they are used -- kept live across the
    // tile loop they cost the two registers that sent a 64-bit value to scratch, and its reload inside the loop is a vector-memory
    // operation whose wait drains the C stores and the next panel in flight: 27-point 1-dof grid, N = 64 .. 256, 8 .. 16 % of the launch)
    constexpr bool kRematC = RM == 1 && SETS == 2;
    auto off_rin = [&](int t) -> unsigned {
        if constexpr (kRematC) { unsigned c = coff[t]; asm volatile("" : "+v"(c)); return (c * (unsigned)ldc_in + 4u * (unsigned)q) * 4u; }
        else return cvoff_rin[t];
    };
    auto off_rout = [&](int t) -> unsigned {
        if constexpr (kRematC) { unsigned c = coff[t]; asm volatile("" : "+v"(c)); return (c * (unsigned)ldc + 4u * (unsigned)q) * 4u; }
        else return cvoff_rout[t];
    };
    const float *cp_in[SETS];   // RM == 2: this lane's 16 bytes of tile 0, as 64-bit addresses
    float *cp_out[SETS];
#pragma unroll
    for (int t = 0; t < SETS; ++t) {
        cp_in[t] = Cin + (int64_t)coff[t] * ldc_in + 4 * q;
        cp_out[t] = Cout + (int64_t)coff[t] * ldc + 4 * q;
    }
    auto ptr_in = [&](int t) -> const float * {   // (RM == 2: the 64-bit lane addresses likewise)
        if constexpr (RM == 2 && SETS == 2) { unsigned c = coff[t]; asm volatile("" : "+v"(c)); return Cin + (int64_t)c * ldc_in + 4 * q; }
        else return cp_in[t];
    };
    auto ptr_out = [&](int t) -> float * {
        if constexpr (RM == 2 && SETS == 2) { unsigned c = coff[t]; asm volatile("" : "+v"(c)); return Cout + (int64_t)c * ldc + 4 * q; }
        else return cp_out[t];
    };
    // C_in of the FIRST super tile is requested here, in the same round trip as the panel and the row entries (for a
    // matrix of a few thousand rows the whole kernel is three round trips: one more is 15 % of its time)
    float cin[SETS][H][4];
    f32x4 cinv[SETS];
#pragma unroll
    for (int t = 0; t < SETS; ++t) {
        cinv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (CROW) {
            if (!RM || cvalid || st_begin + 1 < nsuper) {
                if constexpr (RM == 2) aload4p(cinv[t], ptr_in(t) + (int64_t)st_begin * 16);
                else aload4(cinv[t], Cin + (int64_t)st_begin * ct_in, off_rin(t));
            }
        } else if (cvalid || st_begin + 1 < nsuper) {
#pragma unroll
            for (int h = 0; h < H; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) aload1(cin[t][h][j], Cin + ((int64_t)st_begin * NTT + h * 16 + j) * ldc_in, cvoff_in[t]);
        }
    }
    if constexpr (!DMA) store_panel();
    // (a use of the row registers HERE makes the compiler wait for their loads before the loop; otherwise its wait
    // bookkeeping carries them into the loop as "possibly pending" and every batch waits for younger loads)
#pragma unroll
    for (int t = 0; t < SETS; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(av[t][b]));
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing the compiler tracks is outstanding from here on
    if constexpr (TIMED) t2 = clock64();                // second round trip done: row entries, first panel (and C_in)
    long long t_rows = 0;
    for (int st = st_begin; st < st_end; ++st) {
        const int64_t col0 = (int64_t)st * NTT;
        // ---- the panel of this tile (LDS-DMA form), then the requests that fly under its row loop: its C_in, the next panel
        // (register form).  C_in is requested AFTER the panel has landed: the compiler drains everything outstanding before the
        // barrier that publishes the panel, and C_in (always an HBM miss) in front of that drain added its latency to the panel's
        // (mostly L2 hits) at the top of every tile; behind it, it has the whole row loop to arrive.
        if constexpr (DMA) {
            if (st != st_begin) {
                __syncthreads();               // every wave is done reading the previous panel
                dma_panel(st);
                __syncthreads();               // (the compiler drains the DMA before the barrier)
            }
        }
        if (st != st_begin) {
#pragma unroll
            for (int t = 0; t < SETS; ++t) {
                if constexpr (CROW) {
                    if (!RM || cvalid || st + 1 < nsuper) {
                        if constexpr (RM == 2) aload4p(cinv[t], ptr_in(t) + (int64_t)st * 16);
                        else aload4(cinv[t], Cin + (int64_t)st * ct_in, off_rin(t));
                    }
                } else if (cvalid || st + 1 < nsuper) {
#pragma unroll
                    for (int h = 0; h < H; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j) aload1(cin[t][h][j], Cin + (col0 + h * 16 + j) * ldc_in, cvoff_in[t]);
                }
            }
        }
        if constexpr (!DMA) {
            if (st + 1 < st_end) load_panel(st + 1, false);
        }
        f32x4 acc[SETS][H];
#pragma unroll
        for (int t = 0; t < SETS; ++t) {
#pragma unroll
        for (int h = 0; h < H; ++h) acc[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
        // a batch whose 16 entries are live for every lane of the wavefront runs without predicates
#define SX_RBATCH(b)                                                                                  \
        if constexpr ((b) < NB) {                                                                      \
        if (__builtin_amdgcn_ballot_w64(len[t] >= ((b) + 1) * BATCH) == __builtin_amdgcn_ballot_w64(true)) { \
            float vb[4] = {av[t][b].x, av[t][b].y, av[t][b].z, av[t][b].w};                            \
            wide_batch<H, EXACT>(ai[t][b], vb, pq[t], acc[t]);                                         \
        } else if (len[t] > (b) * BATCH) {                                                            \
            float vb[4] = {av[t][b].x, av[t][b].y, av[t][b].z, av[t][b].w};                            \
            const int cnt = len[t] - (b) * BATCH;                                                      \
            wide_quad<0, H, EXACT>(ai[t][b], vb, pq[t], acc[t]); pin<H>(acc[t]);                       \
            if (cnt > 4) { wide_quad<1, H, EXACT>(ai[t][b], vb, pq[t], acc[t]); pin<H>(acc[t]); }      \
            if (cnt > 8) { wide_quad<2, H, EXACT>(ai[t][b], vb, pq[t], acc[t]); pin<H>(acc[t]); }      \
            if (cnt > 12) { wide_quad<3, H, EXACT>(ai[t][b], vb, pq[t], acc[t]); pin<H>(acc[t]); }     \
        }                                                                                              \
        }
        SX_RBATCH(0) SX_RBATCH(1) SX_RBATCH(2) SX_RBATCH(3) SX_RBATCH(4) SX_RBATCH(5)
#undef SX_RBATCH
        for (int pos = NB * BATCH; pos < len[t]; pos += BATCH) {   // rows longer than NB batches: the rest from the stream
            const f32x4 v = *reinterpret_cast<const f32x4 *>(pv + loff[t] + pos);
            const uint2 w = *reinterpret_cast<const uint2 *>(pi + loffi[t] + pos);
            int ix[4] = {(int)(w.x & 0xffffu), (int)(w.x >> 16), (int)(w.y & 0xffffu), (int)(w.y >> 16)};
            float vx[4] = {v.x, v.y, v.z, v.w};
            const int cnt = len[t] - pos;
            wide_quad<0, H, EXACT>(ix, vx, pq[t], acc[t]); pin<H>(acc[t]);
            if (cnt > 4) { wide_quad<1, H, EXACT>(ix, vx, pq[t], acc[t]); pin<H>(acc[t]); }
            if (cnt > 8) { wide_quad<2, H, EXACT>(ix, vx, pq[t], acc[t]); pin<H>(acc[t]); }
            if (cnt > 12) { wide_quad<3, H, EXACT>(ix, vx, pq[t], acc[t]); pin<H>(acc[t]); }
        }
        }

        if constexpr (TIMED) {
            asm volatile("" : "+v"(acc[0][0]) : : "memory");
            t_rows += clock64() - (st == st_begin ? t2 : t3);
        }
        // ---- drain: C_in and the next panel have landed (and the previous super tile's stores are acknowledged)
        if constexpr (H == 2) {
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(cin[0][0][0]), "+v"(cin[0][0][1]), "+v"(cin[0][0][2]), "+v"(cin[0][0][3]), "+v"(cin[0][1][0]), "+v"(cin[0][1][1]),
                           "+v"(cin[0][1][2]), "+v"(cin[0][1][3])
                         :
                         : "memory");
        } else if constexpr (CROW && SETS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(cinv[0]), "+v"(cinv[SETS - 1]) : : "memory");
        } else if constexpr (CROW) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(cinv[0]) : : "memory");
        } else if constexpr (SETS == 2) {
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(cin[0][0][0]), "+v"(cin[0][0][1]), "+v"(cin[0][0][2]), "+v"(cin[0][0][3]), "+v"(cin[SETS - 1][0][0]),
                           "+v"(cin[SETS - 1][0][1]), "+v"(cin[SETS - 1][0][2]), "+v"(cin[SETS - 1][0][3])
                         :
                         : "memory");
        } else {
            static_assert((H == 1 || H == 2) && SETS <= 2, "the drain names the C_in registers explicitly");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(cin[0][0][0]), "+v"(cin[0][0][1]), "+v"(cin[0][0][2]), "+v"(cin[0][0][3]) : : "memory");
        }
        // The same wait also covers the NEXT panel's B rows (load_panel(st + 1, false) wrote bv / bs behind the compiler's back): name
        // every one of those registers in a volatile asm right behind the wait (volatile asms keep their order), so that no use of
        // them -- e.g. the copies that pack bs[u][h][0..3] into the 16-byte operand of store_panel -- can be scheduled above it.
        if constexpr (!DMA) {
#pragma unroll
            for (int u = 0; u < MAXD; ++u)
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    if constexpr (BCOL) asm volatile("" : "+v"(bs[u][h][0]), "+v"(bs[u][h][1]), "+v"(bs[u][h][2]), "+v"(bs[u][h][3]));
                    else asm volatile("" : "+v"(bv[u][h]));
                }
        }
        // ---- C straight from the accumulators
#pragma unroll
        for (int t = 0; t < SETS; ++t) {
            if constexpr (CROW) {
                if (cwrite[t] && (!RM || cvalid || st + 1 < nsuper)) {
                    const f32x4 o = {epilogue<EXACT>(alpha, acc[t][0].x, beta, cinv[t].x), epilogue<EXACT>(alpha, acc[t][0].y, beta, cinv[t].y),
                                     epilogue<EXACT>(alpha, acc[t][0].z, beta, cinv[t].z), epilogue<EXACT>(alpha, acc[t][0].w, beta, cinv[t].w)};
                    if constexpr (RM == 2) astore4p(ptr_out(t) + (int64_t)st * 16, o);
                    else astore4(Cout + (int64_t)st * ct_out, off_rout(t), o);
                }
            } else if (cwrite[t] && (cvalid || st + 1 < nsuper)) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const float a4[4] = {acc[t][h].x, acc[t][h].y, acc[t][h].z, acc[t][h].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        astore1(Cout + (col0 + h * 16 + j) * ldc, cvoff_out[t], epilogue<EXACT>(alpha, a4[j], beta, cin[t][h][j]));
                }
            }
        }
        if constexpr (!DMA) {
            if (st + 1 < st_end) {
                __syncthreads();               // every wave is done reading this super tile's panel
                store_panel();
                __syncthreads();
            }
        }
        if constexpr (TIMED) t3 = clock64();
    }
    if constexpr (TIMED) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last tile's C stores are acknowledged
        const long long t4 = clock64();
        if (dbg && tid == 0 && (blockIdx.x & 15) == 5) {
            atomicAdd((unsigned long long *)&dbg[0], (unsigned long long)(t1 - t0));            // kernel arguments + block meta, extents, dictionary
            atomicAdd((unsigned long long *)&dbg[1], (unsigned long long)(t2 - t1));            // row entries + first panel + barrier
            atomicAdd((unsigned long long *)&dbg[2], (unsigned long long)t_rows);               // row loops of all tiles
            atomicAdd((unsigned long long *)&dbg[3], (unsigned long long)(t4 - t2 - t_rows));   // drains, C stores, panel turnover
            atomicAdd((unsigned long long *)&dbg[4], 1ull);
            atomicAdd((unsigned long long *)&dbg[5], (unsigned long long)(wall_clock64() - w0));
        }
    }
}

}  // namespace sx
