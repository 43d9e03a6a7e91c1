// spmm_csr_kernels.h -- CDNA4 (gfx950) kernels that replace the TAPA/HLS processing elements of
// the reference accelerator (src/sextans.cpp:836-984, task inventory in SURVEY.md 2.2).
//
// Mapping of the reference's dataflow onto a 64-lane wavefront machine:
//
//   reference (FPGA)                                   here (MI355X)
//   -----------------------------------------------    ------------------------------------------
//   read_A: packed non-zero stream, re-read per         CSR col_idx/val streamed with coalesced
//     8-column N tile (sextans.cpp:75-100)                global loads into LDS, once per N tile
//   read_B + PEG_Bmtx local_B window                    B repacked to row-major N-tile panels
//     (sextans.cpp:102-126, 337, 353-381)                 (repack_b_panels), gathered with one
//                                                         16-byte load per lane per non-zero
//   PEcore_Bmtx: val * B[col][0..7] (:285-295)          per-lane float4 multiply
//   PEG_Cmtx URAM accumulators, row%64 -> PE            per-row register accumulators: LPR lanes
//     (sextans.cpp:425-460, sparse_helper.h:370)          own one row, each lane 4 columns
//   FloatvMultConst x2 + FloatvAddFloatv                fused epilogue alpha*acc + beta*c_in,
//     (sextans.cpp:196-233), read_C, write_C              C tile transposed through LDS so the
//                                                         column-major loads/stores coalesce
//
// Arithmetic order (parity): a row's products are formed and added in ascending CSR order by ONE
// lane per output element, each product rounded to fp32 before the add when EXACT (the order and
// rounding of cpu_spmm_CSR, sparse_helper.h:279-289, which SURVEY.md 3.3 shows is also the
// accelerator's).  The TU is compiled with -ffp-contract=off; the non-exact variant calls fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sx {

constexpr int kBlock = 256;   // 4 wavefronts
typedef float f32x4 __attribute__((ext_vector_type(4)));   // plain vector: stays in VGPRs when staged

// Bijective XCD-aware remap of a linear workgroup id: the dispatcher places workgroup b on XCD
// b % 8, so giving each XCD a contiguous chunk of logical ids keeps neighbouring row blocks
// (which share B rows in banded matrices, and the A stream across N tiles) on one L2.  Speed
// only -- correctness never depends on placement.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned xcd = b & 7u, i = b >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

template <bool EXACT>
__device__ __forceinline__ float mac(float acc, float a, float b) {
    if constexpr (EXACT) {
        const float p = a * b;   // rounded product (-ffp-contract=off), then rounded add
        return acc + p;
    } else {
        return __builtin_fmaf(a, b, acc);
    }
}

template <bool EXACT>
__device__ __forceinline__ void mac4(float4 &acc, float a, const float4 &b) {
    acc.x = mac<EXACT>(acc.x, a, b.x);
    acc.y = mac<EXACT>(acc.y, a, b.y);
    acc.z = mac<EXACT>(acc.z, a, b.z);
    acc.w = mac<EXACT>(acc.w, a, b.w);
}

template <bool EXACT>
__device__ __forceinline__ float epilogue(float alpha, float acc, float beta, float cin) {
    if constexpr (EXACT) {
        const float t0 = alpha * acc;   // FloatvMultConst (sextans.cpp:196-216): separate multiplies
        const float t1 = beta * cin;
        return t0 + t1;                 // FloatvAddFloatv (sextans.cpp:218-233)
    } else {
        return __builtin_fmaf(alpha, acc, beta * cin);
    }
}

// ------------------------------------------------------------------------------------------------
// Row-group gather kernel.
//   LPR   lanes per row; each lane owns 4 consecutive output columns -> N tile NT = 4*LPR.
//   CH    non-zeros staged in LDS per chunk (8 bytes each).
//   Each workgroup owns RB = 256/LPR consecutive rows and one N tile.  Linear grid of
//   nrowblk * ntiles workgroups; logical id -> (rowblk = id / ntiles, tile = id % ntiles).
// Bp: panels, panel t is a row-major K x NT matrix at Bp + t*panel_stride.
// Cin/Cout: column-major, already offset to the first column of tile 0 of this launch.
// RM (round 5, sextans_spmm_device_rm): the caller's ROW-major operands -- Bp = B at the launch's first column, panel_stride = its
// leading dimension (B row c of tile t = the NT floats at Bp + c * ldb + t * NT: what a panel row is, without the repack pass), C rows
// of ldc_in / ldc floats: a lane's 4 accumulators are 16 consecutive bytes of its C row, so the tile goes out (and C_in comes in)
// with one 16-byte access per lane and the LDS transpose is not needed.
// ------------------------------------------------------------------------------------------------
template <int LPR, int CH, bool EXACT, bool STAGE, bool RM = false>
__global__ __launch_bounds__(kBlock) void spmm_csr_rowgroup(
    const int *__restrict__ row_ptr, const int *__restrict__ row_end, const int *__restrict__ col_idx,
    const float *__restrict__ val, const float *__restrict__ Bp, int64_t panel_stride, const float *Cin,
    int64_t ldc_in, float *Cout, int64_t ldc, int row_begin, int M, int ntiles, int nrowblk, float alpha, float beta,
    int use_xcd_remap, const unsigned char *__restrict__ skip, const int *__restrict__ groups) {
    // groups (may be null; the split form of a mixed plan): the launch covers only the listed groups of 128 rows (those that hold a row
    // of this kernel's) -- workgroup id / ntiles walks the list, 128 / RB workgroups per group.
    // skip (may be null): rows whose C is produced by the piece path (long rows taken out of the main matrix).
    // Row r holds entries [row_ptr[r], row_end[r]): row_end = row_ptr + 1 for a CSR matrix; the pieces of hub rows
    // (long-row splitting) come with their own end array and are not contiguous from one piece to the next.
    // Rows [row_begin, M) are processed; C pointers address row_begin as their row 0 (row-range calls
    // of the multi-GPU pipeline write a packed slab chunk); row_ptr is indexed with the global row.
    constexpr int NT = 4 * LPR;
    constexpr int RB = kBlock / LPR;
    constexpr int TS = RB + 1;   // padded row stride of the transposed C tile in LDS
    static_assert(CH * 2 >= NT * TS, "chunk buffer must also hold the C tile");
    __shared__ __attribute__((aligned(16))) int smem[CH * 2];

    const unsigned nwg = (unsigned)nrowblk * (unsigned)ntiles;
    unsigned wg = blockIdx.x;
    if (use_xcd_remap) wg = xcd_remap(wg, nwg);
    int rowblk = (int)(wg / (unsigned)ntiles);
    const int tile = (int)(wg % (unsigned)ntiles);
    if (groups) {
        constexpr int PER = (128 / RB) > 0 ? 128 / RB : 1;
        rowblk = groups[rowblk / PER] * PER + rowblk % PER;
    }

    const int tid = threadIdx.x;
    const int slot = tid / LPR;
    const int q = tid % LPR;
    const int row0 = row_begin + rowblk * RB;
    const int row = row0 + slot;

    const float *bq = RM ? Bp + (int64_t)tile * NT + 4 * q : Bp + (int64_t)tile * panel_stride + 4 * q;
    auto brow = [&](int c) -> const float4 * {   // my 16 bytes of B row c
        if constexpr (RM) return reinterpret_cast<const float4 *>(bq + (int64_t)c * panel_stride);
        else return reinterpret_cast<const float4 *>(bq + (int64_t)c * NT);
    };
    int j = 0, jend = 0;
    // rows named by `skip` are somebody else's: piece-path rows are empty in the main matrix anyway; in the split form of a MIXED plan
    // they are the rows of the dictionary blocks, with all their entries -- not walked here, and a workgroup without a row of its own
    // leaves before it stages anything
    const bool mine = row < M && !(skip && skip[row]);
    if (skip && __syncthreads_count(mine) == 0) return;
    if (mine) { j = row_ptr[row]; jend = row_end[row]; }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cin4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool cwrite = RM && mine;
    if constexpr (RM)   // C_in early: in flight under the row loop
        if (cwrite) cin4 = *reinterpret_cast<const float4 *>(Cin + (int64_t)(row - row_begin) * ldc_in + tile * NT + 4 * q);

    if constexpr (STAGE) {
        int2 *s_nz = reinterpret_cast<int2 *>(smem);
        const int bs = row_ptr[row0];
        const int be = row_ptr[min(row0 + RB, M)];
        for (int cs = bs; cs < be; cs += CH) {
            const int n = min(CH, be - cs);
            for (int i = tid; i < n; i += kBlock)
                s_nz[i] = make_int2(col_idx[cs + i], __float_as_int(val[cs + i]));
            __syncthreads();
            const int hi = min(jend, cs + n);
            while (j + 4 <= hi) {
                const int2 e0 = s_nz[j - cs], e1 = s_nz[j - cs + 1], e2 = s_nz[j - cs + 2],
                           e3 = s_nz[j - cs + 3];
                const float4 b0 = *brow(e0.x);
                const float4 b1 = *brow(e1.x);
                const float4 b2 = *brow(e2.x);
                const float4 b3 = *brow(e3.x);
                mac4<EXACT>(acc, __int_as_float(e0.y), b0);
                mac4<EXACT>(acc, __int_as_float(e1.y), b1);
                mac4<EXACT>(acc, __int_as_float(e2.y), b2);
                mac4<EXACT>(acc, __int_as_float(e3.y), b3);
                j += 4;
            }
            while (j < hi) {
                const int2 e = s_nz[j - cs];
                const float4 b = *brow(e.x);
                mac4<EXACT>(acc, __int_as_float(e.y), b);
                ++j;
            }
            __syncthreads();
        }
    } else {
        // Direct variant: the LPR lanes of a row fetch LPR consecutive non-zeros with one load each
        // and exchange them with wave shuffles (no LDS staging, no workgroup barriers).
        while (j < jend) {
            const int mine = j + q;
            int c = 0;
            float a = 0.f;
            if (mine < jend) { c = col_idx[mine]; a = val[mine]; }
            const int cnt = min(LPR, jend - j);
            float4 b[LPR];
#pragma unroll
            for (int u = 0; u < LPR; ++u) {
                const int cu = __shfl(c, u, LPR);
                b[u] = (u < cnt) ? *brow(cu)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < LPR; ++u) {
                const float au = __shfl(a, u, LPR);
                if (u < cnt) mac4<EXACT>(acc, au, b[u]);
            }
            j += LPR;
        }
    }

    if constexpr (RM) {
        if (cwrite)
            *reinterpret_cast<float4 *>(Cout + (int64_t)(row - row_begin) * ldc + tile * NT + 4 * q) =
                make_float4(epilogue<EXACT>(alpha, acc.x, beta, cin4.x), epilogue<EXACT>(alpha, acc.y, beta, cin4.y),
                            epilogue<EXACT>(alpha, acc.z, beta, cin4.z), epilogue<EXACT>(alpha, acc.w, beta, cin4.w));
        return;
    }
    // Transpose the RB x NT tile through LDS so that column-major C traffic is coalesced.
    float *s_c = reinterpret_cast<float *>(smem);
    s_c[(4 * q + 0) * TS + slot] = acc.x;
    s_c[(4 * q + 1) * TS + slot] = acc.y;
    s_c[(4 * q + 2) * TS + slot] = acc.z;
    s_c[(4 * q + 3) * TS + slot] = acc.w;
    __syncthreads();
    const int64_t col0 = (int64_t)tile * NT;
#pragma unroll
    for (int i = 0; i < (RB * NT) / kBlock; ++i) {
        const int e = tid + i * kBlock;
        const int n = e / RB, r = e % RB;
        const int orow = row0 + r;
        if (orow < M && !(skip && skip[orow])) {
            const int64_t lr = (int64_t)(orow - row_begin);
            Cout[lr + (col0 + n) * ldc] = epilogue<EXACT>(alpha, s_c[n * TS + r], beta, Cin[lr + (col0 + n) * ldc_in]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-panel kernel ("the dense B panel staged into LDS"): the CDNA4 form of the reference's on-chip
// B window (PEG_Bmtx local_B, sextans.cpp:337,353-381) fed by a packed non-zero stream with
// window-local column indices (edge_list_64bit, sparse_helper.h:419-443).  Input is the
// row-bucketed packed form of A built by panel_plan.cpp:
//   dictionary block: the block's distinct B rows are copied ONCE from the row-major B panel in
//     global memory into LDS (64-byte rows for N-tile 16, ascending column order so the copy reads
//     whole 128-byte lines); every non-zero then reads its B row from LDS through a 16-bit local
//     index -- 6 bytes of A stream per non-zero instead of 8 and no per-non-zero L1/L2 gather;
//   direct block (no reuse / too many distinct columns): 32-bit columns, B rows gathered from
//     global memory as in spmm_csr_rowgroup.
// The A stream is NOT staged through LDS: every row starts on a 4-entry boundary, each of the LPR
// lanes of a row fetches 4 entries with one 8-byte (indices) + one 16-byte (values) load, and the
// entries are handed round the row group with DPP quad broadcasts (no LDS, no barrier).  LDS holds
// only the B panel and the C tile, so ~4 workgroups fit per CU.
// Arithmetic and per-row order are those of spmm_csr_rowgroup: bit-identical to cpu_spmm_CSR.
// ------------------------------------------------------------------------------------------------
template <int LPR, int S>
__device__ __forceinline__ int slot_bcast(int x) {
    static_assert(S >= 0 && S < LPR, "source lane inside the row group");
    if constexpr (LPR == 4) {
        return __builtin_amdgcn_update_dpp(0, x, S * 0x55, 0xF, 0xF, true);   // quad_perm [S,S,S,S]
    } else if constexpr (LPR == 2) {
        return __builtin_amdgcn_update_dpp(0, x, S | (S << 2) | ((2 + S) << 4) | ((2 + S) << 6), 0xF, 0xF, true);
    } else {
        return __shfl(x, S, LPR);
    }
}

// MIXED = the plan also has "direct" blocks (no dictionary: 32-bit columns, B rows gathered from global
// memory).  Matrices whose every block has reuse run the MIXED = false instantiation: no second code path,
// so no control-flow joins at which the waitcnt insertion has to drain all outstanding loads.
// BCOL (dictionary-only): Bp is the caller's COLUMN-MAJOR B (panel_stride = its leading dimension) and the
// panel is staged from it with four 4-byte loads per B row and lane instead of one 16-byte load from the
// repacked panel.  For matrices whose B fits the L2s this saves the repack launch, which is a third of a
// step when the whole SpMM is a few microseconds (nasa4704).
template <int LPR, bool EXACT, bool MIXED, bool BCOL = false>
#ifndef SX_PANEL_MIXED_WGS
#define SX_PANEL_MIXED_WGS 4
#endif
__global__ __launch_bounds__(kBlock, (LPR == 8 ? 2 : (MIXED && LPR == 4 ? SX_PANEL_MIXED_WGS : 4))) void spmm_csr_panel(
    const int2 *__restrict__ slot_info, const unsigned short *__restrict__ p_idx16,
    const int *__restrict__ p_col32, const float *__restrict__ p_val, const int *__restrict__ blk_row,
    const int *__restrict__ dict_cnt, const int *__restrict__ blk_dict, int dict_stride,
    const float *__restrict__ Bp, int64_t panel_stride, const float *Cin, int64_t ldc_in, float *Cout,
    int64_t ldc, int ntiles, int nblk, float alpha, float beta, int use_xcd_remap, int panel_floats,
    long long *dbg, int blk_begin, int row_base, const unsigned char *__restrict__ skip, const int2 *__restrict__ slot_ioff, int pad_rows) {
    // slot_ioff (may be null): {start of the slot's index list in p_idx16, shift in bytes added to every offset} when lists of consecutive
    // rows that are equal up to a shift are stored once; pad_rows: +1.0f rows at the end of the panel (1, or kPlanPadRows with shared lists).
    // Blocks [blk_begin, blk_begin + nblk) of the plan are processed (row-range calls of the multi-GPU pipeline
    // cut at block boundaries); the C pointers address row `row_base` as their row 0.
    const long long t0 = dbg ? clock64() : 0;   // dbg: optional phase timing (engine option "phase_timing")
    const long long w0 = dbg ? wall_clock64() : 0;   // 100 MHz constant clock: calibrates the cycle counts
    static_assert(!(MIXED && BCOL), "the column-major staging exists for dictionary-only plans");
    constexpr int NT = 4 * LPR;
    constexpr int RB = kBlock / LPR;
    constexpr int TS = RB + 1;
    constexpr int BATCH = 4 * LPR;            // entries a row group holds per fetch (4 per lane)
    constexpr int OPT = (RB * NT) / kBlock;   // outputs per thread
    // Dynamic LDS: max(panel_floats, NT*TS) floats.  The B panel lives there while rows are streamed; the
    // C tile of the epilogue reuses the same bytes after a barrier (36 KiB per workgroup => 4 per CU;
    // a separate 4 KiB tile made it 41 024 bytes, 64 too many for the fourth workgroup).
    extern __shared__ __attribute__((aligned(16))) int smem[];
    float *panel = reinterpret_cast<float *>(smem);
    float *s_c = reinterpret_cast<float *>(smem);

    const unsigned nwg = (unsigned)nblk * (unsigned)ntiles;
    unsigned wg = blockIdx.x;
    if (use_xcd_remap) wg = xcd_remap(wg, nwg);
    const int blk = blk_begin + (int)(wg / (unsigned)ntiles);
    const int tile = (int)(wg % (unsigned)ntiles);

    const int tid = threadIdx.x;
    const int slot = tid / LPR;
    const int q = tid % LPR;
    // Everything the block needs first sits at addresses that depend on the block NUMBER only (fixed
    // strides per block), so the scalar block meta, this slot's row extent and its share of the
    // dictionary are all requested in the same round trip; the entries and the B rows follow in the
    // second one.  Slots past the block's last row read {0, 0}; dictionary slots past its last entry
    // repeat the last column (they rewrite the same panel bytes).
    constexpr int MAXD = 9;   // dictionary capacity = MAXD * RB entries (36 KiB panel)
    const int row0 = blk_row[blk];
    const int row1 = blk_row[blk + 1];        // row1 - row0 <= RB
    const int nu = dict_cnt[blk];
    const bool use_dict = MIXED ? nu > 0 : true;
    const int2 si = slot_info[(int64_t)blk * RB + slot];      // {first packed entry, entries} of this slot's row
    int dix[MAXD];
    {
        // dict_stride is a multiple of RB: the clamp is on the uniform part of the address only (branch-free:
        // all loads stay in one basic block and in flight together)
        const int *bd = blk_dict + (int64_t)blk * dict_stride + slot;
#pragma unroll
        for (int u = 0; u < MAXD; ++u) dix[u] = bd[min(u * RB, dict_stride - RB)];
    }

    const float *bq = Bp + (int64_t)tile * panel_stride + 4 * q;
    const int len = si.y;
    const int64_t off = si.x;
    const int2 io2 = slot_ioff ? slot_ioff[(int64_t)blk * RB + slot] : make_int2(si.x, 0);
    const int64_t offi = io2.x;

    // this lane's 4 entries of a batch: indices (unpacked to int) and values
    auto fetch = [&](int pos, int (&oi)[4], float (&ov)[4]) {
        const int64_t o = off + pos + 4 * q;   // stream is padded: reads past the row stay in bounds
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p_val + o);
        ov[0] = v.x; ov[1] = v.y; ov[2] = v.z; ov[3] = v.w;
        if (use_dict) {
            // dictionary entries carry the BYTE offset of their B row in the panel (index * NT * 4 < 64 Ki)
            const uint2 w = *reinterpret_cast<const uint2 *>(p_idx16 + offi + pos + 4 * q);
            oi[0] = (int)(w.x & 0xffffu); oi[1] = (int)(w.x >> 16);
            oi[2] = (int)(w.y & 0xffffu); oi[3] = (int)(w.y >> 16);
        } else {
            const int4 c = *reinterpret_cast<const int4 *>(p_col32 + o);
            oi[0] = c.x; oi[1] = c.y; oi[2] = c.z; oi[3] = c.w;
        }
    };
    int ix[4], nx[4];
    float vx[4], nv[4];
    // Dictionary-only instantiation: the B-row loads of the panel (mostly L2 hits) are issued BEFORE the
    // row-stream loads (always HBM): vmcnt retires in order, so the LDS writes of the panel would otherwise
    // wait for the slower stream loads queued ahead of them.
    constexpr int MAXD_ = 9;
    f32x4 bv[MIXED ? 1 : MAXD_];
    if constexpr (!MIXED) {
#pragma unroll
        for (int u = 0; u < MAXD_; ++u) {
            if constexpr (BCOL) {
                const float *bc = Bp + ((int64_t)tile * NT + 4 * q) * panel_stride + dix[u];
                bv[u].x = bc[0]; bv[u].y = bc[panel_stride]; bv[u].z = bc[2 * panel_stride];
                bv[u].w = bc[3 * panel_stride];
            } else {
                bv[u] = *reinterpret_cast<const f32x4 *>(bq + (int64_t)dix[u] * NT);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    fetch(0, ix, vx);
    // Dictionary-only instantiation: PFT further batches of the row stay in flight as raw loads (6
    // registers each); see the row loop below.
    constexpr int PFT = MIXED ? 0 : 3;
    f32x4 rv[PFT > 0 ? PFT : 1];
    uint2 rw[PFT > 0 ? PFT : 1];
    auto fetch_raw = [&](int pos, f32x4 &v, uint2 &w) {
        const int64_t o = off + pos + 4 * q;
        v = *reinterpret_cast<const f32x4 *>(p_val + o);
        w = *reinterpret_cast<const uint2 *>(p_idx16 + offi + pos + 4 * q);
    };
    auto unpack_raw = [&](const f32x4 &v, const uint2 &w) {
        vx[0] = v.x; vx[1] = v.y; vx[2] = v.z; vx[3] = v.w;
        ix[0] = (int)(w.x & 0xffffu); ix[1] = (int)(w.x >> 16);
        ix[2] = (int)(w.y & 0xffffu); ix[3] = (int)(w.y >> 16);
    };
    if constexpr (PFT > 0) {
#pragma unroll
        for (int d = 0; d < PFT; ++d) fetch_raw((d + 1) * BATCH, rv[d], rw[d]);
    }

    // C_in for this thread's outputs: issued now, consumed in the epilogue (clamped addresses).
    const int64_t col0 = (int64_t)tile * NT;
    float cin[OPT];
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
        const int e = tid + i * kBlock;
        const int n = e / RB, r = e % RB;
        cin[i] = Cin[(int64_t)(min(row0 + r, row1 - 1) - row_base) + (col0 + n) * ldc_in];
    }

    const long long t1 = dbg ? clock64() : 0;
    if constexpr (!MIXED) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < MAXD_; ++u)
            *reinterpret_cast<f32x4 *>(panel + max(min(slot + min(u * RB, dict_stride - RB), nu - 1), 0) * NT + 4 * q) =
                bv[u];
        for (int i = tid; i < pad_rows * NT; i += kBlock) panel[panel_floats - pad_rows * NT + i] = 1.0f;   // the rows padding entries (value -0.0f) point at
        __syncthreads();
    } else if (use_dict) {
        // Stage the block's distinct B rows: slot s copies dictionary entries s, s+RB, ... (indices were
        // requested at kernel entry; chunks past the stride repeat the last chunk and entries past the
        // dictionary repeat its last column, so duplicates rewrite the same bytes).
#pragma unroll
        for (int h0 = 0; h0 < MAXD; h0 += 5) {
            f32x4 v[5];
#pragma unroll
            for (int u = 0; u < 5; ++u)
                if (h0 + u < MAXD) v[u] = *reinterpret_cast<const f32x4 *>(bq + (int64_t)dix[h0 + u] * NT);
#pragma unroll
            for (int u = 0; u < 5; ++u)
                if (h0 + u < MAXD)
                    *reinterpret_cast<f32x4 *>(
                        panel + max(min(slot + min((h0 + u) * RB, dict_stride - RB), nu - 1), 0) * NT + 4 * q) = v[u];
        }
        for (int i = tid; i < pad_rows * NT; i += kBlock) panel[panel_floats - pad_rows * NT + i] = 1.0f;   // the rows padding entries (value -0.0f) point at
        __syncthreads();
    }

    const long long t2 = dbg ? clock64() : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *pq = reinterpret_cast<const float *>(reinterpret_cast<const char *>(panel + 4 * q) + io2.y);

// One sub-batch = 8 consecutive entries starting at entry `base` of the current batch: broadcast the
// entries held by lane (k/4) of the row group, issue all B-row reads (BROW: LDS panel or global
// gather -- kept as two separate code paths so the LDS reads stay ds_read_b128, not flat loads), then
// accumulate in order.
#define SX_BROW_LDS(idx) (*reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(pq) + (idx)))
#define SX_BROW_GLB(idx) (*reinterpret_cast<const float4 *>(bq + (int64_t)(idx) * NT))
#define SX_STEP_DECL(k) int i##k = 0; float a##k = 0.f; float4 b##k = make_float4(0.f, 0.f, 0.f, 0.f);
#define SX_STEP_LOAD(BROW, k, base, live)                                                            \
    if constexpr ((base) + (k) < BATCH) {                                                            \
        i##k = slot_bcast<LPR, ((base) + (k)) / 4>(ix[(k) % 4]);                                     \
        a##k = __int_as_float(slot_bcast<LPR, ((base) + (k)) / 4>(__float_as_int(vx[(k) % 4])));     \
        if (live) b##k = BROW(i##k);                                                                 \
    }
#define SX_STEP_MAC(k, base, live) \
    if constexpr ((base) + (k) < BATCH) { if (live) mac4<EXACT>(acc, a##k, b##k); }
#define SX_SUB(BROW, base, cnt)                                                                          \
    {                                                                                                    \
        SX_STEP_DECL(0) SX_STEP_DECL(1) SX_STEP_DECL(2) SX_STEP_DECL(3) SX_STEP_DECL(4) SX_STEP_DECL(5)  \
        SX_STEP_DECL(6) SX_STEP_DECL(7)                                                                  \
        SX_STEP_LOAD(BROW, 0, base, (base) + 0 < (cnt)) SX_STEP_LOAD(BROW, 1, base, (base) + 1 < (cnt)) \
        SX_STEP_LOAD(BROW, 2, base, (base) + 2 < (cnt)) SX_STEP_LOAD(BROW, 3, base, (base) + 3 < (cnt)) \
        SX_STEP_LOAD(BROW, 4, base, (base) + 4 < (cnt)) SX_STEP_LOAD(BROW, 5, base, (base) + 5 < (cnt)) \
        SX_STEP_LOAD(BROW, 6, base, (base) + 6 < (cnt)) SX_STEP_LOAD(BROW, 7, base, (base) + 7 < (cnt)) \
        SX_STEP_MAC(0, base, (base) + 0 < (cnt)) SX_STEP_MAC(1, base, (base) + 1 < (cnt))               \
        SX_STEP_MAC(2, base, (base) + 2 < (cnt)) SX_STEP_MAC(3, base, (base) + 3 < (cnt))               \
        SX_STEP_MAC(4, base, (base) + 4 < (cnt)) SX_STEP_MAC(5, base, (base) + 5 < (cnt))               \
        SX_STEP_MAC(6, base, (base) + 6 < (cnt)) SX_STEP_MAC(7, base, (base) + 7 < (cnt))               \
    }
// Tail of a dictionary row: its remaining entries come in whole groups of 4 (one lane's fetch), the
// padding inside the last group is exact-safe, so the predicate is per group, not per entry.
#define SX_QUAD(BROW, j, cnt)                                                                            \
    if constexpr ((j) < LPR - 1) {                                                                       \
        if (4 * (j) < (cnt)) {                                                                           \
            SX_STEP_DECL(0) SX_STEP_DECL(1) SX_STEP_DECL(2) SX_STEP_DECL(3)                              \
            SX_STEP_LOAD(BROW, 0, 4 * (j), true) SX_STEP_LOAD(BROW, 1, 4 * (j), true)                    \
            SX_STEP_LOAD(BROW, 2, 4 * (j), true) SX_STEP_LOAD(BROW, 3, 4 * (j), true)                    \
            SX_STEP_MAC(0, 4 * (j), true) SX_STEP_MAC(1, 4 * (j), true)                                  \
            SX_STEP_MAC(2, 4 * (j), true) SX_STEP_MAC(3, 4 * (j), true)                                  \
        }                                                                                                \
    }
#define SX_TAIL_ENTRIES(BROW, cnt) \
    { SX_SUB(BROW, 0, cnt) SX_SUB(BROW, 8, cnt) SX_SUB(BROW, 16, cnt) SX_SUB(BROW, 24, cnt) }
#define SX_TAIL_QUADS(BROW, cnt)                                                                  \
    { SX_QUAD(BROW, 0, cnt) SX_QUAD(BROW, 1, cnt) SX_QUAD(BROW, 2, cnt) SX_QUAD(BROW, 3, cnt)     \
      SX_QUAD(BROW, 4, cnt) SX_QUAD(BROW, 5, cnt) SX_QUAD(BROW, 6, cnt) }
#define SX_ROW_LOOP(BROW, TAIL)                                                                \
    {                                                                                          \
        int pos = 0;                                                                           \
        /* full batches: every entry is live, no per-entry predicate */                        \
        while (pos + BATCH <= len) {                                                           \
            fetch(pos + BATCH, nx, nv); /* next batch in flight while this one is consumed */  \
            SX_SUB(BROW, 0, BATCH) SX_SUB(BROW, 8, BATCH) SX_SUB(BROW, 16, BATCH) SX_SUB(BROW, 24, BATCH) \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { ix[e] = nx[e]; vx[e] = nv[e]; }    \
            pos += BATCH;                                                                      \
        }                                                                                      \
        /* tail: fewer than BATCH entries left, predicated per entry */                        \
        if (pos < len) {                                                                       \
            const int cnt = len - pos;                                                         \
            TAIL(BROW, cnt)                                                                    \
        }                                                                                      \
    }
    if constexpr (!MIXED) {
        // Ring of PFT batches ahead.  The steady-state body has no branches, so the compiler can wait for
        // exactly the oldest fetch (vmcnt(2 * (PFT - 1))) instead of draining everything at a join.
        int pos = 0;
        while (pos + PFT * BATCH <= len) {
#pragma unroll
            for (int d = 0; d < PFT; ++d) {
                SX_SUB(SX_BROW_LDS, 0, BATCH) SX_SUB(SX_BROW_LDS, 8, BATCH)
                SX_SUB(SX_BROW_LDS, 16, BATCH) SX_SUB(SX_BROW_LDS, 24, BATCH)
                unpack_raw(rv[d], rw[d]);
                fetch_raw(pos + (PFT + 1) * BATCH, rv[d], rw[d]);
                pos += BATCH;
            }
        }
        // fewer than PFT full batches left: they are already in the ring, nothing new is requested
#pragma unroll
        for (int d = 0; d < PFT - 1; ++d) {
            if (pos + BATCH <= len) {
                SX_SUB(SX_BROW_LDS, 0, BATCH) SX_SUB(SX_BROW_LDS, 8, BATCH)
                SX_SUB(SX_BROW_LDS, 16, BATCH) SX_SUB(SX_BROW_LDS, 24, BATCH)
                unpack_raw(rv[d], rw[d]);
                pos += BATCH;
            }
        }
        if (pos < len) {
            const int cnt = len - pos;
            SX_TAIL_QUADS(SX_BROW_LDS, cnt)
        }
    } else {
        if (use_dict) SX_ROW_LOOP(SX_BROW_LDS, SX_TAIL_QUADS) else SX_ROW_LOOP(SX_BROW_GLB, SX_TAIL_ENTRIES)
    }
#undef SX_ROW_LOOP
#undef SX_TAIL_QUADS
#undef SX_TAIL_ENTRIES
#undef SX_QUAD
#undef SX_SUB
#undef SX_STEP_MAC
#undef SX_STEP_LOAD
#undef SX_STEP_DECL
#undef SX_BROW_GLB
#undef SX_BROW_LDS

    const long long t3 = dbg ? clock64() : 0;
    __syncthreads();                          // every wave is done reading the panel: reuse it as the C tile
    s_c[(4 * q + 0) * TS + slot] = acc.x;
    s_c[(4 * q + 1) * TS + slot] = acc.y;
    s_c[(4 * q + 2) * TS + slot] = acc.z;
    s_c[(4 * q + 3) * TS + slot] = acc.w;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
        const int e = tid + i * kBlock;
        const int n = e / RB, r = e % RB;
        const int orow = row0 + r;
        if (orow < row1 && !(skip && skip[orow])) {   // skip: rows whose C is produced by the piece path
            const int64_t o = (int64_t)(orow - row_base) + (col0 + n) * ldc;
            Cout[o] = epilogue<EXACT>(alpha, s_c[n * TS + r], beta, cin[i]);
        }
    }
    if (dbg && (tid & 63) == 0 && (blockIdx.x & 127) == 5) {   // 1 workgroup in 128: negligible perturbation
        const long long t4 = clock64();
        atomicAdd((unsigned long long *)&dbg[0], (unsigned long long)(t1 - t0));   // meta, row extents, first entries
        atomicAdd((unsigned long long *)&dbg[1], (unsigned long long)(t2 - t1));   // dictionary -> B rows -> LDS, barrier
        atomicAdd((unsigned long long *)&dbg[2], (unsigned long long)(t3 - t2));   // row streaming / compute
        atomicAdd((unsigned long long *)&dbg[3], (unsigned long long)(t4 - t3));   // C tile + epilogue
        atomicAdd((unsigned long long *)&dbg[4], 1ull);
        atomicAdd((unsigned long long *)&dbg[5], (unsigned long long)(wall_clock64() - w0));   // same span, 10 ns ticks
    }
}

// ------------------------------------------------------------------------------------------------
// Piece kernel: one row group (LPR lanes x 4 columns) per piece [vbeg[v], vend[v]) of a long row, raw sums into
// the scratch matrix P (no epilogue).  Pieces arrive sorted by length, so the row groups of a workgroup finish
// together; what matters inside a piece is memory-level parallelism, because the adds of one row are a serial
// chain anyway: a batch is 8 entries (8 / LPR per lane), the 8 B-row gathers of batch k + 1 are issued before the
// multiply-adds of batch k (two register sets), and the entries of batch k + 3 are requested before those of
// batch k + 1 are used (three entry sets; the loop is unrolled over the six phases of the two rotations).
// Order inside a piece = CSR order; one lane per output element (exact).
// ------------------------------------------------------------------------------------------------
template <int LPR, bool EXACT, bool RM = false>   // RM: Bp = the caller's row-major B at the launch's first column, panel_stride = its leading dimension
__global__ __launch_bounds__(kBlock) void spmm_csr_pieces(const int *__restrict__ vbeg, const int *__restrict__ vend,
                                                          const int *__restrict__ col_idx, const float *__restrict__ val,
                                                          const float *__restrict__ Bp, int64_t panel_stride, float *P,
                                                          int64_t ldp, int v_begin, int v_end, int ntiles, const int *__restrict__ colpos) {
    // colpos (may be null): the panels hold row colpos[k] of B at row k -- the permuted panels of the reordered form (reorder_kernels.h)
    constexpr int NT = 4 * LPR;
    constexpr int RB = kBlock / LPR;
    constexpr int E = LPR >= 8 ? 1 : 8 / LPR;   // entries per lane and batch: 8 gathers per batch whatever the tile width
    constexpr int BATCH = E * LPR;
    const int blk = (int)(blockIdx.x / (unsigned)ntiles), tile = (int)(blockIdx.x % (unsigned)ntiles);
    const int tid = threadIdx.x, slot = tid / LPR, q = tid % LPR;
    const int v = v_begin + blk * RB + slot;
    int j = 0, jend = 0;
    if (v < v_end) { j = vbeg[v]; jend = vend[v]; }
    const float *bq = RM ? Bp + (int64_t)tile * NT + 4 * q : Bp + (int64_t)tile * panel_stride + 4 * q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    struct Ent { int c[E]; float a[E]; };
    // this lane's E entries of the batch starting at p (clamped inside the piece: never out of bounds; entries
    // past the end are never multiplied)
    auto fetch = [&](int p, Ent &x) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = max(min(p + E * q + e, jend - 1), 0);
            x.c[e] = colpos ? colpos[col_idx[i]] : col_idx[i];
            x.a[e] = val[i];
        }
    };
    auto gather = [&](const Ent &x, float4 (&b)[BATCH]) {
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int cu = __shfl(x.c[u % E], u / E, LPR);   // entry u of the batch sits in lane u / E, slot u % E
            b[u] = *reinterpret_cast<const float4 *>(bq + (int64_t)cu * (RM ? panel_stride : (int64_t)NT));
        }
    };
    auto macs = [&](const Ent &x, const float4 (&b)[BATCH], int cnt) {
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const float au = __shfl(x.a[u % E], u / E, LPR);
            if (u < cnt) mac4<EXACT>(acc, au, b[u]);
        }
    };
    if (j < jend) {
        Ent e0, e1, e2, t;
        float4 bA[BATCH], bB[BATCH];
        fetch(j, e0); fetch(j + BATCH, e1); fetch(j + 2 * BATCH, e2);
        gather(e0, bA);
        int pos = j;
        // entry sets rotate e0 -> e1 -> e2, row sets bA <-> bB: six phases until both are back where they started
        while (pos < jend) {
            gather(e1, bB); fetch(pos + 3 * BATCH, t); macs(e0, bA, jend - pos); e0 = t; pos += BATCH;
            if (pos >= jend) break;
            gather(e2, bA); fetch(pos + 3 * BATCH, t); macs(e1, bB, jend - pos); e1 = t; pos += BATCH;
            if (pos >= jend) break;
            gather(e0, bB); fetch(pos + 3 * BATCH, t); macs(e2, bA, jend - pos); e2 = t; pos += BATCH;
            if (pos >= jend) break;
            gather(e1, bA); fetch(pos + 3 * BATCH, t); macs(e0, bB, jend - pos); e0 = t; pos += BATCH;
            if (pos >= jend) break;
            gather(e2, bB); fetch(pos + 3 * BATCH, t); macs(e1, bA, jend - pos); e1 = t; pos += BATCH;
            if (pos >= jend) break;
            gather(e0, bA); fetch(pos + 3 * BATCH, t); macs(e2, bB, jend - pos); e2 = t; pos += BATCH;
        }
    }
    if (v < v_end) {
        float *o = P + (int64_t)v + (int64_t)(tile * NT + 4 * q) * ldp;
        o[0] = acc.x; o[ldp] = acc.y; o[2 * ldp] = acc.z; o[3 * ldp] = acc.w;
    }
}

// ------------------------------------------------------------------------------------------------
// Piece path for long rows (engine options "bucket_rows" / "split_rows").  Long rows are taken out of the main
// matrix (the main kernels see them as empty and, told by `skip`, do not write their C); their entries form
// pieces -- the whole row (bucketed rows) or chunks of T entries (hub rows of power-law matrices) -- which the
// row-group kernel sums as virtual rows (alpha = 1, beta = 0) into the scratch matrix P; this kernel folds the
// pieces of each long row IN ORDER and applies the epilogue from C_in.  A one-piece row is bit-identical to
// cpu_spmm_CSR; a split row's sum is re-associated (within the stated 1e-4 bound).  One thread per (row, column).
// ------------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ __launch_bounds__(kBlock) void fold_hub_pieces(const int *__restrict__ vfirst, const int *__restrict__ hub_row,
                                                          const float *__restrict__ P, int64_t ldp, const float *Cin,
                                                          int64_t ldc_in, float *Cout, int64_t ldc, int hub_begin, int nhub,
                                                          int N, int row_base, float alpha, float beta, int c_rm) {
    // c_rm: C row-major (C[r * ldc + n]; consecutive threads then take the columns of one row)
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (int64_t)nhub * N) return;
    const int k = hub_begin + (c_rm ? (int)(t / N) : (int)(t % nhub)), n = c_rm ? (int)(t % N) : (int)(t / nhub);
    const int v0 = vfirst[k], v1 = vfirst[k + 1];
    float acc = P[(int64_t)v0 + n * ldp];
    for (int v = v0 + 1; v < v1; ++v) acc = acc + P[(int64_t)v + n * ldp];
    const int64_t r = (int64_t)(hub_row[k] - row_base);
    if (c_rm) Cout[r * ldc + n] = epilogue<EXACT>(alpha, acc, beta, Cin[r * ldc_in + n]);
    else Cout[r + n * ldc] = epilogue<EXACT>(alpha, acc, beta, Cin[r + n * ldc_in]);
}

// ------------------------------------------------------------------------------------------------
// Exact chains for VERY long rows (hub rows of power-law matrices, border rows of arrow / KKT systems) in strict-order mode.
// The reference's sum of a row is one serial chain of rounded adds (cpu_spmm_CSR, sparse_helper.h:279-289; the PEs
// accumulate one entry at a time too, sextans.cpp:425-446): bit-identical results allow no re-association, so a 400 000-entry
// row cannot be summed in parallel pieces.  What CAN be parallel is everything except the adds: all rounded products
// a_j * B[col_j][n] of the row are formed at full memory-level parallelism (every entry independent) and only the adds walk the
// row in order, one lane per output column: 400 000 entries x ~0.2 ns instead of ~32 ms through the piece kernel, whose serial
// chain also contains the B-row gathers (~80 ns per entry).  Round 3 first did this in two launches through a scratch matrix
// (chain_products + chain_sum: 2.66 ms on the power-law case); chain_fused below does it in one workgroup per row and N tile.
constexpr int kChainCE = 64;   // entries per chunk

// ------------------------------------------------------------------------------------------------
// chain_fused<NT, EXACT>: the exact chain of one very long row x one N tile (NT = 16 / 8 columns) in ONE workgroup, without the
// scratch matrix: wavefront 0 CONSUMES -- lane n adds the rounded products of column n in order (ds_read_b128 = 4 entries, 4
// dependent adds; the one part of the row that has to be serial: 4.5 cycles per entry, tools/valu_bench.hip) -- and the other
// wavefronts PRODUCE them, 64 entries (one chunk) at a time, into an LDS ring.  chain_products + chain_sum above did the same
// in two launches through global memory (the product pass sat in front of the longest row's chain).
// No workgroup barrier in the steady state: a first fused version stepped all wavefronts through one barrier per chunk, and the
// barrier + per-step bookkeeping cost the consumer as much as its 64 adds (700 cycles per chunk, 2.07 ms for a 399 302-entry row;
// tools/chain_bench.hip: 16 reads + 64 dependent adds alone are 440 cycles; this kernel: 520).  Here every producer wavefront is a pipeline of its own:
//   * two groups of GW producer wavefronts take turns, even chunks / odd chunks; inside a chunk a wavefront owns 64 / GW entries
//     (all N-tile columns of them: one task = one entry x 4 columns per lane);
//   * its val / col entries, its B rows and its requests are private (own LDS rings, own vmcnt): stage 1 requests val + col of
//     chunk m + 2D (one dword LDS-DMA, addresses clamped to the row's last entry), stage 2 reads col and requests the B rows of
//     chunk m + D (16-byte LDS-DMA), stage 3 multiplies chunk m into the shared product ring and publishes "m + 1 chunks done";
//   * the consumer reads the eight progress words when it runs out of known-complete chunks and publishes how many chunks it has
//     taken into registers; a producer looks at that word before it overwrites a ring slot.  (LDS operations of a wavefront are
//     executed in order: data before flag on the writing side, flag before data on the reading side.)
// ------------------------------------------------------------------------------------------------
constexpr int kChainD = 7;                                  // producer steps between a request and its use (a step = two chunks of
                                                            // consumer time, ~0.35 us; a random B-row gather ~2.5 us under load)
constexpr int kChainNVC = 16, kChainNBR = 8, kChainNPR = 8; // per-wavefront val/col ring and B-row ring, shared product ring (chunks)
constexpr int chain_fused_group_waves(int NT) { return NT >= 16 ? 4 : 2; }      // = tasks per chunk / 64
constexpr int chain_fused_threads(int NT) { return 64 * (1 + 2 * chain_fused_group_waves(NT)); }
constexpr int chain_fused_lds_bytes(int NT) {
    return 2 * chain_fused_group_waves(NT) * (kChainNVC * 256 + kChainNBR * 1024) + kChainNPR * NT * (kChainCE + 4) * 4 + 64;
}
static_assert(kChainNVC >= 2 * kChainD + 1 && kChainNBR >= kChainD + 1 && kChainNPR >= 4, "ring sizes");

template <int NT, bool EXACT>
__global__ __launch_bounds__(chain_fused_threads(NT)) void chain_fused(
    const int *__restrict__ crow, const int *__restrict__ cbeg, const long long *__restrict__ coff, const int *__restrict__ perm,
    const int *__restrict__ col_idx, const float *__restrict__ val, const float *__restrict__ Bp, int64_t panel_stride, int panel_width, const float *Cin,
    int64_t ldc_in, float *Cout, int64_t ldc, int col0, int ntiles, int k0, int row_base, float alpha, float beta, int c_rm) {
    // c_rm (row-major operands): C[r * ldc + n]; the caller's row-major B is "one panel" whose rows are panel_width = ldb floats apart
    // (NT = 16 or 8 columns per workgroup: a 32-column panel is walked as two 16-column halves -- `ntiles` counts NT-column tiles,
    // `panel_width` = floats per B row of the panel they live in)
    static_assert(NT == 16 || NT == 8, "one task per producer lane and chunk");
    constexpr int CE = kChainCE;                        // 64 entries per chunk
    constexpr int QN = NT / 4;                          // tasks (lanes) per entry: one per 4 columns
    constexpr int GW = chain_fused_group_waves(NT);     // producer wavefronts per group
    constexpr int EPW = CE / GW;                        // entries of a chunk per producer wavefront (64 lanes = EPW entries x QN)
    static_assert(EPW * QN == 64 && 2 * EPW <= 64, "a wavefront's tasks; val and col of its entries fit one dword request");
    constexpr int D = kChainD, NVC = kChainNVC, NBR = kChainNBR, NPR = kChainNPR;
    constexpr int PCS = CE + 4;                         // floats per product column (68: 16-byte rows for the consumer, 2-way banks for the writers)
    constexpr int NPW = 2 * GW;
    extern __shared__ __attribute__((aligned(16))) char chain_lds[];   // (more than the 64 KiB a static array may have)
    // [per producer wavefront: val/col ring NVC x {64 dwords: val[EPW], col[EPW], unused}][B-row ring NBR x 64 x 16 B] [product ring] [flags]
    constexpr int kWaveBytes = NVC * 256 + NBR * 1024;
    float *s_pr = reinterpret_cast<float *>(chain_lds + NPW * kWaveBytes);                 // [NPR][NT][PCS]
    typedef __attribute__((address_space(3))) volatile int lds_vint;   // (explicitly LDS: a generic volatile pointer captured by the lambdas below does not compile)
    lds_vint *s_flag = (lds_vint *)((__attribute__((address_space(3))) char *)chain_lds + NPW * kWaveBytes + NPR * NT * PCS * 4);
    // s_flag[0 .. NPW-1]: chunks done per producer wavefront; s_flag[NPW]: chunks taken by the consumer
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (perm: whole-matrix launches walk the chain rows longest first; row-range launches take them as they come)
    const int kr = (int)(blockIdx.x / (unsigned)ntiles), tile = (int)(blockIdx.x % (unsigned)ntiles);
    const int k = perm ? perm[kr] : k0 + kr;
    const long long len = coff[k + 1] - coff[k];
    const int nch = (int)((len + CE - 1) / CE);
    const int nfull = (int)(len / CE);                  // chunks with all 64 entries
    const int j0 = cbeg[k];
    const int jlast = j0 + (int)len - 1;
    if (tid <= NPW) s_flag[tid] = 0;
    __syncthreads();                                    // (the only barrier)
    if (wave == 0) {
        // ---------------- consumer ----------------
        float acc = 0.f;
        if (tid < NT) {
            f32x4 x[CE / 4], xn[CE / 4];
            int avail = 0;                              // chunks 0 .. avail-1 are known to be complete in the product ring
            auto wait_for = [&](int need) {             // until chunk `need - 1` is complete
                while (avail < need) {
                    int p0 = 0x7fffffff, p1 = 0x7fffffff;
#pragma unroll
                    for (int w = 0; w < GW; ++w) { p0 = min(p0, s_flag[w]); p1 = min(p1, s_flag[GW + w]); }
                    avail = __builtin_amdgcn_readfirstlane(min(2 * p0, 2 * p1 + 1));   // (uniform: keep it in a scalar register)
                    if (avail < need) __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
            };
            auto load = [&](int ch, f32x4 (&dst)[CE / 4]) {
                const f32x4 *np_ = reinterpret_cast<const f32x4 *>(s_pr + ((ch & (NPR - 1)) * NT + tid) * PCS);
#pragma unroll
                for (int u = 0; u < CE / 4; ++u) dst[u] = np_[u];
            };
            wait_for(1);
            load(0, x);
            // One chunk: `cur` = chunk c (already in registers), `nxt` <- chunk c + 1, then the 64 adds of chunk c.  The loop body is
            // kept free of everything that is not these 16 reads and 64 adds -- a wavefront issues one instruction per ~4.5 cycles
            // whatever it is, and a taken branch costs several of those:
            //   * ONE register pin for all of `cur` (it was read a whole chunk ago: the compiler waits for it here, where that is free,
            //     instead of in front of the first add; sixteen separate pins became fifteen s_waitcnt instructions);
            //   * the "taken" word is written by every active lane (same value, same address: no exec juggling);
            //   * the progress words are only read when the known-complete chunks run out (scalar compare, not taken).
#define SX_PIN16(r) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                                      "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]))
#define SX_CONSUME_FULL(c, cur, nxt)                                                                              \
            {                                                                                                     \
                SX_PIN16(cur);                                                                                    \
                s_flag[NPW] = (c) + 1;                         /* chunks 0 .. c are in registers: their ring slots are free */ \
                if (__builtin_expect((c) + 2 > avail, 0)) wait_for((c) + 2);                                      \
                load((c) + 1, nxt);                                                                               \
                asm volatile("" : "+v"(acc) : : "memory");   /* the reads are ISSUED before the adds start (memory clobber: the  */ \
                                                             /* loads stay above; acc in/out: the adds stay below) ...          */ \
                _Pragma("unroll") for (int u = 0; u < CE / 4; ++u) {                                              \
                    acc = acc + cur[u].x; acc = acc + cur[u].y; acc = acc + cur[u].z; acc = acc + cur[u].w;        \
                }                                                                                                 \
                asm volatile("" : "+v"(acc));   /* ... and the next chunk's pin and flag write stay BEHIND them (volatile asm  */ \
                                                /* statements keep their order): hoisted above, they would wait for the reads */ \
            }
            static_assert(CE / 4 == 16, "SX_PIN16");
            const int nmain = min(nfull, nch - 1);          // chunks that are full AND have a successor
            int c = 0;
            // Eight chunks per trip: the ring slot of every read is a compile-time constant (ds_read immediate offsets from one
            // base register, no address arithmetic), the "taken" word is written every other chunk.
            static_assert(NPR == 8, "the unrolled loop walks the product ring once per trip");
            const f32x4 *ring_me = reinterpret_cast<const f32x4 *>(s_pr + tid * PCS);
#define SX_CONSUME8(i, cur, nxt)                                                                                  \
            {                                                                                                     \
                SX_PIN16(cur);                                                                                    \
                if (((i) & 1) == 0) s_flag[NPW] = c + (i) + 1;                                                    \
                if (__builtin_expect(c + (i) + 2 > avail, 0)) wait_for(c + (i) + 2);                              \
                {                                                                                                 \
                    const f32x4 *np_ = ring_me + (((i) + 1) & (NPR - 1)) * (NT * PCS / 4);                        \
                    _Pragma("unroll") for (int u = 0; u < CE / 4; ++u) nxt[u] = np_[u];                           \
                }                                                                                                 \
                asm volatile("" : "+v"(acc) : : "memory");                                                        \
                _Pragma("unroll") for (int u = 0; u < CE / 4; ++u) {                                              \
                    acc = acc + cur[u].x; acc = acc + cur[u].y; acc = acc + cur[u].z; acc = acc + cur[u].w;        \
                }                                                                                                 \
                asm volatile("" : "+v"(acc));                                                                     \
            }
            static_assert((NT * PCS) % 4 == 0, "ring slots are whole f32x4");
            for (; c + 8 <= nmain; c += 8) {
                SX_CONSUME8(0, x, xn) SX_CONSUME8(1, xn, x) SX_CONSUME8(2, x, xn) SX_CONSUME8(3, xn, x)
                SX_CONSUME8(4, x, xn) SX_CONSUME8(5, xn, x) SX_CONSUME8(6, x, xn) SX_CONSUME8(7, xn, x)
            }
#undef SX_CONSUME8
            for (; c + 2 <= nmain; c += 2) {
                SX_CONSUME_FULL(c, x, xn)
                SX_CONSUME_FULL(c + 1, xn, x)
            }
            for (; c < nch; ++c) {                          // the last one to three chunks: `x` is the current one
                SX_PIN16(x);
                s_flag[NPW] = c + 1;
                if (c + 1 < nch) { wait_for(c + 2); load(c + 1, xn); }
                if (c < nfull) {
#pragma unroll
                    for (int u = 0; u < CE / 4; ++u) { acc = acc + x[u].x; acc = acc + x[u].y; acc = acc + x[u].z; acc = acc + x[u].w; }
                } else {   // the row's last, partial chunk: straight from the ring (nobody overwrites the last chunk's slot)
                    const float *cf = s_pr + ((c & (NPR - 1)) * NT + tid) * PCS;
                    const int cnt = (int)(len - (long long)nfull * CE);
                    for (int u = 0; u < cnt; ++u) acc = acc + cf[u];
                }
                if (c + 1 < nch) {
                    SX_PIN16(xn);
#pragma unroll
                    for (int u = 0; u < CE / 4; ++u) x[u] = xn[u];
                }
            }
#undef SX_CONSUME_FULL
#undef SX_PIN16
            const int64_t r = (int64_t)(crow[k] - row_base);
            const int64_t n = (int64_t)col0 + (int64_t)tile * NT + tid;
            if (c_rm) Cout[r * ldc + n] = epilogue<EXACT>(alpha, acc, beta, Cin[r * ldc_in + n]);
            else Cout[r + n * ldc] = epilogue<EXACT>(alpha, acc, beta, Cin[r + n * ldc_in]);
        }
        return;
    }
    // ---------------- producers ----------------
    const int pw = wave - 1;                            // producer number 0 .. NPW-1
    const int grp = pw / GW, wl = pw % GW;              // group (0: even chunks, 1: odd chunks), wavefront inside the group
    char *my = chain_lds + pw * kWaveBytes;             // private: val/col ring, then B-row ring
    const unsigned my_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)my);
    const int e_in_wave = lane / QN, q = lane % QN;     // my task: entry wl * EPW + e_in_wave of a chunk, columns 4q .. 4q+3 of the tile
    const int sub_per_panel = panel_width / NT;
    const float *bq = Bp + (int64_t)(tile / sub_per_panel) * panel_stride + (tile % sub_per_panel) * NT + 4 * q;
    const int pofs = (4 * q) * PCS + wl * EPW + e_in_wave;
    const int nmine = nch > grp ? (nch - grp + 1) / 2 : 0;   // chunks of my parity: grp, grp + 2, ...
    auto dma4 = [](const void *g, unsigned lds_base) {      // 4 bytes per lane -> lds_base + 4 * lane
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lds_base) : "memory");
    };
    auto dma16 = [](const void *g, unsigned lds_base) {     // 16 bytes per lane -> lds_base + 16 * lane
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lds_base) : "memory");
    };
    int taken = 0;                                      // chunks the consumer is known to have taken into registers
    // step m: stage 1 for my chunk number m + 2D, stage 2 for m + D, stage 3 for m (my chunk number i = chunk 2i + grp of the row)
    for (int m = -2 * D; m < nmine; ++m) {
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * (D - 1)) : "memory");   // everything I requested D steps ago has landed
        const int m1 = m + 2 * D, m2 = m + D;
        // LDS reads first (the requests below are compiler barriers): col of stage 2, a and the B row of stage 3
        int col = 0;
        if (m2 >= 0) col = reinterpret_cast<const int *>(my + (m2 & (NVC - 1)) * 256)[EPW + e_in_wave];
        float a = 0.f;
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (m >= 0) {
            a = reinterpret_cast<const float *>(my + (m & (NVC - 1)) * 256)[e_in_wave];
            b = *reinterpret_cast<const f32x4 *>(my + NVC * 256 + (m & (NBR - 1)) * 1024 + lane * 16);
        }
        {   // stage 1: lanes 0 .. EPW-1 fetch val, EPW .. 2 EPW - 1 col of my entries of chunk m1 (the rest repeat)
            const int ch = min(2 * m1 + grp, nch - 1);
            const int j = min(j0 + ch * CE + wl * EPW + (lane % EPW), jlast);
            const void *g = (lane / EPW) & 1 ? (const void *)(col_idx + j) : (const void *)(val + j);
            dma4(g, my_lds + (m1 & (NVC - 1)) * 256);
        }
        // stage 2: B rows of chunk m2 (before the pipeline is full: row 0)
        dma16(bq + (int64_t)col * panel_width, my_lds + NVC * 256 + (m2 & (NBR - 1)) * 1024);
        if (m >= 0) {   // stage 3
            const int ch = 2 * m + grp;
            while (ch - taken >= NPR) {                 // the slot still holds a chunk the consumer has not taken
                taken = s_flag[NPW];
                if (ch - taken >= NPR) __builtin_amdgcn_s_sleep(4);
            }
            asm volatile("" ::: "memory");
            float *d = s_pr + (ch & (NPR - 1)) * (NT * PCS) + pofs;
            d[0] = a * b.x; d[PCS] = a * b.y; d[2 * PCS] = a * b.z; d[3 * PCS] = a * b.w;
            asm volatile("" ::: "memory");
            if (lane == 0) s_flag[pw] = m + 1;          // (after the products: LDS operations of a wavefront execute in order)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the over-requested chunks: nothing may still target LDS when the wavefront ends
}

// ------------------------------------------------------------------------------------------------
// B repack: column-major K x N (leading dimension ldb) -> row-major panels of width W.
// Panel t (columns col_base + t*W ...) is written at Bp + t*K*W as K rows of W floats.  The
// reference does the equivalent re-layout for its HBM channels on the host
// (sextans-host.cpp:150-177); here it is one coalesced device pass.
// Workgroup = 256 consecutive k of one panel.
// ------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(kBlock) void repack_b_panels(const float *__restrict__ B, int64_t ldb,
                                                          float *__restrict__ Bp, int K,
                                                          int col_base, int k_begin, int k_end, int ncols, const unsigned char *__restrict__ touched) {
    // touched (may be null): one byte per 64 rows of B -- segments without a single column index of the matrix are skipped (a rank of a
    // row-partitioned SpMM whose slab reaches a few far-away columns still repacks 1 / world of B, not everything between min and max)
    // ncols: columns of B from col_base on; a last panel that reaches past them is filled with zeros there (N = 16 t + 8 run as t + 1
    // 16-column tiles: the padded columns multiply zeros and are never stored, spmm_panel_v2.h `last_cols`)
    // Rows [k_begin, k_end) of B only: the rows the matrix of this engine has columns in (a rank of a row-partitioned SpMM
    // over a banded matrix touches 1 / world of B plus a halo: engine_plan.hip, ensure_col_range); panels keep their absolute
    // addressing (row k of tile t at Bp[t K W + k W]).
    __shared__ float s[W][kBlock + 1];
    const int tid = threadIdx.x;
    const int k0 = k_begin + blockIdx.x * kBlock;
    const int t = blockIdx.y;
    const int k = k0 + tid;
    const float *src = B + (int64_t)(col_base + t * W) * ldb;
    if (touched && !(touched[k0 >> 6] | touched[(k0 >> 6) + 1] | touched[(k0 >> 6) + 2] | touched[(k0 >> 6) + 3])) return;   // (k_begin is a multiple of 64; uniform)
    if ((t + 1) * W > ncols) {   // (uniform: the zero-padded last panel only -- a per-element test in the loop below cost 35 % of the pass)
        if (k < k_end)
            for (int c = 0; c < W; ++c) s[c][tid] = t * W + c < ncols ? src[(int64_t)c * ldb + k] : 0.f;
    } else if (k < k_end) {
#pragma unroll
        for (int c = 0; c < W; ++c) s[c][tid] = src[(int64_t)c * ldb + k];
    }
    __syncthreads();
    float *dst = Bp + (int64_t)t * K * W + (int64_t)k0 * W;
    const int nk = min(kBlock, k_end - k0);
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int e = tid + i * kBlock;   // linear element of the 256 x W chunk
        const int kk = e / W, c = e % W;
        if (kk < nk) dst[e] = s[c][kk];
    }
}

}  // namespace sx
