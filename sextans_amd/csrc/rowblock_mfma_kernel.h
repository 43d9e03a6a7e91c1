// rowblock_mfma_kernel.h -- fp32 SpMM of DENSE ROW BLOCKS on the fp32 matrix cores (option "mfma_dense_tiles" = 2; round 6).
//
// north_star: "feeds MFMA only where a tile is actually dense" -- here without a precision trade.  The reference's PEs multiply and
// accumulate in fp32 (sextans.cpp:285-295, 425-446); gfx950's v_mfma_f32_16x16x4_f32 does the same arithmetic as a k-ordered chain
// of fused multiply-adds, D = fma(a_k3, b_k3, fma(a_k2, b_k2, fma(a_k1, b_k1, fma(a_k0, b_k0, C)))) with one rounding per step and no
// wider accumulator (MI355X guide, "FP32-input MFMA").  A row whose entries are walked in ascending column order through such
// instructions therefore gets EXACTLY the bits of the engine's "exact" = 0 kernels (acc = fmaf(a, b, acc) in CSR order, epilogue
// fmaf(alpha, acc, beta * c_in)): the zero entries that pad a fragment contribute fmaf(0, b, acc) = acc for finite b.
//
// Unit of routing = a block of 16 consecutive rows (never a part of a row: a row is summed by ONE kernel, in ONE order).  Its
// columns are covered by GROUPS of 4 consecutive columns (4 c .. 4 c + 3), ascending; every group is one 16 x 4 fragment of A, stored
// in MFMA operand order (64 floats: lane l holds A[row0 + (l & 15)][4 c + (l >> 4)]).  fill = entries / (64 x groups); blocks whose
// fill reaches "dense_tile_fill_x100" % are routed here, all other rows stay on the CSR kernels, which skip the routed rows.
// A fully dense 32 x 32 tile is 2 blocks x 8 groups at fill 1; a 6-dof FEM row block reaches ~0.6, a 3-dof one ~0.35.
//
// Per group and block: ONE coalesced 256-byte load of the A fragment; per group and tile ONE coalesced load of the B fragment -- lane l
// reads B[4 c + (l >> 4)][n0 + (l & 15)] from the row-major B panels the engine repacks anyway (panel row = PW floats: the four rows of
// a group are 4 x 64 consecutive bytes; PW = 16: one 256-byte run) -- and one MFMA, D^T = B^T-fragment x A^T-fragment, so that a
// register of D holds 16 consecutive rows of one column of C (64-byte runs of column-major C).  The next entry's fragments are
// requested before the current entry's MFMAs issue.
// Rate: 2 * 16 * 16 * 4 flop per 32 cycles per SIMD = the fp32 vector peak; at N >= 64 the VALU kernels reach ~1/4 of it (two issue
// slots per multiply-add pair plus the operand moves), which is where this path pays.  At N <= 32 it is bound by the A stream:
// 256 bytes per group = 4 / fill bytes per entry against 4.3 - 6 of the packed CSR forms.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "spmm_csr_kernels.h"   // xcd_remap

namespace sx {

typedef float rb_f32x4 __attribute__((ext_vector_type(4)));

// One wavefront = one SUPER BLOCK (up to 4 routed blocks = 64 rows, consecutive in the routed order) x NT tiles of 16 columns.
// The super block walks the UNION of its blocks' column groups, ascending: per union entry the B fragments are loaded ONCE (NT
// coalesced loads) and every block that owns the group (4-bit mask) multiplies its own A fragment into its own accumulators.  A first
// form with one block per wavefront loaded the B fragments once per MFMA and was bound by L2 -> L1 traffic at 24 TFLOP/s on fully
// dense blocks (profiles/r06_rowblock_mfma_v1.jsonl); the VALU kernels reuse a B row across the 64 rows of a row block through LDS,
// this form reuses it across the same 64 rows through registers.
// The walk is BRANCH-FREE: a block that does not own a group multiplies zeros (a buffer load whose offset lies outside its resource
// returns 0) -- fmaf(0, b, acc) = acc -- instead of skipping the instruction.  The first form of this loop branched on the owner mask
// and the compiler answered with 128 accumulator moves between register files and a full vmcnt(0) in front of every MFMA group per
// iteration; straight-line code keeps the accumulators where the MFMAs want them and lets the wait-count pass count the prefetches.
// Every fragment load is a BUFFER load -- resource + per-lane offset that never changes + a scalar offset per entry -- so that an entry
// costs scalar arithmetic only: the flat-address form spent ~60 vector instructions per entry on 64-bit address arithmetic, which
// share the issue port with the 16 MFMAs (matrix cores 38 - 45 % busy, profiles/r06_rowblock_mfma_pmc.txt).
template <int NT>
__global__ __launch_bounds__(256) void spmm_rowblock_mfma_f32(const int *__restrict__ rb_row0, const int *__restrict__ rb_gptr, const int *__restrict__ sb_uptr,
                                                              const int2 *__restrict__ sb_u, const float *__restrict__ rb_A, const float *__restrict__ Bp,
                                                              int64_t panel_stride, int PW, int K, const float *Cin, int64_t ldc_in, float *Cout, int64_t ldc,
                                                              int nrb, int nsb, int ntile_groups, int ncols_panel, int ncols, int row_begin, int row_end,
                                                              float alpha, float beta) {
    const int lane = threadIdx.x & 63;
    // (workgroup b runs on XCD b % 8: every XCD gets a contiguous run of super blocks, so that neighbours -- which share B rows, and
    // the tile groups of one super block, which share its A fragments -- meet in ONE L2.  Without it 95 % of the L2 requests of a
    // block-tridiagonal matrix missed: profiles/r06_rowblock_mfma_pmc.txt)
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const int sbi = __builtin_amdgcn_readfirstlane((int)(wg / (unsigned)ntile_groups) * 4 + (int)(threadIdx.x >> 6));
    const int tg = (int)(wg % (unsigned)ntile_groups);
    if (sbi >= nsb) return;
    const int nb_here = min(4, nrb - 4 * sbi);
    int row0[4];
    const int gbase = __builtin_amdgcn_readfirstlane(rb_gptr[4 * sbi]);
    const int gend = __builtin_amdgcn_readfirstlane(rb_gptr[4 * sbi + nb_here]);
    int fo[4];   // byte offset of every block's next A fragment inside this super block's run of fragments (stored in ascending group order)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rb = min(4 * sbi + q, nrb - 1);
        row0[q] = q < nb_here ? __builtin_amdgcn_readfirstlane(rb_row0[rb]) : -1000;
        fo[q] = (__builtin_amdgcn_readfirstlane(rb_gptr[rb]) - gbase) * 256;
    }
    if (row0[0] >= row_end || (row0[nb_here - 1] + 16 <= row_begin)) return;   // (row-range calls: super blocks outside the range; row0 ascends)
    const int u0 = __builtin_amdgcn_readfirstlane(sb_uptr[sbi]), u1 = __builtin_amdgcn_readfirstlane(sb_uptr[sbi + 1]);
    if (u0 >= u1) return;   // (cannot happen: a routed block has entries)
    const int kq = lane >> 4, li = lane & 15;
    // A: one resource over the super block's fragments; a lane's offset is 4 * lane, or far outside for a block that does not own the entry
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(rb_A + (int64_t)gbase * 64), 0, (gend - gbase) * 256, 0x00020000);
    // B: one resource per tile = the panel that holds it; a lane's offset inside a group of 4 panel rows never changes
    __amdgpu_buffer_rsrc_t rbp[NT];
    int bvo[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int c0 = (tg * NT + i) * 16;
        const int c0c = c0 < ncols_panel ? c0 : 0;                 // (a tile beyond the panels reads tile 0's: never stored)
        const int p = c0c / PW, pc = c0c - p * PW + li;
        // (the resource ends with the panel: the rows 4 c + kq >= K of the LAST group of a K that is no multiple of 4 read as 0 -- behind row
        // K - 1 lies the next panel, or memory no repack ever wrote, and 0 x NaN would not be 0)
        rbp[i] = __builtin_amdgcn_make_buffer_rsrc((void *)(Bp + (int64_t)p * panel_stride), 0, K * PW * 4, 0x00020000);
        bvo[i] = (kq * PW + (pc < PW ? pc : 0)) * 4;               // (8-column tail panel: the lanes of columns 8 .. 15 read column 0's, never stored)
    }
    rb_f32x4 acc[4][NT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[q][i] = rb_f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane4 = lane * 4;
    auto fetch = [&](int u, float (&a)[4], float (&b)[NT]) {   // entry u, or past the end: the last entry again with no owner
        const int2 e = sb_u[min(u, u1 - 1)];
        const unsigned m = u < u1 ? (unsigned)e.y : 0u;
        const int bso = e.x * 16 * PW;                             // bytes from the panel's first row to row 4 c (added to the lane offset: the range check sees it)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned bit = (m >> q) & 1u;
            a[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, lane4 + (bit ? 0 : 0x40000000), fo[q], 0));
            fo[q] += bit ? 256 : 0;
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) b[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbp[i], bvo[i] + bso, 0, 0));
    };
    auto multiply = [&](const float (&a)[4], const float (&b)[NT]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[q][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i], a[q], acc[q][i], 0, 0, 0);
    };
    // A ring of four register sets, three entries ahead (24 loads in flight per wavefront), with scheduling barriers between the phases:
    // the loads of entry u + 3 are ISSUED before the MFMAs of entry u and awaited only three entries later.  (Left to itself the
    // scheduler sinks a prefetch below the MFMAs to shorten live ranges and the loop becomes load -> wait -> 16 MFMAs.)
    float a0[4], b0[NT], a1[4], b1[NT], a2[4], b2[NT], a3[4], b3[NT];
    fetch(u0, a0, b0);
    fetch(u0 + 1, a1, b1);
    fetch(u0 + 2, a2, b2);
    for (int u = u0; u < u1; u += 4) {   // (entries past the end multiply zeros: nothing changes)
        fetch(u + 3, a3, b3);
        __builtin_amdgcn_sched_barrier(0);
        multiply(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(u + 4, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        multiply(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(u + 5, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        multiply(a2, b2);
        __builtin_amdgcn_sched_barrier(0);
        fetch(u + 6, a2, b2);
        __builtin_amdgcn_sched_barrier(0);
        multiply(a3, b3);
        __builtin_amdgcn_sched_barrier(0);
    }
    // D[n_local = 4 * (lane >> 4) + r][row_local = lane & 15]: 16 consecutive rows of one column per register.  All C_in values of a
    // block are requested at once from clamped addresses (16 NT / 4 loads in flight), then masked stores.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q >= nb_here) break;
        const int row = row0[q] + li;
        const bool row_ok = row >= row_begin && row < row_end;
        const int64_t r_in = (int64_t)(min(max(row, row_begin), row_end - 1) - row_begin);
        float cin[NT][4];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = min((tg * NT + i) * 16 + 4 * kq + r, ncols - 1);
                cin[i][r] = Cin[r_in + (int64_t)c * ldc_in];
            }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = (tg * NT + i) * 16 + 4 * kq + r;
                if (row_ok && c < ncols) Cout[r_in + (int64_t)c * ldc] = __builtin_fmaf(alpha, acc[q][i][r], beta * cin[i][r]);   // = epilogue<false> of the CSR kernels
            }
    }
}

// A fragments from the CSR arrays, on the device: one wavefront per routed block; every entry finds its group by binary search in the
// block's ascending group list and lands at fragment position (column & 3) * 16 + local row.  (The fragment buffer is zeroed first.)
__global__ __launch_bounds__(256) void rowblock_fill_fragments(const int *__restrict__ rp, const int *__restrict__ ci, const float *__restrict__ v,
                                                               const int *__restrict__ rb_row0, const int *__restrict__ rb_gptr, const int *__restrict__ rb_gcol,
                                                               float *__restrict__ rb_A, int nrb) {
    const int lane = threadIdx.x & 63;
    const int rbi = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rbi >= nrb) return;
    const int row0 = rb_row0[rbi], g0 = rb_gptr[rbi], g1 = rb_gptr[rbi + 1];
    for (int lr = 0; lr < 16; ++lr) {
        const int j0 = rp[row0 + lr], j1 = rp[row0 + lr + 1];
        for (int j = j0 + lane; j < j1; j += 64) {
            const int c = ci[j], c4 = c >> 2;
            int lo = g0, hi = g1 - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (rb_gcol[mid] < c4) lo = mid + 1; else hi = mid;
            }
            // (duplicate (row, column) entries of a caller's matrix: summed in fp32 like everywhere else would be a different rounding --
            // the builder does not route blocks that hold duplicates, so a plain store is exact)
            rb_A[(int64_t)lo * 64 + (c & 3) * 16 + lr] = v[j];
        }
    }
}

// skip[r] = 1 for the rows of the routed blocks (the CSR kernels never write them)
__global__ __launch_bounds__(256) void rowblock_mark_skip(const int *__restrict__ rb_row0, int nrb, unsigned char *__restrict__ skip) {
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrb * 16) return;
    skip[rb_row0[i >> 4] + (i & 15)] = 1;
}

}  // namespace sx
