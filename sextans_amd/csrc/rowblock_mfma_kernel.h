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
// One wavefront = one routed block x NT tiles of 16 columns of C.  Per group: ONE coalesced 256-byte load of the A fragment, and per
// tile ONE coalesced load of the B fragment -- lane l reads B[4 c + (l >> 4)][n0 + (l & 15)] from the row-major B panels the engine
// repacks anyway (panel row = PW floats: the four rows of a group are 4 x 64 consecutive bytes; PW = 16: one 256-byte run) -- and one
// MFMA, D^T = B^T-fragment x A^T-fragment, so that a register of D holds 16 consecutive rows of one column of C (64-byte runs of
// column-major C).  The next group's fragments are requested before the current group's MFMAs issue.
// Rate: 2 * 16 * 16 * 4 flop per 32 cycles per SIMD = the fp32 vector peak; at N >= 64 the VALU kernels reach ~1/4 of it (two issue
// slots per multiply-add pair plus the operand moves), which is where this path pays.  At N <= 32 it is bound by the A stream:
// 256 bytes per group = 4 / fill bytes per entry against 4.3 - 6 of the packed CSR forms.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sx {

typedef float rb_f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void spmm_rowblock_mfma_f32(const int *__restrict__ rb_row0, const int *__restrict__ rb_gptr, const int *__restrict__ rb_gcol,
                                                              const float *__restrict__ rb_A, const float *__restrict__ Bp, int64_t panel_stride, int PW, int K,
                                                              const float *Cin, int64_t ldc_in, float *Cout, int64_t ldc, int nrb, int ntile_groups, int ncols_panel,
                                                              int ncols, int row_begin, int row_end, float alpha, float beta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rbi = (int)(blockIdx.x / (unsigned)ntile_groups) * 4 + wave;
    const int tg = (int)(blockIdx.x % (unsigned)ntile_groups);
    if (rbi >= nrb) return;
    const int row0 = rb_row0[rbi];
    if (row0 + 16 <= row_begin || row0 >= row_end) return;   // (row-range calls: blocks outside the range)
    const int g0 = rb_gptr[rbi], g1 = rb_gptr[rbi + 1];
    const int kq = lane >> 4, li = lane & 15;
    // per tile: where this lane's B element sits inside a panel row, and whether the column exists in the panels at all
    int64_t boff[NT];
    bool bok[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int c = (tg * NT + i) * 16 + li;
        bok[i] = c < ncols_panel;
        const int p = c / PW, pc = c - p * PW;
        boff[i] = (int64_t)p * panel_stride + pc;
    }
    rb_f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = rb_f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_b = [&](int c4, float (&b)[NT]) {
        const int k = 4 * c4 + kq;
        const bool kok = k < K;
        const int64_t rowoff = (int64_t)k * PW;
#pragma unroll
        for (int i = 0; i < NT; ++i) b[i] = (kok && bok[i]) ? Bp[boff[i] + rowoff] : 0.f;
    };
    float a_cur = 0.f, b_cur[NT];
    if (g0 < g1) {
        a_cur = rb_A[(int64_t)g0 * 64 + lane];
        load_b(rb_gcol[g0], b_cur);
    }
    for (int g = g0; g < g1; ++g) {
        float a_nxt = 0.f, b_nxt[NT];
        if (g + 1 < g1) {   // the next group's fragments are in flight while this group's MFMAs issue
            a_nxt = rb_A[(int64_t)(g + 1) * 64 + lane];
            load_b(rb_gcol[g + 1], b_nxt);
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i) b_nxt[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_cur[i], a_cur, acc[i], 0, 0, 0);
        a_cur = a_nxt;
#pragma unroll
        for (int i = 0; i < NT; ++i) b_cur[i] = b_nxt[i];
    }
    // D[n_local = 4 * (lane >> 4) + r][row_local = lane & 15]: 16 consecutive rows of one column per register
    const int row = row0 + li;
    if (row < row_begin || row >= row_end) return;
    const int64_t r_in = (int64_t)(row - row_begin);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = (tg * NT + i) * 16 + 4 * kq + r;
            if (c < ncols) {
                const float cin = Cin[r_in + (int64_t)c * ldc_in];
                Cout[r_in + (int64_t)c * ldc] = __builtin_fmaf(alpha, acc[i][r], beta * cin);   // = epilogue<false> of the CSR kernels
            }
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
