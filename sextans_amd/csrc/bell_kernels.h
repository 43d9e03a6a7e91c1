// bell_kernels.h -- blocked-ELL bf16 SpMM on the CDNA4 matrix cores (BASELINE config 5: "blocked-ELL
// / row-bucketed variant that feeds MFMA only where a tile is actually dense").  No analogue in the
// reference (its PEs are scalar fp32 MACs, sextans.cpp:285-295); same C = alpha*A*B + beta*C contract
// and column-major fp32 C.
//
// A: M x K in dense 32x32 bf16 blocks, ell_width block slots per block row (block_col = -1: empty).
// B: K x N bf16 column-major.  Both are repacked once into MFMA FRAGMENT ORDER so that every operand
// load of a wave is one contiguous 1 KiB read (16 B per lane):
//   A block  -> [kstep 0..1][lane 0..63][8 bf16] :  lane l holds A[m = l%32][16*s + 8*(l/32) .. +8)
//   B k-block-> [ntile][kstep][lane][8 bf16]     :  lane l holds B[32*kb + 16*s + 8*(l/32) .. +8)][n0 + l%32]
// The product is computed transposed, D = (B-tile)^T x (A-block)^T with
// v_mfma_f32_32x32x16_bf16(Bfrag, Afrag, acc): the accumulator lane index then runs along m, so for
// every accumulator register 32 lanes store 32 consecutive rows of one column of column-major C
// (128-byte coalesced stores) instead of 32 different columns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Row-major 32x32 bf16 blocks -> fragment order.  One thread per (block, kstep, lane).
__global__ __launch_bounds__(256) void bell_repack_a(const unsigned short *__restrict__ src,
                                                     u32x4 *__restrict__ dst, int64_t nblocks) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nblocks * 128) return;
    const int64_t blk = t >> 7;
    const int s = (int)(t >> 6) & 1, l = (int)t & 63;
    const unsigned short *p = src + blk * 1024 + (l & 31) * 32 + 16 * s + 8 * (l >> 5);
    dst[t] = *reinterpret_cast<const u32x4 *>(p);
}

// Column-major K x N bf16 (ldb % 8 == 0) -> fragment order.  One thread per (kb, ntile, kstep, lane).
__global__ __launch_bounds__(256) void bell_repack_b(const unsigned short *__restrict__ B, int64_t ldb,
                                                     u32x4 *__restrict__ dst, int kblocks, int ntiles) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)kblocks * ntiles * 128) return;
    const int l = (int)t & 63, s = (int)(t >> 6) & 1;
    const int64_t r = t >> 7;
    const int nt = (int)(r % ntiles);
    const int64_t kb = r / ntiles;
    const unsigned short *p = B + (int64_t)(nt * 32 + (l & 31)) * ldb + kb * 32 + 16 * s + 8 * (l >> 5);
    dst[t] = *reinterpret_cast<const u32x4 *>(p);
}

// fp32 column-major K x N (any ldb >= K) -> bf16 (round to nearest even) in fragment order, rows k >= K read as
// zero: the B operand of the dense-tile path of the CSR dispatcher, where the caller's B is fp32 and K need not
// be a multiple of 32.  One thread per (kb, ntile, kstep, lane).
__device__ __forceinline__ unsigned f32_to_bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__global__ __launch_bounds__(256) void bell_repack_b_f32(const float *__restrict__ B, int64_t ldb, int K,
                                                         u32x4 *__restrict__ dst, int kblocks, int ntiles) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)kblocks * ntiles * 128) return;
    const int l = (int)t & 63, s = (int)(t >> 6) & 1;
    const int64_t r = t >> 7;
    const int nt = (int)(r % ntiles);
    const int64_t kb = r / ntiles;
    const int64_t k0 = kb * 32 + 16 * s + 8 * (l >> 5);
    const float *p = B + (int64_t)(nt * 32 + (l & 31)) * ldb + k0;
    unsigned h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (k0 + e < K) ? f32_to_bf16_bits(p[e]) : 0u;
    u32x4 w;
    w.x = h[0] | (h[1] << 16); w.y = h[2] | (h[3] << 16); w.z = h[4] | (h[5] << 16); w.w = h[6] | (h[7] << 16);
    dst[t] = w;
}

// C_out rows [row0, M) of every column = (alpha * 0) + (beta * C_in): the rows below the last full 32-row block
// row, which the MFMA pass of the dense-tile path does not write.
__global__ __launch_bounds__(256) void scale_tail_rows(const float *Cin, int64_t ldc_in, float *Cout, int64_t ldc, int row0,
                                                       int M, int N, float alpha, float beta) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int rows = M - row0;
    if (t >= (int64_t)rows * N) return;
    const int r = row0 + (int)(t % rows), n = (int)(t / rows);
    const float t0 = alpha * 0.0f, t1 = beta * Cin[(int64_t)r + n * ldc_in];
    Cout[(int64_t)r + n * ldc] = t0 + t1;
}

// One wavefront = one block row x NSUB*32 columns.  Operand fragments of the next block are in flight
// while the current block's MFMAs issue.
template <int NSUB>
__global__ __launch_bounds__(256) void spmm_bell_mfma(
    const int *__restrict__ block_col, const bf16x8 *__restrict__ Af, const bf16x8 *__restrict__ Bf,
    const float *Cin, int64_t ldc_in, float *Cout, int64_t ldc, int mblocks, int ell_width, int ntiles,
    float alpha, float beta) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int ngroups = ntiles / NSUB;
    const int br = wave / ngroups;
    const int ng = wave % ngroups;
    if (br >= mblocks) return;

    f32x16 acc[NSUB];
#pragma unroll
    for (int t = 0; t < NSUB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int *bc_row = block_col + (int64_t)br * ell_width;
    const bf16x8 *a_row = Af + (int64_t)br * ell_width * 128 + lane;
    auto b_ptr = [&](int bc) { return Bf + ((int64_t)bc * ntiles + ng * NSUB) * 128 + lane; };

    bf16x8 a_cur[2], b_cur[NSUB][2], a_nxt[2], b_nxt[NSUB][2];
    int bc = ell_width > 0 ? bc_row[0] : -1;
    if (bc >= 0) {
        a_cur[0] = a_row[0]; a_cur[1] = a_row[64];
        const bf16x8 *bp = b_ptr(bc);
#pragma unroll
        for (int t = 0; t < NSUB; ++t) { b_cur[t][0] = bp[t * 128]; b_cur[t][1] = bp[t * 128 + 64]; }
    }
    for (int s = 0; s < ell_width; ++s) {
        const int bc_next = (s + 1 < ell_width) ? bc_row[s + 1] : -1;
        if (bc_next >= 0) {
            const bf16x8 *ap = a_row + (int64_t)(s + 1) * 128;
            a_nxt[0] = ap[0]; a_nxt[1] = ap[64];
            const bf16x8 *bp = b_ptr(bc_next);
#pragma unroll
            for (int t = 0; t < NSUB; ++t) { b_nxt[t][0] = bp[t * 128]; b_nxt[t][1] = bp[t * 128 + 64]; }
        }
        if (bc >= 0) {
#pragma unroll
            for (int t = 0; t < NSUB; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_cur[t][0], a_cur[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_cur[t][1], a_cur[1], acc[t], 0, 0, 0);
            }
        }
        bc = bc_next;
        a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1];
#pragma unroll
        for (int t = 0; t < NSUB; ++t) { b_cur[t][0] = b_nxt[t][0]; b_cur[t][1] = b_nxt[t][1]; }
    }

    // D[i = n_local][j = m_local]: lane -> m_local = lane % 32, register r -> n_local.
    const int64_t m = (int64_t)br * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < NSUB; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int64_t o = m + (int64_t)((ng * NSUB + t) * 32 + nl) * ldc;
            const float t0 = alpha * acc[t][r];
            const float t1 = beta * Cin[m + (int64_t)((ng * NSUB + t) * 32 + nl) * ldc_in];
            Cout[o] = t0 + t1;
        }
    }
}

// Full-width variant for N = 256: one wavefront = one block row x all 8 column tiles, so every A block is
// requested ONCE (spmm_bell_mfma<4> runs two wavefronts per block row, each fetching the block: the second fetch
// hits L1/L2 but is still a line request, and line requests are what bounds this kernel).  The 32x256
// accumulator is 128 registers; operands are pipelined per 16-wide k-step (A fragment + 8 B fragments = 36
// registers in flight while the previous k-step's 8 MFMAs issue): ~210 registers, 2 wavefronts per SIMD.
__global__ __launch_bounds__(256, 2) void spmm_bell_mfma_n256(
    const int *__restrict__ block_col, const bf16x8 *__restrict__ Af, const bf16x8 *__restrict__ Bf,
    const float *Cin, int64_t ldc_in, float *Cout, int64_t ldc, int mblocks, int ell_width, float alpha, float beta,
    int br_begin) {
    constexpr int NT8 = 8;
    const int lane = threadIdx.x & 63;
    const int br = __builtin_amdgcn_readfirstlane(br_begin + (int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (br >= mblocks) return;
    f32x16 acc[NT8];
#pragma unroll
    for (int t = 0; t < NT8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int *bc_row = block_col + (int64_t)br * ell_width;
    const bf16x8 *a_row = Af + (int64_t)br * ell_width * 128 + lane;
    // k-step index q = 2 * slot + ks; fragments of k-step q: A at a_row[slot * 128 + ks * 64], B tile t at
    // Bf[(bc * 8 + t) * 128 + ks * 64 + lane]
    bf16x8 a_c, b_c[NT8], a_n, b_n[NT8];
    const int nq = 2 * ell_width;
    auto load = [&](int q, bf16x8 &a, bf16x8 (&b)[NT8]) -> int {
        const int slot = q >> 1, ks = q & 1;
        const int bc = bc_row[slot];
        if (bc >= 0) {
            a = __builtin_nontemporal_load(a_row + (int64_t)slot * 128 + ks * 64);   // A is read once: keep it out of the caches B lives in
            const bf16x8 *bp = Bf + ((int64_t)bc * NT8) * 128 + ks * 64 + lane;
#pragma unroll
            for (int t = 0; t < NT8; ++t) b[t] = bp[t * 128];
        }
        return bc;
    };
    int bc_c = nq > 0 ? load(0, a_c, b_c) : -1;
    for (int q = 0; q < nq; ++q) {
        const int bc_n = q + 1 < nq ? load(q + 1, a_n, b_n) : -1;
        if (bc_c >= 0) {
#pragma unroll
            for (int t = 0; t < NT8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_c[t], a_c, acc[t], 0, 0, 0);
        }
        bc_c = bc_n;
        a_c = a_n;
#pragma unroll
        for (int t = 0; t < NT8; ++t) b_c[t] = b_n[t];
    }
    const int64_t m = (int64_t)br * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float t0 = alpha * acc[t][r];
            const float t1 = beta * Cin[m + (int64_t)(t * 32 + nl) * ldc_in];
            Cout[m + (int64_t)(t * 32 + nl) * ldc] = t0 + t1;
        }
    }
}

}  // namespace sx
