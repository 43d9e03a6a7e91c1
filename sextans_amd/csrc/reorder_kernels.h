// reorder_kernels.h -- the three streaming passes of the REORDERED form of the LDS-panel SpMM (engine_plan.hip: ensure_cluster_plan,
// graph_cluster.hip): when the plan visits the rows in a graph-clustered order and relabels the columns to match,
//   * B is repacked into row-major 16-column panels whose row k' = colpos[k] (repack_b_panels_perm): the dictionary of a row
//     block is then (mostly) a run of consecutive panel rows -- whole 128-byte lines -- instead of 64-byte rows scattered over K;
//   * C_in is gathered into a block-major staging buffer Cs[tile][slot of the row][16] (permute_c_in) and C_out scattered back
//     from it (permute_c_out): with a column-major C (leading dimension M) the 64 rows of a clustered block are 64 unrelated
//     addresses per column -- 4-byte accesses, one line each -- while the staging buffer gives every lane of spmm_csr_panel_v2
//     ONE 16-byte load and ONE 16-byte store per tile.
// All three move 64-byte rows (4 lanes x 16 bytes) on the permuted side and whole 1 KiB runs on the column-major side.
// The reference lays B and C out for its kernel on the host, outside the timed call (sextans-host.cpp:150-195, 264-270); here the
// passes are inside the timed step.  Included by engine.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spmm_csr_kernels.h"

namespace sx {

// Workgroup = 256 consecutive k of one 16-column panel t: Bp[t][colpos[k]][0..15] = B[k][col_base + 16 t + 0..15].
__global__ __launch_bounds__(kBlock) void repack_b_panels_perm(const float *__restrict__ B, int64_t ldb, float *__restrict__ Bp, int K,
                                                               int col_base, const int *__restrict__ colpos) {
    __shared__ float s[16][kBlock + 1];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * kBlock;
    const int t = blockIdx.y;
    const float *src = B + (int64_t)(col_base + t * 16) * ldb;
    if (k0 + tid < K) {
#pragma unroll
        for (int c = 0; c < 16; ++c) s[c][tid] = src[(int64_t)c * ldb + k0 + tid];
    }
    __syncthreads();
    float *dst = Bp + (int64_t)t * K * 16;
    const int q = tid & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = (tid >> 2) + 64 * i;
        if (k0 + kk < K) {
            const f32x4 v = {s[4 * q][kk], s[4 * q + 1][kk], s[4 * q + 2][kk], s[4 * q + 3][kk]};
            *reinterpret_cast<f32x4 *>(dst + (int64_t)colpos[k0 + kk] * 16 + 4 * q) = v;
        }
    }
}

// Workgroup = 256 consecutive rows of one 16-column tile t: Cs[t][cpos[r]][0..15] = C[r][col_base + 16 t + 0..15].
__global__ __launch_bounds__(kBlock) void permute_c_in(const float *__restrict__ C, int64_t ldc, float *__restrict__ Cs, int64_t tile_stride,
                                                       const int *__restrict__ cpos, int M, int col_base) {
    __shared__ float s[16][kBlock + 1];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * kBlock;
    const int t = blockIdx.y;
    const float *src = C + (int64_t)(col_base + t * 16) * ldc;
    if (r0 + tid < M) {
#pragma unroll
        for (int c = 0; c < 16; ++c) s[c][tid] = src[(int64_t)c * ldc + r0 + tid];
    }
    __syncthreads();
    float *dst = Cs + (int64_t)t * tile_stride;
    const int q = tid & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = (tid >> 2) + 64 * i;
        if (r0 + rr < M) {
            const f32x4 v = {s[4 * q][rr], s[4 * q + 1][rr], s[4 * q + 2][rr], s[4 * q + 3][rr]};
            *reinterpret_cast<f32x4 *>(dst + (int64_t)cpos[r0 + rr] * 16 + 4 * q) = v;
        }
    }
}

// The reverse: C[r][col_base + 16 t + 0..15] = Cs[t][cpos[r]][0..15].
__global__ __launch_bounds__(kBlock) void permute_c_out(const float *__restrict__ Cs, int64_t tile_stride, float *__restrict__ C, int64_t ldc,
                                                        const int *__restrict__ cpos, int M, int col_base) {
    __shared__ float s[16][kBlock + 1];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * kBlock;
    const int t = blockIdx.y;
    const float *src = Cs + (int64_t)t * tile_stride;
    const int q = tid & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = (tid >> 2) + 64 * i;
        if (r0 + rr < M) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (int64_t)cpos[r0 + rr] * 16 + 4 * q);
            s[4 * q][rr] = v.x; s[4 * q + 1][rr] = v.y; s[4 * q + 2][rr] = v.z; s[4 * q + 3][rr] = v.w;
        }
    }
    __syncthreads();
    float *dst = C + (int64_t)(col_base + t * 16) * ldc;
    if (r0 + tid < M) {
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[(int64_t)c * ldc + r0 + tid] = s[c][tid];
    }
}

}  // namespace sx
