// reorder_kernels.h -- the three streaming passes of the REORDERED form of the LDS-panel SpMM (engine_plan.hip: ensure_cluster_plan,
// graph_cluster.hip): when the plan visits the rows in a graph-clustered order and relabels the columns to match,
//   * B is repacked into row-major 16-column panels whose row k' = colpos[k] (repack_b_panels_perm): the dictionary of a row
//     block is then (mostly) a run of consecutive panel rows -- whole 128-byte lines -- instead of 64-byte rows scattered over K;
//   * C goes through a ROW-major staging buffer Cs[tile][row][16] (repack_b_panels<16> applied to C_in on the way in,
//     tiles_to_colmajor on the way out): with a column-major C (leading dimension M) the 64 rows of a clustered block are 64
//     unrelated addresses per column -- 4-byte accesses, one line each -- while a row of the staging buffer is 64 contiguous
//     bytes: every lane of spmm_csr_panel_v2 issues ONE 16-byte load and ONE 16-byte store per tile, the four lanes of a row
//     together one 64-byte segment (the kernel's slot -> row table says where).  The staging passes themselves are plain
//     streaming transposes (no permutation; first version: staging in block order, both passes scattering 64-byte rows --
//     4.2-4.4 TB/s against 6.3 TB/s for the streaming form).
// The reference lays B and C out for its kernel on the host, outside the timed call (sextans-host.cpp:150-195, 264-270); here the
// passes are inside the timed step.  Included by engine.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spmm_csr_kernels.h"

namespace sx {

// Workgroup = 256 consecutive k of one 16-column panel t: Bp[t][colpos[k]][0..15] = B[k][col_base + 16 t + 0..15].
__global__ __launch_bounds__(kBlock) void repack_b_panels_perm(const float *__restrict__ B, int64_t ldb, float *__restrict__ Bp, int K,
                                                               int col_base, const int *__restrict__ colpos, int k_begin, int k_end, int ncols,
                                                               const unsigned char *__restrict__ touched) {
    __shared__ float s[16][kBlock + 1];
    const int tid = threadIdx.x;
    const int k0 = k_begin + blockIdx.x * kBlock;
    const int t = blockIdx.y;
    const float *src = B + (int64_t)(col_base + t * 16) * ldb;
    if (touched && !(touched[k0 >> 6] | touched[(k0 >> 6) + 1] | touched[(k0 >> 6) + 2] | touched[(k0 >> 6) + 3])) return;   // (see repack_b_panels)
    if ((t + 1) * 16 > ncols) {   // (ncols: see repack_b_panels -- uniform test, the zero-padded last panel only)
        if (k0 + tid < k_end)
            for (int c = 0; c < 16; ++c) s[c][tid] = t * 16 + c < ncols ? src[(int64_t)c * ldb + k0 + tid] : 0.f;
    } else if (k0 + tid < k_end) {
#pragma unroll
        for (int c = 0; c < 16; ++c) s[c][tid] = src[(int64_t)c * ldb + k0 + tid];
    }
    __syncthreads();
    float *dst = Bp + (int64_t)t * K * 16;
    const int q = tid & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = (tid >> 2) + 64 * i;
        if (k0 + kk < k_end) {
            const f32x4 v = {s[4 * q][kk], s[4 * q + 1][kk], s[4 * q + 2][kk], s[4 * q + 3][kk]};
            *reinterpret_cast<f32x4 *>(dst + (int64_t)colpos[k0 + kk] * 16 + 4 * q) = v;
        }
    }
}

// C staging: Cs[t][r][0..15] = C[r][col_base + 16 t + 0..15] is repack_b_panels<16> applied to C (rows in their NATURAL order: the
// kernel scatters / gathers 64-byte rows itself, where the latency hides behind everything else it has in flight); this is the way
// back.  Workgroup = 256 consecutive rows of one tile.
__global__ __launch_bounds__(kBlock) void tiles_to_colmajor(const float *__restrict__ Cs, float *__restrict__ C, int64_t ldc, int M,
                                                            int col_base, int ncols) {
    // ncols: columns of C from col_base on (the last tile of N = 16 t + 8 has 8: its other staging columns are not written back)
    // 16-byte accesses on both sides: a lane reads 4 consecutive floats of the staging chunk and writes 4 consecutive ROWS of one
    // column (rows of the LDS tile are 260 floats apart: 16-byte aligned, conflict-free for both access patterns).
    constexpr int LD = kBlock + 4;
    __shared__ __attribute__((aligned(16))) float s[16 * LD];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * kBlock;
    const int t = blockIdx.y;
    const float *src = Cs + (int64_t)t * M * 16 + (int64_t)r0 * 16;
    const int nr = min(kBlock, M - r0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e4 = tid + i * kBlock;          // float4 number e4 of the 256 x 16 chunk: row e4 / 4, columns 4 (e4 % 4) ..
        const int rr = e4 >> 2, c4 = (e4 & 3) * 4;
        if (rr < nr) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (int64_t)e4 * 4);
            s[(c4 + 0) * LD + rr] = v.x; s[(c4 + 1) * LD + rr] = v.y; s[(c4 + 2) * LD + rr] = v.z; s[(c4 + 3) * LD + rr] = v.w;
        }
    }
    __syncthreads();
    float *dst = C + (int64_t)(col_base + t * 16) * ldc + r0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = tid + i * kBlock;           // group g: column g / 64, rows 4 (g % 64) .. + 3
        const int c = g >> 6, rr = (g & 63) * 4;
        if (t * 16 + c >= ncols) continue;
        float *d = dst + (int64_t)c * ldc + rr;
        if (rr + 3 < nr) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(&s[c * LD + rr]);
            if ((reinterpret_cast<uintptr_t>(d) & 15) == 0) *reinterpret_cast<f32x4 *>(d) = v;
            else { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
        } else {
            for (int k = 0; rr + k < nr; ++k) d[k] = s[c * LD + rr + k];
        }
    }
}

}  // namespace sx
