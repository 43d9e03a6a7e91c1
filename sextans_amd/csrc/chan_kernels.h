// chan_kernels.h -- device-side conversion between the accelerator's dense channel layouts
// (sextans-host.cpp:152-195, :264-270) and the column-major B / C the SpMM kernels consume.
// Pure streaming passes (HBM-bound, 8 bytes moved per element); one thread moves the 8 floats
// that are contiguous in a channel so both sides see >= 32-byte segments.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sx {

typedef float f32x4c __attribute__((ext_vector_type(4)));

// B, 8 channels: column n -> channel n % 8 at k + colsize * (n / 8).  Thread = (k, n).
// B, 4 channels: columns (2c, 2c+1) of a tile share channel c in alternating runs of 8 rows.
__global__ void __launch_bounds__(256)
chan_unpack_b(const float *__restrict__ ch, int64_t chan_len, int64_t colsize, int num_ch_b, int K, int N,
              float *__restrict__ B) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (k >= K) return;
    int c;
    int64_t pos;
    if (num_ch_b == 8) {
        c = n & 7;
        pos = k + colsize * (n >> 3);
    } else {
        c = (n >> 1) & 3;
        pos = (k >> 3) * 16 + (n & 1) * 8 + (k & 7) + colsize * (n >> 3);
    }
    B[k + (int64_t)K * n] = ch[(int64_t)c * chan_len + pos];
}

// C_in: thread = (row m, N-tile t): 8 contiguous floats of channel m % 8 -> 8 columns of C.
__global__ void __launch_bounds__(256)
chan_unpack_c(const float *__restrict__ ch, int64_t chan_len, int64_t colsize, int M, int N,
              float *__restrict__ C) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (m >= M) return;
    const float *src = ch + (int64_t)(m & 7) * chan_len + colsize * t + (m >> 3) * 8;
    const f32x4c lo = *(const f32x4c *)src, hi = *(const f32x4c *)(src + 4);
    float *dst = C + m + (int64_t)M * (t * 8);
    dst[0] = lo.x;               dst[(int64_t)M] = lo.y;
    dst[(int64_t)M * 2] = lo.z;  dst[(int64_t)M * 3] = lo.w;
    dst[(int64_t)M * 4] = hi.x;  dst[(int64_t)M * 5] = hi.y;
    dst[(int64_t)M * 6] = hi.z;  dst[(int64_t)M * 7] = hi.w;
}

// C_out: thread = (padded row m < colsize, tile t).  Rows >= M receive `pad` (= alpha*0 + beta*0:
// the accelerator computes them from zero partial sums and zero-padded C_in, sextans.cpp:196-233).
__global__ void __launch_bounds__(256)
chan_pack_c(const float *__restrict__ C, int M, int N, int64_t chan_len, int64_t colsize, float pad,
            float *__restrict__ ch) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (m >= colsize) return;
    f32x4c lo, hi;
    if (m < M) {
        const float *src = C + m + (int64_t)M * (t * 8);
        lo.x = src[0];               lo.y = src[(int64_t)M];
        lo.z = src[(int64_t)M * 2];  lo.w = src[(int64_t)M * 3];
        hi.x = src[(int64_t)M * 4];  hi.y = src[(int64_t)M * 5];
        hi.z = src[(int64_t)M * 6];  hi.w = src[(int64_t)M * 7];
    } else {
        lo = f32x4c{pad, pad, pad, pad};
        hi = lo;
    }
    float *dst = ch + (int64_t)(m & 7) * chan_len + colsize * t + (m >> 3) * 8;
    *(f32x4c *)dst = lo;
    *(f32x4c *)(dst + 4) = hi;
}

}  // namespace sx
