// synth.hip -- deterministic synthetic inputs for the measurement harness (BASELINE config 4:
// "Synthetic 4M x 4M CSR, ~0.001% density").  Not part of the reference.
//
// Everything is a pure function of (seed, row, position) through a counter-based splitmix64 hash,
// so any row range can be generated independently, on the host or on the device, with identical
// bits: ranks of a multi-GPU run generate only their own row slice directly in HBM, and the CPU
// baseline generates the same rows on the host.
//
//   row length  : Poisson(mean) by integer inverse-CDF (64-bit thresholds computed once on the
//                 host in long double and handed to both generators), clipped to [0, min(K, 511)]
//   columns     : `len` draws floor(u * K / 2^64), sorted ascending, duplicates pushed to the next
//                 free column, then clamped from the top so all stay < K (strictly increasing)
//   values      : U[-1, 1) with 24 random bits: k * 2^-23 - 1 (exact in fp32)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sextans_amd.h"

#define SX_HD __host__ __device__ __forceinline__

namespace {

constexpr int kTable = 512;
constexpr uint64_t kValSalt = 0x76616c7565ULL;   // "value"

SX_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
SX_HD uint64_t rnd(uint64_t seed, uint64_t a, uint64_t b) {
    return splitmix64(splitmix64(seed ^ (a * 0xD6E8FEB86659FD93ULL)) + b);
}
SX_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
SX_HD float u01m1(uint64_t bits) {
    return (float)(uint32_t)(bits >> 40) * (1.0f / 8388608.0f) - 1.0f;
}
SX_HD int row_len(const uint64_t *table, uint64_t seed, int row, int K) {
    const uint64_t u = rnd(seed, (uint64_t)row, 0);
    int len = 0;
    while (len < kTable - 1 && u >= table[len]) ++len;
    return len < K ? len : K;
}
SX_HD void fill_row(uint64_t seed, int row, int K, int len, int *c, float *v) {
    for (int i = 0; i < len; ++i) {
        const int x = (int)mulhi64(rnd(seed, (uint64_t)row, 1 + (uint64_t)i), (uint64_t)K);
        int p = i;
        while (p > 0 && c[p - 1] > x) { c[p] = c[p - 1]; --p; }
        c[p] = x;
    }
    for (int i = 1; i < len; ++i)
        if (c[i] <= c[i - 1]) c[i] = c[i - 1] + 1;
    for (int i = len - 1; i >= 0; --i) {
        const int cap = K - 1 - (len - 1 - i);
        if (c[i] > cap) c[i] = cap;
    }
    for (int i = 0; i < len; ++i) v[i] = u01m1(rnd(seed ^ kValSalt, (uint64_t)row, (uint64_t)i));
}

void poisson_table(double mean, uint64_t *t) {
    long double p = expl(-(long double)mean), cdf = 0.0L;
    const long double two64 = 18446744073709551616.0L;
    for (int i = 0; i < kTable; ++i) {
        cdf += p;
        long double x = cdf * two64;
        t[i] = (x >= two64 - 1.0L || cdf >= 1.0L) ? UINT64_MAX : (uint64_t)x;
        p = p * (long double)mean / (long double)(i + 1);
    }
    t[kTable - 1] = UINT64_MAX;
}

__global__ void k_row_len(const uint64_t *table, uint64_t seed, int r0, int nrows, int K, int *lens) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) lens[i] = row_len(table, seed, r0 + i, K);
}
__global__ void k_fill_rows(uint64_t seed, int r0, int nrows, int K, const int *rp, int *col,
                            float *val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) fill_row(seed, r0 + i, K, rp[i + 1] - rp[i], col + rp[i], val + rp[i]);
}
__global__ void k_uniform(float *dst, int64_t n, uint64_t seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = u01m1(rnd(seed, (uint64_t)i, 0x51));
}

#define SY_HIP(call) do { if ((call) != hipSuccess) return SEXTANS_ERR_HIP; } while (0)

}  // namespace

extern "C" {

int sextans_gen_csr_host(int M, int K, double mean_nnz, uint64_t seed, int r0, int r1, int **row_ptr,
                         int **col_idx, float **val, int64_t *nnz) {
    if (M < 0 || K <= 0 || r0 < 0 || r1 < r0 || r1 > M || mean_nnz <= 0 || mean_nnz > 300 ||
        !row_ptr || !col_idx || !val || !nnz)
        return SEXTANS_ERR_INVALID;
    std::vector<uint64_t> table(kTable);
    poisson_table(mean_nnz, table.data());
    const int nrows = r1 - r0;
    int *rp = (int *)malloc(sizeof(int) * ((size_t)nrows + 1));
    if (!rp) return SEXTANS_ERR_ALLOC;
    int64_t tot = 0;
    rp[0] = 0;
    for (int i = 0; i < nrows; ++i) {
        tot += row_len(table.data(), seed, r0 + i, K);
        if (tot > 0x7fffffffLL) { free(rp); return SEXTANS_ERR_INVALID; }
        rp[i + 1] = (int)tot;
    }
    int *c = (int *)malloc(sizeof(int) * (size_t)(tot ? tot : 1));
    float *v = (float *)malloc(sizeof(float) * (size_t)(tot ? tot : 1));
    if (!c || !v) { free(rp); free(c); free(v); return SEXTANS_ERR_ALLOC; }
    for (int i = 0; i < nrows; ++i) fill_row(seed, r0 + i, K, rp[i + 1] - rp[i], c + rp[i], v + rp[i]);
    *row_ptr = rp; *col_idx = c; *val = v; *nnz = tot;
    return SEXTANS_OK;
}

int sextans_gen_csr_device(int device, int M, int K, double mean_nnz, uint64_t seed, int r0, int r1,
                           int **d_row_ptr, int **d_col_idx, float **d_val, int64_t *nnz) {
    if (M < 0 || K <= 0 || r0 < 0 || r1 < r0 || r1 > M || mean_nnz <= 0 || mean_nnz > 300 ||
        !d_row_ptr || !d_col_idx || !d_val || !nnz)
        return SEXTANS_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return SEXTANS_ERR_NO_DEVICE;
    SY_HIP(hipSetDevice(device));
    std::vector<uint64_t> table(kTable);
    poisson_table(mean_nnz, table.data());
    uint64_t *d_table = nullptr;
    SY_HIP(hipMalloc((void **)&d_table, sizeof(uint64_t) * kTable));
    SY_HIP(hipMemcpy(d_table, table.data(), sizeof(uint64_t) * kTable, hipMemcpyHostToDevice));
    const int nrows = r1 - r0;
    int *d_rp = nullptr;
    SY_HIP(hipMalloc((void **)&d_rp, sizeof(int) * ((size_t)nrows + 1)));
    const unsigned grid = (unsigned)((nrows + 255) / 256);
    if (nrows) hipLaunchKernelGGL(k_row_len, dim3(grid), dim3(256), 0, 0, d_table, seed, r0, nrows, K, d_rp + 1);
    std::vector<int> rp((size_t)nrows + 1, 0);
    if (nrows) SY_HIP(hipMemcpy(rp.data() + 1, d_rp + 1, sizeof(int) * (size_t)nrows, hipMemcpyDeviceToHost));
    int64_t tot = 0;
    for (int i = 0; i < nrows; ++i) {
        tot += rp[(size_t)i + 1];
        if (tot > 0x7fffffffLL) { (void)hipFree(d_table); (void)hipFree(d_rp); return SEXTANS_ERR_INVALID; }
        rp[(size_t)i + 1] = (int)tot;
    }
    SY_HIP(hipMemcpy(d_rp, rp.data(), sizeof(int) * ((size_t)nrows + 1), hipMemcpyHostToDevice));
    int *d_c = nullptr;
    float *d_v = nullptr;
    SY_HIP(hipMalloc((void **)&d_c, sizeof(int) * (size_t)(tot ? tot : 1)));
    SY_HIP(hipMalloc((void **)&d_v, sizeof(float) * (size_t)(tot ? tot : 1)));
    if (nrows) hipLaunchKernelGGL(k_fill_rows, dim3(grid), dim3(256), 0, 0, seed, r0, nrows, K, d_rp, d_c, d_v);
    SY_HIP(hipDeviceSynchronize());
    SY_HIP(hipFree(d_table));
    *d_row_ptr = d_rp; *d_col_idx = d_c; *d_val = d_v; *nnz = tot;
    return SEXTANS_OK;
}

int sextans_gen_uniform_host(float *dst, int64_t n, uint64_t seed) {
    if (!dst || n < 0) return SEXTANS_ERR_INVALID;
    for (int64_t i = 0; i < n; ++i) dst[i] = u01m1(rnd(seed, (uint64_t)i, 0x51));
    return SEXTANS_OK;
}

int sextans_gen_uniform_device(int device, float *d_dst, int64_t n, uint64_t seed, void *stream) {
    if (!d_dst || n < 0) return SEXTANS_ERR_INVALID;
    SY_HIP(hipSetDevice(device));
    if (n == 0) return SEXTANS_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_uniform, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_dst, n, seed);
    SY_HIP(hipGetLastError());
    return SEXTANS_OK;
}

}  // extern "C"
