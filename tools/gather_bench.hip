// gather_bench.hip -- microbenchmarks that bound the SpMM gather kernel on MI355X:
//   copy      : float4 streaming copy (HBM ceiling calibration, expect ~6.3 TB/s of 8.0 spec)
//   gather G T: random G-byte records (G = 64 or 128, G/16 lanes per record, 16 B per lane) from a
//               table of T MiB, 8 independent loads in flight per lane, indices from a hash
//               (no index stream traffic).  Reports useful GB/s = records * G / time.
// This is the ceiling for the B-row gathers of spmm_csr_rowgroup on a matrix without locality
// (BASELINE config 4): N=16 -> 64-byte B rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31);
}

__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) b[i] = a[i];
}

template <int LPR, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ tab, uint64_t nrec, int per_group,
                                                float* __restrict__ out) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int grp = tid / LPR, q = tid % LPR;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < per_group; it += UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t r = __umul64hi(mix(((uint64_t)grp << 20) + it + u), nrec);
            const float4* p = tab + r * LPR + q;
            if (NT) { const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); v[u] = make_float4(t.x, t.y, t.z, t.w); }
            else v[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[tid] = acc.x;
}

template <int LPR, int UNROLL, bool NT>
double run_gather(const float4* tab, uint64_t nrec, float* out, int blocks, int per_group) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<LPR, UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, tab, nrec, per_group, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 3;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_gather<LPR, UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, tab, nrec, per_group, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)blocks * 256 / LPR * per_group;
    return recs * LPR * 16 / (ms / reps * 1e-3) / 1e9;
}

int main() {
    const size_t maxB = (size_t)4096 << 20;
    float4 *a, *b; float* out;
    CK(hipMalloc(&a, maxB)); CK(hipMalloc(&b, maxB)); CK(hipMalloc(&out, 64 << 20));
    CK(hipMemset(a, 1, maxB)); CK(hipMemset(b, 0, maxB));
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        size_t n = maxB / 16;
        hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy 4GiB->4GiB: %.1f GB/s (read+write)\n", 2.0 * maxB / (ms / 5 * 1e-3) / 1e9);
    }
    const int blocks = 256 * 8 * 4;
    for (size_t mib : {32, 128, 256, 512, 1024, 4096}) {
        const size_t bytes = mib << 20;
        printf("table %5zu MiB:", mib);
        printf("  64B u4 %7.1f", run_gather<4, 4, false>(a, bytes / 64, out, blocks, 256));
        printf("  64B u8 %7.1f", run_gather<4, 8, false>(a, bytes / 64, out, blocks, 256));
        printf("  64B u8 nt %7.1f", run_gather<4, 8, true>(a, bytes / 64, out, blocks, 256));
        printf("  128B u8 %7.1f", run_gather<8, 8, false>(a, bytes / 128, out, blocks, 256));
        printf("  32B u8 %7.1f", run_gather<2, 8, false>(a, bytes / 32, out, blocks, 256));
        printf("  256B u8 %7.1f GB/s\n", run_gather<16, 8, false>(a, bytes / 256, out, blocks, 256));
    }
    return 0;
}
