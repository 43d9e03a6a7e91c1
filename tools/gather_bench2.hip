// gather_bench2.hip -- can a 64-byte random gather avoid the 128-byte line fetch on gfx950?
// Variants: allocation {normal, uncached (hipDeviceMallocUncached), finegrained} x load cache bits
// {plain, nt, sc1, sc0 sc1, sc0 sc1 nt}.  Kernel template names differ so rocprofv3 PMC rows separate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31);
}
template <int MODE> __device__ __forceinline__ f32x4 ld(const f32x4* p) {
    f32x4 v;
    if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    if (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
// 8 loads in flight per lane: issue all, single wait
template <int MODE> __device__ __forceinline__ void ld8(const f32x4* const (&p)[8], f32x4 (&v)[8]) {
#define L(i, bits) "global_load_dwordx4 %" #i ", %" #bits
    if (MODE == 0) asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %9, off\n\tglobal_load_dwordx4 %2, %10, off\n\tglobal_load_dwordx4 %3, %11, off\n\tglobal_load_dwordx4 %4, %12, off\n\tglobal_load_dwordx4 %5, %13, off\n\tglobal_load_dwordx4 %6, %14, off\n\tglobal_load_dwordx4 %7, %15, off\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
    if (MODE == 1) asm volatile("global_load_dwordx4 %0, %8, off nt\n\tglobal_load_dwordx4 %1, %9, off nt\n\tglobal_load_dwordx4 %2, %10, off nt\n\tglobal_load_dwordx4 %3, %11, off nt\n\tglobal_load_dwordx4 %4, %12, off nt\n\tglobal_load_dwordx4 %5, %13, off nt\n\tglobal_load_dwordx4 %6, %14, off nt\n\tglobal_load_dwordx4 %7, %15, off nt\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\tglobal_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\tglobal_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\tglobal_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\tglobal_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1 nt\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1 nt\n\tglobal_load_dwordx4 %2, %10, off sc0 sc1 nt\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1 nt\n\tglobal_load_dwordx4 %4, %12, off sc0 sc1 nt\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1 nt\n\tglobal_load_dwordx4 %6, %14, off sc0 sc1 nt\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
    if (MODE == 5) asm volatile("global_load_dwordx4 %0, %8, off sc0\n\tglobal_load_dwordx4 %1, %9, off sc0\n\tglobal_load_dwordx4 %2, %10, off sc0\n\tglobal_load_dwordx4 %3, %11, off sc0\n\tglobal_load_dwordx4 %4, %12, off sc0\n\tglobal_load_dwordx4 %5, %13, off sc0\n\tglobal_load_dwordx4 %6, %14, off sc0\n\tglobal_load_dwordx4 %7, %15, off sc0\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
}

template <int LPR, int MODE, int ALLOC>
__global__ __launch_bounds__(256) void k_gather(const f32x4* __restrict__ tab, uint64_t nrec, int per_group, float* __restrict__ out) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int grp = tid / LPR, q = tid % LPR;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < per_group; it += 8) {
        const f32x4* p[8]; f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = tab + __umul64hi(mix(((uint64_t)grp << 20) + it + u), nrec) * LPR + q;
        ld8<MODE>(p, v);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[tid] = acc.x;
}

template <int LPR, int MODE, int ALLOC>
void run(const char* name, const f32x4* tab, size_t bytes, float* out) {
    const int blocks = 256 * 8 * 4, per_group = 128;
    const uint64_t nrec = bytes / (LPR * 16);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<LPR, MODE, ALLOC>), dim3(blocks), dim3(256), 0, 0, tab, nrec, per_group, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_gather<LPR, MODE, ALLOC>), dim3(blocks), dim3(256), 0, 0, tab, nrec, per_group, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)blocks * 256 / LPR * per_group;
    printf("  %-34s %8.1f GB/s useful\n", name, recs * LPR * 16 / (ms / 3 * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = (size_t)1024 << 20;
    float* out; CK(hipMalloc(&out, 64 << 20));
    f32x4 *t0, *t1 = nullptr, *t2 = nullptr;
    CK(hipMalloc(&t0, bytes)); CK(hipMemset(t0, 1, bytes));
    hipError_t e1 = hipExtMallocWithFlags((void**)&t1, bytes, hipDeviceMallocUncached);
    hipError_t e2 = hipExtMallocWithFlags((void**)&t2, bytes, hipDeviceMallocFinegrained);
    printf("uncached alloc: %s, finegrained alloc: %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    if (t1) CK(hipMemset(t1, 1, bytes));
    if (t2) CK(hipMemset(t2, 1, bytes));
    printf("64B records, 1 GiB table, normal alloc:\n");
    run<4, 0, 0>("plain", t0, bytes, out); run<4, 1, 0>("nt", t0, bytes, out); run<4, 2, 0>("sc1", t0, bytes, out);
    run<4, 3, 0>("sc0 sc1", t0, bytes, out); run<4, 4, 0>("sc0 sc1 nt", t0, bytes, out); run<4, 5, 0>("sc0", t0, bytes, out);
    printf("128B records, normal alloc:\n");
    run<8, 0, 0>("plain", t0, bytes, out); run<8, 3, 0>("sc0 sc1", t0, bytes, out);
    if (t1) { printf("64B records, UNCACHED alloc:\n"); run<4, 0, 1>("plain", t1, bytes, out); run<4, 1, 1>("nt", t1, bytes, out); run<4, 3, 1>("sc0 sc1", t1, bytes, out);
              printf("128B records, UNCACHED alloc:\n"); run<8, 0, 1>("plain", t1, bytes, out);
              printf("32B records, UNCACHED alloc:\n"); run<2, 0, 1>("plain", t1, bytes, out); }
    if (t2) { printf("64B records, FINEGRAINED alloc:\n"); run<4, 0, 2>("plain", t2, bytes, out); run<4, 3, 2>("sc0 sc1", t2, bytes, out); }
    return 0;
}
