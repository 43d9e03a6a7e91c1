// tools/valu_bench.hip -- cycles per VALU instruction on gfx950 (is packed fp32 full rate? what do DPP moves cost?)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_bench tools/valu_bench.hip && tools/bin/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, long long *cycles, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
    f2 p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {   // 64 independent-ish v_pk_mul_f32 (8 chains)
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if constexpr (MODE == 1) {   // 64 v_mul_f32 (8 chains)
            asm volatile(REP8("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if constexpr (MODE == 2) {   // 32 x (pk_mul into temp, dependent pk_add into one of 4 accumulators): the kernel's pattern
            f2 t0_, t1_, t2_, t3_;
            asm volatile(REP8("v_pk_mul_f32 %4, %8, %9\n v_pk_mul_f32 %5, %8, %9\n v_pk_mul_f32 %6, %8, %9\n v_pk_mul_f32 %7, %8, %9\n"
                              "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %6\n v_pk_add_f32 %3, %3, %7\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_) : "v"(p4), "v"(q));
        } else if constexpr (MODE == 3) {   // same, chain-major order as the compiler emits: mul, add(dep), mul, add(dep) on ONE accumulator
            f2 t0_;
            asm volatile(REP8("v_pk_mul_f32 %1, %2, %3\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %2, %3\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n"
                              "v_pk_mul_f32 %1, %2, %3\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %2, %3\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n")
                         : "+v"(p0), "=&v"(t0_) : "v"(p4), "v"(q));
        } else if constexpr (MODE == 4) {   // 64 v_mov_b32_dpp quad_perm
            asm volatile(REP8("v_mov_b32_dpp %0, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %2, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %4, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %6, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if constexpr (MODE == 5) {   // 64 v_fma_f32 (8 chains)
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if constexpr (MODE == 6) {   // 64 v_pk_fma_f32 (8 chains)
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                              "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if constexpr (MODE == 8) {   // ONE dependent chain of 64 v_add_f32 (the exact-chain consumer's pattern)
            asm volatile(REP64("v_add_f32 %0, %0, %1\n") : "+v"(a0) : "v"(b0));
        } else if constexpr (MODE == 9) {   // the same chain with an independent ds-free VALU instruction between the adds
            asm volatile(REP64("v_add_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n") : "+v"(a0), "+v"(a1) : "v"(b0));
        } else if constexpr (MODE == 10) {  // two independent chains interleaved
            asm volatile(REP64("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n") : "+v"(a0), "+v"(a1) : "v"(b0));
        } else if constexpr (MODE == 7) {   // the kernel's pattern with scalar ops: 4 v_mul + 4 dependent v_add per 4 floats
            float t0_, t1_, t2_, t3_;
            asm volatile(REP8("v_mul_f32 %4, %8, %9\n v_mul_f32 %5, %8, %9\n v_mul_f32 %6, %8, %9\n v_mul_f32 %7, %8, %9\n"
                              "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_) : "v"(a4), "v"(b0));
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) cycles[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
void run(const char *name, int threads, int ninstr) {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16 * 8);
    const int iters = 2000;
    k<MODE><<<256, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double per = (double)h[0] / ((double)iters * ninstr);
    printf("%-34s %4d thr/CU (%d waves/SIMD): %6.2f clk per instr per wave => %5.2f clk per instr per SIMD; kernel %.3f ms\n", name, threads,
           threads / 256, per, per / (threads / 256.0), ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512, 1024}) {
        run<1>("v_mul_f32 x64 (8 chains)", threads, 64);
        run<5>("v_fma_f32 x64 (8 chains)", threads, 64);
        run<0>("v_pk_mul_f32 x64 (8 chains)", threads, 64);
        run<6>("v_pk_fma_f32 x64 (8 chains)", threads, 64);
        run<2>("pk_mul x4 + dep pk_add x4 (x8)", threads, 64);
        run<3>("pk_mul,nop,pk_add one chain (x32)", threads, 64);
        run<7>("v_mul x4 + dep v_add x4 (x8)", threads, 64);
        run<4>("v_mov_b32_dpp x64", threads, 64);
        run<8>("v_add_f32 x64, ONE dependent chain", threads, 64);
        run<9>("dep v_add + indep v_mul (x64)", threads, 128);
        run<10>("two dep v_add chains interleaved", threads, 128);
    }
    return 0;
}
