// tools/chain_bench.hip -- what does the exact-chain consumer's loop cost?  One wavefront: per 64-entry chunk 16 ds_read_b128 of the
// NEXT chunk, then 64 dependent v_add_f32 on the current one.  Variants: active lanes 64 / 16, with / without the LDS reads,
// adds from registers of a previous ds_read vs from constants.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/chain_bench tools/chain_bench.hip && tools/bin/chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, long long *cycles, int chunks) {
    __shared__ __attribute__((aligned(16))) float ring[4][16][68];
    for (int i = threadIdx.x; i < 4 * 16 * 68; i += 64) (&ring[0][0][0])[i] = 1e-9f * i;
    __syncthreads();
    float acc = 0.f;
    const int lanes = (MODE & 1) ? 16 : 64;
    const int tid = threadIdx.x;
    long long t0 = 0, t1 = 0;
    if (tid < lanes) {
        f32x4 x[16], xn[16];
        const int col = tid & 15;
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = reinterpret_cast<const f32x4 *>(&ring[0][col][0])[u];
        t0 = clock64();
#define STEP(c, cur, nxt)                                                                          \
        {                                                                                          \
            _Pragma("unroll") for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(cur[u]));        \
            if (!(MODE & 2)) {                                                                     \
                const f32x4 *np_ = reinterpret_cast<const f32x4 *>(&ring[((c) + 1) & 3][col][0]);  \
                _Pragma("unroll") for (int u = 0; u < 16; ++u) nxt[u] = np_[u];                    \
            }                                                                                      \
            if (MODE & 4) {   /* adds as one asm block: no compiler scheduling in between */       \
                _Pragma("unroll") for (int u = 0; u < 16; ++u)                                     \
                    asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4" \
                                 : "+v"(acc) : "v"(cur[u].x), "v"(cur[u].y), "v"(cur[u].z), "v"(cur[u].w)); \
            } else {                                                                               \
                _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                   \
                    acc = acc + cur[u].x; acc = acc + cur[u].y; acc = acc + cur[u].z; acc = acc + cur[u].w; \
                }                                                                                  \
            }                                                                                      \
        }
        for (int c = 0; c < chunks; c += 2) {
            STEP(c, x, xn)
            if (MODE & 2) { STEP(c + 1, x, xn) } else { STEP(c + 1, xn, x) }
        }
        t1 = clock64();
    }
    out[blockIdx.x * 64 + tid] = acc;
    if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int MODE>
void run(const char *name) {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 64 * 4); (void)hipMalloc(&cyc, 8);
    const int chunks = 4000;
    k<MODE><<<256, 64>>>(out, cyc, chunks);
    (void)hipDeviceSynchronize();
    long long h = 0; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-70s %7.1f cycles per chunk (64 adds)\n", name, (double)h / chunks);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0>("64 lanes, 16 ds_read_b128 of the next chunk + 64 dependent adds");
    run<1>("16 lanes, 16 ds_read_b128 of the next chunk + 64 dependent adds");
    run<2>("64 lanes, no LDS reads, 64 dependent adds");
    run<3>("16 lanes, no LDS reads, 64 dependent adds");
    run<4>("64 lanes, reads + adds as asm blocks");
    run<5>("16 lanes, reads + adds as asm blocks");
    return 0;
}
