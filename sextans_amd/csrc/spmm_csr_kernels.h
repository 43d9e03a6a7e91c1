// spmm_csr_kernels.h -- CDNA4 (gfx950) kernels that replace the TAPA/HLS processing elements of
// the reference accelerator (src/sextans.cpp:836-984, task inventory in SURVEY.md 2.2).
//
// Mapping of the reference's dataflow onto a 64-lane wavefront machine:
//
//   reference (FPGA)                                   here (MI355X)
//   -----------------------------------------------    ------------------------------------------
//   read_A: packed non-zero stream, re-read per         CSR col_idx/val streamed with coalesced
//     8-column N tile (sextans.cpp:75-100)                global loads into LDS, once per N tile
//   read_B + PEG_Bmtx local_B window                    B repacked to row-major N-tile panels
//     (sextans.cpp:102-126, 337, 353-381)                 (repack_b_panels), gathered with one
//                                                         16-byte load per lane per non-zero
//   PEcore_Bmtx: val * B[col][0..7] (:285-295)          per-lane float4 multiply
//   PEG_Cmtx URAM accumulators, row%64 -> PE            per-row register accumulators: LPR lanes
//     (sextans.cpp:425-460, sparse_helper.h:370)          own one row, each lane 4 columns
//   FloatvMultConst x2 + FloatvAddFloatv                fused epilogue alpha*acc + beta*c_in,
//     (sextans.cpp:196-233), read_C, write_C              C tile transposed through LDS so the
//                                                         column-major loads/stores coalesce
//
// Arithmetic order (parity): a row's products are formed and added in ascending CSR order by ONE
// lane per output element, each product rounded to fp32 before the add when EXACT (the order and
// rounding of cpu_spmm_CSR, sparse_helper.h:279-289, which SURVEY.md 3.3 shows is also the
// accelerator's).  The TU is compiled with -ffp-contract=off; the non-exact variant calls fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sx {

constexpr int kBlock = 256;   // 4 wavefronts

// Bijective XCD-aware remap of a linear workgroup id: the dispatcher places workgroup b on XCD
// b % 8, so giving each XCD a contiguous chunk of logical ids keeps neighbouring row blocks
// (which share B rows in banded matrices, and the A stream across N tiles) on one L2.  Speed
// only -- correctness never depends on placement.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned xcd = b & 7u, i = b >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

template <bool EXACT>
__device__ __forceinline__ float mac(float acc, float a, float b) {
    if constexpr (EXACT) {
        const float p = a * b;   // rounded product (-ffp-contract=off), then rounded add
        return acc + p;
    } else {
        return __builtin_fmaf(a, b, acc);
    }
}

template <bool EXACT>
__device__ __forceinline__ void mac4(float4 &acc, float a, const float4 &b) {
    acc.x = mac<EXACT>(acc.x, a, b.x);
    acc.y = mac<EXACT>(acc.y, a, b.y);
    acc.z = mac<EXACT>(acc.z, a, b.z);
    acc.w = mac<EXACT>(acc.w, a, b.w);
}

template <bool EXACT>
__device__ __forceinline__ float epilogue(float alpha, float acc, float beta, float cin) {
    if constexpr (EXACT) {
        const float t0 = alpha * acc;   // FloatvMultConst (sextans.cpp:196-216): separate multiplies
        const float t1 = beta * cin;
        return t0 + t1;                 // FloatvAddFloatv (sextans.cpp:218-233)
    } else {
        return __builtin_fmaf(alpha, acc, beta * cin);
    }
}

// ------------------------------------------------------------------------------------------------
// Row-group gather kernel.
//   LPR   lanes per row; each lane owns 4 consecutive output columns -> N tile NT = 4*LPR.
//   CH    non-zeros staged in LDS per chunk (8 bytes each).
//   Each workgroup owns RB = 256/LPR consecutive rows and one N tile.  Linear grid of
//   nrowblk * ntiles workgroups; logical id -> (rowblk = id / ntiles, tile = id % ntiles).
// Bp: panels, panel t is a row-major K x NT matrix at Bp + t*panel_stride.
// Cin/Cout: column-major, already offset to the first column of tile 0 of this launch.
// ------------------------------------------------------------------------------------------------
template <int LPR, int CH, bool EXACT, bool STAGE>
__global__ __launch_bounds__(kBlock) void spmm_csr_rowgroup(
    const int *__restrict__ row_ptr, const int *__restrict__ col_idx, const float *__restrict__ val,
    const float *__restrict__ Bp, int64_t panel_stride, const float *Cin, float *Cout, int64_t ldc,
    int M, int ntiles, int nrowblk, float alpha, float beta, int use_xcd_remap) {
    constexpr int NT = 4 * LPR;
    constexpr int RB = kBlock / LPR;
    constexpr int TS = RB + 1;   // padded row stride of the transposed C tile in LDS
    static_assert(CH * 2 >= NT * TS, "chunk buffer must also hold the C tile");
    __shared__ __attribute__((aligned(16))) int smem[CH * 2];

    const unsigned nwg = (unsigned)nrowblk * (unsigned)ntiles;
    unsigned wg = blockIdx.x;
    if (use_xcd_remap) wg = xcd_remap(wg, nwg);
    const int rowblk = (int)(wg / (unsigned)ntiles);
    const int tile = (int)(wg % (unsigned)ntiles);

    const int tid = threadIdx.x;
    const int slot = tid / LPR;
    const int q = tid % LPR;
    const int row0 = rowblk * RB;
    const int row = row0 + slot;

    const float *bq = Bp + (int64_t)tile * panel_stride + 4 * q;
    int j = 0, jend = 0;
    if (row < M) { j = row_ptr[row]; jend = row_ptr[row + 1]; }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    if constexpr (STAGE) {
        int2 *s_nz = reinterpret_cast<int2 *>(smem);
        const int bs = row_ptr[row0];
        const int be = row_ptr[min(row0 + RB, M)];
        for (int cs = bs; cs < be; cs += CH) {
            const int n = min(CH, be - cs);
            for (int i = tid; i < n; i += kBlock)
                s_nz[i] = make_int2(col_idx[cs + i], __float_as_int(val[cs + i]));
            __syncthreads();
            const int hi = min(jend, cs + n);
            while (j + 4 <= hi) {
                const int2 e0 = s_nz[j - cs], e1 = s_nz[j - cs + 1], e2 = s_nz[j - cs + 2],
                           e3 = s_nz[j - cs + 3];
                const float4 b0 = *reinterpret_cast<const float4 *>(bq + (int64_t)e0.x * NT);
                const float4 b1 = *reinterpret_cast<const float4 *>(bq + (int64_t)e1.x * NT);
                const float4 b2 = *reinterpret_cast<const float4 *>(bq + (int64_t)e2.x * NT);
                const float4 b3 = *reinterpret_cast<const float4 *>(bq + (int64_t)e3.x * NT);
                mac4<EXACT>(acc, __int_as_float(e0.y), b0);
                mac4<EXACT>(acc, __int_as_float(e1.y), b1);
                mac4<EXACT>(acc, __int_as_float(e2.y), b2);
                mac4<EXACT>(acc, __int_as_float(e3.y), b3);
                j += 4;
            }
            while (j < hi) {
                const int2 e = s_nz[j - cs];
                const float4 b = *reinterpret_cast<const float4 *>(bq + (int64_t)e.x * NT);
                mac4<EXACT>(acc, __int_as_float(e.y), b);
                ++j;
            }
            __syncthreads();
        }
    } else {
        // Direct variant: the LPR lanes of a row fetch LPR consecutive non-zeros with one load each
        // and exchange them with wave shuffles (no LDS staging, no workgroup barriers).
        while (j < jend) {
            const int mine = j + q;
            int c = 0;
            float a = 0.f;
            if (mine < jend) { c = col_idx[mine]; a = val[mine]; }
            const int cnt = min(LPR, jend - j);
            float4 b[LPR];
#pragma unroll
            for (int u = 0; u < LPR; ++u) {
                const int cu = __shfl(c, u, LPR);
                b[u] = (u < cnt) ? *reinterpret_cast<const float4 *>(bq + (int64_t)cu * NT)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < LPR; ++u) {
                const float au = __shfl(a, u, LPR);
                if (u < cnt) mac4<EXACT>(acc, au, b[u]);
            }
            j += LPR;
        }
    }

    // Transpose the RB x NT tile through LDS so that column-major C traffic is coalesced.
    float *s_c = reinterpret_cast<float *>(smem);
    s_c[(4 * q + 0) * TS + slot] = acc.x;
    s_c[(4 * q + 1) * TS + slot] = acc.y;
    s_c[(4 * q + 2) * TS + slot] = acc.z;
    s_c[(4 * q + 3) * TS + slot] = acc.w;
    __syncthreads();
    const int64_t col0 = (int64_t)tile * NT;
#pragma unroll
    for (int i = 0; i < (RB * NT) / kBlock; ++i) {
        const int e = tid + i * kBlock;
        const int n = e / RB, r = e % RB;
        const int orow = row0 + r;
        if (orow < M) {
            const int64_t o = (int64_t)orow + (col0 + n) * ldc;
            Cout[o] = epilogue<EXACT>(alpha, s_c[n * TS + r], beta, Cin[o]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// B repack: column-major K x N (leading dimension ldb) -> row-major panels of width W.
// Panel t (columns col_base + t*W ...) is written at Bp + t*K*W as K rows of W floats.  The
// reference does the equivalent re-layout for its HBM channels on the host
// (sextans-host.cpp:150-177); here it is one coalesced device pass.
// Workgroup = 256 consecutive k of one panel.
// ------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(kBlock) void repack_b_panels(const float *__restrict__ B, int64_t ldb,
                                                          float *__restrict__ Bp, int K,
                                                          int col_base) {
    __shared__ float s[W][kBlock + 1];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * kBlock;
    const int t = blockIdx.y;
    const int k = k0 + tid;
    const float *src = B + (int64_t)(col_base + t * W) * ldb;
    if (k < K) {
#pragma unroll
        for (int c = 0; c < W; ++c) s[c][tid] = src[(int64_t)c * ldb + k];
    }
    __syncthreads();
    float *dst = Bp + (int64_t)t * K * W + (int64_t)k0 * W;
    const int nk = min(kBlock, K - k0);
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int e = tid + i * kBlock;   // linear element of the 256 x W chunk
        const int kk = e / W, c = e % W;
        if (kk < nk) dst[e] = s[c][kk];
    }
}

}  // namespace sx
