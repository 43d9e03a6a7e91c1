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

#include "spmm_csr_kernels.h"   // xcd_remap

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

// ------------------------------------------------------------------------------------------------
// spmm_bell_mfma_shared -- N = 256, for matrices whose block rows SHARE block columns (block-banded / block-FEM
// structure): the case north_star's "MFMA only where a tile is actually dense ... MFMA utilisation against the roofline"
// is about.  The kernels above fetch every operand of every block from global memory per wavefront: with uniformly
// random block columns (BASELINE config 5) nothing else is possible, every 32x32 block needs its own 32x256 B tile and
// the launch is bound by line requests.  Here a workgroup owns kShRows (8) consecutive block rows (one wavefront
// each, 32 x 256 accumulators = 128 registers) and walks the UNION of their block columns: the 16 KiB B tile of a block
// column is copied ONCE per workgroup from global memory into an LDS ring (LDS-DMA, global_load_lds_dwordx4: no
// staging registers) and multiplied by up to kShRows A blocks, whose fragments come straight from HBM (read once,
// non-temporal).  Software pipeline, hand-counted: at step s the tile and the A fragments of step s + kShDepth are requested;
// every step issues exactly kShVmemPerStep vector-memory instructions per wavefront (dummy A loads when the wavefront's
// row has no block at that column), so `s_waitcnt vmcnt((kShDepth - 1) * kShVmemPerStep)` at the top of a step means "everything of
// step s has landed, the later steps may still be in flight" -- the compiler cannot count these (inline asm), which is the
// point: its own bookkeeping would drain the prefetches at every barrier (cdna_hip_programming.md, LDS-DMA section).
// ------------------------------------------------------------------------------------------------
constexpr int kShRows = 8;                 // block rows per workgroup = wavefronts (512 threads, one workgroup per CU)
constexpr int kShRing = 7;                 // B tiles in the LDS ring
constexpr int kShDepth = 6;                // a tile / A block is requested this many steps before it is used: ~5 x 1 k cycles of
                                           // MFMA work per step covers the HBM round trip (a 3-tile ring, requests 2 steps
                                           // ahead, measured 22 % MFMA utilisation: every step waited for memory)
constexpr int kShTileBytes = 32 * 256 * 2; // one block column of B for N = 256, fragment order: [ntile 8][kstep 2][lane 64][16 B]
constexpr int kShMaxRowCols = 2048;        // ELL entries of the 8 block rows of a workgroup (8 * ell_width <= 2048)
constexpr int kShMaxUnion = 1024;          // distinct block columns of a workgroup (checked by the engine before it picks this kernel)
constexpr int kShVmemPerStep = 4;          // per wavefront and step: 2 LDS-DMA pieces of the tile (1 KiB each) + 2 A fragments
constexpr int kShThreads = kShRows * 64;
static_assert(kShRing >= kShDepth + 1, "the slot requested at step s must not be the one being multiplied");

__device__ __forceinline__ void sh_dma_1k(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void sh_load_frag(bf16x8 &dst, const void *gsrc) {
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(gsrc) : "memory");
}

__global__ __launch_bounds__(kShThreads, 2) void spmm_bell_mfma_shared(
    const int *__restrict__ block_col, const bf16x8 *__restrict__ Af, const bf16x8 *__restrict__ Bf, const float *Cin,
    int64_t ldc_in, float *Cout, int64_t ldc, int mblocks, int ell_width, float alpha, float beta, int dbg) {
    // dbg (engine option "bell_debug", measurements only -- results are wrong): 1 = no global-memory requests in the
    // main loop, 2 = no MFMAs (the LDS reads stay), 4 = no per-step barrier, 8 = no main loop, 16 = no epilogue
    constexpr int NT8 = 8;
    constexpr int T = kShThreads;
    extern __shared__ __attribute__((aligned(16))) char sh_lds[];   // [ring: kShRing tiles][union list][ELL rows]
    int *ulist = reinterpret_cast<int *>(sh_lds + kShRing * kShTileBytes);       // [kShMaxUnion + kShDepth + 1]
    short *wslot = reinterpret_cast<short *>(ulist + kShMaxUnion + 8);              // [rows][kShMaxUnion + 8]: ELL slot of my row at step u, -1 = none
    int *rowcols = reinterpret_cast<int *>(wslot + kShRows * (kShMaxUnion + 8));    // [rows][ell_width]
    int *tmp = reinterpret_cast<int *>(sh_lds);                                     // the ring is free while the lists are built
    __shared__ int s_scan[T];
    __shared__ int s_nun;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware placement: workgroup b runs on XCD b % 8 and every XCD has its own L2, so neighbouring groups of block
    // rows -- which share all but 8 of their B tiles in a banded matrix -- must sit on the SAME XCD to find each other's
    // tiles in L2 (dispatch order: 7.5 % L2 hit rate, 152 M fabric requests per launch for 89 M algorithmic lines)
    const int br0 = (int)xcd_remap(blockIdx.x, gridDim.x) * kShRows;
    const int br = br0 + wave;
    const bool have_row = br < mblocks;

    // ---- union of the block columns of the workgroup's block rows, ascending.  Bitmap over the range of columns the
    // workgroup touches (one atomicOr per entry, prefix popcount, expand): ~10 barriers.  The first version sorted the
    // concatenated rows (bitonic, 66 barrier-separated stages): 0.65 ms of a 3.2 ms launch with one workgroup per CU and
    // nobody to overlap it with.  Ranges wider than the bitmap (uniformly random columns) still take the sort.
    const int n_in = kShRows * ell_width;
    __shared__ int s_min, s_max;
    if (tid == 0) { s_min = 0x7fffffff; s_max = -1; }
    __syncthreads();
    {
        int lo = 0x7fffffff, hi = -1;
        for (int i = tid; i < n_in; i += T) {
            int v = -1;
            if (br0 + i / ell_width < mblocks) v = block_col[(int64_t)(br0 + i / ell_width) * ell_width + i % ell_width];
            rowcols[i] = v;                                         // (rows past the matrix hold -1)
            if (v >= 0) { lo = min(lo, v); hi = max(hi, v); }
        }
        if (hi >= 0) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
    }
    __syncthreads();
    const int cmin = s_min, range = s_max >= 0 ? s_max - s_min + 1 : 0;
    constexpr int kBmWords = 1024;                                  // bitmap: 32768 block columns of range
    if (range <= kBmWords * 32) {
        unsigned *bm = reinterpret_cast<unsigned *>(tmp);
        int *pref = tmp + kBmWords;
        const int nwords = (range + 31) >> 5;
        for (int i = tid; i < nwords; i += T) bm[i] = 0u;
        __syncthreads();
        for (int i = tid; i < n_in; i += T) {
            const int v = rowcols[i];
            if (v >= 0) atomicOr(&bm[(v - cmin) >> 5], 1u << ((v - cmin) & 31));
        }
        __syncthreads();
        const int per = (nwords + T - 1) / T, w0 = tid * per, w1 = min(nwords, w0 + per);
        int cnt = 0;
        for (int w = w0; w < w1; ++w) cnt += __builtin_popcount(bm[w]);
        s_scan[tid] = cnt;
        __syncthreads();
        for (int d = 1; d < T; d <<= 1) {
            const int v = tid >= d ? s_scan[tid - d] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        int o = s_scan[tid] - cnt;
        for (int w = w0; w < w1; ++w) {
            pref[w] = o;
            unsigned bits = bm[w];
            while (bits) {
                const int bpos = __builtin_ctz(bits);
                bits &= bits - 1;
                if (o < kShMaxUnion) ulist[o] = cmin + 32 * w + bpos;
                ++o;
            }
        }
        if (tid == T - 1) s_nun = s_scan[T - 1];
    } else {
        int P = T;
        while (P < n_in) P <<= 1;
        for (int i = tid; i < P; i += T) tmp[i] = (i < n_in && rowcols[i] >= 0) ? rowcols[i] : 0x7fffffff;
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < P; i += T) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const int a = tmp[i], c = tmp[ixj];
                        if ((a > c) == ((i & k) == 0)) { tmp[i] = c; tmp[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        const int per = P / T, i0 = tid * per;
        int cnt = 0;
        for (int i = i0; i < i0 + per; ++i) cnt += (tmp[i] != 0x7fffffff && (i == 0 || tmp[i] != tmp[i - 1])) ? 1 : 0;
        s_scan[tid] = cnt;
        __syncthreads();
        for (int d = 1; d < T; d <<= 1) {
            const int v = tid >= d ? s_scan[tid - d] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        int w = s_scan[tid] - cnt;
        for (int i = i0; i < i0 + per; ++i)
            if (tmp[i] != 0x7fffffff && (i == 0 || tmp[i] != tmp[i - 1])) { if (w < kShMaxUnion) ulist[w] = tmp[i]; ++w; }
        if (tid == T - 1) s_nun = s_scan[T - 1];
    }
    __syncthreads();
    const int nun = min(s_nun, kShMaxUnion);   // (the engine only picks this kernel when every group's union fits)

    f32x16 acc[NT8];
#pragma unroll
    for (int t = 0; t < NT8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- per wavefront: which ELL slot of my block row (if any) sits at step u of the union walk.  Everything the main
    // loop needs to know about the structure is then two LDS words per step, read together with the B fragments (one
    // wait): the first version walked the ELL row with dependent LDS reads between the barrier and the first MFMA --
    // serial latency that no other wavefront could cover, because the barrier puts all of them in the same phase.
    short *my_slot = wslot + wave * (kShMaxUnion + 8);
    for (int i = lane; i < nun + kShDepth + 1; i += 64) my_slot[i] = -1;
    for (int i = tid; i < kShDepth + 1; i += T) ulist[nun + i] = nun > 0 ? ulist[nun - 1] : 0;   // steps past the end re-request the last tile
    for (int sl = lane; sl < ell_width; sl += 64) {
        const int c = rowcols[wave * ell_width + sl];
        if (c >= 0) {
            int lo = 0, hi = nun - 1;                       // position of c in the sorted union list
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (ulist[mid] < c) lo = mid + 1; else hi = mid; }
            my_slot[lo] = (short)sl;
        }
    }
    __syncthreads();
    const bf16x8 *a_row = Af + (int64_t)(have_row ? br : 0) * ell_width * 128 + lane;
    const unsigned ring0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)sh_lds);
    // this wavefront's eighth of a tile: 2 pieces of 1 KiB
    constexpr int kPieces = kShTileBytes / 1024 / kShRows;
    static_assert(kPieces + 2 == kShVmemPerStep, "vector-memory instructions per step");
    const char *bsrc_lane = reinterpret_cast<const char *>(Bf) + (size_t)wave * (kPieces * 1024) + (size_t)lane * 16;
    bf16x8 a0[kShRing], a1[kShRing];
    bool has[kShRing];
#pragma unroll
    for (int p = 0; p < kShRing; ++p) has[p] = false;
    // requests of one step (block column uc, my ELL slot sl or -1) into ring slot / register set p: always
    // kShVmemPerStep vector-memory instructions, dummy A loads when my row has no block there
    auto issue = [&](int uc, int sl, int p) {
        if (dbg & 1) { has[p] = sl >= 0; return; }
        const char *src = bsrc_lane + (size_t)uc * kShTileBytes;
#pragma unroll
        for (int k = 0; k < kPieces; ++k) sh_dma_1k(src + k * 1024, ring0 + p * kShTileBytes + wave * (kPieces * 1024) + k * 1024);
        const bf16x8 *ap = a_row + (int64_t)max(sl, 0) * 128;
        sh_load_frag(a0[p], ap);
        sh_load_frag(a1[p], ap + 64);
        has[p] = sl >= 0;
    };
    int uc_n = 0, sl_n = -1;                       // block column / ELL slot of the step that is issued next
    if (nun > 0 && !(dbg & 8)) {
#pragma unroll
        for (int d = 0; d < kShDepth; ++d)
            issue(__builtin_amdgcn_readfirstlane(ulist[d]), __builtin_amdgcn_readfirstlane((int)my_slot[d]), d);
        uc_n = __builtin_amdgcn_readfirstlane(ulist[kShDepth]);
        sl_n = __builtin_amdgcn_readfirstlane((int)my_slot[kShDepth]);
        for (int s = 0; s < nun; s += kShRing) {
#define SX_SH_STEP(p)                                                                                          \
            if (s + (p) < nun) {                                                                               \
                /* everything of step s+p has landed (mine); after the barrier: everybody's */                \
                asm volatile("s_waitcnt vmcnt(20)" : "+v"(a0[p]), "+v"(a1[p]) : : "memory");                    \
                static_assert(kShVmemPerStep * (kShDepth - 1) == 20, "the wait above is written out");          \
                if (!(dbg & 4)) __builtin_amdgcn_s_barrier();                                                  \
                asm volatile("" ::: "memory");                                                                 \
                /* LDS: the 16 fragments of this step's tile + the two structure words of the next request; the */ \
                /* global-memory requests of step s+p+depth are issued while those reads are in flight          */ \
                const char *tile = sh_lds + (p) * kShTileBytes + lane * 16;                                    \
                bf16x8 bf[2 * NT8];                                                                            \
                _Pragma("unroll") for (int i = 0; i < 2 * NT8; ++i)                                            \
                    bf[i] = *reinterpret_cast<const bf16x8 *>(tile + ((i % NT8) * 2 + i / NT8) * 1024);        \
                int v_uc = ulist[s + (p) + kShDepth + 1], v_sl = (int)my_slot[s + (p) + kShDepth + 1];          \
                asm volatile("" ::: "memory");                                                                 \
                issue(uc_n, sl_n, ((p) + kShDepth) % kShRing);                                                  \
                asm volatile("" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]), "+v"(bf[4]), "+v"(bf[5]), "+v"(bf[6]), \
                                  "+v"(bf[7]), "+v"(bf[8]), "+v"(bf[9]), "+v"(bf[10]), "+v"(bf[11]), "+v"(bf[12]),         \
                                  "+v"(bf[13]), "+v"(bf[14]), "+v"(bf[15]), "+v"(v_uc), "+v"(v_sl));                          \
                uc_n = __builtin_amdgcn_readfirstlane(v_uc);                                                   \
                sl_n = __builtin_amdgcn_readfirstlane(v_sl);                                                   \
                if (has[p] && !(dbg & 2)) {                                                                    \
                    _Pragma("unroll") for (int t = 0; t < NT8; ++t)                                            \
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[t], a0[p], acc[t], 0, 0, 0);       \
                    _Pragma("unroll") for (int t = 0; t < NT8; ++t)                                            \
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[NT8 + t], a1[p], acc[t], 0, 0, 0); \
                }                                                                                              \
            }
            SX_SH_STEP(0) SX_SH_STEP(1) SX_SH_STEP(2) SX_SH_STEP(3) SX_SH_STEP(4) SX_SH_STEP(5) SX_SH_STEP(6)
#undef SX_SH_STEP
            static_assert(kShRing == 7, "the step loop is written out for a ring of seven");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the over-issued steps: nothing of mine may still target LDS / registers
    if (!have_row || (dbg & 16)) return;
    // Epilogue: one workgroup per CU means nobody covers this phase, so its memory latency is paid as rarely as the
    // registers allow: C_in of FOUR column tiles (64 loads per lane) in flight at a time -- the A / B-fragment registers
    // of the main loop are free now -- instead of one round trip per tile.
    const int64_t m = (int64_t)br * 32 + (lane & 31);
#pragma unroll
    for (int t0 = 0; t0 < NT8; t0 += 4) {
        float cin[4][16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                cin[t][r] = Cin[m + (int64_t)((t0 + t) * 32 + nl) * ldc_in];
            }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float x0 = alpha * acc[t0 + t][r];
                const float x1 = beta * cin[t][r];
                Cout[m + (int64_t)((t0 + t) * 32 + nl) * ldc] = x0 + x1;
            }
    }
}

// Sharing factor of a blocked-ELL matrix: sum over groups of kShRows block rows of |union of their block columns|
// (one thread per group; rows sorted strictly ascending with -1 = empty: a k-way merge; anything else sets *irregular).
__global__ __launch_bounds__(256) void bell_union_count(const int *__restrict__ block_col, int mblocks, int ell_width,
                                                        unsigned long long *total_union, unsigned long long *total_blocks,
                                                        unsigned long long *max_union, unsigned long long *irregular) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int br0 = g * kShRows;
    if (br0 >= mblocks) return;
    int pos[kShRows];
#pragma unroll
    for (int r = 0; r < kShRows; ++r) pos[r] = 0;
    // The union walk (and spmm_bell_mfma_shared, which keeps ONE ELL slot per (block row, union position)) needs every block row's
    // block columns strictly ascending.  sextans_set_matrix_bell* does not demand that -- the per-wavefront kernels sum duplicate
    // and unsorted slots like any others -- so a row that lists a column twice or out of order marks the matrix irregular and the
    // engine keeps it off the shared kernel.
    for (int r = 0; r < kShRows; ++r) {
        if (br0 + r >= mblocks) continue;
        const int *row = block_col + (int64_t)(br0 + r) * ell_width;
        int prev = -1;
        bool bad = false;
        for (int i = 0; i < ell_width; ++i) {
            const int c = row[i];
            if (c < 0) continue;
            bad |= c <= prev;
            prev = c;
        }
        if (bad) { atomicMax(irregular, 1ull); return; }
    }
    unsigned long long uni = 0, blocks = 0;
    while (true) {
        int best = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < kShRows; ++r) {
            if (br0 + r >= mblocks) continue;
            const int *row = block_col + (int64_t)(br0 + r) * ell_width;
            while (pos[r] < ell_width && row[pos[r]] < 0) ++pos[r];
            if (pos[r] < ell_width) best = min(best, row[pos[r]]);
        }
        if (best == 0x7fffffff) break;
        ++uni;
#pragma unroll
        for (int r = 0; r < kShRows; ++r) {
            if (br0 + r >= mblocks) continue;
            const int *row = block_col + (int64_t)(br0 + r) * ell_width;
            while (pos[r] < ell_width && (row[pos[r]] < 0 || row[pos[r]] == best)) { blocks += row[pos[r]] == best; ++pos[r]; }
        }
    }
    atomicAdd(total_union, uni);
    atomicAdd(total_blocks, blocks);
    atomicMax(max_union, uni);
}

}  // namespace sx
