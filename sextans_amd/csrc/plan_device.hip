// plan_device.hip -- the packed row-bucketed form of A (panel_plan.h) built ON THE DEVICE.
//
// The reference schedules and packs its non-zero stream on the host before the accelerator runs
// (generate_edge_list_for_all_PEs + edge_list_64bit, sparse_helper.h:345-473, sextans-host.cpp:114-148); round 1/2 of
// this engine did the same: copy the matrix to the host, pack on all cores, upload -- 2.3 s for a 318 M non-zero
// matrix against a 0.7 ms SpMM.  Here the CSR arrays never leave HBM:
//   pass A  padded row lengths -> stream offset of every row (two-level scan);
//   pass B  block formation: one workgroup per PART of kPlanPartBlocks * RB rows walks its rows in order and grows a
//           block while the distinct columns (LDS hash set) fit the panel -- the same greedy rule as
//           build_panel_plan, whose parts are the same fixed row ranges, so the block list is identical;
//   pass C  one workgroup per BLOCK: hash the block's columns, sort them (bitonic, LDS) into its dictionary, write the
//           strided dictionary / per-slot row extents / 16-bit byte-offset stream the kernels read.
// Result: byte-identical to the host builder (tests/test_plan_device_gpu.py), in milliseconds.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <vector>

#include "plan_device.h"
#include "thread_stream.h"

namespace sx {
namespace {

constexpr int kHT = 4096;   // hash slots: >= largest panel capacity (1152 at 2 lanes per row) + one 256-entry chunk, load < 0.35
constexpr int kEmpty = -1;

__device__ __forceinline__ unsigned hash_col(int c) { return ((unsigned)c * 2654435761u) >> 20; }   // 12 bits

// returns true when `col` was not in the set yet
__device__ __forceinline__ bool hs_insert(int *keys, int *stamp, int col, int st) {
    unsigned h = hash_col(col) & (kHT - 1);
    while (true) {
        const int prev = atomicCAS(&keys[h], kEmpty, col);
        if (prev == kEmpty) { stamp[h] = st; return true; }
        if (prev == col) return false;
        h = (h + 1) & (kHT - 1);
    }
}

__device__ __forceinline__ int hs_find(const int *keys, int col) {
    unsigned h = hash_col(col) & (kHT - 1);
    while (keys[h] != col) h = (h + 1) & (kHT - 1);
    return (int)h;
}

// ---- pass A: padded lengths.  part_sum[p] = padded entries of rows [p*PR, (p+1)*PR)
__global__ __launch_bounds__(256) void plan_part_sums(const int *__restrict__ rp, int M, int PR, long long *part_sum) {
    __shared__ long long s[256];
    const int p = blockIdx.x, r0 = p * PR, r1 = min(M, r0 + PR);
    long long acc = 0;
    for (int r = r0 + threadIdx.x; r < r1; r += 256) acc += (long long)((rp[r + 1] - rp[r] + 3) & ~3);
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part_sum[p] = s[0];
}

// row_off[r] for the rows of part p (exclusive scan inside the part on top of part_base[p]); row_off[M] by the last part
__global__ __launch_bounds__(256) void plan_row_off(const int *__restrict__ rp, int M, int PR, const int *__restrict__ part_base,
                                                    int *__restrict__ row_off) {
    __shared__ int s[256];
    const int p = blockIdx.x, r0 = p * PR, r1 = min(M, r0 + PR);
    const int per = (PR + 255) / 256;
    const int a = min(r1, r0 + (int)threadIdx.x * per), b = min(r1, a + per);
    int acc = 0;
    for (int r = a; r < b; ++r) acc += (rp[r + 1] - rp[r] + 3) & ~3;
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {   // inclusive scan (Hillis-Steele)
        const int v = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0;
        __syncthreads();
        s[threadIdx.x] += v;
        __syncthreads();
    }
    int off = part_base[p] + s[threadIdx.x] - acc;
    for (int r = a; r < b; ++r) { row_off[r] = off; off += (rp[r + 1] - rp[r] + 3) & ~3; }
    if (r1 == M && threadIdx.x == 255) row_off[M] = part_base[p] + s[255];
}

// ---- pass B: block formation, one workgroup per part
__global__ __launch_bounds__(256) void plan_blocks(const int *__restrict__ rp, const int *__restrict__ ci, int M, int PR, int RB,
                                                   int cap, double min_reuse, int *__restrict__ pb_row, int *__restrict__ pb_cnt,
                                                   int *__restrict__ part_nblk, const unsigned char *__restrict__ cut, int *__restrict__ part_capcuts) {
    // part_capcuts: blocks of this part that ended because the next row's columns no longer fit the dictionary capacity
    // cut (may be null): rows at which a block MUST start (brick boundaries of the clustered row order, row_cluster.hip)
    __shared__ int keys[kHT], stamp[kHT];
    __shared__ int s_count;
    const int tid = threadIdx.x;
    const int p = blockIdx.x, part_begin = p * PR, part_end = min(M, part_begin + PR);
    int nb = 0, capcuts = 0;
    for (int r = part_begin; r < part_end;) {
        for (int i = tid; i < kHT; i += 256) keys[i] = kEmpty;
        if (tid == 0) s_count = 0;
        __syncthreads();
        int e = r, uniq = 0;
        bool fits = true;
        while (e < part_end && e - r < RB && !(cut && e > r && cut[e])) {
            const int before = uniq;
            bool over = false;
            const int j1 = rp[e + 1];
            for (int j0 = rp[e]; j0 < j1 && !over; j0 += 256) {   // (a chunk at a time: the set never holds more than cap + 256 keys)
                const int j = j0 + tid;
                if (j < j1 && hs_insert(keys, stamp, ci[j], e - r)) atomicAdd(&s_count, 1);
                __syncthreads();
                uniq = s_count;
                __syncthreads();                                  // everybody has read the count before it moves again
                over = uniq > cap;
            }
            if (over) {                                           // row e does not fit: its keys (stamp == e - r) are not part
                uniq = before;                                    // of the block; the set is rebuilt for the next block anyway
                if (e == r) fits = false;                         // a single row already exceeds the panel
                ++capcuts;
                break;
            }
            ++e;
        }
        if (e == r) e = r + 1;                                    // the oversized row forms a (direct) block of its own
        const long long n = (long long)rp[e] - rp[r];
        // (a block cut short by the end of its part is judged by fit alone: a one-row remnant has no reuse of its own,
        // and a single direct block would push the whole matrix onto the slower mixed kernel instantiation)
        const bool remnant = (e == part_end || (cut && e < part_end && cut[e])) && e - r < RB;   // (a forced cut counts as one)
        const bool use_dict = fits && uniq > 0 && ((double)n >= min_reuse * (double)uniq || remnant);
        if (tid == 0) { pb_row[(long long)p * PR + nb] = r; pb_cnt[(long long)p * PR + nb] = use_dict ? uniq : 0; }
        ++nb;
        r = e;
        __syncthreads();
    }
    if (tid == 0) { part_nblk[p] = nb; part_capcuts[p] = capcuts; }
}

// blocks of all parts -> contiguous arrays + plan statistics
__global__ __launch_bounds__(256) void plan_compact(const int *__restrict__ rp, int M, int PR, const int *__restrict__ pb_row,
                                                    const int *__restrict__ pb_cnt, const int *__restrict__ part_nblk,
                                                    const int *__restrict__ part_blk_base, int nblk, int *__restrict__ blk_row,
                                                    int *__restrict__ dict_cnt, int *stats /* max_dict, mixed, longest row */,
                                                    unsigned long long *nnz_panel /* [0] covered non-zeros, [1] sum of the dictionaries */) {
    const int p = blockIdx.x;
    const int nb = part_nblk[p], base = part_blk_base[p];
    int mx = 0, mixed = 0;
    unsigned long long covered = 0, dsum = 0;
    for (int i = threadIdx.x; i < nb; i += 256) {
        const int r0 = pb_row[(long long)p * PR + i];
        const int r1 = i + 1 < nb ? pb_row[(long long)p * PR + i + 1] : min(M, (p + 1) * PR);
        const int c = pb_cnt[(long long)p * PR + i];
        blk_row[base + i] = r0;
        dict_cnt[base + i] = c;
        const long long n = (long long)rp[r1] - rp[r0];
        mx = max(mx, c);
        dsum += (unsigned long long)c;
        if (c > 0) covered += (unsigned long long)n;
        else if (n > 0) mixed = 1;
    }
    if (p == 0 && threadIdx.x == 0) blk_row[nblk] = M;
    int mlen = 0;                                   // longest row of the part (the kernels size their register-resident batches by it)
    for (int r = p * PR + threadIdx.x; r < min(M, (p + 1) * PR); r += 256) mlen = max(mlen, rp[r + 1] - rp[r]);
    if (mlen) atomicMax(&stats[2], mlen);
    if (mx) atomicMax(&stats[0], mx);
    if (mixed) atomicOr(&stats[1], 1);
    if (covered) atomicAdd(nnz_panel, covered);
    if (dsum) atomicAdd(nnz_panel + 1, dsum);
}

// ---- pass C: one workgroup per block
template <int P>   // P = power of two >= panel capacity (sort width)
__global__ __launch_bounds__(256) void plan_emit(const int *__restrict__ rp, const int *__restrict__ ci, const float *__restrict__ va,
                                                 const int *__restrict__ row_off, const int *__restrict__ blk_row,
                                                 const int *__restrict__ dict_cnt, int RB, int dstride, unsigned row_bytes,
                                                 unsigned pad_off, int *__restrict__ bdict, int *__restrict__ slot_info,
                                                 unsigned short *__restrict__ idx16, int *__restrict__ col32, float *__restrict__ pval, int sets) {
    // sets: a block has sets * RB row slots (one dictionary; the kernel walks them RB at a time)
    __shared__ int keys[kHT], rankv[kHT];
    __shared__ int sorted[P];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int r0 = blk_row[b], r1 = blk_row[b + 1], nu = dict_cnt[b];
    const int lpr = 256 / RB, q = tid % lpr;
    // per-slot row extents {first packed entry, entries}: dictionary rows are consumed in whole groups of 4 entries
    // (exact-safe padding), direct rows keep their true length; slots past the block's last row stay {0, 0}
    if (q == 0)
        for (int slot = tid / lpr; slot < RB * sets; slot += RB) {
            int2 si = make_int2(0, 0);
            if (r0 + slot < r1) {
                const int len = rp[r0 + slot + 1] - rp[r0 + slot];
                si = make_int2(row_off[r0 + slot], nu > 0 ? (len + 3) & ~3 : len);
            }
            reinterpret_cast<int2 *>(slot_info)[(long long)b * RB * sets + slot] = si;
        }
    if (nu == 0) {   // direct block: 32-bit columns, B rows gathered from global memory by the kernel
        for (int i = tid; i < dstride; i += 256) bdict[(long long)b * dstride + i] = 0;
        for (int slot = tid / lpr; slot < RB * sets; slot += RB)
            if (r0 + slot < r1) {
                const int row = r0 + slot, j0 = rp[row], len = rp[row + 1] - j0, o0 = row_off[row];
                for (int e = q; e < len; e += lpr) { pval[o0 + e] = va[j0 + e]; col32[o0 + e] = ci[j0 + e]; }
            }
        return;
    }
    for (int i = tid; i < kHT; i += 256) keys[i] = kEmpty;
    for (int i = tid; i < P; i += 256) sorted[i] = INT_MAX;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int jb = rp[r0], je = rp[r1];
    for (int j = jb + tid; j < je; j += 256) {
        const int c = ci[j];
        if (hs_insert(keys, rankv, c, 0)) sorted[atomicAdd(&s_n, 1)] = c;   // (s_n ends at nu: pass B counted the same set)
    }
    __syncthreads();
    // bitonic sort, ascending, P elements
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int a = sorted[i], c = sorted[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { sorted[i] = c; sorted[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // dictionary at a fixed stride per block, last column repeated; rank of every column for the re-encoding
    for (int i = tid; i < dstride; i += 256) bdict[(long long)b * dstride + i] = sorted[min(i, nu - 1)];
    for (int i = tid; i < nu; i += 256) rankv[hs_find(keys, sorted[i])] = i;
    __syncthreads();
    for (int slot = tid / lpr; slot < RB * sets; slot += RB) {
        if (r0 + slot >= r1) break;
        const int row = r0 + slot, j0 = rp[row], len = rp[row + 1] - j0, o0 = row_off[row], plen = (len + 3) & ~3;
        for (int e = q; e < plen; e += lpr) {
            if (e < len) {
                pval[o0 + e] = va[j0 + e];
                idx16[o0 + e] = (unsigned short)((unsigned)rankv[hs_find(keys, ci[j0 + e])] * row_bytes);
            } else {   // padding consumed by the kernel: -0.0f against the +1.0f panel row
                pval[o0 + e] = -0.0f;
                idx16[o0 + e] = (unsigned short)pad_off;
            }
        }
    }
}

// ---- index-list sharing: consecutive rows of a block whose 16-bit index lists are identical UP TO A CONSTANT SHIFT keep ONE copy of
// the list.  Shift 0: the dof rows of one mesh node (same columns, hence the same dictionary ranks).  Shift > 0: the next node along a
// grid line -- every column moves by the same number of dictionary rows, so every 16-bit byte offset moves by the same multiple of
// the panel's row size; the kernel adds the slot's shift to its LDS base address once, the inner loop is unchanged.  Padding entries
// (the +1.0f row behind the dictionary) are shifted with the rest, so the panel carries kPlanPadRows such rows and a chain of shared
// lists never shifts further than that.  The values stay per row; the index stream shrinks from 2 bytes per non-zero to 2 / (rows per
// list): the packed form from 6 to ~4.2 bytes per non-zero on grid-ordered matrices, 4.67 on renumbered 3-dof ones -- the LDS-panel
// kernels run within 20-25 % of the achievable HBM bandwidth and the A stream is 60 % of what they move.
// one wavefront per (block, slot): is this slot's list the previous slot's, shifted by `step` bytes?
__global__ __launch_bounds__(256) void share_detect(long long nslots, int RB, unsigned pad_off, unsigned row_bytes, const int2 *__restrict__ slot_info,
                                                    const unsigned short *__restrict__ idx16, int *__restrict__ cand, int *__restrict__ step) {
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= nslots) return;
    const int2 me = slot_info[i];
    bool eq = false;
    int d = 0;
    if ((i % RB) != 0 && me.y > 0) {
        const int2 pr = slot_info[i - 1];
        if (pr.y == me.y) {
            d = (int)idx16[me.x] - (int)idx16[pr.x];             // (entry 0 is never padding: rows are padded at their end)
            bool diff = false;
            for (int e = lane; e < me.y; e += 64) {
                const unsigned a = idx16[(long long)me.x + e], b = idx16[(long long)pr.x + e];
                diff |= (a == pad_off || b == pad_off) ? a != b : (int)a - (int)b != d;
            }
            eq = __ballot(diff) == 0ull && d >= 0 && (unsigned)d % row_bytes == 0u;
        }
    }
    if (lane == 0) { cand[i] = eq ? 1 : 0; step[i] = eq ? d : 0; }
}
// one thread per block: chains of shared lists, total shift bounded by the spare +1.0f rows
__global__ __launch_bounds__(256) void share_chain(int nblk, int RB, int max_shift, const int2 *__restrict__ slot_info, const int *__restrict__ cand,
                                                   const int *__restrict__ step, int *__restrict__ root, int *__restrict__ shift,
                                                   int *__restrict__ own_len) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    long long r = (long long)b * RB;
    int cum = 0;
    for (int s = 0; s < RB; ++s) {
        const long long i = (long long)b * RB + s;
        if (s > 0 && cand[i] && cum + step[i] <= max_shift) { cum += step[i]; }
        else { r = i; cum = 0; }
        root[i] = (int)(r - (long long)b * RB);
        shift[i] = cum;
        own_len[i] = r == i ? slot_info[i].y : 0;
    }
}
__global__ __launch_bounds__(256) void share_compact(long long nslots, int RB, const int2 *__restrict__ slot_info, const int *__restrict__ root,
                                                     const int *__restrict__ shift, const int *__restrict__ noff,
                                                     const unsigned short *__restrict__ idx16, unsigned short *__restrict__ out, int2 *__restrict__ ioff) {
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= nslots) return;
    const long long ri = i / RB * RB + root[i];
    if (lane == 0) ioff[i] = make_int2(noff[ri], shift[i]);
    if (ri != i) return;
    const int2 me = slot_info[i];
    for (int e = lane; e < me.y; e += 64) out[(long long)noff[i] + e] = idx16[(long long)me.x + e];
}

#define PD_HIP(call)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            char buf_[256];                                                                                    \
            snprintf(buf_, sizeof buf_, "%s failed: %s (plan_device.hip:%d)", #call, hipGetErrorString(e_), __LINE__); \
            err = buf_;                                                                                        \
            return 2;                                                                                          \
        }                                                                                                      \
    } while (0)

struct Scratch {   // freed on every exit path
    std::vector<void *> p;
    ~Scratch() { for (void *q : p) (void)hipFree(q); }
    template <class T> hipError_t alloc(T **out, size_t n) {
        hipError_t e = hipMalloc((void **)out, sizeof(T) * (n ? n : 1));
        if (e == hipSuccess) p.push_back(*out);
        return e;
    }
};

}  // namespace

namespace {
__global__ __launch_bounds__(256) void csr_validate(const int *__restrict__ rp, const int *__restrict__ ci, int M, int K, long long nnz,
                                                    int *bad) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    int b = 0;
    if (t == 0 && (rp[0] != 0 || (long long)rp[M] != nnz)) b |= 1;
    for (long long r = t; r < M; r += stride)
        if (rp[r + 1] < rp[r]) b |= 1;
    for (long long j = t; j < nnz; j += stride)
        if ((unsigned)ci[j] >= (unsigned)K) b |= 2;
    if (b) atomicOr(bad, b);
}
}  // namespace

int validate_csr_device(int M, int K, int64_t nnz, const int *d_rp, const int *d_ci, int *bad, std::string &err) {
    *bad = 0;
    int *d_bad = nullptr;
    PD_HIP(hipMalloc((void **)&d_bad, sizeof(int)));
    hipError_t e1 = hipMemset(d_bad, 0, sizeof(int));
    hipLaunchKernelGGL(csr_validate, dim3(2048), dim3(256), 0, nullptr, d_rp, d_ci, M, K, (long long)nnz, d_bad);
    hipError_t e2 = hipMemcpy(bad, d_bad, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    PD_HIP(e1);
    PD_HIP(e2);
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void col_minmax(const int *__restrict__ ci, long long nnz, int *lo, int *hi) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    int a = 0x7fffffff, b = -1;
    for (long long j = t; j < nnz; j += stride) { const int c = ci[j]; a = min(a, c); b = max(b, c); }
    for (int off = 32; off > 0; off >>= 1) { a = min(a, __shfl_xor(a, off)); b = max(b, __shfl_xor(b, off)); }
    if ((threadIdx.x & 63) == 0 && b >= 0) { atomicMin(lo, a); atomicMax(hi, b); }
}
}  // namespace

// smallest and largest column index of a device CSR matrix (lo > hi when it has no entries)
int column_range_device(int64_t nnz, const int *d_ci, int *lo, int *hi, std::string &err) {
    *lo = 0x7fffffff; *hi = -1;
    if (nnz <= 0) return 0;
    int *d = nullptr;
    PD_HIP(hipMalloc((void **)&d, 2 * sizeof(int)));
    const int init[2] = {0x7fffffff, -1};
    hipError_t e1 = hipMemcpy(d, init, sizeof init, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(col_minmax, dim3(2048), dim3(256), 0, nullptr, d_ci, (long long)nnz, d, d + 1);
    int out[2] = {0x7fffffff, -1};
    hipError_t e2 = hipMemcpy(out, d, sizeof out, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    PD_HIP(e1);
    PD_HIP(e2);
    *lo = out[0]; *hi = out[1];
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void col_touch(const int *__restrict__ ci, long long nnz, unsigned char *__restrict__ flag) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    for (long long j = t; j < nnz; j += stride) flag[ci[j] >> 6] = 1;     // (every writer stores the same value)
}
}  // namespace

// flag[k / 64] = 1 where the matrix has an entry in columns [64 (k / 64), 64 (k / 64) + 64): the 64-row segments of B a call has to
// repack.  ceil(K / 64) + 4 bytes on the device (the tail zero), caller frees.
int column_touch_flags_device(int K, int64_t nnz, const int *d_ci, unsigned char **d_flag, int64_t *touched_segments, std::string &err) {
    *d_flag = nullptr;
    if (touched_segments) *touched_segments = 0;
    const size_t n = (size_t)(K + 63) / 64 + 4;
    unsigned char *f = nullptr;
    PD_HIP(hipMalloc((void **)&f, n));
    hipError_t e = hipMemsetAsync(f, 0, n, nullptr);
    if (nnz > 0) hipLaunchKernelGGL(col_touch, dim3(2048), dim3(256), 0, nullptr, d_ci, (long long)nnz, f);
    std::vector<unsigned char> host(n);
    hipError_t e2 = hipMemcpy(host.data(), f, n, hipMemcpyDeviceToHost);
    if (e != hipSuccess || e2 != hipSuccess) { (void)hipFree(f); PD_HIP(e); PD_HIP(e2); }
    if (touched_segments) for (unsigned char x : host) *touched_segments += x;
    *d_flag = f;
    return 0;
}

void free_device_plan(DevicePlan &d) {
    (void)hipFree(d.d_blk_row); (void)hipFree(d.d_dict_cnt); (void)hipFree(d.d_dict); (void)hipFree(d.d_slot_info);
    (void)hipFree(d.d_idx16); (void)hipFree(d.d_col32); (void)hipFree(d.d_val); (void)hipFree(d.d_ioff);
    d = DevicePlan();
}

// 0 = built; 1 = not representable (padded stream exceeds 32-bit entry offsets): caller keeps the row-group kernel;
// 2 = HIP error (err set)
int build_panel_plan_device(int M, int K, const int *d_rp, const int *d_ci, const float *d_v, int lpr, int max_unique,
                            double min_reuse, DevicePlan &out, std::string &err, const unsigned char *d_cut, bool share_index_lists, int sets) {
    (void)K;
    free_device_plan(out);
    if (sets < 1 || (sets > 1 && lpr != 4)) { err = "row sets per block: 4 lanes per row only"; return 2; }
    const int RB = 256 / lpr;           // row slots per set (= per workgroup pass)
    const int RBS = RB * sets;          // row slots per block
    const int PR = kPlanPartBlocks * RBS;
    const int nparts = M > 0 ? (M + PR - 1) / PR : 0;
    out.rows_per_block = RBS;
    out.sets = sets;
    out.lpr = lpr;
    const unsigned row_bytes = 16u * (unsigned)lpr;
    const unsigned pad_off = (unsigned)max_unique * row_bytes;   // the +1.0f row sits right behind a full dictionary
    if (pad_off > 0xffffu) { err = "panel capacity does not fit 16-bit byte offsets"; return 2; }
    Scratch tmp;
    if (M == 0) {
        PD_HIP(hipMalloc((void **)&out.d_blk_row, sizeof(int)));
        PD_HIP(hipMemset(out.d_blk_row, 0, sizeof(int)));
        PD_HIP(hipMalloc((void **)&out.d_dict_cnt, sizeof(int)));
        PD_HIP(hipMalloc((void **)&out.d_dict, sizeof(int)));
        PD_HIP(hipMalloc((void **)&out.d_slot_info, sizeof(int)));
        PD_HIP(hipMalloc((void **)&out.d_idx16, sizeof(unsigned short) * kPlanTailPad));
        PD_HIP(hipMalloc((void **)&out.d_col32, sizeof(int)));
        PD_HIP(hipMalloc((void **)&out.d_val, sizeof(float) * kPlanTailPad));
        PD_HIP(hipMemset(out.d_idx16, 0, sizeof(unsigned short) * kPlanTailPad));
        PD_HIP(hipMemset(out.d_val, 0, sizeof(float) * kPlanTailPad));
        out.h_blk_row.assign(1, 0);
        out.dict_stride = RB;
        out.stream_len = kPlanTailPad;
        return 0;
    }
    // ---- pass A
    long long *d_part_sum = nullptr;
    int *d_part_base = nullptr, *d_row_off = nullptr;
    PD_HIP(tmp.alloc(&d_part_sum, (size_t)nparts));
    PD_HIP(tmp.alloc(&d_part_base, (size_t)nparts));
    PD_HIP(tmp.alloc(&d_row_off, (size_t)M + 1));
    hipLaunchKernelGGL(plan_part_sums, dim3((unsigned)nparts), dim3(256), 0, nullptr, d_rp, M, PR, d_part_sum);
    std::vector<long long> h_sum((size_t)nparts);
    PD_HIP(hipMemcpy(h_sum.data(), d_part_sum, sizeof(long long) * (size_t)nparts, hipMemcpyDeviceToHost));
    std::vector<int> h_base((size_t)nparts);
    long long total = 0;
    for (int p = 0; p < nparts; ++p) { h_base[(size_t)p] = (int)total; total += h_sum[(size_t)p]; }
    if (total > 0x7fffffffLL - 4096) return 1;
    PD_HIP(hipMemcpy(d_part_base, h_base.data(), sizeof(int) * (size_t)nparts, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(plan_row_off, dim3((unsigned)nparts), dim3(256), 0, nullptr, d_rp, M, PR, d_part_base, d_row_off);
    // ---- pass B
    int *d_pb_row = nullptr, *d_pb_cnt = nullptr, *d_part_nblk = nullptr, *d_part_blk_base = nullptr;
    PD_HIP(tmp.alloc(&d_pb_row, (size_t)nparts * (size_t)PR));
    PD_HIP(tmp.alloc(&d_pb_cnt, (size_t)nparts * (size_t)PR));
    PD_HIP(tmp.alloc(&d_part_nblk, (size_t)nparts));
    PD_HIP(tmp.alloc(&d_part_blk_base, (size_t)nparts));
    int *d_part_capcuts = nullptr;
    PD_HIP(tmp.alloc(&d_part_capcuts, (size_t)nparts));
    hipLaunchKernelGGL(plan_blocks, dim3((unsigned)nparts), dim3(256), 0, nullptr, d_rp, d_ci, M, PR, RBS, max_unique, min_reuse,
                       d_pb_row, d_pb_cnt, d_part_nblk, d_cut, d_part_capcuts);
    std::vector<int> h_nblk((size_t)nparts), h_bbase((size_t)nparts), h_capcuts((size_t)nparts);
    PD_HIP(hipMemcpy(h_nblk.data(), d_part_nblk, sizeof(int) * (size_t)nparts, hipMemcpyDeviceToHost));
    PD_HIP(hipMemcpy(h_capcuts.data(), d_part_capcuts, sizeof(int) * (size_t)nparts, hipMemcpyDeviceToHost));
    out.capacity_cuts = 0;
    for (int c : h_capcuts) out.capacity_cuts += c;
    long long nblk = 0;
    for (int p = 0; p < nparts; ++p) { h_bbase[(size_t)p] = (int)nblk; nblk += h_nblk[(size_t)p]; }
    PD_HIP(hipMemcpy(d_part_blk_base, h_bbase.data(), sizeof(int) * (size_t)nparts, hipMemcpyHostToDevice));
    out.nblk = (int)nblk;
    int *d_stats = nullptr;
    unsigned long long *d_cov = nullptr;
    PD_HIP(tmp.alloc(&d_stats, 3));
    PD_HIP(tmp.alloc(&d_cov, 2));
    PD_HIP(hipMemset(d_stats, 0, 3 * sizeof(int)));
    PD_HIP(hipMemset(d_cov, 0, 2 * sizeof(unsigned long long)));
    PD_HIP(hipMalloc((void **)&out.d_blk_row, sizeof(int) * ((size_t)nblk + 1)));
    PD_HIP(hipMalloc((void **)&out.d_dict_cnt, sizeof(int) * (size_t)nblk));
    hipLaunchKernelGGL(plan_compact, dim3((unsigned)nparts), dim3(256), 0, nullptr, d_rp, M, PR, d_pb_row, d_pb_cnt, d_part_nblk,
                       d_part_blk_base, (int)nblk, out.d_blk_row, out.d_dict_cnt, d_stats, d_cov);
    int h_stats[3] = {0, 0, 0};
    unsigned long long h_cov[2] = {0, 0};
    PD_HIP(hipMemcpy(h_stats, d_stats, sizeof h_stats, hipMemcpyDeviceToHost));
    PD_HIP(hipMemcpy(h_cov, d_cov, sizeof h_cov, hipMemcpyDeviceToHost));
    out.max_dict = h_stats[0];
    out.mixed = h_stats[1] != 0;
    out.max_row_len = h_stats[2];
    out.nnz_in_panel_blocks = (int64_t)h_cov[0];
    out.total_dict = (int64_t)h_cov[1];
    out.h_blk_row.resize((size_t)nblk + 1);
    PD_HIP(hipMemcpy(out.h_blk_row.data(), out.d_blk_row, sizeof(int) * ((size_t)nblk + 1), hipMemcpyDeviceToHost));
    // ---- pass C
    int dstride = ((out.max_dict + RB - 1) / RB) * RB;
    if (dstride < RB) dstride = RB;
    out.dict_stride = dstride;
    const size_t stream = (size_t)total + kPlanTailPad;
    out.stream_len = (int64_t)stream;
    PD_HIP(hipMalloc((void **)&out.d_dict, sizeof(int) * (size_t)nblk * (size_t)dstride));
    PD_HIP(hipMalloc((void **)&out.d_slot_info, sizeof(int) * (size_t)nblk * (size_t)RBS * 2));
    PD_HIP(hipMalloc((void **)&out.d_idx16, sizeof(unsigned short) * stream));
    PD_HIP(hipMalloc((void **)&out.d_val, sizeof(float) * stream));
    PD_HIP(hipMalloc((void **)&out.d_col32, sizeof(int) * (out.mixed ? stream : 1)));   // only direct blocks read it
    PD_HIP(hipMemsetAsync(out.d_idx16, 0, sizeof(unsigned short) * stream, nullptr));
    PD_HIP(hipMemsetAsync(out.d_val, 0, sizeof(float) * stream, nullptr));
    if (out.mixed) PD_HIP(hipMemsetAsync(out.d_col32, 0, sizeof(int) * stream, nullptr));
    auto emit = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), 0, nullptr, d_rp, d_ci, d_v, d_row_off, out.d_blk_row,
                           out.d_dict_cnt, RB, dstride, row_bytes, pad_off, out.d_dict, out.d_slot_info, out.d_idx16,
                           out.d_col32, out.d_val, sets);
    };
    if (max_unique <= 512) emit(plan_emit<512>);
    else if (max_unique <= 1024) emit(plan_emit<1024>);
    else emit(plan_emit<2048>);
    PD_HIP(hipGetLastError());
    PD_HIP(hipDeviceSynchronize());
    out.idx_len = out.stream_len;
    if (share_index_lists && !out.mixed && nblk > 0 && lpr == 4) {
        const long long nslots = (long long)nblk * RBS;
        int *d_cand = nullptr, *d_step = nullptr, *d_root = nullptr, *d_shift = nullptr, *d_len = nullptr, *d_noff = nullptr;
        PD_HIP(tmp.alloc(&d_cand, (size_t)nslots));
        PD_HIP(tmp.alloc(&d_step, (size_t)nslots));
        PD_HIP(tmp.alloc(&d_root, (size_t)nslots));
        PD_HIP(tmp.alloc(&d_shift, (size_t)nslots));
        PD_HIP(tmp.alloc(&d_len, (size_t)nslots + 1));
        PD_HIP(tmp.alloc(&d_noff, (size_t)nslots + 1));
        PD_HIP(hipMemsetAsync(d_len + nslots, 0, sizeof(int), nullptr));
        hipLaunchKernelGGL(share_detect, dim3((unsigned)((nslots + 3) / 4)), dim3(256), 0, nullptr, nslots, RBS, pad_off, row_bytes,
                           (const int2 *)out.d_slot_info, out.d_idx16, d_cand, d_step);
        hipLaunchKernelGGL(share_chain, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, nullptr, (int)nblk, RBS,
                           (int)((kPlanPadRows - 1) * row_bytes), (const int2 *)out.d_slot_info, d_cand, d_step, d_root, d_shift, d_len);
        void *scan_tmp = nullptr;
        size_t bytes = 0;
        PD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_len, d_noff, (int)(nslots + 1), nullptr));
        PD_HIP(tmp.alloc((char **)&scan_tmp, bytes));
        PD_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, bytes, d_len, d_noff, (int)(nslots + 1), nullptr));
        int kept = 0;
        PD_HIP(hipMemcpy(&kept, d_noff + nslots, sizeof(int), hipMemcpyDeviceToHost));
        if ((long long)kept * 10 <= (long long)total * 9) {   // at least a tenth of the index stream goes: worth one more table
            unsigned short *nidx = nullptr;
            PD_HIP(hipMalloc((void **)&nidx, sizeof(unsigned short) * ((size_t)kept + kPlanTailPad)));
            hipError_t e = hipMalloc((void **)&out.d_ioff, sizeof(int2) * (size_t)nslots);
            if (e != hipSuccess) { (void)hipFree(nidx); PD_HIP(e); }
            PD_HIP(hipMemsetAsync(nidx, 0, sizeof(unsigned short) * ((size_t)kept + kPlanTailPad), nullptr));
            hipLaunchKernelGGL(share_compact, dim3((unsigned)((nslots + 3) / 4)), dim3(256), 0, nullptr, nslots, RBS, (const int2 *)out.d_slot_info,
                               d_root, d_shift, d_noff, out.d_idx16, nidx, (int2 *)out.d_ioff);
            e = hipDeviceSynchronize();
            if (e != hipSuccess) { (void)hipFree(nidx); PD_HIP(e); }
            (void)hipFree(out.d_idx16);
            out.d_idx16 = nidx;
            out.idx_len = (int64_t)kept + kPlanTailPad;
        }
    }
    return 0;
}

}  // namespace sx
