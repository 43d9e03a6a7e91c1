// graph_cluster.hip -- structure-agnostic row clustering for the LDS-panel plan, built on the device.
//
// Why: a row block's LDS panel is as large as the block's dictionary (its distinct columns) and is copied once per 16-column N
// tile, and a block is a run of CONSECUTIVE rows of the plan's row order.  When the numbering of the matrix has no locality (an
// arbitrary node ordering of a mesh: SuiteSparse files carry whatever order their generator wrote) 64 consecutive rows share
// nothing: 22 rows fill the 576-row panel, every dictionary row is a separate 128-byte fabric request for 64 useful bytes, and the
// kernel runs at a third of its speed.  The rows of a matrix are independent -- ANY order of the rows gives the same sums, bit for
// bit -- so the plan may visit them in an order in which consecutive rows are neighbours in the matrix graph.  The reference
// schedules its non-zeros for the same purpose, keeping the on-chip B window hot (generate_edge_list_for_all_PEs,
// sparse_helper.h:345-403: row % 64 interleaving over PEs, 4096-column windows).
//
// How (cluster_rows_graph_device): multilevel pairwise aggregation, the coarsening half of a multilevel graph partitioner.
//   weights   t(r, c) = 1 + |cols(r) & cols(c)| for every non-zero (r, c), c read as a row (M == K): the neighbourhood two
//             adjacent rows share ("triangles on the edge").  With unit weights every neighbour of a stencil row ties and the
//             pairs are random (face / edge / corner neighbours alike): measured on a randomly renumbered 3-dof 27-point mesh,
//             9.8 dictionary rows per matrix row against 7.1 with these weights (natural grid order: 9.3; hand-made bricks: 6.2).
//   level     every cluster accumulates, in an LDS hash table, the weight of its non-zeros towards each neighbouring cluster
//             that still fits (size(A) + size(B) <= 2, 4, 8 ... rows) and keeps its four best partners by
//             weight / sqrt(size(B)); five handshake rounds then pair clusters that choose each other (first still-unmatched
//             candidate each round); a pair's rows become contiguous in the order.  Sizes double per level, so 64-row blocks
//             take six levels and the order keeps refining up to `max_cluster_rows` (neighbouring blocks of the order are
//             neighbours in the graph: they run on one XCD at about the same time and share B lines in its L2).
//   orient    each member of a merged pair may be laid down reversed: the halves that become adjacent in the order are the two
//             through which the clusters are connected most strongly, so a run of rows that straddles two clusters takes their
//             touching ends (the merge tree's leaf order becomes a space-filling-curve-like order of the graph).
//   cost      one pass over the non-zeros per level; ~22 levels + the weights: ~0.3 s for 318 M non-zeros (plan time, outside every
//             timed region like the reference's scheduling).
// Everything is integer arithmetic on the device; ties are broken by a hash of the (unordered) pair, so the order is a
// deterministic function of the matrix.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "graph_cluster.h"
#include "thread_stream.h"

namespace sx {
namespace {

#define GC_HIP(x)                                                                                       \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return 2; }       \
    } while (0)

struct Scratch {   // device allocations freed on every exit path (keep() hands one over to the caller)
    std::vector<void *> p;
    ~Scratch() { for (void *q : p) (void)hipFree(q); }
    template <class T> hipError_t alloc(T **out, size_t n) {
        hipError_t e = hipMalloc((void **)out, sizeof(T) * (n ? n : 1));
        if (e == hipSuccess) p.push_back(*out);
        return e;
    }
    void keep(void *q) { p.erase(std::remove(p.begin(), p.end(), q), p.end()); }
};

constexpr int kTriMaxLen = 256;   // rows longer than this: unit weights (their neighbourhoods are not compared)
constexpr int kTriHT = 512;       // hash slots per wavefront for one row's columns (load <= 0.5)
constexpr int kNbHT = 512;        // hash slots per wavefront for a cluster's neighbouring clusters
constexpr int kCand = 4;          // partners kept per cluster and level

__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ void set_insert(int *tab, int c) {
    unsigned h = mix32((unsigned)c) & (kTriHT - 1);
    for (int probe = 0; probe < kTriHT; ++probe) {
        const int prev = atomicCAS(&tab[h], -1, c);
        if (prev == -1 || prev == c) return;
        h = (h + 1) & (kTriHT - 1);
    }
}
__device__ __forceinline__ bool set_has(const int *tab, int c) {
    unsigned h = mix32((unsigned)c) & (kTriHT - 1);
    for (int probe = 0; probe < kTriHT; ++probe) {
        const int k = tab[h];
        if (k == c) return true;
        if (k == -1) return false;
        h = (h + 1) & (kTriHT - 1);
    }
    return false;
}

// One wavefront per row r: t[j] = min(255, 1 + |cols(r) & cols(c_j)|) for every entry j of the row whose column c_j is a row
// (c_j < M, != r) of at most kTriMaxLen entries; 1 otherwise.
__global__ __launch_bounds__(256) void tri_weights(int M, const int *__restrict__ rp, const int *__restrict__ ci, unsigned char *__restrict__ t) {
    __shared__ int tabs[4][kTriHT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= M) return;                       // (no workgroup barrier below: wavefronts are independent)
    int *tab = tabs[wave];
    const int j0 = rp[r], len = rp[r + 1] - j0;
    if (len > kTriMaxLen) {
        for (int e = lane; e < len; e += 64) t[j0 + e] = 1;
        return;
    }
    for (int i = lane; i < kTriHT; i += 64) tab[i] = -1;
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < len; e += 64) set_insert(tab, ci[j0 + e]);
    __builtin_amdgcn_wave_barrier();
    for (int e0 = 0; e0 < len; e0 += 64) {
        const int e = e0 + lane;
        int k0 = 0, lc = 0;
        if (e < len) {
            const int c = ci[j0 + e];
            if ((unsigned)c < (unsigned)M && c != r) { k0 = rp[c]; lc = min(rp[c + 1] - k0, kTriMaxLen); }
        }
        const int n_here = min(64, len - e0);
        unsigned mine = 0;
        for (int i = 0; i < n_here; ++i) {
            const int ck0 = __shfl(k0, i), clc = __shfl(lc, i);   // wave-uniform
            unsigned cnt = 0;
            for (int k = 0; k < clc; k += 64) {
                const bool hit = (k + lane < clc) && set_has(tab, ci[ck0 + k + lane]);
                cnt += (unsigned)__popcll(__ballot(hit));
            }
            if (lane == i) mine = cnt;
        }
        if (e < len) t[j0 + e] = (unsigned char)min(255u, 1u + mine);
    }
}

// Sampled version of the same count: the share of a neighbour row's columns that the sampled row has too.
__global__ __launch_bounds__(256) void probe_shared(int M, const int *__restrict__ rp, const int *__restrict__ ci, int nsample,
                                                    unsigned long long *acc /* [0] shared, [1] compared, [2] near entries, [3] entries, [4] mirrored, [5] tested */) {
    __shared__ int tabs[4][kTriHT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + wave;
    if (s >= nsample) return;
    const int r = (int)((long long)M / 8 + (long long)s * (3LL * M / 4) / nsample);
    int *tab = tabs[wave];
    const int j0 = rp[r], len = rp[r + 1] - j0;
    if (len < 2 || len > kTriMaxLen) return;
    for (int i = lane; i < kTriHT; i += 64) tab[i] = -1;
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < len; e += 64) set_insert(tab, ci[j0 + e]);
    __builtin_amdgcn_wave_barrier();
    {   // locality of the numbering: entries within M / 64 of the diagonal
        unsigned long long near = 0;
        for (int e = lane; e < len; e += 64) {
            const long long d = (long long)ci[j0 + e] - r;
            near += (d < 0 ? -d : d) < (long long)M / 64 + 1 ? 1u : 0u;
        }
        for (int off = 32; off > 0; off >>= 1) near += __shfl_xor((unsigned)near, off);
        if (lane == 0) { atomicAdd(&acc[2], near); atomicAdd(&acc[3], (unsigned long long)len); }
    }
    unsigned long long shared = 0, compared = 0, mirrored = 0, tested = 0;
    for (int q = 0; q < 8; ++q) {                          // eight neighbours spread over the row
        const int c = ci[j0 + (int)((long long)q * len / 8)];
        if ((unsigned)c >= (unsigned)M || c == r) continue;
        const int k0 = rp[c], lc_all = rp[c + 1] - k0, lc = min(lc_all, kTriMaxLen);
        for (int k = 0; k < lc; k += 64) {
            const bool hit = (k + lane < lc) && set_has(tab, ci[k0 + k + lane]);
            shared += (unsigned long long)__popcll(__ballot(hit));
        }
        compared += (unsigned long long)lc;
        bool back = false;                                 // is the pattern symmetric: does row c hold r?
        for (int k = 0; k < min(lc_all, 4096); k += 64) back |= (k + lane < lc_all) && ci[k0 + k + lane] == r;
        mirrored += __ballot(back) ? 1u : 0u;
        ++tested;
    }
    if (lane == 0 && compared) { atomicAdd(&acc[0], shared); atomicAdd(&acc[1], compared); }
    if (lane == 0 && tested) { atomicAdd(&acc[4], mirrored); atomicAdd(&acc[5], tested); }
}

// Do consecutive rows have neighbouring columns?  Pairs (r, r + 1) sampled over the matrix: share of their j-th entries whose
// columns differ by at most 32 (what makes the B loads of spmm_csr_colwise coalesce).
__global__ __launch_bounds__(256) void probe_coherence(int M, const int *__restrict__ rp, const int *__restrict__ ci, int nsample,
                                                       unsigned long long *acc /* [0] close pairs, [1] pairs */) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= nsample || M < 2) return;
    const int r = (int)((long long)s * (M - 1) / nsample);
    const int a0 = rp[r], a1 = rp[r + 1], b1 = rp[r + 2];
    const int n = min(min(a1 - a0, b1 - a1), 64);
    unsigned long long close = 0;
    for (int j = 0; j < n; ++j) {
        const long long d = (long long)ci[a1 + j] - ci[a0 + j];
        close += (d < 0 ? -d : d) <= 32 ? 1u : 0u;
    }
    if (n > 0) { atomicAdd(&acc[0], close); atomicAdd(&acc[1], (unsigned long long)n); }
}

// ---- one aggregation level -----------------------------------------------------------------------------------------------
// Clusters are numbered in the current order: cluster A = rows ord[cstart[A] .. cstart[A + 1]); cinfo[row] = {cluster, its size}.
// One wavefront per cluster: weights towards every neighbouring cluster that still fits under `limit` rows, then the kCand best.
__global__ __launch_bounds__(256) void level_candidates(int nc, int M, const int *__restrict__ cstart, const int *__restrict__ ord,
                                                        const int2 *__restrict__ cinfo, const int *__restrict__ rp,
                                                        const int *__restrict__ ci, const unsigned char *__restrict__ t, int limit,
                                                        unsigned salt, const int *__restrict__ matched, int *__restrict__ cand,
                                                        unsigned *__restrict__ candw) {
    // (matched: clusters paired by an earlier pass of this level -- they keep their lists and are nobody's candidates any more)
    // candw[(A * kCand + k) * 4 + 2 * ha + hb]: weight between half ha of A and half hb of candidate k (halves of the clusters' rows
    // in the current order): what decides the ORIENTATION of a merged pair (level_orient)
    __shared__ int keys[4][kNbHT];
    __shared__ unsigned vals[4][4 * kNbHT];
    __shared__ int sizes[4][kNbHT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int A = blockIdx.x * 4 + wave;
    if (A >= nc || matched[A]) return;
    int *kk = keys[wave];
    unsigned *vv = vals[wave];
    int *ss = sizes[wave];
    const int s0 = cstart[A], s1 = cstart[A + 1], sizeA = s1 - s0;
    if (sizeA >= limit) {                                   // nothing fits any more
        if (lane < kCand) cand[(long long)A * kCand + lane] = -1;
        return;
    }
    for (int i = lane; i < kNbHT; i += 64) kk[i] = -1;
    for (int i = lane; i < 4 * kNbHT; i += 64) vv[i] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (int p = s0; p < s1; ++p) {
        const int r = ord[p];
        const int j0 = rp[r], len = rp[r + 1] - j0;
        const int ha = 2 * (p - s0) >= sizeA ? 1 : 0;
        for (int e = lane; e < len; e += 64) {
            const int c = ci[j0 + e];
            if ((unsigned)c >= (unsigned)M) continue;
            const int2 inf = cinfo[c];
            const int sizeB = inf.y & 0x3fffffff, hb = (int)((unsigned)inf.y >> 30);
            if (inf.x == A || sizeA + sizeB > limit || matched[inf.x]) continue;
            unsigned h = mix32((unsigned)inf.x) & (kNbHT - 1);
            for (int probe = 0; probe < kNbHT; ++probe) {   // (a full table drops the entry: a heuristic loses a candidate)
                const int prev = atomicCAS(&kk[h], -1, inf.x);
                if (prev == -1 || prev == inf.x) { atomicAdd(&vv[(2 * ha + hb) * kNbHT + h], (unsigned)t[j0 + e]); ss[h] = sizeB; break; }
                h = (h + 1) & (kNbHT - 1);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // score = weight / sqrt(partner size) (equal weights: the smaller partner, which keeps the sizes balanced), ties by a hash
    // of the unordered pair (both ends see the same value, so mutual choices are likely)
    constexpr int PER = kNbHT / 64;
    unsigned long long sc[PER];
    int id[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int slot = lane + 64 * i;
        id[i] = kk[slot];
        sc[i] = 0ull;
        if (id[i] >= 0) {
            const unsigned wsum = vv[slot] + vv[kNbHT + slot] + vv[2 * kNbHT + slot] + vv[3 * kNbHT + slot];
            const float f = (float)wsum * __frsqrt_rn((float)ss[slot]);
            const unsigned lo = (unsigned)min(A, id[i]), hi = (unsigned)max(A, id[i]);
            const unsigned tie = mix32(lo * 0x9E3779B1u + mix32(hi + salt));
            sc[i] = ((unsigned long long)__float_as_uint(f) << 32) | tie;
        }
    }
    for (int k = 0; k < kCand; ++k) {
        unsigned long long best = 0ull;
        int bid = -1;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (sc[i] > best) { best = sc[i]; bid = id[i]; }
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned olo = __shfl_xor((unsigned)(best & 0xffffffffull), off), ohi = __shfl_xor((unsigned)(best >> 32), off);
            const int oid = __shfl_xor(bid, off);
            const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
            if (other > best || (other == best && oid > bid)) { best = other; bid = oid; }
        }
        if (lane == 0) cand[(long long)A * kCand + k] = best ? bid : -1;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (id[i] == bid && bid >= 0) {                 // the lane that holds the winner's slot writes its four half-to-half weights
                const int slot = lane + 64 * i;
                for (int q = 0; q < 4; ++q) candw[((long long)A * kCand + k) * 4 + q] = vv[q * kNbHT + slot];
                sc[i] = 0ull;
            }
    }
}

// ---- the same for the TOP of the merge tree.  Once few clusters are left a cluster holds 10^4 .. 10^6 rows and one wavefront per
// cluster would walk them alone for seconds; the coarse graph, however, is tiny: with nc <= kDenseMax clusters the weights fit a
// dense nc x nc x 4 array.  One wavefront per ROW adds the row's entries (pre-aggregated per neighbouring cluster in LDS) with
// global atomics, then one wavefront per cluster picks its kCand best partners from its line of the array.
constexpr int kDenseMax = 2048;
constexpr int kRowHT = 128;

__global__ __launch_bounds__(256) void level_dense_accumulate(int M, int nc, const int *__restrict__ ord, const int2 *__restrict__ cinfo,
                                                              const int *__restrict__ rp, const int *__restrict__ ci,
                                                              const unsigned char *__restrict__ t, int limit, const int *__restrict__ matched,
                                                              unsigned *__restrict__ W) {
    __shared__ int keys[4][kRowHT];
    __shared__ unsigned vals[4][4 * kRowHT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wave;
    if (p >= M) return;
    int *kk = keys[wave];
    unsigned *vv = vals[wave];
    const int r = ord[p];
    const int2 me = cinfo[r];
    const int A = me.x, sizeA = me.y & 0x3fffffff, ha = (int)((unsigned)me.y >> 30);
    if (sizeA >= limit || matched[A]) return;
    for (int i = lane; i < kRowHT; i += 64) kk[i] = -1;
    for (int i = lane; i < 4 * kRowHT; i += 64) vv[i] = 0u;
    __builtin_amdgcn_wave_barrier();
    const int j0 = rp[r], len = rp[r + 1] - j0;
    for (int e = lane; e < len; e += 64) {
        const int c = ci[j0 + e];
        if ((unsigned)c >= (unsigned)M) continue;
        const int2 inf = cinfo[c];
        const int sizeB = inf.y & 0x3fffffff, hb = (int)((unsigned)inf.y >> 30);
        if (inf.x == A || sizeA + sizeB > limit || matched[inf.x]) continue;
        unsigned h = mix32((unsigned)inf.x) & (kRowHT - 1);
        bool placed = false;
        for (int probe = 0; probe < kRowHT && !placed; ++probe) {
            const int prev = atomicCAS(&kk[h], -1, inf.x);
            if (prev == -1 || prev == inf.x) { atomicAdd(&vv[hb * kRowHT + h], (unsigned)t[j0 + e]); placed = true; }
            else h = (h + 1) & (kRowHT - 1);
        }
        if (!placed) atomicAdd(&W[((size_t)A * nc + inf.x) * 4 + 2 * ha + hb], (unsigned)t[j0 + e]);   // (more than 128 neighbouring clusters)
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kRowHT; i += 64) {
        const int B = kk[i];
        if (B < 0) continue;
        for (int hb = 0; hb < 2; ++hb) {
            const unsigned v = vv[hb * kRowHT + i];
            if (v) atomicAdd(&W[((size_t)A * nc + B) * 4 + 2 * ha + hb], v);
        }
    }
}

__global__ __launch_bounds__(256) void level_dense_candidates(int nc, const int *__restrict__ cstart, const unsigned *__restrict__ W, int limit,
                                                              unsigned salt, const int *__restrict__ matched, int *__restrict__ cand,
                                                              unsigned *__restrict__ candw) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int A = blockIdx.x * 4 + wave;
    if (A >= nc || matched[A]) return;
    const int sizeA = cstart[A + 1] - cstart[A];
    int chosen[kCand];
#pragma unroll
    for (int k = 0; k < kCand; ++k) chosen[k] = -1;
    for (int k = 0; k < kCand; ++k) {
        unsigned long long best = 0ull;
        int bid = -1;
        if (sizeA < limit)
            for (int B = lane; B < nc; B += 64) {
                if (B == A) continue;
                bool taken = false;
#pragma unroll
                for (int q = 0; q < kCand; ++q) taken |= chosen[q] == B;
                if (taken) continue;
                const unsigned *w = W + ((size_t)A * nc + B) * 4;
                const unsigned wsum = w[0] + w[1] + w[2] + w[3];
                const int sizeB = cstart[B + 1] - cstart[B];
                if (!wsum || sizeA + sizeB > limit) continue;
                const float f = (float)wsum * __frsqrt_rn((float)sizeB);
                const unsigned lo = (unsigned)min(A, B), hi = (unsigned)max(A, B);
                const unsigned long long sc = ((unsigned long long)__float_as_uint(f) << 32) | mix32(lo * 0x9E3779B1u + mix32(hi + salt));
                if (sc > best || (sc == best && B > bid)) { best = sc; bid = B; }
            }
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned olo = __shfl_xor((unsigned)(best & 0xffffffffull), off), ohi = __shfl_xor((unsigned)(best >> 32), off);
            const int oid = __shfl_xor(bid, off);
            const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
            if (other > best || (other == best && oid > bid)) { best = other; bid = oid; }
        }
        chosen[k] = best ? bid : -1;
        if (lane == 0) {
            cand[(long long)A * kCand + k] = chosen[k];
            for (int q = 0; q < 4; ++q) candw[((long long)A * kCand + k) * 4 + q] = chosen[k] >= 0 ? W[((size_t)A * nc + chosen[k]) * 4 + q] : 0u;
        }
    }
}

__global__ __launch_bounds__(256) void level_count_unmatched(int nc, const int *__restrict__ matched, const int *__restrict__ cand, int *count) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    const bool open = a < nc && !matched[a] && cand[(long long)a * kCand] >= 0;   // unmatched although it had somebody to merge with
    const unsigned long long b = __ballot(open);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (int)__popcll(b));
}
__global__ __launch_bounds__(256) void level_reset(int nc, int *matched, int *mate) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a < nc) { matched[a] = 0; mate[a] = -1; }
}
__global__ __launch_bounds__(256) void level_propose(int nc, const int *__restrict__ cand, const int *__restrict__ matched, int *__restrict__ want) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= nc) return;
    int w = -1;
    if (!matched[a])
        for (int k = 0; k < kCand && w < 0; ++k) {
            const int b = cand[(long long)a * kCand + k];
            if (b >= 0 && !matched[b]) w = b;
        }
    want[a] = w;
}
__global__ __launch_bounds__(256) void level_accept(int nc, const int *__restrict__ want, int *matched, int *mate) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= nc) return;
    const int b = want[a];
    if (b > a && want[b] == a) { matched[a] = 1; matched[b] = 1; mate[a] = b; mate[b] = a; }
}
// Orientation of every merged pair: the leader's rows come first; each member may be laid down reversed, so that the halves that
// end up ADJACENT in the order are the two the clusters are most strongly connected through (weight between the tail half of the
// first and the head half of the second).  Applied at every level this makes the final order locality-preserving at every scale --
// a run of 64 consecutive rows that straddles two clusters takes the touching ends of both (measured: -6 % panel rows).
__global__ __launch_bounds__(256) void level_orient(int nc, const int *__restrict__ mate, const int *__restrict__ cand,
                                                    const unsigned *__restrict__ candw, int *__restrict__ flip) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= nc) return;
    const int m = mate[a];
    if (m < 0) { flip[a] = 0; return; }
    if (m < a) return;                                     // the leader (first in the order) decides for both
    unsigned w[4] = {0, 0, 0, 0};
    for (int k = 0; k < kCand; ++k)
        if (cand[(long long)a * kCand + k] == m)
            for (int q = 0; q < 4; ++q) w[q] = candw[((long long)a * kCand + k) * 4 + q];
    // w[2 ha + hb]; layouts: (A, B) joins A's tail half (1) to B's head half (0); (A, rev B): 1-1; (rev A, B): 0-0; (rev A, rev B): 0-1
    const unsigned opt[4] = {w[2], w[3], w[0], w[1]};
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (opt[i] > opt[best]) best = i;
    flip[a] = best >> 1;
    flip[m] = best & 1;
}
// leader of a pair = its member that comes first in the order; new cluster sizes at the leaders
__global__ __launch_bounds__(256) void level_leaders(int nc, const int *__restrict__ mate, const int *__restrict__ cstart, int *is_leader,
                                                     int *new_size) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a > nc) return;
    if (a == nc) { is_leader[a] = 0; new_size[a] = 0; return; }   // (scans run over nc + 1 elements: the last one yields the totals)
    const int m = mate[a];
    const bool lead = m < 0 || m > a;
    is_leader[a] = lead ? 1 : 0;
    new_size[a] = lead ? (cstart[a + 1] - cstart[a]) + (m >= 0 ? cstart[m + 1] - cstart[m] : 0) : 0;
}
__global__ __launch_bounds__(256) void level_move(int M, int nc, const int *__restrict__ ord, const int2 *__restrict__ cinfo,
                                                  const int *__restrict__ cstart, const int *__restrict__ mate, const int *__restrict__ flip,
                                                  const int *__restrict__ new_idx, const int *__restrict__ new_start, int *__restrict__ ord2,
                                                  int2 *__restrict__ cinfo2, int *__restrict__ cstart2) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= M) return;
    const int r = ord[p];
    const int a = cinfo[r].x;
    const int m = mate[a];
    const int lead = (m >= 0 && m < a) ? m : a;
    const int own = cstart[a + 1] - cstart[a];
    const int o = flip[a] ? own - 1 - (p - cstart[a]) : p - cstart[a];
    const int lead_size = cstart[lead + 1] - cstart[lead];
    const int total = lead_size + ((m >= 0) ? (lead == a ? cstart[m + 1] - cstart[m] : cstart[a + 1] - cstart[a]) : 0);
    const int on = lead == a ? o : lead_size + o;          // offset inside the new cluster
    const int np = new_start[lead] + on;
    ord2[np] = r;
    cinfo2[r] = make_int2(new_idx[lead], total | ((2 * on >= total ? 1 : 0) << 30));   // size + which half of its cluster the row is in
    if (on == 0) cstart2[new_idx[lead]] = new_start[lead];
    if (p == 0) cstart2[new_idx[nc]] = M;   // new_idx[nc] = number of new clusters (exclusive scan over nc + 1 elements)
}

__global__ __launch_bounds__(256) void init_level0(int M, int *ord, int2 *cinfo, int *cstart) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < M) { ord[r] = r; cinfo[r] = make_int2(r, 1); cstart[r] = r; }
    if (r == M) cstart[r] = M;
}

// ---- block refinement ----------------------------------------------------------------------------------------------------
// The merge tree gives an order in which runs of rows are compact; cutting it every 64 rows still leaves blocks whose boundary rows
// sit on the wrong side.  A few sweeps of capacity-constrained label propagation repair that: the order is cut into blocks of `per`
// (< 64) rows, every row counts its neighbours per block and asks to move to the block that holds more of them than its own; per
// target block the requests with the highest gain are granted while the block has room (<= 64 rows).  Deterministic: requests are
// radix-sorted by (target, gain, row), a request's rank inside its target's segment decides.  Measured on renumbered meshes
// (prototype and device agree): -13 .. -14 % dictionary rows for 3 % more blocks (3-dof 27-point: 7.3 -> 6.3 per matrix row, hand-made
// bricks 6.2).
constexpr int kRefHT = 128;

__global__ __launch_bounds__(256) void refine_init(int M, int per, const int *__restrict__ order, int *__restrict__ blk, int *__restrict__ pos) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < M) { const int r = order[p]; blk[r] = p / per; pos[r] = p; }
}
__global__ __launch_bounds__(256) void refine_sizes(int M, const int *__restrict__ blk, int *size) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < M) atomicAdd(&size[blk[r]], 1);
}
// one wavefront per row: key = target block << 32 | (0xffffffff - gain) for rows that want to move this sweep, all-ones otherwise
__global__ __launch_bounds__(256) void refine_requests(int M, const int *__restrict__ rp, const int *__restrict__ ci, const int *__restrict__ blk,
                                                       int sweep, unsigned long long *__restrict__ key, int *__restrict__ val) {
    __shared__ int keys[4][kRefHT];
    __shared__ int cnts[4][kRefHT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= M) return;
    if (lane == 0) { key[r] = ~0ull; val[r] = r; }
    if (((mix32((unsigned)r) >> 7) + (unsigned)sweep) & 1u) return;       // half of the rows per sweep: neighbours do not swap past each other
    int *kk = keys[wave], *cc = cnts[wave];
    for (int i = lane; i < kRefHT; i += 64) { kk[i] = -1; cc[i] = 0; }
    __builtin_amdgcn_wave_barrier();
    const int own = blk[r];
    const int j0 = rp[r], len = min(rp[r + 1] - j0, 4 * kTriMaxLen);
    for (int e = lane; e < len; e += 64) {
        const int c = ci[j0 + e];
        if ((unsigned)c >= (unsigned)M || c == r) continue;
        const int b = blk[c];
        unsigned h = mix32((unsigned)b) & (kRefHT - 1);
        for (int probe = 0; probe < kRefHT; ++probe) {
            const int prev = atomicCAS(&kk[h], -1, b);
            if (prev == -1 || prev == b) { atomicAdd(&cc[h], 1); break; }
            h = (h + 1) & (kRefHT - 1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    int own_cnt = 0, best_cnt = 0, best_b = 0x7fffffff;
    for (int i = lane; i < kRefHT; i += 64) {
        const int b = kk[i];
        if (b < 0) continue;
        if (b == own) own_cnt = cc[i];
        else if (cc[i] > best_cnt || (cc[i] == best_cnt && b < best_b)) { best_cnt = cc[i]; best_b = b; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        own_cnt = max(own_cnt, __shfl_xor(own_cnt, off));
        const int oc = __shfl_xor(best_cnt, off), ob = __shfl_xor(best_b, off);
        if (oc > best_cnt || (oc == best_cnt && ob < best_b)) { best_cnt = oc; best_b = ob; }
    }
    if (lane == 0 && best_cnt > own_cnt)
        key[r] = ((unsigned long long)(unsigned)best_b << 32) | (unsigned long long)(0xffffffffu - (unsigned)(best_cnt - own_cnt));
}
__global__ __launch_bounds__(256) void refine_segments(int M, const unsigned long long *__restrict__ skey, int *__restrict__ seg_start) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M || skey[i] == ~0ull) return;
    const unsigned t = (unsigned)(skey[i] >> 32);
    if (i == 0 || (unsigned)(skey[i - 1] >> 32) != t) seg_start[t] = i;
}
__global__ __launch_bounds__(256) void refine_grant(int M, int cap, const unsigned long long *__restrict__ skey, const int *__restrict__ sval,
                                                    const int *__restrict__ seg_start, const int *__restrict__ size_old, int *size_new, int *blk,
                                                    int *moved) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M || skey[i] == ~0ull) return;
    const int t = (int)(skey[i] >> 32), r = sval[i];
    if (i - seg_start[t] >= cap - size_old[t]) return;                    // the block is full for this sweep (departures are not counted: safe)
    const int own = blk[r];
    if (size_old[own] <= 1) return;
    blk[r] = t;
    atomicAdd(&size_new[t], 1);
    atomicSub(&size_new[own], 1);
    atomicAdd(moved, 1);
}
__global__ __launch_bounds__(256) void refine_final_keys(int M, const int *__restrict__ blk, const int *__restrict__ pos, unsigned long long *key, int *val) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < M) { key[r] = ((unsigned long long)(unsigned)blk[r] << 32) | (unsigned)pos[r]; val[r] = r; }
}
__global__ __launch_bounds__(256) void refine_cuts(int M, const unsigned long long *__restrict__ skey, unsigned char *cut) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < M) cut[i] = (i == 0 || (skey[i] >> 32) != (skey[i - 1] >> 32)) ? 1 : 0;
}

// ---- column order ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void first_touch(int M, const int *__restrict__ rp, const int *__restrict__ ci, const int *__restrict__ ord,
                                                   int *first) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wave;
    if (p >= M) return;
    const int r = ord[p], j0 = rp[r], len = rp[r + 1] - j0;
    for (int e = lane; e < len; e += 64) atomicMin(&first[ci[j0 + e]], p);
}
__global__ __launch_bounds__(256) void iota_fill(int n, int *v, int *w, int fill) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { v[i] = i; if (w) w[i] = fill; }
}
__global__ __launch_bounds__(256) void invert_perm(int n, const int *__restrict__ order, int *__restrict__ pos) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) pos[order[i]] = i;
}
__global__ __launch_bounds__(256) void relabel(long long nnz, int *ci, const int *__restrict__ colpos) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j < nnz) ci[j] = colpos[ci[j]];
}
inline unsigned blocks_for(long long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

int probe_shared_neighbourhood_device(int M, const int *d_rp, const int *d_ci, int nsample, double *shared_fraction, double *near_fraction,
                                      std::string &err, double *symmetric_fraction) {
    *shared_fraction = 0.0;
    *near_fraction = 0.0;
    if (symmetric_fraction) *symmetric_fraction = 1.0;
    if (M < 16 || nsample < 1) return 0;
    Scratch tmp;
    unsigned long long *d_acc = nullptr, h_acc[6] = {0, 0, 0, 0, 0, 0};
    GC_HIP(tmp.alloc(&d_acc, 6));
    GC_HIP(hipMemset(d_acc, 0, sizeof h_acc));
    hipLaunchKernelGGL(probe_shared, dim3(blocks_for(nsample, 4)), dim3(256), 0, nullptr, M, d_rp, d_ci, nsample, d_acc);
    GC_HIP(hipMemcpy(h_acc, d_acc, sizeof h_acc, hipMemcpyDeviceToHost));
    if (h_acc[1]) *shared_fraction = (double)h_acc[0] / (double)h_acc[1];
    if (h_acc[3]) *near_fraction = (double)h_acc[2] / (double)h_acc[3];
    if (symmetric_fraction && h_acc[5]) *symmetric_fraction = (double)h_acc[4] / (double)h_acc[5];
    return 0;
}

int probe_row_coherence_device(int M, const int *d_rp, const int *d_ci, int nsample, double *close_fraction, std::string &err) {
    *close_fraction = 0.0;
    if (M < 2 || nsample < 1) return 0;
    Scratch tmp;
    unsigned long long *d_acc = nullptr, h_acc[2] = {0, 0};
    GC_HIP(tmp.alloc(&d_acc, 2));
    GC_HIP(hipMemset(d_acc, 0, sizeof h_acc));
    hipLaunchKernelGGL(probe_coherence, dim3(blocks_for(nsample, 256)), dim3(256), 0, nullptr, M, d_rp, d_ci, nsample, d_acc);
    GC_HIP(hipMemcpy(h_acc, d_acc, sizeof h_acc, hipMemcpyDeviceToHost));
    if (h_acc[1]) *close_fraction = (double)h_acc[0] / (double)h_acc[1];
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void snapshot_clusters(int M, const int2 *__restrict__ cinfo, int *__restrict__ id) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < M) id[r] = cinfo[r].x;
}
}  // namespace

int cluster_rows_graph_device(int M, int K, int64_t nnz, const int *d_rp, const int *d_ci, int max_cluster_rows, int **d_order,
                              std::string &err, const unsigned char *d_weights, int snapshot_limit, int **d_snapshot) {
    *d_order = nullptr;
    int *snap = nullptr;
    bool snapped = false;
    if (d_snapshot) *d_snapshot = nullptr;
    if (M != K || M < 2 || nnz <= 0) return 1;
    Scratch tmp;
    unsigned char *t = nullptr;
    int *ord[2] = {nullptr, nullptr}, *cstart[2] = {nullptr, nullptr};
    int2 *cinfo[2] = {nullptr, nullptr};
    int *cand = nullptr, *matched = nullptr, *want = nullptr, *mate = nullptr, *is_leader = nullptr, *new_idx = nullptr, *new_size = nullptr,
        *new_start = nullptr;
    GC_HIP(tmp.alloc(&t, (size_t)nnz));
    if (d_snapshot) GC_HIP(tmp.alloc(&snap, (size_t)M));
    for (int i = 0; i < 2; ++i) {
        GC_HIP(tmp.alloc(&ord[i], (size_t)M));
        GC_HIP(tmp.alloc(&cstart[i], (size_t)M + 1));
        GC_HIP(tmp.alloc(&cinfo[i], (size_t)M));
    }
    GC_HIP(tmp.alloc(&cand, (size_t)M * kCand));
    unsigned *candw = nullptr, *W = nullptr;
    int *flip = nullptr, *d_count = nullptr;
    GC_HIP(tmp.alloc(&d_count, 1));
    GC_HIP(tmp.alloc(&candw, (size_t)M * kCand * 4));
    GC_HIP(tmp.alloc(&flip, (size_t)M));
    GC_HIP(tmp.alloc(&matched, (size_t)M));
    GC_HIP(tmp.alloc(&want, (size_t)M));
    GC_HIP(tmp.alloc(&mate, (size_t)M));
    GC_HIP(tmp.alloc(&is_leader, (size_t)M + 1));
    GC_HIP(tmp.alloc(&new_idx, (size_t)M + 1));
    GC_HIP(tmp.alloc(&new_size, (size_t)M + 1));
    GC_HIP(tmp.alloc(&new_start, (size_t)M + 1));
    void *scan_tmp = nullptr;
    size_t scan_bytes = 0;
    GC_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, is_leader, new_idx, M + 1, nullptr));
    GC_HIP(tmp.alloc((char **)&scan_tmp, scan_bytes));

    if (d_weights) GC_HIP(hipMemcpyAsync(t, d_weights, (size_t)nnz, hipMemcpyDeviceToDevice, nullptr));
    else hipLaunchKernelGGL(tri_weights, dim3(blocks_for(M, 4)), dim3(256), 0, nullptr, M, d_rp, d_ci, t);
    hipLaunchKernelGGL(init_level0, dim3(blocks_for((long long)M + 1, 256)), dim3(256), 0, nullptr, M, ord[0], cinfo[0], cstart[0]);
    int nc = M, cur = 0, level = 0;
    const bool trace = getenv("SEXTANS_CLUSTER_TRACE") != nullptr;
    double t_prev = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    for (long long limit = 2; limit <= (long long)max_cluster_rows && nc > 1; limit *= 2, ++level) {
        hipLaunchKernelGGL(level_reset, dim3(blocks_for(nc, 256)), dim3(256), 0, nullptr, nc, matched, mate);
        // Up to three passes per level: clusters still unmatched after the handshake rounds of a pass (their four candidates went to
        // others) get fresh candidates among the clusters that are still free.  Without them a third of the clusters stayed single
        // per level, the sizes spread over four orders of magnitude and the tree never closed (measured: 12 469 clusters left).
        for (int pass = 0; pass < 3; ++pass) {
            const unsigned salt = (unsigned)(level * 3 + pass) * 0x632BE5ABu;
            if (nc > kDenseMax) {
                hipLaunchKernelGGL(level_candidates, dim3(blocks_for(nc, 4)), dim3(256), 0, nullptr, nc, M, cstart[cur], ord[cur], cinfo[cur], d_rp,
                                   d_ci, t, (int)limit, salt, matched, cand, candw);
            } else {
                if (!W) GC_HIP(tmp.alloc(&W, (size_t)kDenseMax * kDenseMax * 4));
                GC_HIP(hipMemsetAsync(W, 0, sizeof(unsigned) * (size_t)nc * nc * 4, nullptr));
                hipLaunchKernelGGL(level_dense_accumulate, dim3(blocks_for(M, 4)), dim3(256), 0, nullptr, M, nc, ord[cur], cinfo[cur], d_rp, d_ci, t,
                                   (int)limit, matched, W);
                hipLaunchKernelGGL(level_dense_candidates, dim3(blocks_for(nc, 4)), dim3(256), 0, nullptr, nc, cstart[cur], W, (int)limit, salt,
                                   matched, cand, candw);
            }
            for (int round = 0; round < 5; ++round) {
                hipLaunchKernelGGL(level_propose, dim3(blocks_for(nc, 256)), dim3(256), 0, nullptr, nc, cand, matched, want);
                hipLaunchKernelGGL(level_accept, dim3(blocks_for(nc, 256)), dim3(256), 0, nullptr, nc, want, matched, mate);
            }
            if (pass == 2) break;
            int open = 0;
            GC_HIP(hipMemsetAsync(d_count, 0, sizeof(int), nullptr));
            hipLaunchKernelGGL(level_count_unmatched, dim3(blocks_for(nc, 256)), dim3(256), 0, nullptr, nc, matched, cand, d_count);
            GC_HIP(hipMemcpy(&open, d_count, sizeof(int), hipMemcpyDeviceToHost));
            if (open * 16 < nc) break;                     // fewer than 6 % could still pair up
        }
        hipLaunchKernelGGL(level_orient, dim3(blocks_for(nc, 256)), dim3(256), 0, nullptr, nc, mate, cand, candw, flip);
        hipLaunchKernelGGL(level_leaders, dim3(blocks_for((long long)nc + 1, 256)), dim3(256), 0, nullptr, nc, mate, cstart[cur], is_leader,
                           new_size);
        GC_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, is_leader, new_idx, nc + 1, nullptr));
        GC_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, new_size, new_start, nc + 1, nullptr));
        hipLaunchKernelGGL(level_move, dim3(blocks_for(M, 256)), dim3(256), 0, nullptr, M, nc, ord[cur], cinfo[cur], cstart[cur], mate, flip,
                           new_idx, new_start, ord[cur ^ 1], cinfo[cur ^ 1], cstart[cur ^ 1]);
        int nc_new = 0;
        GC_HIP(hipMemcpy(&nc_new, new_idx + nc, sizeof(int), hipMemcpyDeviceToHost));
        if (trace) {
            const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            fprintf(stderr, "graph clustering: level %d limit %lld clusters %d -> %d  %.3f s\n", level, limit, nc, nc_new, now - t_prev);
            t_prev = now;
        }
        cur ^= 1;
        if (nc_new <= 0 || nc_new > nc) { err = "graph clustering: inconsistent level"; return 2; }
        if (snap && !snapped && limit >= snapshot_limit) {   // cluster of every row once clusters hold up to snapshot_limit rows: later levels only concatenate (and reverse) whole clusters
            hipLaunchKernelGGL(snapshot_clusters, dim3(blocks_for(M, 256)), dim3(256), 0, nullptr, M, cinfo[cur], snap);
            snapped = true;
        }
        const bool stalled = nc_new == nc || (limit >= 4096 && (long long)(nc - nc_new) * 32 < nc);
        nc = nc_new;
        if (stalled && limit >= 64) break;   // nothing (or only a trickle) merges any more: disconnected pieces, rows without neighbours
    }
    if (snap && !snapped) hipLaunchKernelGGL(snapshot_clusters, dim3(blocks_for(M, 256)), dim3(256), 0, nullptr, M, cinfo[cur], snap);
    GC_HIP(hipDeviceSynchronize());
    GC_HIP(hipGetLastError());
    tmp.keep(ord[cur]);
    *d_order = ord[cur];
    if (snap) { tmp.keep(snap); *d_snapshot = snap; }
    return 0;
}

int refine_blocks_device(int M, const int *d_rp, const int *d_ci, int *d_order, int per, int cap, int sweeps, unsigned char **d_cut,
                         std::string &err) {
    *d_cut = nullptr;
    if (M < 2 || per < 1 || per > cap) return 1;
    Scratch tmp;
    const int nb = (M + per - 1) / per;
    int *blk = nullptr, *pos = nullptr, *size[2] = {nullptr, nullptr}, *seg = nullptr, *val = nullptr, *sval = nullptr, *moved = nullptr;
    unsigned long long *key = nullptr, *skey = nullptr;
    unsigned char *cut = nullptr;
    GC_HIP(tmp.alloc(&blk, (size_t)M));
    GC_HIP(tmp.alloc(&pos, (size_t)M));
    GC_HIP(tmp.alloc(&size[0], (size_t)nb));
    GC_HIP(tmp.alloc(&size[1], (size_t)nb));
    GC_HIP(tmp.alloc(&seg, (size_t)nb));
    GC_HIP(tmp.alloc(&val, (size_t)M));
    GC_HIP(tmp.alloc(&sval, (size_t)M));
    GC_HIP(tmp.alloc(&key, (size_t)M));
    GC_HIP(tmp.alloc(&skey, (size_t)M));
    GC_HIP(tmp.alloc(&moved, 1));
    GC_HIP(tmp.alloc(&cut, (size_t)M));
    void *sort_tmp = nullptr;
    size_t bytes = 0;
    GC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, key, skey, val, sval, M, 0, 64, nullptr));
    GC_HIP(tmp.alloc((char **)&sort_tmp, bytes));
    hipLaunchKernelGGL(refine_init, dim3(blocks_for(M, 256)), dim3(256), 0, nullptr, M, per, d_order, blk, pos);
    GC_HIP(hipMemsetAsync(size[0], 0, sizeof(int) * (size_t)nb, nullptr));
    hipLaunchKernelGGL(refine_sizes, dim3(blocks_for(M, 256)), dim3(256), 0, nullptr, M, blk, size[0]);
    int cur = 0;
    for (int sweep = 0; sweep < sweeps; ++sweep) {
        hipLaunchKernelGGL(refine_requests, dim3(blocks_for(M, 4)), dim3(256), 0, nullptr, M, d_rp, d_ci, blk, sweep, key, val);
        GC_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp, bytes, key, skey, val, sval, M, 0, 64, nullptr));
        GC_HIP(hipMemcpyAsync(size[cur ^ 1], size[cur], sizeof(int) * (size_t)nb, hipMemcpyDeviceToDevice, nullptr));
        GC_HIP(hipMemsetAsync(moved, 0, sizeof(int), nullptr));
        hipLaunchKernelGGL(refine_segments, dim3(blocks_for(M, 256)), dim3(256), 0, nullptr, M, skey, seg);
        hipLaunchKernelGGL(refine_grant, dim3(blocks_for(M, 256)), dim3(256), 0, nullptr, M, cap, skey, sval, seg, size[cur], size[cur ^ 1], blk, moved);
        cur ^= 1;
        int h_moved = 0;
        GC_HIP(hipMemcpy(&h_moved, moved, sizeof(int), hipMemcpyDeviceToHost));
        if (getenv("SEXTANS_CLUSTER_TRACE")) fprintf(stderr, "graph clustering: refinement sweep %d moved %d rows\n", sweep, h_moved);
        if ((long long)h_moved * 2000 < M) break;
    }
    hipLaunchKernelGGL(refine_final_keys, dim3(blocks_for(M, 256)), dim3(256), 0, nullptr, M, blk, pos, key, val);
    GC_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp, bytes, key, skey, val, d_order, M, 0, 64, nullptr));
    hipLaunchKernelGGL(refine_cuts, dim3(blocks_for(M, 256)), dim3(256), 0, nullptr, M, skey, cut);
    GC_HIP(hipDeviceSynchronize());
    GC_HIP(hipGetLastError());
    tmp.keep(cut);
    *d_cut = cut;
    return 0;
}

int column_first_touch_order_device(int M, int K, const int *d_rp, const int *d_ci, const int *d_order, int **d_colpos, std::string &err) {
    *d_colpos = nullptr;
    if (K <= 0) return 1;
    Scratch tmp;
    int *first = nullptr, *first_sorted = nullptr, *cols = nullptr, *cols_sorted = nullptr, *pos = nullptr;
    GC_HIP(tmp.alloc(&first, (size_t)K));
    GC_HIP(tmp.alloc(&first_sorted, (size_t)K));
    GC_HIP(tmp.alloc(&cols, (size_t)K));
    GC_HIP(tmp.alloc(&cols_sorted, (size_t)K));
    GC_HIP(tmp.alloc(&pos, (size_t)K));
    hipLaunchKernelGGL(iota_fill, dim3(blocks_for(K, 256)), dim3(256), 0, nullptr, K, cols, first, 0x7fffffff);
    if (M > 0) hipLaunchKernelGGL(first_touch, dim3(blocks_for(M, 4)), dim3(256), 0, nullptr, M, d_rp, d_ci, d_order, first);
    void *sort_tmp = nullptr;
    size_t bytes = 0;
    GC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, first, first_sorted, cols, cols_sorted, K, 0, 32, nullptr));
    GC_HIP(tmp.alloc((char **)&sort_tmp, bytes));
    GC_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp, bytes, first, first_sorted, cols, cols_sorted, K, 0, 32, nullptr));   // stable: ties by column
    hipLaunchKernelGGL(invert_perm, dim3(blocks_for(K, 256)), dim3(256), 0, nullptr, K, cols_sorted, pos);
    GC_HIP(hipDeviceSynchronize());
    GC_HIP(hipGetLastError());
    tmp.keep(pos);
    *d_colpos = pos;
    return 0;
}

int relabel_columns_device(int64_t nnz, int *d_ci, const int *d_colpos, std::string &err) {
    if (nnz > 0) hipLaunchKernelGGL(relabel, dim3(blocks_for(nnz, 256)), dim3(256), 0, nullptr, (long long)nnz, d_ci, d_colpos);
    GC_HIP(hipDeviceSynchronize());
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void slab_count(int M, const int *__restrict__ rp, const int *__restrict__ ci, int off, int *__restrict__ cnt) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    int n = 0;
    for (int j = rp[r]; j < rp[r + 1]; ++j) n += (unsigned)(ci[j] - off) < (unsigned)M;
    cnt[r] = n;
}
__global__ __launch_bounds__(256) void slab_fill(int M, const int *__restrict__ rp, const int *__restrict__ ci, int off, const int *__restrict__ orp,
                                                 int *__restrict__ oci) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    int o = orp[r];
    for (int j = rp[r]; j < rp[r + 1]; ++j) {
        const int c = ci[j] - off;
        if ((unsigned)c < (unsigned)M) oci[o++] = c;
    }
}
}  // namespace

int local_square_pattern_device(int M, const int *d_rp, const int *d_ci, int row_offset, int **out_rp, int **out_ci, int64_t *out_nnz,
                                std::string &err) {
    *out_rp = *out_ci = nullptr;
    *out_nnz = 0;
    if (M <= 0) return 1;
    Scratch tmp;
    int *cnt = nullptr, *orp = nullptr, *oci = nullptr;
    GC_HIP(tmp.alloc(&cnt, (size_t)M + 1));
    GC_HIP(tmp.alloc(&orp, (size_t)M + 1));
    GC_HIP(hipMemsetAsync(cnt + M, 0, sizeof(int), nullptr));
    hipLaunchKernelGGL(slab_count, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, nullptr, M, d_rp, d_ci, row_offset, cnt);
    void *scan_tmp = nullptr;
    size_t bytes = 0;
    GC_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, cnt, orp, M + 1, nullptr));
    GC_HIP(tmp.alloc((char **)&scan_tmp, bytes));
    GC_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, bytes, cnt, orp, M + 1, nullptr));
    int total = 0;
    GC_HIP(hipMemcpy(&total, orp + M, sizeof(int), hipMemcpyDeviceToHost));
    GC_HIP(tmp.alloc(&oci, (size_t)std::max(total, 1)));
    hipLaunchKernelGGL(slab_fill, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, nullptr, M, d_rp, d_ci, row_offset, orp, oci);
    GC_HIP(hipDeviceSynchronize());
    tmp.keep(orp); tmp.keep(oci);
    *out_rp = orp; *out_ci = oci; *out_nnz = total;
    return 0;
}

// ---- row-similarity graph of a RECTANGULAR matrix ----------------------------------------------------------------------------
// cluster_rows_graph_device reads a column index as the row of a neighbour: M == K.  The reference schedules ANY M x K matrix for
// its on-chip window (sparse_helper.h:345-403, K and M independent); the LP / least-squares / rectangular matrices of SuiteSparse
// have no "row c".  What two rows of any matrix can share is COLUMNS: the rows are clustered over the graph in which row r is joined
// to the kRowSimDeg rows that share the most columns with it.
//   transpose  the pattern sorted by column (hipcub radix sort of (column, row) pairs, stable: rows ascending inside a column);
//   candidates one wavefront per row r: kRowSimCols of its columns spread over the row; from each column's row list a window of up to
//              64 rows centred on r itself (binary search: in a numbering with locality the nearby rows are the similar ones, in a
//              random numbering any window is as good as another) is counted into an LDS hash table -- a row that shares many
//              columns with r turns up in many of the lists;
//   weights    the kRowSimDeg most frequent candidates get their exact overlap |cols(r) & cols(r')| (the same measure tri_weights
//              uses for square matrices), 1 .. 255;
//   output     a square M x M pattern with exactly kRowSimDeg slots per row (-1 = empty slot; every consumer skips indices outside
//              [0, M)) + one weight byte per slot: cluster_rows_graph_device(..., d_weights) and refine_blocks_device run on it.
namespace {
constexpr int kRowSimDeg = 16, kRowSimCols = 16, kRowSimHT = 1024;

__global__ __launch_bounds__(256) void expand_row_ids(int M, const int *__restrict__ rp, int *__restrict__ rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= M) return;
    for (int j = rp[r] + lane; j < rp[r + 1]; j += 64) rows[j] = r;
}
__global__ __launch_bounds__(256) void fixed_degree_row_ptr(int n, int deg, int *rp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) rp[i] = i * deg;
}
__global__ __launch_bounds__(256) void column_starts(int K, long long nnz, const int *__restrict__ sorted_cols, int *__restrict__ cp) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c > K) return;
    long long lo = 0, hi = nnz;                     // first position whose column is >= c
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (sorted_cols[mid] < c) lo = mid + 1; else hi = mid; }
    cp[c] = (int)lo;
}
__global__ __launch_bounds__(256) void row_similarity(int M, int K, const int *__restrict__ rp, const int *__restrict__ ci, const int *__restrict__ cp,
                                                      const int *__restrict__ crow, int *__restrict__ g_ci, unsigned char *__restrict__ g_w,
                                                      unsigned long long *__restrict__ acc /* [0] best overlap, [1] row length, [2] near, [3] entries */) {
    __shared__ int keys[4][kRowSimHT];
    __shared__ int cnts[4][kRowSimHT];
    __shared__ int tabs[4][kTriHT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= M) return;
    int *kk = keys[wave], *cc = cnts[wave], *tab = tabs[wave];
    const int j0 = rp[r], len = rp[r + 1] - j0;
    if (len == 0) {
        if (lane < kRowSimDeg) { g_ci[(long long)r * kRowSimDeg + lane] = -1; g_w[(long long)r * kRowSimDeg + lane] = 0; }
        return;
    }
    for (int i = lane; i < kRowSimHT; i += 64) { kk[i] = -1; cc[i] = 0; }
    for (int i = lane; i < kTriHT; i += 64) tab[i] = -1;
    __builtin_amdgcn_wave_barrier();
    const bool small = len <= kTriMaxLen;
    if (small) for (int e = lane; e < len; e += 64) set_insert(tab, ci[j0 + e]);
    const int S = min(len, kRowSimCols);
    for (int s = 0; s < S; ++s) {
        const int c = ci[j0 + (int)((long long)s * len / S)];
        const int l0 = cp[c], L = cp[c + 1] - l0;
        int start = 0, n = L;
        if (L > 64) {                               // window of 64 rows around r's own position in the column's (ascending) row list
            int lo = 0, hi = L;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (crow[l0 + mid] < r) lo = mid + 1; else hi = mid; }
            start = min(max(lo - 32, 0), L - 64);
            n = 64;
        }
        if (lane < n) {
            const int q = crow[l0 + start + lane];
            if (q != r) {
                unsigned h = mix32((unsigned)q) & (kRowSimHT - 1);
                for (int probe = 0; probe < kRowSimHT; ++probe) {
                    const int prev = atomicCAS(&kk[h], -1, q);
                    if (prev == -1 || prev == q) { atomicAdd(&cc[h], 1); break; }
                    h = (h + 1) & (kRowSimHT - 1);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int PER = kRowSimHT / 64;
    unsigned long long sc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int slot = lane + 64 * i, q = kk[slot];
        sc[i] = 0ull;
        if (q >= 0) {
            const unsigned lo = (unsigned)min(r, q), hi = (unsigned)max(r, q);
            sc[i] = ((unsigned long long)(unsigned)cc[slot] << 44) | ((unsigned long long)(mix32(lo * 0x9E3779B1u + mix32(hi)) & 0xfffu) << 32) | (unsigned)q;
        }
    }
    unsigned best_overlap = 0;
    for (int k = 0; k < kRowSimDeg; ++k) {
        unsigned long long best = 0ull;
#pragma unroll
        for (int i = 0; i < PER; ++i) best = sc[i] > best ? sc[i] : best;
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned olo = __shfl_xor((unsigned)(best & 0xffffffffull), off), ohi = __shfl_xor((unsigned)(best >> 32), off);
            const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
            best = other > best ? other : best;
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) if (sc[i] == best) sc[i] = 0ull;
        int q = -1;
        unsigned w = 0;
        if (best) {
            q = (int)(unsigned)(best & 0xffffffffull);
            const int k0 = rp[q], lq = rp[q + 1] - k0;
            if (small && lq <= kTriMaxLen) {
                for (int e = 0; e < lq; e += 64) {
                    const bool hit = (e + lane < lq) && set_has(tab, ci[k0 + e + lane]);
                    w += (unsigned)__popcll(__ballot(hit));
                }
            } else w = (unsigned)(best >> 44);          // long rows: the sampled count
            best_overlap = max(best_overlap, w);
            w = min(255u, max(1u, w));
        }
        if (lane == 0) { g_ci[(long long)r * kRowSimDeg + k] = q; g_w[(long long)r * kRowSimDeg + k] = (unsigned char)w; }
    }
    if ((r & 63) == 0) {                                // statistics from every 64th row
        unsigned near = 0;
        const long long diag = (long long)r * K / M;
        for (int e = lane; e < len; e += 64) {
            const long long d = (long long)ci[j0 + e] - diag;
            near += (d < 0 ? -d : d) < (long long)K / 64 + 1 ? 1u : 0u;
        }
        for (int off = 32; off > 0; off >>= 1) near += __shfl_xor(near, off);
        if (lane == 0) {
            atomicAdd(&acc[0], (unsigned long long)min(best_overlap, (unsigned)len)); atomicAdd(&acc[1], (unsigned long long)len);
            atomicAdd(&acc[2], (unsigned long long)near); atomicAdd(&acc[3], (unsigned long long)len);
        }
    }
}
}  // namespace

int row_similarity_graph_device(int M, int K, int64_t nnz, const int *d_rp, const int *d_ci, int **g_rp, int **g_ci, unsigned char **g_w,
                                int64_t *g_nnz, double *shared_fraction, double *near_fraction, std::string &err) {
    *g_rp = *g_ci = nullptr; *g_w = nullptr; *g_nnz = 0;
    *shared_fraction = *near_fraction = 0.0;
    if (M < 2 || K < 1 || nnz <= 0 || nnz > 0x7fffffffLL || (int64_t)M * kRowSimDeg > 0x7fffffffLL) return 1;
    Scratch tmp;
    int *rows = nullptr, *cols = nullptr, *srows = nullptr, *scols = nullptr, *cp = nullptr, *orp = nullptr, *oci = nullptr;
    unsigned char *ow = nullptr;
    unsigned long long *d_acc = nullptr, h_acc[4] = {0, 0, 0, 0};
    GC_HIP(tmp.alloc(&rows, (size_t)nnz));
    GC_HIP(tmp.alloc(&cols, (size_t)nnz));
    GC_HIP(tmp.alloc(&srows, (size_t)nnz));
    GC_HIP(tmp.alloc(&scols, (size_t)nnz));
    GC_HIP(tmp.alloc(&cp, (size_t)K + 1));
    GC_HIP(tmp.alloc(&orp, (size_t)M + 1));
    GC_HIP(tmp.alloc(&oci, (size_t)M * kRowSimDeg));
    GC_HIP(tmp.alloc(&ow, (size_t)M * kRowSimDeg));
    GC_HIP(tmp.alloc(&d_acc, 4));
    GC_HIP(hipMemsetAsync(d_acc, 0, sizeof h_acc, nullptr));
    hipLaunchKernelGGL(expand_row_ids, dim3(blocks_for(M, 4)), dim3(256), 0, nullptr, M, d_rp, rows);
    GC_HIP(hipMemcpyAsync(cols, d_ci, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToDevice, nullptr));
    int bits = 1;
    while (bits < 32 && (1LL << bits) < (long long)K) ++bits;
    void *sort_tmp = nullptr;
    size_t bytes = 0;
    GC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, cols, scols, rows, srows, (int)nnz, 0, bits, nullptr));
    GC_HIP(tmp.alloc((char **)&sort_tmp, bytes));
    GC_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp, bytes, cols, scols, rows, srows, (int)nnz, 0, bits, nullptr));   // stable: rows ascending per column
    hipLaunchKernelGGL(column_starts, dim3(blocks_for((long long)K + 1, 256)), dim3(256), 0, nullptr, K, (long long)nnz, scols, cp);
    hipLaunchKernelGGL(fixed_degree_row_ptr, dim3(blocks_for((long long)M + 1, 256)), dim3(256), 0, nullptr, M + 1, kRowSimDeg, orp);
    hipLaunchKernelGGL(row_similarity, dim3(blocks_for(M, 4)), dim3(256), 0, nullptr, M, K, d_rp, d_ci, cp, srows, oci, ow, d_acc);
    GC_HIP(hipMemcpy(h_acc, d_acc, sizeof h_acc, hipMemcpyDeviceToHost));
    GC_HIP(hipDeviceSynchronize());
    GC_HIP(hipGetLastError());
    if (h_acc[1]) *shared_fraction = (double)h_acc[0] / (double)h_acc[1];
    if (h_acc[3]) *near_fraction = (double)h_acc[2] / (double)h_acc[3];
    tmp.keep(orp); tmp.keep(oci); tmp.keep(ow);
    *g_rp = orp; *g_ci = oci; *g_w = ow; *g_nnz = (int64_t)M * kRowSimDeg;
    return 0;
}

// ---- symmetrised graph ----------------------------------------------------------------------------------------------------------
// The handshake matching of cluster_rows_graph_device pairs two clusters when each is the other's best still-free candidate.  With
// SYMMETRIC weights both ends see the same score for a pair, locally heaviest pairs always exist and every round matches a good share
// of the clusters.  With an unsymmetric pattern (15 % of the mirror entries missing on the holdout class) A's best is B, B's best is
// C, C's best is A: measured 54.7 M panel rows against 29.6 M for the symmetric pattern of the same mesh, 124 M under a random
// numbering.  So a graph with an unsymmetric pattern is clustered over G + G^T: every row gets, behind its own entries, the rows that
// point at it and that it does not hold itself (found through the pattern sorted by column: stable radix sort, sources ascending).
// Weights (optional, one byte per entry) travel with the mirrored entries.  Entries outside [0, M) are dropped.  Deterministic.
namespace {
constexpr int kSymMaxIn = 4096;     // mirrored entries considered per row (a hub COLUMN of the matrix would be a hub row of the transpose)

__global__ __launch_bounds__(256) void sym_keys(int M, const int *__restrict__ rp, const int *__restrict__ ci, int *__restrict__ key, int *__restrict__ eid,
                                                int *__restrict__ src) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= M) return;
    for (int j = rp[r] + lane; j < rp[r + 1]; j += 64) {
        const int c = ci[j];
        key[j] = (unsigned)c < (unsigned)M ? c : M;       // (invalid entries sort behind every column)
        eid[j] = j;
        src[j] = r;
    }
}
// FILL = false: cnt[r] = entries of row r of G + G^T; FILL = true: write them at out_rp[r]: own (valid) entries in their order, then
// the mirrored ones in ascending source order
template <bool FILL>
__global__ __launch_bounds__(256) void sym_rows(int M, const int *__restrict__ rp, const int *__restrict__ ci, const unsigned char *__restrict__ w,
                                                const int *__restrict__ cp, const int *__restrict__ sorted_eid, const int *__restrict__ src,
                                                int *__restrict__ cnt, const int *__restrict__ out_rp, int *__restrict__ out_ci,
                                                unsigned char *__restrict__ out_w) {
    __shared__ int tabs[4][kTriHT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= M) return;
    int *tab = tabs[wave];
    const int j0 = rp[r], len = rp[r + 1] - j0;
    const bool use_set = len <= kTriMaxLen;
    for (int i = lane; i < kTriHT; i += 64) tab[i] = -1;
    __builtin_amdgcn_wave_barrier();
    int n = 0;                                            // (wave-uniform) entries written / counted so far
    long long o = FILL ? out_rp[r] : 0;
    for (int e0 = 0; e0 < len; e0 += 64) {
        const int e = e0 + lane;
        const int c = e < len ? ci[j0 + e] : -1;
        const bool ok = (unsigned)c < (unsigned)M;
        if (ok && use_set) set_insert(tab, c);
        const unsigned long long m = __ballot(ok);
        if (FILL && ok) {
            const int at = n + (int)__popcll(m & ((1ull << lane) - 1ull));
            out_ci[o + at] = c;
            if (out_w) out_w[o + at] = w[j0 + e];
        }
        n += (int)__popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    const int i0 = cp[r], nin = min(cp[r + 1] - i0, kSymMaxIn);
    for (int k0 = 0; k0 < nin; k0 += 64) {
        const int k = k0 + lane;
        int q = -1, e = 0;
        if (k < nin) { e = sorted_eid[i0 + k]; q = src[e]; }
        const bool ok = q >= 0 && q != r && !(use_set && set_has(tab, q));
        const unsigned long long m = __ballot(ok);
        if (FILL && ok) {
            const int at = n + (int)__popcll(m & ((1ull << lane) - 1ull));
            out_ci[o + at] = q;
            if (out_w) out_w[o + at] = w[e];
        }
        n += (int)__popcll(m);
    }
    if (!FILL && lane == 0) cnt[r] = n;
}
}  // namespace

int symmetrize_graph_device(int M, int64_t nnz, const int *d_rp, const int *d_ci, const unsigned char *d_w, int **s_rp, int **s_ci,
                            unsigned char **s_w, int64_t *s_nnz, std::string &err) {
    *s_rp = *s_ci = nullptr; *s_nnz = 0;
    if (s_w) *s_w = nullptr;
    if (M < 1 || nnz <= 0 || nnz > 0x3fffffffLL) return 1;
    Scratch tmp;
    int *key = nullptr, *skey = nullptr, *eid = nullptr, *seid = nullptr, *src = nullptr, *cp = nullptr, *cnt = nullptr, *orp = nullptr, *oci = nullptr;
    unsigned char *ow = nullptr;
    GC_HIP(tmp.alloc(&key, (size_t)nnz));
    GC_HIP(tmp.alloc(&skey, (size_t)nnz));
    GC_HIP(tmp.alloc(&eid, (size_t)nnz));
    GC_HIP(tmp.alloc(&seid, (size_t)nnz));
    GC_HIP(tmp.alloc(&src, (size_t)nnz));
    GC_HIP(tmp.alloc(&cp, (size_t)M + 2));
    GC_HIP(tmp.alloc(&cnt, (size_t)M + 1));
    GC_HIP(tmp.alloc(&orp, (size_t)M + 1));
    hipLaunchKernelGGL(sym_keys, dim3(blocks_for(M, 4)), dim3(256), 0, nullptr, M, d_rp, d_ci, key, eid, src);
    int bits = 1;
    while (bits < 32 && (1LL << bits) <= (long long)M) ++bits;
    void *sort_tmp = nullptr;
    size_t bytes = 0;
    GC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, key, skey, eid, seid, (int)nnz, 0, bits, nullptr));
    GC_HIP(tmp.alloc((char **)&sort_tmp, bytes));
    GC_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp, bytes, key, skey, eid, seid, (int)nnz, 0, bits, nullptr));   // stable: sources ascending per target
    hipLaunchKernelGGL(column_starts, dim3(blocks_for((long long)M + 2, 256)), dim3(256), 0, nullptr, M + 1, (long long)nnz, skey, cp);
    GC_HIP(hipMemsetAsync(cnt + M, 0, sizeof(int), nullptr));
    hipLaunchKernelGGL(sym_rows<false>, dim3(blocks_for(M, 4)), dim3(256), 0, nullptr, M, d_rp, d_ci, d_w, cp, seid, src, cnt, (const int *)nullptr,
                       (int *)nullptr, (unsigned char *)nullptr);
    void *scan_tmp = nullptr;
    size_t sbytes = 0;
    GC_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, sbytes, cnt, orp, M + 1, nullptr));
    GC_HIP(tmp.alloc((char **)&scan_tmp, sbytes));
    GC_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, sbytes, cnt, orp, M + 1, nullptr));
    int total = 0;
    GC_HIP(hipMemcpy(&total, orp + M, sizeof(int), hipMemcpyDeviceToHost));
    if (total <= 0) return 1;
    GC_HIP(tmp.alloc(&oci, (size_t)total));
    if (d_w && s_w) GC_HIP(tmp.alloc(&ow, (size_t)total));
    hipLaunchKernelGGL(sym_rows<true>, dim3(blocks_for(M, 4)), dim3(256), 0, nullptr, M, d_rp, d_ci, d_w, cp, seid, src, (int *)nullptr, orp, oci, ow);
    GC_HIP(hipDeviceSynchronize());
    GC_HIP(hipGetLastError());
    tmp.keep(orp); tmp.keep(oci);
    if (ow) { tmp.keep(ow); *s_w = ow; }
    *s_rp = orp; *s_ci = oci; *s_nnz = total;
    return 0;
}

// ---- run graph ---------------------------------------------------------------------------------------------------------------
// A matrix in a numbering WITH locality (a SuiteSparse file in the order its generator wrote, RCM) whose row blocks are nevertheless
// cut short by the panel capacity: 64 CONSECUTIVE rows are a 1-D run of the numbering and share less than 64 well-chosen rows would
// (holdout class: 11.7 dictionary rows per matrix row in file order, 7.4 fully clustered).  The full clustering pays for its gain
// with the reordered form -- C rows scattered, two passes over C through a staging buffer -- which needs >= 40 % fewer panel rows to
// win.  The middle way keeps the numbering's locality where the kernel needs it: the units that are clustered are RUNS of `run`
// consecutive rows (one wavefront's 16 row slots: its C accesses stay 64-byte runs per column, no staging, natural B panels), and a
// row block is 4 runs chosen over the graph of runs.  This builds that graph: node R = rows [R run, R run + run); R -> c / run for
// every entry of its rows, deduplicated, self loops dropped.  One wavefront per run, LDS hash set, count pass + fill pass.
namespace {
constexpr int kRunHT = 1024;      // hash slots per run (<= 512 distinct neighbouring runs)

template <bool FILL>
__global__ __launch_bounds__(256) void run_graph_rows(int Mr, int run, const int *__restrict__ rp, const int *__restrict__ ci, int *__restrict__ cnt,
                                                      const int *__restrict__ out_rp, int *__restrict__ out_ci, unsigned char *__restrict__ out_w,
                                                      int *__restrict__ overflow) {
    // out_w: how many entries of the run's rows lie in the neighbouring run (1 .. 255): runs that are strongly connected share their
    // neighbourhood -- with unit weights every neighbouring run of a mesh ties
    __shared__ int tabs[4][kRunHT];
    __shared__ int cnts[4][kRunHT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int R = blockIdx.x * 4 + wave;
    if (R >= Mr) return;
    int *tab = tabs[wave], *cc = cnts[wave];
    for (int i = lane; i < kRunHT; i += 64) { tab[i] = -1; cc[i] = 0; }
    __builtin_amdgcn_wave_barrier();
    const int j0 = rp[R * run], j1 = rp[(R + 1) * run];
    int distinct = 0;
    for (int base = j0; base < j1; base += 64) {          // (whole wavefronts: the ballot below)
        const int j = base + lane;
        bool fresh = false;
        if (j < j1) {
            const int q = ci[j] / run;
            if (q != R && q < Mr) {
                unsigned h = mix32((unsigned)q) & (kRunHT - 1);
                for (int probe = 0; probe < kRunHT; ++probe) {
                    const int prev = atomicCAS(&tab[h], -1, q);
                    if (prev == -1 || prev == q) { fresh = prev == -1; if (FILL) atomicAdd(&cc[h], 1); break; }
                    h = (h + 1) & (kRunHT - 1);
                }
            }
        }
        distinct += (int)__popcll(__ballot(fresh));
    }
    __builtin_amdgcn_wave_barrier();
    if (!FILL) {
        if (lane == 0) { cnt[R] = distinct; if (distinct > kRunHT / 2) atomicAdd(overflow, 1); }
        return;
    }
    int o = out_rp[R];
    for (int i0 = 0; i0 < kRunHT; i0 += 64) {
        const int q = tab[i0 + lane];
        const unsigned long long m = __ballot(q >= 0);
        if (q >= 0) {
            const int at = o + (int)__popcll(m & ((1ull << lane) - 1ull));
            out_ci[at] = q;
            out_w[at] = (unsigned char)min(255, max(1, cc[i0 + lane]));
        }
        o += (int)__popcll(m);
    }
}
// order[i * run + j] = order_r[i] * run + j; rows past the last whole run keep their places at the end; cut = 1 every `per` runs
__global__ __launch_bounds__(256) void expand_run_order(int M, int Mr, int run, int per, const int *__restrict__ order_r, const int *__restrict__ group,
                                                        int *__restrict__ order, unsigned char *__restrict__ cut) {
    // group (may be null): cluster of every run at the level where clusters hold up to `per` runs -- a block starts where it changes
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= M) return;
    const int i = p / run, j = p % run;
    order[p] = i < Mr ? order_r[i] * run + j : p;
    bool c = j == 0 && (i == 0 || i == Mr);
    if (j == 0 && i > 0 && i < Mr) c = group ? group[order_r[i]] != group[order_r[i - 1]] : i % per == 0;
    cut[p] = c ? 1 : 0;
}
}  // namespace

int run_graph_device(int M, int run, const int *d_rp, const int *d_ci, int **r_rp, int **r_ci, unsigned char **r_w, int64_t *r_nnz, int *Mr_out,
                     std::string &err) {
    *r_rp = *r_ci = nullptr; *r_w = nullptr; *r_nnz = 0; *Mr_out = 0;
    const int Mr = M / run;
    if (Mr < 2 || run < 1) return 1;
    Scratch tmp;
    int *cnt = nullptr, *orp = nullptr, *oci = nullptr, *d_over = nullptr;
    unsigned char *ow = nullptr;
    GC_HIP(tmp.alloc(&cnt, (size_t)Mr + 1));
    GC_HIP(tmp.alloc(&orp, (size_t)Mr + 1));
    GC_HIP(tmp.alloc(&d_over, 1));
    GC_HIP(hipMemsetAsync(d_over, 0, sizeof(int), nullptr));
    GC_HIP(hipMemsetAsync(cnt + Mr, 0, sizeof(int), nullptr));
    hipLaunchKernelGGL(run_graph_rows<false>, dim3(blocks_for(Mr, 4)), dim3(256), 0, nullptr, Mr, run, d_rp, d_ci, cnt, (const int *)nullptr, (int *)nullptr, (unsigned char *)nullptr, d_over);
    void *scan_tmp = nullptr;
    size_t bytes = 0;
    GC_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, cnt, orp, Mr + 1, nullptr));
    GC_HIP(tmp.alloc((char **)&scan_tmp, bytes));
    GC_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp, bytes, cnt, orp, Mr + 1, nullptr));
    int total = 0, over = 0;
    GC_HIP(hipMemcpy(&total, orp + Mr, sizeof(int), hipMemcpyDeviceToHost));
    GC_HIP(hipMemcpy(&over, d_over, sizeof(int), hipMemcpyDeviceToHost));
    if (over || total <= 0) return 1;                     // (runs with more than 512 neighbouring runs: not a matrix for this)
    GC_HIP(tmp.alloc(&oci, (size_t)total));
    GC_HIP(tmp.alloc(&ow, (size_t)total));
    hipLaunchKernelGGL(run_graph_rows<true>, dim3(blocks_for(Mr, 4)), dim3(256), 0, nullptr, Mr, run, d_rp, d_ci, (int *)nullptr, orp, oci, ow, (int *)nullptr);
    GC_HIP(hipDeviceSynchronize());
    GC_HIP(hipGetLastError());
    tmp.keep(orp); tmp.keep(oci); tmp.keep(ow);
    *r_rp = orp; *r_ci = oci; *r_w = ow; *r_nnz = total; *Mr_out = Mr;
    return 0;
}

int expand_run_order_device(int M, int Mr, int run, int runs_per_block, const int *d_order_r, const int *d_group, int **d_order, unsigned char **d_cut,
                            std::string &err) {
    *d_order = nullptr; *d_cut = nullptr;
    Scratch tmp;
    int *o = nullptr;
    unsigned char *c = nullptr;
    GC_HIP(tmp.alloc(&o, (size_t)M));
    GC_HIP(tmp.alloc(&c, (size_t)M));
    hipLaunchKernelGGL(expand_run_order, dim3(blocks_for(M, 256)), dim3(256), 0, nullptr, M, Mr, run, runs_per_block, d_order_r, d_group, o, c);
    GC_HIP(hipDeviceSynchronize());
    tmp.keep(o); tmp.keep(c);
    *d_order = o; *d_cut = c;
    return 0;
}

}  // namespace sx
