// row_cluster.hip -- graph-compact row blocks for the LDS-panel plan.
//
// The LDS panel of a row block is as large as the block's dictionary (its distinct columns), and it is copied once per 16-column N
// tile.  The plan forms blocks from CONSECUTIVE rows; for a 3-D mesh matrix in natural ordering 64 consecutive rows are a 1-D run of
// nodes whose neighbourhoods barely overlap: 57 rows of the 27-point / 3-dof FEM matrix need 576 B rows (10 per row), 18 GB of L2 -> LDS
// panel traffic per SpMM at N = 128 and B re-read ~4x from HBM at N = 16.  The rows of a matrix are independent -- any order of the rows
// gives the same sums -- so the plan may visit them brick by brick: a 4 x 4 x 4 brick of grid lines holds 192 rows whose 64-row
// blocks need ~330 B rows each.  Measured with the rows permuted by hand (tools/perm_exp.py): N = 16 -5 %, N = 32 -13 %, N = 128 -15 %.
// The reference schedules its non-zeros for the same reason -- to keep the on-chip B window hot (generate_edge_list_for_all_PEs,
// sparse_helper.h:345-403) -- with a different mechanism (row % 64 interleaving over PEs, 4096-column windows).
//
// This file: (1) infer the grid strides from the column offsets of sampled rows (host, a few hundred rows); (2) sort the rows by
// (brick, position inside the brick) on the device; (3) gather the CSR rows in that order for the plan builder; (4) the per-slot row
// table the kernel uses to address C.  Matrices without such a structure (unstructured meshes, random columns) are left alone.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <map>

#include "row_cluster.h"
#include "thread_stream.h"

namespace sx {
namespace {

#define RC_HIP(x)                                                                                     \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return 2; }     \
    } while (0)

// offsets of one row -> twice the centres of its column clusters (consecutive columns, gap <= 4, form a cluster)
void cluster_centres(const std::vector<long long> &off, std::vector<long long> &centres) {
    centres.clear();
    size_t i = 0;
    while (i < off.size()) {
        size_t j = i;
        while (j + 1 < off.size() && off[j + 1] - off[j] <= 4) ++j;
        centres.push_back(off[i] + off[j]);   // TWICE the centre (exact for clusters of any width; spacings are halved below)
        i = j + 1;
    }
}

// Runs along a grid line: nr = ceil(s2 / run_cap) runs of nearly equal length <= run_cap, so that every brick of b2 x b3 such runs has
// at most run_cap * b2 * b3 rows and no run is a small remainder (s2 = 330, run_cap 16: 21 runs of 15 / 16 rows).
__global__ __launch_bounds__(256) void brick_keys(int M, long long s2, long long s3, int b2, int b3, long long nr, long long n2, int run_cap, int sg,
                                                  unsigned long long *keys, int *vals) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    long long c3 = 0, rem = r;
    if (s3 > 0) { c3 = r / s3; rem = r % s3; }
    const long long c2 = rem / s2, c1 = rem % s2;
    const long long run = (c1 * nr) / s2, run_start = (run * s2 + nr - 1) / nr;
    // bricks are laid out in groups of sg x sg brick columns (lines x planes), all runs of a line inside a column: the workgroups that run
    // together on an XCD then share B rows in both cross directions, not only along the line
    const unsigned long long c2b = (unsigned long long)(c2 / b2), c3b = (unsigned long long)(c3 / b3), g = (unsigned long long)sg;
    const unsigned long long n2g = ((unsigned long long)n2 + g - 1) / g;
    const unsigned long long brick = (((c3b / g) * n2g + (c2b / g)) * (g * g) + ((c3b % g) * g + (c2b % g))) * (unsigned long long)nr +
                                     (unsigned long long)run;
    const unsigned long long inner = ((unsigned long long)(c3 % b3) * b2 + (unsigned long long)(c2 % b2)) * (unsigned long long)run_cap + (unsigned long long)(c1 - run_start);
    keys[r] = brick * 4096ull + inner;          // (b2 * b3 * run_cap <= 4096)
    vals[r] = r;
}

__global__ __launch_bounds__(256) void brick_cuts(int M, const unsigned long long *__restrict__ sorted_keys, unsigned char *cut) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < M) cut[i] = (i == 0 || (sorted_keys[i] >> 12) != (sorted_keys[i - 1] >> 12)) ? 1 : 0;
}

__global__ __launch_bounds__(256) void perm_lengths(int M, const int *__restrict__ rp, const int *__restrict__ perm, long long *len) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < M) { const int r = perm[i]; len[i] = (long long)rp[r + 1] - rp[r]; }
    if (i == M) len[i] = 0;
}

__global__ __launch_bounds__(256) void perm_row_ptr(int M, const long long *__restrict__ off, int *__restrict__ nrp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i <= M) nrp[i] = (int)off[i];
}

// one 16-lane group per row
__global__ __launch_bounds__(256) void perm_gather(int M, const int *__restrict__ rp, const int *__restrict__ ci, const float *__restrict__ va,
                                                   const int *__restrict__ perm, const int *__restrict__ nrp, int *__restrict__ nci,
                                                   float *__restrict__ nva) {
    const int i = blockIdx.x * 16 + threadIdx.x / 16, l = threadIdx.x % 16;
    if (i >= M) return;
    const int r = perm[i], s = rp[r], n = rp[r + 1] - s, d = nrp[i];
    for (int e = l; e < n; e += 16) { nci[d + e] = ci[s + e]; nva[d + e] = va[s + e]; }
}

__global__ __launch_bounds__(256) void slot_rows(int nblk, int RB, const int *__restrict__ blk_row, const int *__restrict__ perm,
                                                 int *__restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)nblk * RB) return;
    const int b = (int)(t / RB), s = (int)(t % RB);
    const int r0 = blk_row[b], r1 = blk_row[b + 1];
    out[t] = r1 > r0 ? perm[min(r0 + s, r1 - 1)] : 0;
}

}  // namespace

bool detect_grid_strides(int M, const std::vector<int> &rows, const std::vector<std::vector<int>> &cols, GridStrides *out) {
    // per sampled row: s2 = smallest distance between neighbouring cluster centres, s3 = smallest distance between neighbouring
    // GROUPS of clusters (groups = centres closer than 1.5 x s2); the matrix has the structure when 3/4 of the rows agree
    std::map<long long, int> v2, v3;
    int used = 0;
    std::vector<long long> off, c, g;
    for (size_t k = 0; k < rows.size(); ++k) {
        if (cols[k].size() < 3) continue;
        off.assign(cols[k].begin(), cols[k].end());
        for (auto &x : off) x -= rows[k];
        cluster_centres(off, c);
        if (c.size() < 3) continue;
        long long s2 = 0;
        for (size_t i = 1; i < c.size(); ++i) { const long long d = c[i] - c[i - 1]; if (!s2 || d < s2) s2 = d; }
        if (s2 < 16) continue;                              // (doubled: strides below 8 rows are not worth bricks)
        g.clear();
        size_t i = 0;
        while (i < c.size()) {
            size_t j = i;
            while (j + 1 < c.size() && 2 * (c[j + 1] - c[j]) <= 3 * s2) ++j;
            g.push_back((c[i] + c[j]) / 2);   // (still in doubled units)
            i = j + 1;
        }
        long long s3 = 0;
        for (size_t q = 1; q < g.size(); ++q) { const long long d = g[q] - g[q - 1]; if (!s3 || d < s3) s3 = d; }
        if ((s2 & 1) || (s3 & 1)) continue;                 // (doubled units: a true stride is even here)
        ++used;
        ++v2[s2 / 2];
        ++v3[g.size() >= 2 ? s3 / 2 : 0];
    }
    if (used < 8) return false;
    auto mode = [](const std::map<long long, int> &m, int *cnt) { long long best = 0; *cnt = 0; for (auto &kv : m) if (kv.second > *cnt) { *cnt = kv.second; best = kv.first; } return best; };
    int n2 = 0, n3 = 0;
    const long long s2 = mode(v2, &n2), s3 = mode(v3, &n3);
    if (4 * n2 < 3 * used || 4 * n3 < 3 * used) return false;
    if (s2 < 8 || s2 > M) return false;
    if (s3 != 0 && (s3 < 2 * s2 || s3 > M)) return false;
    out->s2 = s2;
    out->s3 = s3;
    return true;
}

int build_brick_order_device(int M, GridStrides s, int run_rows, int b2, int b3, int super_group, int **d_perm, unsigned char **d_cut,
                             std::string &err) {
    if (super_group < 1) super_group = 1;
    *d_perm = nullptr;
    *d_cut = nullptr;
    if (M <= 0 || run_rows < 1 || (long long)run_rows * b2 * b3 > 4096) { err = "brick too large"; return 2; }
    const long long nr = (s.s2 + run_rows - 1) / run_rows;                       // runs (bricks) along a grid line
    const long long lines = s.s3 > 0 ? (s.s3 + s.s2 - 1) / s.s2 : ((long long)M + s.s2 - 1) / s.s2;
    const long long n2 = (lines + b2 - 1) / b2;
    unsigned long long *k_in = nullptr, *k_out = nullptr;
    int *v_in = nullptr, *v_out = nullptr;
    unsigned char *cut = nullptr;
    void *tmp = nullptr;
    size_t bytes = 0;
    auto cleanup = [&]() { (void)hipFree(k_in); (void)hipFree(k_out); (void)hipFree(v_in); (void)hipFree(tmp); };
    hipError_t e = hipMalloc((void **)&k_in, sizeof(unsigned long long) * (size_t)M);
    if (e == hipSuccess) e = hipMalloc((void **)&k_out, sizeof(unsigned long long) * (size_t)M);
    if (e == hipSuccess) e = hipMalloc((void **)&v_in, sizeof(int) * (size_t)M);
    if (e == hipSuccess) e = hipMalloc((void **)&v_out, sizeof(int) * (size_t)M);
    if (e == hipSuccess) e = hipMalloc((void **)&cut, (size_t)M);
    if (e != hipSuccess) { cleanup(); (void)hipFree(v_out); (void)hipFree(cut); err = hipGetErrorString(e); return 2; }
    hipLaunchKernelGGL(brick_keys, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, nullptr, M, s.s2, s.s3, b2, b3, nr, n2, run_rows, super_group, k_in, v_in);
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k_in, k_out, v_in, v_out, M, 0, 64, nullptr);
    if (e == hipSuccess) e = hipMalloc(&tmp, bytes);
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(tmp, bytes, k_in, k_out, v_in, v_out, M, 0, 64, nullptr);
    if (e == hipSuccess) hipLaunchKernelGGL(brick_cuts, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, nullptr, M, k_out, cut);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    cleanup();
    if (e != hipSuccess) { (void)hipFree(v_out); (void)hipFree(cut); err = hipGetErrorString(e); return 2; }
    *d_perm = v_out;
    *d_cut = cut;
    return 0;
}

int permute_csr_rows_device(int M, int64_t nnz, const int *d_rp, const int *d_ci, const float *d_v, const int *d_perm, int **o_rp,
                            int **o_ci, float **o_v, std::string &err) {
    *o_rp = nullptr; *o_ci = nullptr; *o_v = nullptr;
    long long *len = nullptr, *off = nullptr;
    void *tmp = nullptr;
    size_t bytes = 0;
    int *nrp = nullptr, *nci = nullptr;
    float *nva = nullptr;
    auto fail = [&](hipError_t e) { (void)hipFree(len); (void)hipFree(off); (void)hipFree(tmp); (void)hipFree(nrp); (void)hipFree(nci); (void)hipFree(nva);
                                    err = hipGetErrorString(e); return 2; };
    hipError_t e = hipMalloc((void **)&len, sizeof(long long) * ((size_t)M + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&off, sizeof(long long) * ((size_t)M + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&nrp, sizeof(int) * ((size_t)M + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&nci, sizeof(int) * (size_t)std::max<int64_t>(nnz, 1));
    if (e == hipSuccess) e = hipMalloc((void **)&nva, sizeof(float) * (size_t)std::max<int64_t>(nnz, 1));
    if (e != hipSuccess) return fail(e);
    hipLaunchKernelGGL(perm_lengths, dim3((unsigned)((M + 256) / 256)), dim3(256), 0, nullptr, M, d_rp, d_perm, len);
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, len, off, M + 1, nullptr);
    if (e == hipSuccess) e = hipMalloc(&tmp, bytes);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, bytes, len, off, M + 1, nullptr);
    if (e != hipSuccess) return fail(e);
    hipLaunchKernelGGL(perm_row_ptr, dim3((unsigned)((M + 256) / 256)), dim3(256), 0, nullptr, M, off, nrp);
    hipLaunchKernelGGL(perm_gather, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, nullptr, M, d_rp, d_ci, d_v, d_perm, nrp, nci, nva);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(e);
    (void)hipFree(len); (void)hipFree(off); (void)hipFree(tmp);
    *o_rp = nrp; *o_ci = nci; *o_v = nva;
    return 0;
}

int build_slot_rows_device(int nblk, int RB, const int *d_blk_row, const int *d_perm, int **d_slot_row, std::string &err) {
    *d_slot_row = nullptr;
    int *out = nullptr;
    RC_HIP(hipMalloc((void **)&out, sizeof(int) * (size_t)std::max(1, nblk) * (size_t)RB));
    if (nblk > 0)
        hipLaunchKernelGGL(slot_rows, dim3((unsigned)(((long long)nblk * RB + 255) / 256)), dim3(256), 0, nullptr, nblk, RB, d_blk_row, d_perm, out);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(out); err = hipGetErrorString(e); return 2; }
    *d_slot_row = out;
    return 0;
}

}  // namespace sx
