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
//   columns     : `len` draws uniform over [0,K) (bandwidth 0) or over the band [row-bw, row+bw]
//                 (bandwidth bw > 0: FEM-like locality), sorted ascending, duplicates pushed to the next
//                 free column, then clamped from the top so all stay < K (strictly increasing)
//   values      : U[-1, 1) with 24 random bits: k * 2^-23 - 1 (exact in fp32)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "graph_cluster.h"
#include "plan_device.h"
#include "row_cluster.h"
#include "sextans_amd.h"
#include "thread_stream.h"

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
// Generator spec shared by host and device code.  kind 0: Poisson row lengths, columns uniform over
// [0,K) (bw == 0) or over the band [row-bw, row+bw] (bw > 0).  kind 1: 3-D finite-element-like
// matrix (27-point node stencil on an nx*ny*nz grid, dof unknowns per node, dense dof x dof blocks):
// the structure of SuiteSparse FEM matrices such as Boeing/pcrystk02 (3 dof, ~69 nnz/row).
// kind 2: power-law row lengths (web/social-graph-like skew, what the sweep harness needs for its load-balancing
// cases): P(len >= x) = (xmin / x)^tail for x in [xmin, max_len], drawn by integer inverse CDF over
// quarter-octave buckets (thresholds and bucket edges are computed once on the host in long double and handed
// to both generators, so host and device agree bit for bit); columns are one uniform draw from each of `len`
// equal strata of [0, K) -- distinct, ascending, O(len) -- values as for kind 0.
struct Spec {
    int kind, K, bw, nx, ny, nz, dof;
    uint64_t seed;
    const int *p_rp = nullptr, *p_ci = nullptr;   // kind 5: the pattern P (host pointers for the host generator, device pointers on the device)
};

SX_HD uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    memcpy(&u, &f, 4);
#endif
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
constexpr uint64_t kBellSalt = 0x62656c6cULL;   // "bell"

SX_HD int fem_neighbors(const Spec &sp, int node, int *nb) {
    const int x = node % sp.nx, y = (node / sp.nx) % sp.ny, z = node / (sp.nx * sp.ny);
    int n = 0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx, yy = y + dy, zz = z + dz;
                if (xx < 0 || xx >= sp.nx || yy < 0 || yy >= sp.ny || zz < 0 || zz >= sp.nz) continue;
                if (nb) nb[n] = xx + sp.nx * (yy + sp.ny * zz);
                ++n;
            }
    return n;
}

// kind 3: 2-D stencil on an nx * ny grid, `bw` = 5 (von Neumann) or 9 (Moore) points, dof unknowns per node: the structure
// of 2-D PDE discretisations in SuiteSparse (ecology, G2_circuit-like grids, Poisson problems).
SX_HD int stencil2d_neighbors(const Spec &sp, int node, int *nb) {
    const int x = node % sp.nx, y = node / sp.nx;
    int n = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            if (sp.bw == 5 && dx != 0 && dy != 0) continue;
            const int xx = x + dx, yy = y + dy;
            if (xx < 0 || xx >= sp.nx || yy < 0 || yy >= sp.ny) continue;
            if (nb) nb[n] = xx + sp.nx * yy;
            ++n;
        }
    return n;
}
// kind 4: KKT / arrow block structure  [[H, A^T, U], [A, 0, U], [V, V, D]]  of constrained optimisation and bordered
// systems: n = sp.nx variables with a pentadiagonal H, m = n / 2 constraints each tying variables 2j, 2j+1, 2j+2, and
// a = sp.ny border ("arrow") rows/columns: every row carries the a border columns, every border row holds every 16th
// column of the rest plus the border block.  Short rows without locality between blocks + a few very long rows.
SX_HD int kkt_row(const Spec &sp, int row, int *c) {
    const int n = sp.nx, m = n / 2, a = sp.ny;
    int k = 0;
    if (row < n) {
        for (int d = -2; d <= 2; ++d) { const int cc = row + d; if (cc >= 0 && cc < n) { if (c) c[k] = cc; ++k; } }
        const int j1 = row / 2;
        if ((row & 1) == 0 && row >= 2 && j1 - 1 < m) { if (c) c[k] = n + j1 - 1; ++k; }
        if (j1 < m) { if (c) c[k] = n + j1; ++k; }
    } else if (row < n + m) {
        const int j = row - n;
        for (int d = 0; d < 3; ++d) { const int cc = 2 * j + d; if (cc < n) { if (c) c[k] = cc; ++k; } }
    } else {
        for (int cc = (row - n - m) % 16; cc < n + m; cc += 16) { if (c) c[k] = cc; ++k; }
    }
    for (int e = 0; e < a; ++e) { if (c) c[k] = n + m + e; ++k; }
    return k;
}

// kind 5: kron(T_n, P) -- n = sp.nx copies of a given pm x pk pattern P (sp.ny x sp.nz; the holdout class uses the pattern of a real
// SuiteSparse matrix, nasa4704) on the block diagonal, each coupled to its two neighbours through the same pattern (T_n tridiagonal):
// row i * pm + p holds, for j = i - 1, i, i + 1 inside [0, n), the columns j * pk + P[p][*] -- ascending.  Values U(-1, 1) keyed by
// (row, column of the UNMODIFIED product), halved in the off-diagonal blocks ("the same pattern scaled").  Variants (sp.bw, bit set):
//   1  rectangular: every third column (c % 3 == 2) is dropped with its entries, the rest renumbered c' = 2 * (c / 3) + c % 3;
//   2  unsymmetric pattern: 30 % of the strictly lower entries (c < row; hash of (row, c)) are dropped.
constexpr uint64_t kKronDropSalt = 0x64726f70ULL;   // "drop"
SX_HD bool kron_keep(const Spec &sp, int row, int64_t c) {
    if ((sp.bw & 1) && c % 3 == 2) return false;
    if ((sp.bw & 2) && c < (int64_t)row && mulhi64(rnd(sp.seed ^ kKronDropSalt, (uint64_t)row, (uint64_t)c), 10) < 3) return false;
    return true;
}
SX_HD int kron_row(const Spec &sp, int row, int *c, float *v) {
    const int pm = sp.ny, pk = sp.nz, i = row / pm, p = row % pm;
    const int q0 = sp.p_rp[p], q1 = sp.p_rp[p + 1];
    int k = 0;
    for (int j = i - 1; j <= i + 1; ++j) {
        if (j < 0 || j >= sp.nx) continue;
        for (int q = q0; q < q1; ++q) {
            const int64_t col = (int64_t)j * pk + sp.p_ci[q];
            if (!kron_keep(sp, row, col)) continue;
            if (c) {
                c[k] = (int)((sp.bw & 1) ? 2 * (col / 3) + col % 3 : col);
                const float x = u01m1(rnd(sp.seed ^ kValSalt, (uint64_t)row, (uint64_t)col));
                v[k] = j == i ? x : 0.5f * x;
            }
            ++k;
        }
    }
    return k;
}

SX_HD int row_len(const Spec &sp, const uint64_t *table, int row) {
    if (sp.kind == 5) return kron_row(sp, row, nullptr, nullptr);
    if (sp.kind == 1) return fem_neighbors(sp, row / sp.dof, nullptr) * sp.dof;
    if (sp.kind == 3) return stencil2d_neighbors(sp, row / sp.dof, nullptr) * sp.dof;
    if (sp.kind == 4) return kkt_row(sp, row, nullptr);
    if (sp.kind == 2) {
        // table[0 .. nb) = thresholds (CDF at the upper edge of bucket b, 64-bit fixed point), table[kTable/2 + b] =
        // lower edge of bucket b (edges[nb] = max_len + 1); nb = sp.bw
        const uint64_t u = rnd(sp.seed, (uint64_t)row, 0);
        int b = 0;
        while (b < sp.bw - 1 && u >= table[b]) ++b;
        const uint64_t lo = table[kTable / 2 + b], hi = table[kTable / 2 + b + 1];
        const int len = (int)(lo + mulhi64(rnd(sp.seed, (uint64_t)row, 0x70776c), hi - lo));
        return len < sp.K ? len : sp.K;
    }
    const uint64_t u = rnd(sp.seed, (uint64_t)row, 0);
    int len = 0;
    while (len < kTable - 1 && u >= table[len]) ++len;
    int span = sp.K;
    if (sp.bw > 0) {
        const int lo = row - sp.bw < 0 ? 0 : row - sp.bw;
        const int hi = row + sp.bw >= sp.K ? sp.K - 1 : row + sp.bw;
        span = hi >= lo ? hi - lo + 1 : (2 * sp.bw + 1 < sp.K ? 2 * sp.bw + 1 : sp.K);
    }
    return len < span ? len : span;
}

SX_HD void fill_row(const Spec &sp, int row, int len, int *c, float *v) {
    if (sp.kind == 5) { kron_row(sp, row, c, v); return; }
    if (sp.kind == 1) {
        int nb[27];
        const int n = fem_neighbors(sp, row / sp.dof, nb);   // ascending node order by construction
        int i = 0;
        for (int a = 0; a < n; ++a)
            for (int e = 0; e < sp.dof; ++e) c[i++] = nb[a] * sp.dof + e;
    } else if (sp.kind == 3) {
        int nb[9];
        const int n = stencil2d_neighbors(sp, row / sp.dof, nb);   // ascending node order by construction
        int i = 0;
        for (int a = 0; a < n; ++a)
            for (int e = 0; e < sp.dof; ++e) c[i++] = nb[a] * sp.dof + e;
    } else if (sp.kind == 4) {
        kkt_row(sp, row, c);
    } else if (sp.kind == 2) {
        for (int i = 0; i < len; ++i) {
            const int64_t lo = (int64_t)i * sp.K / len, hi = (int64_t)(i + 1) * sp.K / len;   // len <= K: hi > lo
            c[i] = (int)(lo + (int64_t)mulhi64(rnd(sp.seed, (uint64_t)row, 1 + (uint64_t)i), (uint64_t)(hi - lo)));
        }
    } else {
        // bw == 0: columns uniform over [0, K); bw > 0: banded, uniform over [row-bw, row+bw] clipped.
        int lo = 0, span = sp.K;
        if (sp.bw > 0) {
            lo = row - sp.bw < 0 ? 0 : row - sp.bw;
            int hi = row + sp.bw >= sp.K ? sp.K - 1 : row + sp.bw;
            if (hi < lo) { hi = sp.K - 1; lo = sp.K - 1 - 2 * sp.bw < 0 ? 0 : sp.K - 1 - 2 * sp.bw; }
            span = hi - lo + 1;
        }
        for (int i = 0; i < len; ++i) {
            const int x = lo + (int)mulhi64(rnd(sp.seed, (uint64_t)row, 1 + (uint64_t)i), (uint64_t)span);
            int p = i;
            while (p > 0 && c[p - 1] > x) { c[p] = c[p - 1]; --p; }
            c[p] = x;
        }
        for (int i = 1; i < len; ++i)
            if (c[i] <= c[i - 1]) c[i] = c[i - 1] + 1;
        for (int i = len - 1; i >= 0; --i) {
            const int cap = lo + span - 1 - (len - 1 - i);
            if (c[i] > cap) c[i] = cap;
        }
    }
    for (int i = 0; i < len; ++i) v[i] = u01m1(rnd(sp.seed ^ kValSalt, (uint64_t)row, (uint64_t)i));
}

thread_local uint64_t g_powerlaw_table[kTable];   // filled by the kind-2 entry points right before gen_host/gen_device

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

// kind 2 tables (see row_len): returns the number of buckets.
int powerlaw_table(int xmin, double tail, int max_len, uint64_t *t) {
    const long double two64 = 18446744073709551616.0L;
    int nb = 0;
    long double edge = (long double)xmin;
    std::vector<uint64_t> edges;
    while (nb < kTable / 2 - 2) {
        const uint64_t e = (uint64_t)edge;
        if (!edges.empty() && e <= edges.back()) { edge *= 1.189207115002721L; continue; }   // 2^(1/4)
        if (e > (uint64_t)max_len) break;
        edges.push_back(e);
        ++nb;
        edge *= 1.189207115002721L;
    }
    edges.push_back((uint64_t)max_len + 1);
    for (int b = 0; b < nb; ++b) {
        // P(len < upper edge) = 1 - (xmin / upper)^tail ; the last bucket takes the rest
        const long double cdf = b == nb - 1 ? 1.0L : 1.0L - powl((long double)xmin / (long double)edges[(size_t)b + 1], (long double)tail);
        const long double x = cdf * two64;
        t[b] = (cdf >= 1.0L || x >= two64 - 1.0L) ? UINT64_MAX : (uint64_t)x;
    }
    for (int b = 0; b <= nb; ++b) t[kTable / 2 + b] = edges[(size_t)b];
    return nb;
}

__global__ void k_row_len(Spec sp, const uint64_t *table, int r0, int nrows, int *lens) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) lens[i] = row_len(sp, table, r0 + i);
}
__global__ void k_fill_rows(Spec sp, int r0, int nrows, const int *rp, int *col, float *val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) fill_row(sp, r0 + i, rp[i + 1] - rp[i], col + rp[i], val + rp[i]);
}
__global__ void k_uniform(float *dst, int64_t n, uint64_t seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = u01m1(rnd(seed, (uint64_t)i, 0x51));
}

// ---- P A P^T on the device (measurement infrastructure: meshes in arbitrary node orders, sextans_amd/meshgen.py) ----------------
// rows already gathered in the new order and columns relabelled: sort every row's entries by column (values travel along).
// One wavefront per row of <= 256 entries (bitonic network over 256 LDS slots), one workgroup per longer row (<= 4096).
template <int P, int T>   // P slots, T threads cooperating on one row
__device__ __forceinline__ void sort_row_lds(int *key, float *val, int n, int t) {
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < P; i += T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int a = key[i], b = key[ixj];
                    if ((a > b) == ((i & k) == 0)) {
                        key[i] = b; key[ixj] = a;
                        const float va = val[i]; val[i] = val[ixj]; val[ixj] = va;
                    }
                }
            }
            if (T == 64) __builtin_amdgcn_wave_barrier(); else __syncthreads();
        }
    (void)n;
}
__global__ __launch_bounds__(256) void k_sort_rows_wave(int M, const int *__restrict__ rp, int *ci, float *va) {
    __shared__ int key[4][256];
    __shared__ float val[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= M) return;
    const int j0 = rp[r], n = rp[r + 1] - j0;
    if (n < 2 || n > 256) return;
    for (int i = lane; i < 256; i += 64) { key[wave][i] = i < n ? ci[j0 + i] : 0x7fffffff; val[wave][i] = i < n ? va[j0 + i] : 0.f; }
    __builtin_amdgcn_wave_barrier();
    sort_row_lds<256, 64>(key[wave], val[wave], n, lane);
    for (int i = lane; i < n; i += 64) { ci[j0 + i] = key[wave][i]; va[j0 + i] = val[wave][i]; }
}
__global__ __launch_bounds__(256) void k_sort_rows_wg(const int *__restrict__ rows, const int *__restrict__ rp, int *ci, float *va) {
    __shared__ int key[4096];
    __shared__ float val[4096];
    const int r = rows[blockIdx.x], t = threadIdx.x;
    const int j0 = rp[r], n = rp[r + 1] - j0;
    for (int i = t; i < 4096; i += 256) { key[i] = i < n ? ci[j0 + i] : 0x7fffffff; val[i] = i < n ? va[j0 + i] : 0.f; }
    __syncthreads();
    sort_row_lds<4096, 256>(key, val, n, t);
    for (int i = t; i < n; i += 256) { ci[j0 + i] = key[i]; va[j0 + i] = val[i]; }
}

#define SY_HIP(call) do { if ((call) != hipSuccess) return SEXTANS_ERR_HIP; } while (0)

int gen_host(const Spec &sp, double mean, int r0, int r1, int **row_ptr, int **col_idx, float **val,
             int64_t *nnz) {
    std::vector<uint64_t> table(kTable);
    if (sp.kind == 2) memcpy(table.data(), g_powerlaw_table, sizeof(uint64_t) * kTable);
    else poisson_table(mean, table.data());
    const int nrows = r1 - r0;
    int *rp = (int *)malloc(sizeof(int) * ((size_t)nrows + 1));
    if (!rp) return SEXTANS_ERR_ALLOC;
    int64_t tot = 0;
    rp[0] = 0;
    for (int i = 0; i < nrows; ++i) {
        tot += row_len(sp, table.data(), r0 + i);
        if (tot > 0x7fffffffLL) { free(rp); return SEXTANS_ERR_INVALID; }
        rp[i + 1] = (int)tot;
    }
    int *c = (int *)malloc(sizeof(int) * (size_t)(tot ? tot : 1));
    float *v = (float *)malloc(sizeof(float) * (size_t)(tot ? tot : 1));
    if (!c || !v) { free(rp); free(c); free(v); return SEXTANS_ERR_ALLOC; }
    for (int i = 0; i < nrows; ++i) fill_row(sp, r0 + i, rp[i + 1] - rp[i], c + rp[i], v + rp[i]);
    *row_ptr = rp; *col_idx = c; *val = v; *nnz = tot;
    return SEXTANS_OK;
}

int gen_device(int device, const Spec &sp, double mean, int r0, int r1, int **d_row_ptr,
               int **d_col_idx, float **d_val, int64_t *nnz) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return SEXTANS_ERR_NO_DEVICE;
    SY_HIP(hipSetDevice(device));
    std::vector<uint64_t> table(kTable);
    if (sp.kind == 2) memcpy(table.data(), g_powerlaw_table, sizeof(uint64_t) * kTable);
    else poisson_table(mean, table.data());
    uint64_t *d_table = nullptr;
    SY_HIP(hipMalloc((void **)&d_table, sizeof(uint64_t) * kTable));
    SY_HIP(hipMemcpy(d_table, table.data(), sizeof(uint64_t) * kTable, hipMemcpyHostToDevice));
    const int nrows = r1 - r0;
    int *d_rp = nullptr;
    SY_HIP(hipMalloc((void **)&d_rp, sizeof(int) * ((size_t)nrows + 1)));
    const unsigned grid = (unsigned)((nrows + 255) / 256);
    if (nrows) hipLaunchKernelGGL(k_row_len, dim3(grid), dim3(256), 0, 0, sp, d_table, r0, nrows, d_rp + 1);
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
    if (nrows) hipLaunchKernelGGL(k_fill_rows, dim3(grid), dim3(256), 0, 0, sp, r0, nrows, d_rp, d_c, d_v);
    SY_HIP(hipDeviceSynchronize());
    SY_HIP(hipFree(d_table));
    *d_row_ptr = d_rp; *d_col_idx = d_c; *d_val = d_v; *nnz = tot;
    return SEXTANS_OK;
}

__global__ void k_uniform_bf16(uint16_t *dst, int64_t n, uint64_t seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = f32_to_bf16_rne(u01m1(rnd(seed, (uint64_t)i, 0x51)));
}
// block columns of one block row: ell_width distinct sorted values in [0, kblocks)
__global__ void k_bell_cols(Spec sp, int mblocks, int ell_width, int *block_col) {
    const int br = blockIdx.x * blockDim.x + threadIdx.x;
    if (br >= mblocks) return;
    int *c = block_col + (int64_t)br * ell_width;
    for (int i = 0; i < ell_width; ++i) {
        const int x = (int)mulhi64(rnd(sp.seed, (uint64_t)br, 1 + (uint64_t)i), (uint64_t)sp.K);
        int p = i;
        while (p > 0 && c[p - 1] > x) { c[p] = c[p - 1]; --p; }
        c[p] = x;
    }
    for (int i = 1; i < ell_width; ++i)
        if (c[i] <= c[i - 1]) c[i] = c[i - 1] + 1;
    for (int i = ell_width - 1; i >= 0; --i) {
        const int cap = sp.K - 1 - (ell_width - 1 - i);
        if (c[i] > cap) c[i] = cap;
    }
}
SX_HD uint16_t bell_value(uint64_t seed, int64_t slot, int e) {
    return f32_to_bf16_rne(u01m1(rnd(seed ^ kBellSalt, (uint64_t)slot, (uint64_t)e)));
}
__global__ void k_bell_vals(uint64_t seed, int64_t nslots, uint16_t *val) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread = 8 consecutive values
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nslots * 128; i += stride) {
        const int64_t slot = i >> 7;
        const int e0 = (int)(i & 127) * 8;
        uint16_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bell_value(seed, slot, e0 + e);
        uint4 w;
        w.x = v[0] | ((uint32_t)v[1] << 16); w.y = v[2] | ((uint32_t)v[3] << 16);
        w.z = v[4] | ((uint32_t)v[5] << 16); w.w = v[6] | ((uint32_t)v[7] << 16);
        *reinterpret_cast<uint4 *>(val + i * 8) = w;
    }
}

bool fem_ok(int nx, int ny, int nz, int dof) {
    return nx > 0 && ny > 0 && nz > 0 && dof > 0 && dof <= 8 &&
           (int64_t)nx * ny * nz * dof <= 0x7fffffffLL;
}

}  // namespace

extern "C" {

int sextans_gen_csr_host(int M, int K, double mean_nnz, int bandwidth, uint64_t seed, int r0, int r1,
                         int **row_ptr, int **col_idx, float **val, int64_t *nnz) {
    if (M < 0 || K <= 0 || r0 < 0 || r1 < r0 || r1 > M || mean_nnz <= 0 || mean_nnz > 300 || bandwidth < 0 ||
        !row_ptr || !col_idx || !val || !nnz)
        return SEXTANS_ERR_INVALID;
    const Spec sp{0, K, bandwidth, 0, 0, 0, 0, seed};
    return gen_host(sp, mean_nnz, r0, r1, row_ptr, col_idx, val, nnz);
}

int sextans_gen_csr_device(int device, int M, int K, double mean_nnz, int bandwidth, uint64_t seed, int r0,
                           int r1, int **d_row_ptr, int **d_col_idx, float **d_val, int64_t *nnz) {
    if (M < 0 || K <= 0 || r0 < 0 || r1 < r0 || r1 > M || mean_nnz <= 0 || mean_nnz > 300 || bandwidth < 0 ||
        !d_row_ptr || !d_col_idx || !d_val || !nnz)
        return SEXTANS_ERR_INVALID;
    const Spec sp{0, K, bandwidth, 0, 0, 0, 0, seed};
    return gen_device(device, sp, mean_nnz, r0, r1, d_row_ptr, d_col_idx, d_val, nnz);
}

static int powerlaw_spec(int M, int K, int xmin, int tail_x100, int max_len, uint64_t seed, int r0, int r1, Spec *sp) {
    if (M < 0 || K <= 0 || r0 < 0 || r1 < r0 || r1 > M || xmin < 1 || tail_x100 < 50 || tail_x100 > 1000 ||
        max_len < xmin)
        return SEXTANS_ERR_INVALID;
    if (max_len > K) max_len = K;
    if (max_len < xmin) return SEXTANS_ERR_INVALID;
    const int nb = powerlaw_table(xmin, tail_x100 / 100.0, max_len, g_powerlaw_table);
    *sp = Spec{2, K, nb, 0, 0, 0, 0, seed};
    return SEXTANS_OK;
}

int sextans_gen_powerlaw_host(int M, int K, int xmin, int tail_x100, int max_len, uint64_t seed, int r0, int r1,
                              int **row_ptr, int **col_idx, float **val, int64_t *nnz) {
    if (!row_ptr || !col_idx || !val || !nnz) return SEXTANS_ERR_INVALID;
    Spec sp;
    if (int rc = powerlaw_spec(M, K, xmin, tail_x100, max_len, seed, r0, r1, &sp)) return rc;
    return gen_host(sp, 1.0, r0, r1, row_ptr, col_idx, val, nnz);
}

int sextans_gen_powerlaw_device(int device, int M, int K, int xmin, int tail_x100, int max_len, uint64_t seed, int r0,
                                int r1, int **d_row_ptr, int **d_col_idx, float **d_val, int64_t *nnz) {
    if (!d_row_ptr || !d_col_idx || !d_val || !nnz) return SEXTANS_ERR_INVALID;
    Spec sp;
    if (int rc = powerlaw_spec(M, K, xmin, tail_x100, max_len, seed, r0, r1, &sp)) return rc;
    return gen_device(device, sp, 1.0, r0, r1, d_row_ptr, d_col_idx, d_val, nnz);
}

int sextans_gen_fem3d_host(int nx, int ny, int nz, int dof, uint64_t seed, int r0, int r1, int **row_ptr,
                           int **col_idx, float **val, int64_t *nnz) {
    if (!fem_ok(nx, ny, nz, dof) || !row_ptr || !col_idx || !val || !nnz) return SEXTANS_ERR_INVALID;
    const int M = nx * ny * nz * dof;
    if (r0 < 0 || r1 < r0 || r1 > M) return SEXTANS_ERR_INVALID;
    const Spec sp{1, M, 0, nx, ny, nz, dof, seed};
    return gen_host(sp, 1.0, r0, r1, row_ptr, col_idx, val, nnz);
}

int sextans_gen_fem3d_device(int device, int nx, int ny, int nz, int dof, uint64_t seed, int r0, int r1,
                             int **d_row_ptr, int **d_col_idx, float **d_val, int64_t *nnz) {
    if (!fem_ok(nx, ny, nz, dof) || !d_row_ptr || !d_col_idx || !d_val || !nnz) return SEXTANS_ERR_INVALID;
    const int M = nx * ny * nz * dof;
    if (r0 < 0 || r1 < r0 || r1 > M) return SEXTANS_ERR_INVALID;
    const Spec sp{1, M, 0, nx, ny, nz, dof, seed};
    return gen_device(device, sp, 1.0, r0, r1, d_row_ptr, d_col_idx, d_val, nnz);
}

int sextans_gen_stencil2d_host(int nx, int ny, int points, int dof, uint64_t seed, int r0, int r1, int **row_ptr, int **col_idx,
                               float **val, int64_t *nnz) {
    if (nx <= 0 || ny <= 0 || dof <= 0 || dof > 8 || (points != 5 && points != 9) || (int64_t)nx * ny * dof > 0x7fffffffLL ||
        !row_ptr || !col_idx || !val || !nnz)
        return SEXTANS_ERR_INVALID;
    const int M = nx * ny * dof;
    if (r0 < 0 || r1 < r0 || r1 > M) return SEXTANS_ERR_INVALID;
    const Spec sp{3, M, points, nx, ny, 1, dof, seed};
    return gen_host(sp, 1.0, r0, r1, row_ptr, col_idx, val, nnz);
}

int sextans_gen_stencil2d_device(int device, int nx, int ny, int points, int dof, uint64_t seed, int r0, int r1, int **d_row_ptr,
                                 int **d_col_idx, float **d_val, int64_t *nnz) {
    if (nx <= 0 || ny <= 0 || dof <= 0 || dof > 8 || (points != 5 && points != 9) || (int64_t)nx * ny * dof > 0x7fffffffLL ||
        !d_row_ptr || !d_col_idx || !d_val || !nnz)
        return SEXTANS_ERR_INVALID;
    const int M = nx * ny * dof;
    if (r0 < 0 || r1 < r0 || r1 > M) return SEXTANS_ERR_INVALID;
    const Spec sp{3, M, points, nx, ny, 1, dof, seed};
    return gen_device(device, sp, 1.0, r0, r1, d_row_ptr, d_col_idx, d_val, nnz);
}

static bool kkt_ok(int n, int arrow) { return n >= 4 && arrow >= 0 && arrow <= 64 && (int64_t)n + n / 2 + arrow <= 0x7fffffffLL; }

int sextans_gen_kkt_host(int n, int arrow, uint64_t seed, int r0, int r1, int **row_ptr, int **col_idx, float **val, int64_t *nnz) {
    if (!kkt_ok(n, arrow) || !row_ptr || !col_idx || !val || !nnz) return SEXTANS_ERR_INVALID;
    const int M = n + n / 2 + arrow;
    if (r0 < 0 || r1 < r0 || r1 > M) return SEXTANS_ERR_INVALID;
    const Spec sp{4, M, 0, n, arrow, 1, 1, seed};
    return gen_host(sp, 1.0, r0, r1, row_ptr, col_idx, val, nnz);
}

int sextans_gen_kkt_device(int device, int n, int arrow, uint64_t seed, int r0, int r1, int **d_row_ptr, int **d_col_idx,
                           float **d_val, int64_t *nnz) {
    if (!kkt_ok(n, arrow) || !d_row_ptr || !d_col_idx || !d_val || !nnz) return SEXTANS_ERR_INVALID;
    const int M = n + n / 2 + arrow;
    if (r0 < 0 || r1 < r0 || r1 > M) return SEXTANS_ERR_INVALID;
    const Spec sp{4, M, 0, n, arrow, 1, 1, seed};
    return gen_device(device, sp, 1.0, r0, r1, d_row_ptr, d_col_idx, d_val, nnz);
}

// kron(T_n, P): see kron_row above.  The pattern is validated here (monotone row_ptr, ascending in-range columns).
static int kron_spec(int n, int pm, int pk, const int *p_row_ptr, const int *p_col_idx, int variant, uint64_t seed, int r0, int r1, Spec *sp) {
    if (n < 1 || pm < 1 || pk < 1 || !p_row_ptr || !p_col_idx || variant < 0 || variant > 3 || (int64_t)n * pm > 0x7fffffffLL ||
        (int64_t)n * pk > 0x7fffffffLL || r0 < 0 || r1 < r0 || (int64_t)r1 > (int64_t)n * pm || p_row_ptr[0] != 0)
        return SEXTANS_ERR_INVALID;
    for (int p = 0; p < pm; ++p) {
        if (p_row_ptr[p + 1] < p_row_ptr[p]) return SEXTANS_ERR_INVALID;
        for (int q = p_row_ptr[p]; q < p_row_ptr[p + 1]; ++q)
            if (p_col_idx[q] < 0 || p_col_idx[q] >= pk || (q > p_row_ptr[p] && p_col_idx[q] <= p_col_idx[q - 1])) return SEXTANS_ERR_INDEX;
    }
    const int64_t K0 = (int64_t)n * pk;
    const int K = (int)((variant & 1) ? 2 * (K0 / 3) + (K0 % 3 < 2 ? K0 % 3 : 2) : K0);
    *sp = Spec{5, K, variant, n, pm, pk, 1, seed};
    return SEXTANS_OK;
}

int sextans_gen_kron_host(int n, int pm, int pk, const int *p_row_ptr, const int *p_col_idx, int variant, uint64_t seed, int r0, int r1,
                          int **row_ptr, int **col_idx, float **val, int64_t *nnz, int *K_out) {
    if (!row_ptr || !col_idx || !val || !nnz) return SEXTANS_ERR_INVALID;
    Spec sp;
    if (int rc = kron_spec(n, pm, pk, p_row_ptr, p_col_idx, variant, seed, r0, r1, &sp)) return rc;
    sp.p_rp = p_row_ptr; sp.p_ci = p_col_idx;
    if (K_out) *K_out = sp.K;
    return gen_host(sp, 1.0, r0, r1, row_ptr, col_idx, val, nnz);
}

int sextans_gen_kron_device(int device, int n, int pm, int pk, const int *p_row_ptr, const int *p_col_idx, int variant, uint64_t seed,
                            int r0, int r1, int **d_row_ptr, int **d_col_idx, float **d_val, int64_t *nnz, int *K_out) {
    if (!d_row_ptr || !d_col_idx || !d_val || !nnz) return SEXTANS_ERR_INVALID;
    Spec sp;
    if (int rc = kron_spec(n, pm, pk, p_row_ptr, p_col_idx, variant, seed, r0, r1, &sp)) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SEXTANS_ERR_NO_DEVICE;
    SY_HIP(hipSetDevice(device));
    int *d_prp = nullptr, *d_pci = nullptr;
    const size_t pn = (size_t)p_row_ptr[pm];
    if (hipMalloc((void **)&d_prp, sizeof(int) * ((size_t)pm + 1)) != hipSuccess || hipMalloc((void **)&d_pci, sizeof(int) * (pn ? pn : 1)) != hipSuccess ||
        hipMemcpy(d_prp, p_row_ptr, sizeof(int) * ((size_t)pm + 1), hipMemcpyHostToDevice) != hipSuccess ||
        (pn && hipMemcpy(d_pci, p_col_idx, sizeof(int) * pn, hipMemcpyHostToDevice) != hipSuccess)) {
        (void)hipFree(d_prp); (void)hipFree(d_pci);
        return SEXTANS_ERR_HIP;
    }
    sp.p_rp = d_prp; sp.p_ci = d_pci;
    if (K_out) *K_out = sp.K;
    const int rc = gen_device(device, sp, 1.0, r0, r1, d_row_ptr, d_col_idx, d_val, nnz);
    (void)hipFree(d_prp); (void)hipFree(d_pci);
    return rc;
}

int sextans_gen_bell_host(int M, int K, int ell_width, uint64_t seed, int **block_col, uint16_t **block_val) {
    if (M <= 0 || K <= 0 || M % 32 || K % 32 || ell_width <= 0 || ell_width > K / 32 || !block_col || !block_val)
        return SEXTANS_ERR_INVALID;
    const int mb = M / 32, kb = K / 32;
    const int64_t nslots = (int64_t)mb * ell_width;
    int *c = (int *)malloc(sizeof(int) * (size_t)nslots);
    uint16_t *v = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)nslots * 1024);
    if (!c || !v) { free(c); free(v); return SEXTANS_ERR_ALLOC; }
    for (int br = 0; br < mb; ++br) {
        int *row = c + (int64_t)br * ell_width;
        for (int i = 0; i < ell_width; ++i) {
            const int x = (int)mulhi64(rnd(seed, (uint64_t)br, 1 + (uint64_t)i), (uint64_t)kb);
            int p = i;
            while (p > 0 && row[p - 1] > x) { row[p] = row[p - 1]; --p; }
            row[p] = x;
        }
        for (int i = 1; i < ell_width; ++i)
            if (row[i] <= row[i - 1]) row[i] = row[i - 1] + 1;
        for (int i = ell_width - 1; i >= 0; --i) {
            const int cap = kb - 1 - (ell_width - 1 - i);
            if (row[i] > cap) row[i] = cap;
        }
    }
    for (int64_t sl = 0; sl < nslots; ++sl)
        for (int e = 0; e < 1024; ++e) v[sl * 1024 + e] = bell_value(seed, sl, e);
    *block_col = c; *block_val = v;
    return SEXTANS_OK;
}

int sextans_gen_bell_device(int device, int M, int K, int ell_width, uint64_t seed, int **d_block_col,
                            uint16_t **d_block_val) {
    if (M <= 0 || K <= 0 || M % 32 || K % 32 || ell_width <= 0 || ell_width > K / 32 || !d_block_col ||
        !d_block_val)
        return SEXTANS_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SEXTANS_ERR_NO_DEVICE;
    SY_HIP(hipSetDevice(device));
    const int mb = M / 32, kb = K / 32;
    const int64_t nslots = (int64_t)mb * ell_width;
    int *c = nullptr;
    uint16_t *v = nullptr;
    SY_HIP(hipMalloc((void **)&c, sizeof(int) * (size_t)nslots));
    SY_HIP(hipMalloc((void **)&v, sizeof(uint16_t) * (size_t)nslots * 1024));
    const Spec sp{0, kb, 0, 0, 0, 0, 0, seed};
    hipLaunchKernelGGL(k_bell_cols, dim3((unsigned)((mb + 255) / 256)), dim3(256), 0, 0, sp, mb, ell_width, c);
    hipLaunchKernelGGL(k_bell_vals, dim3(65536), dim3(256), 0, 0, seed, nslots, v);
    SY_HIP(hipDeviceSynchronize());
    *d_block_col = c; *d_block_val = v;
    return SEXTANS_OK;
}

// Block-banded blocked-ELL: block row br holds the 2*half_width + 1 consecutive block columns centred on its diagonal
// block (window shifted inwards at the matrix edges, so every row is full): neighbouring block rows share all but one of
// their block columns -- the structure (banded / FEM-like at block level) where a B tile staged once per workgroup is
// reused by several block rows.  Values as in the uniform generator (slot-indexed, seeded).
SX_HD int bell_band_start(int br, int mb, int kb, int W) {
    const int64_t centre = (int64_t)br * kb / mb;
    int64_t s0 = centre - W / 2;
    if (s0 < 0) s0 = 0;
    if (s0 > kb - W) s0 = kb - W;
    return (int)s0;
}
__global__ void k_bell_band_cols(int mb, int kb, int W, int *block_col) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)mb * W) return;
    const int br = (int)(t / W), i = (int)(t % W);
    block_col[t] = bell_band_start(br, mb, kb, W) + i;
}

int sextans_gen_bell_banded_host(int M, int K, int half_width, uint64_t seed, int **block_col, uint16_t **block_val) {
    const int W = 2 * half_width + 1;
    if (M <= 0 || K <= 0 || M % 32 || K % 32 || half_width < 0 || W > K / 32 || !block_col || !block_val) return SEXTANS_ERR_INVALID;
    const int mb = M / 32, kb = K / 32;
    const int64_t nslots = (int64_t)mb * W;
    int *c = (int *)malloc(sizeof(int) * (size_t)nslots);
    uint16_t *v = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)nslots * 1024);
    if (!c || !v) { free(c); free(v); return SEXTANS_ERR_ALLOC; }
    for (int br = 0; br < mb; ++br)
        for (int i = 0; i < W; ++i) c[(int64_t)br * W + i] = bell_band_start(br, mb, kb, W) + i;
    for (int64_t sl = 0; sl < nslots; ++sl)
        for (int e = 0; e < 1024; ++e) v[sl * 1024 + e] = bell_value(seed, sl, e);
    *block_col = c; *block_val = v;
    return SEXTANS_OK;
}

int sextans_gen_bell_banded_device(int device, int M, int K, int half_width, uint64_t seed, int **d_block_col,
                                   uint16_t **d_block_val) {
    const int W = 2 * half_width + 1;
    if (M <= 0 || K <= 0 || M % 32 || K % 32 || half_width < 0 || W > K / 32 || !d_block_col || !d_block_val) return SEXTANS_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SEXTANS_ERR_NO_DEVICE;
    SY_HIP(hipSetDevice(device));
    const int mb = M / 32, kb = K / 32;
    const int64_t nslots = (int64_t)mb * W;
    int *c = nullptr;
    uint16_t *v = nullptr;
    SY_HIP(hipMalloc((void **)&c, sizeof(int) * (size_t)nslots));
    SY_HIP(hipMalloc((void **)&v, sizeof(uint16_t) * (size_t)nslots * 1024));
    hipLaunchKernelGGL(k_bell_band_cols, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, 0, mb, kb, W, c);
    hipLaunchKernelGGL(k_bell_vals, dim3(65536), dim3(256), 0, 0, seed, nslots, v);
    SY_HIP(hipDeviceSynchronize());
    *d_block_col = c; *d_block_val = v;
    return SEXTANS_OK;
}

int sextans_gen_uniform_bf16_host(uint16_t *dst, int64_t n, uint64_t seed) {
    if (!dst || n < 0) return SEXTANS_ERR_INVALID;
    for (int64_t i = 0; i < n; ++i) dst[i] = f32_to_bf16_rne(u01m1(rnd(seed, (uint64_t)i, 0x51)));
    return SEXTANS_OK;
}

int sextans_gen_uniform_bf16_device(int device, uint16_t *d_dst, int64_t n, uint64_t seed, void *stream) {
    if (!d_dst || n < 0) return SEXTANS_ERR_INVALID;
    SY_HIP(hipSetDevice(device));
    if (n == 0) return SEXTANS_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_uniform_bf16, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_dst, n, seed);
    SY_HIP(hipGetLastError());
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

__global__ __launch_bounds__(256) void slice_row_ptr(int n, const int *__restrict__ rp, int *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i <= n) out[i] = rp[i] - rp[0];
}

int sextans_csr_slice_rows_device(int device, int r0, int r1, const int *d_row_ptr, int **o_row_ptr, int64_t *first_entry, int64_t *nnz) {
    if (r0 < 0 || r1 < r0 || !d_row_ptr || !o_row_ptr || !first_entry || !nnz) return SEXTANS_ERR_INVALID;
    SY_HIP(hipSetDevice(device));
    int ends[2] = {0, 0};
    SY_HIP(hipMemcpy(&ends[0], d_row_ptr + r0, sizeof(int), hipMemcpyDeviceToHost));
    SY_HIP(hipMemcpy(&ends[1], d_row_ptr + r1, sizeof(int), hipMemcpyDeviceToHost));
    int *out = nullptr;
    SY_HIP(hipMalloc((void **)&out, sizeof(int) * ((size_t)(r1 - r0) + 1)));
    hipLaunchKernelGGL(slice_row_ptr, dim3((unsigned)((r1 - r0 + 256) / 256)), dim3(256), 0, nullptr, r1 - r0, d_row_ptr + r0, out);
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipFree(out); return SEXTANS_ERR_HIP; }
    *o_row_ptr = out;
    *first_entry = ends[0];
    *nnz = (int64_t)ends[1] - ends[0];
    return SEXTANS_OK;
}

int sextans_csr_permute_symmetric_device(int device, int M, int64_t nnz, const int *d_row_ptr, const int *d_col_idx, const float *d_val,
                                         const int *new_of_old, int **o_row_ptr, int **o_col_idx, float **o_val) {
    if (M < 0 || nnz < 0 || !d_row_ptr || !new_of_old || !o_row_ptr || !o_col_idx || !o_val) return SEXTANS_ERR_INVALID;
    SY_HIP(hipSetDevice(device));
    {   // a public entry point: the columns are relabelled through a table of M entries, so they must be rows
        int bad = 0;
        std::string verr;
        if (M > 0 && sx::validate_csr_device(M, M, nnz, d_row_ptr, d_col_idx, &bad, verr)) return SEXTANS_ERR_HIP;
        if (bad) return (bad & 1) ? SEXTANS_ERR_INVALID : SEXTANS_ERR_INDEX;
    }
    std::vector<int> old_of_new((size_t)M, -1), rp((size_t)M + 1);
    for (int i = 0; i < M; ++i) {
        const int p = new_of_old[i];
        if (p < 0 || p >= M || old_of_new[(size_t)p] != -1) return SEXTANS_ERR_INVALID;   // not a permutation
        old_of_new[(size_t)p] = i;
    }
    int *d_order = nullptr, *d_new = nullptr, *d_long = nullptr;
    auto drop = [&]() { (void)hipFree(d_order); (void)hipFree(d_new); (void)hipFree(d_long); };
    std::string err;
    if (hipMalloc((void **)&d_order, sizeof(int) * (size_t)(M ? M : 1)) != hipSuccess || hipMalloc((void **)&d_new, sizeof(int) * (size_t)(M ? M : 1)) != hipSuccess) { drop(); return SEXTANS_ERR_HIP; }
    if (M && (hipMemcpy(d_order, old_of_new.data(), sizeof(int) * (size_t)M, hipMemcpyHostToDevice) != hipSuccess ||
              hipMemcpy(d_new, new_of_old, sizeof(int) * (size_t)M, hipMemcpyHostToDevice) != hipSuccess)) { drop(); return SEXTANS_ERR_HIP; }
    int *nrp = nullptr, *nci = nullptr;
    float *nva = nullptr;
    auto fail = [&](int rc) { drop(); (void)hipFree(nrp); (void)hipFree(nci); (void)hipFree(nva); return rc; };
    if (sx::permute_csr_rows_device(M, nnz, d_row_ptr, d_col_idx, d_val, d_order, &nrp, &nci, &nva, err)) return fail(SEXTANS_ERR_HIP);
    if (sx::relabel_columns_device(nnz, nci, d_new, err)) return fail(SEXTANS_ERR_HIP);
    if (hipMemcpy(rp.data(), nrp, sizeof(int) * ((size_t)M + 1), hipMemcpyDeviceToHost) != hipSuccess) return fail(SEXTANS_ERR_HIP);
    std::vector<int> long_rows;
    for (int r = 0; r < M; ++r) {
        const int n = rp[(size_t)r + 1] - rp[(size_t)r];
        if (n > 4096) return fail(SEXTANS_ERR_INVALID);      // (this tool sorts rows of up to 4096 entries)
        if (n > 256) long_rows.push_back(r);
    }
    if (M) hipLaunchKernelGGL(k_sort_rows_wave, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, 0, M, nrp, nci, nva);
    if (!long_rows.empty()) {
        if (hipMalloc((void **)&d_long, sizeof(int) * long_rows.size()) != hipSuccess ||
            hipMemcpy(d_long, long_rows.data(), sizeof(int) * long_rows.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(SEXTANS_ERR_HIP);
        hipLaunchKernelGGL(k_sort_rows_wg, dim3((unsigned)long_rows.size()), dim3(256), 0, 0, d_long, nrp, nci, nva);
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail(SEXTANS_ERR_HIP);
    drop();
    *o_row_ptr = nrp; *o_col_idx = nci; *o_val = nva;
    return SEXTANS_OK;
}

}  // extern "C"
