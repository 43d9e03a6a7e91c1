// engine_dist.hip -- native multi-GPU form of the SpMM behind the C ABI.
// ------------------------------------------------------------------------------------------------
// Native multi-GPU form (north_star: "A row-range partitioned across the GPUs of one node, B replicated,
// RCCL all-gather of C panels over xGMI") behind the C ABI, for callers that have no torch.distributed.
// RCCL is bound at run time (dlopen "librccl.so.1"): the single-GPU entry points never need it.
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>

#include "engine_state.h"

using namespace sxe;

namespace {
struct Id128 { char b[128]; };   // ncclUniqueId, passed to ncclCommInitRank BY VALUE
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
std::string g_rccl_error;   // written once, inside the call_once below
void rccl_bind(Rccl &r) {
    const char *env = getenv("SEXTANS_RCCL_PATH");
    for (const char *name : {env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        if (!name || !*name) continue;
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) {
        const char *why = dlerror();
        g_rccl_error = std::string("RCCL not found: ") + (why ? why : "dlopen failed");
        return;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
        g_rccl_error = "RCCL library lacks ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclAllGather";
        dlclose(r.lib); r.lib = nullptr;
    }
}
Rccl *rccl() {   // one thread per GPU is the documented model: the binding happens exactly once whoever comes first
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_bind(r); });
    if (!r.lib) { g_last_error = g_rccl_error; return nullptr; }
    return &r;
}
int rccl_check(int rc, const char *what) {
    if (rc == 0) return SEXTANS_OK;
    Rccl *r = rccl();
    g_last_error = std::string(what) + " failed: " + (r && r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
    return SEXTANS_ERR_HIP;
}

// S[g][n][0 .. len_g) -> C[(row0_g + i) + n * ldc]: one thread per staged element; `meta` = {row0, len} per rank.
__global__ __launch_bounds__(256) void dist_unpack_slabs(const float *__restrict__ S, int64_t lmax, int N,
                                                         const int2 *__restrict__ meta, float *C, int64_t ldc) {
    const int g = blockIdx.z, n = blockIdx.y;
    const int2 m = meta[g];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < m.y) C[(int64_t)m.x + i + (int64_t)n * ldc] = S[((int64_t)g * N + n) * lmax + i];
}
}  // namespace

extern "C" {

int sextans_dist_unique_id(char id[128]) {
    if (!id) return SEXTANS_ERR_INVALID;
    Rccl *r = rccl();
    if (!r) return SEXTANS_ERR_STATE;
    return rccl_check(r->GetUniqueId(id), "ncclGetUniqueId");
}

int sextans_dist_comm_init(void **comm, int device, int world, int rank, const char id[128]) {
    if (!comm || !id || world < 1 || rank < 0 || rank >= world) return SEXTANS_ERR_INVALID;
    if (int rc = check_device(device)) return rc;
    Rccl *r = rccl();
    if (!r) return SEXTANS_ERR_STATE;
    SX_HIP(hipSetDevice(device));
    Id128 u;
    memcpy(u.b, id, 128);
    return rccl_check(r->CommInitRank(comm, world, u, rank), "ncclCommInitRank");
}

int sextans_dist_comm_destroy(void *comm) {
    Rccl *r = rccl();
    if (!r || !comm) return SEXTANS_ERR_INVALID;
    return rccl_check(r->CommDestroy(comm), "ncclCommDestroy");
}

int sextans_dist_spmm(sextans_handle_t h, void *comm, int world, int rank, const int *row_ranges, int N, float alpha,
                      const float *d_B, int64_t ldb, float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out,
                      int64_t ldc, int nchunks, void *stream) {
    if (!h || !comm || world < 1 || rank < 0 || rank >= world || !row_ranges || N <= 0 || (N % 8) || !d_B || !d_C_in ||
        !d_C_out)
        return SEXTANS_ERR_INVALID;
    if (!h->d_rp) return SEXTANS_ERR_STATE;
    Rccl *r = rccl();
    if (!r) return SEXTANS_ERR_STATE;
    // ranges must tile [0, M_total) in rank order and this rank's range must be the engine's matrix
    int64_t M_total = 0;
    for (int g = 0; g < world; ++g) {
        if (row_ranges[2 * g] != (int)M_total || row_ranges[2 * g + 1] < row_ranges[2 * g]) return SEXTANS_ERR_INVALID;
        M_total = row_ranges[2 * g + 1];
    }
    const int row0 = row_ranges[2 * rank], m_loc = row_ranges[2 * rank + 1] - row0;
    if (m_loc != h->M || ldc < M_total || ldc_in < M_total || ldb < h->K) return SEXTANS_ERR_INVALID;
    SX_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    // (where this rank's rows sit in the matrix: lets the graph clustering run on the slab -- used by whole-slab calls, nchunks = 1)
    if (h->opt_row_offset != row0) (void)sextans_set_option(h, "row_offset", row0);
    if (nchunks < 1) nchunks = 1;
    if (nchunks > 16) nchunks = 16;
    // Chunk c of rank g = local rows [cuts[g][c], cuts[g][c+1]).  Every rank snaps its OWN interior cuts to the
    // boundaries its kernels want (sextans_align_row: row blocks of the LDS-panel plan, wavefronts of the window kernel,
    // so every chunk keeps the whole-matrix kernel) and the cut positions are exchanged once per (partition, N, chunk
    // count) with a small ncclAllGather; they are cached in the engine afterwards.
    std::vector<int> key(row_ranges, row_ranges + 2 * world);
    key.push_back(N); key.push_back(nchunks); key.push_back(rank);
    if (h->dist_cut_key != key) {
        {   // Non-zeros of the whole matrix = sum over ranks: the automatic hub-split threshold ("split_rows" = -1) is
            // derived from it, so a rank cuts a hub row into the same pieces as one GPU holding every row would and the
            // N-GPU result equals the 1-GPU result bit for bit (a row lives on exactly one rank).
            int *d_nz = nullptr;
            SX_HIP(hipMalloc((void **)&d_nz, sizeof(int) * 2 * (size_t)world));
            const int mine_nz[2] = {(int)(h->nnz & 0x7fffffff), (int)(h->nnz >> 31)};
            SX_HIP(hipMemcpyAsync(d_nz + 2 * (size_t)rank, mine_nz, sizeof mine_nz, hipMemcpyHostToDevice, s));
            const int rc = rccl_check(r->AllGather(d_nz + 2 * (size_t)rank, d_nz, 2, 2 /* ncclInt32 */, comm, s), "ncclAllGather(nnz)");
            std::vector<int> all_nz(2 * (size_t)world);
            hipError_t e1 = rc ? hipSuccess : hipMemcpyAsync(all_nz.data(), d_nz, sizeof(int) * all_nz.size(), hipMemcpyDeviceToHost, s);
            hipError_t e2 = hipStreamSynchronize(s);
            (void)hipFree(d_nz);
            if (rc) return rc;
            SX_HIP(e1);
            SX_HIP(e2);
            int64_t total = 0;
            for (int g = 0; g < world; ++g) total += (int64_t)all_nz[2 * (size_t)g] + ((int64_t)all_nz[2 * (size_t)g + 1] << 31);
            h->opt_global_nnz = total;
        }
        std::vector<int> mine((size_t)nchunks + 1, 0);
        mine[(size_t)nchunks] = m_loc;
        for (int c = 1; c < nchunks; ++c) {
            int a = (int)((int64_t)m_loc * c / nchunks);
            if (int rc = sextans_align_row(h, N, a, &a)) return rc;
            mine[(size_t)c] = std::min(std::max(a, mine[(size_t)c - 1]), m_loc);
        }
        int *d_cuts = nullptr;
        SX_HIP(hipMalloc((void **)&d_cuts, sizeof(int) * (size_t)world * ((size_t)nchunks + 1)));
        SX_HIP(hipMemcpyAsync(d_cuts + (size_t)rank * (nchunks + 1), mine.data(), sizeof(int) * mine.size(),
                              hipMemcpyHostToDevice, s));
        const int rc = rccl_check(r->AllGather(d_cuts + (size_t)rank * (nchunks + 1), d_cuts, (size_t)nchunks + 1, 2 /* ncclInt32 */,
                                               comm, s), "ncclAllGather(cuts)");
        std::vector<int> all((size_t)world * ((size_t)nchunks + 1));
        hipError_t e1 = rc ? hipSuccess : hipMemcpyAsync(all.data(), d_cuts, sizeof(int) * all.size(), hipMemcpyDeviceToHost, s);
        hipError_t e2 = hipStreamSynchronize(s);
        (void)hipFree(d_cuts);
        if (rc) return rc;
        SX_HIP(e1);
        SX_HIP(e2);
        for (int g = 0; g < world; ++g) {   // what arrived must be a monotone cut list of that rank's range
            const int len = row_ranges[2 * g + 1] - row_ranges[2 * g];
            const int *cg = all.data() + (size_t)g * (nchunks + 1);
            if (cg[0] != 0 || cg[nchunks] != len) return SEXTANS_ERR_STATE;
            for (int c = 0; c < nchunks; ++c)
                if (cg[c + 1] < cg[c]) return SEXTANS_ERR_STATE;
        }
        h->dist_cuts = all;
        h->dist_cut_key = key;
    }
    auto cut = [&](int g, int c) { return h->dist_cuts[(size_t)g * (nchunks + 1) + (size_t)c]; };
    std::vector<int64_t> lmax((size_t)nchunks, 1), off((size_t)nchunks + 1, 0);
    for (int c = 0; c < nchunks; ++c) {
        for (int g = 0; g < world; ++g) lmax[(size_t)c] = std::max<int64_t>(lmax[(size_t)c], cut(g, c + 1) - cut(g, c));
        off[(size_t)c + 1] = off[(size_t)c] + (int64_t)world * N * lmax[(size_t)c];
    }
    // staging + per-chunk {row0, len} tables (ints, kept behind the float staging area)
    const size_t meta_floats = (size_t)nchunks * (size_t)world * 2;
    if (h->stage_cap < (size_t)off[(size_t)nchunks] + meta_floats) h->dist_meta_at = nullptr;   // new buffer: tables gone
    if (int rc = ensure(&h->d_stage, &h->stage_cap, (size_t)off[(size_t)nchunks] + meta_floats)) return rc;
    if (!h->comm_stream) SX_HIP(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    while (h->dist_events.size() < (size_t)nchunks + 1) {
        hipEvent_t e;
        SX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->dist_events.push_back(e);
    }
    std::vector<int> meta(meta_floats);
    for (int c = 0; c < nchunks; ++c)
        for (int g = 0; g < world; ++g) {
            meta[((size_t)c * world + g) * 2] = row_ranges[2 * g] + cut(g, c);
            meta[((size_t)c * world + g) * 2 + 1] = cut(g, c + 1) - cut(g, c);
        }
    int *d_meta = reinterpret_cast<int *>(h->d_stage + off[(size_t)nchunks]);
    if (h->dist_meta != meta || h->dist_meta_at != d_meta) {   // the row tables change only with the partition
        SX_HIP(hipMemcpyAsync(d_meta, meta.data(), sizeof(int) * meta.size(), hipMemcpyHostToDevice, s));
        SX_HIP(hipStreamSynchronize(s));   // `meta` is a host temporary; later calls with the same ranges skip this
        h->dist_meta = meta;
        h->dist_meta_at = d_meta;
    }
    bool first = true;
    for (int c = 0; c < nchunks; ++c) {
        float *S = h->d_stage + off[(size_t)c];
        const int c0 = cut(rank, c), c1 = cut(rank, c + 1);
        float *mine = S + (size_t)rank * N * lmax[(size_t)c];
        if (c1 > c0) {
            if (int rc = sextans_spmm_device_rows(h, N, alpha, d_B, ldb, beta, d_C_in + row0 + c0, ldc_in, mine,
                                                  lmax[(size_t)c], c0, c1, first ? 0 : SEXTANS_ROWS_REUSE_B_PANELS, stream))
                return rc;
            first = false;
        }
        // the all-gather of chunk c runs on the communication stream while the SpMM of chunk c+1 runs on `stream`; its
        // slabs are unpacked into column-major C right behind it on the same stream, i.e. under all-gather c+1 / SpMM c+2,
        // so only the last chunk's unpack is exposed
        SX_HIP(hipEventRecord(h->dist_events[(size_t)c], s));
        SX_HIP(hipStreamWaitEvent(h->comm_stream, h->dist_events[(size_t)c], 0));
        if (int rc = rccl_check(r->AllGather(mine, S, (size_t)N * (size_t)lmax[(size_t)c], 7 /* ncclFloat */, comm,
                                             h->comm_stream), "ncclAllGather"))
            return rc;
        const unsigned gx = (unsigned)((lmax[(size_t)c] + 255) / 256);
        hipLaunchKernelGGL(dist_unpack_slabs, dim3(gx, (unsigned)N, (unsigned)world), dim3(256), 0, h->comm_stream, S,
                           lmax[(size_t)c], N, reinterpret_cast<const int2 *>(d_meta) + (size_t)c * world, d_C_out, ldc);
    }
    SX_HIP(hipEventRecord(h->dist_events[(size_t)nchunks], h->comm_stream));
    SX_HIP(hipStreamWaitEvent(s, h->dist_events[(size_t)nchunks], 0));
    SX_HIP(hipGetLastError());
    return SEXTANS_OK;
}

}  // extern "C"
