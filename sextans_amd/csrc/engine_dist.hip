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
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;   // (optional: unequal row-major slabs)
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
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
    r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
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
    // comm == NULL is allowed for world == 1: the rank's chunks are computed, staged and unpacked exactly as in a multi-rank run, only
    // the collectives are skipped (a 1-rank all-gather is a no-op): single-GPU callers without RCCL, and tools/rank_slabs.py
    if (!h || (!comm && world != 1) || world < 1 || rank < 0 || rank >= world || !row_ranges || N <= 0 || (N % 8) || !d_B || !d_C_in ||
        !d_C_out)
        return SEXTANS_ERR_INVALID;
    if (!h->d_rp) return SEXTANS_ERR_STATE;
    Rccl *r = comm ? rccl() : nullptr;
    if (comm && !r) return SEXTANS_ERR_STATE;
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
    // (set once per partition: a change frees the clustered plan, and the first call after it rebuilds it -- ~0.3 s for 318 M non-zeros,
    // inside that call: issue one warm-up call before timing, as bench.py does.  A 1-rank "world" keeps what the caller set: its slab
    // may be a range of a larger matrix -- tools/rank_slabs.py.)
    if (world > 1 && h->opt_row_offset != row0)
        if (int rc = sextans_set_option(h, "row_offset", row0)) return rc;
    if (nchunks < 1) nchunks = 1;
    if (nchunks > 16) nchunks = 16;
    // Clustered-order chunks (round 5): when this rank's slab runs on a graph-clustered plan, chunks are ranges of the plan's row
    // BLOCKS and every chunk keeps the reordered form (engine.hip: cc_*) -- if every rank of the partition can do the same.
    // (before anything is planned: the long-row thresholds follow the whole matrix's non-zeros, and a change rebuilds every packed form)
    std::vector<int> nnz_key(row_ranges, row_ranges + 2 * world);
    nnz_key.push_back(rank);
    if (comm && h->dist_nnz_key != nnz_key) {   // Non-zeros of the whole matrix = sum over ranks: the automatic hub-split threshold ("split_rows" = -1) is
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
        if (total != h->opt_global_nnz)
            if (int rc = sextans_set_option(h, "global_nnz", total)) return rc;
        h->dist_nnz_key = nnz_key;
    }
    bool want_cc = false;
    if (nchunks > 1)
        if (int rc = cc_prepare(h, N, &want_cc)) return rc;
    if (want_cc && h->psc.plan_nblk < nchunks) want_cc = false;
    // Chunk c of rank g = local rows [cuts[g][c], cuts[g][c+1]).  Every rank snaps its OWN interior cuts to the
    // boundaries its kernels want (sextans_align_row: row blocks of the LDS-panel plan, wavefronts of the window kernel,
    // so every chunk keeps the whole-matrix kernel) and the cut positions are exchanged once per (partition, N, chunk
    // count) with a small ncclAllGather; they are cached in the engine afterwards.
    std::vector<int> key(row_ranges, row_ranges + 2 * world);
    key.push_back(N); key.push_back(nchunks); key.push_back(rank); key.push_back(want_cc ? 1 : 0);
    int m_max = 0;
    for (int g = 0; g < world; ++g) m_max = std::max(m_max, row_ranges[2 * g + 1] - row_ranges[2 * g]);
    if (h->dist_cut_key != key) {
        bool all_cc = want_cc;
        if (comm) {   // does every rank want clustered-order chunks?  (one int per rank)
            int *d_f = nullptr;
            SX_HIP(hipMalloc((void **)&d_f, sizeof(int) * (size_t)world));
            const int mine_f = want_cc ? 1 : 0;
            SX_HIP(hipMemcpyAsync(d_f + rank, &mine_f, sizeof(int), hipMemcpyHostToDevice, s));
            const int rc = rccl_check(r->AllGather(d_f + rank, d_f, 1, 2 /* ncclInt32 */, comm, s), "ncclAllGather(mode)");
            std::vector<int> f((size_t)world, 0);
            hipError_t e1 = rc ? hipSuccess : hipMemcpyAsync(f.data(), d_f, sizeof(int) * (size_t)world, hipMemcpyDeviceToHost, s);
            hipError_t e2 = hipStreamSynchronize(s);
            (void)hipFree(d_f);
            if (rc) return rc;
            SX_HIP(e1);
            SX_HIP(e2);
            for (int g = 0; g < world; ++g) all_cc = all_cc && f[(size_t)g] != 0;
        }
        h->dist_cc = all_cc;
        if (all_cc) {   // every rank's position -> global row table, exchanged once per partition: [world][m_max] ints
            const size_t need = (size_t)world * (size_t)m_max;
            if (h->dist_rows_cap < need) {
                (void)hipFree(h->d_dist_rows);
                h->d_dist_rows = nullptr; h->dist_rows_cap = 0;
                SX_HIP(hipMalloc((void **)&h->d_dist_rows, sizeof(int) * std::max<size_t>(need, 1)));
                h->dist_rows_cap = need;
            }
            cc_table(h, row0, h->d_dist_rows + (size_t)rank * m_max, s);
            if (comm)
                if (int rc = rccl_check(r->AllGather(h->d_dist_rows + (size_t)rank * m_max, h->d_dist_rows, (size_t)m_max, 2 /* ncclInt32 */, comm, s),
                                        "ncclAllGather(row tables)"))
                    return rc;
            SX_HIP(hipStreamSynchronize(s));
        }
        std::vector<int> mine((size_t)nchunks + 1, 0);
        mine[(size_t)nchunks] = m_loc;
        for (int c = 1; c < nchunks; ++c) {
            if (all_cc) {   // positions of the clustered order at block boundaries
                mine[(size_t)c] = h->psc.h_blk_row[(size_t)((int64_t)h->psc.plan_nblk * c / nchunks)];
                continue;
            }
            int a = (int)((int64_t)m_loc * c / nchunks);
            if (int rc = sextans_align_row(h, N, a, &a)) return rc;
            mine[(size_t)c] = std::min(std::max(a, mine[(size_t)c - 1]), m_loc);
        }
        std::vector<int> all((size_t)world * ((size_t)nchunks + 1));
        if (!comm) {
            all = mine;
        } else {
        int *d_cuts = nullptr;
        SX_HIP(hipMalloc((void **)&d_cuts, sizeof(int) * (size_t)world * ((size_t)nchunks + 1)));
        SX_HIP(hipMemcpyAsync(d_cuts + (size_t)rank * (nchunks + 1), mine.data(), sizeof(int) * mine.size(),
                              hipMemcpyHostToDevice, s));
        const int rc = rccl_check(r->AllGather(d_cuts + (size_t)rank * (nchunks + 1), d_cuts, (size_t)nchunks + 1, 2 /* ncclInt32 */,
                                               comm, s), "ncclAllGather(cuts)");
        hipError_t e1 = rc ? hipSuccess : hipMemcpyAsync(all.data(), d_cuts, sizeof(int) * all.size(), hipMemcpyDeviceToHost, s);
        hipError_t e2 = hipStreamSynchronize(s);
        (void)hipFree(d_cuts);
        if (rc) return rc;
        SX_HIP(e1);
        SX_HIP(e2);
        }
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
    const bool cc = h->dist_cc;
    if (cc)   // row-major staging of the whole C: received slabs are scattered into it, one streaming pass writes column-major C at the end
        if (int rc = ensure(&h->d_Cfull, &h->Cfull_cap, (size_t)M_total * (size_t)N)) return rc;
    bool first = true;
    for (int c = 0; c < nchunks; ++c) {
        float *S = h->d_stage + off[(size_t)c];
        const int c0 = cut(rank, c), c1 = cut(rank, c + 1);
        float *mine = S + (size_t)rank * N * lmax[(size_t)c];
        if (cc) {
            if (first) cc_pre(h, N, d_B, ldb, d_C_in + row0, ldc_in, s);
            first = false;
            const int b0 = (int)((int64_t)h->psc.plan_nblk * c / nchunks), b1 = c + 1 == nchunks ? h->psc.plan_nblk : (int)((int64_t)h->psc.plan_nblk * (c + 1) / nchunks);
            if (int rc = cc_chunk(h, N, alpha, beta, b0, b1, h->d_dist_rows + (size_t)rank * m_max, row0, mine, lmax[(size_t)c], s)) return rc;
        } else if (c1 > c0) {
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
        if (comm)
            if (int rc = rccl_check(r->AllGather(mine, S, (size_t)N * (size_t)lmax[(size_t)c], 7 /* ncclFloat */, comm,
                                                 h->comm_stream), "ncclAllGather"))
                return rc;
        if (cc) {   // slabs hold rows in the senders' clustered order: 64-byte rows to their places in the staging buffer of the whole C
            for (int g = 0; g < world; ++g)
                cc_scatter(S + (size_t)g * N * lmax[(size_t)c], lmax[(size_t)c], h->d_dist_rows + (size_t)g * m_max + cut(g, c), cut(g, c + 1) - cut(g, c),
                           h->d_Cfull, M_total * 16, N, h->comm_stream);
            continue;
        }
        const unsigned gx = (unsigned)((lmax[(size_t)c] + 255) / 256);
        hipLaunchKernelGGL(dist_unpack_slabs, dim3(gx, (unsigned)N, (unsigned)world), dim3(256), 0, h->comm_stream, S,
                           lmax[(size_t)c], N, reinterpret_cast<const int2 *>(d_meta) + (size_t)c * world, d_C_out, ldc);
    }
    if (cc) cc_finish(h->d_Cfull, d_C_out, ldc, (int)M_total, N, h->comm_stream);
    SX_HIP(hipEventRecord(h->dist_events[(size_t)nchunks], h->comm_stream));
    SX_HIP(hipStreamWaitEvent(s, h->dist_events[(size_t)nchunks], 0));
    SX_HIP(hipGetLastError());
    return SEXTANS_OK;
}

// Row-major form (round 5).  The rows [row0, row1) of a row-major M_total x N matrix with ldc == N ARE one contiguous run: the rank's
// SpMM (sextans_spmm_device_rm: no repack of the replicated B, no staging, natural / brick / graph-clustered plans alike -- every
// kernel writes a row where it belongs) lands in its place inside d_C_out and the exchange is an IN-PLACE all-gather -- one
// ncclAllGather when the ranges have equal lengths, one group of ncclBroadcast (root g sends its run) for nnz-balanced ranges.
// Nothing is packed, nothing unpacked: what sextans_dist_spmm spends per rank on the B repack (O(K N) on EVERY rank, it does not
// shrink with the world size: 17 of 85 us per rank at 8 ranks on the 4M-row FEM matrix, 125 of 279 us on its randomly numbered form,
// profiles/r05_rank_slab_times.json) and on the unpack pass over all of C is gone.  ldc > N (rows not adjacent): the runs travel
// through a packed staging copy (two strided copies around the same collective).
int sextans_dist_spmm_rm(sextans_handle_t h, void *comm, int world, int rank, const int *row_ranges, int N, float alpha, const float *d_B,
                         int64_t ldb, float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out, int64_t ldc, void *stream) {
    if (!h || (!comm && world != 1) || world < 1 || rank < 0 || rank >= world || !row_ranges || N <= 0 || (N % 8) || !d_B || !d_C_in || !d_C_out ||
        ldb < N || ldc_in < N || ldc < N)
        return SEXTANS_ERR_INVALID;
    if (!h->d_rp) return SEXTANS_ERR_STATE;
    Rccl *r = comm ? rccl() : nullptr;
    if (comm && !r) return SEXTANS_ERR_STATE;
    int64_t M_total = 0;
    bool equal = true;
    for (int g = 0; g < world; ++g) {
        if (row_ranges[2 * g] != (int)M_total || row_ranges[2 * g + 1] < row_ranges[2 * g]) return SEXTANS_ERR_INVALID;
        M_total = row_ranges[2 * g + 1];
        equal = equal && row_ranges[2 * g + 1] - row_ranges[2 * g] == row_ranges[1] - row_ranges[0];
    }
    const int row0 = row_ranges[2 * rank], m_loc = row_ranges[2 * rank + 1] - row0;
    if (m_loc != h->M) return SEXTANS_ERR_INVALID;
    if (getenv("SEXTANS_DIST_BROADCAST_RUNS")) equal = false;   // (tests: the grouped-broadcast exchange on ranges of equal length too)
    if (comm && !equal && (!r->Broadcast || !r->GroupStart || !r->GroupEnd)) {
        g_last_error = "RCCL library lacks ncclBroadcast / ncclGroupStart / ncclGroupEnd (row ranges of unequal length)";
        return SEXTANS_ERR_STATE;
    }
    SX_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (world > 1 && h->opt_row_offset != row0)
        if (int rc = sextans_set_option(h, "row_offset", row0)) return rc;
    std::vector<int> nnz_key(row_ranges, row_ranges + 2 * world);
    nnz_key.push_back(rank);
    if (comm && h->dist_nnz_key != nnz_key) {   // the long-row thresholds follow the whole matrix's non-zeros (as in sextans_dist_spmm)
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
        if (total != h->opt_global_nnz)
            if (int rc = sextans_set_option(h, "global_nnz", total)) return rc;
        h->dist_nnz_key = nnz_key;
    }
    if (m_loc > 0)
        if (int rc = sextans_spmm_device_rm(h, N, alpha, d_B, ldb, beta, d_C_in + (int64_t)row0 * ldc_in, ldc_in, d_C_out + (int64_t)row0 * ldc, ldc, stream))
            return rc;
    if (!comm) return SEXTANS_OK;
    float *X = d_C_out;   // where the runs are exchanged: C_out itself, or a packed copy of it
    const bool packed = ldc != N;
    if (packed) {
        if (int rc = ensure(&h->d_stage, &h->stage_cap, (size_t)M_total * (size_t)N)) return rc;
        h->dist_meta_at = nullptr;   // (the column-major form keeps its row tables behind its staging area)
        X = h->d_stage;
        if (m_loc > 0)
            SX_HIP(hipMemcpy2DAsync(X + (int64_t)row0 * N, sizeof(float) * (size_t)N, d_C_out + (int64_t)row0 * ldc, sizeof(float) * (size_t)ldc, sizeof(float) * (size_t)N,
                                    (size_t)m_loc, hipMemcpyDeviceToDevice, s));
    }
    if (equal) {
        const size_t count = (size_t)m_loc * (size_t)N;
        if (int rc = rccl_check(r->AllGather(X + (size_t)rank * count, X, count, 7 /* ncclFloat */, comm, s), "ncclAllGather(row-major C)")) return rc;
    } else {
        if (int rc = rccl_check(r->GroupStart(), "ncclGroupStart")) return rc;
        int rc = SEXTANS_OK;
        for (int g = 0; g < world && rc == SEXTANS_OK; ++g) {
            const size_t count = (size_t)(row_ranges[2 * g + 1] - row_ranges[2 * g]) * (size_t)N;
            float *run = X + (int64_t)row_ranges[2 * g] * N;
            if (count) rc = rccl_check(r->Broadcast(run, run, count, 7 /* ncclFloat */, g, comm, s), "ncclBroadcast(row-major C)");
        }
        const int rc2 = rccl_check(r->GroupEnd(), "ncclGroupEnd");
        if (rc) return rc;
        if (rc2) return rc2;
    }
    if (packed)   // the other ranks' rows: two strided copies around this rank's own run
        for (int part = 0; part < 2; ++part) {
            const int64_t a = part ? row0 + m_loc : 0, b = part ? M_total : row0;
            if (b > a)
                SX_HIP(hipMemcpy2DAsync(d_C_out + a * ldc, sizeof(float) * (size_t)ldc, X + a * N, sizeof(float) * (size_t)N, sizeof(float) * (size_t)N, (size_t)(b - a),
                                        hipMemcpyDeviceToDevice, s));
        }
    SX_HIP(hipGetLastError());
    return SEXTANS_OK;
}

// Blocked-ELL bf16 (BASELINE config 5) over several GPUs -- SURVEY 8e: "Config 5 likewise (block-row ranges)".  Block rows are
// independent (every wavefront owns 32 rows of C), so the engine of a rank holds the block rows of its row range
// (sextans_set_matrix_bell[_device] with M = its rows; ranges are multiples of 32), B (bf16) is replicated, and the rank's fp32 slab is
// written PACKED (ldc = longest range) into its slot of the staging buffer, moved by ONE ncclAllGather and written into column-major
// C_out by one HBM-local pass -- the column-major CSR form without chunks: at N = 256 the slab of a rank is 128 MB at 8 ranks, the
// kernel 27 ms / world.  Every element is computed by the same wavefront code on the same operands as on one GPU: bit-identical to
// sextans_spmm_bell_device on the whole matrix.  comm == NULL with world == 1: the same without the collective.
int sextans_dist_spmm_bell(sextans_handle_t h, void *comm, int world, int rank, const int *row_ranges, int N, float alpha, const uint16_t *d_B,
                           int64_t ldb, float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out, int64_t ldc, void *stream) {
    if (!h || (!comm && world != 1) || world < 1 || rank < 0 || rank >= world || !row_ranges || N <= 0 || (N % 32) || !d_B || !d_C_in || !d_C_out)
        return SEXTANS_ERR_INVALID;
    if (!h->d_bell_Af) return SEXTANS_ERR_STATE;
    Rccl *r = comm ? rccl() : nullptr;
    if (comm && !r) return SEXTANS_ERR_STATE;
    int64_t M_total = 0, lmax = 1;
    for (int g = 0; g < world; ++g) {
        if (row_ranges[2 * g] != (int)M_total || row_ranges[2 * g + 1] < row_ranges[2 * g] || (row_ranges[2 * g + 1] % 32)) return SEXTANS_ERR_INVALID;
        lmax = std::max<int64_t>(lmax, row_ranges[2 * g + 1] - row_ranges[2 * g]);
        M_total = row_ranges[2 * g + 1];
    }
    const int row0 = row_ranges[2 * rank], m_loc = row_ranges[2 * rank + 1] - row0;
    if (m_loc != h->bell_M || ldc < M_total || ldc_in < M_total) return SEXTANS_ERR_INVALID;
    SX_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t slab = (size_t)N * (size_t)lmax, meta_floats = (size_t)world * 2;
    if (h->stage_cap < (size_t)world * slab + meta_floats) h->dist_meta_at = nullptr;
    if (int rc = ensure(&h->d_stage, &h->stage_cap, (size_t)world * slab + meta_floats)) return rc;
    std::vector<int> meta(meta_floats);
    for (int g = 0; g < world; ++g) { meta[2 * (size_t)g] = row_ranges[2 * g]; meta[2 * (size_t)g + 1] = row_ranges[2 * g + 1] - row_ranges[2 * g]; }
    int *d_meta = reinterpret_cast<int *>(h->d_stage + (size_t)world * slab);
    if (h->dist_meta != meta || h->dist_meta_at != d_meta) {   // the row table changes only with the partition
        SX_HIP(hipMemcpyAsync(d_meta, meta.data(), sizeof(int) * meta.size(), hipMemcpyHostToDevice, s));
        SX_HIP(hipStreamSynchronize(s));
        h->dist_meta = meta;
        h->dist_meta_at = d_meta;
    }
    float *mine = h->d_stage + (size_t)rank * slab;
    if (int rc = sextans_spmm_bell_device2(h, N, alpha, d_B, ldb, beta, d_C_in + row0, ldc_in, mine, lmax, stream)) return rc;
    if (comm)
        if (int rc = rccl_check(r->AllGather(mine, h->d_stage, slab, 7 /* ncclFloat */, comm, s), "ncclAllGather(blocked-ELL C)")) return rc;
    hipLaunchKernelGGL(dist_unpack_slabs, dim3((unsigned)((lmax + 255) / 256), (unsigned)N, (unsigned)world), dim3(256), 0, s, h->d_stage, lmax, N,
                       reinterpret_cast<const int2 *>(d_meta), d_C_out, ldc);
    SX_HIP(hipGetLastError());
    return SEXTANS_OK;
}

}  // extern "C"
