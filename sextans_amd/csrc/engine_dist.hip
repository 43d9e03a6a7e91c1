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
// One binding per process, made at the first use or by sextans_dist_bind_library; replaced only while no communicator is alive.
std::mutex g_bind_mutex;
Rccl g_r;
bool g_bound = false;
int g_live_comms = 0;
std::string g_bind_path, g_rccl_error;
void rccl_bind_locked() {
    Rccl r;
    g_rccl_error.clear();
    const char *env = getenv("SEXTANS_RCCL_PATH");
    const char *explicit_path = g_bind_path.empty() ? nullptr : g_bind_path.c_str();
    for (const char *name : {explicit_path, explicit_path ? nullptr : env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        if (!name || !*name) continue;
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib || explicit_path) break;   // (an explicitly bound library that does not load is an error, not a reason to fall back)
    }
    if (!r.lib) {
        const char *why = dlerror();
        g_rccl_error = std::string("RCCL not found: ") + (why ? why : "dlopen failed");
        g_r = Rccl();
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
        dlclose(r.lib);
        r = Rccl();
    }
    g_r = r;
}
Rccl *rccl() {   // one thread per GPU is the documented model: the binding happens exactly once whoever comes first
    std::lock_guard<std::mutex> lk(g_bind_mutex);
    if (!g_bound) { rccl_bind_locked(); g_bound = true; }
    if (!g_r.lib) { g_last_error = g_rccl_error; return nullptr; }
    return &g_r;
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

// ---- control collectives ------------------------------------------------------------------------------------------------------------
// Every control exchange carries the sender's STATUS next to its payload: `n` ints per rank + one.  It returns SEXTANS_OK only when
// every rank's status was 0 (`all` then holds world * n ints); otherwise this rank's own code if it had failed, SEXTANS_ERR_PEER if
// only others had -- on every rank at the same point of the sequence, so that no rank walks into the next collective alone.
int exchange(sextans_engine *h, Rccl *r, void *comm, int world, int rank, const int *mine, int n, std::vector<int> &all, int status, hipStream_t s,
             const char *what) {
    ++h->dist_exchanges;
    const size_t per = (size_t)n + 1;
    std::vector<int> me(per, 0), buf(per * (size_t)world, 0);
    for (int i = 0; i < n; ++i) me[(size_t)i] = status ? 0 : mine[i];
    me[(size_t)n] = status;
    int *d = nullptr;
    SX_HIP(hipMalloc((void **)&d, sizeof(int) * buf.size()));
    hipError_t e0 = hipMemcpyAsync(d + per * (size_t)rank, me.data(), sizeof(int) * per, hipMemcpyHostToDevice, s);
    const int rc = e0 != hipSuccess ? SEXTANS_OK : rccl_check(r->AllGather(d + per * (size_t)rank, d, per, 2 /* ncclInt32 */, comm, s), what);
    hipError_t e1 = (rc || e0 != hipSuccess) ? hipSuccess : hipMemcpyAsync(buf.data(), d, sizeof(int) * buf.size(), hipMemcpyDeviceToHost, s);
    hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (rc) return rc;
    SX_HIP(e0);
    SX_HIP(e1);
    SX_HIP(e2);
    bool peer_failed = false;
    all.assign((size_t)n * (size_t)world, 0);
    for (int g = 0; g < world; ++g) {
        peer_failed = peer_failed || buf[per * (size_t)g + (size_t)n] != 0;
        for (int i = 0; i < n; ++i) all[(size_t)g * n + (size_t)i] = buf[per * (size_t)g + (size_t)i];
    }
    if (status) return status;
    if (peer_failed) {
        g_last_error = std::string(what) + ": another rank reported a failure";
        return SEXTANS_ERR_PEER;
    }
    return SEXTANS_OK;
}

struct Partition {   // what the row ranges say about this rank
    int64_t M_total = 0;
    int row0 = 0, m_loc = 0, m_max = 0;
    bool equal = true;
};
int read_partition(int world, int rank, const int *row_ranges, int multiple_of, Partition &p) {
    p = Partition();
    for (int g = 0; g < world; ++g) {
        if (row_ranges[2 * g] != (int)p.M_total || row_ranges[2 * g + 1] < row_ranges[2 * g] || (row_ranges[2 * g + 1] % multiple_of)) return SEXTANS_ERR_INVALID;
        p.M_total = row_ranges[2 * g + 1];
        p.m_max = std::max(p.m_max, row_ranges[2 * g + 1] - row_ranges[2 * g]);
        p.equal = p.equal && row_ranges[2 * g + 1] - row_ranges[2 * g] == row_ranges[1] - row_ranges[0];
    }
    p.row0 = row_ranges[2 * rank];
    p.m_loc = row_ranges[2 * rank + 1] - p.row0;
    return SEXTANS_OK;
}

// Non-zeros of the whole matrix = sum over ranks: the automatic hub-split threshold ("split_rows" = -1) is derived from it, so a rank
// cuts a hub row into the same pieces as one GPU holding every row would and the N-GPU result equals the 1-GPU result bit for bit (a
// row lives on exactly one rank).  Exchanged once per (ranges, rank); before anything is planned: a change rebuilds every packed form.
int exchange_nnz(sextans_engine *h, Rccl *r, void *comm, int world, int rank, const int *row_ranges, bool force, int status, hipStream_t s, bool *exchanged) {
    if (!comm) return status;
    std::vector<int> nnz_key(row_ranges, row_ranges + 2 * world);
    nnz_key.push_back(rank);
    if (!force && h->dist_nnz_key == nnz_key) return status;
    const int mine_nz[2] = {(int)(h->nnz & 0x7fffffff), (int)(h->nnz >> 31)};
    std::vector<int> all_nz;
    *exchanged = true;
    if (int rc = exchange(h, r, comm, world, rank, mine_nz, 2, all_nz, status, s, "ncclAllGather(nnz)")) return rc;
    int64_t total = 0;
    for (int g = 0; g < world; ++g) total += (int64_t)all_nz[2 * (size_t)g] + ((int64_t)all_nz[2 * (size_t)g + 1] << 31);
    if (total != h->opt_global_nnz)
        if (int rc = sextans_set_option(h, "global_nnz", total)) return rc;
    h->dist_nnz_key = nnz_key;
    return SEXTANS_OK;
}

// The last step of every preparation that exchanged anything: all ranks learn whether all ranks are ready.
int settle(sextans_engine *h, Rccl *r, void *comm, int world, int rank, int status, bool exchanged, hipStream_t s) {
    if (!comm || !exchanged) return status;
    std::vector<int> none;
    const int rc = exchange(h, r, comm, world, rank, nullptr, 0, none, status, s, "ncclAllGather(status)");
    if (rc) { h->dist_cut_key.clear(); h->dist_nnz_key.clear(); }   // nothing of a failed preparation is trusted later
    return rc;
}

// ---- column-major CSR form: everything sextans_dist_spmm needs before its first launch ------------------------------------------------
struct CmSetup {
    Partition p;
    std::vector<int64_t> lmax, off;
    int *d_meta = nullptr;
    bool cc = false;
};
int setup_cm(sextans_engine *h, Rccl *r, void *comm, int world, int rank, const int *row_ranges, int N, int &nchunks, hipStream_t s, bool force, CmSetup &out) {
    bool exchanged = false;
    int st = read_partition(world, rank, row_ranges, 1, out.p);
    const Partition &p = out.p;
    if (!st && p.m_loc != h->M) st = SEXTANS_ERR_INVALID;
    if (!st && !h->d_rp) st = SEXTANS_ERR_STATE;
    if (st && !(force && comm)) return st;   // (a collective preparation carries it through the first exchange instead: every rank must learn of it)
    // (where this rank's rows sit in the matrix: lets the graph clustering run on the slab -- used by whole-slab calls, nchunks = 1; a
    // change frees the clustered plan.  A 1-rank "world" keeps what the caller set: its slab may be a range of a larger matrix --
    // tools/rank_slabs.py.)
    if (!st && world > 1 && h->opt_row_offset != p.row0) st = sextans_set_option(h, "row_offset", p.row0);
    if (nchunks < 1) nchunks = 1;
    if (nchunks > 16) nchunks = 16;
    if (int rc = exchange_nnz(h, r, comm, world, rank, row_ranges, force, st, s, &exchanged)) return rc;
    // Clustered-order chunks (round 5): when this rank's slab runs on a graph-clustered plan, chunks are ranges of the plan's row
    // BLOCKS and every chunk keeps the reordered form (engine.hip: cc_*) -- if every rank of the partition can do the same.
    bool want_cc = false;
    if (nchunks > 1) {
        st = cc_prepare(h, N, &want_cc);
    } else if (h->M > 0) {   // (whole-slab calls: the plan is built here, not inside the first launch)
        std::vector<Seg> plan;
        int W = 0;
        bool up = false, uw = false;
        st = prepare(h, N, plan, W, up, uw, true);
    }
    if (want_cc && h->psc.plan_nblk < nchunks) want_cc = false;
    // Chunk c of rank g = local rows [cuts[g][c], cuts[g][c+1]).  Every rank snaps its OWN interior cuts to the
    // boundaries its kernels want (sextans_align_row: row blocks of the LDS-panel plan, wavefronts of the window kernel,
    // so every chunk keeps the whole-matrix kernel) and the cut positions are exchanged once per (partition, N, chunk
    // count) with a small ncclAllGather; they are cached in the engine afterwards.
    std::vector<int> key(row_ranges, row_ranges + 2 * world);
    key.push_back(N); key.push_back(nchunks); key.push_back(rank); key.push_back(want_cc ? 1 : 0);
    if (force || h->dist_cut_key != key || st) {
        h->dist_cut_key.clear();
        const size_t need = (size_t)world * (size_t)p.m_max;
        if (!st && want_cc && h->dist_rows_cap < need) {   // every rank's position -> global row table: [world][m_max] ints (allocated before the flag exchange carries the status)
            (void)hipFree(h->d_dist_rows);
            h->d_dist_rows = nullptr; h->dist_rows_cap = 0;
            if (hipMalloc((void **)&h->d_dist_rows, sizeof(int) * std::max<size_t>(need, 1)) != hipSuccess) { (void)hipGetLastError(); st = SEXTANS_ERR_HIP; g_last_error = "hipMalloc(row tables) failed"; }
            else h->dist_rows_cap = need;
        }
        bool all_cc = want_cc && !st;
        if (comm) {   // does every rank want clustered-order chunks?  (one int per rank)
            const int mine_f = want_cc ? 1 : 0;
            std::vector<int> f;
            exchanged = true;
            if (int rc = exchange(h, r, comm, world, rank, &mine_f, 1, f, st, s, "ncclAllGather(mode)")) return rc;
            for (int g = 0; g < world; ++g) all_cc = all_cc && f[(size_t)g] != 0;
        } else if (st) {
            return st;
        }
        h->dist_cc = all_cc;
        if (all_cc) {   // tables exchanged once per partition
            cc_table(h, p.row0, h->d_dist_rows + (size_t)rank * p.m_max, s);
            if (comm)
                if (int rc = rccl_check(r->AllGather(h->d_dist_rows + (size_t)rank * p.m_max, h->d_dist_rows, (size_t)p.m_max, 2 /* ncclInt32 */, comm, s),
                                        "ncclAllGather(row tables)"))
                    return rc;
            ++h->dist_exchanges;
            SX_HIP(hipStreamSynchronize(s));
        }
        std::vector<int> mine((size_t)nchunks + 1, 0);
        mine[(size_t)nchunks] = p.m_loc;
        for (int c = 1; c < nchunks && !st; ++c) {
            if (all_cc) {   // positions of the clustered order at block boundaries
                mine[(size_t)c] = h->psc.h_blk_row[(size_t)((int64_t)h->psc.plan_nblk * c / nchunks)];
                continue;
            }
            int a = (int)((int64_t)p.m_loc * c / nchunks);
            st = sextans_align_row(h, N, a, &a);
            mine[(size_t)c] = std::min(std::max(a, mine[(size_t)c - 1]), p.m_loc);
        }
        std::vector<int> all;
        if (!comm) {
            if (st) return st;
            all = mine;
        } else {
            exchanged = true;
            if (int rc = exchange(h, r, comm, world, rank, mine.data(), nchunks + 1, all, st, s, "ncclAllGather(cuts)")) return rc;
        }
        for (int g = 0; g < world && !st; ++g) {   // what arrived must be a monotone cut list of that rank's range
            const int len = row_ranges[2 * g + 1] - row_ranges[2 * g];
            const int *cg = all.data() + (size_t)g * (nchunks + 1);
            if (cg[0] != 0 || cg[nchunks] != len) st = SEXTANS_ERR_STATE;
            for (int c = 0; c < nchunks && !st; ++c)
                if (cg[c + 1] < cg[c]) st = SEXTANS_ERR_STATE;
        }
        if (!st) { h->dist_cuts = all; h->dist_cut_key = key; }
    }
    auto workspaces = [&]() -> int {
        auto cut = [&](int g, int c) { return h->dist_cuts[(size_t)g * (nchunks + 1) + (size_t)c]; };
        out.lmax.assign((size_t)nchunks, 1);
        out.off.assign((size_t)nchunks + 1, 0);
        for (int c = 0; c < nchunks; ++c) {
            for (int g = 0; g < world; ++g) out.lmax[(size_t)c] = std::max<int64_t>(out.lmax[(size_t)c], cut(g, c + 1) - cut(g, c));
            out.off[(size_t)c + 1] = out.off[(size_t)c] + (int64_t)world * N * out.lmax[(size_t)c];
        }
        // staging + per-chunk {row0, len} tables (ints, kept behind the float staging area)
        const size_t meta_floats = (size_t)nchunks * (size_t)world * 2;
        if (h->stage_cap < (size_t)out.off[(size_t)nchunks] + meta_floats) h->dist_meta_at = nullptr;   // new buffer: tables gone
        if (int rc = ensure(&h->d_stage, &h->stage_cap, (size_t)out.off[(size_t)nchunks] + meta_floats)) return rc;
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
        out.d_meta = reinterpret_cast<int *>(h->d_stage + out.off[(size_t)nchunks]);
        if (h->dist_meta != meta || h->dist_meta_at != out.d_meta) {   // the row tables change only with the partition
            SX_HIP(hipMemcpyAsync(out.d_meta, meta.data(), sizeof(int) * meta.size(), hipMemcpyHostToDevice, s));
            ++h->dist_exchanges;
            SX_HIP(hipStreamSynchronize(s));   // `meta` is a host temporary; later calls with the same ranges skip this
            h->dist_meta = meta;
            h->dist_meta_at = out.d_meta;
        }
        out.cc = h->dist_cc;
        if (out.cc)   // row-major staging of the whole C: received slabs are scattered into it, one streaming pass writes column-major C at the end
            if (int rc = ensure(&h->d_Cfull, &h->Cfull_cap, (size_t)p.M_total * (size_t)N)) return rc;
        return SEXTANS_OK;
    };
    if (!st) st = workspaces();
    return settle(h, r, comm, world, rank, st, exchanged, s);
}

// ---- row-major CSR form -----------------------------------------------------------------------------------------------------------------
int setup_rm(sextans_engine *h, Rccl *r, void *comm, int world, int rank, const int *row_ranges, int N, bool packed, hipStream_t s, bool force, Partition &p) {
    bool exchanged = false;
    int st = read_partition(world, rank, row_ranges, 1, p);
    if (!st && p.m_loc != h->M) st = SEXTANS_ERR_INVALID;
    if (!st && !h->d_rp) st = SEXTANS_ERR_STATE;
    if (st && !(force && comm)) return st;
    if (!st && world > 1 && h->opt_row_offset != p.row0) st = sextans_set_option(h, "row_offset", p.row0);
    if (int rc = exchange_nnz(h, r, comm, world, rank, row_ranges, force, st, s, &exchanged)) return rc;
    if (force && h->M > 0) {   // (the lazy form leaves the plan to sextans_spmm_device_rm, which builds it the same way)
        std::vector<Seg> plan;
        int W = 0;
        bool up = false, uw = false;
        st = rm_plan(h, N, plan, W, up, uw, s);
    }
    if (!st && packed && comm) {
        st = ensure(&h->d_stage, &h->stage_cap, (size_t)p.M_total * (size_t)N);
        h->dist_meta_at = nullptr;   // (the column-major form keeps its row tables behind its staging area)
    }
    return settle(h, r, comm, world, rank, st, exchanged || (force && comm), s);
}

// ---- blocked-ELL form -------------------------------------------------------------------------------------------------------------------
struct BellSetup {
    Partition p;
    int64_t lmax = 1;
    size_t slab = 0;
    int *d_meta = nullptr;
};
int setup_bell(sextans_engine *h, Rccl *r, void *comm, int world, int rank, const int *row_ranges, int N, hipStream_t s, bool force, BellSetup &out) {
    int st = read_partition(world, rank, row_ranges, 32, out.p);
    const Partition &p = out.p;
    if (!st && p.m_loc != h->bell_M) st = SEXTANS_ERR_INVALID;
    if (!st && !h->d_bell_Af) st = SEXTANS_ERR_STATE;
    if (st && !(force && comm)) return st;
    auto workspaces = [&]() -> int {
        out.lmax = std::max<int64_t>(1, p.m_max);
        out.slab = (size_t)N * (size_t)out.lmax;
        const size_t meta_floats = (size_t)world * 2;
        if (h->stage_cap < (size_t)world * out.slab + meta_floats) h->dist_meta_at = nullptr;
        if (int rc = ensure(&h->d_stage, &h->stage_cap, (size_t)world * out.slab + meta_floats)) return rc;
        std::vector<int> meta(meta_floats);
        for (int g = 0; g < world; ++g) { meta[2 * (size_t)g] = row_ranges[2 * g]; meta[2 * (size_t)g + 1] = row_ranges[2 * g + 1] - row_ranges[2 * g]; }
        out.d_meta = reinterpret_cast<int *>(h->d_stage + (size_t)world * out.slab);
        if (h->dist_meta != meta || h->dist_meta_at != out.d_meta) {   // the row table changes only with the partition
            SX_HIP(hipMemcpyAsync(out.d_meta, meta.data(), sizeof(int) * meta.size(), hipMemcpyHostToDevice, s));
            ++h->dist_exchanges;
            SX_HIP(hipStreamSynchronize(s));
            h->dist_meta = meta;
            h->dist_meta_at = out.d_meta;
        }
        return SEXTANS_OK;
    };
    if (!st) st = workspaces();
    return settle(h, r, comm, world, rank, st, force && comm, s);
}

int dist_args_ok(sextans_handle_t h, void *comm, int world, int rank, const int *row_ranges) {
    return h && (comm || world == 1) && world >= 1 && rank >= 0 && rank < world && row_ranges;
}
}  // namespace

extern "C" {

int sextans_dist_bind_library(const char *path) {
    std::lock_guard<std::mutex> lk(g_bind_mutex);
    if (g_live_comms > 0) {
        g_last_error = "sextans_dist_bind_library: communicators of the current library are still alive";
        return SEXTANS_ERR_STATE;
    }
    // (the previous library stays mapped: RCCL does not survive a dlclose with its proxy threads, and a handle costs nothing)
    g_bind_path = path && *path ? path : "";
    rccl_bind_locked();
    g_bound = true;
    if (!g_r.lib) { g_last_error = g_rccl_error; return SEXTANS_ERR_STATE; }
    return SEXTANS_OK;
}

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
    { std::lock_guard<std::mutex> lk(g_bind_mutex); ++g_live_comms; }   // (before the call: it blocks until every rank has joined)
    const int rc = rccl_check(r->CommInitRank(comm, world, u, rank), "ncclCommInitRank");
    if (rc) { std::lock_guard<std::mutex> lk(g_bind_mutex); --g_live_comms; }
    return rc;
}

int sextans_dist_comm_destroy(void *comm) {
    Rccl *r = rccl();
    if (!r || !comm) return SEXTANS_ERR_INVALID;
    const int rc = rccl_check(r->CommDestroy(comm), "ncclCommDestroy");
    std::lock_guard<std::mutex> lk(g_bind_mutex);
    if (g_live_comms > 0) --g_live_comms;
    return rc;
}

int sextans_dist_prepare(sextans_handle_t h, void *comm, int world, int rank, const int *row_ranges, int N, int nchunks, int form, void *stream) {
    if (!dist_args_ok(h, comm, world, rank, row_ranges) || N <= 0) return SEXTANS_ERR_INVALID;
    if (form != SEXTANS_DIST_CSR_COLMAJOR && form != SEXTANS_DIST_CSR_ROWMAJOR && form != SEXTANS_DIST_BELL) return SEXTANS_ERR_INVALID;
    if (N % (form == SEXTANS_DIST_BELL ? 32 : 8)) return SEXTANS_ERR_INVALID;
    Rccl *r = comm ? rccl() : nullptr;
    if (comm && !r) return SEXTANS_ERR_STATE;
    SX_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (form == SEXTANS_DIST_CSR_COLMAJOR) {
        CmSetup cs;
        return setup_cm(h, r, comm, world, rank, row_ranges, N, nchunks, s, true, cs);
    }
    if (form == SEXTANS_DIST_CSR_ROWMAJOR) {
        Partition p;
        return setup_rm(h, r, comm, world, rank, row_ranges, N, nchunks < 0, s, true, p);
    }
    BellSetup bs;
    return setup_bell(h, r, comm, world, rank, row_ranges, N, s, true, bs);
}

int sextans_dist_spmm(sextans_handle_t h, void *comm, int world, int rank, const int *row_ranges, int N, float alpha,
                      const float *d_B, int64_t ldb, float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out,
                      int64_t ldc, int nchunks, void *stream) {
    // comm == NULL is allowed for world == 1: the rank's chunks are computed, staged and unpacked exactly as in a multi-rank run, only
    // the collectives are skipped (a 1-rank all-gather is a no-op): single-GPU callers without RCCL, and tools/rank_slabs.py
    if (!dist_args_ok(h, comm, world, rank, row_ranges) || N <= 0 || (N % 8) || !d_B || !d_C_in || !d_C_out) return SEXTANS_ERR_INVALID;
    if (!h->d_rp) return SEXTANS_ERR_STATE;
    Rccl *r = comm ? rccl() : nullptr;
    if (comm && !r) return SEXTANS_ERR_STATE;
    SX_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    // Everything that exchanges, plans, allocates or synchronises -- nothing of it after sextans_dist_prepare for the same
    // (ranges, N, nchunks), or after the first call (the keys are cached in the engine):
    CmSetup cs;
    if (int rc = setup_cm(h, r, comm, world, rank, row_ranges, N, nchunks, s, false, cs)) return rc;
    const Partition &p = cs.p;
    const int64_t M_total = p.M_total;
    const int row0 = p.row0, m_max = p.m_max;
    if (ldc < M_total || ldc_in < M_total || ldb < h->K) return SEXTANS_ERR_INVALID;
    auto cut = [&](int g, int c) { return h->dist_cuts[(size_t)g * (nchunks + 1) + (size_t)c]; };
    const std::vector<int64_t> &lmax = cs.lmax, &off = cs.off;
    int *d_meta = cs.d_meta;
    const bool cc = cs.cc;
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
    if (!dist_args_ok(h, comm, world, rank, row_ranges) || N <= 0 || (N % 8) || !d_B || !d_C_in || !d_C_out || ldb < N || ldc_in < N || ldc < N)
        return SEXTANS_ERR_INVALID;
    if (!h->d_rp) return SEXTANS_ERR_STATE;
    Rccl *r = comm ? rccl() : nullptr;
    if (comm && !r) return SEXTANS_ERR_STATE;
    SX_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const bool packed = ldc != N;
    Partition p;
    if (int rc = setup_rm(h, r, comm, world, rank, row_ranges, N, packed, s, false, p)) return rc;
    const int64_t M_total = p.M_total;
    const int row0 = p.row0, m_loc = p.m_loc;
    const bool equal = p.equal && h->opt_dist_broadcast_runs == 0;   // (tests: the grouped-broadcast exchange on ranges of equal length too)
    if (comm && !equal && (!r->Broadcast || !r->GroupStart || !r->GroupEnd)) {
        g_last_error = "RCCL library lacks ncclBroadcast / ncclGroupStart / ncclGroupEnd (row ranges of unequal length)";
        return SEXTANS_ERR_STATE;
    }
    if (m_loc > 0)
        if (int rc = sextans_spmm_device_rm(h, N, alpha, d_B, ldb, beta, d_C_in + (int64_t)row0 * ldc_in, ldc_in, d_C_out + (int64_t)row0 * ldc, ldc, stream))
            return rc;
    if (!comm) return SEXTANS_OK;
    float *X = d_C_out;   // where the runs are exchanged: C_out itself, or a packed copy of it
    if (packed) {
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
    if (!dist_args_ok(h, comm, world, rank, row_ranges) || N <= 0 || (N % 32) || !d_B || !d_C_in || !d_C_out) return SEXTANS_ERR_INVALID;
    if (!h->d_bell_Af) return SEXTANS_ERR_STATE;
    Rccl *r = comm ? rccl() : nullptr;
    if (comm && !r) return SEXTANS_ERR_STATE;
    SX_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    BellSetup bs;
    if (int rc = setup_bell(h, r, comm, world, rank, row_ranges, N, s, false, bs)) return rc;
    const int64_t M_total = bs.p.M_total, lmax = bs.lmax;
    if (ldc < M_total || ldc_in < M_total) return SEXTANS_ERR_INVALID;
    float *mine = h->d_stage + (size_t)rank * bs.slab;
    if (int rc = sextans_spmm_bell_device2(h, N, alpha, d_B, ldb, beta, d_C_in + bs.p.row0, ldc_in, mine, lmax, stream)) return rc;
    if (comm)
        if (int rc = rccl_check(r->AllGather(mine, h->d_stage, bs.slab, 7 /* ncclFloat */, comm, s), "ncclAllGather(blocked-ELL C)")) return rc;
    hipLaunchKernelGGL(dist_unpack_slabs, dim3((unsigned)((lmax + 255) / 256), (unsigned)N, (unsigned)world), dim3(256), 0, s, h->d_stage, lmax, N,
                       reinterpret_cast<const int2 *>(bs.d_meta), d_C_out, ldc);
    SX_HIP(hipGetLastError());
    return SEXTANS_OK;
}

}  // extern "C"
