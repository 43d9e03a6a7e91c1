// engine.hip -- the SpMM engine behind the C ABI (include/sextans_amd.h): replaces the
// tapa::invoke(Sextans, ...) boundary of the reference (sextans-host.cpp:237-251, sextans.h:20-26).
//
// One engine = one HIP device.  The CSR arrays are uploaded once (sextans_set_matrix_csr).  The first SpMM on a
// matrix (or sextans_align_row / sextans_get_stat) prepares it on the host, outside every timed region like the
// reference's scheduling and packing (sextans-host.cpp:114-148): matrix as set -> [dense 32x32 tiles to bf16 MFMA,
// opt-in] -> source -> [long rows to the piece path] -> main matrix -> packed row-bucketed form (LDS-panel kernel)
// where row blocks reuse B rows.  Every spmm call then enqueues on the caller's stream, without host
// synchronisation in the device-resident form: (1) B repack column-major -> N-tile panels, (2) the main kernel
// (row-group gather / LDS panel / K-window sweep) [+ MFMA pass over the dense tiles, + piece kernel and in-order
// fold for long rows], (3) for the multi-GPU entry the RCCL all-gather of the C slabs, chunk-pipelined.
// There is NO CPU fallback anywhere in this file: no device => SEXTANS_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <type_traits>
#include <vector>

#include "bell_kernels.h"
#include "chan_kernels.h"
#include "panel_plan.h"
#include "plan_device.h"
#include "row_cluster.h"
#include "sextans_amd.h"
#include "spmm_csr_kernels.h"
#include "spmm_panel_v2.h"
#include "spmm_window_kernel.h"
#include "window_plan.h"

namespace {

thread_local std::string g_last_error;

#define SX_HIP(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            char buf_[512];                                                             \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call,                 \
                     hipGetErrorString(e_), __FILE__, __LINE__);                        \
            g_last_error = buf_;                                                        \
            return SEXTANS_ERR_HIP;                                                     \
        }                                                                               \
    } while (0)

struct EventPair { hipEvent_t a, b; };
constexpr int kRowsNoFuseB = 0x100;   // internal flag of sextans_spmm_device_rows: always stage from the repacked panel

}  // namespace

struct sextans_engine {
    int device = 0;
    int num_cus = 256;
    // matrix
    int M = 0, K = 0;
    int64_t nnz = 0;
    const int *d_rp = nullptr, *d_ci = nullptr;
    const float *d_v = nullptr;
    bool owns_matrix = false;
    bool device_matrix_checked = false;   // a caller-provided device matrix has been validated (row_ptr monotone, columns < K)
    // workspaces
    std::set<const void *> big_lds_kernels;   // kernels whose dynamic-LDS limit has been raised ON THIS ENGINE'S DEVICE (the
                                              // attribute is per device: a process-wide flag breaks the second GPU of a process)
    float *d_Bp = nullptr;
    size_t Bp_cap = 0;              // floats
    int bp_layout = 0;              // main panel width of the last repack into d_Bp (0 = none)
    float *d_B = nullptr, *d_Cin = nullptr, *d_Cout = nullptr;   // host-path staging
    size_t B_cap = 0, C_cap = 0;
    hipStream_t host_stream = nullptr;                           // stream of the host-buffer entry points
    float *d_chB = nullptr, *d_chC = nullptr;                    // accelerator channel layouts (sextans_invoke)
    size_t chB_cap = 0, chC_cap = 0;
    // block-dictionary plan for the LDS-panel kernel (built lazily, per lanes_per_row)
    // One packed form per lanes_per_row value (2 / 4 / 8): the active one below, the others parked in plan_stash, so
    // callers that alternate between N classes (N = 8 -> 2 lanes, N >= 16 -> 4) do not rebuild on every switch.
    struct PanelState {
        int plan_lpr = 0;               // 0 = no plan
        int64_t plan_min_reuse = -1;
        // d_dict_ptr: entries per block dictionary; d_dict: dictionaries at stride plan_dict_stride;
        // d_row_off: {first packed entry, entries} per (block, slot)
        int *d_dict_ptr = nullptr, *d_dict = nullptr, *d_blk_row = nullptr, *d_row_off = nullptr;
        int *d_pcol32 = nullptr;
        float *d_pval = nullptr;
        int plan_nblk = 0;
        std::vector<int> h_blk_row;     // host copy of the plan's block boundaries (row-range calls, sextans_align_row)
        unsigned short *d_lidx = nullptr;
        double plan_panel_frac = 0.0;   // share of non-zeros living in dictionary blocks
        double plan_narrow_frac = 0.0;  // sampled share of non-zeros in row blocks that meet the N <= 16 threshold ("panel_min_reuse_x100")
        int plan_max_dict = 0;          // largest block dictionary (entries)
        int plan_max_row = 0;           // longest row of the planned matrix
        int64_t plan_stream_len = 0, plan_nnz_panel = 0;
        int plan_pad_row = 0;           // panel row holding +1.0f for the padding entries = panel capacity in rows
        int plan_dict_stride = 0;       // ints per block in d_dict (dictionaries padded to a common stride)
        bool plan_mixed = false;        // some block with non-zeros has no dictionary (global-gather path needed)
        bool plan_built = false;        // false: only the sampled verdict exists (no packed stream)
    };
    PanelState ps;                      // active
    // The same plan over the rows in CLUSTERED order (row_cluster.hip: brick by brick for grid-stencil matrices), 4 lanes per row,
    // used by spmm_csr_panel_v2 for whole-matrix calls; row-range calls and every other kernel keep the natural-order plan above.
    PanelState psc;
    int *d_slot_row = nullptr;          // psc: row of the main matrix per (block, slot)
    int cluster_state = 0;              // 0 not evaluated, 1 in use, -1 rejected (no grid structure / no gain)
    int64_t cluster_s2 = 0, cluster_s3 = 0;
    int64_t plan_total_dict = 0, cluster_total_dict = 0;   // sum of the block dictionaries: natural order / clustered order
    PanelState plan_stash[3];           // parked, indexed by lanes_per_row 2 / 4 / 8 -> 0 / 1 / 2
    // K-windowed accumulator-resident plan (spmm_csr_window; built lazily)
    uint2 *d_wstream = nullptr;
    int *d_wstep0 = nullptr;
    int win_nwaves = 0, win_rw = 0;
    int64_t win_padded = 0;         // stream entries including padding
    int win_state = 0;              // 0 = not evaluated, 1 = built, -1 = rejected (skewed rows / K too large)
    int64_t win_built_rows = -1, win_built_cols = -1;
    double plan_build_s = 0.0;      // host seconds spent building packed forms of A for the current matrix
    // "MFMA only where a tile is actually dense" (options "mfma_dense_tiles" / "dense_tile_fill_x100"): 32x32 tiles of
    // the main matrix whose fill reaches the threshold, as a blocked-ELL bf16 side matrix; the CSR kernels keep the rest
    int dense_mb = 0, dense_W = 0;  // full block rows, ELL width (0 = no dense tile / not extracted)
    double dense_share = 0.0;       // blocks per distinct block column in groups of 8 block rows of the dense-tile matrix
    int dense_max_union = 0;
    int *d_dense_col = nullptr;
    void *d_dense_Af = nullptr;
    int64_t dense_tiles = 0, dense_nnz = 0;
    int64_t dense_built_mfma = -2, dense_built_fill = -2;
    // blocked-ELL bf16 matrix (MFMA path)
    int bell_M = 0, bell_K = 0, bell_W = 0;
    int bell_max_union = 0;         // largest number of distinct block columns inside a group of 8 block rows
    double bell_share = 0.0;        // blocks per distinct block column inside groups of 8 block rows (1 = no sharing, 8 = identical rows)
    const int *d_bell_col = nullptr;
    int *d_bell_col_owned = nullptr;
    void *d_bell_Af = nullptr;      // A blocks in MFMA fragment order (owned)
    void *d_bell_Bf = nullptr;      // B in fragment order (workspace)
    size_t bell_Bf_cap = 0;         // bytes
    // Long rows leave the "main" matrix -- the CSR arrays every kernel and plan works on, equal to the arrays
    // above when there are none -- and go through the piece path (rows sorted by length, one row group per piece):
    //   bucketed rows (longer than the bucket threshold L0, option "bucket_rows"): ONE piece, summed in order =
    //     still bit-identical to cpu_spmm_CSR; they only leave so that a workgroup of the main kernel never waits
    //     for one long row among 63 short ones;
    //   hub rows (longer than the split threshold T, option "split_rows"): pieces of T entries summed in parallel
    //     and folded in order = re-associated (stated tolerance), reported by sextans_reassociated_rows.
    // Chain of matrices: the matrix as set (d_rp / d_ci / d_v) -> [dense 32x32 tiles cut out, when routed to MFMA] ->
    // "source" (s_*) -> [long rows emptied] -> "main" (m_*).  Without dense tiles / long rows the stages alias.
    const int *s_rp = nullptr, *s_ci = nullptr;
    const float *s_v = nullptr;
    int64_t s_nnz = 0;
    int *d_srp = nullptr, *d_sci = nullptr;   // owned copy of the source (exists only when tiles were cut out)
    float *d_sv = nullptr;
    const int *m_rp = nullptr, *m_ci = nullptr;
    const float *m_v = nullptr;
    int64_t m_nnz = 0;
    int *d_mrp = nullptr, *d_mci = nullptr;   // owned compacted copy (exists only when rows left)
    float *d_mv = nullptr;
    unsigned char *d_skip = nullptr;          // 1 = the row's C is written by the piece path, not by the main kernel
    struct PieceTable {                       // pieces [begin, end) in d_ci / d_v, first piece per long row, the rows
        int *d_vrp = nullptr, *d_vend = nullptr, *d_vfirst = nullptr, *d_row = nullptr;
        std::vector<int> h_row, h_vfirst;
    };
    PieceTable by_len, by_row;                // sorted by length (whole-matrix calls: balanced workgroups) / by row (row ranges)
    // exact chains (strict order, "exact_chain" = 1): rows longer than the automatic threshold leave the piece tables too
    // and are summed by chain_fused -- still one serial chain of rounded adds per (row, column), bit-identical
    int nchain = 0;
    int *d_chain_row = nullptr, *d_chain_beg = nullptr, *d_chain_perm = nullptr;   // perm: chain rows by length, longest first
    long long *d_chain_off = nullptr;                   // prefix of the lengths
    std::vector<int> h_chain_row;
    std::vector<long long> h_chain_off;
    int64_t chain_T = 0;
    int64_t chain_built_opt = -2;
    hipStream_t aux_stream = nullptr;         // the chain kernels need one or two wavefronts for ~1 ms: they run beside the main kernel
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<int> h_split_rows;            // ascending: rows cut into more than one piece
    int nhub = 0;                             // long rows (bucketed + split)
    int split_nv = 0;                         // pieces of all long rows
    int64_t split_T = 0, bucket_L0 = 0;       // thresholds in effect (0 = none)
    int64_t split_built_opt = -2, bucket_built_opt = -2, split_built_gnnz = -2;   // option values the state above was built for
    float *d_P = nullptr;
    size_t P_cap = 0;
    long long *d_dbg = nullptr;     // 8 counters for phase timing (option "phase_timing")
    // native multi-GPU form (sextans_dist_spmm): slab staging S[chunk][world][N][lmax_chunk], communication stream
    float *d_stage = nullptr;
    size_t stage_cap = 0;
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> dist_events;
    std::vector<int> dist_cut_key, dist_cuts;   // (ranges, N, nchunks, rank) the chunk cuts of all ranks were exchanged for
    std::vector<int> dist_meta;     // {first row, rows} per (chunk, rank) as last uploaded, and where
    const int *dist_meta_at = nullptr;
    // options
    int64_t opt_kernel = 0, opt_lpr = 0, opt_stage = 1, opt_xcd = 1, opt_exact = 1, opt_profile = 0;   // opt_lpr 0 = auto
    int64_t opt_cols_per_lane = 0;      // LDS-panel kernel: output columns per lane.  4 (= 0, the default) = 16-column tiles;
                                        // 8 = register-blocked 32-column super tiles (spmm_csr_panel_v2<2>: 2 workgroups per
                                        // CU -- measured slower than 4 columns per lane at 4 workgroups per CU)
    int64_t opt_cluster_shape = 0;      // measurement switch: brick shape run_rows * 10000 + lines * 100 + planes (0 = 16 x 2 x 2 / 16 x 4)
    int64_t opt_cluster_group = 3;      // bricks are laid out in groups of g x g brick columns (A/B on the 4M-row FEM matrix: g = 3)
    int64_t opt_row_cluster = -1;       // clustered-order plan for spmm_csr_panel_v2 (ensure_cluster_plan): -1 auto, 0 never, 1 whenever found
    int64_t opt_small_v2 = 1;           // measurement switch: 0 = small matrices keep the full-capacity, 4-deep form of spmm_csr_panel_v2
    int64_t opt_panel_v2 = -1;          // 16-column tiles on the register-resident form (spmm_csr_panel_v2<1>: row entries
                                        // loaded once per block, panels by LDS-DMA, tile loop inside the workgroup, C stored
                                        // straight from the accumulators): 1 = yes, 0 = no (spmm_csr_panel), -1 = auto: yes
                                        // unless the column-major staging of small matrices applies ("fuse_b")
    int64_t opt_tiles_per_wg = 0;       // wide kernel: super tiles one workgroup walks (A stream from HBM once per that many
                                        // columns); 0 = auto: all of N while the launch still fills the chip several times
    int64_t opt_fuse_b = 1;             // panel kernel may stage from column-major B (small matrices: no repack launch)
    int64_t opt_split_rows = 0;         // 0 (default) = never: every row is summed in strict CSR order, bit-identical to
                                        // cpu_spmm_CSR; > 0: rows longer than this are split (re-associated, opt-in);
                                        // -1 = opt in with the automatic threshold max(1024, global nnz / 16384)
    int64_t opt_global_nnz = 0;         // multi-GPU: non-zeros of the WHOLE matrix (0 = this engine's matrix is the whole
                                        // matrix), so every rank derives the same split threshold as a single GPU would
    int64_t opt_exact_chain = 1;        // strict order ("split_rows" = 0): rows longer than max(1024, nnz / 16384) are summed as
                                        // exact chains (all products in parallel, one lane per column adds them in order);
                                        // 0 = such rows stay on the piece path (one row group, ~80 ns per entry)
    int64_t opt_bucket_rows = -1;       // > 0: rows longer than this take the piece path unsplit (still exact); 0 = off;
                                        // -1 = max(32, 2 * mean row length)
    int64_t opt_phase_timing = 0;       // 1: panel kernel accumulates per-phase wave cycles (debug aid)
    int64_t opt_min_reuse_x100 = 200;   // a block uses the LDS panel if nnz >= 2 * distinct columns (measured: a 1-dof 3-D
                                        // stencil, reuse 2.9, runs 18 % faster on the panel kernel; FEM/banded classes unchanged)
    int64_t opt_min_reuse_wide_x100 = 150;   // the same threshold for N >= 32: with more columns per B row read the panel pays
                                        // earlier (2-D 5-point stencil, reuse 1.65: N = 128 2.28 ms vs 2.60 ms on the gather kernel,
                                        // N = 16 0.370 vs 0.350 ms)
    int64_t opt_win_rows = 319;         // rows per wavefront of the window kernel (+1 dummy row: 4 x 320 x 32 B = 40 KiB)
    int64_t opt_win_cols = 65536;       // columns per K window (x 32 B = 2 MiB of the 8-column panel: half an XCD's L2)
    int64_t opt_win_unroll = 8;         // steps in flight per ring (4 or 8)
    int64_t opt_win_auto = 0;           // 1: "kernel" 0 may pick the window kernel from the fabric-byte model
    int64_t opt_bell_shared = -1;       // N = 256: workgroups of 8 block rows share each B tile through an LDS ring
                                        // (spmm_bell_mfma_shared).  1 = always, 0 = never, -1 = when the 8 block rows of a
                                        // workgroup share block columns (blocks per distinct column >= 1.5)
    int64_t opt_bell_debug = 0;         // measurements only (wrong results): ablation bits of spmm_bell_mfma_shared
    int64_t opt_bell_gen = 0;           // block rows per launch of the wide kernel (0 = all in one launch)
    int64_t opt_bell_wide = 1;          // 1 (default): N = 256 runs one wavefront per block row over all 8 column tiles
                                        // (A requested once, non-temporal); 0: two wavefronts of 4 tiles each
    int64_t opt_mfma_dense = 0;         // 1: dense 32x32 tiles run on the bf16 MFMA path (the caller opts into bf16 rounding
                                        // of those tiles and of B for them); 0: they are only counted (get_stat)
    int64_t opt_dense_fill_x100 = 50;   // a tile is dense when it holds >= this percentage of its 1024 positions
    // profiling
    std::vector<EventPair> ev_kernel, ev_repack;
    const char *last_kernel = "none";
};

namespace {

int check_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_last_error = "hipGetDeviceCount: no HIP device";
        return SEXTANS_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return SEXTANS_ERR_INVALID;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return SEXTANS_ERR_NO_DEVICE;
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        g_last_error = std::string("device is ") + p.gcnArchName + ", engine is built for gfx950 only";
        return SEXTANS_ERR_NO_DEVICE;
    }
    return SEXTANS_OK;
}

void free_panel_state(sextans_engine::PanelState &p) {
    (void)hipFree(p.d_dict_ptr); (void)hipFree(p.d_dict); (void)hipFree(p.d_lidx); (void)hipFree(p.d_blk_row);
    (void)hipFree(p.d_row_off); (void)hipFree(p.d_pcol32); (void)hipFree(p.d_pval);
    p = sextans_engine::PanelState();
}

void free_plan(sextans_engine *h) {   // every packed form of the current main matrix
    free_panel_state(h->ps);
    for (auto &p : h->plan_stash) free_panel_state(p);
    free_panel_state(h->psc);
    (void)hipFree(h->d_slot_row);
    h->d_slot_row = nullptr;
    h->cluster_state = 0;
    h->cluster_s2 = h->cluster_s3 = 0;
    h->plan_total_dict = h->cluster_total_dict = 0;
}

void free_window(sextans_engine *h) {
    (void)hipFree(h->d_wstream); (void)hipFree(h->d_wstep0);
    h->d_wstream = nullptr; h->d_wstep0 = nullptr;
    h->win_nwaves = h->win_rw = 0;
    h->win_padded = 0;
    h->win_state = 0;
    h->win_built_rows = h->win_built_cols = -1;
}

void free_bell(sextans_engine *h) {
    (void)hipFree(h->d_bell_col_owned); (void)hipFree(h->d_bell_Af);
    h->d_bell_col_owned = nullptr; h->d_bell_col = nullptr; h->d_bell_Af = nullptr;
    h->bell_M = h->bell_K = h->bell_W = 0;
}

void free_split(sextans_engine *h) {   // long-row state: main matrix, skip flags, piece tables (built from the source)
    for (auto *t : {&h->by_len, &h->by_row}) {
        (void)hipFree(t->d_vrp); (void)hipFree(t->d_vend); (void)hipFree(t->d_vfirst); (void)hipFree(t->d_row);
        *t = sextans_engine::PieceTable();
    }
    (void)hipFree(h->d_chain_row); (void)hipFree(h->d_chain_beg); (void)hipFree(h->d_chain_off); (void)hipFree(h->d_chain_perm);
    h->d_chain_row = h->d_chain_beg = h->d_chain_perm = nullptr; h->d_chain_off = nullptr;
    h->nchain = 0; h->h_chain_row.clear(); h->h_chain_off.clear(); h->chain_T = 0; h->chain_built_opt = -2;
    (void)hipFree(h->d_mrp); (void)hipFree(h->d_mci); (void)hipFree(h->d_mv); (void)hipFree(h->d_skip);
    h->d_mrp = h->d_mci = nullptr;
    h->d_mv = nullptr;
    h->d_skip = nullptr;
    h->h_split_rows.clear();
    h->nhub = h->split_nv = 0;
    h->split_T = h->bucket_L0 = 0;
    h->split_built_opt = h->bucket_built_opt = h->split_built_gnnz = -2;
    h->m_rp = h->s_rp; h->m_ci = h->s_ci; h->m_v = h->s_v; h->m_nnz = h->s_nnz;
}

void free_dense(sextans_engine *h) {   // dense-tile state and everything downstream of the source matrix
    (void)hipFree(h->d_dense_col); (void)hipFree(h->d_dense_Af);
    h->d_dense_col = nullptr; h->d_dense_Af = nullptr;
    h->dense_mb = h->dense_W = 0;
    h->dense_tiles = h->dense_nnz = 0;
    h->dense_built_mfma = h->dense_built_fill = -2;
    (void)hipFree(h->d_srp); (void)hipFree(h->d_sci); (void)hipFree(h->d_sv);
    h->d_srp = h->d_sci = nullptr;
    h->d_sv = nullptr;
    h->s_rp = h->d_rp; h->s_ci = h->d_ci; h->s_v = h->d_v; h->s_nnz = h->nnz;
    free_split(h);
}

void free_matrix(sextans_engine *h) {
    free_plan(h);
    free_dense(h);
    free_window(h);
    h->plan_build_s = 0.0;
    if (h->owns_matrix) {
        (void)hipFree((void *)h->d_rp);
        (void)hipFree((void *)h->d_ci);
        (void)hipFree((void *)h->d_v);
    }
    h->d_rp = h->d_ci = nullptr;
    h->d_v = nullptr;
    h->owns_matrix = false;
    h->device_matrix_checked = false;
    h->m_rp = h->m_ci = h->s_rp = h->s_ci = nullptr; h->m_v = h->s_v = nullptr; h->m_nnz = h->s_nnz = 0;
    h->dist_cut_key.clear();   // chunk cuts are aligned to the packed forms of one matrix
    h->bp_layout = 0;          // B panels belong to one (K, B)
}

int ensure(float **p, size_t *cap, size_t need) {
    if (*cap >= need && *p) return SEXTANS_OK;
    if (*p) SX_HIP(hipFree(*p));
    *p = nullptr; *cap = 0;
    SX_HIP(hipMalloc((void **)p, (need ? need : 1) * sizeof(float)));
    *cap = need;
    return SEXTANS_OK;
}

struct Prof {
    sextans_engine *h; std::vector<EventPair> *vec; hipStream_t s; bool on; EventPair ep{};
    Prof(sextans_engine *h_, std::vector<EventPair> *v, hipStream_t s_) : h(h_), vec(v), s(s_), on(h_->opt_profile != 0) {
        if (on) {
            (void)hipEventCreate(&ep.a); (void)hipEventCreate(&ep.b);
            (void)hipEventRecord(ep.a, s);
        }
    }
    ~Prof() {
        if (on) { (void)hipEventRecord(ep.b, s); vec->push_back(ep); }
    }
};

template <int W>
void launch_repack(const float *dB, int64_t ldb, float *dBp, int K, int col_base, int ntiles,
                   hipStream_t s) {
    dim3 grid((unsigned)((K + sx::kBlock - 1) / sx::kBlock), (unsigned)ntiles);
    hipLaunchKernelGGL(sx::repack_b_panels<W>, grid, dim3(sx::kBlock), 0, s, dB, ldb, dBp, K,
                       col_base);
}

template <int LPR>
void launch_rowgroup(sextans_engine *h, const int *rp, const int *rend, const int *ci, const float *va, bool pieces,
                     const unsigned char *skip, const float *dBp,
                     const float *dCin, int64_t ldc_in, float *dCout, int64_t ldc, int row_begin, int row_end, int ntiles,
                     float alpha, float beta, hipStream_t s) {
    constexpr int RB = sx::kBlock / LPR;
    constexpr int CH = 2048;
    const int nrowblk = (row_end - row_begin + RB - 1) / RB;
    if (nrowblk <= 0) return;
    const unsigned nwg = (unsigned)nrowblk * (unsigned)ntiles;
    const int64_t pstride = (int64_t)h->K * 4 * LPR;
    const int xcd = (int)h->opt_xcd;
#define SX_LAUNCH(EX, ST)                                                                       \
    hipLaunchKernelGGL((sx::spmm_csr_rowgroup<LPR, CH, EX, ST>), dim3(nwg), dim3(sx::kBlock), 0, \
                       s, rp, rend, ci, va, dBp, pstride, dCin, ldc_in, dCout, ldc, row_begin,          \
                       row_end, ntiles, nrowblk, alpha, beta, xcd, skip)
    // The LDS-staged A stream walks a block's non-zeros in order, which serialises row groups when rows
    // are long pieces of one hub row (split mode): there every row group streams its own piece directly.
    const bool stage = h->opt_stage && !pieces;
    if (h->opt_exact) { if (stage) SX_LAUNCH(true, true); else SX_LAUNCH(true, false); }
    else              { if (stage) SX_LAUNCH(false, true); else SX_LAUNCH(false, false); }
#undef SX_LAUNCH
}

constexpr int kPanelFloats = 9216;    // at most 36 KiB of LDS for the B panel (576 rows at N-tile 16)

template <class T>
int upload(T **dst, const std::vector<T> &src) {
    SX_HIP(hipMalloc((void **)dst, sizeof(T) * (src.empty() ? 1 : src.size())));
    if (!src.empty()) SX_HIP(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    return SEXTANS_OK;
}

struct PlanTimer {   // accumulates host seconds spent packing A (reported by sextans_get_stat "plan_build_s")
    sextans_engine *h; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit PlanTimer(sextans_engine *h_) : h(h_) {}
    ~PlanTimer() { h->plan_build_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// Device copy of the CSR arrays -> host, validated: the host-side plan builders index arrays of size K with
// the column indices and trust row_ptr to be monotonic (a matrix handed over with
// sextans_set_matrix_csr_device has not been looked at by anybody yet).
// (level 2: the main matrix the kernels work on; 1: the source of the long-row split; 0: the matrix as the caller set it)
int read_back_row_ptr(sextans_engine *h, std::vector<int> &rp, int level = 2) {
    rp.resize((size_t)h->M + 1);
    const int *src = level == 2 ? h->m_rp : level == 1 ? h->s_rp : h->d_rp;
    SX_HIP(hipMemcpy(rp.data(), src, sizeof(int) * ((size_t)h->M + 1), hipMemcpyDeviceToHost));
    if (rp[0] != 0 || (int64_t)rp[(size_t)h->M] != (level == 2 ? h->m_nnz : level == 1 ? h->s_nnz : h->nnz)) return SEXTANS_ERR_INVALID;
    for (int r = 0; r < h->M; ++r)
        if (rp[(size_t)r + 1] < rp[(size_t)r]) return SEXTANS_ERR_INVALID;
    return SEXTANS_OK;
}
int read_back_entries(sextans_engine *h, std::vector<int> &ci, std::vector<float> &va, int level = 2) {
    const int64_t nnz = level == 2 ? h->m_nnz : level == 1 ? h->s_nnz : h->nnz;
    const size_t n1 = (size_t)(nnz ? nnz : 1);
    ci.assign(n1, 0); va.assign(n1, 0.f);
    if (nnz) {
        SX_HIP(hipMemcpy(ci.data(), level == 2 ? h->m_ci : level == 1 ? h->s_ci : h->d_ci, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost));
        SX_HIP(hipMemcpy(va.data(), level == 2 ? h->m_v : level == 1 ? h->s_v : h->d_v, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToHost));
    }
    const unsigned K = (unsigned)h->K;
    unsigned bad = 0;
    for (int64_t j = 0; j < nnz; ++j) bad |= (unsigned)((unsigned)ci[(size_t)j] >= K);
    return bad ? SEXTANS_ERR_INDEX : SEXTANS_OK;
}

// Build (or reuse) the packed row-bucketed form of A for `lpr` lanes per row.  The CSR arrays are read
// back from the device copy, so this works for host- and device-provided matrices alike; it runs once
// per matrix ("upload once"), outside any timed region, like the reference's host-side scheduling
// and packing (generate_edge_list_for_all_PEs + edge_list_64bit, sextans-host.cpp:114-148).
// Cheap pre-test on a sample of row blocks: share of sampled non-zeros that sit in blocks with
// nnz >= min_reuse * distinct columns.  Lets "auto" skip the full plan build on matrices without
// reuse (e.g. uniformly random columns).
int64_t plan_key(const sextans_engine *h) { return h->opt_min_reuse_x100 * 100000 + h->opt_min_reuse_wide_x100; }

int sample_reuse(sextans_engine *h, int RB, int max_unique, double min_reuse, double min_reuse2, double *frac, double *frac2) {
    const int nblk = (h->M + RB - 1) / RB;
    const int nsample = std::min(nblk, 512);
    std::vector<int> rp;
    if (int rc = read_back_row_ptr(h, rp)) return rc;
    int64_t tot = 0, good = 0, good2 = 0;
    std::vector<int> cols;
    for (int sidx = 0; sidx < nsample; ++sidx) {
        const int b = (int)((int64_t)sidx * nblk / nsample);
        const int r0 = b * RB, r1 = std::min(h->M, r0 + RB);
        const int j0 = rp[(size_t)r0], j1 = rp[(size_t)r1];
        if (j1 <= j0) continue;
        cols.resize((size_t)(j1 - j0));
        SX_HIP(hipMemcpy(cols.data(), h->m_ci + j0, sizeof(int) * cols.size(), hipMemcpyDeviceToHost));
        std::sort(cols.begin(), cols.end());
        const int64_t distinct = std::unique(cols.begin(), cols.end()) - cols.begin();
        tot += j1 - j0;
        // a block larger than the panel is split by the real builder; its reuse ratio carries over
        if ((double)(j1 - j0) >= min_reuse * (double)distinct) good += j1 - j0;
        if ((double)(j1 - j0) >= min_reuse2 * (double)distinct) good2 += j1 - j0;
        (void)max_unique;
    }
    *frac = tot ? (double)good / (double)tot : 0.0;
    *frac2 = tot ? (double)good2 / (double)tot : 0.0;
    return SEXTANS_OK;
}

// Build (or reuse) the packed row-bucketed form of A for `lpr` lanes per row.  The CSR arrays are read
// back from the device copy, so this works for host- and device-provided matrices alike; it runs once
// per matrix ("upload once"), outside any timed region, like the reference's host-side scheduling
// and packing (generate_edge_list_for_all_PEs + edge_list_64bit, sextans-host.cpp:114-148).
int ensure_plan(sextans_engine *h, int lpr, bool force) {
    if (h->ps.plan_lpr == lpr && h->ps.plan_min_reuse == plan_key(h) && (h->ps.plan_built || !force))
        return SEXTANS_OK;
    if (h->ps.plan_lpr != lpr) {   // park the active form, bring back the one for this lane count (if any)
        auto idx = [](int l) { return l == 2 ? 0 : l == 4 ? 1 : 2; };
        if (h->ps.plan_lpr) std::swap(h->ps, h->plan_stash[idx(h->ps.plan_lpr)]);
        if (h->ps.plan_lpr != lpr) std::swap(h->ps, h->plan_stash[idx(lpr)]);
        if (h->ps.plan_lpr && h->ps.plan_lpr != lpr) {   // displaced a third form: park it in its own slot
            std::swap(h->ps, h->plan_stash[idx(h->ps.plan_lpr)]);
            free_panel_state(h->ps);
        }
        if (h->ps.plan_lpr == lpr && h->ps.plan_min_reuse == plan_key(h) && (h->ps.plan_built || !force))
            return SEXTANS_OK;
    }
    free_panel_state(h->ps);
    PlanTimer timer(h);
    // A matrix handed over with sextans_set_matrix_csr_device has not been looked at by anybody yet: the kernels gather B
    // rows by column index and the builders trust row_ptr to be monotone, so it is validated once, on the device.
    if (!h->owns_matrix && !h->device_matrix_checked) {
        int bad = 0;
        std::string verr;
        if (sx::validate_csr_device(h->M, h->K, h->nnz, h->d_rp, h->d_ci, &bad, verr)) { g_last_error = verr; return SEXTANS_ERR_HIP; }
        if (bad) return (bad & 1) ? SEXTANS_ERR_INVALID : SEXTANS_ERR_INDEX;
        h->device_matrix_checked = true;
    }
    const int RB = sx::kBlock / lpr;
    // two thresholds: "panel_min_reuse_x100" decides for N <= 16, "panel_min_reuse_wide_x100" for N >= 32 (prepare()); the plan
    // is built once, for the lower of the two, so that alternating N never rebuilds it
    const double narrow = (double)h->opt_min_reuse_x100 / 100.0;
    const double min_reuse = std::min(narrow, (double)h->opt_min_reuse_wide_x100 / 100.0);
    double narrow_frac = 1.0;
    if (!force) {
        double frac = 0.0;
        if (int rc = sample_reuse(h, RB, kPanelFloats / (4 * lpr), min_reuse, narrow, &frac, &narrow_frac)) return rc;
        if (frac < 0.5) {   // no reuse worth an LDS panel: remember the verdict, skip the build
            h->ps.plan_lpr = lpr;
            h->ps.plan_min_reuse = plan_key(h);
            h->ps.plan_panel_frac = frac * 0.999;
            h->ps.plan_narrow_frac = narrow_frac * 0.999;
            h->ps.plan_built = false;
            return SEXTANS_OK;
        }
    }
    // The packed form is built on the device (plan_device.hip): the CSR arrays never leave HBM.
    sx::DevicePlan dp;
    std::string err;
    const int cap = kPanelFloats / (4 * lpr);
    const int brc = sx::build_panel_plan_device(h->M, h->K, h->m_rp, h->m_ci, h->m_v, lpr, cap, min_reuse, dp, err);
    if (brc == 2) { g_last_error = err; sx::free_device_plan(dp); return SEXTANS_ERR_HIP; }
    h->ps.plan_lpr = lpr;
    h->ps.plan_min_reuse = plan_key(h);
    h->ps.plan_narrow_frac = narrow_frac;
    if (brc == 1) {   // rows padded to 4 entries exceed 32-bit entry offsets: row-group kernel only
        h->ps.plan_panel_frac = 0.0;
        h->ps.plan_built = false;
        return SEXTANS_OK;
    }
    if (dp.dict_stride > 9 * RB) { sx::free_device_plan(dp); return SEXTANS_ERR_STATE; }   // capacity = 9 * RB by construction
    h->ps.plan_nblk = dp.nblk;
    h->ps.plan_dict_stride = dp.dict_stride;
    h->ps.plan_mixed = dp.mixed;
    h->ps.d_blk_row = dp.d_blk_row; h->ps.d_dict_ptr = dp.d_dict_cnt; h->ps.d_dict = dp.d_dict; h->ps.d_row_off = dp.d_slot_info;
    h->ps.d_lidx = dp.d_idx16; h->ps.d_pcol32 = dp.d_col32; h->ps.d_pval = dp.d_val;
    h->ps.h_blk_row.swap(dp.h_blk_row);
    h->ps.plan_stream_len = dp.stream_len;
    h->ps.plan_panel_frac = h->m_nnz ? (double)dp.nnz_in_panel_blocks / (double)h->m_nnz : 0.0;
    h->ps.plan_nnz_panel = dp.nnz_in_panel_blocks;
    h->ps.plan_max_dict = dp.max_dict;
    h->ps.plan_max_row = dp.max_row_len;
    h->ps.plan_pad_row = cap;
    h->ps.plan_built = true;
    if (lpr == 4) h->plan_total_dict = dp.total_dict;
    return SEXTANS_OK;
}

// Clustered-order plan (see PanelState psc): built once per matrix, after the natural-order plan for 4 lanes per row exists and is
// dictionary-only.  Option "row_cluster": -1 = when the matrix has grid-stencil structure AND the clustered plan copies at least
// 15 % fewer B rows into LDS; 1 = whenever the structure is found; 0 = never.
int ensure_cluster_plan(sextans_engine *h) {
    if (h->cluster_state != 0) return SEXTANS_OK;
    h->cluster_state = -1;
    if (h->opt_row_cluster == 0 || !h->ps.plan_built || h->ps.plan_lpr != 4 || h->ps.plan_mixed || h->M < 4096) return SEXTANS_OK;
    PlanTimer timer(h);
    // ---- grid strides from the columns of ~128 rows out of the middle half of the matrix
    std::vector<int> rp;
    if (int rc = read_back_row_ptr(h, rp)) return rc;
    std::vector<int> rows;
    std::vector<std::vector<int>> cols;
    const int nsample = 128;
    for (int k = 0; k < nsample; ++k) {
        const int r = (int)((int64_t)h->M / 4 + (int64_t)k * (h->M / 2) / nsample);
        const int j0 = rp[(size_t)r], j1 = rp[(size_t)r + 1];
        if (j1 <= j0 || j1 - j0 > 4096) continue;
        std::vector<int> c((size_t)(j1 - j0));
        SX_HIP(hipMemcpy(c.data(), h->m_ci + j0, sizeof(int) * c.size(), hipMemcpyDeviceToHost));
        rows.push_back(r);
        cols.push_back(std::move(c));
    }
    sx::GridStrides gs;
    if (!sx::detect_grid_strides(h->M, rows, cols, &gs)) return SEXTANS_OK;
    h->cluster_s2 = gs.s2;
    h->cluster_s3 = gs.s3;
    // ---- bricks of <= 64 rows = one row block each: a run of 15 / 16 rows of a grid line x 2 lines x 2 planes (3-D), x 4 lines (2-D).
    // Runs of consecutive rows keep a wavefront's C accesses (its 16 row slots) on consecutive rows -- 64-byte runs per column as in
    // natural order (12-row runs gave away half of the gain at N = 128, where C is half of the traffic); the plan builder starts a
    // block at every brick (`cut`), so blocks and bricks coincide.
    int run_rows = 16, b2 = gs.s3 > 0 ? 2 : 4, b3 = gs.s3 > 0 ? 2 : 1;
    if (h->opt_cluster_shape > 0) {   // EXPERIMENT: run_rows * 10000 + b2 * 100 + b3
        run_rows = (int)(h->opt_cluster_shape / 10000); b2 = (int)(h->opt_cluster_shape / 100 % 100); b3 = (int)(h->opt_cluster_shape % 100);
    }
    std::string err;
    int *d_perm = nullptr, *prp = nullptr, *pci = nullptr;
    unsigned char *d_cut = nullptr;
    float *pv = nullptr;
    auto drop = [&]() { (void)hipFree(d_perm); (void)hipFree(d_cut); (void)hipFree(prp); (void)hipFree(pci); (void)hipFree(pv); };
    if (sx::build_brick_order_device(h->M, gs, run_rows, b2, b3, (int)h->opt_cluster_group, &d_perm, &d_cut, err)) { g_last_error = err; drop(); return SEXTANS_ERR_HIP; }
    if (sx::permute_csr_rows_device(h->M, h->m_nnz, h->m_rp, h->m_ci, h->m_v, d_perm, &prp, &pci, &pv, err)) {
        g_last_error = err; drop(); return SEXTANS_ERR_HIP;
    }
    sx::DevicePlan dp;
    const int lpr = 4, RB = sx::kBlock / lpr, cap = kPanelFloats / (4 * lpr);
    const double min_reuse = std::min((double)h->opt_min_reuse_x100, (double)h->opt_min_reuse_wide_x100) / 100.0;
    const int brc = sx::build_panel_plan_device(h->M, h->K, prp, pci, pv, lpr, cap, min_reuse, dp, err, d_cut);
    (void)hipFree(prp); (void)hipFree(pci); (void)hipFree(pv); (void)hipFree(d_cut);
    prp = pci = nullptr; pv = nullptr; d_cut = nullptr;
    if (brc == 2) { g_last_error = err; sx::free_device_plan(dp); drop(); return SEXTANS_ERR_HIP; }
    h->cluster_total_dict = dp.total_dict;
    const bool gain = (double)dp.total_dict <= 0.85 * (double)h->plan_total_dict;
    if (brc != 0 || dp.mixed || dp.dict_stride > 9 * RB || dp.max_dict > sx::kWideMaxDict || (h->opt_row_cluster < 0 && !gain)) {
        sx::free_device_plan(dp); drop();
        return SEXTANS_OK;
    }
    if (sx::build_slot_rows_device(dp.nblk, RB, dp.d_blk_row, d_perm, &h->d_slot_row, err)) {
        g_last_error = err; sx::free_device_plan(dp); drop(); return SEXTANS_ERR_HIP;
    }
    (void)hipFree(d_perm);
    sextans_engine::PanelState &c = h->psc;
    c.plan_lpr = lpr;
    c.plan_min_reuse = plan_key(h);
    c.plan_nblk = dp.nblk;
    c.plan_dict_stride = dp.dict_stride;
    c.plan_mixed = false;
    c.d_blk_row = dp.d_blk_row; c.d_dict_ptr = dp.d_dict_cnt; c.d_dict = dp.d_dict; c.d_row_off = dp.d_slot_info;
    c.d_lidx = dp.d_idx16; c.d_pcol32 = dp.d_col32; c.d_pval = dp.d_val;
    c.h_blk_row.swap(dp.h_blk_row);
    c.plan_stream_len = dp.stream_len;
    c.plan_panel_frac = h->ps.plan_panel_frac;
    c.plan_narrow_frac = h->ps.plan_narrow_frac;
    c.plan_nnz_panel = dp.nnz_in_panel_blocks;
    c.plan_max_dict = dp.max_dict;
    c.plan_max_row = dp.max_row_len;
    c.plan_pad_row = cap;
    c.plan_built = true;
    h->cluster_state = 1;
    return SEXTANS_OK;
}

// dBp: repacked panel (bcol_ld == 0) or the caller's column-major B at this segment's first column with its
// leading dimension bcol_ld (dictionary-only plans, small B: no repack launch).
template <int LPR>
void launch_panel(sextans_engine *h, const float *dBp, const float *dCin, int64_t ldc_in, float *dCout,
                  int64_t ldc, int ntiles, float alpha, float beta, hipStream_t s, int64_t bcol_ld, int blk_begin,
                  int blk_end, int row_base) {
    constexpr int RB = sx::kBlock / LPR;
    constexpr int NT = 4 * LPR;
    const int nblk = blk_end - blk_begin;
    if (nblk <= 0) return;
    const unsigned nwg = (unsigned)nblk * (unsigned)ntiles;
    const int64_t pstride = bcol_ld > 0 ? bcol_ld : (int64_t)h->K * NT;
    const int xcd = (int)h->opt_xcd;
    // LDS = B panel sized for the largest dictionary of this matrix (rounded to 1 KiB) + C tile.
    const int panel_floats = (h->ps.plan_pad_row + 1) * NT;   // dictionary capacity + the +1.0f row the padding entries address
    const int tile_floats = NT * (RB + 1);   // the C tile reuses the panel bytes
    const size_t lds = (size_t)(panel_floats > tile_floats ? panel_floats : tile_floats) * sizeof(int);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(sx::kBlock), lds, s, (const int2 *)h->ps.d_row_off, h->ps.d_lidx,
                           h->ps.d_pcol32, h->ps.d_pval, h->ps.d_blk_row, h->ps.d_dict_ptr, h->ps.d_dict, h->ps.plan_dict_stride, dBp,
                           pstride, dCin, ldc_in, dCout, ldc, ntiles, nblk, alpha, beta, xcd, panel_floats,
                           (long long *)h->d_dbg, blk_begin, row_base, (const unsigned char *)h->d_skip);
    };
    if (h->ps.plan_mixed) {
        if (h->opt_exact) go(sx::spmm_csr_panel<LPR, true, true>);
        else go(sx::spmm_csr_panel<LPR, false, true>);
    } else if (bcol_ld > 0) {
        if (h->opt_exact) go(sx::spmm_csr_panel<LPR, true, false, true>);
        else go(sx::spmm_csr_panel<LPR, false, false, true>);
    } else {
        if (h->opt_exact) go(sx::spmm_csr_panel<LPR, true, false>);
        else go(sx::spmm_csr_panel<LPR, false, false>);
    }
}

// Kernels that need more than the default 64 KiB of dynamic LDS: raise the limit once per (engine = device, kernel).
int allow_big_lds(sextans_engine *h, const void *kern, int bytes) {
    if (h->big_lds_kernels.count(kern)) return SEXTANS_OK;
    SX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    h->big_lds_kernels.insert(kern);
    return SEXTANS_OK;
}

// Wide-N form of the panel kernel (spmm_panel_v2.h): `nsuper` super tiles of 32 columns starting at the pointers
// given; dictionary-only plans built for 4 lanes per row.
template <int H>
int launch_panel_v2(sextans_engine *h, const float *dBp, const float *dCin, int64_t ldc_in, float *dCout, int64_t ldc,
                      int nsuper, float alpha, float beta, hipStream_t s, int64_t bcol_ld, int blk_begin, int blk_end,
                      int row_base, bool clustered = false) {
    // (clustered: the plan over the rows in brick order, whole-matrix calls only; its slot -> row table addresses C)
    const sextans_engine::PanelState &P = clustered ? h->psc : h->ps;
    const int *slot_row = clustered ? h->d_slot_row : nullptr;
    const int nblk = blk_end - blk_begin;
    if (nblk <= 0 || nsuper <= 0) return SEXTANS_OK;
    int tpw = (int)h->opt_tiles_per_wg;
    if (tpw <= 0) {   // all of N in one workgroup while that still leaves >= 4 rounds of workgroups (2 per CU)
        const int64_t rounds = (int64_t)nblk * nsuper / ((int64_t)8 * h->num_cus);
        tpw = (int)std::max<int64_t>(1, std::min<int64_t>(nsuper, rounds));
    }
    tpw = std::min(tpw, nsuper);
    const int ngrp = (nsuper + tpw - 1) / tpw;
    const int64_t pstride = bcol_ld > 0 ? bcol_ld : (int64_t)h->K * 16;
    const size_t lds = (size_t)H * sx::kWideHalfBytes;
    auto go = [&](auto kern) -> int {
        if (int rc = allow_big_lds(h, reinterpret_cast<const void *>(kern), (int)lds)) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk * (unsigned)ngrp), dim3(sx::kBlock), lds, s, (const int2 *)P.d_row_off,
                           P.d_lidx, P.d_pval, P.d_blk_row, P.d_dict_ptr, P.d_dict, P.plan_dict_stride,
                           dBp, pstride, dCin, ldc_in, dCout, ldc, nsuper, tpw, nblk, alpha, beta, (int)h->opt_xcd,
                           P.plan_pad_row, blk_begin, row_base, (const unsigned char *)h->d_skip, (long long *)h->d_dbg, slot_row);
        return SEXTANS_OK;
    };
    // register-resident batches (16 entries each) per row: from the mean row length of the main matrix, so that matrices
    // with short rows (1-dof stencils: 27 entries) do not fetch six batches per row
    const int64_t mean_len = h->M > 0 ? h->m_nnz / h->M : 0;
    int nb = mean_len + 8 <= 32 ? 2 : mean_len + 8 <= 64 ? 4 : 6;
    {   // ... corrected by the longest row: no more batches than any row has, and one more when that makes EVERY row
        // register-resident (nasa4704: mean 22, longest 42 -- a quarter of the wavefronts otherwise finish a row from the
        // stream, one L2 round trip per 16 entries, and their workgroup waits for them)
        const int nb_max = std::max(1, (P.plan_max_row + 15) / 16);
        if (nb_max <= nb) nb = nb_max <= 2 ? 2 : nb_max <= 3 ? 3 : nb_max <= 4 ? 4 : 6;
        else if (nb == 2 && nb_max == 3) nb = 3;
    }
    if constexpr (H > 1) {
        if (bcol_ld > 0) return h->opt_exact ? go(sx::spmm_csr_panel_v2<H, 6, true, true>) : go(sx::spmm_csr_panel_v2<H, 6, false, true>);
        return h->opt_exact ? go(sx::spmm_csr_panel_v2<H, 6, true, false>) : go(sx::spmm_csr_panel_v2<H, 6, false, false>);
    } else {
        // small matrices staged from column-major B: dictionary capacity from the plan (5 x 64 covers nasa4704's 300)
        const bool small_dict = bcol_ld > 0 && P.plan_max_dict <= 5 * 64;
        if (h->opt_phase_timing && h->d_dbg && h->opt_exact) {   // diagnostic instantiations: the forms the dispatcher uses most
            if (small_dict && nb == 3 && h->opt_small_v2 != 0) return go(sx::spmm_csr_panel_v2<H, 3, true, true, true, 5>);
            if (bcol_ld > 0) return go(sx::spmm_csr_panel_v2<H, 2, true, true, true>);
            if (nb == 6) return go(sx::spmm_csr_panel_v2<H, 6, true, false, true>);
        }
        if (small_dict && h->opt_small_v2 != 0) {
            if (nb == 3) return h->opt_exact ? go(sx::spmm_csr_panel_v2<H, 3, true, true, false, 5>) : go(sx::spmm_csr_panel_v2<H, 3, false, true, false, 5>);
            return h->opt_exact ? go(sx::spmm_csr_panel_v2<H, 2, true, true, false, 5>) : go(sx::spmm_csr_panel_v2<H, 2, false, true, false, 5>);
        }
        if (bcol_ld > 0) return h->opt_exact ? go(sx::spmm_csr_panel_v2<H, 2, true, true>) : go(sx::spmm_csr_panel_v2<H, 2, false, true>);
        if (nb == 3) return h->opt_exact ? go(sx::spmm_csr_panel_v2<H, 3, true, false>) : go(sx::spmm_csr_panel_v2<H, 3, false, false>);
        if (nb == 2) return h->opt_exact ? go(sx::spmm_csr_panel_v2<H, 2, true, false>) : go(sx::spmm_csr_panel_v2<H, 2, false, false>);
        if (nb == 4) return h->opt_exact ? go(sx::spmm_csr_panel_v2<H, 4, true, false>) : go(sx::spmm_csr_panel_v2<H, 4, false, false>);
        return h->opt_exact ? go(sx::spmm_csr_panel_v2<H, 6, true, false>) : go(sx::spmm_csr_panel_v2<H, 6, false, false>);
    }
}

// Traffic model behind the automatic choice between the gather kernel and the window kernel for matrices
// without B-row reuse (bytes crossing the L2 <-> memory fabric per SpMM):
//   gather: every non-zero pulls max(128, 4 * tile width) bytes of B (a 64-byte B row still costs a
//           128-byte line) + the 8-byte CSR entry per N tile;
//   window: every XCD streams the whole 8-column panel once per sweep, sweeps = rows / rows whose partial
//           sums the chip holds in LDS at once (at least 1), + the 8-byte stream entry, per 8-column tile.
bool window_pays(const sextans_engine *h, int N, int64_t padded) {
    // Measured on MI355X (profiles/r02_window_kernel_*.txt): the model below counts fabric BYTES, but both
    // kernels are bound by line REQUESTS (~57 G/s beyond L2, ~135 G/s from L2), the sweep issues two 32-byte
    // row reads per non-zero at N = 16 where the gather issues one 64-byte read, and without a chip-wide
    // window barrier the wavefronts drift apart by more than the 4 MiB L2 holds (L2 hit rate 20 %).  The
    // window kernel never won a measurement, so "auto" only considers it when option "window_auto" is set.
    if (!h->opt_win_auto) return false;
    if (N > 24 || h->m_nnz == 0) return false;
    const double K = (double)h->K, nnz = (double)h->m_nnz, M = (double)h->M;
    if (K * N * 4.0 <= 48.0 * 1048576.0) return false;   // B (nearly) fits the L2s: gathers stay on chip
    double gather = 0.0;
    int rest = N;
    for (int w : {16, 8}) { const int nt = rest / w; gather += nt * nnz * (std::max(128.0, 4.0 * w) + 8.0); rest -= nt * w; }
    const double live = (double)h->num_cus * 16.0 * (double)h->opt_win_rows;
    const double sweeps = std::max(1.0, M / live);
    const double window = (N / 8) * (sweeps * 8.0 * K * 32.0 + 8.0 * (double)padded);
    return window < 0.75 * gather;
}

// Build (or reuse) the K-windowed stream of A.  force: "kernel" = 3 (no pay-off / skew test).
int ensure_window(sextans_engine *h, bool force) {
    if (h->win_state != 0 && h->win_built_rows == h->opt_win_rows && h->win_built_cols == h->opt_win_cols &&
        (h->win_state == 1 || !force))
        return SEXTANS_OK;
    free_window(h);
    PlanTimer timer(h);
    h->win_built_rows = h->opt_win_rows;
    h->win_built_cols = h->opt_win_cols;
    h->win_state = -1;
    const int RW = (int)h->opt_win_rows;
    if (RW < 1 || RW > sx::kWinMaxRowsPerWave || h->opt_win_cols < 1 || h->opt_win_cols > 0x7fffffff ||
        (int64_t)h->K > ((int64_t)1 << sx::kWinColBits) || h->m_nnz == 0)
        return SEXTANS_OK;
    std::vector<int> rp, ci;
    std::vector<float> va;
    if (int rc = read_back_row_ptr(h, rp)) return rc;
    if (!force && (double)sx::window_plan_padded_lower_bound(h->M, rp.data(), RW) > 1.3 * (double)h->m_nnz)
        return SEXTANS_OK;   // skewed rows: one row per step would be mostly padding
    if (int rc = read_back_entries(h, ci, va)) return rc;
    sx::WindowPlan plan;
    if (!sx::build_window_plan(h->M, h->K, rp.data(), ci.data(), va.data(), RW, (int)h->opt_win_cols, plan))
        return SEXTANS_OK;
    if (!force && (double)plan.padded > 1.35 * (double)h->m_nnz) return SEXTANS_OK;
    static_assert(sizeof(sx::WinEntry) == sizeof(uint2), "stream entries are loaded as uint2");
    SX_HIP(hipMalloc((void **)&h->d_wstream, sizeof(uint2) * plan.stream.size()));
    SX_HIP(hipMemcpy(h->d_wstream, plan.stream.data(), sizeof(uint2) * plan.stream.size(), hipMemcpyHostToDevice));
    if (int rc = upload(&h->d_wstep0, plan.wave_step0)) return rc;
    h->win_nwaves = plan.nwaves;
    h->win_rw = RW;
    h->win_padded = plan.padded;
    h->win_state = 1;
    return SEXTANS_OK;
}

// dBp8: N/8 row-major K x 8 panels.  Rows [wave_begin * RW, min(M, wave_end * RW)); the C pointers address
// row `row_base` as their row 0.
void launch_window(sextans_engine *h, const float *dBp8, const float *dCin, int64_t ldc_in, float *dCout,
                   int64_t ldc, int ntiles, int wave_begin, int wave_end, int row_base, float alpha, float beta,
                   hipStream_t s) {
    const int nwg = (wave_end - wave_begin + sx::kWinWaves - 1) / sx::kWinWaves;
    if (nwg <= 0) return;
    const size_t lds = (size_t)sx::kWinWaves * (size_t)(h->win_rw + 1) * sx::kWinNT * sizeof(float);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg * (unsigned)ntiles), dim3(sx::kWinWaves * 64), lds, s,
                           (const sx::u32x2 *)h->d_wstream, (const int *)h->d_wstep0, dBp8, (int64_t)h->K * sx::kWinNT, dCin,
                           ldc_in, dCout, ldc, h->M, h->win_rw, wave_begin, wave_end, nwg, row_base, alpha, beta,
                           (const unsigned char *)h->d_skip);
    };
    if (h->opt_win_unroll == 4) { if (h->opt_exact) go(sx::spmm_csr_window<true, 4>); else go(sx::spmm_csr_window<false, 4>); }
    else                        { if (h->opt_exact) go(sx::spmm_csr_window<true, 8>); else go(sx::spmm_csr_window<false, 8>); }
}

}  // namespace

extern "C" {

const char *sextans_last_error(void) { return g_last_error.c_str(); }

int sextans_device_count(int *count) {
    if (!count) return SEXTANS_ERR_INVALID;
    *count = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SEXTANS_ERR_NO_DEVICE;
    int ok = 0;
    for (int d = 0; d < n; ++d)
        if (check_device(d) == SEXTANS_OK) ++ok;
    *count = ok;
    return ok > 0 ? SEXTANS_OK : SEXTANS_ERR_NO_DEVICE;
}

int sextans_create(sextans_handle_t *out, int device) {
    if (!out) return SEXTANS_ERR_INVALID;
    *out = nullptr;
    if (int rc = check_device(device)) return rc;
    SX_HIP(hipSetDevice(device));
    auto *h = new sextans_engine();
    h->device = device;
    {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, device) == hipSuccess && p.multiProcessorCount > 0)
            h->num_cus = p.multiProcessorCount;
    }
    {   // load the device code now (the runtime does it lazily at the first launch, ~1 ms, which would
        // otherwise land inside the first timed SpMM)
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&sx::repack_b_panels<16>));
        (void)hipGetLastError();
    }
    *out = h;
    return SEXTANS_OK;
}

int sextans_destroy(sextans_handle_t h) {
    if (!h) return SEXTANS_ERR_INVALID;
    (void)hipSetDevice(h->device);
    free_matrix(h);
    free_bell(h);
    (void)hipFree(h->d_bell_Bf);
    (void)hipFree(h->d_Bp); (void)hipFree(h->d_B); (void)hipFree(h->d_Cin); (void)hipFree(h->d_Cout);
    sextans_profile_reset(h);
    (void)hipFree(h->d_P);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    (void)hipFree(h->d_dbg);
    (void)hipFree(h->d_chB); (void)hipFree(h->d_chC);
    (void)hipFree(h->d_stage);
    for (hipEvent_t e : h->dist_events) (void)hipEventDestroy(e);
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    if (h->host_stream) (void)hipStreamDestroy(h->host_stream);
    delete h;
    return SEXTANS_OK;
}

static int64_t *option_slot(sextans_handle_t h, const char *key) {
    if (!strcmp(key, "kernel")) return &h->opt_kernel;
    if (!strcmp(key, "lanes_per_row")) return &h->opt_lpr;
    if (!strcmp(key, "stage_a")) return &h->opt_stage;
    if (!strcmp(key, "xcd_remap")) return &h->opt_xcd;
    if (!strcmp(key, "exact")) return &h->opt_exact;
    if (!strcmp(key, "profile")) return &h->opt_profile;
    if (!strcmp(key, "panel_min_reuse_x100")) return &h->opt_min_reuse_x100;
    if (!strcmp(key, "panel_min_reuse_wide_x100")) return &h->opt_min_reuse_wide_x100;
    if (!strcmp(key, "phase_timing")) return &h->opt_phase_timing;
    if (!strcmp(key, "split_rows")) return &h->opt_split_rows;
    if (!strcmp(key, "bucket_rows")) return &h->opt_bucket_rows;
    if (!strcmp(key, "global_nnz")) return &h->opt_global_nnz;
    if (!strcmp(key, "exact_chain")) return &h->opt_exact_chain;
    if (!strcmp(key, "fuse_b")) return &h->opt_fuse_b;
    if (!strcmp(key, "cols_per_lane")) return &h->opt_cols_per_lane;
    if (!strcmp(key, "tiles_per_wg")) return &h->opt_tiles_per_wg;
    if (!strcmp(key, "panel_v2")) return &h->opt_panel_v2;
    if (!strcmp(key, "small_v2")) return &h->opt_small_v2;
    if (!strcmp(key, "row_cluster")) return &h->opt_row_cluster;
    if (!strcmp(key, "cluster_shape")) return &h->opt_cluster_shape;
    if (!strcmp(key, "cluster_group")) return &h->opt_cluster_group;
    if (!strcmp(key, "window_rows")) return &h->opt_win_rows;
    if (!strcmp(key, "window_cols")) return &h->opt_win_cols;
    if (!strcmp(key, "window_unroll")) return &h->opt_win_unroll;
    if (!strcmp(key, "window_auto")) return &h->opt_win_auto;
    if (!strcmp(key, "bell_wide")) return &h->opt_bell_wide;
    if (!strcmp(key, "bell_generation")) return &h->opt_bell_gen;
    if (!strcmp(key, "bell_shared")) return &h->opt_bell_shared;
    if (!strcmp(key, "bell_debug")) return &h->opt_bell_debug;
    if (!strcmp(key, "mfma_dense_tiles")) return &h->opt_mfma_dense;
    if (!strcmp(key, "dense_tile_fill_x100")) return &h->opt_dense_fill_x100;
    return nullptr;
}

int sextans_set_option(sextans_handle_t h, const char *key, int64_t value) {
    if (!h || !key) return SEXTANS_ERR_INVALID;
    int64_t *slot = option_slot(h, key);
    if (!slot) return SEXTANS_ERR_INVALID;
    if (slot == &h->opt_lpr && value != 0 && value != 2 && value != 4 && value != 8) return SEXTANS_ERR_INVALID;
    if (slot == &h->opt_win_rows && (value < 1 || value > sx::kWinMaxRowsPerWave)) return SEXTANS_ERR_INVALID;
    if (slot == &h->opt_win_cols && (value < 1 || value > 0x7fffffff)) return SEXTANS_ERR_INVALID;
    if (slot == &h->opt_win_unroll && value != 4 && value != 8) return SEXTANS_ERR_INVALID;
    if (slot == &h->opt_cols_per_lane && value != 0 && value != 4 && value != 8) return SEXTANS_ERR_INVALID;
    if (slot == &h->opt_tiles_per_wg && (value < 0 || value > 1024)) return SEXTANS_ERR_INVALID;
    if ((slot == &h->opt_win_rows || slot == &h->opt_win_cols) && *slot != value) {
        (void)hipSetDevice(h->device);
        free_window(h);   // the stream is built for one (rows per wavefront, window) pair
    }
    if ((slot == &h->opt_row_cluster || slot == &h->opt_cluster_shape || slot == &h->opt_cluster_group || slot == &h->opt_min_reuse_x100 || slot == &h->opt_min_reuse_wide_x100) && *slot != value) {
        (void)hipSetDevice(h->device);   // the clustered-order plan is (re)considered under the new setting
        free_panel_state(h->psc);
        (void)hipFree(h->d_slot_row);
        h->d_slot_row = nullptr;
        h->cluster_state = 0;
    }
    if (*slot != value) h->dist_cut_key.clear();   // chunk cuts are aligned to the packed forms the options select (all
                                                   // ranks of a partition must change options together: the cut
                                                   // exchange is a collective)
    *slot = value;
    if (slot == &h->opt_phase_timing) {
        (void)hipSetDevice(h->device);
        if (value && !h->d_dbg && hipMalloc((void **)&h->d_dbg, 64) != hipSuccess) return SEXTANS_ERR_HIP;
        if (h->d_dbg) (void)hipMemset(h->d_dbg, 0, 64);
        if (!value && h->d_dbg) { (void)hipFree(h->d_dbg); h->d_dbg = nullptr; }
    }
    return SEXTANS_OK;
}

int sextans_phase_timing_read(sextans_handle_t h, int64_t out[8]) {
    if (!h || !out || !h->d_dbg) return SEXTANS_ERR_STATE;
    SX_HIP(hipSetDevice(h->device));
    SX_HIP(hipDeviceSynchronize());
    SX_HIP(hipMemcpy(out, h->d_dbg, 64, hipMemcpyDeviceToHost));
    return SEXTANS_OK;
}

int sextans_get_option(sextans_handle_t h, const char *key, int64_t *value) {
    if (!h || !key || !value) return SEXTANS_ERR_INVALID;
    int64_t *slot = option_slot(h, key);
    if (!slot) return SEXTANS_ERR_INVALID;
    *value = *slot;
    return SEXTANS_OK;
}

int sextans_set_matrix_csr(sextans_handle_t h, int M, int K, int64_t nnz, const int *row_ptr,
                           const int *col_idx, const float *val) {
    if (!h || M < 0 || K < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!col_idx || !val)))
        return SEXTANS_ERR_INVALID;
    if (nnz > 0x7fffffffLL) return SEXTANS_ERR_INVALID;   // 32-bit row_ptr like the reference
    if (row_ptr[0] != 0 || row_ptr[M] != (int)nnz) return SEXTANS_ERR_INVALID;
    // the kernels gather B rows by column index and the plan builders index host arrays of size K with them
    for (int r = 0; r < M; ++r)
        if (row_ptr[r + 1] < row_ptr[r]) return SEXTANS_ERR_INVALID;
    {
        unsigned bad = 0;
        for (int64_t j = 0; j < nnz; ++j) bad |= (unsigned)((unsigned)col_idx[j] >= (unsigned)K);
        if (bad) return SEXTANS_ERR_INDEX;
    }
    SX_HIP(hipSetDevice(h->device));
    free_matrix(h);
    int *rp = nullptr, *ci = nullptr;
    float *v = nullptr;
    SX_HIP(hipMalloc((void **)&rp, sizeof(int) * ((size_t)M + 1)));
    SX_HIP(hipMalloc((void **)&ci, sizeof(int) * (size_t)(nnz ? nnz : 1)));
    SX_HIP(hipMalloc((void **)&v, sizeof(float) * (size_t)(nnz ? nnz : 1)));
    SX_HIP(hipMemcpy(rp, row_ptr, sizeof(int) * ((size_t)M + 1), hipMemcpyHostToDevice));
    if (nnz) {
        SX_HIP(hipMemcpy(ci, col_idx, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        SX_HIP(hipMemcpy(v, val, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice));
    }
    h->d_rp = rp; h->d_ci = ci; h->d_v = v;
    h->owns_matrix = true;
    h->M = M; h->K = K; h->nnz = nnz;
    free_dense(h);   // source = main = the matrix itself until the dense-tile and long-row tests have run
    return SEXTANS_OK;
}

int sextans_set_matrix_csr_device(sextans_handle_t h, int M, int K, int64_t nnz,
                                  const int *d_row_ptr, const int *d_col_idx, const float *d_val) {
    if (!h || M < 0 || K < 0 || nnz < 0 || !d_row_ptr) return SEXTANS_ERR_INVALID;
    if (nnz > 0x7fffffffLL) return SEXTANS_ERR_INVALID;
    SX_HIP(hipSetDevice(h->device));
    free_matrix(h);
    h->d_rp = d_row_ptr; h->d_ci = d_col_idx; h->d_v = d_val;
    h->owns_matrix = false;
    h->M = M; h->K = K; h->nnz = nnz;
    free_dense(h);
    return SEXTANS_OK;
}

int sextans_spmm_device(sextans_handle_t h, int N, float alpha, const float *d_B, int64_t ldb,
                        float beta, const float *d_C_in, float *d_C_out, int64_t ldc,
                        void *stream) {
    return sextans_spmm_device2(h, N, alpha, d_B, ldb, beta, d_C_in, ldc, d_C_out, ldc, stream);
}

namespace {
struct Seg { int width, col0, ntiles; };

// Long-row test + piece tables (see the engine struct).  Thresholds:
//   L0 ("bucket_rows"; -1 = max(32, 2 * mean row length)): a workgroup of the row-group / panel kernels owns 32-128
//     consecutive rows and lives as long as its longest row, so one 100-entry row among 15-entry rows wastes 85 % of
//     the workgroup; rows above L0 are processed in a second launch in order of length instead.  Regular matrices
//     (Poisson, FEM, nasa4704) have no such rows and take none of this path.
//   T ("split_rows"; -1 = max(1024, nnz / 16384); 0 = never): the adds of one row are a serial chain and its B
//     rows arrive at best ~16 per memory round trip, i.e. ~0.05-0.1 us per entry: a 400 000-entry hub row would hold
//     one row group for tens of milliseconds.  Rows above T are cut into pieces of T entries that are summed in
//     parallel and folded in order (re-associated).  Measured on a 1M-row power-law matrix (33 M nnz, longest row
//     399 302): T = 512 / 1024 / 2021 -> 0.81 / 0.74 / 0.77 ms with 4964 / 2190 / 978 rows re-associated (uniform
//     matrix of the same size: 0.64 ms), so the larger threshold costs nothing and touches fewer rows.

int ensure_split(sextans_engine *h) {
    if (h->split_built_opt == h->opt_split_rows && h->bucket_built_opt == h->opt_bucket_rows &&
        h->split_built_gnnz == h->opt_global_nnz && h->chain_built_opt == h->opt_exact_chain)
        return SEXTANS_OK;
    free_split(h);
    free_plan(h);      // the packed forms are built from the main matrix
    free_window(h);
    h->split_built_opt = h->opt_split_rows;
    h->bucket_built_opt = h->opt_bucket_rows;
    h->split_built_gnnz = h->opt_global_nnz;
    h->chain_built_opt = h->opt_exact_chain;
    if (h->M == 0 || h->s_nnz == 0) return SEXTANS_OK;
    int64_t T = h->opt_split_rows, L0 = h->opt_bucket_rows;
    // strict order: rows above the automatic threshold become exact chains instead of one-piece rows
    const int64_t Tc = (h->opt_split_rows == 0 && h->opt_exact_chain)
                           ? std::max<int64_t>(1024, std::max<int64_t>(h->opt_global_nnz, h->s_nnz) / 16384) : INT64_MAX;
    // the automatic threshold follows the non-zeros of the whole matrix: a rank of a row-partitioned SpMM ("global_nnz")
    // then cuts a hub row into the same pieces as a single GPU holding all rows => bitwise equal results
    if (T < 0) T = std::max<int64_t>(1024, std::max<int64_t>(h->opt_global_nnz, h->s_nnz) / 16384);
    if (L0 < 0) L0 = std::max<int64_t>(32, 2 * (h->s_nnz / h->M));
    if (T == 0) T = INT64_MAX;                 // never split
    if (L0 == 0) L0 = std::min(T, Tc);         // no bucketing: only rows that must be split / chained leave
    if (L0 > std::min(T, Tc)) L0 = std::min(T, Tc);
    if (L0 == INT64_MAX) return SEXTANS_OK;
    PlanTimer timer(h);
    std::vector<int> rp;
    if (int rc = read_back_row_ptr(h, rp, 1)) return rc;
    std::vector<int> rows;                     // ascending
    for (int r = 0; r < h->M; ++r)
        if ((int64_t)rp[(size_t)r + 1] - rp[(size_t)r] > L0) rows.push_back(r);
    if (rows.empty()) return SEXTANS_OK;
    {   // bucketing alone (no row that must be split) is only worth three extra launches when the long rows carry
        // a visible share of the work: a handful of rows just above L0 in a regular matrix stay where they are
        int64_t long_nnz = 0, longest = 0;
        for (int r : rows) {
            const int64_t len = (int64_t)rp[(size_t)r + 1] - rp[(size_t)r];
            long_nnz += len;
            longest = std::max(longest, len);
        }
        if (longest <= std::min(T, Tc) && h->opt_bucket_rows < 0 && long_nnz * 50 < h->s_nnz) return SEXTANS_OK;
    }
    // chain rows leave the piece tables
    std::vector<int> chain_rows, piece_rows;
    for (int r : rows) ((int64_t)rp[(size_t)r + 1] - rp[(size_t)r] > Tc ? chain_rows : piece_rows).push_back(r);
    if (!chain_rows.empty()) {
        std::vector<int> beg;
        std::vector<long long> off(1, 0);
        for (int r : chain_rows) {
            const long long len = rp[(size_t)r + 1] - rp[(size_t)r];
            beg.push_back(rp[(size_t)r]);
            off.push_back(off.back() + len);
        }
        if (int rc = upload(&h->d_chain_row, chain_rows)) return rc;
        if (int rc = upload(&h->d_chain_beg, beg)) return rc;
        if (int rc = upload(&h->d_chain_off, off)) return rc;
        {   // launch order of whole-matrix calls: longest chain first (a workgroup lives as long as its row is; one per CU)
            std::vector<int> perm(chain_rows.size());
            for (size_t i = 0; i < perm.size(); ++i) perm[i] = (int)i;
            std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return off[(size_t)a + 1] - off[(size_t)a] > off[(size_t)b + 1] - off[(size_t)b]; });
            if (int rc = upload(&h->d_chain_perm, perm)) return rc;
        }
        h->h_chain_row = chain_rows;
        h->h_chain_off = off;
        h->nchain = (int)chain_rows.size();
        h->chain_T = Tc;
    }
    std::vector<int> ci;
    std::vector<float> va;
    if (int rc = read_back_entries(h, ci, va, 1)) return rc;
    // piece tables in two row orders
    auto build = [&](const std::vector<int> &order, sextans_engine::PieceTable &t) -> int {
        std::vector<int> vrp, vend, vfirst;
        for (int r : order) {
            vfirst.push_back((int)vrp.size());
            const int64_t j0 = rp[(size_t)r], j1 = rp[(size_t)r + 1];
            const int64_t step = (j1 - j0 > T) ? T : (j1 - j0);
            for (int64_t j = j0; j < j1; j += step) { vrp.push_back((int)j); vend.push_back((int)std::min(j + step, j1)); }
        }
        vfirst.push_back((int)vrp.size());
        if (int rc = upload(&t.d_vrp, vrp)) return rc;
        if (int rc = upload(&t.d_vend, vend)) return rc;
        if (int rc = upload(&t.d_vfirst, vfirst)) return rc;
        if (int rc = upload(&t.d_row, order)) return rc;
        t.h_row = order;
        t.h_vfirst = vfirst;
        return SEXTANS_OK;
    };
    std::vector<int> by_len = piece_rows;
    std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int b) {
        return rp[(size_t)a + 1] - rp[(size_t)a] > rp[(size_t)b + 1] - rp[(size_t)b];
    });
    if (int rc = build(by_len, h->by_len)) return rc;
    if (int rc = build(piece_rows, h->by_row)) return rc;
    // main matrix: long rows emptied; skip flags
    std::vector<int> mrp((size_t)h->M + 1, 0);
    std::vector<unsigned char> skip((size_t)h->M, 0);
    {
        size_t k = 0, w = 0;
        for (int r = 0; r < h->M; ++r) {
            const int j0 = rp[(size_t)r], j1 = rp[(size_t)r + 1];
            if (k < rows.size() && rows[k] == r) {
                skip[(size_t)r] = 1;
                if ((int64_t)j1 - j0 > T) h->h_split_rows.push_back(r);
                ++k;
            } else {
                if (w != (size_t)j0) {
                    std::copy(ci.begin() + j0, ci.begin() + j1, ci.begin() + (ptrdiff_t)w);
                    std::copy(va.begin() + j0, va.begin() + j1, va.begin() + (ptrdiff_t)w);
                }
                w += (size_t)(j1 - j0);
            }
            mrp[(size_t)r + 1] = (int)w;
        }
        ci.resize(w ? w : 1); va.resize(w ? w : 1);
        h->m_nnz = (int64_t)w;
    }
    if (int rc = upload(&h->d_mrp, mrp)) return rc;
    if (int rc = upload(&h->d_mci, ci)) return rc;
    if (int rc = upload(&h->d_mv, va)) return rc;
    if (int rc = upload(&h->d_skip, skip)) return rc;
    h->m_rp = h->d_mrp; h->m_ci = h->d_mci; h->m_v = h->d_mv;
    h->nhub = (int)piece_rows.size();
    h->split_nv = h->by_len.h_vfirst.back();
    h->split_T = T == INT64_MAX ? 0 : T;
    h->bucket_L0 = L0;
    return SEXTANS_OK;
}

// "MFMA only where a tile is actually dense" (north_star).  Counts the 32x32 tiles of the main matrix whose fill
// reaches the threshold (always: get_stat "dense_tile_fraction" = share of the non-zeros sitting in such tiles) and,
// when the caller has opted into bf16 for them ("mfma_dense_tiles" = 1), cuts them out of the main matrix into a
// blocked-ELL bf16 side matrix for spmm_bell_mfma; the CSR kernels keep the remainder in fp32.  Only full 32-row
// block rows are searched; at most 256 dense tiles per block row (the densest columns first come first served).
int ensure_dense(sextans_engine *h) {
    if (h->dense_built_mfma == h->opt_mfma_dense && h->dense_built_fill == h->opt_dense_fill_x100) return SEXTANS_OK;
    if (h->dense_W > 0 || h->opt_mfma_dense) {   // the source matrix may change: everything downstream starts again
        free_plan(h);
        free_window(h);
    }
    const bool had_tiles = h->dense_W > 0;
    if (had_tiles || h->opt_mfma_dense) free_dense(h);
    else { h->dense_tiles = h->dense_nnz = 0; }
    h->dense_built_mfma = h->opt_mfma_dense;
    h->dense_built_fill = h->opt_dense_fill_x100;
    const int mb = h->M / 32;
    if (mb == 0 || h->nnz == 0) return SEXTANS_OK;
    const int64_t thr = std::max<int64_t>(1, (h->opt_dense_fill_x100 * 1024 + 99) / 100);
    if (h->nnz < thr) return SEXTANS_OK;
    PlanTimer timer(h);
    std::vector<int> rp, ci;
    std::vector<float> va;
    if (int rc = read_back_row_ptr(h, rp, 0)) return rc;
    {   // cheap exit: a block row with fewer than `thr` entries cannot hold a dense tile
        bool any = false;
        for (int br = 0; br < mb && !any; ++br) any = (int64_t)rp[(size_t)br * 32 + 32] - rp[(size_t)br * 32] >= thr;
        if (!any) return SEXTANS_OK;
    }
    if (!h->opt_mfma_dense) {
        // report only: estimate from a sample of block rows (a few small copies instead of reading the matrix back)
        const int nsample = std::min(mb, 512);
        int64_t tot = 0, in_dense = 0, tiles = 0;
        std::vector<int> cols;
        for (int sidx = 0; sidx < nsample; ++sidx) {
            const int br = (int)((int64_t)sidx * mb / nsample);
            const int j0 = rp[(size_t)br * 32], j1 = rp[(size_t)br * 32 + 32];
            tot += j1 - j0;
            if (j1 - j0 < thr) continue;
            cols.resize((size_t)(j1 - j0));
            SX_HIP(hipMemcpy(cols.data(), h->d_ci + j0, sizeof(int) * cols.size(), hipMemcpyDeviceToHost));
            for (int &c : cols) {
                if ((unsigned)c >= (unsigned)h->K) return SEXTANS_ERR_INDEX;
                c >>= 5;
            }
            std::sort(cols.begin(), cols.end());
            for (size_t a = 0; a < cols.size();) {
                size_t b = a;
                while (b < cols.size() && cols[b] == cols[a]) ++b;
                if ((int64_t)(b - a) >= thr) { in_dense += (int64_t)(b - a); ++tiles; }
                a = b;
            }
        }
        // scaled to the whole matrix
        h->dense_nnz = tot ? (int64_t)((double)in_dense / (double)tot * (double)h->nnz) : 0;
        h->dense_tiles = (int64_t)((double)tiles * (double)mb / (double)nsample);
        return SEXTANS_OK;
    }
    if (int rc = read_back_entries(h, ci, va, 0)) return rc;
    // pass 1: dense tile columns per block row
    std::vector<std::vector<int>> dense((size_t)mb);
    std::vector<int> cols;
    int W = 0;
    for (int br = 0; br < mb; ++br) {
        const int j0 = rp[(size_t)br * 32], j1 = rp[(size_t)br * 32 + 32];
        if (j1 - j0 < thr) continue;
        cols.assign(ci.begin() + j0, ci.begin() + j1);
        for (int &c : cols) c >>= 5;
        std::sort(cols.begin(), cols.end());
        for (size_t a = 0; a < cols.size();) {
            size_t b = a;
            while (b < cols.size() && cols[b] == cols[a]) ++b;
            if ((int64_t)(b - a) >= thr && dense[(size_t)br].size() < 256) {
                dense[(size_t)br].push_back(cols[a]);
                h->dense_nnz += (int64_t)(b - a);
            }
            a = b;
        }
        h->dense_tiles += (int64_t)dense[(size_t)br].size();
        W = std::max(W, (int)dense[(size_t)br].size());
    }
    if (W == 0) return SEXTANS_OK;   // nothing to route
    // The device form is blocked-ELL (mb x W slots of 2 KiB): one crowded block row sets W for all of them, so bound
    // the padded size; a matrix that would need more keeps its dense tiles on the fp32 kernels (reported, not routed).
    if ((int64_t)mb * W * 2048 > ((int64_t)8 << 30)) {
        g_last_error = "mfma_dense_tiles: blocked-ELL form of the dense tiles would exceed 8 GiB; tiles stay on the fp32 kernels";
        return SEXTANS_OK;
    }
    // pass 2: tile values, stored COMPACTLY on the host (one 32x32 fp32 tile per dense tile, not per ELL slot: fp32 sums
    // of duplicates, rounded to bf16 once) + the remainder as the new main matrix
    std::vector<int64_t> tile0((size_t)mb + 1, 0);   // first compact tile of every block row
    for (int br = 0; br < mb; ++br) tile0[(size_t)br + 1] = tile0[(size_t)br] + (int64_t)dense[(size_t)br].size();
    std::vector<int> bcol((size_t)mb * W, -1);
    std::vector<float> blk((size_t)tile0[(size_t)mb] * 1024, 0.0f);
    std::vector<int> mrp((size_t)h->M + 1, 0);
    size_t w = 0;
    for (int r = 0; r < h->M; ++r) {
        const int br = r >> 5;
        const std::vector<int> *d = br < mb ? &dense[(size_t)br] : nullptr;
        for (int j = rp[(size_t)r]; j < rp[(size_t)r + 1]; ++j) {
            int slot = -1;
            if (d && !d->empty()) {
                const auto it = std::lower_bound(d->begin(), d->end(), ci[(size_t)j] >> 5);
                if (it != d->end() && *it == (ci[(size_t)j] >> 5)) slot = (int)(it - d->begin());
            }
            if (slot >= 0) {
                blk[(((size_t)tile0[(size_t)br] + (size_t)slot) * 32 + (size_t)(r & 31)) * 32 + (size_t)(ci[(size_t)j] & 31)] += va[(size_t)j];
            } else {
                ci[w] = ci[(size_t)j]; va[w] = va[(size_t)j]; ++w;
            }
        }
        mrp[(size_t)r + 1] = (int)w;
    }
    for (int br = 0; br < mb; ++br)
        for (size_t sl = 0; sl < dense[(size_t)br].size(); ++sl) bcol[(size_t)br * W + sl] = dense[(size_t)br][sl];
    std::vector<uint16_t> bval((size_t)mb * W * 1024, 0);   // ELL slots without a tile stay +0.0
    for (int br = 0; br < mb; ++br)
        for (size_t sl = 0; sl < dense[(size_t)br].size(); ++sl) {
            const float *src = blk.data() + ((size_t)tile0[(size_t)br] + sl) * 1024;
            uint16_t *dst = bval.data() + ((size_t)br * W + sl) * 1024;
            for (int i = 0; i < 1024; ++i) {
                uint32_t u;
                memcpy(&u, &src[i], 4);
                if ((u & 0x7fffffffu) > 0x7f800000u) { dst[i] = (uint16_t)((u >> 16) | 0x40u); continue; }
                u += 0x7fffu + ((u >> 16) & 1u);
                dst[i] = (uint16_t)(u >> 16);
            }
        }
    std::vector<float>().swap(blk);
    ci.resize(w ? w : 1); va.resize(w ? w : 1);
    // the remainder becomes the source matrix of the long-row split (which has not run yet for this source)
    if (int rc = upload(&h->d_srp, mrp)) return rc;
    if (int rc = upload(&h->d_sci, ci)) return rc;
    if (int rc = upload(&h->d_sv, va)) return rc;
    h->s_rp = h->d_srp; h->s_ci = h->d_sci; h->s_v = h->d_sv; h->s_nnz = (int64_t)w;
    h->m_rp = h->s_rp; h->m_ci = h->s_ci; h->m_v = h->s_v; h->m_nnz = h->s_nnz;
    if (int rc = upload(&h->d_dense_col, bcol)) return rc;
    uint16_t *d_val = nullptr;
    SX_HIP(hipMalloc((void **)&d_val, bval.size() * 2));
    SX_HIP(hipMemcpy(d_val, bval.data(), bval.size() * 2, hipMemcpyHostToDevice));
    const int64_t nslots = (int64_t)mb * W;
    SX_HIP(hipMalloc(&h->d_dense_Af, (size_t)nslots * 2048));
    hipLaunchKernelGGL(sx::bell_repack_a, dim3((unsigned)((nslots * 128 + 255) / 256)), dim3(256), 0, nullptr, d_val,
                       (sx::u32x4 *)h->d_dense_Af, nslots);
    SX_HIP(hipDeviceSynchronize());
    (void)hipFree(d_val);
    h->dense_mb = mb;
    h->dense_W = W;
    {   // do neighbouring block rows share tile columns (block-diagonal / banded dense structure)?  Then N = 256 runs the
        // LDS-shared MFMA kernel
        unsigned long long *d_cnt = nullptr, h_cnt[3] = {0, 0, 0};
        SX_HIP(hipMalloc((void **)&d_cnt, 3 * sizeof(unsigned long long)));
        SX_HIP(hipMemset(d_cnt, 0, 3 * sizeof(unsigned long long)));
        const int groups = (mb + sx::kShRows - 1) / sx::kShRows;
        hipLaunchKernelGGL(sx::bell_union_count, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, nullptr, h->d_dense_col, mb, W,
                           d_cnt, d_cnt + 1, d_cnt + 2);
        const hipError_t e = hipMemcpy(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost);
        (void)hipFree(d_cnt);
        SX_HIP(e);
        h->dense_share = h_cnt[0] ? (double)h_cnt[1] / (double)h_cnt[0] : 0.0;
        h->dense_max_union = (int)h_cnt[2];
    }
    return SEXTANS_OK;
}

// Everything that may allocate or run host-side preprocessing for an N-column SpMM: B-panel workspace,
// N-tile plan, and (for kernel != 1) the packed row-bucketed form of A.  Idempotent; called by
// sextans_spmm_device2 and, ahead of the timed region, by sextans_spmm_host.
int prepare(sextans_engine *h, int N, std::vector<Seg> &plan, int &W, bool &use_panel, bool &use_window) {
    if (int rc = ensure_dense(h)) return rc;   // first the dense tiles leave (when the caller routes them to MFMA) ...
    if (int rc = ensure_split(h)) return rc;   // ... then the long rows; the packed forms below are built from what remains
    if (h->dense_W > 0 && N % 32 == 0) {
        const size_t need = (size_t)((h->K + 31) / 32) * 32 * (size_t)N * 2;
        if (h->bell_Bf_cap < need) {
            if (h->d_bell_Bf) SX_HIP(hipFree(h->d_bell_Bf));
            h->d_bell_Bf = nullptr; h->bell_Bf_cap = 0;
            SX_HIP(hipMalloc(&h->d_bell_Bf, need));
            h->bell_Bf_cap = need;
        }
    }
    if (h->nhub > 0)
        if (int rc = ensure(&h->d_P, &h->P_cap, (size_t)h->split_nv * (size_t)N)) return rc;
    if (h->nchain > 0) {
        if (!h->aux_stream) {
            SX_HIP(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));   // (a high-priority stream was measured: no difference)
            SX_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            SX_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
    }
    if (h->Bp_cap < (size_t)h->K * (size_t)N || !h->d_Bp) h->bp_layout = 0;   // new workspace: nothing to reuse
    if (int rc = ensure(&h->d_Bp, &h->Bp_cap, (size_t)h->K * (size_t)N)) return rc;
    // main tile width: 4*lanes_per_row, but never wider than N itself (N = 8 -> 2 lanes per row); then
    // 16- and 8-wide tiles for the remainder (N is a multiple of 8, the reference's N-tile
    // granularity: sextans.cpp:57-60).
    // "lanes_per_row" 0 = auto: 4 lanes (16-column tiles) for the panel kernel -- measured best on the FEM class
    // (config 3, N=128: 23 us with 4 lanes, 31 us with 8) -- and 8 lanes (32-column tiles) for the gather kernel
    // once N >= 32: a 128-byte B row is one fabric request where two 64-byte tiles are two (uniform 4M matrix,
    // N = 32/64/128: 3.15/6.9/15.2 ms with 8 lanes against 6.1/12.6/27.7 ms with 4).
    int lpr = h->opt_lpr ? (int)h->opt_lpr : 4;
    auto tiles = [&]() {
        while (lpr > 2 && 4 * lpr > N) lpr /= 2;
        W = 4 * lpr;
        plan.clear();
        int col = 0;
        for (int w : {W, 16, 8}) {
            if (w > W) continue;
            const int nt = (N - col) / w;
            if (nt > 0) { plan.push_back({w, col, nt}); col += nt * w; }
        }
    };
    tiles();
    // Kernel choice: "kernel" 1 = row-group gather, 2 = LDS panel, 0 = auto (panel when at least half
    // of the non-zeros sit in row blocks whose B rows are reused -- "only where a tile has reuse").
    use_panel = false;
    if (h->opt_kernel != 1 && h->m_nnz > 0) {
        if (int rc = ensure_plan(h, lpr, h->opt_kernel == 2)) return rc;
        // (here, not at launch time: prepare() runs before a hipGraph capture starts, and the builder copies to the host)
        if (lpr == 4 && h->ps.plan_built)
            if (int rc = ensure_cluster_plan(h)) return rc;
        use_panel = h->ps.plan_built && ((h->opt_kernel == 2) || (h->ps.plan_panel_frac >= 0.5 && (N >= 32 || h->ps.plan_narrow_frac >= 0.5)));
    }
    if (!h->opt_lpr && !use_panel && N >= 32 && lpr != 8) { lpr = 8; tiles(); }
    // "kernel" 3 = K-windowed accumulator-resident kernel; auto picks it for matrices without B-row reuse
    // whose B does not fit the L2s when the traffic model says the sweep moves fewer bytes than the gather.
    use_window = false;
    if (h->m_nnz > 0 && (h->opt_kernel == 3 || (h->opt_kernel == 0 && !use_panel))) {
        const bool force = h->opt_kernel == 3;
        if (force || (h->win_state >= 0 && window_pays(h, N, h->win_state == 1 ? h->win_padded : h->m_nnz))) {
            if (int rc = ensure_window(h, force)) return rc;
            use_window = h->win_state == 1 && (force || window_pays(h, N, h->win_padded));
        }
    }
    return SEXTANS_OK;
}
}  // namespace

int sextans_align_row(sextans_handle_t h, int N, int row, int *aligned) {
    if (!h || !aligned || N <= 0 || (N % 8) != 0 || row < 0 || row > h->M) return SEXTANS_ERR_INVALID;
    if (!h->d_rp) return SEXTANS_ERR_STATE;
    SX_HIP(hipSetDevice(h->device));
    std::vector<Seg> plan;
    int W = 0;
    bool use_panel = false, use_window = false;
    if (int rc = prepare(h, N, plan, W, use_panel, use_window)) return rc;
    *aligned = row;
    if (row == h->M) return SEXTANS_OK;
    if (use_window) *aligned = row / h->win_rw * h->win_rw;
    else if (use_panel && !h->ps.h_blk_row.empty())
        *aligned = *(std::upper_bound(h->ps.h_blk_row.begin(), h->ps.h_blk_row.end(), row) - 1);
    return SEXTANS_OK;
}

int sextans_spmm_device2(sextans_handle_t h, int N, float alpha, const float *d_B, int64_t ldb,
                         float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out, int64_t ldc,
                         void *stream) {
    if (!h) return SEXTANS_ERR_INVALID;
    return sextans_spmm_device_rows(h, N, alpha, d_B, ldb, beta, d_C_in, ldc_in, d_C_out, ldc, 0, h->M, 0,
                                    stream);
}

}  // extern "C"
namespace {
// Hub rows inside [row_begin, row_end): pieces summed as virtual rows by the row-group kernel from B panels of width
// 4 * LPR at dBp (ntiles panels), then folded in order into the C the main kernel has already written.
const char *kernel_name(int main, bool hubs, bool dense) {   // static strings for sextans_last_kernel
    static const char *names[4][2][2] = {
        {{"spmm_csr_rowgroup", "spmm_csr_rowgroup+dense_tiles_mfma"},
         {"spmm_csr_rowgroup+hub_pieces", "spmm_csr_rowgroup+hub_pieces+dense_tiles_mfma"}},
        {{"spmm_csr_panel", "spmm_csr_panel+dense_tiles_mfma"},
         {"spmm_csr_panel+hub_pieces", "spmm_csr_panel+hub_pieces+dense_tiles_mfma"}},
        {{"spmm_csr_window", "spmm_csr_window+dense_tiles_mfma"},
         {"spmm_csr_window+hub_pieces", "spmm_csr_window+hub_pieces+dense_tiles_mfma"}},
        {{"spmm_csr_panel_v2", "spmm_csr_panel_v2+dense_tiles_mfma"},
         {"spmm_csr_panel_v2+hub_pieces", "spmm_csr_panel_v2+hub_pieces+dense_tiles_mfma"}}};
    return names[main][hubs ? 1 : 0][dense ? 1 : 0];
}

// Exact chains of the chain rows [c0, c1) (chain_fused, spmm_csr_kernels.h): products from the repacked B panels (segment by
// segment, like the piece kernel) and the serial sum of every (row, column) in one workgroup, epilogue included.
void launch_chains(sextans_engine *h, const std::vector<Seg> &plan, const float *dCin, int64_t ldc_in, float *dCout, int64_t ldc,
                   int N, int c0, int c1, int row_base, float alpha, float beta, hipStream_t s) {
    // one workgroup per (chain row, 16- or 8-column tile): chain_fused
    {
        for (const Seg &g : plan) {
            const float *bp = h->d_Bp + (size_t)h->K * (size_t)g.col0;
            const int NT = g.width >= 16 ? 16 : 8;
            const int ntiles = g.ntiles * (g.width / NT);
            auto go = [&](auto kern, int lds, int threads) {
                (void)allow_big_lds(h, reinterpret_cast<const void *>(kern), lds);
                hipLaunchKernelGGL(kern, dim3((unsigned)(c1 - c0) * (unsigned)ntiles), dim3((unsigned)threads), (size_t)lds, s, h->d_chain_row,
                                   h->d_chain_beg, h->d_chain_off, (c0 == 0 && c1 == h->nchain) ? h->d_chain_perm : (const int *)nullptr, h->s_ci, h->s_v, bp, (int64_t)h->K * g.width, g.width, dCin, ldc_in, dCout,
                                   ldc, g.col0, ntiles, c0, row_base, alpha, beta);
            };
#define SX_FUSED(W) if (h->opt_exact) go(sx::chain_fused<W, true>, sx::chain_fused_lds_bytes(W), sx::chain_fused_threads(W)); \
                    else go(sx::chain_fused<W, false>, sx::chain_fused_lds_bytes(W), sx::chain_fused_threads(W))
            if (NT == 16) { SX_FUSED(16); } else { SX_FUSED(8); }
#undef SX_FUSED
        }
    }
}

template <int LPR>
void launch_hub_pieces(sextans_engine *h, const sextans_engine::PieceTable &t, const float *dBp, int ntiles, int col0,
                       int v0, int v1, hipStream_t s) {
    constexpr int RB = sx::kBlock / LPR;
    const int nblk = (v1 - v0 + RB - 1) / RB;
    if (nblk <= 0) return;
    float *P = h->d_P + (int64_t)col0 * h->split_nv;
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk * (unsigned)ntiles), dim3(sx::kBlock), 0, s, t.d_vrp, t.d_vend, h->s_ci,
                           h->s_v, dBp, (int64_t)h->K * 4 * LPR, P, (int64_t)h->split_nv, v0, v1, ntiles);
    };
    if (h->opt_exact) go(sx::spmm_csr_pieces<LPR, true>); else go(sx::spmm_csr_pieces<LPR, false>);
}
}  // namespace
extern "C" {

int sextans_get_stat(sextans_handle_t h, const char *key, double *value) {
    if (!h || !key || !value) return SEXTANS_ERR_INVALID;
    if (h->d_rp) {   // figures about the packed forms refer to the current options: bring the cheap ones up to date
        SX_HIP(hipSetDevice(h->device));
        if (int rc = ensure_dense(h)) return rc;
        if (int rc = ensure_split(h)) return rc;
    }
    if (!strcmp(key, "plan_build_s")) *value = h->plan_build_s;
    else if (!strcmp(key, "window_padded_entries")) *value = (double)h->win_padded;
    else if (!strcmp(key, "window_state")) *value = (double)h->win_state;
    else if (!strcmp(key, "reassociated_rows")) *value = (double)h->h_split_rows.size();
    else if (!strcmp(key, "piece_path_rows")) *value = (double)(h->nhub + h->nchain);
    else if (!strcmp(key, "exact_chain_rows")) *value = (double)h->nchain;
    else if (!strcmp(key, "chain_threshold")) *value = (double)h->chain_T;
    else if (!strcmp(key, "split_threshold")) *value = (double)h->split_T;
    else if (!strcmp(key, "bucket_threshold")) *value = (double)h->bucket_L0;
    else if (!strcmp(key, "dense_tiles")) *value = (double)h->dense_tiles;
    else if (!strcmp(key, "dense_tile_fraction")) *value = h->nnz > 0 ? (double)h->dense_nnz / (double)h->nnz : 0.0;
    else if (!strcmp(key, "dense_tiles_on_mfma")) *value = h->dense_W > 0 ? 1.0 : 0.0;
    else if (!strcmp(key, "bell_share")) *value = h->bell_share;
    else if (!strcmp(key, "row_cluster")) *value = (double)h->cluster_state;          // 1 in use, -1 rejected, 0 not evaluated yet
    else if (!strcmp(key, "grid_stride_line")) *value = (double)h->cluster_s2;
    else if (!strcmp(key, "grid_stride_plane")) *value = (double)h->cluster_s3;
    else if (!strcmp(key, "panel_rows_natural")) *value = (double)h->plan_total_dict;  // B rows copied into LDS per N tile, natural order
    else if (!strcmp(key, "panel_rows_clustered")) *value = (double)h->cluster_total_dict;
    else if (!strcmp(key, "panel_fraction")) *value = h->ps.plan_panel_frac;
    else if (!strcmp(key, "panel_blocks")) *value = (double)h->ps.plan_nblk;
    else return SEXTANS_ERR_INVALID;
    return SEXTANS_OK;
}

int sextans_export_plan(sextans_handle_t h, int lanes_per_row, sextans_packed *out) {
    if (!h || !out || (lanes_per_row != 2 && lanes_per_row != 4 && lanes_per_row != 8)) return SEXTANS_ERR_INVALID;
    if (!h->d_rp) return SEXTANS_ERR_STATE;
    SX_HIP(hipSetDevice(h->device));
    if (int rc = ensure_dense(h)) return rc;
    if (int rc = ensure_split(h)) return rc;
    if (int rc = ensure_plan(h, lanes_per_row, true)) return rc;
    if (!h->ps.plan_built) return SEXTANS_ERR_STATE;
    const auto &ps = h->ps;
    const int M = h->M, nblk = ps.plan_nblk, RB = sx::kBlock / lanes_per_row;
    const size_t L = (size_t)ps.plan_stream_len;
    memset(out, 0, sizeof *out);
    out->M = M; out->K = h->K; out->nnz = h->m_nnz; out->lanes_per_row = lanes_per_row; out->nblk = nblk;
    out->stream_len = (int64_t)L; out->max_dict = ps.plan_max_dict; out->nnz_in_panel_blocks = ps.plan_nnz_panel;
    std::vector<int> rp((size_t)M + 1), cnt((size_t)nblk), bd((size_t)nblk * ps.plan_dict_stride);
    SX_HIP(hipMemcpy(rp.data(), h->m_rp, sizeof(int) * rp.size(), hipMemcpyDeviceToHost));
    if (nblk) {
        SX_HIP(hipMemcpy(cnt.data(), ps.d_dict_ptr, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost));
        SX_HIP(hipMemcpy(bd.data(), ps.d_dict, sizeof(int) * bd.size(), hipMemcpyDeviceToHost));
    }
    auto alloc = [](size_t bytes) { return calloc(bytes ? bytes : 1, 1); };
    out->blk_row = (int *)alloc(sizeof(int) * ((size_t)nblk + 1));
    out->dict_ptr = (int *)alloc(sizeof(int) * ((size_t)nblk + 1));
    out->row_off = (int *)alloc(sizeof(int) * ((size_t)M + 1));
    out->idx16 = (uint16_t *)alloc(sizeof(uint16_t) * L);
    out->col32 = (int *)alloc(sizeof(int) * L);
    out->val = (float *)alloc(sizeof(float) * L);
    size_t ndict = 0;
    for (int b = 0; b < nblk; ++b) ndict += (size_t)cnt[(size_t)b];
    out->dict = (int *)alloc(sizeof(int) * ndict);
    if (!out->blk_row || !out->dict_ptr || !out->row_off || !out->idx16 || !out->col32 || !out->val || !out->dict) {
        sextans_packed_free(out);
        return SEXTANS_ERR_ALLOC;
    }
    memcpy(out->blk_row, ps.h_blk_row.data(), sizeof(int) * ((size_t)nblk + 1));
    size_t w = 0;
    for (int b = 0; b < nblk; ++b) {
        out->dict_ptr[b] = (int)w;
        for (int i = 0; i < cnt[(size_t)b]; ++i) out->dict[w++] = bd[(size_t)b * ps.plan_dict_stride + (size_t)i];
    }
    out->dict_ptr[nblk] = (int)w;
    for (int r = 0; r < M; ++r) out->row_off[r + 1] = out->row_off[r] + ((rp[(size_t)r + 1] - rp[(size_t)r] + 3) & ~3);
    SX_HIP(hipMemcpy(out->idx16, ps.d_lidx, sizeof(uint16_t) * L, hipMemcpyDeviceToHost));
    SX_HIP(hipMemcpy(out->val, ps.d_pval, sizeof(float) * L, hipMemcpyDeviceToHost));
    if (ps.plan_mixed) SX_HIP(hipMemcpy(out->col32, ps.d_pcol32, sizeof(int) * L, hipMemcpyDeviceToHost));
    // device stream: byte offset of the B row in the panel; public form: dictionary index, 0xFFFF in the padding
    const unsigned row_bytes = 16u * (unsigned)lanes_per_row, pad_off = (unsigned)ps.plan_pad_row * row_bytes;
    for (size_t i = 0; i < L; ++i) out->idx16[i] = out->idx16[i] == pad_off ? (uint16_t)0xFFFF : (uint16_t)(out->idx16[i] / row_bytes);
    (void)RB;
    return SEXTANS_OK;
}

int sextans_reassociated_rows(sextans_handle_t h, int *rows, int capacity, int *count) {
    if (!h || !count || capacity < 0 || (capacity > 0 && !rows)) return SEXTANS_ERR_INVALID;
    if (!h->d_rp) return SEXTANS_ERR_STATE;
    SX_HIP(hipSetDevice(h->device));
    if (int rc = ensure_dense(h)) return rc;
    if (int rc = ensure_split(h)) return rc;
    *count = (int)h->h_split_rows.size();
    for (int i = 0; i < *count && i < capacity; ++i) rows[i] = h->h_split_rows[(size_t)i];
    return SEXTANS_OK;
}

int sextans_spmm_device_rows(sextans_handle_t h, int N, float alpha, const float *d_B, int64_t ldb,
                             float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out, int64_t ldc,
                             int row_begin, int row_end, int flags, void *stream) {
    if (!h || N <= 0 || (N % 8) != 0 || !d_B || !d_C_in || !d_C_out) return SEXTANS_ERR_INVALID;
    if (!h->d_rp) return SEXTANS_ERR_STATE;
    if (row_begin < 0 || row_end < row_begin || row_end > h->M) return SEXTANS_ERR_INVALID;
    const bool whole = row_begin == 0 && row_end == h->M;
    const int nrows = row_end - row_begin;
    if (ldb < h->K || ldc < nrows || ldc_in < nrows) return SEXTANS_ERR_INVALID;
    SX_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (nrows == 0) {
        // an empty range still opens a sequence: without the reuse flag the caller announces a new B, and the next call
        // (which will carry the flag) must not find the panels of some earlier B
        if (!(flags & SEXTANS_ROWS_REUSE_B_PANELS)) h->bp_layout = 0;
        return SEXTANS_OK;
    }
    // A call without the reuse flag announces a new B: whatever panels the workspace holds are stale from here on,
    // also when THIS call does not repack (column-major staging, fuse_b) -- a later chunk of the same pipelined SpMM
    // that does need panels must not find those of an earlier B.
    if (!(flags & SEXTANS_ROWS_REUSE_B_PANELS)) h->bp_layout = 0;
    std::vector<Seg> plan;
    int W = 0;
    bool use_panel = false, use_window = false;
    if (int rc = prepare(h, N, plan, W, use_panel, use_window)) return rc;
    if (h->dense_W > 0) {
        // Dense tiles first, on the matrix cores: C_out = alpha * (A_dense * bf16(B)) + beta * C_in for the full block
        // rows (and alpha * 0 + beta * C_in below them); the CSR kernels then add alpha * (A_rest * B) on top
        // (their beta becomes 1, their C_in the partial result): one C pass per kernel, no extra combine launch.
        if (!whole || (N % 32) != 0) {
            g_last_error = "mfma_dense_tiles = 1 needs whole-matrix calls and N % 32 == 0";
            return SEXTANS_ERR_INVALID;
        }
        const int kblocks = (h->K + 31) / 32, ntiles = N / 32;
        const int64_t threads = (int64_t)kblocks * ntiles * 128;
        hipLaunchKernelGGL(sx::bell_repack_b_f32, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_B, ldb, h->K,
                           (sx::u32x4 *)h->d_bell_Bf, kblocks, ntiles);
        const auto *Af = (const sx::bf16x8 *)h->d_dense_Af;
        const auto *Bf = (const sx::bf16x8 *)h->d_bell_Bf;
#define SX_BELL(NSUB)                                                                                               \
    {                                                                                                               \
        const int64_t waves = (int64_t)h->dense_mb * (ntiles / NSUB);                                               \
        hipLaunchKernelGGL((sx::spmm_bell_mfma<NSUB>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, h->d_dense_col, \
                           Af, Bf, d_C_in, ldc_in, d_C_out, ldc, h->dense_mb, h->dense_W, ntiles, alpha, beta);            \
    }
        const bool shared = ntiles == 8 && h->opt_bell_shared != 0 && sx::kShRows * h->dense_W <= sx::kShMaxRowCols &&
                            h->dense_max_union <= sx::kShMaxUnion && (h->opt_bell_shared == 1 || h->dense_share >= 1.5);
        if (shared) {
            constexpr size_t lds = (size_t)sx::kShRing * sx::kShTileBytes + (size_t)(sx::kShMaxUnion + 8) * (sizeof(int) + sx::kShRows * sizeof(short)) +
                                   (size_t)sx::kShMaxRowCols * sizeof(int);
            if (int rc = allow_big_lds(h, reinterpret_cast<const void *>(sx::spmm_bell_mfma_shared), (int)lds)) return rc;
            hipLaunchKernelGGL(sx::spmm_bell_mfma_shared, dim3((unsigned)((h->dense_mb + sx::kShRows - 1) / sx::kShRows)),
                               dim3(sx::kShThreads), lds, s, h->d_dense_col, Af, Bf, d_C_in, ldc_in, d_C_out, ldc, h->dense_mb, h->dense_W,
                               alpha, beta, 0);
        } else if (ntiles % 4 == 0) SX_BELL(4) else if (ntiles % 2 == 0) SX_BELL(2) else SX_BELL(1)
#undef SX_BELL
        const int row0 = h->dense_mb * 32;
        if (row0 < h->M) {
            const int64_t tot = (int64_t)(h->M - row0) * N;
            hipLaunchKernelGGL(sx::scale_tail_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, d_C_in, ldc_in,
                               d_C_out, ldc, row0, h->M, N, alpha, beta);
        }
        beta = 1.0f; d_C_in = d_C_out; ldc_in = ldc;
    }
    // a row range keeps the panel kernel when it starts and ends on row-block boundaries of the plan
    int blk0 = 0, blk1 = h->ps.plan_nblk;
    if (use_panel && !whole) {
        const auto &br = h->ps.h_blk_row;
        const auto i0 = std::lower_bound(br.begin(), br.end(), row_begin), i1 = std::lower_bound(br.begin(), br.end(), row_end);
        if (i0 == br.end() || *i0 != row_begin || i1 == br.end() || *i1 != row_end) use_panel = false;
        else { blk0 = (int)(i0 - br.begin()); blk1 = (int)(i1 - br.begin()); }
    }
    // ... and the window kernel when it starts and ends on wavefront (rows-per-wave) boundaries
    if (use_window && (row_begin % h->win_rw != 0 || (row_end % h->win_rw != 0 && row_end != h->M))) use_window = false;
    // long rows of this range: entries [hub0, hub1) of a piece table, pieces [v0, v1).  Whole-matrix calls walk the
    // table sorted by length (workgroups of equally long pieces), row ranges the one sorted by row.
    const sextans_engine::PieceTable &pt = whole ? h->by_len : h->by_row;
    int hub0 = 0, hub1 = 0, v0 = 0, v1 = 0;
    if (h->nhub > 0) {
        if (whole) { hub1 = h->nhub; }
        else {
            hub0 = (int)(std::lower_bound(pt.h_row.begin(), pt.h_row.end(), row_begin) - pt.h_row.begin());
            hub1 = (int)(std::lower_bound(pt.h_row.begin(), pt.h_row.end(), row_end) - pt.h_row.begin());
        }
        v0 = pt.h_vfirst[(size_t)hub0]; v1 = pt.h_vfirst[(size_t)hub1];
    }
    const bool hubs = hub1 > hub0;
    int ch0 = 0, ch1 = h->nchain;              // chain rows of this range
    if (h->nchain > 0 && !whole) {
        ch0 = (int)(std::lower_bound(h->h_chain_row.begin(), h->h_chain_row.end(), row_begin) - h->h_chain_row.begin());
        ch1 = (int)(std::lower_bound(h->h_chain_row.begin(), h->h_chain_row.end(), row_end) - h->h_chain_row.begin());
    }
    const bool chains = ch1 > ch0;
    auto fold = [&]() {
        const int64_t tot = (int64_t)(hub1 - hub0) * N;
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, pt.d_vfirst, pt.d_row, h->d_P,
                               (int64_t)h->split_nv, d_C_in, ldc_in, d_C_out, ldc, hub0, hub1 - hub0, N, row_begin, alpha, beta);
        };
        if (h->opt_exact) go(sx::fold_hub_pieces<true>); else go(sx::fold_hub_pieces<false>);
    };
    if (use_window) {
        // B in 8-column panels (the reference's N tile), then one tile-major launch
        if (!(flags & SEXTANS_ROWS_REUSE_B_PANELS) || h->bp_layout != 8) {
            Prof p(h, &h->ev_repack, s);
            launch_repack<8>(d_B, ldb, h->d_Bp, h->K, 0, N / 8, s);
            h->bp_layout = 8;
        }
        {
            Prof p(h, &h->ev_kernel, s);
            const int w0 = row_begin / h->win_rw, w1 = (row_end + h->win_rw - 1) / h->win_rw;
            launch_window(h, h->d_Bp, d_C_in, ldc_in, d_C_out, ldc, N / 8, w0, w1, row_begin, alpha, beta, s);
            if (hubs) { launch_hub_pieces<2>(h, pt, h->d_Bp, N / 8, 0, v0, v1, s); fold(); }
            if (chains) {
                const std::vector<Seg> p8{{8, 0, N / 8}};
                launch_chains(h, p8, d_C_in, ldc_in, d_C_out, ldc, N, ch0, ch1, row_begin, alpha, beta, s);
            }
            h->last_kernel = kernel_name(2, hubs || chains, h->dense_W > 0);
        }
        SX_HIP(hipGetLastError());
        return SEXTANS_OK;
    }
    // Small B (fits the L2s), dictionary-only plan, one N segment: the panel kernel stages straight from the
    // caller's column-major B and the repack launch disappears.
    const bool fuse_b = use_panel && !h->ps.plan_mixed && h->opt_fuse_b && !(flags & kRowsNoFuseB) && plan.size() == 1 &&
                        plan[0].width == W && !hubs && !chains &&
                        (size_t)h->K * (size_t)N * sizeof(float) <= ((size_t)16 << 20);
    // (a reuse request is honoured only if the panels in the workspace have this layout: row-range calls of
    // one pipelined SpMM may alternate between the window kernel's 8-column panels and these)
    const bool skip_repack = fuse_b || ((flags & SEXTANS_ROWS_REUSE_B_PANELS) != 0 && h->bp_layout == W);

    if (!skip_repack) {
        h->bp_layout = W;
        Prof p(h, &h->ev_repack, s);
        for (const Seg &g : plan) {
            float *dst = h->d_Bp + (size_t)h->K * (size_t)g.col0;
            switch (g.width) {
                case 32: launch_repack<32>(d_B, ldb, dst, h->K, g.col0, g.ntiles, s); break;
                case 16: launch_repack<16>(d_B, ldb, dst, h->K, g.col0, g.ntiles, s); break;
                default: launch_repack<8>(d_B, ldb, dst, h->K, g.col0, g.ntiles, s); break;
            }
        }
    }
    {
        Prof p(h, &h->ev_kernel, s);
        bool v2_used = false;
        if (chains) {
            // the chains need one or two wavefronts for about a millisecond: on their own stream, beside the main kernel
            // (fork after the B panels are in place, join before the call's work on `s` is considered complete)
            SX_HIP(hipEventRecord(h->ev_fork, s));
            SX_HIP(hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0));
            launch_chains(h, plan, d_C_in, ldc_in, d_C_out, ldc, N, ch0, ch1, row_begin, alpha, beta, h->aux_stream);
            SX_HIP(hipEventRecord(h->ev_join, h->aux_stream));
        }
        for (const Seg &g : plan) {
            const float *bp = h->d_Bp + (size_t)h->K * (size_t)g.col0;
            const float *cin = d_C_in + (int64_t)g.col0 * ldc_in;
            float *cout = d_C_out + (int64_t)g.col0 * ldc;
            const bool panel_here = use_panel && g.width == W;   // the plan is built for width W
            const float *bsrc = fuse_b ? d_B + (int64_t)g.col0 * ldb : bp;
            const int64_t bld = fuse_b ? ldb : 0;
            // N >= 32 on a dictionary-only plan at 4 lanes per row: register-blocked 32-column super tiles with the
            // tile loop inside the workgroup; an odd 16-column tile at the end goes to the plain panel kernel
            // (32-bit byte offsets inside the wide kernels: panels, C columns)
            const bool wide_ok = h->ps.plan_max_dict <= sx::kWideMaxDict && (int64_t)h->K * 64 < ((int64_t)1 << 32) &&
                                 std::max(ldc, ldc_in) * 64 < ((int64_t)1 << 32) && (!fuse_b || ldb * 64 < ((int64_t)1 << 32));
            if (panel_here && g.width == 16 && !h->ps.plan_mixed && h->opt_cols_per_lane == 8 && g.ntiles >= 2 && wide_ok) {
                const int nsuper = g.ntiles / 2;
                if (int rc = launch_panel_v2<2>(h, bsrc, cin, ldc_in, cout, ldc, nsuper, alpha, beta, s, bld, blk0, blk1, row_begin))
                    return rc;
                if (g.ntiles & 1) {
                    const int64_t c0 = (int64_t)nsuper * 32;
                    launch_panel<4>(h, fuse_b ? bsrc + c0 * ldb : bsrc + c0 * (int64_t)h->K, cin + c0 * ldc_in, ldc_in, cout + c0 * ldc,
                                    ldc, 1, alpha, beta, s, bld, blk0, blk1, row_begin);
                }
                v2_used = true;
                if (hubs) launch_hub_pieces<4>(h, pt, bp, g.ntiles, g.col0, v0, v1, s);
                continue;
            }
            // (column-major staging keeps the round-1 kernel unless the rows are short: then the register-resident form fits
            // 128 registers together with a panel in registers)
            const bool short_rows = h->M > 0 && h->m_nnz / h->M + 8 <= 32;
            if (panel_here && g.width == 16 && !h->ps.plan_mixed && h->opt_panel_v2 != 0 && (!fuse_b || short_rows) && wide_ok) {
                // whole-matrix calls on repacked panels: the plan over the rows in clustered (brick) order when the matrix has one
                const bool clustered = whole && !fuse_b && h->cluster_state == 1;
                if (int rc = launch_panel_v2<1>(h, bsrc, cin, ldc_in, cout, ldc, g.ntiles, alpha, beta, s, bld, clustered ? 0 : blk0,
                                                clustered ? h->psc.plan_nblk : blk1, row_begin, clustered))
                    return rc;
                v2_used = true;
                if (hubs) launch_hub_pieces<4>(h, pt, bp, g.ntiles, g.col0, v0, v1, s);
                continue;
            }
#define SX_SEG(L)                                                                                                       \
    if (panel_here) launch_panel<L>(h, bsrc, cin, ldc_in, cout, ldc, g.ntiles, alpha, beta, s, bld, blk0, blk1, row_begin);  \
    else launch_rowgroup<L>(h, h->m_rp, h->m_rp + 1, h->m_ci, h->m_v, false, h->d_skip, bp, cin, ldc_in, cout, ldc, row_begin, row_end, \
                            g.ntiles, alpha, beta, s);                                                                   \
    if (hubs) launch_hub_pieces<L>(h, pt, bp, g.ntiles, g.col0, v0, v1, s);
            switch (g.width) {
                case 32: SX_SEG(8) break;
                case 16: SX_SEG(4) break;
                default: SX_SEG(2) break;
            }
#undef SX_SEG
        }
        if (hubs) fold();
        if (chains) SX_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
        h->last_kernel = kernel_name(v2_used ? 3 : use_panel ? 1 : 0, hubs || chains, h->dense_W > 0);
    }
    SX_HIP(hipGetLastError());
    return SEXTANS_OK;
}

}  // extern "C"
namespace {
// The timed region of the host-buffer entry points: [pre] + rp_time x SpMM from h->d_B / h->d_Cin into
// h->d_Cout + [post], on the engine's own stream.  B is the same in every repeat, so its panel repack runs
// in the first one only (the reference re-lays B out on the host, outside its timed region:
// sextans-host.cpp:150-177).  The repeats are captured once into a hipGraph (instantiated outside the timed
// region) and replayed with a single launch: the loop is launch-bound for small matrices (nasa4704: 4.3 us
// per repeat replayed vs 7.8 us launched one by one).
template <class Pre, class Post>
int run_repeats(sextans_engine *h, int N, float alpha, float beta, int rp_time, Pre pre, Post post, double *ns) {
    if (!h->host_stream) {
        SX_HIP(hipStreamCreateWithFlags(&h->host_stream, hipStreamNonBlocking));
        // first use of a stream sets up its hardware queue (~1 ms): keep that out of the timed region
        SX_HIP(hipMemsetAsync(h->d_Cout, 0, 4, h->host_stream));
        SX_HIP(hipStreamSynchronize(h->host_stream));
    }
    hipStream_t cs = h->host_stream;
    // `count` repeats; the first one lays B out in panels, the others reuse them.  Loops of four or more
    // repeats always use the panel-staged kernel (one repack amortised) instead of the column-major staging.
    const int nofuse = rp_time >= 4 ? kRowsNoFuseB : 0;
    auto enqueue = [&](int count) -> int {
        for (int r = 0; r < count; ++r)
            if (int rc = sextans_spmm_device_rows(h, N, alpha, h->d_B, h->K, beta, h->d_Cin, h->M, h->d_Cout, h->M,
                                                  0, h->M, (r ? SEXTANS_ROWS_REUSE_B_PANELS : 0) | nofuse, (void *)cs))
                return rc;
        return SEXTANS_OK;
    };
    struct Cleanup {   // released on every exit path
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Cleanup() {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } c;
    const bool use_graph = !h->opt_profile && !h->opt_phase_timing;
    const int per_graph = rp_time < 128 ? rp_time : 128;   // bound the graph; long loops replay it
    if (use_graph) {
        // relaxed mode: the enqueue path calls hipSetDevice / hipGetLastError, which thread-local capture rejects
        SX_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
        const int rc = enqueue(per_graph);
        const hipError_t ce = hipStreamEndCapture(cs, &c.graph);
        if (rc) return rc;
        SX_HIP(ce);
        SX_HIP(hipGraphInstantiate(&c.exec, c.graph, nullptr, nullptr, 0));
    }
    SX_HIP(hipEventCreate(&c.e0));
    SX_HIP(hipEventCreate(&c.e1));
    SX_HIP(hipEventRecord(c.e0, cs));
    if constexpr (!std::is_same<Pre, std::nullptr_t>::value) pre(cs);
    if (use_graph) {
        for (int done = 0; done + per_graph <= rp_time; done += per_graph) SX_HIP(hipGraphLaunch(c.exec, cs));
        if (int rc = enqueue(rp_time % per_graph)) return rc;
    } else if (int rc = enqueue(rp_time)) {
        return rc;
    }
    if constexpr (!std::is_same<Post, std::nullptr_t>::value) post(cs);
    SX_HIP(hipGetLastError());
    SX_HIP(hipEventRecord(c.e1, cs));
    SX_HIP(hipEventSynchronize(c.e1));
    float ms = 0.f;
    SX_HIP(hipEventElapsedTime(&ms, c.e0, c.e1));
    *ns = (double)ms * 1e6;
    return SEXTANS_OK;
}
}  // namespace
extern "C" {

int sextans_spmm_host(sextans_handle_t h, int N, float alpha, const float *B, float beta, float *C,
                      int rp_time, double *elapsed_ns) {
    if (!h || !B || !C || N <= 0 || (N % 8) != 0) return SEXTANS_ERR_INVALID;
    if (!h->d_rp) return SEXTANS_ERR_STATE;
    if (rp_time < 1) rp_time = 1;
    SX_HIP(hipSetDevice(h->device));
    const size_t nB = (size_t)h->K * (size_t)N, nC = (size_t)h->M * (size_t)N;
    if (int rc = ensure(&h->d_B, &h->B_cap, nB)) return rc;
    size_t ccap = h->C_cap;
    if (int rc = ensure(&h->d_Cin, &ccap, nC)) return rc;
    if (int rc = ensure(&h->d_Cout, &h->C_cap, nC)) return rc;
    SX_HIP(hipMemcpy(h->d_B, B, nB * sizeof(float), hipMemcpyHostToDevice));
    SX_HIP(hipMemcpy(h->d_Cin, C, nC * sizeof(float), hipMemcpyHostToDevice));
    {   // allocations and the one-time packing of A stay outside the timed region
        std::vector<Seg> plan; int W = 0; bool up = false, uw = false;
        if (int rc = prepare(h, N, plan, W, up, uw)) return rc;
    }
    double ns = 0.0;
    if (int rc = run_repeats(h, N, alpha, beta, rp_time, nullptr, nullptr, &ns)) return rc;
    if (elapsed_ns) *elapsed_ns = ns;
    SX_HIP(hipMemcpy(C, h->d_Cout, nC * sizeof(float), hipMemcpyDeviceToHost));
    return SEXTANS_OK;
}

int sextans_set_matrix_edges(sextans_handle_t h, const int32_t *edge_list_ptr, const uint64_t *const *edge_list_ch,
                             int NUM_ITE, int NUM_A_LEN, int M, int K) {
    if (!h || !edge_list_ptr || !edge_list_ch || NUM_ITE < 0 || M < 0 || K < 0) return SEXTANS_ERR_INVALID;
    // NUM_ITE = ceil(K / 4096) (sextans-host.cpp:221): checked BEFORE edge_list_ptr[NUM_ITE] is read
    if ((int64_t)NUM_ITE != ((int64_t)K + SEXTANS_EDGES_WINDOW - 1) / SEXTANS_EDGES_WINDOW) return SEXTANS_ERR_INVALID;
    if (edge_list_ptr[NUM_ITE] != NUM_A_LEN) return SEXTANS_ERR_INVALID;
    int64_t nnz = 0;
    int *rp = nullptr, *ci = nullptr;
    float *v = nullptr;
    if (int rc = sextans_edges_decode_csr(edge_list_ptr, edge_list_ch, NUM_ITE, M, K, &nnz, &rp, &ci, &v))
        return rc;
    const int rc = sextans_set_matrix_csr(h, M, K, nnz, rp, ci, v);
    free(rp); free(ci); free(v);
    return rc;
}

int sextans_invoke(sextans_handle_t h, const int32_t *edge_list_ptr, const uint64_t *const *edge_list_ch,
                   const float *const *mat_B_ch, int num_ch_b, const float *const *mat_C_ch_in,
                   float *const *mat_C_ch, int NUM_ITE, int NUM_A_LEN, int M, int K, int P_N, int alpha_u,
                   int beta_u, double *elapsed_ns) {
    const int N = P_N & 0xFFFF;                      // sextans-host.cpp:223, sextans.cpp:203
    int rp_time = (int)((unsigned)P_N >> 16);
    if (rp_time < 1) rp_time = 1;
    if (!h || !mat_B_ch || !mat_C_ch_in || !mat_C_ch || N <= 0 || (N % 8) || (num_ch_b != 4 && num_ch_b != 8) ||
        M < 0 || K < 0)
        return SEXTANS_ERR_INVALID;
    if (edge_list_ptr) {
        if (int rc = sextans_set_matrix_edges(h, edge_list_ptr, edge_list_ch, NUM_ITE, NUM_A_LEN, M, K)) return rc;
    } else if (!h->d_rp) {
        return SEXTANS_ERR_STATE;
    } else if (h->M != M || h->K != K) {
        return SEXTANS_ERR_INVALID;
    }
    float alpha, beta;
    memcpy(&alpha, &alpha_u, 4);                     // raw fp32 bits, sextans-host.cpp:225-229
    memcpy(&beta, &beta_u, 4);
    SX_HIP(hipSetDevice(h->device));
    const int64_t b_cs = sextans_chan_b_colsize(K, num_ch_b), b_len = sextans_chan_b_len(K, N, num_ch_b);
    const int64_t c_cs = sextans_chan_c_colsize(M), c_len = sextans_chan_c_len(M, N);
    const int64_t b_used = b_cs * (N / 8), c_used = c_cs * (N / 8);
    const size_t nB = (size_t)K * (size_t)N, nC = (size_t)M * (size_t)N;
    if (int rc = ensure(&h->d_chB, &h->chB_cap, (size_t)b_len * num_ch_b)) return rc;
    if (int rc = ensure(&h->d_chC, &h->chC_cap, (size_t)c_len * 8)) return rc;
    if (int rc = ensure(&h->d_B, &h->B_cap, nB)) return rc;
    size_t ccap = h->C_cap;
    if (int rc = ensure(&h->d_Cin, &ccap, nC)) return rc;
    if (int rc = ensure(&h->d_Cout, &h->C_cap, nC)) return rc;
    for (int c = 0; c < num_ch_b; ++c) {
        if (!mat_B_ch[c]) return SEXTANS_ERR_INVALID;
        SX_HIP(hipMemcpy(h->d_chB + (size_t)c * b_len, mat_B_ch[c], sizeof(float) * (size_t)b_used,
                         hipMemcpyHostToDevice));
    }
    for (int c = 0; c < 8; ++c) {
        if (!mat_C_ch_in[c] || !mat_C_ch[c]) return SEXTANS_ERR_INVALID;
        SX_HIP(hipMemcpy(h->d_chC + (size_t)c * c_len, mat_C_ch_in[c], sizeof(float) * (size_t)c_used,
                         hipMemcpyHostToDevice));
    }
    {
        std::vector<Seg> plan; int W = 0; bool up = false, uw = false;
        if (int rc = prepare(h, N, plan, W, up, uw)) return rc;
    }
    const float pad = alpha * 0.0f + beta * 0.0f;    // what the accelerator writes into rows M .. colsize-1
    auto pre = [&](hipStream_t cs) {
        if (K > 0)
            sx::chan_unpack_b<<<dim3((unsigned)((K + 255) / 256), (unsigned)N), 256, 0, cs>>>(
                h->d_chB, b_len, b_cs, num_ch_b, K, N, h->d_B);
        if (M > 0)
            sx::chan_unpack_c<<<dim3((unsigned)((M + 255) / 256), (unsigned)(N / 8)), 256, 0, cs>>>(
                h->d_chC, c_len, c_cs, M, N, h->d_Cin);
    };
    auto post = [&](hipStream_t cs) {
        if (c_cs > 0)
            sx::chan_pack_c<<<dim3((unsigned)((c_cs + 255) / 256), (unsigned)(N / 8)), 256, 0, cs>>>(
                h->d_Cout, M, N, c_len, c_cs, pad, h->d_chC);
    };
    double ns = 0.0;
    if (M > 0) {
        if (int rc = run_repeats(h, N, alpha, beta, rp_time, pre, post, &ns)) return rc;
    } else {
        pre(nullptr); post(nullptr);
        SX_HIP(hipDeviceSynchronize());
    }
    if (elapsed_ns) *elapsed_ns = ns;
    for (int c = 0; c < 8; ++c)
        SX_HIP(hipMemcpy(mat_C_ch[c], h->d_chC + (size_t)c * c_len, sizeof(float) * (size_t)c_used,
                         hipMemcpyDeviceToHost));
    return SEXTANS_OK;
}

int sextans_spmm_csr(int M, int N, int K, int NNZ, float ALPHA, const int *CSRRowPtr,
                     const int *CSRColIndex, const float *CSRVal, const float *mat_B, float BETA,
                     float *mat_C) {
    sextans_handle_t h = nullptr;
    if (int rc = sextans_create(&h, 0)) return rc;
    int rc = sextans_set_matrix_csr(h, M, K, NNZ, CSRRowPtr, CSRColIndex, CSRVal);
    if (!rc) rc = sextans_spmm_host(h, N, ALPHA, mat_B, BETA, mat_C, 1, nullptr);
    sextans_destroy(h);
    return rc;
}

int sextans_set_matrix_bell_device(sextans_handle_t h, int M, int K, int ell_width,
                                   const int *d_block_col, const uint16_t *d_block_val) {
    if (!h || M <= 0 || K <= 0 || (M % 32) || (K % 32) || ell_width <= 0 || !d_block_col || !d_block_val)
        return SEXTANS_ERR_INVALID;
    SX_HIP(hipSetDevice(h->device));
    free_bell(h);
    const int64_t nslots = (int64_t)(M / 32) * ell_width;
    SX_HIP(hipMalloc(&h->d_bell_Af, (size_t)nslots * 2048));
    const int64_t threads = nslots * 128;
    hipLaunchKernelGGL(sx::bell_repack_a, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, nullptr,
                       d_block_val, (sx::u32x4 *)h->d_bell_Af, nslots);
    SX_HIP(hipDeviceSynchronize());
    h->d_bell_col = d_block_col;
    h->bell_M = M; h->bell_K = K; h->bell_W = ell_width;
    {   // do the block rows of a workgroup share block columns?  (decides between the per-wavefront kernels and the
        // LDS-shared one, "MFMA only where a tile is actually dense" + reuse)
        unsigned long long *d_cnt = nullptr, h_cnt[3] = {0, 0, 0};
        SX_HIP(hipMalloc((void **)&d_cnt, 3 * sizeof(unsigned long long)));
        SX_HIP(hipMemset(d_cnt, 0, 3 * sizeof(unsigned long long)));
        const int groups = (M / 32 + sx::kShRows - 1) / sx::kShRows;
        hipLaunchKernelGGL(sx::bell_union_count, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, nullptr, d_block_col, M / 32,
                           ell_width, d_cnt, d_cnt + 1, d_cnt + 2);
        const hipError_t e = hipMemcpy(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost);
        (void)hipFree(d_cnt);
        SX_HIP(e);
        h->bell_share = h_cnt[0] ? (double)h_cnt[1] / (double)h_cnt[0] : 0.0;
        h->bell_max_union = (int)h_cnt[2];
    }
    return SEXTANS_OK;
}

int sextans_set_matrix_bell(sextans_handle_t h, int M, int K, int ell_width, const int *block_col,
                            const uint16_t *block_val) {
    if (!h || M <= 0 || K <= 0 || (M % 32) || (K % 32) || ell_width <= 0 || !block_col || !block_val)
        return SEXTANS_ERR_INVALID;
    SX_HIP(hipSetDevice(h->device));
    const size_t nslots = (size_t)(M / 32) * (size_t)ell_width;
    int *d_col = nullptr;
    uint16_t *d_val = nullptr;
    SX_HIP(hipMalloc((void **)&d_col, nslots * sizeof(int)));
    SX_HIP(hipMalloc((void **)&d_val, nslots * 2048));
    SX_HIP(hipMemcpy(d_col, block_col, nslots * sizeof(int), hipMemcpyHostToDevice));
    SX_HIP(hipMemcpy(d_val, block_val, nslots * 2048, hipMemcpyHostToDevice));
    int rc = sextans_set_matrix_bell_device(h, M, K, ell_width, d_col, d_val);
    (void)hipFree(d_val);
    if (rc) { (void)hipFree(d_col); return rc; }
    h->d_bell_col_owned = d_col;
    return SEXTANS_OK;
}

int sextans_spmm_bell_device(sextans_handle_t h, int N, float alpha, const uint16_t *d_B, int64_t ldb,
                             float beta, const float *d_C_in, float *d_C_out, int64_t ldc, void *stream) {
    if (!h || N <= 0 || (N % 32) || !d_B || !d_C_in || !d_C_out) return SEXTANS_ERR_INVALID;
    if (!h->d_bell_Af) return SEXTANS_ERR_STATE;
    if (ldb < h->bell_K || (ldb % 8) || ldc < h->bell_M) return SEXTANS_ERR_INVALID;
    SX_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int kblocks = h->bell_K / 32, mblocks = h->bell_M / 32, ntiles = N / 32;
    const size_t need = (size_t)h->bell_K * (size_t)N * 2;
    if (h->bell_Bf_cap < need) {
        if (h->d_bell_Bf) SX_HIP(hipFree(h->d_bell_Bf));
        h->d_bell_Bf = nullptr; h->bell_Bf_cap = 0;
        SX_HIP(hipMalloc(&h->d_bell_Bf, need));
        h->bell_Bf_cap = need;
    }
    {
        Prof p(h, &h->ev_repack, s);
        const int64_t threads = (int64_t)kblocks * ntiles * 128;
        hipLaunchKernelGGL(sx::bell_repack_b, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_B,
                           ldb, (sx::u32x4 *)h->d_bell_Bf, kblocks, ntiles);
    }
    {
        Prof p(h, &h->ev_kernel, s);
        const auto *Af = (const sx::bf16x8 *)h->d_bell_Af;
        const auto *Bf = (const sx::bf16x8 *)h->d_bell_Bf;
#define SX_BELL(NSUB)                                                                                  \
    {                                                                                                  \
        const int64_t waves = (int64_t)mblocks * (ntiles / NSUB);                                      \
        hipLaunchKernelGGL((sx::spmm_bell_mfma<NSUB>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, \
                           h->d_bell_col, Af, Bf, d_C_in, ldc, d_C_out, ldc, mblocks, h->bell_W, ntiles,    \
                           alpha, beta);                                                                      \
    }
        const bool shared = ntiles == 8 && h->opt_bell_shared != 0 && sx::kShRows * h->bell_W <= sx::kShMaxRowCols &&
                            h->bell_max_union <= sx::kShMaxUnion &&
                            (h->opt_bell_shared == 1 || h->bell_share >= 1.5);
        if (shared) {
            constexpr size_t lds = (size_t)sx::kShRing * sx::kShTileBytes + (size_t)(sx::kShMaxUnion + 8) * (sizeof(int) + sx::kShRows * sizeof(short)) +
                                  (size_t)sx::kShMaxRowCols * sizeof(int);
            if (int rc = allow_big_lds(h, reinterpret_cast<const void *>(sx::spmm_bell_mfma_shared), (int)lds)) return rc;
            hipLaunchKernelGGL(sx::spmm_bell_mfma_shared, dim3((unsigned)((mblocks + sx::kShRows - 1) / sx::kShRows)), dim3(sx::kShThreads), lds,
                               s, h->d_bell_col, Af, Bf, d_C_in, ldc, d_C_out, ldc, mblocks, h->bell_W, alpha, beta, (int)h->opt_bell_debug);
            h->last_kernel = "spmm_bell_mfma_shared";
            SX_HIP(hipGetLastError());
            return SEXTANS_OK;
        } else if (ntiles == 8 && h->opt_bell_wide) {
            // "bell_generation" = G > 0: launches of G block rows, so that the wavefronts of a launch start at block
            // column 0 together and sweep K side by side (experiment: does the Infinity Cache then serve the B tiles?)
            const int G = h->opt_bell_gen > 0 ? (int)h->opt_bell_gen : mblocks;
            for (int b0 = 0; b0 < mblocks; b0 += G) {
                const int nb = std::min(G, mblocks - b0);
                hipLaunchKernelGGL(sx::spmm_bell_mfma_n256, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, s, h->d_bell_col, Af,
                                   Bf, d_C_in, ldc, d_C_out, ldc, std::min(mblocks, b0 + nb), h->bell_W, alpha, beta, b0);
            }
        } else if (ntiles % 4 == 0) SX_BELL(4) else if (ntiles % 2 == 0) SX_BELL(2) else SX_BELL(1)
#undef SX_BELL
        h->last_kernel = "spmm_bell_mfma";
    }
    SX_HIP(hipGetLastError());
    return SEXTANS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Native multi-GPU form (north_star: "A row-range partitioned across the GPUs of one node, B replicated,
// RCCL all-gather of C panels over xGMI") behind the C ABI, for callers that have no torch.distributed.
// RCCL is bound at run time (dlopen "librccl.so.1"): the single-GPU entry points never need it.
// ------------------------------------------------------------------------------------------------
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

int sextans_profile_reset(sextans_handle_t h) {
    if (!h) return SEXTANS_ERR_INVALID;
    for (auto *vec : {&h->ev_kernel, &h->ev_repack}) {
        for (auto &ep : *vec) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
        vec->clear();
    }
    return SEXTANS_OK;
}

int sextans_profile_read(sextans_handle_t h, double *mean_kernel_ns, int64_t *launches,
                         double *mean_repack_ns) {
    if (!h) return SEXTANS_ERR_INVALID;
    SX_HIP(hipSetDevice(h->device));
    auto mean = [&](std::vector<EventPair> &v, double *out) -> int {
        double tot = 0.0;
        for (auto &ep : v) {
            SX_HIP(hipEventSynchronize(ep.b));
            float ms = 0.f;
            SX_HIP(hipEventElapsedTime(&ms, ep.a, ep.b));
            tot += (double)ms * 1e6;
        }
        if (out) *out = v.empty() ? 0.0 : tot / (double)v.size();
        return SEXTANS_OK;
    };
    if (int rc = mean(h->ev_kernel, mean_kernel_ns)) return rc;
    if (int rc = mean(h->ev_repack, mean_repack_ns)) return rc;
    if (launches) *launches = (int64_t)h->ev_kernel.size();
    return SEXTANS_OK;
}

const char *sextans_last_kernel(sextans_handle_t h) { return h ? h->last_kernel : "none"; }

int sextans_device_free(int device, void *d_ptr) {
    SX_HIP(hipSetDevice(device));
    SX_HIP(hipFree(d_ptr));
    return SEXTANS_OK;
}

}  // extern "C"
