// engine_plan.hip -- what the engine prepares ONCE per matrix, outside every timed region, like the reference's host-side
// scheduling and packing (generate_edge_list_for_all_PEs + edge_list_64bit, sextans-host.cpp:114-148): the long-row split
// (bucketed rows, hub pieces, exact chains), the packed row-bucketed plans of the LDS-panel kernels (natural and clustered
// row order; built on the device, plan_device.hip / row_cluster.hip), the K-window stream, and prepare(), which brings all
// of that up to date for an N-column SpMM.
#include <algorithm>
#include <cstring>

#include "engine_state.h"
#include "graph_cluster.h"
#include "panel_plan.h"
#include "plan_device.h"
#include "row_cluster.h"
#include "spmm_csr_kernels.h"
#include "spmm_panel_v2.h"
#include "spmm_window_kernel.h"
#include "window_plan.h"

namespace sxe {

static_assert(sx::kPlanPadRows == sx::kWidePadRows, "the plan builder and the kernels agree on the +1.0f rows of a panel");

namespace {
// mixed plans: flag the rows the gather kernel must NOT write -- rows of blocks that have a dictionary (spmm_csr_panel_v2 writes them)
// and rows on the piece path
__global__ __launch_bounds__(256) void mixed_row_flags(int nblk, const int *__restrict__ blk_row, const int *__restrict__ dict_cnt,
                                                       const unsigned char *__restrict__ piece_rows, unsigned char *__restrict__ out) {
    const int blk = blockIdx.x;
    if (blk >= nblk) return;
    const int r0 = blk_row[blk], r1 = blk_row[blk + 1];
    const bool dict = dict_cnt[blk] > 0;
    for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) out[r] = dict || (piece_rows && piece_rows[r]) ? 1 : 0;
}
// groups of 128 rows with at least one row of the gather kernel's, appended in any order; list[ngroups_max] counts them
__global__ __launch_bounds__(256) void mixed_group_list(int M, const unsigned char *__restrict__ flags, int *__restrict__ list, int *__restrict__ count) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if ((int64_t)g * 128 >= M) return;
    bool any = false;
    for (int r = g * 128; r < min(M, (g + 1) * 128) && !any; ++r) any = flags[r] == 0;
    if (any) list[atomicAdd(count, 1)] = g;
}
}  // namespace

void free_panel_state(sextans_engine::PanelState &p) {
    (void)hipFree(p.d_dict_ptr); (void)hipFree(p.d_dict); (void)hipFree(p.d_lidx); (void)hipFree(p.d_blk_row);
    (void)hipFree(p.d_row_off); (void)hipFree(p.d_pcol32); (void)hipFree(p.d_pval); (void)hipFree(p.d_ioff); (void)hipFree(p.d_rg_skip); (void)hipFree(p.d_rg_groups); (void)hipFree(p.d_dict_blocks);
    p = sextans_engine::PanelState();
}

void free_cluster_plan(sextans_engine *h) {   // the clustered-order plan and its tables; it is reconsidered at the next whole-matrix call
    free_panel_state(h->psc);
    (void)hipFree(h->d_slot_row); (void)hipFree(h->d_colpos); (void)hipFree(h->d_dict_nat);
    h->d_slot_row = h->d_colpos = h->d_dict_nat = nullptr;
    h->cluster_cm_pays = true;
    h->cluster_runs = false;
    (void)hipFree(h->d_chain_ci_perm); (void)hipFree(h->d_chain_beg_c); (void)hipFree(h->d_chain_v_c);
    h->d_chain_ci_perm = h->d_chain_beg_c = nullptr; h->d_chain_v_c = nullptr;
    h->cluster_state = 0;
    h->colwise_state = 0;
    h->row_coherence = 0.0;
    h->cluster_decline = 0;
    h->cluster_total_dict = 0;
    h->cluster_shared = 0.0;
    h->cluster_graph_kind = 0;
    h->pattern_symmetry = 1.0;
}

void free_plan(sextans_engine *h) {   // every packed form of the current main matrix
    h->cluster_rm_tried = false;
    free_panel_state(h->ps);
    for (auto &p : h->plan_stash) free_panel_state(p);
    free_cluster_plan(h);
    h->cluster_s2 = h->cluster_s3 = 0;
    h->plan_total_dict = h->cluster_total_dict = 0;
}

int64_t device_bytes(const sextans_engine *h) {
    auto plan_bytes = [](const sextans_engine::PanelState &p) -> int64_t {
        if (!p.plan_built || !p.plan_lpr) return 0;
        const int64_t rb = (int64_t)(sx::kBlock / p.plan_lpr) * p.plan_sets;
        return (int64_t)p.plan_nblk * (4 * (2 + p.plan_dict_stride + 2 * rb)) +
               (p.stream_released ? 0 : p.plan_stream_len * 4 + p.plan_idx_len * 2 + (p.d_ioff ? (int64_t)p.plan_nblk * rb * 8 : 0) +
                                            (p.plan_mixed ? p.plan_stream_len * 4 : 4));
    };
    int64_t b = 0;
    const int64_t rows = (int64_t)h->M + 1;
    if (h->owns_matrix) b += rows * 4 + h->nnz * 8;
    if (h->d_srp) b += rows * 4 + h->s_nnz * 8;
    if (h->d_mrp) b += rows * 4 + h->m_nnz * 8 + h->M;
    b += ((int64_t)h->split_nv * 8 + (int64_t)h->nhub * 8) * 2 + (int64_t)h->nchain * 24;
    b += plan_bytes(h->ps) + plan_bytes(h->psc);
    for (const auto &p : h->plan_stash) b += plan_bytes(p);
    if (h->d_slot_row) b += (int64_t)h->psc.plan_nblk * 64 * h->psc.plan_sets * 4;
    if (h->d_colpos) b += (int64_t)h->K * 4;
    if (h->d_chain_ci_perm && !h->h_chain_off.empty()) b += (int64_t)h->h_chain_off.back() * 8 + (int64_t)h->nchain * 4;
    if (h->d_wstream) b += h->win_padded * 8 + (int64_t)h->win_nwaves * 4;
    if (h->d_dense_Af) b += (int64_t)h->dense_mb * h->dense_W * (2048 + 4);
    if (h->d_bell_Af) b += (int64_t)(h->bell_M / 32) * h->bell_W * (2048 + (h->d_bell_col_owned ? 4 : 0));
    b += (int64_t)h->bell_Bf_cap;
    b += 4 * (int64_t)(h->Bp_cap + h->B_cap + (h->d_Cin ? h->C_cap : 0) + h->C_cap + h->P_cap + h->stage_cap + h->chB_cap + h->chC_cap + h->Cs_cap + h->rmB_cap + h->rmC_cap + h->Cfull_cap + h->dist_rows_cap);
    return b;
}

void free_window(sextans_engine *h) {
    (void)hipFree(h->d_wstream); (void)hipFree(h->d_wstep0);
    h->d_wstream = nullptr; h->d_wstep0 = nullptr;
    h->win_nwaves = h->win_rw = 0;
    h->win_padded = 0;
    h->win_state = 0;
    h->win_built_rows = h->win_built_cols = -1;
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
    (void)hipFree(h->d_rb_row0); (void)hipFree(h->d_rb_gptr); (void)hipFree(h->d_rb_gcol); (void)hipFree(h->d_rb_A);
    h->d_rb_row0 = h->d_rb_gptr = h->d_rb_gcol = nullptr; h->d_rb_A = nullptr;
    h->rb_n = 0; h->rb_groups = 0;
    (void)hipFree(h->d_sb_uptr); (void)hipFree(h->d_sb_ucol); (void)hipFree(h->d_sb_umask);
    h->d_sb_uptr = h->d_sb_ucol = nullptr; h->d_sb_umask = nullptr;
    h->sb_n = 0; h->sb_entries = 0;
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
    h->col_range_known = false;
    (void)hipFree(h->d_touched);
    h->d_touched = nullptr; h->touched_segments = 0;
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

// Device copy of the CSR arrays -> host, validated: the host-side plan builders index arrays of size K with
// the column indices and trust row_ptr to be monotonic (a matrix handed over with
// sextans_set_matrix_csr_device has not been looked at by anybody yet).
// (level 2: the main matrix the kernels work on; 1: the source of the long-row split; 0: the matrix as the caller set it)
int read_back_row_ptr(sextans_engine *h, std::vector<int> &rp, int level) {
    rp.resize((size_t)h->M + 1);
    const int *src = level == 2 ? h->m_rp : level == 1 ? h->s_rp : h->d_rp;
    SX_HIP(hipMemcpy(rp.data(), src, sizeof(int) * ((size_t)h->M + 1), hipMemcpyDeviceToHost));
    if (rp[0] != 0 || (int64_t)rp[(size_t)h->M] != (level == 2 ? h->m_nnz : level == 1 ? h->s_nnz : h->nnz)) return SEXTANS_ERR_INVALID;
    for (int r = 0; r < h->M; ++r)
        if (rp[(size_t)r + 1] < rp[(size_t)r]) return SEXTANS_ERR_INVALID;
    return SEXTANS_OK;
}
int read_back_entries(sextans_engine *h, std::vector<int> &ci, std::vector<float> &va, int level) {
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

// DevicePlan -> PanelState (the arrays change owner)
void adopt_device_plan(sextans_engine::PanelState &c, sx::DevicePlan &dp, const sextans_engine *h, int lpr, int cap) {
    c.plan_lpr = lpr;
    c.plan_sets = dp.sets;
    c.plan_min_reuse = plan_key(h);
    c.plan_nblk = dp.nblk;
    c.plan_dict_stride = dp.dict_stride;
    c.plan_mixed = dp.mixed;
    c.d_blk_row = dp.d_blk_row; c.d_dict_ptr = dp.d_dict_cnt; c.d_dict = dp.d_dict; c.d_row_off = dp.d_slot_info;
    c.d_lidx = dp.d_idx16; c.d_pcol32 = dp.d_col32; c.d_pval = dp.d_val; c.d_ioff = dp.d_ioff;
    c.plan_idx_len = dp.idx_len;
    c.h_blk_row.swap(dp.h_blk_row);
    c.plan_stream_len = dp.stream_len;
    c.plan_nnz_panel = dp.nnz_in_panel_blocks;
    c.plan_max_dict = dp.max_dict;
    c.plan_max_row = dp.max_row_len;
    c.plan_pad_row = cap;
    c.plan_built = true;
    dp = sx::DevicePlan();
}

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
// Share of the non-zeros in row blocks with reuse from which a whole-matrix call on a MIXED plan runs its split form (dictionary blocks
// on spmm_csr_panel_v2 + the other rows on the gather kernel) instead of the gather kernel alone; everything else keeps 0.5.
constexpr double kSplitMinFrac = 0.15;
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
        // (the split form of a mixed plan pays from a much smaller share on: its blocks with reuse run at the panel kernel's speed, the
        // rest at the gather kernel's, which they would run at anyway -- prepare(): kSplitMinFrac)
        const double build_from = lpr == 4 && h->opt_split_mixed != 0 && h->opt_kernel == 0 ? kSplitMinFrac : 0.5;
        if (frac < build_from) {   // no reuse worth an LDS panel: remember the verdict, skip the build
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
    const int brc = sx::build_panel_plan_device(h->M, h->K, h->m_rp, h->m_ci, h->m_v, lpr, cap, min_reuse, dp, err, nullptr,
                                                lpr == 4 && h->opt_share_index != 0);
    if (brc == 2) { g_last_error = err; sx::free_device_plan(dp); return SEXTANS_ERR_HIP; }
    h->ps.plan_lpr = lpr;
    h->ps.plan_min_reuse = plan_key(h);
    h->ps.plan_narrow_frac = narrow_frac;
    if (brc == 1) {   // rows padded to 4 entries exceed 32-bit entry offsets: row-group kernel only
        h->ps.plan_panel_frac = 0.0;
        h->ps.plan_built = false;
        return SEXTANS_OK;
    }
    // A handful of row blocks without reuse (the tail of a cluster order, a few unstructured rows in a mesh matrix) made the whole plan
    // "mixed" and took the register-resident kernel away from EVERY block: 815 us instead of ~620 for a 4M-row mesh matrix renumbered by
    // a cluster order (round 5).  When the blocks without a dictionary hold under 2 % of the non-zeros they get one as well (a block
    // always fits the panel by construction; a dictionary without reuse costs those few blocks a panel copy they do not amortise).
    if (brc == 0 && dp.mixed && lpr == 4 && h->m_nnz > 0 && (double)dp.nnz_in_panel_blocks >= 0.98 * (double)h->m_nnz) {
        sx::DevicePlan all;
        const int rc2 = sx::build_panel_plan_device(h->M, h->K, h->m_rp, h->m_ci, h->m_v, lpr, cap, 0.0, all, err, nullptr, h->opt_share_index != 0);
        if (rc2 == 0 && !all.mixed && all.dict_stride <= 9 * RB) { sx::free_device_plan(dp); dp = all; all = sx::DevicePlan(); h->ps.plan_all_dict = true; }
        else { sx::free_device_plan(all); if (rc2 == 2) (void)hipGetLastError(); }
    }
    if (dp.dict_stride > 9 * RB) { sx::free_device_plan(dp); return SEXTANS_ERR_STATE; }   // capacity = 9 * RB by construction
    h->ps.plan_nblk = dp.nblk;
    h->ps.plan_dict_stride = dp.dict_stride;
    h->ps.plan_mixed = dp.mixed;
    h->ps.d_blk_row = dp.d_blk_row; h->ps.d_dict_ptr = dp.d_dict_cnt; h->ps.d_dict = dp.d_dict; h->ps.d_row_off = dp.d_slot_info;
    h->ps.d_lidx = dp.d_idx16; h->ps.d_pcol32 = dp.d_col32; h->ps.d_pval = dp.d_val; h->ps.d_ioff = dp.d_ioff;
    h->ps.plan_idx_len = dp.idx_len;
    h->ps.h_blk_row.swap(dp.h_blk_row);
    h->ps.plan_stream_len = dp.stream_len;
    h->ps.plan_panel_frac = h->m_nnz ? (double)dp.nnz_in_panel_blocks / (double)h->m_nnz : 0.0;
    h->ps.plan_nnz_panel = dp.nnz_in_panel_blocks;
    h->ps.plan_max_dict = dp.max_dict;
    h->ps.plan_max_row = dp.max_row_len;
    h->ps.plan_pad_row = cap;
    h->ps.plan_built = true;
    if (lpr == 4) h->plan_total_dict = dp.total_dict;
    if (h->ps.plan_mixed && lpr == 4 && h->ps.plan_nblk > 0) {   // the split form of a mixed plan (engine.hip: split_mixed)
        (void)hipFree(h->ps.d_rg_skip); h->ps.d_rg_skip = nullptr;
        if (hipMalloc((void **)&h->ps.d_rg_skip, (size_t)std::max(h->M, 1)) == hipSuccess) {
            hipLaunchKernelGGL(mixed_row_flags, dim3((unsigned)h->ps.plan_nblk), dim3(256), 0, nullptr, h->ps.plan_nblk, h->ps.d_blk_row, h->ps.d_dict_ptr,
                               (const unsigned char *)h->d_skip, h->ps.d_rg_skip);
            const int ng = (h->M + 127) / 128;
            (void)hipFree(h->ps.d_rg_groups); h->ps.d_rg_groups = nullptr; h->ps.rg_ngroups = 0;
            if (hipMalloc((void **)&h->ps.d_rg_groups, sizeof(int) * ((size_t)ng + 1)) == hipSuccess) {
                SX_HIP(hipMemsetAsync(h->ps.d_rg_groups + ng, 0, sizeof(int), nullptr));
                hipLaunchKernelGGL(mixed_group_list, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, nullptr, h->M, h->ps.d_rg_skip, h->ps.d_rg_groups, h->ps.d_rg_groups + ng);
                SX_HIP(hipMemcpy(&h->ps.rg_ngroups, h->ps.d_rg_groups + ng, sizeof(int), hipMemcpyDeviceToHost));
            } else {
                (void)hipGetLastError();
                (void)hipFree(h->ps.d_rg_skip); h->ps.d_rg_skip = nullptr;
            }
            SX_HIP(hipDeviceSynchronize());
            // the blocks with a dictionary, in order (the register-resident kernel walks this list: every workgroup of its launch works,
            // so the contiguous chunks of workgroups the XCDs get stay balanced)
            std::vector<int> cnt((size_t)h->ps.plan_nblk), list;
            SX_HIP(hipMemcpy(cnt.data(), h->ps.d_dict_ptr, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost));
            for (int b = 0; b < h->ps.plan_nblk; ++b)
                if (cnt[(size_t)b] > 0) list.push_back(b);
            (void)hipFree(h->ps.d_dict_blocks); h->ps.d_dict_blocks = nullptr; h->ps.n_dict_blocks = 0;
            if (h->ps.d_rg_skip && !list.empty() && hipMalloc((void **)&h->ps.d_dict_blocks, sizeof(int) * list.size()) == hipSuccess) {
                SX_HIP(hipMemcpy(h->ps.d_dict_blocks, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice));
                h->ps.n_dict_blocks = (int)list.size();
            } else {
                (void)hipGetLastError();
                (void)hipFree(h->ps.d_rg_skip); h->ps.d_rg_skip = nullptr;   // (no split form)
            }
        } else {
            (void)hipGetLastError();   // (without the table the mixed kernel runs)
        }
    }
    return SEXTANS_OK;
}

// Clustered-order plan (see PanelState psc): built once per matrix, on the first WHOLE-matrix call that could use it.  Two forms:
//   cluster_state 1  grid bricks (row_cluster.hip): matrices with Cartesian-grid stencil structure in natural ordering; strides
//                    inferred from sampled rows, rows sorted brick by brick; C addressed through a slot -> row table, B untouched.
//                    A fast path: one radix sort, and hand-shaped bricks beat the general clustering by ~10 % in panel rows.
//   cluster_state 2  graph clustering (graph_cluster.hip) for everything else that has neighbourhood structure but not in its
//                    numbering: multilevel pairwise aggregation of the rows over the matrix graph, columns relabelled in
//                    first-touch order, B repacked into permuted panels and C staged block-major (reorder_kernels.h).
// Option "row_cluster": -1 = automatic (grid: when the clustered plan copies >= 15 % fewer B rows into LDS; graph: when the
// natural-order plan is missing or its blocks are cut short by the panel capacity, the sampled rows share neighbourhoods, and the
// reordered plan copies >= 40 % fewer B rows -- it pays two extra passes over C); 1 = whenever a clustered plan can be built;
// 2 = graph clustering even for grids; 0 = never.
// It is an optimisation: whatever goes wrong in here (allocation failures included) declines it and the SpMM runs on the
// natural-order forms it already has.
namespace {
// Short rows (<= 2 register-resident batches) whose block dictionaries all fit 320 rows: the clustered plan is packed for a 320-row
// panel, 20.5 KB of LDS per workgroup (spmm_csr_panel_v2<..., DCAP = 5>): a fifth workgroup per CU for launches that are latency-bound.
bool small_panel_fits(const sextans_engine *h, const sx::DevicePlan &dp) {
    return h->opt_small_panel != 0 && !dp.mixed && dp.max_dict <= 5 * 64 && h->M > 0 && h->m_nnz / h->M + 8 <= 48;
}
int cluster_grid(sextans_engine *h) {   // 0 = in use, 1 = declined
    if (!h->ps.plan_built || h->ps.plan_lpr != 4 || h->ps.plan_mixed) return 1;
    // ---- grid strides from the columns of ~128 rows out of the middle half of the matrix
    std::vector<int> rp;
    if (read_back_row_ptr(h, rp)) return 1;
    std::vector<int> rows;
    std::vector<std::vector<int>> cols;
    const int nsample = 128;
    for (int k = 0; k < nsample; ++k) {
        const int r = (int)((int64_t)h->M / 4 + (int64_t)k * (h->M / 2) / nsample);
        const int j0 = rp[(size_t)r], j1 = rp[(size_t)r + 1];
        if (j1 <= j0 || j1 - j0 > 4096) continue;
        std::vector<int> c((size_t)(j1 - j0));
        if (hipMemcpy(c.data(), h->m_ci + j0, sizeof(int) * c.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        rows.push_back(r);
        cols.push_back(std::move(c));
    }
    sx::GridStrides gs;
    if (!sx::detect_grid_strides(h->M, rows, cols, &gs)) return 1;
    h->cluster_s2 = gs.s2;
    h->cluster_s3 = gs.s3;
    // ---- bricks of <= 64 rows = one row block each: a run of 15 / 16 rows of a grid line x 2 lines x 2 planes (3-D), x 4 lines (2-D).
    // Runs of consecutive rows keep a wavefront's C accesses (its 16 row slots) on consecutive rows -- 64-byte runs per column as in
    // natural order (12-row runs gave away half of the gain at N = 128, where C is half of the traffic); the plan builder starts a
    // block at every brick (`cut`), so blocks and bricks coincide.
    int run_rows = 16, b2 = gs.s3 > 0 ? 2 : 4, b3 = gs.s3 > 0 ? 2 : 1;
    // Short rows (every row <= 32 entries = 2 register-resident batches; 3 batches x 2 sets spill at 128 registers): bricks of 128 rows -- 16 x 4 x 2 (3-D), 16 x 8 (2-D) -- as blocks of TWO 64-slot row
    // sets on one dictionary (spmm_panel_v2.h: SETS): 3.4 instead of 4.5 dictionary rows per matrix row on a 27-point mesh, and block
    // meta, prologue round trips and panel are paid once per 128 rows.  Falls back to 64-row bricks when a dictionary outgrows the panel.
    // (measured same-box, tools/exp_r04i.sh: 27-point 1-dof -3.5 % at N = 16, -4 % at N = 128; 2-D 9-point -2 .. -7 % at N = 16 but +3 .. +8 %
    // at N >= 32, where the tile loop re-reads both sets' panels: automatic for 3-D grids only, row_sets = 3 forces it for 2-D grids too)
    int sets = (h->M > 0 && h->ps.plan_max_row <= 32 && (h->opt_row_sets >= 3 || (h->opt_row_sets == 2 && gs.s3 > 0 && h->opt_cluster_shape == 0))) ? 2 : 1;
    if (h->opt_cluster_shape > 0) {   // measurement switch (SEXTANS_DEBUG_OPTIONS): run_rows * 10000 + b2 * 100 + b3, validated by set_option
        run_rows = (int)(h->opt_cluster_shape / 10000); b2 = (int)(h->opt_cluster_shape / 100 % 100); b3 = (int)(h->opt_cluster_shape % 100);
        if (run_rows < 1 || b2 < 1 || b3 < 1) return 1;
    }
    std::string err;
    int *d_perm = nullptr, *prp = nullptr, *pci = nullptr;
    unsigned char *d_cut = nullptr;
    float *pv = nullptr;
    sx::DevicePlan dp;
    auto drop = [&]() { (void)hipFree(d_perm); (void)hipFree(d_cut); (void)hipFree(prp); (void)hipFree(pci); (void)hipFree(pv); sx::free_device_plan(dp); return 1; };
    const int lpr = 4, RB = sx::kBlock / lpr, cap = kPanelFloats / (4 * lpr);
    const double min_reuse = std::min((double)h->opt_min_reuse_x100, (double)h->opt_min_reuse_wide_x100) / 100.0;
    int brc = 1, cap_used = cap;
    for (; sets >= 1; --sets) {
        (void)hipFree(d_perm); (void)hipFree(d_cut); (void)hipFree(prp); (void)hipFree(pci); (void)hipFree(pv);
        d_perm = prp = pci = nullptr; d_cut = nullptr; pv = nullptr;
        sx::free_device_plan(dp);
        const int l2 = sets == 2 && h->opt_cluster_shape == 0 ? (gs.s3 > 0 ? 4 : 8) : b2, l3 = sets == 2 && h->opt_cluster_shape == 0 ? (gs.s3 > 0 ? 2 : 1) : b3;   // (a measurement shape + row_sets = 3: that shape as two row sets)
        if (sx::build_brick_order_device(h->M, gs, run_rows, l2, l3, (int)h->opt_cluster_group, &d_perm, &d_cut, err)) return drop();
        if (sx::permute_csr_rows_device(h->M, h->m_nnz, h->m_rp, h->m_ci, h->m_v, d_perm, &prp, &pci, &pv, err)) return drop();
        cap_used = cap;
        brc = sx::build_panel_plan_device(h->M, h->K, prp, pci, pv, lpr, cap, min_reuse, dp, err, d_cut, h->opt_share_index != 0, sets);
        if (sets == 2 && (brc != 0 || dp.mixed || dp.max_dict > sx::kWideMaxDict || dp.dict_stride > 9 * RB)) continue;   // 128-row bricks do not fit the panel
        // ... which the builder reports as blocks cut by the capacity: it cuts a block when its dictionary is full, so an overflowing
        // 128-row brick comes back as uneven pieces, not as an error -- then 64-row bricks
        if (sets == 2 && dp.capacity_cuts > 0) continue;
        if (brc == 0 && sets == 1 && small_panel_fits(h, dp)) {   // short rows, small dictionaries: packed again for a 320-row panel (same blocks, less LDS)
            sx::free_device_plan(dp);
            cap_used = 5 * RB;
            brc = sx::build_panel_plan_device(h->M, h->K, prp, pci, pv, lpr, cap_used, min_reuse, dp, err, d_cut, h->opt_share_index != 0);
        }
        break;
    }
    if (sets < 1) sets = 1;
    (void)hipFree(prp); (void)hipFree(pci); (void)hipFree(pv); (void)hipFree(d_cut);
    prp = pci = nullptr; pv = nullptr; d_cut = nullptr;
    if (brc != 0) return drop();
    h->cluster_total_dict = dp.total_dict;
    const bool gain = (double)dp.total_dict <= 0.85 * (double)h->plan_total_dict;
    if (dp.mixed || dp.dict_stride > 9 * RB || dp.max_dict > sx::kWideMaxDict || (h->opt_row_cluster < 0 && !gain)) return drop();
    if (sx::build_slot_rows_device(dp.nblk, RB * dp.sets, dp.d_blk_row, d_perm, &h->d_slot_row, err)) return drop();
    (void)hipFree(d_perm);
    adopt_device_plan(h->psc, dp, h, lpr, cap_used);
    h->psc.plan_panel_frac = h->ps.plan_panel_frac;
    h->psc.plan_narrow_frac = h->ps.plan_narrow_frac;
    return 0;
}

// Run-level clustering (round 5; graph_cluster.hip: run_graph_device): the rows stay in runs of 16 consecutive rows -- a wavefront's 16
// row slots, so C is accessed exactly as in natural order and B needs no permutation -- and a row block is 4 runs chosen over the graph
// of runs.  For matrices whose numbering has locality but whose natural blocks are cut short by the panel capacity (real FEM files in
// their file order, RCM orders): the graph-clustered plan copies 25-40 % fewer panel rows there, not enough to pay for the reordered
// form's passes (decline 12); this form has no passes.  Uses the grid-brick machinery (slot -> row table, cluster_state 1).
int cluster_runs(sextans_engine *h) {   // 0 = in use, 1 = declined
    if (!h->ps.plan_built || h->ps.plan_lpr != 4 || h->ps.plan_mixed || h->M != h->K || h->M < 65536 || h->ps.plan_nblk <= 0) return 1;
    if ((double)h->M / h->ps.plan_nblk >= 56.0 || h->dense_W > 0) return 1;               // natural blocks (nearly) full: nothing to gain
    const int lpr = 4, RB = sx::kBlock / lpr, cap = kPanelFloats / (4 * lpr), run = 16;
    std::string err;
    int *g_rp = nullptr, *g_ci = nullptr, *s_rp = nullptr, *s_ci = nullptr, *d_order_r = nullptr, *d_order = nullptr, *prp = nullptr, *pci = nullptr;
    unsigned char *d_cut = nullptr, *g_w = nullptr, *s_w = nullptr;
    float *pv = nullptr;
    sx::DevicePlan dp;
    auto drop = [&]() { for (void *q : {(void *)g_rp, (void *)g_ci, (void *)s_rp, (void *)s_ci, (void *)d_order_r, (void *)d_order, (void *)prp, (void *)pci, (void *)d_cut, (void *)pv, (void *)g_w, (void *)s_w}) (void)hipFree(q);
                        sx::free_device_plan(dp); (void)hipGetLastError(); return 1; };
    int64_t g_nnz = 0, s_nnz = 0;
    int Mr = 0;
    if (sx::run_graph_device(h->M, run, h->m_rp, h->m_ci, &g_rp, &g_ci, &g_w, &g_nnz, &Mr, err)) return drop();
    if (sx::symmetrize_graph_device(Mr, g_nnz, g_rp, g_ci, g_w, &s_rp, &s_ci, &s_w, &s_nnz, err)) return drop();   // (the matching needs symmetric weights)
    // (blocks = the clusters of the level at which they hold up to 4 runs: cutting the final order every 4 runs would straddle them)
    int *d_group = nullptr;
    if (sx::cluster_rows_graph_device(Mr, Mr, s_nnz, s_rp, s_ci, (int)std::min<int64_t>(h->opt_cluster_top, 0x40000000), &d_order_r, err, s_w, RB / run, &d_group)) return drop();
    const int erc = sx::expand_run_order_device(h->M, Mr, run, RB / run, d_order_r, d_group, &d_order, &d_cut, err);
    (void)hipFree(d_group);
    if (erc) return drop();
    if (sx::permute_csr_rows_device(h->M, h->m_nnz, h->m_rp, h->m_ci, h->m_v, d_order, &prp, &pci, &pv, err)) return drop();
    const double min_reuse = std::min((double)h->opt_min_reuse_x100, (double)h->opt_min_reuse_wide_x100) / 100.0;
    if (sx::build_panel_plan_device(h->M, h->K, prp, pci, pv, lpr, cap, min_reuse, dp, err, d_cut, h->opt_share_index != 0) != 0) return drop();
    if (dp.mixed || dp.dict_stride > 9 * RB || dp.max_dict > sx::kWideMaxDict) return drop();
    h->cluster_total_dict = dp.total_dict;
    if (h->opt_run_cluster < 2 && (double)dp.total_dict > 0.9 * (double)h->plan_total_dict) return drop();   // (>= 10 % fewer panel rows; "run_cluster" = 2 keeps it regardless: measurements)
    if (sx::build_slot_rows_device(dp.nblk, RB, dp.d_blk_row, d_order, &h->d_slot_row, err)) return drop();
    adopt_device_plan(h->psc, dp, h, lpr, cap);
    h->psc.plan_panel_frac = h->ps.plan_panel_frac;
    h->psc.plan_narrow_frac = h->ps.plan_narrow_frac;
    for (void *q : {(void *)g_rp, (void *)g_ci, (void *)s_rp, (void *)s_ci, (void *)d_order_r, (void *)d_order, (void *)prp, (void *)pci, (void *)d_cut, (void *)pv, (void *)g_w, (void *)s_w}) (void)hipFree(q);
    return 0;
}

int cluster_graph(sextans_engine *h) {   // 0 = in use, else declined: the reason (stat "cluster_decline")
    // Which graph are the rows clustered over?
    //   (a) square matrix with a (nearly) symmetric pattern: the matrix itself, a column index read as the row of the neighbour;
    //   (b) a row slab of a square matrix whose position is known (option row_offset, set by sextans_dist_spmm): the slab's own
    //       square pattern, everything behind it on the rectangular matrix;
    //   (c) anything else -- rectangular (LP / least-squares matrices; the reference schedules any M x K matrix, sparse_helper.h:
    //       345-403) or square with an unsymmetric pattern (row c says little about row r there: measured on the holdout class with
    //       30 % of its lower entries dropped, graph (a) gave 2 x the panel rows of the symmetric pattern, 4 x under a random
    //       numbering) -- the ROW-SIMILARITY graph: r joined to the rows that share the most columns with it (graph_cluster.hip).
    // Option "row_similarity": -1 automatic, 0 never (rectangular matrices are declined as before round 5), 1 always.
    const bool slab = h->M != h->K && h->opt_row_offset >= 0 && h->opt_row_offset + h->M <= h->K;
    const bool rect = h->M != h->K && !slab;
    if (h->m_nnz <= 0 || (rect && h->opt_row_similarity == 0)) return 1;                 // 1: not square (and the row-similarity graph switched off)
    if (h->dense_W > 0) return 2;                                                       // 2: dense tiles on the MFMA path (rows on the piece / chain paths are fine: they are empty here)
    if ((int64_t)h->K * 64 >= ((int64_t)1 << 32)) return 3;   // 3: 32-bit byte offsets into a K x 16 panel
    const int lpr = 4, RB = sx::kBlock / lpr, cap = kPanelFloats / (4 * lpr);
    std::string err;
    int *g_rp_own = nullptr, *g_ci_own = nullptr;      // the graph the rows are clustered over
    unsigned char *g_w_own = nullptr;                  // ... and its weights, when it brings its own
    const int *g_rp = h->m_rp, *g_ci = h->m_ci;
    int64_t g_nnz = h->m_nnz;
    struct FreeGraph { int *&a, *&b; unsigned char *&c; ~FreeGraph() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); } } free_graph{g_rp_own, g_ci_own, g_w_own};
    // worth trying?  Not when the natural-order plan already fills its blocks (numberings with locality: the grid path or nothing)
    if (h->opt_row_cluster < 0 && h->ps.plan_built && !h->ps.plan_mixed && h->ps.plan_nblk > 0 && (double)h->M / h->ps.plan_nblk >= 50.0) return 4;   // 4: natural blocks are full
    if (slab) {
        if (sx::local_square_pattern_device(h->M, h->m_rp, h->m_ci, (int)h->opt_row_offset, &g_rp_own, &g_ci_own, &g_nnz, err)) return 6;
        g_rp = g_rp_own; g_ci = g_ci_own;
    }
    double shared = 0.0, near = 0.0, sym = 1.0;
    bool rowsim = rect || h->opt_row_similarity > 0;
    if (!rowsim) {
        if (sx::probe_shared_neighbourhood_device(h->M, g_rp, g_ci, 256, &shared, &near, err, &sym)) return 5;
        h->pattern_symmetry = sym;
    }
    auto adopt = [&](int *rp, int *ci, unsigned char *w, int64_t n) {   // replaces the graph owned so far
        (void)hipFree(g_rp_own); (void)hipFree(g_ci_own); (void)hipFree(g_w_own);
        g_rp_own = rp; g_ci_own = ci; g_w_own = w; g_rp = rp; g_ci = ci; g_nnz = n;
    };
    if (rowsim) {
        int *rp = nullptr, *ci = nullptr;
        unsigned char *w = nullptr;
        int64_t n = 0;
        if (sx::row_similarity_graph_device(h->M, h->K, h->m_nnz, h->m_rp, h->m_ci, &rp, &ci, &w, &n, &shared, &near, err)) return 6;
        adopt(rp, ci, w, n);
    }
    if (rowsim || sym < 0.98) {   // the matching needs symmetric weights: G + G^T (the kNN lists of the row-similarity graph are not mutual either)
        int *rp = nullptr, *ci = nullptr;
        unsigned char *w = nullptr;
        int64_t n = 0;
        if (sx::symmetrize_graph_device(h->M, g_nnz, g_rp, g_ci, g_w_own, &rp, &ci, g_w_own ? &w : nullptr, &n, err)) return 6;
        adopt(rp, ci, w, n);
    }
    h->cluster_graph_kind = rowsim ? 2 : sym < 0.98 ? 3 : slab ? 1 : 0;
    h->cluster_shared = shared;
    if (h->opt_row_cluster < 0) {
        // ... and not when neighbouring rows do not share neighbourhoods (random columns: there is nothing to find)
        if (shared < 0.2) return 5;                                                                                               // 5: no shared neighbourhoods
        // 13: short rows in a numbering that has locality -- per row the two extra passes over C move more bytes than the row's
        // non-zeros, and the gather kernel finds its B rows in L2 (2-D 5-point stencils, 1-dof meshes in sweep order: measured equal
        // or slower per step); without locality in the numbering the reordered form wins even there (1-dof mesh, random order: 1.7x)
        if (h->m_nnz / h->M < 20 && near >= 0.5) return 13;
    }
    int *d_order = nullptr, *d_colpos = nullptr, *prp = nullptr, *pci = nullptr;
    float *pv = nullptr;
    unsigned char *d_cut = nullptr;
    sx::DevicePlan dp;
    auto drop = [&](int why) { (void)hipFree(d_order); (void)hipFree(d_colpos); (void)hipFree(d_cut); (void)hipFree(prp); (void)hipFree(pci); (void)hipFree(pv);
                               sx::free_device_plan(dp); return why; };
    if (sx::cluster_rows_graph_device(h->M, h->M, g_nnz, g_rp, g_ci, (int)std::min<int64_t>(h->opt_cluster_top, 0x40000000), &d_order, err, g_w_own)) return drop(6);   // 6 .. 9: a builder failed
    if (h->opt_refine_sweeps > 0)   // boundary rows to the block that holds more of their neighbours (blocks of 62 rows, room for 2 more each)
        if (sx::refine_blocks_device(h->M, g_rp, g_ci, d_order, (int)std::min<int64_t>(RB, std::max<int64_t>(32, h->opt_refine_rows)), RB, (int)h->opt_refine_sweeps, &d_cut, err) == 2) return drop(6);
    const bool relabel = h->opt_relabel_columns != 0;
    if (relabel && sx::column_first_touch_order_device(h->M, h->K, h->m_rp, h->m_ci, d_order, &d_colpos, err)) return drop(7);
    if (sx::permute_csr_rows_device(h->M, h->m_nnz, h->m_rp, h->m_ci, h->m_v, d_order, &prp, &pci, &pv, err)) return drop(8);
    if (relabel && sx::relabel_columns_device(h->m_nnz, pci, d_colpos, err)) return drop(8);
    // Every block that fits gets a dictionary (threshold 0): the tail of the order holds the rows nothing wanted to merge with, and
    // ONE block of such rows without reuse would make the whole plan "mixed".  The reuse test is made on the plan as a whole below.
    const double min_reuse = std::min((double)h->opt_min_reuse_x100, (double)h->opt_min_reuse_wide_x100) / 100.0;
    int brc = sx::build_panel_plan_device(h->M, h->K, prp, pci, pv, lpr, cap, 0.0, dp, err, d_cut, h->opt_share_index != 0);
    int cap_used = cap;
    if (brc == 0 && small_panel_fits(h, dp)) {
        sx::free_device_plan(dp);
        cap_used = 5 * RB;
        brc = sx::build_panel_plan_device(h->M, h->K, prp, pci, pv, lpr, cap_used, 0.0, dp, err, d_cut, h->opt_share_index != 0);
    }
    (void)hipFree(prp); (void)hipFree(pci); (void)hipFree(pv); (void)hipFree(d_cut);
    prp = pci = nullptr; pv = nullptr; d_cut = nullptr;
    if (brc != 0) return drop(9);
    h->cluster_total_dict = dp.total_dict;
    // 10: a row wider than the panel / limits of the 32-bit offsets; 11: the reordered plan has no reuse either; 12: not enough gain
    if (dp.mixed || dp.dict_stride > 9 * RB || dp.max_dict > sx::kWideMaxDict || (int64_t)h->M * 64 >= ((int64_t)1 << 32)) return drop(10);
    if ((double)h->m_nnz < min_reuse * (double)dp.total_dict) return drop(11);
    {   // automatic: worth the two extra passes over C only if it copies >= 40 % fewer B rows than what would run otherwise -- the
        // natural-order plan, or the grid-brick plan when the grid path built one first (ensure_cluster_plan)
        // Row-major calls (sextans_spmm_device_rm) pay no pass for it: there a plan that copies >= 25 % fewer B rows is kept (measured on
        // the holdout class: 36 % fewer panel rows = kernel 686 -> 593 us), for those calls only ("cluster_cm_pays").
        const int64_t ref = h->cluster_ref_dict > 0 ? h->cluster_ref_dict : (h->ps.plan_built ? h->plan_total_dict : 0);
        const bool auto_ref = h->opt_row_cluster < 0 && ref > 0;
        if (auto_ref && (double)dp.total_dict > (h->cluster_for_rm ? 0.75 : 0.6) * (double)ref) return drop(12);
        h->cluster_cm_pays = !(auto_ref && (double)dp.total_dict > 0.6 * (double)ref);
    }
    if (sx::build_slot_rows_device(dp.nblk, RB, dp.d_blk_row, d_order, &h->d_slot_row, err)) return drop(9);
    (void)hipFree(d_order);
    const double covered = h->m_nnz ? (double)dp.nnz_in_panel_blocks / (double)h->m_nnz : 0.0;
    adopt_device_plan(h->psc, dp, h, lpr, cap_used);
    h->psc.plan_panel_frac = covered;
    h->psc.plan_narrow_frac = h->psc.plan_panel_frac;
    h->d_colpos = d_colpos;
    return 0;
}
}  // namespace

// Columns the matrix as set has entries in.  A call repacks only those rows of B: on one GPU that is all of B, but a rank of a
// row-partitioned SpMM over a banded / mesh matrix touches 1 / world of B plus a halo, and the replicated repack was what capped
// the compute phase of the FEM class at 4.2 x on 8 ranks (profiles/r04_rank_slab_times.json).  Rounded outwards to 64 rows.
int ensure_col_range(sextans_engine *h) {
    if (h->col_range_known) return SEXTANS_OK;
    h->col_lo = 0; h->col_hi = h->K;
    h->col_range_known = true;
    if (h->nnz <= 0 || !h->d_ci) return SEXTANS_OK;
    if (!h->owns_matrix && !h->device_matrix_checked) {   // a device matrix nobody has validated yet: its indices are not trusted here
        int bad = 0;
        std::string verr;
        if (sx::validate_csr_device(h->M, h->K, h->nnz, h->d_rp, h->d_ci, &bad, verr)) { g_last_error = verr; return SEXTANS_ERR_HIP; }
        if (bad) return (bad & 1) ? SEXTANS_ERR_INVALID : SEXTANS_ERR_INDEX;
        h->device_matrix_checked = true;
    }
    int lo = 0, hi = -1;
    std::string err;
    if (sx::column_range_device(h->nnz, h->d_ci, &lo, &hi, err)) { (void)hipGetLastError(); return SEXTANS_OK; }
    if (hi >= lo) {
        h->col_lo = std::max(0, lo / 64 * 64);
        h->col_hi = std::min(h->K, (hi / 64 + 1) * 64);
    }
    // ... and inside that range only the 64-row segments that hold one of the matrix's columns (round 5): the slab of a renumbered
    // mesh reaches a handful of far-away columns, which stretched [col_lo, col_hi) over all of B -- 85 us of repack per rank at 8
    // ranks against 17 us for its share (profiles/r05_rank_slab_times.json).  Kept only when it skips at least an eighth.
    {
        unsigned char *f = nullptr;
        int64_t touched = 0;
        if (sx::column_touch_flags_device(h->K, h->nnz, h->d_ci, &f, &touched, err) == 0) {
            const int64_t in_range = (h->col_hi - h->col_lo + 63) / 64;
            if (touched * 8 <= in_range * 7) { h->d_touched = f; h->touched_segments = touched; }
            else (void)hipFree(f);
        } else (void)hipGetLastError();
    }
    return SEXTANS_OK;
}

// Short rows in a numbering with locality: the lane-per-row kernel on the caller's column-major B (spmm_colwise_kernel.h).
int ensure_colwise(sextans_engine *h) {
    if (h->colwise_state != 0) return SEXTANS_OK;
    h->colwise_state = -1;
    if (h->M < 2 || h->m_nnz <= 0 || h->m_nnz / h->M > h->opt_colwise_max_len) return SEXTANS_OK;
    std::string err;
    double close = 0.0;
    if (sx::probe_row_coherence_device(h->M, h->m_rp, h->m_ci, 4096, &close, err)) { (void)hipGetLastError(); return SEXTANS_OK; }
    h->row_coherence = close;
    if (close >= 0.7) h->colwise_state = 1;
    return SEXTANS_OK;
}

// The natural-order plan (4 lanes per row) keeps its meta data -- block boundaries for sextans_align_row, dictionary sizes --
// but its packed stream, 6 bytes per non-zero (1.9 GB for the 318 M-nnz FEM matrix), is only read by row-range calls and by
// whole-matrix calls that cannot use the clustered plan (column-major staging of small B).  Once a clustered plan serves the
// whole-matrix calls of a matrix too large for that staging, the stream is handed back; the first launch that needs it again
// rebuilds it (the builder is deterministic: the same bytes; 75 ms for that matrix) -- a row-range call issued inside the caller's
// own hipGraph capture right after a whole-matrix call would therefore allocate inside the capture: issue one outside first.
void release_plan_streams(sextans_engine *h) {
    sextans_engine::PanelState &p = h->ps;
    if (!p.plan_built || p.stream_released || p.plan_lpr != 4) return;
    if ((size_t)h->K * 16 * sizeof(float) <= ((size_t)16 << 20)) return;   // column-major staging may still pick the natural-order plan
    (void)hipFree(p.d_lidx); (void)hipFree(p.d_pval); (void)hipFree(p.d_pcol32); (void)hipFree(p.d_ioff);
    p.d_lidx = nullptr; p.d_pval = nullptr; p.d_pcol32 = nullptr; p.d_ioff = nullptr;
    p.stream_released = true;
}

int restore_plan_streams(sextans_engine *h) {
    sextans_engine::PanelState &p = h->ps;
    if (!p.stream_released) return SEXTANS_OK;
    PlanTimer timer(h);
    sx::DevicePlan dp;
    std::string err;
    const int lpr = p.plan_lpr, cap = kPanelFloats / (4 * lpr);
    const double min_reuse = std::min((double)h->opt_min_reuse_x100, (double)h->opt_min_reuse_wide_x100) / 100.0;
    const int brc = sx::build_panel_plan_device(h->M, h->K, h->m_rp, h->m_ci, h->m_v, lpr, cap, p.plan_all_dict ? 0.0 : min_reuse, dp, err, nullptr,
                                                lpr == 4 && h->opt_share_index != 0);
    if (brc != 0 || dp.nblk != p.plan_nblk || dp.stream_len != p.plan_stream_len) {
        sx::free_device_plan(dp);
        if (brc == 2) g_last_error = err;
        return brc == 2 ? SEXTANS_ERR_HIP : SEXTANS_ERR_STATE;
    }
    p.d_lidx = dp.d_idx16; p.d_pval = dp.d_val; p.d_pcol32 = dp.d_col32; p.d_ioff = dp.d_ioff;
    p.plan_idx_len = dp.idx_len;
    dp.d_idx16 = nullptr; dp.d_val = nullptr; dp.d_col32 = nullptr; dp.d_ioff = nullptr;
    sx::free_device_plan(dp);
    p.stream_released = false;
    return SEXTANS_OK;
}

__global__ __launch_bounds__(256) void compact_chain_entries(const int *__restrict__ cbeg, const long long *__restrict__ coff, const int *__restrict__ ci,
                                                             const float *__restrict__ va, const int *__restrict__ colpos, int *__restrict__ out_ci,
                                                             float *__restrict__ out_v) {
    const int i = blockIdx.x;
    const long long o = coff[i], len = coff[i + 1] - o;
    const int b = cbeg[i];
    for (long long e = threadIdx.x; e < len; e += 256) { out_ci[o + e] = colpos[ci[b + e]]; out_v[o + e] = va[b + e]; }
}

int ensure_cluster_plan(sextans_engine *h) {
    if (h->cluster_state != 0) return SEXTANS_OK;
    h->cluster_state = -1;
    if (h->opt_row_cluster == 0 || h->M < 4096) return SEXTANS_OK;
    PlanTimer timer(h);
    h->cluster_ref_dict = 0;
    if (h->opt_row_cluster != 2 && cluster_grid(h) == 0) {
        h->cluster_state = 1;
        // A grid whose rows reach into several far-apart column ranges (DOF-MAJOR numbering: all x unknowns, then y, then z -- a row of
        // the 3-dof FEM matrix has its 81 columns in three blocks of the matrix) gets bricks of one unknown each, whose dictionaries are
        // three times the node-major ones: blocks of ~28 rows, 20 dictionary rows per matrix row.  The graph clustering puts the rows of
        // a node back together whatever the numbering: measured on the 3M-row matrix 860 us per step with the bricks, 720 reordered
        // (node-major order: 530).  So when the brick plan's blocks are cut short by the dictionary, the graph plan is built as well and
        // kept if it copies >= 40 % fewer B rows than the bricks.
        const double rows_per_block = (double)h->M / std::max(1, h->psc.plan_nblk);
        if (h->opt_row_cluster < 0 && rows_per_block < 40.0 * h->psc.plan_sets) {
            sextans_engine::PanelState grid = std::move(h->psc);
            int *grid_slot_row = h->d_slot_row;
            const int64_t grid_dict = h->cluster_total_dict;
            h->psc = sextans_engine::PanelState();
            h->d_slot_row = nullptr;
            h->cluster_ref_dict = grid_dict;
            const int why = cluster_graph(h);
            (void)hipGetLastError();
            if (why == 0) {
                free_panel_state(grid);
                (void)hipFree(grid_slot_row);
                h->cluster_state = 2;
            } else {   // keep the bricks
                free_panel_state(h->psc);
                (void)hipFree(h->d_slot_row); (void)hipFree(h->d_colpos);
                h->d_colpos = nullptr;
                h->psc = std::move(grid);
                h->d_slot_row = grid_slot_row;
                h->cluster_total_dict = grid_dict;
                h->cluster_decline = why;
            }
        }
    } else if ((h->cluster_decline = cluster_graph(h)) == 0) h->cluster_state = 2;
    else if (h->cluster_decline == 12 && h->opt_row_cluster < 0 && h->opt_run_cluster != 0 && cluster_runs(h) == 0) {
        h->cluster_state = 1;     // the slot -> row machinery of the grid bricks; the graph plan stays declined (12) for column-major calls
        h->cluster_runs = true;
    }
    if (h->cluster_state == 2 && h->nchain > 0 && h->d_colpos) {
        // the exact-chain kernels read B rows by column index: for the permuted panels of the reordered form they get the chain rows'
        // entries once more, compact, with relabelled columns (a few thousand entries)
        const long long total = h->h_chain_off.empty() ? 0 : h->h_chain_off.back();
        std::vector<int> beg((size_t)h->nchain);
        for (int i = 0; i < h->nchain; ++i) beg[(size_t)i] = (int)h->h_chain_off[(size_t)i];
        bool ok = total > 0 && total < 0x7fffffffLL && hipMalloc((void **)&h->d_chain_ci_perm, sizeof(int) * (size_t)total) == hipSuccess &&
                  hipMalloc((void **)&h->d_chain_v_c, sizeof(float) * (size_t)total) == hipSuccess && upload(&h->d_chain_beg_c, beg) == SEXTANS_OK;
        if (ok) {
            hipLaunchKernelGGL(compact_chain_entries, dim3((unsigned)h->nchain), dim3(256), 0, nullptr, h->d_chain_beg, h->d_chain_off, h->s_ci, h->s_v,
                               h->d_colpos, h->d_chain_ci_perm, h->d_chain_v_c);
            ok = hipDeviceSynchronize() == hipSuccess;
        }
        if (!ok) {   // (declined like every other failure in here)
            free_cluster_plan(h);
            h->cluster_state = -1;
            h->cluster_decline = 9;
        }
    }
    if (h->cluster_state == 1 || (h->cluster_state == 2 && h->cluster_cm_pays)) release_plan_streams(h);
    (void)hipGetLastError();   // a failure in here (out of memory for the sort buffers, ...) only declines the clustered plan
    return SEXTANS_OK;
}

// Kernels that need more than the default 64 KiB of dynamic LDS: raise the limit once per (engine = device, kernel).
int allow_big_lds(sextans_engine *h, const void *kern, int bytes) {
    if (h->big_lds_kernels.count(kern)) return SEXTANS_OK;
    SX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    h->big_lds_kernels.insert(kern);
    return SEXTANS_OK;
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

static int ensure_split_rows(sextans_engine *h);
int ensure_split(sextans_engine *h) {
    const bool fresh = !(h->split_built_opt == h->opt_split_rows && h->bucket_built_opt == h->opt_bucket_rows &&
                         h->split_built_gnnz == h->opt_global_nnz && h->chain_built_opt == h->opt_exact_chain);
    if (int rc = ensure_split_rows(h)) return rc;
    if (fresh && h->rb_n > 0) return mark_rowblock_skip(h);   // rows of blocks routed to the fp32 matrix cores: never written by the CSR kernels
    return SEXTANS_OK;
}
static int ensure_split_rows(sextans_engine *h) {
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
    const bool auto_L0 = L0 < 0;
    if (T == 0) T = INT64_MAX;                 // never split
    if (L0 == 0) L0 = std::min(T, Tc);         // no bucketing: only rows that must be split / chained leave
    if (!auto_L0 && L0 > std::min(T, Tc)) L0 = std::min(T, Tc);
    if (!auto_L0 && L0 == INT64_MAX) return SEXTANS_OK;
    PlanTimer timer(h);
    std::vector<int> rp;
    if (int rc = read_back_row_ptr(h, rp, 1)) return rc;
    int64_t nonempty_rows = 0;
    for (int r = 0; r < h->M; ++r) nonempty_rows += rp[(size_t)r + 1] > rp[(size_t)r];
    if (auto_L0) {   // twice the mean length of the NON-EMPTY rows (a FEM matrix with half of its rows emptied has no long rows for that:
                     // with the mean over all rows every row went to the piece kernel, 913 us per step instead of 351)
        L0 = std::max<int64_t>(32, 2 * (h->s_nnz / std::max<int64_t>(1, nonempty_rows)));
        if (L0 > std::min(T, Tc)) L0 = std::min(T, Tc);
    }
    bool rare_long = false;
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
        // ... and nothing to do either when the rows above L0 are no outliers (a quarter of the non-empty rows or more: bimodal lengths):
        // bucketing would move most of the work to the slower kernel
        if (longest <= std::min(T, Tc) && h->opt_bucket_rows < 0 && (int64_t)rows.size() * 4 >= nonempty_rows) return SEXTANS_OK;
        if (longest <= std::min(T, Tc) && h->opt_bucket_rows < 0 && long_nnz * 50 < h->s_nnz) {
            // ... unless a row cannot fit an LDS panel at all (more entries than the dictionary holds): ONE such row in a mesh matrix
            // makes its block a direct block, the plan "mixed", takes every clustered plan and the register-resident kernel away from the
            // other 4 M rows, and its workgroup runs thousands of entries alone (1.5M-row 3-dof FEM, N = 16: 267 us per step; with three
            // 1000-entry rows 408, with one 5000-entry row 547; renumbered 357 -> 680).  Those rows alone leave for the piece path (one
            // piece each: same order, same rounding).
            constexpr int64_t kPanelRowLimit = 512;
            if (longest <= kPanelRowLimit) return SEXTANS_OK;
            L0 = kPanelRowLimit;
            std::vector<int> keep;
            for (int r : rows)
                if ((int64_t)rp[(size_t)r + 1] - rp[(size_t)r] > L0) keep.push_back(r);
            rows.swap(keep);
            rare_long = true;
        }
    }
    // ... as exact chains where the strict order allows them (the default): a single piece of 5000 entries is one lane group's serial
    // gather loop (~60 ns per entry: 300 us, the tail of the whole SpMM), a chain forms the products in parallel and adds them at
    // ~1.2 ns each
    const int64_t Tc_rows = (rare_long && Tc != INT64_MAX) ? std::min<int64_t>(Tc, L0) : Tc;
    // chain rows leave the piece tables
    std::vector<int> chain_rows, piece_rows;
    for (int r : rows) ((int64_t)rp[(size_t)r + 1] - rp[(size_t)r] > Tc_rows ? chain_rows : piece_rows).push_back(r);
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
        h->chain_T = Tc_rows;
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

// Everything that may allocate or run host-side preprocessing for an N-column SpMM: B-panel workspace,
// N-tile plan, and (for kernel != 1) the packed row-bucketed form of A.  Idempotent; called by
// sextans_spmm_device2 and, ahead of the timed region, by sextans_spmm_host.
int prepare(sextans_engine *h, int N, std::vector<Seg> &plan, int &W, bool &use_panel, bool &use_window, bool whole) {
    if (int rc = ensure_col_range(h)) return rc;
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
    if (h->nhub > 0)   // (16-column granularity: N = 16 t + 8 may run its tail as a half-empty 16-column tile)
        if (int rc = ensure(&h->d_P, &h->P_cap, (size_t)h->split_nv * (size_t)((N + 15) / 16 * 16))) return rc;
    if (h->nchain > 0 || (N >= 32 && h->opt_pipeline_tiles != 0)) {
        if (!h->ev_pipe[0])
            for (hipEvent_t &e : h->ev_pipe) SX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (!h->aux_stream) {
            SX_HIP(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));   // (a high-priority stream was measured: no difference)
            SX_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            SX_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
    }
    const size_t n16 = (size_t)((N + 15) / 16) * 16;   // (N = 16 t + 8: room for the tail as a zero-padded 16-column tile)
    if (!h->lean_prepare && (h->Bp_cap < (size_t)h->K * n16 || !h->d_Bp)) h->bp_layout = 0;   // new workspace: nothing to reuse
    if (!h->lean_prepare)
        if (int rc = ensure(&h->d_Bp, &h->Bp_cap, (size_t)h->K * n16)) return rc;
    // main tile width: 4*lanes_per_row, but never wider than N itself (N = 8 -> 2 lanes per row); then
    // 16- and 8-wide tiles for the remainder (N is a multiple of 8, the reference's N-tile
    // granularity: sextans.cpp:57-60).
    // "lanes_per_row" 0 = auto: 4 lanes (16-column tiles) for the panel kernel -- measured best on the FEM class
    // (config 3, N=128: 23 us with 4 lanes, 31 us with 8) -- and 8 lanes (32-column tiles) for the gather kernel
    // once N >= 32: a 128-byte B row is one fabric request where two 64-byte tiles are two (uniform 4M matrix,
    // N = 32/64/128: 3.15/6.9/15.2 ms with 8 lanes against 6.1/12.6/27.7 ms with 4).
    int lpr = h->opt_lpr ? (int)h->opt_lpr : 4;
    // N = 8 (the reference's own N tile) on a large matrix without long rows: the 16-column plan and the register-resident kernel
    // with a half-empty tile (spmm_panel_v2.h: last_cols) beat the 8-column panel kernel (4M-row FEM: 733 -> ~620 us per step) and
    // save the second packed plan; small B (column-major staging) and matrices with rows on the piece / chain paths keep 2 lanes per row
    bool n8_wide = N == 8 && !h->opt_lpr && h->opt_kernel == 0 && h->opt_panel_v2 != 0 && h->opt_cols_per_lane != 8 && h->nhub == 0 && h->nchain == 0 &&
                   h->dense_W == 0 && (size_t)h->K * 8 * sizeof(float) > ((size_t)16 << 20) && (int64_t)h->K * 64 < ((int64_t)1 << 32) &&
                   h->M > 0 && h->m_nnz / h->M >= 48;   // (shorter rows: measured a loss -- 2-D 9-point x 2 dof 0.48 -> 0.37 of the roofline per step, 43-entry mesh rows 0.38 -> 0.33)
    auto tiles = [&]() {
        while (lpr > 2 && 4 * lpr > N && !(n8_wide && lpr == 4)) lpr /= 2;
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
    if ((h->opt_kernel == 0 || h->opt_kernel == 4) && h->m_nnz > 0)
        if (int rc = ensure_colwise(h)) return rc;
    const bool colwise = h->opt_kernel == 4 || (h->opt_kernel == 0 && h->colwise_state == 1);
    if (h->opt_kernel != 1 && h->opt_kernel != 4 && h->m_nnz > 0 && !(colwise && h->nhub == 0 && h->nchain == 0 && h->dense_W == 0)) {
        if (int rc = ensure_plan(h, lpr, h->opt_kernel == 2)) return rc;
        // (here, not at launch time: prepare() runs before a hipGraph capture starts, and the builders copy to the host and
        // allocate.  Only for whole-matrix calls: engines that serve row ranges -- the chunks of the multi-GPU pipeline -- never
        // use the clustered plan and do not pay for it.)
        if (lpr == 4 && whole && h->opt_kernel != 3) {
            if (int rc = ensure_cluster_plan(h)) return rc;
            if (h->cluster_state == 2 && h->cluster_cm_pays && (N >= 16 || n8_wide) && !h->lean_prepare)   // row-major C staging of the reordered form: ceil(N / 16) tiles of M x 16
                if (int rc = ensure(&h->d_Cs, &h->Cs_cap, (n16 / 16) * (size_t)h->M * 16)) return rc;
        }
        use_panel = h->ps.plan_built && ((h->opt_kernel == 2) || (h->ps.plan_panel_frac >= 0.5 && (N >= 32 || h->ps.plan_narrow_frac >= 0.5)));
        // whole-matrix calls on a mixed plan that has its split form: from kSplitMinFrac on (FEM rows + 3 x / 6 x as many uniformly random
        // rows, panel share 0.40 / 0.25: measured against the gather kernel alone, profiles/r05_mixed_plan_split.txt)
        if (!use_panel && whole && lpr == 4 && h->opt_kernel == 0 && h->ps.plan_built && h->ps.plan_mixed && h->ps.d_rg_skip && h->opt_split_mixed != 0 &&
            h->opt_panel_v2 != 0 && h->ps.plan_max_dict <= sx::kWideMaxDict && h->ps.plan_panel_frac >= kSplitMinFrac &&
            (N >= 32 || h->ps.plan_narrow_frac >= kSplitMinFrac))
            use_panel = true;
        // (a matrix that runs in the reordered form needs no natural-order plan for that: a randomly numbered mesh without any reuse between
        // consecutive rows has none -- its N = 8 calls fell to the gather kernel, 0.06 against 0.37 at N = 16 on the holdout class)
        const bool reorder8 = whole && h->cluster_state == 2 && h->cluster_cm_pays && h->opt_kernel != 1 && h->opt_kernel != 3;
        if (n8_wide && !reorder8 && !(use_panel && !h->ps.plan_mixed && h->ps.plan_max_dict <= sx::kWideMaxDict)) {   // not a case for the wide kernel after all
            n8_wide = false;
            lpr = 4;
            tiles();
            if (int rc = ensure_plan(h, lpr, h->opt_kernel == 2)) return rc;
            use_panel = h->ps.plan_built && ((h->opt_kernel == 2) || (h->ps.plan_panel_frac >= 0.5 && h->ps.plan_narrow_frac >= 0.5));
        }
    }
    // (the reordered form of a graph-clustered matrix runs 16-column tiles whether or not a natural-order plan exists)
    // (a lean prepare -- on behalf of a row-major call -- also keeps 16-column tiles for a clustered plan that is kept for row-major calls
    // only: sextans_spmm_device_rm needs W == 16 to use it, and at N >= 32 the switch to 8 lanes per row below took it to the gather kernel)
    const bool reorder = whole && h->cluster_state == 2 && (h->cluster_cm_pays || h->lean_prepare) && h->opt_kernel != 1 && h->opt_kernel != 3 && lpr == 4;
    if (!h->opt_lpr && !use_panel && !reorder && N >= 32 && lpr != 8) { lpr = 8; tiles(); }
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

}  // namespace sxe
