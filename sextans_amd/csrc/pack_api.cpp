// pack_api.cpp -- host-side C ABI around the packed row-bucketed form of A (panel_plan.h): the
// engine's counterpart of the reference's packed non-zero stream (generate_edge_list_for_all_PEs +
// edge_list_64bit, sparse_helper.h:292-473).  Lets callers and CPU tests build, inspect and decode the
// exact arrays the LDS-panel kernel consumes.
#include <cstdlib>
#include <cstring>

#include "panel_plan.h"
#include "window_plan.h"
#include "sextans_amd.h"

namespace {
template <class T> T *dup(const std::vector<T> &v) {
    T *p = (T *)malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
    if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}
constexpr int kPanelBytes = 36 * 1024;   // must match kPanelFloats * 4 in engine.hip
}  // namespace

extern "C" {

int sextans_pack_csr(int M, int K, const int *row_ptr, const int *col_idx, const float *val,
                     int lanes_per_row, int min_reuse_x100, sextans_packed *out) {
    if (!out || M < 0 || K < 0 || !row_ptr || (lanes_per_row != 2 && lanes_per_row != 4 && lanes_per_row != 8))
        return SEXTANS_ERR_INVALID;
    if (row_ptr[0] != 0) return SEXTANS_ERR_INVALID;
    for (int r = 0; r < M; ++r)
        if (row_ptr[r + 1] < row_ptr[r]) return SEXTANS_ERR_INVALID;
    const int64_t nnz = M > 0 ? row_ptr[M] : 0;
    for (int64_t j = 0; j < nnz; ++j)
        if (col_idx[j] < 0 || col_idx[j] >= K) return SEXTANS_ERR_INDEX;
    sx::PanelPlan p;
    sx::build_panel_plan(M, K, row_ptr, col_idx, val, 256 / lanes_per_row,
                         kPanelBytes / (16 * lanes_per_row), (double)min_reuse_x100 / 100.0, p);
    memset(out, 0, sizeof *out);
    out->M = M; out->K = K; out->nnz = nnz;
    out->lanes_per_row = lanes_per_row;
    out->nblk = (int)p.blk_row.size() - 1;
    out->stream_len = (int64_t)p.val.size();
    out->max_dict = p.max_dict;
    out->nnz_in_panel_blocks = p.nnz_in_panel_blocks;
    out->blk_row = dup(p.blk_row); out->dict_ptr = dup(p.dict_ptr); out->dict = dup(p.dict);
    out->row_off = dup(p.row_off); out->idx16 = dup(p.idx16); out->col32 = dup(p.col32);
    out->val = dup(p.val);
    if (!out->blk_row || !out->dict_ptr || !out->dict || !out->row_off || !out->idx16 || !out->col32 ||
        !out->val) {
        sextans_packed_free(out);
        return SEXTANS_ERR_ALLOC;
    }
    return SEXTANS_OK;
}

void sextans_packed_free(sextans_packed *p) {
    if (!p) return;
    free(p->blk_row); free(p->dict_ptr); free(p->dict); free(p->row_off); free(p->idx16); free(p->col32);
    free(p->val);
    memset(p, 0, sizeof *p);
}

int sextans_unpack_csr(const sextans_packed *p, const int *row_ptr, int *col_idx, float *val) {
    if (!p || !row_ptr || (p->nnz > 0 && (!col_idx || !val))) return SEXTANS_ERR_INVALID;
    for (int b = 0; b < p->nblk; ++b) {
        const int u0 = p->dict_ptr[b], nu = p->dict_ptr[b + 1] - u0;
        for (int r = p->blk_row[b]; r < p->blk_row[b + 1]; ++r) {
            int64_t o = p->row_off[r];
            for (int j = row_ptr[r]; j < row_ptr[r + 1]; ++j, ++o) {
                val[j] = p->val[o];
                if (nu > 0) {
                    if (p->idx16[o] >= nu) return SEXTANS_ERR_INDEX;
                    col_idx[j] = p->dict[u0 + p->idx16[o]];
                } else {
                    col_idx[j] = p->col32[o];
                }
            }
        }
    }
    return SEXTANS_OK;
}

int sextans_window_pack_csr(int M, int K, const int *row_ptr, const int *col_idx, const float *val,
                            int rows_per_wave, int window_cols, sextans_window_packed *out) {
    if (!out || M < 0 || K < 0 || !row_ptr) return SEXTANS_ERR_INVALID;
    if (row_ptr[0] != 0) return SEXTANS_ERR_INVALID;
    for (int r = 0; r < M; ++r)
        if (row_ptr[r + 1] < row_ptr[r]) return SEXTANS_ERR_INVALID;
    const int64_t nnz = M > 0 ? row_ptr[M] : 0;
    for (int64_t j = 0; j < nnz; ++j)
        if (col_idx[j] < 0 || col_idx[j] >= K) return SEXTANS_ERR_INDEX;
    sx::WindowPlan p;
    if (!sx::build_window_plan(M, K, row_ptr, col_idx, val, rows_per_wave, window_cols, p)) return SEXTANS_ERR_INVALID;
    memset(out, 0, sizeof *out);
    out->M = M; out->K = K; out->nnz = nnz;
    out->rows_per_wave = p.rows_per_wave;
    out->window_cols = p.window_cols;
    out->nwaves = p.nwaves;
    out->steps = p.padded / sx::kWinStep;
    out->padded_lower_bound = sx::window_plan_padded_lower_bound(M, row_ptr, rows_per_wave);
    out->wave_step0 = dup(p.wave_step0);
    static_assert(sizeof(sx::WinEntry) == 8, "entry = {fp32 value, row << 23 | column}");
    out->stream = (uint64_t *)malloc(sizeof(uint64_t) * (p.stream.empty() ? 1 : p.stream.size()));
    if (!out->wave_step0 || !out->stream) { sextans_window_packed_free(out); return SEXTANS_ERR_ALLOC; }
    if (!p.stream.empty()) memcpy(out->stream, p.stream.data(), sizeof(uint64_t) * p.stream.size());
    return SEXTANS_OK;
}

void sextans_window_packed_free(sextans_window_packed *p) {
    if (!p) return;
    free(p->wave_step0); free(p->stream);
    memset(p, 0, sizeof *p);
}

int sextans_partition_rows_by_nnz(int M, const int *row_ptr, int world, int *ranges) {
    if (M < 0 || !row_ptr || world < 1 || !ranges) return SEXTANS_ERR_INVALID;
    const int64_t nnz = row_ptr[M];
    int prev = 0;
    for (int g = 0; g < world; ++g) {
        int cut = M;
        if (g + 1 < world) {
            const int64_t target = nnz * (g + 1) / world;
            int lo = prev, hi = M;                      // first row r with row_ptr[r] >= target
            while (lo < hi) {
                const int mid = lo + (hi - lo) / 2;
                if ((int64_t)row_ptr[mid] < target) lo = mid + 1; else hi = mid;
            }
            cut = lo;
        }
        ranges[2 * g] = prev;
        ranges[2 * g + 1] = cut;
        prev = cut;
    }
    return SEXTANS_OK;
}

}  // extern "C"
