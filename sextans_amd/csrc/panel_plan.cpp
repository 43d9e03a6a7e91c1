// panel_plan.cpp -- see panel_plan.h.
#include "panel_plan.h"

#include <algorithm>
#include <thread>

namespace sx {

namespace {

constexpr int kPad = 4;       // entries; every row of the packed stream starts on this boundary
constexpr int kTailPad = 256; // extra zero entries at the end (the kernel keeps up to 4 batches of 32 ahead in flight)

struct Part {
    std::vector<int> blk_row;   // first rows of the blocks of this part (without the final end)
    std::vector<int> counts;    // dictionary size per block (0 = direct)
    std::vector<int> dict;
    int64_t nnz_panel = 0;
    int max_dict = 0;
};

// Rows [r0, r1) -> blocks, dictionaries and packed entries.
void build_part(int K, const int *rp, const int *ci, const float *va, int RB, int r0, int r1,
                int max_unique, double min_reuse, const int *row_off, uint16_t *idx16, int *col32,
                float *pval, Part &out) {
    std::vector<int> stamp((size_t)K, -1), local((size_t)K, 0), uniq;
    int blk_id = 0;
    for (int r = r0; r < r1;) {
        // grow the block row by row while its distinct columns fit the panel
        uniq.clear();
        int e = r;
        bool fits = true;
        while (e < r1 && e - r < RB) {
            const size_t mark = uniq.size();
            bool over = false;
            for (int j = rp[e]; j < rp[e + 1]; ++j) {
                const int c = ci[j];
                if (stamp[(size_t)c] != blk_id) {
                    if ((int)uniq.size() == max_unique) { over = true; break; }
                    stamp[(size_t)c] = blk_id;
                    uniq.push_back(c);
                }
            }
            if (over) {
                for (size_t u = mark; u < uniq.size(); ++u) stamp[(size_t)uniq[u]] = -1;   // undo this row
                uniq.resize(mark);
                if (e == r) fits = false;   // a single row already exceeds the panel
                break;
            }
            ++e;
        }
        if (e == r) e = r + 1;              // the oversized row forms a (direct) block of its own
        const int64_t n = (int64_t)rp[e] - rp[r];
        const bool use_dict = fits && !uniq.empty() && (double)n >= min_reuse * (double)uniq.size();
        out.blk_row.push_back(r);
        if (use_dict) {
            std::sort(uniq.begin(), uniq.end());
            for (size_t u = 0; u < uniq.size(); ++u) local[(size_t)uniq[u]] = (int)u;
            out.counts.push_back((int)uniq.size());
            out.dict.insert(out.dict.end(), uniq.begin(), uniq.end());
            out.nnz_panel += n;
            out.max_dict = std::max(out.max_dict, (int)uniq.size());
        } else {
            out.counts.push_back(0);
        }
        for (int row = r; row < e; ++row) {
            int o = row_off[row];
            for (int j = rp[row]; j < rp[row + 1]; ++j, ++o) {
                pval[o] = va[j];
                if (use_dict) idx16[o] = (uint16_t)local[(size_t)ci[j]];
                else col32[o] = ci[j];
            }
            // Padding of dictionary rows is CONSUMED by the kernel (whole groups of 4 entries): value
            // -0.0f against a panel row of +1.0f gives the product -0.0f, and x + (-0.0f) == x bit for
            // bit for every x, so the sequential sum is untouched.  kPadIndex is replaced by the
            // position of that row when the stream is handed to the device.
            if (use_dict)
                for (; o < row_off[row + 1]; ++o) { pval[o] = -0.0f; idx16[o] = kPadIndex; }
        }
        ++blk_id;
        r = e;
    }
}

}  // namespace

void build_panel_plan(int M, int K, const int *row_ptr, const int *col_idx, const float *val,
                      int rows_per_block, int max_unique, double min_reuse, PanelPlan &out) {
    const int RB = rows_per_block;
    if (max_unique > 65536) max_unique = 65536;   // 16-bit local index
    out = PanelPlan();
    out.rows_per_block = RB;
    out.max_unique = max_unique;
    out.nnz_total = M > 0 ? row_ptr[M] : 0;
    out.row_off.assign((size_t)M + 1, 0);
    for (int r = 0; r < M; ++r) {
        const int len = row_ptr[r + 1] - row_ptr[r];
        out.row_off[(size_t)r + 1] = out.row_off[(size_t)r] + (len + kPad - 1) / kPad * kPad;
    }
    const size_t total = (size_t)out.row_off[(size_t)M] + kTailPad;
    out.idx16.assign(total, 0);
    out.col32.assign(total, 0);
    out.val.assign(total, 0.0f);
    out.blk_row.clear();
    out.dict_ptr.assign(1, 0);
    if (M == 0) { out.blk_row.push_back(0); return; }

    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)std::min<unsigned>(hw ? hw : 1, 16);
    if (out.nnz_total < (1 << 20)) nthreads = 1;   // two K-sized scratch arrays per thread
    nthreads = std::max(1, std::min(nthreads, M / (4 * RB) + 1));
    // split rows at multiples of RB so every part has about the same number of non-zeros; parts
    // start new blocks, so the cut positions only influence block boundaries, never results
    std::vector<int> cut((size_t)nthreads + 1, 0);
    cut[(size_t)nthreads] = M;
    for (int t = 1; t < nthreads; ++t) {
        const int64_t target = out.nnz_total * t / nthreads;
        const int row = (int)(std::lower_bound(row_ptr, row_ptr + M + 1, (int)target) - row_ptr);
        cut[(size_t)t] = std::min(M, std::max(cut[(size_t)t - 1], row / RB * RB));
    }
    std::vector<Part> parts((size_t)nthreads);
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) {
        auto fn = [&, t]() {
            build_part(K, row_ptr, col_idx, val, RB, cut[(size_t)t], cut[(size_t)t + 1], max_unique,
                       min_reuse, out.row_off.data(), out.idx16.data(), out.col32.data(),
                       out.val.data(), parts[(size_t)t]);
        };
        if (nthreads == 1) fn(); else pool.emplace_back(fn);
    }
    for (auto &th : pool) th.join();
    for (auto &p : parts) {
        out.blk_row.insert(out.blk_row.end(), p.blk_row.begin(), p.blk_row.end());
        for (int c : p.counts) out.dict_ptr.push_back(out.dict_ptr.back() + c);
        out.dict.insert(out.dict.end(), p.dict.begin(), p.dict.end());
        out.nnz_in_panel_blocks += p.nnz_panel;
        out.max_dict = std::max(out.max_dict, p.max_dict);
    }
    out.blk_row.push_back(M);
}

}  // namespace sx
