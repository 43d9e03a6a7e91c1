// panel_plan.cpp -- see panel_plan.h.
#include "panel_plan.h"

#include <algorithm>
#include <atomic>
#include <thread>

#include "plan_device.h"

namespace sx {

namespace {

constexpr int kPad = 4;       // entries; every row of the packed stream starts on this boundary
constexpr int kTailPad = kPlanTailPad;   // extra zero entries at the end (the kernels fetch whole batches ahead)

struct Part {
    std::vector<int> blk_row;   // first rows of the blocks of this part (without the final end)
    std::vector<int> counts;    // dictionary size per block (0 = direct)
    std::vector<int> dict;
    int64_t nnz_panel = 0;
    int max_dict = 0;
};

// Rows [r0, r1) -> blocks, dictionaries and packed entries.
void build_part(const int *rp, const int *ci, const float *va, int RB, int r0, int r1,
                int max_unique, double min_reuse, const int *row_off, uint16_t *idx16, int *col32,
                float *pval, Part &out, std::vector<int> &stamp, std::vector<int> &local) {
    // stamp / local: K-sized scratch of the calling thread, reused from part to part (block ids are global row numbers,
    // so a stale stamp of an earlier part never equals a current one)
    std::vector<int> uniq;
    int blk_id = r0;
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
        // (a block cut short by the end of its part is judged by fit alone: a one-row remnant has no reuse of its own,
        // and a single direct block would push the whole matrix onto the slower mixed kernel instantiation)
        const bool remnant = e == r1 && e - r < RB;
        const bool use_dict = fits && !uniq.empty() && ((double)n >= min_reuse * (double)uniq.size() || remnant);
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
        blk_id = e;
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

    // Rows are cut into PARTS of kPlanPartBlocks * RB rows; every part starts a new block.  The parts are a property of
    // the format, not of the machine: the device builder (plan_device.hip) walks the same parts, one workgroup each, so
    // host and device produce the same block list (tests/test_plan_device_gpu.py compares the arrays byte for byte).
    const int PR = kPlanPartBlocks * RB;
    const int nparts = (M + PR - 1) / PR;
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)std::min<unsigned>(hw ? hw : 1, 16);
    if (out.nnz_total < (1 << 20)) nthreads = 1;   // two K-sized scratch arrays per thread
    nthreads = std::max(1, std::min(nthreads, nparts));
    std::vector<Part> parts((size_t)nparts);
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    auto fn = [&]() {
        std::vector<int> stamp((size_t)K, -1), local((size_t)K, 0);
        for (int t = next.fetch_add(1); t < nparts; t = next.fetch_add(1))
            build_part(row_ptr, col_idx, val, RB, t * PR, std::min(M, (t + 1) * PR), max_unique, min_reuse,
                       out.row_off.data(), out.idx16.data(), out.col32.data(), out.val.data(), parts[(size_t)t], stamp, local);
    };
    if (nthreads == 1) fn();
    else for (int t = 0; t < nthreads; ++t) pool.emplace_back(fn);
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
