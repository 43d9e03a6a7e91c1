// window_plan.cpp -- see window_plan.h.
#include "window_plan.h"

#include <algorithm>
#include <thread>

namespace sx {

namespace {

struct Item {
    uint64_t key;     // window << 40 | rank inside (row, window) << 16 | local row
    uint32_t col;
    float val;
};

// List scheduler for one wavefront: entries in canonical order (window, rank, row) are packed into steps
// of 32 with no row twice in a step.  An entry that would collide is parked in a short `deferred` list
// (kept in order, drained first by the following steps); once a row has a parked entry its later entries
// queue up behind it, so per-row order never changes.
struct WaveScheduler {
    std::vector<Item> items, deferred, keep;
    std::vector<int> stamp, parked;

    void run(const int *rp, const int *ci, const float *va, int row0, int row1, int RW, int window_cols,
             std::vector<WinEntry> &out, int &steps) {
        items.clear();
        for (int r = row0; r < row1; ++r) {
            int64_t win = -1;
            uint32_t rank = 0;
            for (int j = rp[r]; j < rp[r + 1]; ++j) {
                const int64_t w = ci[j] / window_cols;
                rank = (w == win) ? rank + 1 : 0;
                win = w;
                items.push_back({(uint64_t)w << 40 | (uint64_t)rank << 16 | (uint64_t)(r - row0), (uint32_t)ci[j],
                                 va[j]});
            }
        }
        std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.key < b.key; });
        stamp.assign((size_t)RW, -1);
        parked.assign((size_t)RW, 0);
        deferred.clear();
        const WinEntry pad = {0.0f, (uint32_t)RW << kWinColBits};
        size_t pos = 0;
        int step = 0;
        constexpr size_t kLookahead = 96;
        while (pos < items.size() || !deferred.empty()) {
            int count = 0;
            auto place = [&](const Item &it) {
                const uint32_t row = (uint32_t)(it.key & 0xffffu);
                out.push_back({it.val, row << kWinColBits | it.col});
                stamp[row] = step;
                ++count;
            };
            keep.clear();
            for (const Item &it : deferred) {
                const uint32_t row = (uint32_t)(it.key & 0xffffu);
                if (count < kWinStep && stamp[row] != step) { place(it); --parked[row]; }
                else keep.push_back(it);
            }
            deferred.swap(keep);
            while (count < kWinStep && pos < items.size() && deferred.size() < kLookahead) {
                const Item &it = items[pos++];
                const uint32_t row = (uint32_t)(it.key & 0xffffu);
                if (parked[row] > 0 || stamp[row] == step) { deferred.push_back(it); ++parked[row]; }
                else place(it);
            }
            for (; count < kWinStep; ++count) out.push_back(pad);
            ++step;
        }
        for (; step % kWinUnroll; ++step)
            for (int i = 0; i < kWinStep; ++i) out.push_back(pad);
        steps = step;
    }
};

}  // namespace

int64_t window_plan_padded_lower_bound(int M, const int *row_ptr, int rows_per_wave) {
    int64_t steps = 0;
    for (int r0 = 0; r0 < M; r0 += rows_per_wave) {
        const int r1 = std::min(M, r0 + rows_per_wave);
        int64_t longest = 0;
        for (int r = r0; r < r1; ++r) longest = std::max<int64_t>(longest, row_ptr[r + 1] - row_ptr[r]);
        const int64_t n = (int64_t)row_ptr[r1] - row_ptr[r0];
        const int64_t s = std::max((n + kWinStep - 1) / kWinStep, longest);
        steps += (s + kWinUnroll - 1) / kWinUnroll * kWinUnroll;
    }
    return steps * kWinStep;
}

bool build_window_plan(int M, int K, const int *row_ptr, const int *col_idx, const float *val,
                       int rows_per_wave, int window_cols, WindowPlan &out) {
    out = WindowPlan();
    if (rows_per_wave < 1 || rows_per_wave > kWinMaxRowsPerWave || window_cols < 1) return false;
    if ((int64_t)K > ((int64_t)1 << kWinColBits)) return false;
    const int RW = rows_per_wave;
    out.rows_per_wave = RW;
    out.window_cols = window_cols;
    out.nwaves = (M + RW - 1) / RW;
    out.nnz = M > 0 ? row_ptr[M] : 0;
    out.wave_step0.assign((size_t)out.nwaves + 1, 0);

    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)std::min<unsigned>(hw ? hw : 1, 64);
    if (out.nnz < (1 << 18)) nthreads = 1;
    nthreads = std::max(1, std::min(nthreads, out.nwaves));
    // contiguous wavefront ranges with about the same number of non-zeros per thread
    std::vector<int> cut((size_t)nthreads + 1, 0);
    cut[(size_t)nthreads] = out.nwaves;
    for (int t = 1; t < nthreads; ++t) {
        const int64_t target = out.nnz * t / nthreads;
        const int row = (int)(std::lower_bound(row_ptr, row_ptr + M + 1, (int)target) - row_ptr);
        cut[(size_t)t] = std::min(out.nwaves, std::max(cut[(size_t)t - 1], row / RW));
    }
    std::vector<std::vector<WinEntry>> part((size_t)nthreads);
    std::vector<int> wave_steps((size_t)out.nwaves, 0);
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) {
        auto fn = [&, t]() {
            WaveScheduler ws;
            auto &dst = part[(size_t)t];
            const int64_t n_est = (int64_t)row_ptr[std::min<int64_t>(M, (int64_t)cut[(size_t)t + 1] * RW)] -
                                  row_ptr[std::min<int64_t>(M, (int64_t)cut[(size_t)t] * RW)];
            dst.reserve((size_t)(n_est + n_est / 16 + 4096));
            for (int g = cut[(size_t)t]; g < cut[(size_t)t + 1]; ++g)
                ws.run(row_ptr, col_idx, val, g * RW, std::min(M, (g + 1) * RW), RW, window_cols, dst,
                       wave_steps[(size_t)g]);
        };
        if (nthreads == 1) fn(); else pool.emplace_back(fn);
    }
    for (auto &th : pool) th.join();
    int64_t steps = 0;
    for (int g = 0; g < out.nwaves; ++g) {
        out.wave_step0[(size_t)g] = (int)steps;
        steps += wave_steps[(size_t)g];
        if (steps > 0x7fffffffLL - kWinTailSteps) return false;
    }
    out.wave_step0[(size_t)out.nwaves] = (int)steps;
    out.padded = steps * kWinStep;
    out.stream.reserve((size_t)(steps + kWinTailSteps) * kWinStep);
    for (auto &p : part) {
        out.stream.insert(out.stream.end(), p.begin(), p.end());
        std::vector<WinEntry>().swap(p);
    }
    out.stream.resize((size_t)(steps + kWinTailSteps) * kWinStep, WinEntry{0.0f, (uint32_t)RW << kWinColBits});
    return true;
}

}  // namespace sx
