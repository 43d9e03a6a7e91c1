// edge_stream.cpp -- the Sextans accelerator's buffer formats, host side.
//
// Writer, reader and container file for the exact buffers the reference host hands to
// tapa::invoke(Sextans, ...) (sextans-host.cpp:237-251), so inputs prepared for the FPGA can be
// consumed by this engine and its outputs produced in the FPGA's layout:
//
//   * scheduled non-zero stream  -- behaviour of generate_edge_list_for_one_PE / _all_PEs
//     (sparse_helper.h:292-403): per 4096-column window and per PE (row % 64), entries are taken in
//     CSC order and each goes to the first free slot that is >= 10 slots after the previous entry
//     of the same row in that window; the 64 PE lists are padded with bubbles to the window's
//     longest list; edge_list_ptr[w+1] = cumulative length.
//   * 64-bit words               -- edge_list_64bit (sparse_helper.h:406-473) for 8 channels:
//     word = col14 << 50 | row18 << 32 | fp32 bits, bubble = 0x3FFFF << 32 (the kernel treats any
//     word with row bit 17 set as a bubble, sextans.cpp:407); PE p lives in channel p % 8 at
//     slot bitrev3(p / 8) of each 8-word group.
//   * dense B / C channel layouts -- sextans-host.cpp:152-195 and :264-270.
//
// Own implementation: the slot search uses a next-free-slot union-find instead of the reference's
// linear probe, per-row state is reset through a touched list instead of an M-sized vector per
// call, and words are produced directly (no intermediate edge structs).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "sextans_amd.h"

namespace {

constexpr int kPE = SEXTANS_EDGES_NUM_PE;          // NUM_CH_SPARSE * 8, sextans-host.cpp:122
constexpr int kWindow = SEXTANS_EDGES_WINDOW;      // WINDOW_SIZE, sextans.h:11
constexpr int kRawDist = 10;                       // DEP_DIST_LOAD_STORE, sextans.h:12
constexpr uint64_t kBubble = 0x3FFFFull << 32;     // sparse_helper.h:427-429
constexpr int kRowLimit = 1 << 17;                 // row bit 17 marks a bubble, sextans.cpp:407

inline int bitrev3(int x) { return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1); }
inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

inline uint64_t make_word(int col_in_window, int row_in_pe, float v) {
    uint32_t bits;
    memcpy(&bits, &v, 4);
    return ((uint64_t)(col_in_window & 0x3FFF) << 50) | ((uint64_t)(row_in_pe & 0x3FFFF) << 32) | bits;
}

// One PE's schedule inside one window.  `slots` holds the words (bubbles where empty).
struct Scheduler {
    std::vector<uint64_t> slots;
    std::vector<int> next_free;     // union-find: smallest free slot >= i
    int find(int i) {
        int r = i;
        while (next_free[r] != r) r = next_free[r];
        while (next_free[i] != r) { int n = next_free[i]; next_free[i] = r; i = n; }
        return r;
    }
    void grow(int n) {                 // make slots [0, n) addressable
        int old = (int)slots.size();
        if (n <= old) return;
        slots.resize(n, kBubble);
        next_free.resize(n + 1);
        for (int i = old; i <= n; ++i) next_free[i] = i;   // sentinel at n is always free
    }
    void clear() { slots.clear(); next_free.assign(1, 0); }
    int place(int earliest, uint64_t w) {
        if (earliest >= (int)slots.size()) grow(earliest + 1);
        int c = find(earliest);
        if (c >= (int)slots.size()) grow(c + 1);
        slots[c] = w;
        next_free[c] = c + 1;
        return c;
    }
};

struct Entry { int row; int col; float val; };

template <class T> T *alloc_zero(int64_t n) { return (T *)calloc((size_t)(n > 0 ? n : 1), sizeof(T)); }

}  // namespace

extern "C" {

int sextans_edges_pack_csc(int M, int K, int nnz, const int *col_ptr, const int *row_idx, const float *val,
                           sextans_edges *out) {
    if (!out || M < 0 || K < 0 || nnz < 0 || !col_ptr) return SEXTANS_ERR_INVALID;
    if (nnz > 0 && (!row_idx || !val)) return SEXTANS_ERR_INVALID;
    if (col_ptr[0] != 0 || col_ptr[K] != nnz) return SEXTANS_ERR_INVALID;
    if (M > 0 && (M - 1) / kPE >= kRowLimit) return SEXTANS_ERR_INVALID;   // row field would read as a bubble
    for (int c = 0; c < K; ++c)
        if (col_ptr[c + 1] < col_ptr[c]) return SEXTANS_ERR_INVALID;
    for (int j = 0; j < nnz; ++j)
        if (row_idx[j] < 0 || row_idx[j] >= M) return SEXTANS_ERR_INDEX;

    const int num_windows = (K + kWindow - 1) / kWindow;
    std::vector<std::vector<uint64_t>> pe_words(kPE);       // concatenated, window-aligned
    std::vector<int> ptr((size_t)num_windows + 1, 0);
    std::vector<std::vector<Entry>> bucket(kPE);
    std::vector<Scheduler> sched(kPE);
    std::vector<int> last_slot((size_t)M, -kRawDist);        // per global row; row r belongs to PE r % 64
    std::vector<std::vector<int>> touched(kPE);

    const unsigned hw = std::thread::hardware_concurrency();
    for (int w = 0; w < num_windows; ++w) {
        const int c0 = w * kWindow, c1 = (c0 + kWindow < K) ? c0 + kWindow : K;
        for (auto &b : bucket) b.clear();
        for (int c = c0; c < c1; ++c)
            for (int j = col_ptr[c]; j < col_ptr[c + 1]; ++j)
                bucket[row_idx[j] % kPE].push_back({row_idx[j], c, val[j]});
        auto run = [&](int p0, int p1) {
            for (int p = p0; p < p1; ++p) {
                Scheduler &s = sched[p];
                s.clear();
                for (const Entry &e : bucket[p]) {
                    int &last = last_slot[e.row];
                    if (last == -kRawDist) touched[p].push_back(e.row);
                    last = s.place(last + kRawDist, make_word(e.col - c0, e.row / kPE, e.val));
                }
                for (int r : touched[p]) last_slot[r] = -kRawDist;
                touched[p].clear();
            }
        };
        const int64_t wn = col_ptr[c1] - col_ptr[c0];
        const int nt = (wn > (1 << 16) && hw > 1) ? (int)(hw < 16 ? hw : 16) : 1;
        if (nt == 1) {
            run(0, kPE);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back(run, kPE * t / nt, kPE * (t + 1) / nt);
            for (auto &t : th) t.join();
        }
        size_t longest = 0;
        for (int p = 0; p < kPE; ++p) {
            pe_words[p].insert(pe_words[p].end(), sched[p].slots.begin(), sched[p].slots.end());
            if (pe_words[p].size() > longest) longest = pe_words[p].size();
        }
        if (longest > (size_t)INT32_MAX) return SEXTANS_ERR_INVALID;
        for (int p = 0; p < kPE; ++p) pe_words[p].resize(longest, kBubble);
        ptr[w + 1] = (int)longest;
    }

    const int64_t total = ptr[num_windows];
    memset(out, 0, sizeof *out);
    out->M = M; out->K = K; out->nnz = nnz;
    out->num_windows = num_windows;
    out->num_a_len = (int)total;
    out->ptr_len = round_up(round_up(num_windows + 1, 16), 1024);      // sextans-host.cpp:131-134
    out->chan_len = round_up(8 * total, 512);                           // sparse_helper.h:412-413
    out->edge_list_ptr = alloc_zero<int32_t>(out->ptr_len);
    bool ok = out->edge_list_ptr != nullptr;
    for (int c = 0; c < SEXTANS_EDGES_NUM_CH; ++c) {
        out->channel[c] = alloc_zero<uint64_t>(out->chan_len);
        ok = ok && out->channel[c];
    }
    if (!ok) { sextans_edges_free(out); return SEXTANS_ERR_ALLOC; }
    memcpy(out->edge_list_ptr, ptr.data(), sizeof(int32_t) * ptr.size());
    for (int p = 0; p < kPE; ++p) {
        uint64_t *ch = out->channel[p % 8] + bitrev3(p / 8);            // sparse_helper.h:458-464
        const uint64_t *src = pe_words[p].data();
        for (int64_t i = 0; i < total; ++i) ch[i * 8] = src[i];
    }
    return SEXTANS_OK;
}

void sextans_edges_free(sextans_edges *e) {
    if (!e) return;
    free(e->edge_list_ptr);
    for (int c = 0; c < SEXTANS_EDGES_NUM_CH; ++c) free(e->channel[c]);
    memset(e, 0, sizeof *e);
}

int sextans_edges_decode_csr(const int32_t *edge_list_ptr, const uint64_t *const *channel, int num_windows,
                             int M, int K, int64_t *nnz_out, int **row_ptr_out, int **col_idx_out,
                             float **val_out) {
    if (!edge_list_ptr || !channel || num_windows < 0 || M < 0 || K < 0 || !nnz_out || !row_ptr_out ||
        !col_idx_out || !val_out)
        return SEXTANS_ERR_INVALID;
    if (num_windows != (K + kWindow - 1) / kWindow) return SEXTANS_ERR_INVALID;
    if (edge_list_ptr[0] != 0) return SEXTANS_ERR_INVALID;
    for (int w = 0; w < num_windows; ++w)
        if (edge_list_ptr[w + 1] < edge_list_ptr[w]) return SEXTANS_ERR_INVALID;
    const int64_t total = edge_list_ptr[num_windows];
    for (int c = 0; c < SEXTANS_EDGES_NUM_CH; ++c)
        if (total > 0 && !channel[c]) return SEXTANS_ERR_INVALID;

    // A row's words all live in one PE stream, so PEs decode independently: count, prefix, scatter
    // in stream order (= the order the accelerator accumulates a row's products in).
    std::vector<int64_t> count((size_t)M + 1, 0);
    int bad[kPE] = {0};
    auto for_pe = [&](int p, auto &&fn) {
        const uint64_t *ch = channel[p % 8] + bitrev3(p / 8);
        int w = 0;
        for (int64_t i = 0; i < total; ++i) {
            while (i >= edge_list_ptr[w + 1]) ++w;
            const uint64_t x = ch[i * 8];
            const uint32_t row18 = (uint32_t)(x >> 32) & 0x3FFFF;
            if (row18 & (1u << 17)) continue;                            // bubble, sextans.cpp:407
            const int64_t row = (int64_t)row18 * kPE + p;
            const int64_t col = (int64_t)w * kWindow + (int64_t)(x >> 50);
            if (row >= M || col >= K) { bad[p] = 1; return; }
            fn((int)row, (int)col, (uint32_t)x);
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (total > (1 << 14) && hw > 1) ? (int)(hw < 16 ? hw : 16) : 1;
    auto parallel_pes = [&](auto &&body) {
        if (nt == 1) { for (int p = 0; p < kPE; ++p) body(p); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] { for (int p = kPE * t / nt; p < kPE * (t + 1) / nt; ++p) body(p); });
        for (auto &t : th) t.join();
    };
    parallel_pes([&](int p) { for_pe(p, [&](int r, int, uint32_t) { ++count[r + 1]; }); });
    for (int p = 0; p < kPE; ++p)
        if (bad[p]) return SEXTANS_ERR_INDEX;
    for (int r = 0; r < M; ++r) count[r + 1] += count[r];
    const int64_t nnz = count[M];
    if (nnz > INT32_MAX) return SEXTANS_ERR_INVALID;
    int *rp = (int *)malloc(sizeof(int) * ((size_t)M + 1));
    int *ci = (int *)malloc(sizeof(int) * (size_t)(nnz ? nnz : 1));
    float *v = (float *)malloc(sizeof(float) * (size_t)(nnz ? nnz : 1));
    if (!rp || !ci || !v) { free(rp); free(ci); free(v); return SEXTANS_ERR_ALLOC; }
    for (int r = 0; r <= M; ++r) rp[r] = (int)count[r];
    std::vector<int> fill(rp, rp + M);                                   // next free position per row
    parallel_pes([&](int p) {
        for_pe(p, [&](int r, int c, uint32_t bits) {
            const int o = fill[r]++;
            ci[o] = c;
            memcpy(&v[o], &bits, 4);
        });
    });
    *nnz_out = nnz; *row_ptr_out = rp; *col_idx_out = ci; *val_out = v;
    return SEXTANS_OK;
}

// ---- container file: the buffers as they are, behind a fixed little-endian header -----------------
//   char magic[8] = "SXTEDGE1"; int32 M, K, num_windows, num_a_len; int64 nnz, ptr_len, chan_len;
//   int32 edge_list_ptr[ptr_len]; uint64 channel[8][chan_len]
namespace {
const char kMagic[8] = {'S', 'X', 'T', 'E', 'D', 'G', 'E', '1'};
struct FileHeader { char magic[8]; int32_t M, K, num_windows, num_a_len; int64_t nnz, ptr_len, chan_len; };
}  // namespace

int sextans_edges_save(const char *path, const sextans_edges *e) {
    if (!path || !e || !e->edge_list_ptr) return SEXTANS_ERR_INVALID;
    FILE *f = fopen(path, "wb");
    if (!f) return SEXTANS_ERR_OPEN;
    FileHeader h;
    memcpy(h.magic, kMagic, 8);
    h.M = e->M; h.K = e->K; h.num_windows = e->num_windows; h.num_a_len = e->num_a_len;
    h.nnz = e->nnz; h.ptr_len = e->ptr_len; h.chan_len = e->chan_len;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1;
    ok = ok && fwrite(e->edge_list_ptr, sizeof(int32_t), (size_t)e->ptr_len, f) == (size_t)e->ptr_len;
    for (int c = 0; c < SEXTANS_EDGES_NUM_CH && ok; ++c)
        ok = fwrite(e->channel[c], sizeof(uint64_t), (size_t)e->chan_len, f) == (size_t)e->chan_len;
    ok = (fclose(f) == 0) && ok;
    return ok ? SEXTANS_OK : SEXTANS_ERR_OPEN;
}

int sextans_edges_load(const char *path, sextans_edges *out) {
    if (!path || !out) return SEXTANS_ERR_INVALID;
    FILE *f = fopen(path, "rb");
    if (!f) return SEXTANS_ERR_OPEN;
    FileHeader h;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, kMagic, 8) != 0) { fclose(f); return SEXTANS_ERR_PARSE; }
    const int nw = h.K >= 0 ? (h.K + kWindow - 1) / kWindow : -1;
    if (h.M < 0 || h.K < 0 || h.num_windows != nw || h.num_a_len < 0 || h.ptr_len < (int64_t)nw + 1 ||
        h.chan_len < 8 * (int64_t)h.num_a_len || h.ptr_len > (1ll << 31) || h.chan_len > (1ll << 34)) {
        fclose(f);
        return SEXTANS_ERR_PARSE;
    }
    memset(out, 0, sizeof *out);
    out->M = h.M; out->K = h.K; out->num_windows = h.num_windows; out->num_a_len = h.num_a_len;
    out->nnz = h.nnz; out->ptr_len = h.ptr_len; out->chan_len = h.chan_len;
    out->edge_list_ptr = alloc_zero<int32_t>(h.ptr_len);
    bool ok = out->edge_list_ptr &&
              fread(out->edge_list_ptr, sizeof(int32_t), (size_t)h.ptr_len, f) == (size_t)h.ptr_len;
    for (int c = 0; c < SEXTANS_EDGES_NUM_CH && ok; ++c) {
        out->channel[c] = alloc_zero<uint64_t>(h.chan_len);
        ok = out->channel[c] &&
             fread(out->channel[c], sizeof(uint64_t), (size_t)h.chan_len, f) == (size_t)h.chan_len;
    }
    fclose(f);
    if (!ok || out->edge_list_ptr[h.num_windows] != h.num_a_len) { sextans_edges_free(out); return SEXTANS_ERR_PARSE; }
    return SEXTANS_OK;
}

// ---- dense channel layouts ---------------------------------------------------------------------------
// B (sextans-host.cpp:152-177): N-tiles of 8 columns; 8 channels: column n in channel n % 8 at
// k + colsize * (n / 8), colsize = round_up(K, 16); 4 channels: columns 2c, 2c+1 of a tile share
// channel c, interleaved in runs of 8 rows, colsize = round_up(K, 8) * 2.
// C (:179-195, :264-270): row m in channel m % 8 at colsize * (n / 8) + (m / 8) * 8 + n % 8,
// colsize = round_up(M, 16).  Every channel is round_up(colsize * N / 8, 1024) floats.
int64_t sextans_chan_b_colsize(int K, int num_ch_b) {
    return num_ch_b == 8 ? round_up(K, 16) : round_up(K, 8) * 2;
}
int64_t sextans_chan_b_len(int K, int N, int num_ch_b) {
    return round_up(sextans_chan_b_colsize(K, num_ch_b) * (N / 8), 1024);
}
int64_t sextans_chan_c_colsize(int M) { return round_up(M, 16); }
int64_t sextans_chan_c_len(int M, int N) { return round_up(sextans_chan_c_colsize(M) * (N / 8), 1024); }

static inline void b_slot(int k, int n, int64_t colsize, int num_ch_b, int &ch, int64_t &pos) {
    if (num_ch_b == 8) {
        ch = n % 8;
        pos = k + colsize * (n / 8);
    } else {
        ch = (n / 2) % 4;
        pos = (int64_t)(k / 8) * 16 + (n % 2) * 8 + k % 8 + colsize * (n / 8);
    }
}

int sextans_chan_pack_b(int K, int N, int num_ch_b, const float *B, float *const *ch) {
    if (K < 0 || N <= 0 || N % 8 || (num_ch_b != 4 && num_ch_b != 8) || !B || !ch) return SEXTANS_ERR_INVALID;
    const int64_t cs = sextans_chan_b_colsize(K, num_ch_b);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            int c; int64_t pos;
            b_slot(k, n, cs, num_ch_b, c, pos);
            ch[c][pos] = B[k + (int64_t)K * n];
        }
    return SEXTANS_OK;
}

int sextans_chan_unpack_b(int K, int N, int num_ch_b, const float *const *ch, float *B) {
    if (K < 0 || N <= 0 || N % 8 || (num_ch_b != 4 && num_ch_b != 8) || !B || !ch) return SEXTANS_ERR_INVALID;
    const int64_t cs = sextans_chan_b_colsize(K, num_ch_b);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            int c; int64_t pos;
            b_slot(k, n, cs, num_ch_b, c, pos);
            B[k + (int64_t)K * n] = ch[c][pos];
        }
    return SEXTANS_OK;
}

int sextans_chan_pack_c(int M, int N, const float *C, float *const *ch) {
    if (M < 0 || N <= 0 || N % 8 || !C || !ch) return SEXTANS_ERR_INVALID;
    const int64_t cs = sextans_chan_c_colsize(M);
    for (int n = 0; n < N; ++n)
        for (int m = 0; m < M; ++m)
            ch[m % 8][cs * (n / 8) + (int64_t)(m / 8) * 8 + n % 8] = C[m + (int64_t)M * n];
    return SEXTANS_OK;
}

int sextans_chan_unpack_c(int M, int N, const float *const *ch, float *C) {
    if (M < 0 || N <= 0 || N % 8 || !C || !ch) return SEXTANS_ERR_INVALID;
    const int64_t cs = sextans_chan_c_colsize(M);
    for (int n = 0; n < N; ++n)
        for (int m = 0; m < M; ++m)
            C[m + (int64_t)M * n] = ch[m % 8][cs * (n / 8) + (int64_t)(m / 8) * 8 + n % 8];
    return SEXTANS_OK;
}

}  // extern "C"
