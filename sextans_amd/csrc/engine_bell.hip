// engine_bell.hip -- the matrix-core side of the engine: blocked-ELL bf16 SpMM (BASELINE config 5, bell_kernels.h) and
// "MFMA only where a tile is actually dense" (ensure_dense: 32x32 tiles of a CSR matrix cut out into a blocked-ELL side matrix).
// No analogue in the reference (scalar fp32 PEs): parity unpinned by it, see DESIGN.md 4.5 / 4.7.
#include <algorithm>
#include <cstring>

#include <thread>

#include "bell_kernels.h"
#include "engine_state.h"
#include "rowblock_mfma_kernel.h"

namespace sxe {

void free_bell(sextans_engine *h) {
    (void)hipFree(h->d_bell_col_owned); (void)hipFree(h->d_bell_Af);
    h->d_bell_col_owned = nullptr; h->d_bell_col = nullptr; h->d_bell_Af = nullptr;
    h->bell_M = h->bell_K = h->bell_W = 0;
}

// "MFMA only where a tile is actually dense" (north_star).  Counts the 32x32 tiles of the main matrix whose fill
// reaches the threshold (always: get_stat "dense_tile_fraction" = share of the non-zeros sitting in such tiles) and,
// when the caller has opted into bf16 for them ("mfma_dense_tiles" = 1), cuts them out of the main matrix into a
// blocked-ELL bf16 side matrix for spmm_bell_mfma; the CSR kernels keep the remainder in fp32.  Only full 32-row
// block rows are searched; at most 256 dense tiles per block row (the densest columns first come first served).
// "mfma_dense_tiles" = 2 (round 6): dense ROW BLOCKS of 16 rows on the fp32 matrix cores (rowblock_mfma_kernel.h) -- bit-identical to the
// "exact" = 0 kernels.  A block is routed when every row of it is strictly ascending in its columns (the chain order of a row IS its
// CSR order; fragments hold one value per (row, column)) and its fill = entries / (64 x groups of 4 columns it touches) reaches the
// threshold.  The routed rows are emptied in the source matrix (the CSR kernels skip them: ensure_split marks them), their groups and
// fragments are stored in block order.  Host passes run on up to 32 threads; the fragments are filled on the device.
int build_rowblocks(sextans_engine *h) {
    const int nb = h->M / 16;
    if (nb == 0 || h->nnz == 0) return SEXTANS_OK;
    if (h->K > (1 << 24)) {   // (the kernel's buffer resources span one B panel: K x 32 floats must stay below 2^31 bytes)
        g_last_error = "mfma_dense_tiles = 2: K > 2^24; rows stay on the fp32 CSR kernels";
        return SEXTANS_OK;
    }
    PlanTimer timer(h);
    std::vector<int> rp, ci;
    std::vector<float> va;
    if (int rc = read_back_row_ptr(h, rp, 0)) return rc;
    if (int rc = read_back_entries(h, ci, va, 0)) return rc;
    const double thr = (double)h->opt_dense_fill_x100 / 100.0;
    std::vector<int> ngroups((size_t)nb, 0);   // groups of a routed block, 0 = not routed
    const unsigned nthreads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    auto parallel_blocks = [&](auto fn) {
        std::vector<std::thread> ts;
        for (unsigned t = 0; t < nthreads; ++t)
            ts.emplace_back([&, t]() {
                const int b0 = (int)((int64_t)nb * t / nthreads), b1 = (int)((int64_t)nb * (t + 1) / nthreads);
                fn(b0, b1);
            });
        for (auto &t : ts) t.join();
    };
    auto groups_of = [&](int b, std::vector<int> &g) -> bool {   // ascending distinct (column >> 2) of block b; false: a row is not strictly ascending
        g.clear();
        for (int r = 16 * b; r < 16 * b + 16; ++r)
            for (int j = rp[(size_t)r]; j < rp[(size_t)r + 1]; ++j) {
                if (j > rp[(size_t)r] && ci[(size_t)j] <= ci[(size_t)j - 1]) return false;
                g.push_back(ci[(size_t)j] >> 2);
            }
        std::sort(g.begin(), g.end());
        g.erase(std::unique(g.begin(), g.end()), g.end());
        return true;
    };
    parallel_blocks([&](int b0, int b1) {
        std::vector<int> g;
        for (int b = b0; b < b1; ++b) {
            const int64_t cnt = (int64_t)rp[(size_t)16 * b + 16] - rp[(size_t)16 * b];
            if (cnt == 0 || (double)cnt < thr * 64.0) continue;             // (even a single group would be below the threshold)
            if (!groups_of(b, g)) continue;
            if ((double)cnt >= thr * 64.0 * (double)g.size() && g.size() <= ((size_t)1 << 20)) ngroups[(size_t)b] = (int)g.size();   // (32-bit byte offsets inside a super block's fragments)
        }
    });
    std::vector<int> row0, gptr(1, 0);
    int64_t total = 0, routed_nnz = 0;
    for (int b = 0; b < nb; ++b)
        if (ngroups[(size_t)b] > 0) {
            row0.push_back(16 * b);
            total += ngroups[(size_t)b];
            if (total > 0x7fffffff / 64) {   // 32-bit group indices / fragment offsets in ints on the host side
                g_last_error = "mfma_dense_tiles = 2: more than 2^25 column groups; rows stay on the fp32 CSR kernels";
                return SEXTANS_OK;
            }
            gptr.push_back((int)total);
            routed_nnz += (int64_t)rp[(size_t)16 * b + 16] - rp[(size_t)16 * b];
        }
    h->dense_tiles = (int64_t)row0.size();
    h->dense_nnz = routed_nnz;
    if (row0.empty()) return SEXTANS_OK;
    std::vector<int> gcol((size_t)total);
    {
        const int nrb = (int)row0.size();
        std::vector<std::thread> ts;
        for (unsigned t = 0; t < nthreads; ++t)
            ts.emplace_back([&, t]() {
                std::vector<int> g;
                for (int i = (int)((int64_t)nrb * t / nthreads); i < (int)((int64_t)nrb * (t + 1) / nthreads); ++i) {
                    groups_of(row0[(size_t)i] / 16, g);
                    std::copy(g.begin(), g.end(), gcol.begin() + gptr[(size_t)i]);
                }
            });
        for (auto &t : ts) t.join();
    }
    if (int rc = upload(&h->d_rb_row0, row0)) return rc;
    if (int rc = upload(&h->d_rb_gptr, gptr)) return rc;
    if (int rc = upload(&h->d_rb_gcol, gcol)) return rc;
    {   // super blocks of 4 routed blocks: the ascending union of their groups + who owns each (the kernel loads a B fragment once per entry)
        const int nrb = (int)row0.size(), nsb = (nrb + 3) / 4;
        std::vector<int> ucount((size_t)nsb, 0);
        auto merge4 = [&](int sb, int *ucol, unsigned char *umask) -> int {   // null outputs: count only
            int pos[4], end[4], n = 0;
            for (int q = 0; q < 4; ++q) {
                const int rb = 4 * sb + q;
                pos[q] = rb < nrb ? gptr[(size_t)rb] : 0;
                end[q] = rb < nrb ? gptr[(size_t)rb + 1] : 0;
            }
            for (;;) {
                int c = 0x7fffffff;
                for (int q = 0; q < 4; ++q)
                    if (pos[q] < end[q]) c = std::min(c, gcol[(size_t)pos[q]]);
                if (c == 0x7fffffff) break;
                unsigned m = 0;
                for (int q = 0; q < 4; ++q)
                    if (pos[q] < end[q] && gcol[(size_t)pos[q]] == c) { m |= 1u << q; ++pos[q]; }
                if (ucol) { ucol[n] = c; umask[n] = (unsigned char)m; }
                ++n;
            }
            return n;
        };
        auto parallel_sb = [&](auto fn) {
            std::vector<std::thread> ts;
            for (unsigned t = 0; t < nthreads; ++t)
                ts.emplace_back([&, t]() {
                    for (int sb = (int)((int64_t)nsb * t / nthreads); sb < (int)((int64_t)nsb * (t + 1) / nthreads); ++sb) fn(sb);
                });
            for (auto &t : ts) t.join();
        };
        parallel_sb([&](int sb) { ucount[(size_t)sb] = merge4(sb, nullptr, nullptr); });
        std::vector<int> uptr((size_t)nsb + 1, 0);
        for (int sb = 0; sb < nsb; ++sb) uptr[(size_t)sb + 1] = uptr[(size_t)sb] + ucount[(size_t)sb];
        std::vector<int> ucol((size_t)uptr[(size_t)nsb]);
        std::vector<unsigned char> umask((size_t)uptr[(size_t)nsb]);
        parallel_sb([&](int sb) { merge4(sb, ucol.data() + uptr[(size_t)sb], umask.data() + uptr[(size_t)sb]); });
        std::vector<int> uent(2 * ucol.size());   // {column group, owner mask} pairs: one 8-byte scalar load per entry
        for (size_t k = 0; k < ucol.size(); ++k) { uent[2 * k] = ucol[k]; uent[2 * k + 1] = umask[k]; }
        if (int rc = upload(&h->d_sb_uptr, uptr)) return rc;
        if (int rc = upload(&h->d_sb_ucol, uent)) return rc;
        h->sb_n = nsb;
        h->sb_entries = uptr[(size_t)nsb];
    }
    SX_HIP(hipMalloc((void **)&h->d_rb_A, sizeof(float) * 64 * (size_t)total));
    SX_HIP(hipMemsetAsync(h->d_rb_A, 0, sizeof(float) * 64 * (size_t)total, hipStreamPerThread));
    hipLaunchKernelGGL(sx::rowblock_fill_fragments, dim3((unsigned)((row0.size() + 3) / 4)), dim3(256), 0, hipStreamPerThread, h->d_rp, h->d_ci, h->d_v, h->d_rb_row0,
                       h->d_rb_gptr, h->d_rb_gcol, h->d_rb_A, (int)row0.size());
    SX_HIP(hipStreamSynchronize(hipStreamPerThread));
    // the remainder: routed rows emptied, as the source matrix of the long-row split
    std::vector<int> mrp((size_t)h->M + 1, 0);
    size_t w = 0;
    for (int r = 0; r < h->M; ++r) {
        const int b = r >> 4;
        if (!(b < nb && ngroups[(size_t)b] > 0)) {
            const int j0 = rp[(size_t)r], j1 = rp[(size_t)r + 1];
            if (w != (size_t)j0) {
                std::copy(ci.begin() + j0, ci.begin() + j1, ci.begin() + (ptrdiff_t)w);
                std::copy(va.begin() + j0, va.begin() + j1, va.begin() + (ptrdiff_t)w);
            }
            w += (size_t)(j1 - j0);
        }
        mrp[(size_t)r + 1] = (int)w;
    }
    ci.resize(w ? w : 1); va.resize(w ? w : 1);
    if (int rc = upload(&h->d_srp, mrp)) return rc;
    if (int rc = upload(&h->d_sci, ci)) return rc;
    if (int rc = upload(&h->d_sv, va)) return rc;
    h->s_rp = h->d_srp; h->s_ci = h->d_sci; h->s_v = h->d_sv; h->s_nnz = (int64_t)w;
    h->m_rp = h->s_rp; h->m_ci = h->s_ci; h->m_v = h->s_v; h->m_nnz = h->s_nnz;
    h->rb_n = (int)row0.size();
    h->rb_groups = total;
    return SEXTANS_OK;
}

int mark_rowblock_skip(sextans_engine *h) {
    if (!h->d_skip) {
        SX_HIP(hipMalloc((void **)&h->d_skip, (size_t)std::max(h->M, 1)));
        SX_HIP(hipMemsetAsync(h->d_skip, 0, (size_t)std::max(h->M, 1), hipStreamPerThread));
    }
    hipLaunchKernelGGL(sx::rowblock_mark_skip, dim3((unsigned)(((int64_t)h->rb_n * 16 + 255) / 256)), dim3(256), 0, hipStreamPerThread, h->d_rb_row0, h->rb_n, h->d_skip);
    SX_HIP(hipStreamSynchronize(hipStreamPerThread));
    return SEXTANS_OK;
}

// The routed blocks of [row_begin, row_end) over the B panels the main path has just laid out (`plan` = the segments of d_Bp).
int launch_rowblocks(sextans_engine *h, const std::vector<Seg> &plan, const float *d_C_in, int64_t ldc_in, float *d_C_out, int64_t ldc, int N, int row_begin,
                     int row_end, float alpha, float beta, hipStream_t s) {
    for (const Seg &g : plan) {
        const int ncols_panel = g.ntiles * g.width;
        const int ncols = std::min(N - g.col0, g.last_cols ? (g.ntiles - 1) * g.width + g.last_cols : ncols_panel);
        const int tiles16 = (ncols + 15) / 16;
        if (tiles16 <= 0) continue;
        const float *bp = h->d_Bp + (size_t)h->K * (size_t)g.col0;
        const float *cin = d_C_in + (int64_t)g.col0 * ldc_in;
        float *cout = d_C_out + (int64_t)g.col0 * ldc;
        auto go = [&](auto kern, int NT) {
            const int tgs = (tiles16 + NT - 1) / NT;
            hipLaunchKernelGGL(kern, dim3((unsigned)((h->sb_n + 3) / 4) * (unsigned)tgs), dim3(256), 0, s, h->d_rb_row0, h->d_rb_gptr, h->d_sb_uptr,
                               (const int2 *)h->d_sb_ucol, h->d_rb_A, bp, (int64_t)h->K * g.width, g.width, h->K, cin, ldc_in, cout, ldc, h->rb_n, h->sb_n,
                               tgs, ncols_panel, ncols, row_begin, row_end, alpha, beta);
        };
        // (a wavefront owns 64 rows x 16 NT columns of C: 16 NT accumulator registers.  NT = 2 = 4 wavefronts per SIMD measured 2 - 4 % ahead of
        // NT = 4 = 2 per SIMD on the dense-block matrix at N = 64 .. 256 -- occupancy over B-fragment reuse; "rowblock_tiles" forces 1 / 4)
        if (tiles16 >= 4 && h->opt_rb_tiles == 4) go(sx::spmm_rowblock_mfma_f32<4>, 4);
        else if (h->opt_rb_tiles == 1) go(sx::spmm_rowblock_mfma_f32<1>, 1);
        else if (tiles16 >= 2) go(sx::spmm_rowblock_mfma_f32<2>, 2);
        else go(sx::spmm_rowblock_mfma_f32<1>, 1);
    }
    SX_HIP(hipGetLastError());
    return SEXTANS_OK;
}

int ensure_dense(sextans_engine *h) {
    if (h->dense_built_mfma == h->opt_mfma_dense && h->dense_built_fill == h->opt_dense_fill_x100) return SEXTANS_OK;
    if (h->dense_W > 0 || h->rb_n > 0 || h->opt_mfma_dense) {   // the source matrix may change: everything downstream starts again
        free_plan(h);
        free_window(h);
    }
    const bool had_tiles = h->dense_W > 0 || h->rb_n > 0;
    if (had_tiles || h->opt_mfma_dense) free_dense(h);
    else { h->dense_tiles = h->dense_nnz = 0; }
    h->dense_built_mfma = h->opt_mfma_dense;
    h->dense_built_fill = h->opt_dense_fill_x100;
    if (h->opt_mfma_dense == 2) return build_rowblocks(h);
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
        unsigned long long *d_cnt = nullptr, h_cnt[4] = {0, 0, 0, 0};
        SX_HIP(hipMalloc((void **)&d_cnt, 4 * sizeof(unsigned long long)));
        SX_HIP(hipMemset(d_cnt, 0, 4 * sizeof(unsigned long long)));
        const int groups = (mb + sx::kShRows - 1) / sx::kShRows;
        hipLaunchKernelGGL(sx::bell_union_count, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, nullptr, h->d_dense_col, mb, W,
                           d_cnt, d_cnt + 1, d_cnt + 2, d_cnt + 3);
        const hipError_t e = hipMemcpy(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost);
        (void)hipFree(d_cnt);
        SX_HIP(e);
        h->dense_share = h_cnt[0] ? (double)h_cnt[1] / (double)h_cnt[0] : 0.0;
        h->dense_max_union = h_cnt[3] ? 0x7fffffff : (int)h_cnt[2];
    }
    return SEXTANS_OK;
}

int launch_dense_tiles(sextans_engine *h, int N, float alpha, const float *d_B, int64_t ldb, float beta, const float *d_C_in,
                       int64_t ldc_in, float *d_C_out, int64_t ldc, hipStream_t s) {
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
    return SEXTANS_OK;
}

}  // namespace sxe

using namespace sxe;

extern "C" {

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
        unsigned long long *d_cnt = nullptr, h_cnt[4] = {0, 0, 0, 0};
        SX_HIP(hipMalloc((void **)&d_cnt, 4 * sizeof(unsigned long long)));
        SX_HIP(hipMemset(d_cnt, 0, 4 * sizeof(unsigned long long)));
        const int groups = (M / 32 + sx::kShRows - 1) / sx::kShRows;
        hipLaunchKernelGGL(sx::bell_union_count, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, nullptr, d_block_col, M / 32,
                           ell_width, d_cnt, d_cnt + 1, d_cnt + 2, d_cnt + 3);
        const hipError_t e = hipMemcpy(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost);
        (void)hipFree(d_cnt);
        SX_HIP(e);
        h->bell_share = h_cnt[0] ? (double)h_cnt[1] / (double)h_cnt[0] : 0.0;
        h->bell_max_union = h_cnt[3] ? 0x7fffffff : (int)h_cnt[2];   // duplicate / unsorted block columns: never the shared kernel
        if (h_cnt[3]) h->bell_share = 0.0;
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
    return sextans_spmm_bell_device2(h, N, alpha, d_B, ldb, beta, d_C_in, ldc, d_C_out, ldc, stream);
}

// (separate leading dimensions of C_in and C_out: a rank of sextans_dist_spmm_bell reads its rows inside the whole C_in and writes a
// packed slab)
int sextans_spmm_bell_device2(sextans_handle_t h, int N, float alpha, const uint16_t *d_B, int64_t ldb, float beta, const float *d_C_in,
                              int64_t ldc_in, float *d_C_out, int64_t ldc, void *stream) {
    if (!h || N <= 0 || (N % 32) || !d_B || !d_C_in || !d_C_out) return SEXTANS_ERR_INVALID;
    if (!h->d_bell_Af) return SEXTANS_ERR_STATE;
    if (ldb < h->bell_K || (ldb % 8) || ldc < h->bell_M || ldc_in < h->bell_M) return SEXTANS_ERR_INVALID;
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
                           h->d_bell_col, Af, Bf, d_C_in, ldc_in, d_C_out, ldc, mblocks, h->bell_W, ntiles,    \
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
                               s, h->d_bell_col, Af, Bf, d_C_in, ldc_in, d_C_out, ldc, mblocks, h->bell_W, alpha, beta, (int)h->opt_bell_debug);
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
                                   Bf, d_C_in, ldc_in, d_C_out, ldc, std::min(mblocks, b0 + nb), h->bell_W, alpha, beta, b0);
            }
        } else if (ntiles % 4 == 0) SX_BELL(4) else if (ntiles % 2 == 0) SX_BELL(2) else SX_BELL(1)
#undef SX_BELL
        h->last_kernel = "spmm_bell_mfma";
    }
    SX_HIP(hipGetLastError());
    return SEXTANS_OK;
}

}  // extern "C"
