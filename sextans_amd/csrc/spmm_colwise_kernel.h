// spmm_colwise_kernel.h -- short-row form of the SpMM for matrices whose NUMBERING has locality: one lane per row, the B and C
// accesses of a wavefront coalesce along the rows of the column-major operands.
//
// The reference packs short rows back to back into each PE's list (sparse_helper.h:292-343) so that a PE never idles on them;
// what costs time here is different: for a 2-D 5-point stencil at N = 16 a row has 40 bytes of non-zeros against 64 bytes of B and
// 128 bytes of C, so the row-group / LDS-panel kernels spend the step on what surrounds the non-zeros -- the B repack into row-major
// panels (512 MB moved, 81 us for K = 4 M: 22 % of the step), the C tile transposes, a per-row prologue and epilogue for five
// multiply-adds per column.  This kernel has none of that:
//   * thread = row r, 16 accumulators = the 16 columns of one N tile; a wavefront = 64 CONSECUTIVE rows;
//   * B is read where the caller left it (column-major, leading dimension ldb): the j-th entries of 64 consecutive rows have
//     columns c_j(r) that move with r in a stencil / banded / mesh-in-generator-order matrix, so the 64 four-byte loads of
//     B[c_j(r) + n ldb] fall into one or two 128-byte lines per column n -- no repack launch, no LDS;
//   * C_in / C_out column-major: 64 consecutive rows of one column are one 256-byte run.
// Every row is summed in ascending CSR order by one lane per output element, product rounded before the add when EXACT: the order
// and rounding of cpu_spmm_CSR (sparse_helper.h:279-289), bit-identical.
// It pays only for SHORT rows (the per-entry cost is 1 + 16 vector-memory instructions per lane; the LDS-panel kernel needs
// 1 LDS read per 4 columns) in numberings WITH locality (else every one of those loads touches 64 lines): the dispatcher uses it
// when the mean row length is <= 6 and sampled consecutive rows have neighbouring columns (engine_plan.hip: ensure_colwise).
// Measured (4M rows, same-box A/B, us per step, first version): 5-point stencil N = 16 361 -> 259, N = 32 592 -> 484, N = 128 2011 -> 1859; 9-point
// (9 per row) 368 -> 407: already a loss; 27-point 450 -> 2489; banded random columns 530 -> 1836 (no coherence between rows).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spmm_csr_kernels.h"


namespace sx {

// rows [row_begin, row_end); C pointers address row_begin as their row 0; grid = (row blocks of 256) x tiles of NC columns from col_base, row blocks
// spread over the XCDs in contiguous chunks (neighbouring row blocks share B lines in L2).
// RM (round 5): ROW-major operands (sextans_spmm_device_rm) -- B[c * ldb + n], C[r * ldc + n]: the NC values of a B row and of a C row are
// contiguous, so every access is a 16-byte load / store (4 per entry instead of 16 four-byte ones) and a wavefront's C rows are one
// contiguous 4 KB run.  Measured on the 4M-row 5-point stencil, N = 16: 237 us against 214 us for the column-major form (each of the four
// load instructions per entry touches all 32 lines of the wavefront's 4 KB) and against ~465 us through column-major copies.  A form with
// FOUR lanes per row and 4 columns each (one load instruction = 1 KB of consecutive B) was built and measured too: 285 us -- every lane
// of a row group fetches the row's columns and values again and four times the wavefronts wait through the same latency chain.
template <bool EXACT, int NC, bool RM = false>   // NC = columns of a tile: 16, or 8 for a remainder tile
__global__ __launch_bounds__(kBlock) void spmm_csr_colwise(const int *__restrict__ rp, const int *__restrict__ ci, const float *__restrict__ va,
                                                           const float *__restrict__ B, int64_t ldb, const float *Cin, int64_t ldc_in, float *Cout,
                                                           int64_t ldc, int row_begin, int row_end, int nrowblk, int col_base, float alpha, float beta,
                                                           int use_xcd_remap, const unsigned char *__restrict__ skip, int tiles_adjacent, int tgroup) {
    // tiles_adjacent > 0 (= the number of tiles; a 1-D grid of nrowblk * tiles workgroups): the tiles of one row block are NEIGHBOURS in
    // the launch order and, through the XCD remap, run on one XCD at about the same time -- the row block's rp / ci / va are read from
    // HBM once instead of once per tile, and in the row-major form the two 64-byte halves of a 128-byte line of B / C are asked for
    // together (with the tile as the slow grid axis they are fetched twice, tiles apart: 5-point stencil 4M rows, row-major N = 32,
    // 0.27 of the roofline against 0.50 at N = 16).
    // tgroup > 1 (row-major form only): `tgroup` neighbouring lanes share a row and take neighbouring tiles of it, so one load instruction
    // of the wavefront covers tgroup * 64 consecutive bytes of every B / C row it touches -- whole 128-byte lines from N = 32 on (one
    // lane per row and tile asks for 64 of the 128 bytes, the other half travels again for the next tile: 0.25 of the roofline at
    // N = 32 .. 128 against 0.48 at N = 16).  The lanes of a group fetch the row's columns and values together (one address: a broadcast).
    unsigned wg = blockIdx.x;
    int tile = blockIdx.y;
    int r;
    if (RM && tgroup > 1) {
        if (use_xcd_remap) wg = xcd_remap(wg, (unsigned)nrowblk);
        const int rows_per = kBlock / tgroup, lr_ = (int)threadIdx.x / tgroup;
        if (lr_ >= rows_per) return;
        r = row_begin + (int)wg * rows_per + lr_;
        tile = (int)blockIdx.y * tgroup + (int)threadIdx.x % tgroup;
    } else {
        if (tiles_adjacent > 0) {
            if (use_xcd_remap) wg = xcd_remap(wg, (unsigned)nrowblk * (unsigned)tiles_adjacent);
            tile = (int)(wg % (unsigned)tiles_adjacent);
            wg /= (unsigned)tiles_adjacent;
        } else if (use_xcd_remap) {
            wg = xcd_remap(wg, (unsigned)nrowblk);
        }
        r = row_begin + (int)wg * kBlock + (int)threadIdx.x;
    }
    if (r >= row_end) return;
    const int col0 = col_base + tile * NC;
    const float *b = RM ? B + col0 : B + (int64_t)col0 * ldb;
    auto load_b = [&](float (&dst)[NC], int c) {
        if constexpr (RM) {
#pragma unroll
            for (int n = 0; n < NC; n += 4) {
                const f32x4 x = *reinterpret_cast<const f32x4 *>(b + (int64_t)c * ldb + n);
                dst[n] = x.x; dst[n + 1] = x.y; dst[n + 2] = x.z; dst[n + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int n = 0; n < NC; ++n) dst[n] = b[c + (int64_t)n * ldb];
        }
    };
    int j = rp[r];
    const int j1 = rp[r + 1];
    float acc[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[n] = 0.f;
    // C_in early: its 16 loads fly under the row loop
    const int64_t lr = r - row_begin;
    float cin[NC];
    if constexpr (RM) {
#pragma unroll
        for (int n = 0; n < NC; n += 4) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(Cin + lr * ldc_in + col0 + n);
            cin[n] = x.x; cin[n + 1] = x.y; cin[n + 2] = x.z; cin[n + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int n = 0; n < NC; ++n) cin[n] = Cin[lr + (int64_t)(col0 + n) * ldc_in];
    }
    // The first PRE entries of the row: columns and values first (one round trip), then the B values of entry e + 1 are requested
    // before the products of entry e are formed (two register sets): a 5-entry row costs ~3 dependent round trips instead of 6.
    // Same-box, 5-point stencil 4M rows: N = 16 250 -> 230 us, N = 32 475 -> 430, N = 128 1 840 -> 1 610 (9 entries per row, forced: 397 -> 351,
    // still behind the panel kernel's 320 per step).  The order of the multiply-adds is unchanged.
    constexpr int PRE = 8;
    {
        int cc[PRE];
        float aa[PRE];
        const int len = min(j1 - j, PRE);
#pragma unroll
        for (int e = 0; e < PRE; ++e) {
            cc[e] = 0; aa[e] = 0.f;
            if (e < len) { cc[e] = ci[j + e]; aa[e] = va[j + e]; }
        }
        constexpr int DEPTH = 1;   // entries whose B values are in flight ahead of the one being multiplied (2 / 3: 118 / 132 registers, measured equal / slower)
        float bv[DEPTH + 1][NC];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d < len) load_b(bv[d], cc[d]);
#pragma unroll
        for (int e = 0; e < PRE; ++e) {
            if (e < len) {
                if (e + DEPTH < PRE && e + DEPTH < len) load_b(bv[(e + DEPTH) % (DEPTH + 1)], cc[(e + DEPTH) % PRE]);
#pragma unroll
                for (int n = 0; n < NC; ++n) acc[n] = mac<EXACT>(acc[n], aa[e], bv[e % (DEPTH + 1)][n]);
            }
        }
        j += len;
    }
    if (j < j1) {
        int c = ci[j];
        float a = va[j];
        while (true) {
            float bv[NC];
            load_b(bv, c);
            const float a0 = a;
            ++j;
            if (j < j1) { c = ci[j]; a = va[j]; }          // the next entry is requested before this one's products are formed
#pragma unroll
            for (int n = 0; n < NC; ++n) acc[n] = mac<EXACT>(acc[n], a0, bv[n]);
            if (j >= j1) break;
        }
    }
    if (skip && skip[r]) return;
    if constexpr (RM) {
#pragma unroll
        for (int n = 0; n < NC; n += 4)
            *reinterpret_cast<f32x4 *>(Cout + lr * ldc + col0 + n) = f32x4{epilogue<EXACT>(alpha, acc[n], beta, cin[n]), epilogue<EXACT>(alpha, acc[n + 1], beta, cin[n + 1]),
                                                                            epilogue<EXACT>(alpha, acc[n + 2], beta, cin[n + 2]), epilogue<EXACT>(alpha, acc[n + 3], beta, cin[n + 3])};
    } else {
#pragma unroll
        for (int n = 0; n < NC; ++n) Cout[lr + (int64_t)(col0 + n) * ldc] = epilogue<EXACT>(alpha, acc[n], beta, cin[n]);
    }
}

}  // namespace sx
