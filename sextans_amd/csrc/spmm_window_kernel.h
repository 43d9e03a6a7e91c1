// spmm_window_kernel.h -- K-windowed SpMM with accumulator-resident output rows: the CDNA4 form of the
// reference's dataflow for inputs without column locality (B window on chip: PEG_Bmtx local_B,
// sextans.cpp:337,353-381; resident partial sums: PEG_Cmtx, sextans.cpp:462-570; pre-bucketed non-zero
// stream: sparse_helper.h:345-403).  Input = the stream built by window_plan.cpp.
//
// One wavefront owns RW consecutive rows for the whole kernel.  Their partial sums for one 8-column N tile
// (the reference's N tile, sextans.cpp:57-60) sit in the wavefront's private LDS slice ((RW + 1) x 32
// bytes; row RW is the dummy row padding entries point at).  The wavefront walks its stream in steps of 32
// entries, one entry per lane PAIR: both lanes read the same 8-byte entry, each gathers its 16-byte half
// of the entry's B row from the 8-column panel (one 32-byte run per pair), then does one LDS
// read-modify-write of its half of the row's accumulator.  No step contains a row twice and LDS operations
// of a wavefront execute in order, so every accumulator sees its products in stream order = ascending
// column order = the order of cpu_spmm_CSR (sparse_helper.h:279-289): bit-identical results.
//
// Wavefronts never talk to each other: no barrier anywhere.  All of them sweep K in the same window order
// at about the same rate, so the XCD's L2 holds the current 2 MiB window of the panel and each 128-byte
// line of B crosses the fabric about once per XCD and sweep.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spmm_csr_kernels.h"

namespace sx {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));   // one stream entry {fp32 value bits, row << 23 | column}: the value sits
                                                                   // in the even register of the pair, where v_pk_mul_f32 can broadcast it
constexpr int kWinNT = 8;        // N-tile width (floats)
constexpr int kWinWaves = 4;     // wavefronts per workgroup

// stream: steps of 32 {val, word}; wave g's steps are [wave_step0[g], wave_step0[g + 1]) (a multiple of
// 3 * UNROLL), followed by at least 3 * UNROLL readable steps.  Bp: row-major K x 8 panels at panel_stride.
// Cin / Cout address row `row_base` as their row 0.  Grid: (wave_end - wave_begin + 3) / 4 workgroups per
// tile, tile-major (all row blocks of tile 0 first: one B panel at a time in the L2).
template <bool EXACT, int UNROLL>
__global__ __launch_bounds__(kWinWaves * 64, 4) void spmm_csr_window(
    const u32x2 *__restrict__ stream, const int *__restrict__ wave_step0, const float *__restrict__ Bp,
    int64_t panel_stride, const float *Cin, int64_t ldc_in, float *Cout, int64_t ldc, int M, int RW,
    int wave_begin, int wave_end, int nwg_per_tile, int row_base, float alpha, float beta,
    const unsigned char *__restrict__ skip) {
    extern __shared__ __attribute__((aligned(16))) float lds_acc[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int tile = blockIdx.x / nwg_per_tile;
    const int g = __builtin_amdgcn_readfirstlane(wave_begin + (int)(blockIdx.x % nwg_per_tile) * kWinWaves + wv);
    if (g >= wave_end) return;   // whole wavefront; nobody waits for it (no barriers in this kernel)

    float *acc = lds_acc + (size_t)wv * (size_t)(RW + 1) * kWinNT;
    for (int i = lane; i < (RW + 1) * 2; i += 64) *reinterpret_cast<f32x4 *>(acc + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};

    const int s0 = wave_step0[g], s1 = wave_step0[g + 1];
    const int p = lane >> 1, q = lane & 1;
    const u32x2 *sp = stream + (int64_t)s0 * 32 + p;
    const float *bq = Bp + (int64_t)tile * panel_stride + 4 * q;
    char *accq = reinterpret_cast<char *>(acc) + 16 * q;

    // While block k (UNROLL steps) is consumed, the B-row gathers of block k + 1 and the entry words of block
    // k + 2 are in flight: a gather is issued UNROLL steps before its data is needed, an entry UNROLL steps
    // before its gather and 2 * UNROLL steps before its accumulation.  An entry therefore lives for two
    // blocks; three register arrays take the roles {being accumulated, gather source, load destination} in
    // turn, so no register ever has to be copied (a rotating ring makes the compiler wait for each fresh load
    // just to move it).  The body is branch-free: waits carry exact vmcnt values (loads retire in order).
    // Reads run up to 2 * UNROLL steps past the wavefront's own stream: into the next wavefront's steps or
    // the zero tail -- valid columns, never accumulated.
    u32x2 e0[UNROLL], e1[UNROLL], e2[UNROLL];
    f32x4 bb[UNROLL];
    auto gather = [&](const u32x2 &e) {
        return *reinterpret_cast<const f32x4 *>(bq + (int64_t)(e.y & 0x7fffffu) * kWinNT);
    };
    auto rmw = [&](const u32x2 &e, const f32x4 &b) {
        char *ap = accq + (e.y >> 23) * (kWinNT * 4);
        f32x4 a = *reinterpret_cast<f32x4 *>(ap);
        const float v = __uint_as_float(e.x);
        a.x = mac<EXACT>(a.x, v, b.x);
        a.y = mac<EXACT>(a.y, v, b.y);
        a.z = mac<EXACT>(a.z, v, b.z);
        a.w = mac<EXACT>(a.w, v, b.w);
        *reinterpret_cast<f32x4 *>(ap) = a;
    };
    // one block: accumulate entries `ec` with the gathered rows, gather for `eg`, refill `ec` two blocks ahead
    auto block = [&](u32x2 (&ec)[UNROLL], const u32x2 (&eg)[UNROLL], u32x2 (&el)[UNROLL]) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u32x2 e = ec[u];
            const f32x4 b = bb[u];
            bb[u] = gather(eg[u]);
            el[u] = __builtin_nontemporal_load(sp + u * 32);
            rmw(e, b);
            __builtin_amdgcn_sched_barrier(0);   // keep every load where it is written: the scheduler otherwise
                                                 // sinks the gathers next to their uses (one register set, vmcnt(0))
        }
        sp += UNROLL * 32;
    };
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) e0[u] = __builtin_nontemporal_load(sp + u * 32);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) e1[u] = __builtin_nontemporal_load(sp + (UNROLL + u) * 32);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) bb[u] = gather(e0[u]);
    sp += 2 * UNROLL * 32;

    for (int s = s0; s < s1; s += 3 * UNROLL) {   // the stream of a wavefront is a multiple of 3 * UNROLL steps
        block(e0, e1, e2);
        block(e1, e2, e0);
        block(e2, e0, e1);
    }

    // Epilogue: lane = row; eight coalesced column-major accesses per row group of 64.
    const int row0 = g * RW;
    const int nrows = min(RW, M - row0);
    const int64_t col0 = (int64_t)tile * kWinNT;
    for (int r = lane; r < nrows; r += 64) {
        if (skip && skip[row0 + r]) continue;   // this row's C comes from the piece path (long rows)
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(acc + r * kWinNT);
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(acc + r * kWinNT + 4);
        const float a8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const int64_t lr = (int64_t)(row0 + r - row_base);
        float c8[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) c8[n] = Cin[lr + (col0 + n) * ldc_in];
#pragma unroll
        for (int n = 0; n < 8; ++n) Cout[lr + (col0 + n) * ldc] = epilogue<EXACT>(alpha, a8[n], beta, c8[n]);
    }
}

}  // namespace sx
