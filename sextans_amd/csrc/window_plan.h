// window_plan.h -- the K-windowed, accumulator-resident form of A consumed by spmm_csr_window.
//
// This is the reference's own dataflow for inputs WITHOUT column locality: Sextans never gathers B per
// non-zero from far memory.  It sweeps K in windows (WINDOW_SIZE = 4096 columns, sextans.h:11,15),
// streams the B window on chip (PEG_Bmtx local_B, sextans.cpp:337,353-381), keeps every output row's
// partial sums resident in on-chip accumulators for the whole sweep (PEG_Cmtx URAM, sextans.cpp:462-570)
// and feeds the PEs a non-zero stream that the host has pre-bucketed per (PE, window)
// (generate_edge_list_for_all_PEs, sparse_helper.h:345-403).  On CDNA4:
//
//   reference                                          here
//   ------------------------------------------------   ------------------------------------------------
//   PE p owns rows with row % 64 == p                   wavefront g owns rows [g*RW, (g+1)*RW): their
//                                                         N-tile-8 partial sums live in its 10 KiB of LDS
//                                                         for the whole sweep (160 KiB per CU = 5120 rows)
//   K window of 4096 columns in BRAM                    K window of 65536 columns = 2 MiB of the 8-column
//                                                         B panel, resident in the XCD's 4 MiB L2: all
//                                                         wavefronts sweep the windows in the same order,
//                                                         so every 128-byte line is fetched about once per
//                                                         XCD and sweep instead of once per non-zero
//   per-(PE, window) edge list, bubbles where a row     per-wavefront stream of 32-entry steps ordered by
//     would hit its own accumulator too early             (window, rank in row, row); no step holds a row
//     (10-slot spacing, sparse_helper.h:318-327)          twice (each step is one LDS read-modify-write
//                                                         per lane pair), padding entries hit a dummy row
//   64-bit word col14 | row18 | fp32                    64-bit word  fp32 , row9 << 23 | col23
//
// Per-row order is the ascending column order of the CSR arrays: windows ascend, and inside a window a
// row's entries keep their order (rank), so the sum is formed exactly as cpu_spmm_CSR forms it
// (sparse_helper.h:279-289) -- bit-identical results.
#pragma once
#include <cstdint>
#include <vector>

namespace sx {

constexpr int kWinStep = 32;           // entries per step (one per lane pair of a wavefront)
constexpr int kWinColBits = 23;        // entry word: row_local << 23 | column  => K <= 2^23
constexpr int kWinMaxRowsPerWave = 510;
constexpr int kWinUnroll = 24;         // every wavefront's stream is padded to a multiple of this many steps
                                       // (the kernel consumes 3 blocks of 4 or 8 steps per loop iteration)
constexpr int kWinTailSteps = 32;      // zero steps after the last stream (the kernel prefetches ahead)

struct WinEntry {
    float val;
    uint32_t word;                     // row_local << 23 | column
};

struct WindowPlan {
    int rows_per_wave = 0;             // RW; the dummy (padding) row has local index RW
    int window_cols = 0;
    int nwaves = 0;                    // ceil(M / RW)
    std::vector<int> wave_step0;       // nwaves + 1: first step of each wavefront's stream
    std::vector<WinEntry> stream;      // (steps + kWinTailSteps) * 32 entries
    int64_t nnz = 0;
    int64_t padded = 0;                // entries including padding (steps * 32, without the tail)
};

// Build the plan on the host (all cores).  Returns false when the matrix cannot be expressed
// (K > 2^23 or more than 2^31 - 1 steps).
// Cheap lower bound of the padded stream length (entries) without building anything: a wavefront needs at
// least max(ceil(nnz / 32), longest row) steps.  The dispatcher rejects skewed matrices with it.
int64_t window_plan_padded_lower_bound(int M, const int *row_ptr, int rows_per_wave);

bool build_window_plan(int M, int K, const int *row_ptr, const int *col_idx, const float *val,
                       int rows_per_wave, int window_cols, WindowPlan &out);

}  // namespace sx
