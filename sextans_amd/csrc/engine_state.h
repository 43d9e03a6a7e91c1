// engine_state.h -- the engine object behind the C ABI (include/sextans_amd.h) and what its translation units share:
//   engine.hip        handles, options, matrices, the SpMM dispatcher (sextans_spmm_device_rows), host-buffer entry points
//   engine_plan.hip   everything prepared once per matrix, outside every timed region: long-row split, packed panel plans
//                     (natural / clustered / reordered), window stream -- the analogue of the reference's host-side scheduling
//                     and packing (sextans-host.cpp:114-148)
//   engine_bell.hip   blocked-ELL bf16 MFMA path (BASELINE config 5) and the dense-tile extraction
//   engine_dist.hip   native multi-GPU entry (RCCL all-gather of C slabs)
// Not a public header.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <set>
#include <string>
#include <vector>

#include "sextans_amd.h"
#include "thread_stream.h"

namespace sxe {
extern thread_local std::string g_last_error;   // text behind sextans_last_error()
}

#define SX_HIP(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            char buf_[512];                                                             \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call,                 \
                     hipGetErrorString(e_), __FILE__, __LINE__);                        \
            sxe::g_last_error = buf_;                                                   \
            return SEXTANS_ERR_HIP;                                                     \
        }                                                                               \
    } while (0)

namespace sxe {
struct EventPair { hipEvent_t a, b; };
constexpr int kRowsNoFuseB = 0x100;   // internal flag of sextans_spmm_device_rows: always stage from the repacked panel
constexpr int kPanelFloats = 9216;    // at most 36 KiB of LDS for the B panel (576 rows at N-tile 16)
struct Seg { int width, col0, ntiles; int last_cols = 0; };   // N is covered by segments of equally wide tiles; last_cols != 0: valid columns of the
                                                               // segment's LAST tile (8: the tail of N = 16 t + 8 merged into the 16-column segment)
}  // namespace sxe

struct sextans_engine {
    int device = 0;
    int num_cus = 256;
    // matrix
    int M = 0, K = 0;
    int64_t nnz = 0;
    const int *d_rp = nullptr, *d_ci = nullptr;
    const float *d_v = nullptr;
    bool owns_matrix = false;
    bool device_matrix_checked = false;   // a caller-provided device matrix has been validated (row_ptr monotone, columns < K)
    // workspaces
    std::set<const void *> big_lds_kernels;   // kernels whose dynamic-LDS limit has been raised ON THIS ENGINE'S DEVICE (the
                                              // attribute is per device: a process-wide flag breaks the second GPU of a process)
    float *d_Bp = nullptr;
    size_t Bp_cap = 0;              // floats
    int bp_layout = 0;              // main panel width of the last repack into d_Bp (0 = none)
    float *d_B = nullptr, *d_Cin = nullptr, *d_Cout = nullptr;   // host-path staging
    size_t B_cap = 0, C_cap = 0;
    hipStream_t host_stream = nullptr;                           // stream of the host-buffer entry points
    float *d_chB = nullptr, *d_chC = nullptr;                    // accelerator channel layouts (sextans_invoke)
    size_t chB_cap = 0, chC_cap = 0;
    // block-dictionary plan for the LDS-panel kernel (built lazily, per lanes_per_row)
    // One packed form per lanes_per_row value (2 / 4 / 8): the active one below, the others parked in plan_stash, so
    // callers that alternate between N classes (N = 8 -> 2 lanes, N >= 16 -> 4) do not rebuild on every switch.
    struct PanelState {
        int plan_lpr = 0;               // 0 = no plan
        bool plan_all_dict = false;     // built with a dictionary for EVERY block (the few blocks without reuse would have made the plan mixed)
        int plan_sets = 1;              // row slots per block = plan_sets * (256 / plan_lpr): 2 for the short-row clustered plan (spmm_panel_v2.h: SETS)
        int64_t plan_min_reuse = -1;
        // d_dict_ptr: entries per block dictionary; d_dict: dictionaries at stride plan_dict_stride;
        // d_row_off: {first packed entry, entries} per (block, slot)
        int *d_dict_ptr = nullptr, *d_dict = nullptr, *d_blk_row = nullptr, *d_row_off = nullptr;
        int *d_pcol32 = nullptr;
        float *d_pval = nullptr;
        int plan_nblk = 0;
        std::vector<int> h_blk_row;     // host copy of the plan's block boundaries (row-range calls, sextans_align_row)
        unsigned short *d_lidx = nullptr;
        int *d_dict_blocks = nullptr;      // mixed plans, split form: the blocks that have a dictionary, ascending (n_dict_blocks of them)
        int n_dict_blocks = 0;
        int *d_rg_groups = nullptr;        // ... and the groups of 128 rows that hold a row of the gather kernel's (rg_ngroups of them)
        int rg_ngroups = 0;
        unsigned char *d_rg_skip = nullptr;   // mixed plans: 1 = the row is NOT the gather kernel's (it lies in a dictionary block, or on the piece path)
        int *d_ioff = nullptr;          // per (block, slot): start of the slot's index list in d_lidx when identical lists of consecutive rows are
                                        // stored once (plan_device.hip: share_index_lists); null = at the slot's first packed entry
        int64_t plan_idx_len = 0;       // entries of d_lidx (= plan_stream_len without sharing)
        double plan_panel_frac = 0.0;   // share of non-zeros living in dictionary blocks
        double plan_narrow_frac = 0.0;  // sampled share of non-zeros in row blocks that meet the N <= 16 threshold ("panel_min_reuse_x100")
        int plan_max_dict = 0;          // largest block dictionary (entries)
        int plan_max_row = 0;           // longest row of the planned matrix
        int64_t plan_stream_len = 0, plan_nnz_panel = 0;
        int plan_pad_row = 0;           // panel row holding +1.0f for the padding entries = panel capacity in rows
        int plan_dict_stride = 0;       // ints per block in d_dict (dictionaries padded to a common stride)
        bool plan_mixed = false;        // some block with non-zeros has no dictionary (global-gather path needed)
        bool plan_built = false;        // false: only the sampled verdict exists (no packed stream)
        bool stream_released = false;   // the packed stream (d_lidx / d_pval / d_pcol32: 6 bytes per non-zero) has been handed back while a
                                        // clustered plan serves the whole-matrix calls; restore_plan_streams() rebuilds it (same bytes)
    };
    PanelState ps;                      // active
    // The same plan over the rows in CLUSTERED order (row_cluster.hip: brick by brick for grid-stencil matrices), 4 lanes per row,
    // used by spmm_csr_panel_v2 for whole-matrix calls; row-range calls and every other kernel keep the natural-order plan above.
    PanelState psc;
    int64_t cluster_ref_dict = 0;       // panel rows of the grid-brick plan while the graph plan is weighed against it (ensure_cluster_plan)
    int *d_slot_row = nullptr;          // psc: row of the main matrix per (block, slot)
    int *d_dict_nat = nullptr;          // psc, graph clustering: the block dictionaries in the CALLER's column numbers (row-major calls read B where it lies)
    bool cluster_runs = false;          // cluster_state 1 by run-level clustering (runs of 16 consecutive rows over the graph of runs), not grid bricks
    bool cluster_for_rm = false;        // the plan is being (re)considered for row-major calls: no passes over C to pay for
    bool lean_prepare = false;         // prepare() on behalf of a row-major call: no B-panel / C-staging workspaces (nothing is repacked or staged there)
    bool cluster_rm_tried = false;      //   ... once per matrix
    bool cluster_cm_pays = true;        // the graph-clustered plan also serves column-major calls (>= 40 % fewer panel rows: it pays two passes over C)
    int *d_colpos = nullptr;            // psc, graph clustering: row of the permuted B panels that holds column c (K ints)
    float *d_Cs = nullptr;              //   ... and the row-major C staging buffer of the reordered form: [N / 16][M][16] floats
    size_t Cs_cap = 0;
    int col_lo = 0, col_hi = 0;         // columns [col_lo, col_hi) the matrix as set has entries in: the only rows of B a call repacks
    bool col_range_known = false;
    unsigned char *d_touched = nullptr; // one byte per 64 rows of B: does the matrix have a column there (null = repack everything in [col_lo, col_hi))
    int64_t touched_segments = 0;
    int colwise_state = 0;              // spmm_csr_colwise for this matrix: 0 not evaluated, 1 short rows in a numbering with locality, -1 no
    double row_coherence = 0.0;         // sampled share of consecutive rows' entries with neighbouring columns
    int cluster_decline = 0;            // why the graph clustering was declined (engine_plan.hip: cluster_graph), 0 = it was not
    double cluster_shared = 0.0;        // sampled share of a neighbour row's columns a row has too (graph clustering pre-test)
    int cluster_graph_kind = 0;         // graph the rows were clustered over: 0 the matrix itself, 1 a slab's own square pattern, 2 row similarity
    double pattern_symmetry = 1.0;      // sampled share of entries (r, c) of a square matrix whose mirror (c, r) exists
    int cluster_state = 0;              // 0 not evaluated, 1 grid bricks in use, 2 graph clustering (reordered form) in use, -1 declined
    int64_t cluster_s2 = 0, cluster_s3 = 0;
    int64_t plan_total_dict = 0, cluster_total_dict = 0;   // sum of the block dictionaries: natural order / clustered order
    PanelState plan_stash[3];           // parked, indexed by lanes_per_row 2 / 4 / 8 -> 0 / 1 / 2
    // K-windowed accumulator-resident plan (spmm_csr_window; built lazily)
    uint2 *d_wstream = nullptr;
    int *d_wstep0 = nullptr;
    int win_nwaves = 0, win_rw = 0;
    int64_t win_padded = 0;         // stream entries including padding
    int win_state = 0;              // 0 = not evaluated, 1 = built, -1 = rejected (skewed rows / K too large)
    int64_t win_built_rows = -1, win_built_cols = -1;
    double plan_build_s = 0.0;      // host seconds spent building packed forms of A for the current matrix
    // "MFMA only where a tile is actually dense" (options "mfma_dense_tiles" / "dense_tile_fill_x100"): 32x32 tiles of
    // the main matrix whose fill reaches the threshold, as a blocked-ELL bf16 side matrix; the CSR kernels keep the rest
    int dense_mb = 0, dense_W = 0;  // full block rows, ELL width (0 = no dense tile / not extracted)
    double dense_share = 0.0;       // blocks per distinct block column in groups of 8 block rows of the dense-tile matrix
    int dense_max_union = 0;
    int *d_dense_col = nullptr;
    void *d_dense_Af = nullptr;
    int64_t dense_tiles = 0, dense_nnz = 0;
    int64_t dense_built_mfma = -2, dense_built_fill = -2;
    // "mfma_dense_tiles" = 2: dense blocks of 16 rows on the fp32 matrix cores (rowblock_mfma_kernel.h); dense_tiles / dense_nnz then count them
    int rb_n = 0;                   // routed blocks
    int64_t rb_groups = 0;          // their 16 x 4 fragments
    int *d_rb_row0 = nullptr, *d_rb_gptr = nullptr, *d_rb_gcol = nullptr;   // first row, fragment range, column group per fragment
    float *d_rb_A = nullptr;        // fragments in MFMA operand order (64 floats each)
    int sb_n = 0;                   // super blocks of 4 routed blocks: ascending union of their column groups + 4-bit owner masks
    int64_t sb_entries = 0;
    int *d_sb_uptr = nullptr, *d_sb_ucol = nullptr;
    unsigned char *d_sb_umask = nullptr;
    // blocked-ELL bf16 matrix (MFMA path)
    int bell_M = 0, bell_K = 0, bell_W = 0;
    int bell_max_union = 0;         // largest number of distinct block columns inside a group of 8 block rows
    double bell_share = 0.0;        // blocks per distinct block column inside groups of 8 block rows (1 = no sharing, 8 = identical rows)
    const int *d_bell_col = nullptr;
    int *d_bell_col_owned = nullptr;
    void *d_bell_Af = nullptr;      // A blocks in MFMA fragment order (owned)
    void *d_bell_Bf = nullptr;      // B in fragment order (workspace)
    size_t bell_Bf_cap = 0;         // bytes
    // Long rows leave the "main" matrix -- the CSR arrays every kernel and plan works on, equal to the arrays
    // above when there are none -- and go through the piece path (rows sorted by length, one row group per piece):
    //   bucketed rows (longer than the bucket threshold L0, option "bucket_rows"): ONE piece, summed in order =
    //     still bit-identical to cpu_spmm_CSR; they only leave so that a workgroup of the main kernel never waits
    //     for one long row among 63 short ones;
    //   hub rows (longer than the split threshold T, option "split_rows"): pieces of T entries summed in parallel
    //     and folded in order = re-associated (stated tolerance), reported by sextans_reassociated_rows.
    // Chain of matrices: the matrix as set (d_rp / d_ci / d_v) -> [dense 32x32 tiles cut out, when routed to MFMA] ->
    // "source" (s_*) -> [long rows emptied] -> "main" (m_*).  Without dense tiles / long rows the stages alias.
    const int *s_rp = nullptr, *s_ci = nullptr;
    const float *s_v = nullptr;
    int64_t s_nnz = 0;
    int *d_srp = nullptr, *d_sci = nullptr;   // owned copy of the source (exists only when tiles were cut out)
    float *d_sv = nullptr;
    const int *m_rp = nullptr, *m_ci = nullptr;
    const float *m_v = nullptr;
    int64_t m_nnz = 0;
    int *d_mrp = nullptr, *d_mci = nullptr;   // owned compacted copy (exists only when rows left)
    float *d_mv = nullptr;
    unsigned char *d_skip = nullptr;          // 1 = the row's C is written by the piece path, not by the main kernel
    struct PieceTable {                       // pieces [begin, end) in d_ci / d_v, first piece per long row, the rows
        int *d_vrp = nullptr, *d_vend = nullptr, *d_vfirst = nullptr, *d_row = nullptr;
        std::vector<int> h_row, h_vfirst;
    };
    PieceTable by_len, by_row;                // sorted by length (whole-matrix calls: balanced workgroups) / by row (row ranges)
    // exact chains (strict order, "exact_chain" = 1): rows longer than the automatic threshold leave the piece tables too
    // and are summed by chain_fused -- still one serial chain of rounded adds per (row, column), bit-identical
    int nchain = 0;
    int *d_chain_row = nullptr, *d_chain_beg = nullptr, *d_chain_perm = nullptr;   // perm: chain rows by length, longest first
    long long *d_chain_off = nullptr;                   // prefix of the lengths
    // the chain rows' entries once more, compact, columns relabelled for the permuted B panels of the reordered form (ensure_cluster_plan)
    int *d_chain_ci_perm = nullptr, *d_chain_beg_c = nullptr;
    float *d_chain_v_c = nullptr;
    std::vector<int> h_chain_row;
    std::vector<long long> h_chain_off;
    int64_t chain_T = 0;
    int64_t chain_built_opt = -2;
    hipStream_t aux_stream = nullptr;         // the chain kernels need one or two wavefronts for ~1 ms: they run beside the main kernel
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_pipe[4] = {nullptr, nullptr, nullptr, nullptr};   // tile-group pipelining (engine.hip): fork / second-group passes done / first kernel done / side stream done
    std::vector<int> h_split_rows;            // ascending: rows cut into more than one piece
    int nhub = 0;                             // long rows (bucketed + split)
    int split_nv = 0;                         // pieces of all long rows
    int64_t split_T = 0, bucket_L0 = 0;       // thresholds in effect (0 = none)
    int64_t split_built_opt = -2, bucket_built_opt = -2, split_built_gnnz = -2;   // option values the state above was built for
    float *d_P = nullptr;
    size_t P_cap = 0;
    long long *d_dbg = nullptr;     // 8 counters for phase timing (option "phase_timing")
    // native multi-GPU form (sextans_dist_spmm): slab staging S[chunk][world][N][lmax_chunk], communication stream
    float *d_stage = nullptr;
    size_t stage_cap = 0;
    int64_t graph_fallbacks = 0;        // host-entry repeat loops whose hipGraph capture failed (another thread touched the legacy stream): launched one by one instead
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> dist_events;
    float *d_rmB = nullptr, *d_rmC = nullptr;   // column-major copies of the row-major entry point's fallback path
    size_t rmB_cap = 0, rmC_cap = 0;
    float *d_Cfull = nullptr;           // clustered-order chunks: row-major staging of the WHOLE C ([N / 16][M_total][16]) the received slabs are scattered into
    size_t Cfull_cap = 0;
    int *d_dist_rows = nullptr;         //   ... and every rank's position -> global row table ([world][longest slab])
    size_t dist_rows_cap = 0;
    bool dist_cc = false;               //   ... in use for the partition of dist_cut_key (every rank agreed)
    int64_t dist_exchanges = 0;         // control collectives + host synchronisations the dist entry points have performed (stat "dist_setup_exchanges")
    std::vector<int> dist_nnz_key;      // (ranges, rank) the whole matrix's non-zero count was exchanged for
    std::vector<int> dist_cut_key, dist_cuts;   // (ranges, N, nchunks, rank) the chunk cuts of all ranks were exchanged for
    std::vector<int> dist_meta;     // {first row, rows} per (chunk, rank) as last uploaded, and where
    const int *dist_meta_at = nullptr;
    // options
    int64_t opt_share_index = 1;        // plans at 4 lanes per row: consecutive rows with identical 16-bit index lists (dof rows of a mesh node) share one copy
    int64_t opt_refine_rows = 62;       // ... rows per block before the refinement (64 - room for rows that move in)
    int64_t opt_refine_sweeps = 8;      // graph clustering: sweeps of the block refinement (0 = blocks are runs of 64 rows of the merge-tree order)
    int64_t opt_run_cluster = 0;        // run-level clustering for matrices whose graph plan is not worth the reordered form: 0 off (measured: no gain), 1 when >= 10 % fewer panel rows, 2 always
    int64_t opt_reordered_xcd = -1;     // measurement switch: workgroup placement of the reordered form (0 round-robin over the XCDs, 1 contiguous chunks, -1 the built-in rule)
    int64_t opt_row_similarity = -1;    // graph clustering over the row-similarity graph: -1 when the matrix is rectangular or its pattern unsymmetric, 0 never, 1 always
    int64_t opt_relabel_columns = 1;    // graph clustering: B rows relabelled in first-touch order (permuted panels); 0 = natural panels
    int64_t opt_colwise_tiles_adjacent = 1;   // lane-per-row kernel, N >= 32: the tiles of a row block neighbours in the launch order (one XCD, same time)
    int64_t opt_split_mixed = 1;         // mixed plans: dictionary blocks on spmm_csr_panel_v2, the rows of the other blocks on the gather kernel (0: spmm_csr_panel<MIXED>)
    int64_t opt_small_panel = 1;        // clustered plans of short-row matrices are packed for a 320-row panel when every dictionary fits (more workgroups per CU)
    int64_t opt_cluster_top = 1 << 30;  // graph clustering: the aggregation stops when clusters reach this many rows.  Default: never -- the whole
                                        // matrix becomes one merge tree, so that each XCD's contiguous chunk of row blocks is one region of the graph and
                                        // B lines are shared inside its L2 (4M-row FEM, random node order, N = 16, same box: 771 us with 4096-row
                                        // clusters in arbitrary order, 727 with 65536, 694 with the full tree)
    int64_t opt_colwise_max_len = 6;    // spmm_csr_colwise is considered for matrices whose mean row length is at most this (5- and 7-point
                                        // stencils; measured on 4M-row matrices: 5 entries per row 361 -> 259 us per step at N = 16, 9 entries 368 -> 407)
    int64_t opt_kernel = 0, opt_lpr = 0, opt_stage = 1, opt_xcd = 1, opt_exact = 1, opt_profile = 0;   // opt_lpr 0 = auto
    int64_t opt_cols_per_lane = 0;      // LDS-panel kernel: output columns per lane.  4 (= 0, the default) = 16-column tiles;
                                        // 8 = register-blocked 32-column super tiles (spmm_csr_panel_v2<2>: 2 workgroups per
                                        // CU -- measured slower than 4 columns per lane at 4 workgroups per CU)
    int64_t opt_cluster_shape = 0;      // measurement switch: brick shape run_rows * 10000 + lines * 100 + planes (0 = 16 x 2 x 2 / 16 x 4)
    int64_t opt_cluster_group = 6;      // bricks are laid out in groups of g x g brick columns (same-box A/B, g = 3 -> 6: FEM 3-dof -1 .. -2 %, 27-point 1-dof -5 %, 2-D 9-point -4 .. -5 %)
    int64_t opt_row_cluster = -1;       // clustered-order plan for spmm_csr_panel_v2 (ensure_cluster_plan): -1 auto, 0 never, 1 whenever one
                                        // can be built, 2 = graph clustering (reordered form) also where the grid bricks would apply
    int64_t opt_row_offset = -1;        // this engine's matrix is the row slab [row_offset, row_offset + M) of a K x K matrix (set by sextans_dist_spmm): lets the
                                        // graph clustering run on a rank's slab (edges to other ranks' rows ignored); -1 = unknown (only square matrices cluster)
    int64_t opt_row_sets = 2;           // clustered grid plan of a short-row matrix (rows <= 32 entries): 128-row bricks = two 64-slot row sets per block and panel; 2 = 3-D grids, 3 = 2-D grids too, 1 = never
    int64_t opt_pipeline_tiles = 0;     // N >= 32 on spmm_csr_panel_v2: 1 = the 16-column tiles run as two groups and the layout passes of the second
                                        // group go to the side stream under the first group's kernel.  Built and measured: the second
                                        // pass over the packed A stream costs more than the hidden repack saves (FEM 4M, N = 128: 3.94 -> 4.09 ms
                                        // per step, N = 32: 1.12 -> 1.52 ms), so the default is 0 = one launch over all tiles
    int64_t opt_small_v2 = 1;           // measurement switch: 0 = small matrices keep the full-capacity, 4-deep form of spmm_csr_panel_v2
    int64_t opt_panel_v2 = -1;          // 16-column tiles on the register-resident form (spmm_csr_panel_v2<1>: row entries
                                        // loaded once per block, panels by LDS-DMA, tile loop inside the workgroup, C stored
                                        // straight from the accumulators): 1 = yes, 0 = no (spmm_csr_panel), -1 = auto: yes
                                        // unless the column-major staging of small matrices applies ("fuse_b")
    int64_t opt_tiles_per_wg = 0;       // wide kernel: super tiles one workgroup walks (A stream from HBM once per that many
                                        // columns); 0 = auto: all of N while the launch still fills the chip several times
    int64_t opt_fuse_b = 1;             // panel kernel may stage from column-major B (small matrices: no repack launch)
    int64_t opt_split_rows = 0;         // 0 (default) = never: every row is summed in strict CSR order, bit-identical to
                                        // cpu_spmm_CSR; > 0: rows longer than this are split (re-associated, opt-in);
                                        // -1 = opt in with the automatic threshold max(1024, global nnz / 16384)
    int64_t opt_global_nnz = 0;         // multi-GPU: non-zeros of the WHOLE matrix (0 = this engine's matrix is the whole
                                        // matrix), so every rank derives the same split threshold as a single GPU would
    int64_t opt_exact_chain = 1;        // strict order ("split_rows" = 0): rows longer than max(1024, nnz / 16384) are summed as
                                        // exact chains (all products in parallel, one lane per column adds them in order);
                                        // 0 = such rows stay on the piece path (one row group, ~80 ns per entry)
    int64_t opt_bucket_rows = -1;       // > 0: rows longer than this take the piece path unsplit (still exact); 0 = off;
                                        // -1 = max(32, 2 * mean row length)
    int64_t opt_phase_timing = 0;       // 1: panel kernel accumulates per-phase wave cycles (debug aid)
    int64_t opt_min_reuse_x100 = 200;   // a block uses the LDS panel if nnz >= 2 * distinct columns (measured: a 1-dof 3-D
                                        // stencil, reuse 2.9, runs 18 % faster on the panel kernel; FEM/banded classes unchanged)
    int64_t opt_min_reuse_wide_x100 = 150;   // the same threshold for N >= 32: with more columns per B row read the panel pays
                                        // earlier (2-D 5-point stencil, reuse 1.65: N = 128 2.28 ms vs 2.60 ms on the gather kernel,
                                        // N = 16 0.370 vs 0.350 ms)
    int64_t opt_win_rows = 319;         // rows per wavefront of the window kernel (+1 dummy row: 4 x 320 x 32 B = 40 KiB)
    int64_t opt_win_cols = 65536;       // columns per K window (x 32 B = 2 MiB of the 8-column panel: half an XCD's L2)
    int64_t opt_win_unroll = 8;         // steps in flight per ring (4 or 8)
    int64_t opt_win_auto = 0;           // 1: "kernel" 0 may pick the window kernel from the fabric-byte model
    int64_t opt_bell_shared = -1;       // N = 256: workgroups of 8 block rows share each B tile through an LDS ring
                                        // (spmm_bell_mfma_shared).  1 = always, 0 = never, -1 = when the 8 block rows of a
                                        // workgroup share block columns (blocks per distinct column >= 1.5)
    int64_t opt_rb_tiles = 0;           // measurements only: tiles of 16 columns per wavefront of the fp32 row-block MFMA kernel (0 = 4 where N allows)
    int64_t opt_mode = 0;               // SEXTANS_MODE_* as last set through option "mode"
    int64_t opt_dist_broadcast_runs = 0;   // measurements / tests only: sextans_dist_spmm_rm exchanges ranges of EQUAL length by grouped broadcasts too
    int64_t opt_bell_debug = 0;         // measurements only (wrong results): ablation bits of spmm_bell_mfma_shared
    int64_t opt_bell_gen = 0;           // block rows per launch of the wide kernel (0 = all in one launch)
    int64_t opt_bell_wide = 1;          // 1 (default): N = 256 runs one wavefront per block row over all 8 column tiles
                                        // (A requested once, non-temporal); 0: two wavefronts of 4 tiles each
    int64_t opt_mfma_dense = 0;         // 1: dense 32x32 tiles run on the bf16 MFMA path (the caller opts into bf16 rounding
                                        // of those tiles and of B for them); 0: they are only counted (get_stat)
    int64_t opt_dense_fill_x100 = 50;   // a tile is dense when it holds >= this percentage of its 1024 positions
    // profiling
    std::vector<sxe::EventPair> ev_kernel, ev_repack, ev_post;   // ev_post: passes behind the kernel (C staging -> C)
    const char *last_kernel = "none";
    std::string last_kernel_buf;        // storage for composed names
};

namespace sxe {

int check_device(int device);
void free_panel_state(sextans_engine::PanelState &p);
void free_plan(sextans_engine *h);
void free_cluster_plan(sextans_engine *h);
int64_t plan_key(const sextans_engine *h);
int64_t device_bytes(const sextans_engine *h);   // what the engine holds in HBM right now (stat "device_bytes")
void free_window(sextans_engine *h);
void free_bell(sextans_engine *h);
void free_split(sextans_engine *h);
void free_dense(sextans_engine *h);
void free_matrix(sextans_engine *h);
int ensure(float **p, size_t *cap, size_t need);
int allow_big_lds(sextans_engine *h, const void *kern, int bytes);
int read_back_row_ptr(sextans_engine *h, std::vector<int> &rp, int level = 2);
int read_back_entries(sextans_engine *h, std::vector<int> &ci, std::vector<float> &va, int level = 2);
int ensure_plan(sextans_engine *h, int lpr, bool force);
int ensure_cluster_plan(sextans_engine *h);
int restore_plan_streams(sextans_engine *h);   // the natural-order plan's packed stream, if it was released
int ensure_colwise(sextans_engine *h);
int ensure_col_range(sextans_engine *h);
int ensure_window(sextans_engine *h, bool force);
bool window_pays(const sextans_engine *h, int N, int64_t padded);
int ensure_split(sextans_engine *h);
int ensure_dense(sextans_engine *h);                       // engine_bell.hip
int prepare(sextans_engine *h, int N, std::vector<Seg> &plan, int &W, bool &use_panel, bool &use_window, bool whole = true);
// dense 32x32 tiles on the matrix cores (engine_bell.hip): C_out = alpha * (A_dense * bf16(B)) + beta * C_in for the full block rows
int launch_dense_tiles(sextans_engine *h, int N, float alpha, const float *d_B, int64_t ldb, float beta, const float *d_C_in,
                       int64_t ldc_in, float *d_C_out, int64_t ldc, hipStream_t s);
// dense row blocks on the fp32 matrix cores (engine_bell.hip): the routed blocks of [row_begin, row_end) from the B panels in d_Bp
int launch_rowblocks(sextans_engine *h, const std::vector<Seg> &plan, const float *d_C_in, int64_t ldc_in, float *d_C_out, int64_t ldc, int N, int row_begin,
                     int row_end, float alpha, float beta, hipStream_t s);
int mark_rowblock_skip(sextans_engine *h);   // after ensure_split: the routed rows join the rows the CSR kernels never write

// clustered-order chunks of sextans_dist_spmm (engine.hip)
int rm_plan(sextans_engine *h, int N, std::vector<Seg> &plan, int &W, bool &use_panel, bool &use_window, hipStream_t s);   // planning half of sextans_spmm_device_rm
int cc_prepare(sextans_engine *h, int N, bool *ok);
void cc_table(sextans_engine *h, int row0, int *d_out, hipStream_t s);
void cc_pre(sextans_engine *h, int N, const float *d_B, int64_t ldb, const float *d_C_in_slab, int64_t ldc_in, hipStream_t s);
int cc_chunk(sextans_engine *h, int N, float alpha, float beta, int b0, int b1, const int *d_rows, int row0, float *slab, int64_t lmax, hipStream_t s);
void cc_scatter(const float *slab, int64_t lmax, const int *d_rows, int n, float *tiles, int64_t tile_stride, int N, hipStream_t s);
void cc_finish(const float *tiles, float *C, int64_t ldc, int M_total, int N, hipStream_t s);

template <class T>
int upload(T **dst, const std::vector<T> &src) {
    SX_HIP(hipMalloc((void **)dst, sizeof(T) * (src.empty() ? 1 : src.size())));
    if (!src.empty()) SX_HIP(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    return SEXTANS_OK;
}

struct PlanTimer {   // accumulates host seconds spent packing A (reported by sextans_get_stat "plan_build_s")
    sextans_engine *h; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit PlanTimer(sextans_engine *h_) : h(h_) {}
    ~PlanTimer() { h->plan_build_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

struct Prof {   // HIP events around a launch group on the launch stream (option "profile")
    sextans_engine *h; std::vector<EventPair> *vec; hipStream_t s; bool on; EventPair ep{};
    Prof(sextans_engine *h_, std::vector<EventPair> *v, hipStream_t s_) : h(h_), vec(v), s(s_), on(h_->opt_profile != 0) {
        if (on) {
            (void)hipEventCreate(&ep.a); (void)hipEventCreate(&ep.b);
            (void)hipEventRecord(ep.a, s);
        }
    }
    ~Prof() {
        if (on) { (void)hipEventRecord(ep.b, s); vec->push_back(ep); }
    }
};

}  // namespace sxe
