// graph_cluster.h -- structure-agnostic row clustering for the LDS-panel plan (csrc/graph_cluster.hip).
//
// The reference schedules the non-zeros of ANY matrix for its on-chip B window (generate_edge_list_for_all_PEs,
// sparse_helper.h:345-403).  The round-3 clustered plan of this engine (row_cluster.hip) needs Cartesian-grid structure in natural
// ordering; this is the general form: a multilevel pairwise aggregation of the rows over the matrix graph, built on the device.
#pragma once
#include <cstdint>
#include <string>

namespace sx {

// Share of a sampled row's columns that a neighbouring row (one of its own column indices, read as a row) has too, averaged
// over `nsample` rows spread over the matrix: ~0 for matrices with random columns, 0.3 .. 0.7 for mesh / stencil matrices in
// ANY numbering.  *near_fraction = share of the sampled rows' entries within M / 64 of the diagonal (does the numbering have
// locality?).  Needs M == K.  Returns non-zero on a HIP error.
// *symmetric_fraction (optional) = share of the sampled entries (r, c) whose mirror (c, r) exists (1.0 = symmetric pattern).
int probe_shared_neighbourhood_device(int M, const int *d_rp, const int *d_ci, int nsample, double *shared_fraction, double *near_fraction,
                                      std::string &err, double *symmetric_fraction = nullptr);

// Share of the j-th entries of sampled consecutive rows (r, r + 1) whose columns differ by at most 32: ~1 for stencil / banded /
// generator-ordered mesh matrices, ~0 for random columns or random numberings (dispatcher test of spmm_csr_colwise).
int probe_row_coherence_device(int M, const int *d_rp, const int *d_ci, int nsample, double *close_fraction, std::string &err);

// order[i] = row of the matrix at position i of the clustered order (M ints on the device, caller frees).
// Rows are merged pairwise, level by level (cluster sizes 1 -> 2 -> 4 ... -> max_cluster_rows), each cluster with the unmatched
// neighbouring cluster it shares the most neighbourhood with; a merged pair's rows become contiguous, so the final order is the
// leaf order of the merge tree: any run of consecutive rows is a graph-compact set.  M == K required (a column index is read as
// the row of the neighbour).  Returns 0 = built, 1 = declined (not square, empty), 2 = HIP error (err set).
// d_weights (optional): one byte per entry of the pattern, the weight of that edge, instead of the shared-neighbourhood count
// computed here (the row-similarity graph below brings its own).
int cluster_rows_graph_device(int M, int K, int64_t nnz, const int *d_rp, const int *d_ci, int max_cluster_rows, int **d_order,
                              std::string &err, const unsigned char *d_weights = nullptr, int snapshot_limit = 0, int **d_snapshot = nullptr);
// (d_snapshot, optional: cluster number of every row at the level where clusters hold up to snapshot_limit rows; later levels only
// concatenate whole clusters, so in the final order a change of that number is a cluster boundary.  M ints on the device, caller frees.)

// Row-similarity graph of a RECTANGULAR matrix (the reference schedules any M x K matrix: sparse_helper.h:345-403): row r joined to
// the 16 rows that share the most columns with it (found through the transposed pattern), weight = shared columns.  A square
// M x M pattern with exactly 16 slots per row (-1 = empty; all consumers skip indices outside [0, M)), g_w one byte per slot; the
// caller frees the three arrays.  *shared_fraction = best overlap / row length over a sample of rows (the pre-test of
// probe_shared_neighbourhood_device for matrices without "row c"), *near_fraction = share of entries near the scaled diagonal.
// Returns 0 = built, 1 = declined, 2 = HIP error.
int row_similarity_graph_device(int M, int K, int64_t nnz, const int *d_rp, const int *d_ci, int **g_rp, int **g_ci, unsigned char **g_w,
                                int64_t *g_nnz, double *shared_fraction, double *near_fraction, std::string &err);

// Block refinement on top of the clustered order: the order is cut into blocks of `per` rows, `sweeps` sweeps of capacity-constrained
// label propagation move boundary rows to the neighbouring block that holds more of their neighbours (at most `cap` rows per block);
// d_order is rewritten (blocks in their old sequence, rows inside a block in their old order) and cut[i] = 1 where a block starts
// (M bytes on the device, caller frees; the plan builder starts a row block there).  Deterministic.
int refine_blocks_device(int M, const int *d_rp, const int *d_ci, int *d_order, int per, int cap, int sweeps, unsigned char **d_cut,
                         std::string &err);

// colpos[c] = new position of column c: columns in the order in which the rows of `order` first touch them (untouched columns
// last), so that the dictionary of a run of consecutive rows is (mostly) a run of consecutive new positions -- whole cache lines
// of the relabelled B panel.  K ints on the device, caller frees.
int column_first_touch_order_device(int M, int K, const int *d_rp, const int *d_ci, const int *d_order, int **d_colpos, std::string &err);

// Row SLAB of a square matrix (rows [row_offset, row_offset + M) of a K x K matrix: what a rank of the row-partitioned SpMM holds): the
// square pattern of the slab's own rows -- entries whose column c lies in [row_offset, row_offset + M), as c - row_offset; edges to
// rows the slab does not hold are dropped.  out_rp (M + 1) / out_ci on the device, caller frees.  The clustering above runs on it.
int local_square_pattern_device(int M, const int *d_rp, const int *d_ci, int row_offset, int **out_rp, int **out_ci, int64_t *out_nnz,
                                std::string &err);

// G + G^T of a square pattern (entries outside [0, M) are dropped): every row keeps its own entries and gets, behind them, the rows
// that point at it and that it does not hold itself; d_w (optional) = one weight byte per entry, mirrored with its entry.  The
// handshake matching of the clustering needs symmetric weights (graph_cluster.hip).  s_rp (M + 1) / s_ci / s_w on the device, caller
// frees.  Returns 0 = built, 1 = declined (empty / too large), 2 = HIP error.
int symmetrize_graph_device(int M, int64_t nnz, const int *d_rp, const int *d_ci, const unsigned char *d_w, int **s_rp, int **s_ci,
                            unsigned char **s_w, int64_t *s_nnz, std::string &err);

// Graph of RUNS of `run` consecutive rows (node R = rows [R run, R run + run), R -> c / run for every entry, deduplicated, no self
// loops) and the expansion of an order of the runs into an order of the rows (+ block cuts every `runs_per_block` runs): the
// run-level clustering of engine_plan.hip (cluster_runs) -- matrices in a numbering with locality whose row blocks are cut short by
// the panel capacity get blocks of 4 well-chosen runs instead of 64 consecutive rows, without the reordered form's passes.
// run_graph_device: 0 = built, 1 = declined (a run with more than 512 neighbouring runs, too few rows), 2 = HIP error.
int run_graph_device(int M, int run, const int *d_rp, const int *d_ci, int **r_rp, int **r_ci, unsigned char **r_w, int64_t *r_nnz, int *Mr_out,
                     std::string &err);   // r_w: entries of the run's rows that lie in the neighbouring run (1 .. 255)
int expand_run_order_device(int M, int Mr, int run, int runs_per_block, const int *d_order_r, const int *d_group, int **d_order, unsigned char **d_cut,
                            std::string &err);

// in place: ci[j] = colpos[ci[j]]
int relabel_columns_device(int64_t nnz, int *d_ci, const int *d_colpos, std::string &err);

}  // namespace sx
