/*
 * sextans_amd.h -- C ABI of the MI355X-native SpMM engine (drop-in boundary for the Sextans path).
 *
 * Plain C types only: pointers, sizes, scalars.  No torch / C++ types cross this boundary.
 * Every entry point names the reference interface it replaces (paths relative to
 * linghaosong/Sextans, branch tapa).  All functions return 0 (SEXTANS_OK) on success or a
 * SEXTANS_ERR_* code; nothing here ever calls exit() (the reference's loader does,
 * sparse_helper.h:100-109,120-123,146-149,181-191).
 *
 * Conventions shared with the reference:
 *   - A is M x K sparse, B is K x N dense, C is M x N dense, C = alpha*A*B + beta*C
 *     (sparse_helper.h:273-277);
 *   - dense matrices are COLUMN MAJOR fp32 (B[k + ldb*n], C[m + ldc*n]);
 *   - indices are 0-based 32-bit ints, values fp32;
 *   - N is used as given by the caller; the CLI rounds it up to a multiple of 8 first
 *     (sextans-host.cpp:51) via sextans_round_up_n().
 *
 * The compute entry points run ONLY on an AMD GPU (gfx950).  There is no CPU fallback: without
 * a usable HIP device they return SEXTANS_ERR_NO_DEVICE.
 */
#ifndef SEXTANS_AMD_H
#define SEXTANS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ error codes */
#define SEXTANS_OK 0
#define SEXTANS_ERR_OPEN 1         /* "Could not open ..."                  sparse_helper.h:181-184 */
#define SEXTANS_ERR_BANNER 2       /* "Could not process Matrix Market banner" sparse_helper.h:100-103 */
#define SEXTANS_ERR_SIZE 3         /* "Could not read Matrix Market format"  sparse_helper.h:105-109 */
#define SEXTANS_ERR_NOT_COORD 4    /* "... is not a coordinate file!"        sparse_helper.h:188-191 */
#define SEXTANS_ERR_COMPLEX 5      /* "complex matrix, not supported yet!"   sparse_helper.h:120-123 */
#define SEXTANS_ERR_INDEX 6        /* index < 1 (reference) or > M/K (ours)  sparse_helper.h:146-149 */
#define SEXTANS_ERR_ALLOC 7
#define SEXTANS_ERR_PARSE 8        /* malformed entry (the reference reads garbage silently) */
#define SEXTANS_ERR_INVALID 9      /* bad argument */
#define SEXTANS_ERR_NO_DEVICE 10   /* no HIP device / wrong architecture: NO CPU fallback */
#define SEXTANS_ERR_HIP 11         /* a HIP runtime call failed; see sextans_last_error() */
#define SEXTANS_ERR_STATE 12       /* e.g. spmm before set_matrix */
#define SEXTANS_ERR_PEER 13        /* a collective preparation failed on ANOTHER rank (sextans_dist_prepare); this rank's own work was fine */

#define SEXTANS_FMT_CSR 0          /* enum MATRIX_FORMAT {CSR, CSC}, sparse_helper.h:20 */
#define SEXTANS_FMT_CSC 1

const char *sextans_error_string(int code);
/* Thread-local text of the last HIP failure (empty string if none). */
const char *sextans_last_error(void);

/* ------------------------------------------------------------------ L2: host sparse library */

/* Replaces read_suitsparse_matrix (sparse_helper.h:169-259): loads a SuiteSparse Matrix-Market
 * coordinate file into CSR (ptr has M+1 entries, idx = columns) or CSC (ptr has K+1 entries,
 * idx = rows).  Same observable semantics: case-insensitive banner, real/integer/pattern,
 * pattern -> 1.0f, entries whose fp32 bits are exactly 0 are dropped (-0.0 is kept), only
 * `symmetric` mirrors off-diagonals, duplicates kept in file order, nnz is the post-drop
 * post-mirror count.  Output arrays are malloc'ed; release with sextans_host_free(). */
int sextans_mtx_read(const char *path, int format, int *M, int *K, int *nnz, int **ptr, int **idx,
                     float **val);
void sextans_host_free(void *p);

/* Loader throughput.  sextans_mtx_read parses with all host cores (env
 * SEXTANS_LOADER_THREADS overrides).  The binary container holds exactly the arrays it returns:
 *   64-byte header "SXTCSR01", int32 format, M, K, nnz, int64 src_size, src_mtime_ns, 24 reserved bytes;
 *   ptr[(format == CSR ? M : K) + 1], idx[nnz], val[nnz]  (little endian).
 * sextans_mtx_read_cached reads `path` through the container `cache_path` (NULL: path + ".csr.sxbin" /
 * ".csc.sxbin"): a container written for the same format from a source file of the same size and
 * mtime is loaded instead of parsing (*cache_hit = 1); otherwise the text is parsed and the container
 * (re)written, best effort.  Results are identical either way. */
int sextans_matrix_save(const char *path, int format, int M, int K, int nnz, const int *ptr, const int *idx,
                        const float *val);
int sextans_matrix_load(const char *path, int *format, int *M, int *K, int *nnz, int **ptr, int **idx,
                        float **val);
int sextans_mtx_read_cached(const char *path, const char *cache_path, int format, int *M, int *K, int *nnz,
                            int **ptr, int **idx, float **val, int *cache_hit);

/* Matrix-Market `real general` writer (tools: holdout matrices written to disk for the CLI); %.9g keeps every fp32 value. */
int sextans_mtx_write(const char *path, int M, int K, const int *row_ptr, const int *col_idx, const float *val);

/* Replaces CSC_2_CSR (sparse_helper.h:475-509).  Caller provides row_ptr[M+1], col_idx[nnz],
 * csr_val[nnz].  Per-row column order = CSC traversal order (ascending columns). */
int sextans_csc_to_csr(int M, int K, int nnz, const int *col_ptr, const int *row_idx,
                       const float *csc_val, int *row_ptr, int *col_idx, float *csr_val);

/* Replaces the dense-operand initialisation of sextans-host.cpp:94-111:
 * B[k + K*n] = 1.0f;  C[m + M*n] = (float)(1.0*(m+1)*(n+1)/M/N). */
void sextans_init_dense_B(int K, int N, float *B);
void sextans_init_dense_C(int M, int N, float *C);

/* tapa::round_up<8>(N), sextans-host.cpp:51. */
int sextans_round_up_n(int N);

/* Replaces the verification loop of sextans-host.cpp:262-289.  Returns the mismatch count
 * (|a-b| / (min(|a|,|b|) + 1e-4) > 1e-4) and writes the percentage; pass iff *percent < 2.0. */
int sextans_verify(int M, int N, const float *c_cpu, const float *c_dev, float *percent);

/* Throughput formula of sextans-host.cpp:219,255-260: 2*N*(nnz+M)/1e9/seconds. */
double sextans_gflops(int M, int N, int64_t nnz, double seconds);

/* Host golden used ONLY by the CLI's built-in self check (the reference's main() runs
 * cpu_spmm_CSR for the same purpose, sextans-host.cpp:206-219).  It is never a fallback for
 * the device path. */
int sextans_selfcheck_golden(int M, int N, int K, float alpha, const int *row_ptr,
                             const int *col_idx, const float *val, const float *B, float beta,
                             float *C_inout);

/* ------------------------------------------------------------------ packed row-bucketed form of A
 *
 * The engine's counterpart of the reference's packed non-zero stream (generate_edge_list_for_all_PEs
 * + edge_list_64bit, sparse_helper.h:292-473; 64-bit words with a window-local 14-bit column): rows
 * bucketed into blocks of <= 256/lanes_per_row rows whose distinct columns fit the 36 KiB LDS panel;
 * per block an ascending dictionary; per non-zero a 16-bit dictionary index (or the 32-bit column in
 * blocks without reuse) + fp32 value, every row starting on a 4-entry boundary.  sextans_set_matrix_*
 * builds the same arrays internally; these entry points expose them (inspection, tests, reuse). */
typedef struct sextans_packed {
    int M, K;
    int64_t nnz;
    int lanes_per_row;            /* 2, 4 or 8 (N tile = 4 * lanes) */
    int nblk;                     /* row blocks */
    int *blk_row;                 /* nblk + 1: block b owns rows [blk_row[b], blk_row[b+1]) */
    int *dict_ptr;                /* nblk + 1 offsets into dict; empty range = "direct" block */
    int *dict;                    /* distinct columns of each dictionary block, ascending */
    int *row_off;                 /* M + 1: first stream entry of each row (multiple of 4) */
    uint16_t *idx16;              /* stream_len: dictionary index per entry (dictionary blocks); 0xFFFF in the
                                     padding of those rows, whose value is -0.0f: the kernel multiplies such an
                                     entry with a +1.0f row, and x + (-0.0f) == x bit for bit */
    int *col32;                   /* stream_len: column per entry (direct blocks) */
    float *val;                   /* stream_len: value per entry (+-0 in padding) */
    int64_t stream_len;
    int max_dict;                 /* largest dictionary */
    int64_t nnz_in_panel_blocks;  /* non-zeros covered by dictionary blocks */
} sextans_packed;

/* min_reuse_x100: a block gets a dictionary only if nnz >= min_reuse_x100/100 * distinct columns. */
int sextans_pack_csr(int M, int K, const int *row_ptr, const int *col_idx, const float *val,
                     int lanes_per_row, int min_reuse_x100, sextans_packed *out);
void sextans_packed_free(sextans_packed *p);
/* Decoder: reconstructs col_idx[nnz], val[nnz] of the CSR matrix with row extents row_ptr. */
int sextans_unpack_csr(const sextans_packed *p, const int *row_ptr, int *col_idx, float *val);

/* K-windowed stream of A for the accumulator-resident kernel ("kernel" = 3; the reference's own dataflow:
 * B window on chip, partial sums resident, non-zeros pre-bucketed per (PE, window) -- sextans.cpp:337,353-381,
 * 462-570; sparse_helper.h:345-403).  Wavefront g owns rows [g*rows_per_wave, (g+1)*rows_per_wave); its
 * stream is steps [wave_step0[g], wave_step0[g+1]) of 32 entries each (a multiple of 24 steps), ordered by
 * (K window, rank of the entry inside its (row, window), row) so that a row's entries keep their CSR order;
 * no step holds a row twice; padding entries carry the local row `rows_per_wave` (a dummy accumulator).
 * Entry (little endian 64 bit): low word = fp32 value bits, high word = local_row << 23 | column.
 * K must be <= 2^23, rows_per_wave <= 510.  The arrays are what sextans_set_matrix_* builds internally. */
typedef struct sextans_window_packed {
    int M, K;
    int64_t nnz;
    int rows_per_wave, window_cols, nwaves;
    int64_t steps;                /* total steps (stream holds steps * 32 entries + a zero tail of 32 steps) */
    int64_t padded_lower_bound;   /* the dispatcher's cheap estimate of steps * 32 */
    int *wave_step0;              /* nwaves + 1 */
    uint64_t *stream;
} sextans_window_packed;
int sextans_window_pack_csr(int M, int K, const int *row_ptr, const int *col_idx, const float *val,
                            int rows_per_wave, int window_cols, sextans_window_packed *out);
void sextans_window_packed_free(sextans_window_packed *p);

/* ------------------------------------------------------------------ the accelerator's own buffer formats
 *
 * Writer / reader for exactly the buffers the reference host prepares for tapa::invoke(Sextans, ...)
 * (sextans-host.cpp:114-204), so inputs prepared for the FPGA can be consumed by this engine
 * (sextans_invoke below) and results produced in the FPGA's C layout.
 *
 *   A: scheduled non-zero stream of 64 PEs (row % 64) in 4096-column windows, 10-slot spacing between
 *      entries of one row inside a window, bubble padding to the window's longest PE list
 *      (generate_edge_list_for_all_PEs, sparse_helper.h:292-403); packed as 64-bit words
 *      col14<<50 | row18<<32 | fp32 in 8 channels, PE p at channel p%8, slot bitrev3(p/8) of each
 *      8-word group (edge_list_64bit, sparse_helper.h:406-473); bubble = 0x3FFFF<<32 (any word with
 *      row bit 17 set is skipped, sextans.cpp:407).  edge_list_ptr[w+1] = stream length after window w.
 *   B: NUM_CH_B (4 in sextans.h:8, or 8) float channels, sextans-host.cpp:152-177.
 *   C: 8 float channels, row m in channel m%8 at colsize*(n/8) + (m/8)*8 + n%8, colsize = round_up(M,16)
 *      (sextans-host.cpp:179-195 for C_in, :264-270 for C_out). */
#define SEXTANS_EDGES_NUM_PE 64
#define SEXTANS_EDGES_NUM_CH 8
#define SEXTANS_EDGES_WINDOW 4096

typedef struct sextans_edges {
    int32_t M, K;
    int32_t num_windows;          /* NUM_ITE   = ceil(K / 4096)               sextans-host.cpp:221 */
    int32_t num_a_len;            /* NUM_A_LEN = edge_list_ptr[num_windows]   sextans-host.cpp:222 */
    int64_t nnz;                  /* non-bubble words */
    int64_t ptr_len;              /* ints in edge_list_ptr, zero padded to a multiple of 1024 (:131-134) */
    int64_t chan_len;             /* words per channel: round_up(8 * num_a_len, 512) (sparse_helper.h:412-413) */
    int32_t *edge_list_ptr;
    uint64_t *channel[SEXTANS_EDGES_NUM_CH];
} sextans_edges;

/* Writer: CSC (what the reference host feeds its scheduler) -> stream.  M must be <= 64 * 2^17. */
int sextans_edges_pack_csc(int M, int K, int nnz, const int *col_ptr, const int *row_idx, const float *val,
                           sextans_edges *out);
void sextans_edges_free(sextans_edges *e);
/* Reader: stream -> CSR with each row's entries in stream order (the order the accelerator accumulates
 * them in; ascending columns for streams written from a CSC matrix).  Arrays are malloc'ed
 * (sextans_host_free).  SEXTANS_ERR_INDEX if a word addresses a row >= M or a column >= K. */
int sextans_edges_decode_csr(const int32_t *edge_list_ptr, const uint64_t *const *channel, int num_windows,
                             int M, int K, int64_t *nnz, int **row_ptr, int **col_idx, float **val);
/* Container file (little endian): "SXTEDGE1", M, K, num_windows, num_a_len (int32), nnz, ptr_len,
 * chan_len (int64), edge_list_ptr[ptr_len], channel[8][chan_len]. */
int sextans_edges_save(const char *path, const sextans_edges *e);
int sextans_edges_load(const char *path, sextans_edges *out);

/* Dense channel layouts (host side).  N must be a multiple of 8; num_ch_b is 4 or 8.  *_len = floats
 * per channel (padded to 1024 as the reference allocates them); pack writes only the addressed
 * positions (callers zero the buffers first, as the reference does). */
int64_t sextans_chan_b_colsize(int K, int num_ch_b);
int64_t sextans_chan_b_len(int K, int N, int num_ch_b);
int64_t sextans_chan_c_colsize(int M);
int64_t sextans_chan_c_len(int M, int N);
int sextans_chan_pack_b(int K, int N, int num_ch_b, const float *B, float *const *channel);
int sextans_chan_unpack_b(int K, int N, int num_ch_b, const float *const *channel, float *B);
int sextans_chan_pack_c(int M, int N, const float *C, float *const *channel);
int sextans_chan_unpack_c(int M, int N, const float *const *channel, float *C);

/* ------------------------------------------------------------------ L1: the SpMM engine (HIP) */

typedef struct sextans_engine *sextans_handle_t;

/* Number of usable gfx950 devices (0 and SEXTANS_ERR_NO_DEVICE when there is none). */
int sextans_device_count(int *count);

/* Engine bound to one HIP device.  Re-entrant per handle; a handle must not be used from two
 * threads at once (the reference is single-threaded). */
int sextans_create(sextans_handle_t *h, int device);
int sextans_destroy(sextans_handle_t h);

/* Tunables.  key: "kernel" (0 auto, 1 row-group gather, 2 LDS panel, 3 K-windowed accumulator-resident
 * sweep), "window_rows" (rows per wavefront of kernel 3, default 319), "window_cols" (columns per K window,
 * default 65536), "window_unroll" (4 or 8 steps in flight), "window_auto" (1: kernel 0 may choose kernel 3 from
 * its fabric-traffic model; default 0 because the sweep never beat the gather kernel on MI355X),
 * "lanes_per_row" (2/4/8, N-tile = 4*lanes; 0 = auto, the default: 4 for the panel kernel, 8 for the gather
 * kernel when N >= 32), "stage_a" (0/1 stage the CSR stream through LDS), "xcd_remap" (0/1), "exact" (1 = no FMA, reference rounding; 0 = allow FMA), "profile" (0/1 hipEvent
 * per-kernel timing), "phase_timing" (0/1, see sextans_phase_timing_read), "bucket_rows" / "split_rows" (long
 * rows, the load-balancing the reference gets from dealing rows to PEs by row % 64: rows longer than L0 =
 * "bucket_rows" leave the main kernel and are processed in a second launch in order of length, still summed in
 * CSR order = bit-identical; rows longer than T = "split_rows" are cut into pieces of T entries that are summed in
 * parallel and folded in order -- THOSE rows are re-associated and meet the stated 1e-4 tolerance instead of bit
 * identity; sextans_reassociated_rows lists them.  Values: > 0 explicit, 0 off, -1 chosen from the matrix:
 * L0 = max(32, 2 * mean row length), T = max(1024, nnz / 16384).  Defaults: "bucket_rows" = -1 (exact, so it costs
 * nothing in parity), "split_rows" = 0 -- EVERY row is summed in strict CSR order and the result is bit-identical to
 * cpu_spmm_CSR unless the caller opts into re-association with "split_rows" = -1 or > 0 (power-law inputs with rows of
 * 10^5 entries may want that).  "exact_chain" (default 1): in strict-order mode a row longer than max(1024, nnz/16384)
 * is summed by the exact-chain kernel -- producer wavefronts form all rounded products of the row, ONE serial chain of
 * rounded adds per output column consumes them, bit-identical to cpu_spmm_CSR; 0 = such rows stay on the piece
 * kernel (the same bits, ~20x slower for a 400 000-entry row).  Matrices without long rows take none of this path.  "global_nnz": non-zeros of the WHOLE matrix when this engine holds a row range of it (multi-GPU);
 * the automatic T is derived from it so that every rank cuts hub rows exactly as one GPU would; sextans_dist_spmm sets
 * it from an exchanged sum; 0 = this engine's matrix is the whole matrix),
 * "fuse_b" (1 = the panel kernel may stage B straight from column-major B when B is <= 16 MiB and every row block has a dictionary, saving the
 * repack launch; default 1), "panel_min_reuse_x100" (a row block uses the LDS panel when
 * nnz >= value/100 * distinct columns; default 200; decides for N <= 16) and "panel_min_reuse_wide_x100" (the same for
 * N >= 32, default 150: with more columns per B row the panel pays earlier; the packed plan is built once, for the lower
 * of the two), "panel_v2" (-1 auto / 0 / 1: the register-resident form of the panel
 * kernel, spmm_csr_panel_v2), "tiles_per_wg" (N tiles one workgroup of that kernel walks; 0 = auto),
 * "row_cluster" (-1 auto / 0 never / 1 whenever a clustered plan can be built / 2 graph clustering also for grids): the plan of
 *   spmm_csr_panel_v2 may visit the rows in a clustered order on whole-matrix calls -- the rows of a matrix are independent, so
 *   every sum keeps its order and the result stays bit-identical to cpu_spmm_CSR.  Two forms: (1) matrices with Cartesian-grid
 *   stencil structure in natural ordering are visited brick by brick (csrc/row_cluster.hip; stats "grid_stride_line",
 *   "grid_stride_plane"); (2) matrices whose numbering has no locality but whose graph has (meshes in an arbitrary node order) are
 *   aggregated over the matrix graph on the device (csrc/graph_cluster.hip), their columns relabelled in first-touch order, and the
 *   SpMM runs in its REORDERED form: B repacked into permuted panels, C staged block-major (two extra passes over C inside the call;
 *   sextans_last_kernel = "spmm_csr_panel_v2_reordered").  Any M x K since round 5 ("row_similarity"); rows on the long-row paths (pieces, exact chains) are fine.  Stats: "row_cluster"
 *   (1 grid bricks / 2 graph clustering in use, -1 declined, 0 not evaluated yet -- it is evaluated by the first whole-matrix
 *   SpMM with N >= 16), "panel_rows_natural", "panel_rows_clustered" (B rows copied into LDS per 16-column tile), "panel_blocks",
 *   "panel_blocks_clustered", "cluster_shared_fraction" (sampled pre-test of form 2).  The same reason the reference schedules its
 *   non-zeros: keeping the on-chip B window hot (sparse_helper.h:345-403).
 *   Tunables of the two forms (defaults are the measured best; every setting is bit-identical): "small_panel" (1: clustered plans
 *   of short-row matrices are packed for a 320-row panel when every dictionary fits), "row_sets" (2: short-row 3-D grid matrices --
 *   every row <= 32 entries -- use 128-row bricks as two 64-slot row sets on one panel; 3 = 2-D grids too; 1 = never; stat
 *   "row_sets"), "refine_sweeps" (8) / "refine_rows" (62): block refinement of the graph-clustered order, "relabel_columns" (1),
 *   "cluster_top" (depth of the merge tree).
 * "run_cluster" (default 0 = off; 1 = when it copies >= 10 % fewer panel rows; 2 = always): run-level clustering -- matrices in a
 *   numbering WITH locality whose natural row blocks are cut short by the panel capacity and whose graph-clustered plan is not worth
 *   the reordered form's passes ("cluster_decline" 12) keep runs of 16 consecutive rows together and build each row block from up to 4
 *   runs chosen over the graph of runs (no staging, natural B panels; stat "cluster_runs").  Built and measured in round 5: 11.6 % fewer
 *   panel rows on the holdout class, kernel 687 -> 706 us (the C accesses become 64-byte runs, neighbouring blocks stop sharing B lines
 *   in L2): OFF by default, kept as a tested negative result.
 * "row_similarity" (-1 auto / 0 never / 1 always): which graph form (2) clusters the rows over.  A square matrix with a symmetric
 *   pattern is its own graph (column c = row c).  RECTANGULAR matrices have no such reading: their rows are joined to the 16 rows that share the most columns with them
 *   (found through the transposed pattern, csrc/graph_cluster.hip: row_similarity_graph_device) -- the reference schedules any
 *   M x K matrix for its on-chip window too (sparse_helper.h:345-403).  A square matrix with an unsymmetric pattern (sampled: stat
 *   "pattern_symmetry" < 0.98) is clustered over A + A^T -- the pairwise matching needs symmetric weights.  Stat
 *   "cluster_graph_kind": 0 the matrix itself, 1 a row slab's own square pattern ("row_offset"), 2 row similarity, 3 A + A^T.
 * "row_offset" (default -1): the matrix of this engine is the row slab [row_offset, row_offset + M) of a K x K matrix (what a rank of
 *   sextans_dist_spmm holds, which sets it): the graph clustering of "row_cluster" then runs on the slab's own square pattern.
 * "share_index" (default 1): consecutive rows of a block whose 16-bit index lists are equal up to a constant shift keep one copy
 *   of the list (stats "index_stream_entries", "value_stream_entries"); the exported plan carries every row's own list.
 * "colwise_max_len" (default 6; "kernel" = 4 forces it): rows of at most this mean length in a numbering with locality (stat
 *   "row_coherence" >= 0.7) run on the lane-per-row kernel over the caller's column-major operands (no B repack; stat "colwise").
 * "split_mixed" (default 1; 0 = one launch of spmm_csr_panel<MIXED>): a MIXED plan -- some row blocks reuse B rows and have a dictionary,
 *   others (rows with uniformly random columns) do not -- runs in two launches: the blocks with a dictionary on spmm_csr_panel_v2, the
 *   rows of the others on the gather kernel; both walk compact lists.  Whole-matrix calls, column-major and row-major operands (the
 *   latter without copies).  Whole-matrix calls take the split form from 15 % of the non-zeros in blocks with reuse on (the gather
 *   kernel alone below that; every other use of the LDS-panel plan keeps its 50 % threshold).  Stat "mixed_plan": 0 no / 1 mixed, one
 *   launch / 2 mixed, split form.  Bit-identical either way.
 * "colwise_tiles_adjacent" (default 1; 0 = the tile as the slow grid axis, 2 = neighbours at every N): where the lane-per-row kernel puts
 *   the 16-column tiles of a row at N >= 32.  Row-major operands: groups of T lanes per row take neighbouring tiles, so a wavefront's
 *   loads cover whole 128-byte lines (0.25 -> 0.51 of the roofline at N = 32 .. 256); column-major: the two tiles of N = 32 are
 *   neighbours in the launch order (the row block's CSR entries come from HBM once).  Bit-identical either way.
 * "cluster_group" (default 6), "cluster_shape" (0 = default bricks): layout tunables of form (1); measurement switches (below).
 * "small_v2" (default 1: small matrices staged from column-major B size the launch's dictionary capacity and register-resident
 * batches from the plan; 0 = the full-capacity form, for measurements),
 * "cols_per_lane" (0/4 = 16-column tiles at 4 workgroups per CU, the default; 8 = 32-column super tiles at 2 per CU),
 * "bell_shared" (blocked-ELL, N = 256: -1 = use the union-walk kernel spmm_bell_mfma_shared when 8 consecutive block rows
 * share block columns, stat "bell_share" >= 1.5; 0 never; 1 whenever the unions fit), "bell_debug" (measurements only:
 * ablation bits of that kernel, results are wrong when non-zero).  Unknown keys -> SEXTANS_ERR_INVALID.
 * MEASUREMENT SWITCHES -- "bell_debug", "cluster_shape", "cluster_group", "phase_timing", "reordered_xcd", "dist_broadcast_runs", "rowblock_tiles" -- are not part of the drop-in surface:
 * setting one to anything but its default returns SEXTANS_ERR_INVALID unless the process runs with SEXTANS_DEBUG_OPTIONS=1 in
 * its environment (tools/ do; "bell_debug" corrupts C on purpose). */
/* "mfma_dense_tiles" / "dense_tile_fill_x100": north_star's "MFMA only where a tile is actually dense".  The engine
 * always counts the 32x32 tiles of A whose fill reaches dense_tile_fill_x100 % (default 50) -- sextans_get_stat
 * "dense_tiles", "dense_tile_fraction" (share of the non-zeros in such tiles; estimated from a sample of up to 512
 * block rows while mfma_dense_tiles = 0, exact once the tiles are extracted).  With mfma_dense_tiles = 1 the caller
 * opts into bf16 for them: those tiles (values rounded to bf16) times bf16(B) run on v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation, the rest of A stays on the fp32 CSR kernels, whose epilogue adds the two parts; results then
 * meet the blocked-ELL tolerance (tests/test_dense_tiles_gpu.py), not bit identity.  Needs N % 32 == 0 and
 * whole-matrix calls.  Default 0: fp32 everywhere, bit-identical to cpu_spmm_CSR.
 * mfma_dense_tiles = 2 (round 6): NO precision trade -- dense blocks of 16 consecutive rows run on the FP32 matrix cores
 * (v_mfma_f32_16x16x4_f32, csrc/rowblock_mfma_kernel.h; the reference's PEs multiply and accumulate in fp32 too, sextans.cpp:285-295,
 * 425-446).  A block is routed when its rows are strictly ascending in their columns and its fill = entries / (64 x the groups of 4
 * consecutive columns it touches) reaches dense_tile_fill_x100 %; routed rows are summed ONLY there, walked in ascending column order,
 * and since the instruction is a k-ordered chain of fused multiply-adds the result is BIT-IDENTICAL to the "exact" = 0 kernels
 * (acc = fmaf(a, b, acc), epilogue fmaf(alpha, acc, beta * c)) -- i.e. inside |d| <= 1e-4 * (|alpha| sum|a b| + |beta c|) of
 * cpu_spmm_CSR -- for finite B (a padding zero times an infinite B entry would be NaN where the CSR kernels see no entry at all).  Any N
 * (multiple of 8), whole-matrix and row-range calls, every rank of the multi-GPU forms; "dense_tiles" / "dense_tile_fraction" then
 * count the routed blocks / the share of the non-zeros in them.  MEASURED (round 6, profiles/r06_rowblock_mfma.jsonl): on a
 * block-tridiagonal matrix of fully dense 32 x 32 blocks the path runs the step at 0.75 / 0.43 / 0.35 / 0.30 of the HBM roofline at
 * N = 16 / 64 / 128 / 256 against 0.79 / 0.47 / 0.38 / 0.32 of the VALU kernels with "exact" = 0 (the matrix cores are ~40 % busy, the
 * loop waits for memory and for issue slots in equal parts, and the matrix is within 1.4 x of its memory bound anyway), and FEM
 * matrices fill their fragments to 0.34 (3 dof) - 0.52 (6 dof) only, so two to three times as many multiply-adds are issued as the
 * matrix holds: 1.5 - 2.7 x slower there.  An option for experiments, not a default.
 * Use it with "exact" = 0 so that routed and unrouted rows follow one rounding rule. */
/* ACCURACY MODES (round 6) -- one documented switch instead of three options:
 *   sextans_set_option(h, "mode", SEXTANS_MODE_STRICT)  default.  "exact" = 1, "split_rows" = 0, "mfma_dense_tiles" = 0: every product
 *       rounded, every row summed in CSR order: BIT-IDENTICAL to cpu_spmm_CSR (sparse_helper.h:262-290), stronger than the reference's
 *       own pass criterion.
 *   sextans_set_option(h, "mode", SEXTANS_MODE_FAST)    "exact" = 0 (fused multiply-adds), "split_rows" = -1 (hub rows above
 *       max(1024, nnz / 16384) entries are cut into pieces summed in parallel and folded in order); "mfma_dense_tiles" stays 0 (its
 *       fp32 form, 2, is bit-identical to this mode and may be added by hand; measured, it does not pay yet).  GUARANTEE, per output element:
 *           |C_fast - C_ref| <= 1e-4 * (|alpha| * sum_j |a_ij * b_jn| + |beta * c_in|)
 *       (the condition-aware form of north_star's "within 1e-4 relative error"; the reference's own check is a tolerance
 *       too, sextans-host.cpp:272-282).  Measured distance is ~1e-7 of that scale: fp32 roundoff of a different, but still fp32,
 *       summation.  Deterministic: the same call gives the same bits every time, on every rank count.
 * sextans_get_option(h, "mode", &v): SEXTANS_MODE_STRICT / SEXTANS_MODE_FAST when the three options stand as that mode set them,
 * -1 when they were set apart by hand. */
#define SEXTANS_MODE_STRICT 0
#define SEXTANS_MODE_FAST 1
int sextans_set_option(sextans_handle_t h, const char *key, int64_t value);
int sextans_get_option(sextans_handle_t h, const char *key, int64_t *value);
/* Rows of the current matrix whose sums are re-associated under the current "split_rows" setting (ascending);
 * writes at most `capacity` of them, *count = how many there are.  sextans_get_stat: "reassociated_rows",
 * "split_threshold". */
int sextans_reassociated_rows(sextans_handle_t h, int *rows, int capacity, int *count);
/* The packed row-bucketed form the ENGINE built for the current matrix at `lanes_per_row` (on the device since round 3:
 * the CSR arrays never leave HBM, csrc/plan_device.hip), read back in the public layout of sextans_pack_csr -- the host
 * builder of the same format; the two are byte-identical (tests/test_plan_device_gpu.py).  Free with
 * sextans_packed_free.  Analogue being replaced: generate_edge_list_for_all_PEs + edge_list_64bit on the host,
 * sparse_helper.h:345-473, sextans-host.cpp:114-148. */
int sextans_export_plan(sextans_handle_t h, int lanes_per_row, struct sextans_packed *out);
/* The order in which the clustered plan ("row_cluster") visits the rows of the current matrix: order[i] = row at position i (M ints,
 * host); *clustered (optional) = 0 none (order = identity) / 1 grid bricks / 2 graph clustering.  For callers that can renumber their
 * matrix once -- P A P^T with new_of_old[order[i]] = i, what FEM packages do with RCM: contiguous row ranges of the renumbered matrix are
 * compact pieces of the matrix graph, which is what the row-range partition of sextans_dist_spmm needs on a mesh whose numbering has
 * no locality.  The reference fixes the order of its non-zeros once on the host as well (sparse_helper.h:345-403). */
int sextans_export_row_order(sextans_handle_t h, int *order, int *clustered);
/* Read-only figures about the matrix currently set.  key: "plan_build_s" (seconds spent so far
 * building packed forms of A -- on the device for the panel plan, on the host for the window stream; outside every timed
 * region like the reference's scheduling/packing, sextans-host.cpp:114-148), "window_padded_entries",
 * "window_state" (0 not evaluated, 1 built, -1 rejected), "panel_fraction", "panel_blocks", "piece_path_rows",
 * "reassociated_rows", "bucket_threshold", "split_threshold", "dense_tiles", "dense_tile_fraction",
 * "dense_tiles_on_mfma", "row_cluster" and the other clustering figures listed with that option, "device_bytes" (bytes of
 * device memory the engine holds right now: matrix copies, packed plans, workspaces), "col_range_lo" / "col_range_hi" (the rows
 * of B the matrix has columns in: only those are repacked -- a rank of a row-partitioned SpMM over a banded matrix touches
 * 1 / world of B plus a halo), "cluster_decline" (why the graph clustering was not used: 1 not square and "row_similarity" = 0, 2 long-row paths,
 * 3 offsets, 4 natural blocks full, 5 no shared neighbourhoods, 6..9 a builder failed, 10..12 plan unusable / no gain,
 * 13 short rows in a local numbering). */
int sextans_get_stat(sextans_handle_t h, const char *key, double *value);

/* Upload a CSR matrix (host pointers) once; later spmm calls reuse the device copy.  This is
 * the analogue of the reference's FPGA-side preparation (generate_edge_list_for_all_PEs +
 * edge_list_64bit, sextans-host.cpp:114-148), which is likewise outside its timed region. */
int sextans_set_matrix_csr(sextans_handle_t h, int M, int K, int64_t nnz, const int *row_ptr,
                           const int *col_idx, const float *val);
/* Same, for arrays that already live on the engine's device (not copied, not owned). */
int sextans_set_matrix_csr_device(sextans_handle_t h, int M, int K, int64_t nnz,
                                  const int *d_row_ptr, const int *d_col_idx, const float *d_val);

/* Replaces tapa::invoke(Sextans, ...) (sextans-host.cpp:237-251, prototype sextans.h:20-26) with
 * the argument meaning of cpu_spmm_CSR (sparse_helper.h:262-272): host buffers in, C updated in
 * place, column major, ld = K for B and M for C.  The kernel runs rp_time (>=1) times, every
 * repeat from the same C input (the reference reads C_in and writes a separate C_out), and
 * *elapsed_ns receives the device time of ALL repeats (tapa::invoke returns the same;
 * sextans-host.cpp:252 divides by rp_time).  Host<->device copies are outside *elapsed_ns.  B is laid
 * out in panels once (first repeat); the repeats are replayed as one hipGraph. */
int sextans_spmm_host(sextans_handle_t h, int N, float alpha, const float *B, float beta, float *C,
                      int rp_time, double *elapsed_ns);

/* Device-resident form: all pointers are device pointers on the engine's device, column major
 * with explicit leading dimensions (ldb >= K, ldc >= M).  d_C_in and d_C_out may alias.
 * Enqueues on `stream` (a hipStream_t passed as void*) and returns without synchronising.  NULL = the CALLING THREAD's default
 * stream: the library is built with per-thread default streams (it never touches HIP's process-wide legacy stream, see
 * INTEGRATION.md section 9); that stream is a blocking stream, so it still orders itself after work the caller put on the legacy
 * stream. */
int sextans_spmm_device(sextans_handle_t h, int N, float alpha, const float *d_B, int64_t ldb,
                        float beta, const float *d_C_in, float *d_C_out, int64_t ldc, void *stream);

/* Same with separate leading dimensions for C_in and C_out (multi-GPU: a rank reads its rows of the
 * full column-major C_in, ldc_in = M, and writes a packed M_local x N slab, ldc_out = M_local). */
int sextans_spmm_device2(sextans_handle_t h, int N, float alpha, const float *d_B, int64_t ldb,
                         float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out,
                         int64_t ldc_out, void *stream);

/* ROW-MAJOR operands (round 5): B is K x N with B[k * ldb + n], C_in / C_out are M x N with C[m * ldc + n] (ld >= N; C_in and C_out may
 * alias).  The reference lays B and C out for its kernel on the host, outside the timed call (sextans-host.cpp:150-195, 264-270); the
 * column-major entry points above pay for that inside the call (B repack; two more passes over C in the reordered form).  Here the
 * caller's B IS the kernel's panel (N = 16: exactly; N > 16: tile t = columns 16 t .. at row stride ldb) and C is read and written in
 * the caller's rows, 16 bytes per lane: no layout pass on the LDS-panel paths (natural-order, grid-brick and graph-clustered plans;
 * sextans_last_kernel = "spmm_csr_panel_v2_rowmajor[_clustered]") nor on the lane-per-row kernel ("spmm_csr_colwise_rowmajor"), and a
 * graph-clustered plan is used from 25 % fewer panel rows on instead of 40 %.  Needs 16-byte aligned pointers and ld % 4 == 0 (panel
 * paths: also K * ldb < 2^32 and M * ldc < 2^30); everything else (gather kernel, rows on the long-row paths, dense tiles on MFMA,
 * unaligned operands, N = 512 on 4 M rows) goes through column-major copies in the engine's workspaces.  Same arithmetic, same order:
 * bit-identical to cpu_spmm_CSR. */
int sextans_spmm_device_rm(sextans_handle_t h, int N, float alpha, const float *d_B, int64_t ldb, float beta, const float *d_C_in,
                           int64_t ldc_in, float *d_C_out, int64_t ldc, void *stream);

/* Everything the first SpMM call for (matrix, options, N, layout) would build inside itself -- packed row-bucketed plan, clustered row
 * order, long-row tables, workspaces, side streams -- built NOW, outside any timed region or hipGraph capture (round 6).  The
 * analogue of the reference preparing its streams before tapa::invoke (sextans-host.cpp:114-204).  Optional: the compute entry
 * points prepare lazily.  `stream` is only used for the few device passes of the builders (it is synchronised). */
#define SEXTANS_LAYOUT_COLMAJOR 0
#define SEXTANS_LAYOUT_ROWMAJOR 1
int sextans_prepare(sextans_handle_t h, int N, int layout, void *stream);

/* Row-range form: computes rows [row_begin, row_end) only.  d_C_in / d_C_out address row_begin as
 * their row 0 (ldc_in, ldc_out >= row_end - row_begin).  Used to pipeline a rank's slab in chunks so the
 * all-gather of chunk i overlaps the SpMM of chunk i+1.  flags: SEXTANS_ROWS_REUSE_B_PANELS = the B
 * panels repacked by the previous call on this handle are still valid (same B, same N): skip the repack.
 * A range that starts and ends on sextans_align_row boundaries (or at 0 / M) keeps the whole-matrix kernel;
 * any other range uses the row-group gather kernel. */
#define SEXTANS_ROWS_REUSE_B_PANELS 1
int sextans_spmm_device_rows(sextans_handle_t h, int N, float alpha, const float *d_B, int64_t ldb,
                             float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out,
                             int64_t ldc_out, int row_begin, int row_end, int flags, void *stream);

/* Largest row <= `row` at which a row-range call of an N-column SpMM on this matrix can be cut without losing
 * the kernel the dispatcher would use for the whole matrix (a wavefront boundary of the window kernel, a row
 * block boundary of the LDS-panel kernel; any row for the gather kernel).  row == M is returned unchanged.
 * Builds the packed forms of A if they do not exist yet. */
int sextans_align_row(sextans_handle_t h, int N, int row, int *aligned);

/* Drop-in for the kernel-invoke call itself: the argument list of
 *   tapa::invoke(Sextans, bitstream, edge_list_ptr, edge_list_ch[8], mat_B_ch[NUM_CH_B], mat_C_ch_in[8],
 *                mat_C_ch[8], NUM_ITE, NUM_A_LEN, M, K, P_N, alpha_u, beta_u)
 * (sextans-host.cpp:237-251, prototype sextans.h:20-26) on the host buffers the reference host prepared.
 * P_N = (rp_time << 16) | N, alpha_u / beta_u = fp32 bit patterns.  The stream is decoded and uploaded
 * (edge_list_ptr == NULL: keep the matrix set by the previous sextans_set_matrix_edges / sextans_invoke on
 * this handle, which must have the same M and K); B and C_in channels are uploaded and converted on the
 * device; C_out channels receive colsize * N/8 floats each, padded rows included (alpha*0 + beta*0, as the
 * accelerator writes them).  *elapsed_ns = device time of the layout conversions + all rp_time repeats.
 * Results are bit-identical to cpu_spmm_CSR on the decoded matrix. */
int sextans_set_matrix_edges(sextans_handle_t h, const int32_t *edge_list_ptr, const uint64_t *const *edge_list_ch,
                             int NUM_ITE, int NUM_A_LEN, int M, int K);
int sextans_invoke(sextans_handle_t h, const int32_t *edge_list_ptr, const uint64_t *const *edge_list_ch,
                   const float *const *mat_B_ch, int num_ch_b, const float *const *mat_C_ch_in,
                   float *const *mat_C_ch, int NUM_ITE, int NUM_A_LEN, int M, int K, int P_N, int alpha_u,
                   int beta_u, double *elapsed_ns);

/* ---- Multi-GPU form behind the C ABI (north_star: A row-range partitioned across the GPUs of one node, B
 * replicated, RCCL all-gather of C panels over xGMI).  The reference is single-device; its only
 * sharding is rows -> PEs with B broadcast (sparse_helper.h:370, sextans.cpp:916-927), the same independence of
 * output rows used here.  One process (or thread) per GPU; RCCL is bound at run time (librccl.so.1, or the
 * library named by SEXTANS_RCCL_PATH), so single-GPU callers never need it.
 *
 *   sextans_dist_unique_id   rank 0 creates the 128-byte RCCL id and hands it to the other ranks by its own
 *                            means (MPI_Bcast, a file, a socket);
 *   sextans_dist_comm_init   every rank: communicator of `world` ranks on HIP device `device`;
 *   sextans_dist_spmm        the engine holds THIS rank's rows [row_ranges[2*rank], row_ranges[2*rank+1]) of the
 *                            M_total x K matrix (ranges tile [0, M_total) in rank order, unequal lengths allowed:
 *                            nnz-balanced splits); d_B is the full K x N matrix, d_C_in / d_C_out the full
 *                            column-major M_total x N matrices on this rank's device.  The rank's slab is
 *                            computed in `nchunks` row chunks (cut at sextans_align_row boundaries, exchanged between
 *                            the ranks once per partition) into a staging buffer; the all-gather of chunk i
 *                            (ncclAllGather on the engine's communication stream) overlaps the SpMM of chunk
 *                            i+1; a final pass writes every rank's rows into d_C_out.  Enqueued on `stream`,
 *                            returns without synchronising (host syncs only the first time a partition is used:
 *                            cut exchange and row tables).  d_C_out holds C = alpha*A*B + beta*C_in for ALL rows on
 *                            every rank.
 *                            CLUSTERED-ORDER CHUNKS (round 5): a rank whose slab runs on a graph-clustered plan ("row_cluster":
 *                            a renumbered mesh) keeps the reordered form with nchunks > 1 -- a chunk is then a range of the plan's
 *                            row blocks, its packed slab holds the rows in clustered order, and every rank scatters the received
 *                            slabs through the senders' position -> row tables (exchanged once per partition) into a row-major
 *                            staging copy of C that one streaming pass turns into column-major d_C_out.  Used when every rank of
 *                            the partition can (one flag per rank is exchanged); N % 16 == 0.
 *                            comm == NULL with world == 1: the same pipeline without collectives (single-GPU callers without RCCL). */
/* Contiguous row ranges with (nearly) equal non-zero counts for `world` ranks: ranges[2g], ranges[2g+1] = rows of rank g
 * (split points by binary search in row_ptr; host function, no device needed). */
int sextans_partition_rows_by_nnz(int M, const int *row_ptr, int world, int *ranges);
/* Which library the collectives come from (round 6).  By default the first of $SEXTANS_RCCL_PATH, librccl.so.1, /opt/rocm/lib/librccl.so.1,
 * librccl.so that dlopen finds, bound at the first use.  sextans_dist_bind_library(path) binds `path` instead (NULL or "": back to the
 * default search) -- a site's own RCCL build, or the loopback communicator of this repository's tests (tests/fake_rccl.cpp: ranks as
 * threads of one process on one device, so that world > 1 runs on a single-GPU box).  The library must export ncclGetUniqueId,
 * ncclCommInitRank, ncclCommDestroy, ncclAllGather (and ncclBroadcast, ncclGroupStart, ncclGroupEnd for row-major slabs of unequal
 * length).  SEXTANS_ERR_STATE while communicators created through sextans_dist_comm_init are alive. */
int sextans_dist_bind_library(const char *path);
/* Everything a sextans_dist_spmm* call would otherwise do INSIDE ITS FIRST INVOCATION for a partition -- exchange of the ranks' non-zero
 * counts (long-row thresholds follow the whole matrix), "row_offset", the plan build of this rank's slab (0.3-0.7 s for 3e8 non-zeros),
 * the cut-list / clustered-order flag / position -> row table exchanges of the chunk pipeline, workspaces, streams and events -- as ONE
 * collective call outside any timed region.  `form` selects the entry point it prepares for:
 *   SEXTANS_DIST_CSR_COLMAJOR  sextans_dist_spmm      (nchunks as in that call)
 *   SEXTANS_DIST_CSR_ROWMAJOR  sextans_dist_spmm_rm   (nchunks ignored; ldc of the later call assumed == N unless nchunks < 0: packed copy)
 *   SEXTANS_DIST_BELL          sextans_dist_spmm_bell (nchunks ignored)
 * Every rank of the communicator must call it (it is a collective).  It ends with an exchange of the ranks' status: it returns
 * SEXTANS_OK on ALL ranks or an error on ALL ranks -- a rank whose own work failed gets its own code, the others SEXTANS_ERR_PEER --
 * and no rank is left waiting in a collective for a rank that gave up.  After it, the dist call for the same (ranges, N, nchunks)
 * performs no host synchronisation and no control collective (stat "dist_setup_exchanges" stays where it was).  Optional: the dist
 * calls still prepare lazily when it was not called.  comm == NULL with world == 1: the local part alone. */
#define SEXTANS_DIST_CSR_COLMAJOR 0
#define SEXTANS_DIST_CSR_ROWMAJOR 1
#define SEXTANS_DIST_BELL 2
int sextans_dist_prepare(sextans_handle_t h, void *comm, int world, int rank, const int *row_ranges, int N, int nchunks, int form, void *stream);
int sextans_dist_unique_id(char id[128]);
int sextans_dist_comm_init(void **comm, int device, int world, int rank, const char id[128]);
int sextans_dist_comm_destroy(void *comm);
int sextans_dist_spmm(sextans_handle_t h, void *comm, int world, int rank, const int *row_ranges, int N, float alpha,
                      const float *d_B, int64_t ldb, float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out,
                      int64_t ldc, int nchunks, void *stream);
/* The same for ROW-major operands (round 5; d_B K x N with ldb >= N, d_C_in / d_C_out M_total x N with ldc_in / ldc >= N, all of it on
 * every rank): the rank's rows are computed by sextans_spmm_device_rm straight into their place in d_C_out -- rows [r0, r1) of a
 * row-major matrix with ldc == N are ONE contiguous run -- and exchanged by an in-place all-gather on `stream` (ncclAllGather for
 * ranges of equal length, a group of ncclBroadcast otherwise): no repack of the replicated B on every rank, no staging buffer, no
 * unpack pass, and every plan kind (natural, bricks, graph-clustered) keeps its whole-slab kernel.  ldc > N: through a packed copy.
 * C_in is read in this rank's rows only.  comm == NULL with world == 1: the SpMM alone. */
int sextans_dist_spmm_rm(sextans_handle_t h, void *comm, int world, int rank, const int *row_ranges, int N, float alpha, const float *d_B,
                         int64_t ldb, float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out, int64_t ldc, void *stream);

/* One-shot convenience with exactly cpu_spmm_CSR's argument list (sparse_helper.h:262-272):
 * create + upload + run + download + destroy on device 0. */
int sextans_spmm_csr(int M, int N, int K, int NNZ, float ALPHA, const int *CSRRowPtr,
                     const int *CSRColIndex, const float *CSRVal, const float *mat_B, float BETA,
                     float *mat_C);

/* ---- Blocked-ELL bf16 path (BASELINE config 5; no counterpart in the reference, whose PEs are scalar
 * fp32 MACs): A is M x K in dense 32x32 bf16 blocks, `ell_width` block slots per block row;
 * block_col[br*ell_width + s] is the block column of slot s (or -1 = empty slot), block_val holds the
 * slots' 32x32 bf16 values row-major (1024 values per slot).  B is bf16 COLUMN MAJOR K x N (ldb % 8 == 0),
 * C fp32 column major, C = alpha*A*B + beta*C with fp32 accumulation on v_mfma_f32_32x32x16_bf16.
 * M, K, N must be multiples of 32.  bf16 values are passed as their 16-bit patterns (uint16_t). */
int sextans_set_matrix_bell(sextans_handle_t h, int M, int K, int ell_width, const int *block_col,
                            const uint16_t *block_val);
int sextans_set_matrix_bell_device(sextans_handle_t h, int M, int K, int ell_width,
                                   const int *d_block_col, const uint16_t *d_block_val);
int sextans_spmm_bell_device(sextans_handle_t h, int N, float alpha, const uint16_t *d_B, int64_t ldb,
                             float beta, const float *d_C_in, float *d_C_out, int64_t ldc, void *stream);
/* The same with separate leading dimensions for C_in and C_out (a rank of sextans_dist_spmm_bell reads its rows inside the whole C_in
 * and writes a packed slab). */
int sextans_spmm_bell_device2(sextans_handle_t h, int N, float alpha, const uint16_t *d_B, int64_t ldb, float beta, const float *d_C_in,
                              int64_t ldc_in, float *d_C_out, int64_t ldc, void *stream);
/* Blocked-ELL over the GPUs of one node (block-row ranges): the engine holds the block rows of
 * [row_ranges[2*rank], row_ranges[2*rank+1]) -- multiples of 32, tiling [0, M_total) in rank order -- d_B is the whole bf16 K x N matrix,
 * d_C_in / d_C_out the whole column-major fp32 M_total x N matrices on this rank's device.  The rank's slab is written packed into a
 * staging buffer, one ncclAllGather on `stream` moves all slabs, one pass writes d_C_out: complete on every rank, stream-ordered,
 * bit-identical to sextans_spmm_bell_device on the whole matrix.  comm == NULL with world == 1: no collective. */
int sextans_dist_spmm_bell(sextans_handle_t h, void *comm, int world, int rank, const int *row_ranges, int N, float alpha, const uint16_t *d_B,
                           int64_t ldb, float beta, const float *d_C_in, int64_t ldc_in, float *d_C_out, int64_t ldc, void *stream);

/* Per-kernel device timing collected while option "profile" = 1: mean duration in ns of the
 * dominant SpMM kernel launches since the last reset, and how many were timed. */
int sextans_profile_read(sextans_handle_t h, double *mean_kernel_ns, int64_t *launches,
                         double *mean_repack_ns);
/* Mean duration of what a call launches BEHIND its SpMM kernel (the reordered form's C staging -> C pass); 0 launches otherwise. */
int sextans_profile_read_post(sextans_handle_t h, double *mean_post_ns, int64_t *launches);
int sextans_profile_reset(sextans_handle_t h);

/* Debug aid (option "phase_timing" = 1): wave cycles of the panel kernel's phases summed over a 1/128
 * sample of workgroups: {meta+extents+first entries, dictionary->B rows->LDS, row streaming, C tile
 * + epilogue, sampled waves, 0, 0, 0} since the option was set. */
int sextans_phase_timing_read(sextans_handle_t h, int64_t out[8]);

/* Name of the kernel the dispatcher last used (static string), for logs and rocprof matching. */
const char *sextans_last_kernel(sextans_handle_t h);

/* ------------------------------------------------------------------ synthetic inputs (bench) */

/* Deterministic counter-based synthetic CSR (BASELINE config 4): row lengths Poisson(mean_nnz)
 * from an integer inverse-CDF table, columns uniform without replacement and sorted (bandwidth 0:
 * over [0,K); bandwidth bw > 0: over the band [row-bw, row+bw], a FEM/SuiteSparse-like matrix with
 * locality), values U(-1,1).  Host and device generators produce identical bits for the same (seed,row).
 * Host form generates rows [r0,r1) only (row_ptr has r1-r0+1 entries starting at 0). */
int sextans_gen_csr_host(int M, int K, double mean_nnz, int bandwidth, uint64_t seed, int r0, int r1,
                         int **row_ptr, int **col_idx, float **val, int64_t *nnz);
/* Device form: allocates device arrays on `device` (free with sextans_device_free). */
int sextans_gen_csr_device(int device, int M, int K, double mean_nnz, int bandwidth, uint64_t seed, int r0,
                           int r1, int **d_row_ptr, int **d_col_idx, float **d_val, int64_t *nnz);
/* 3-D finite-element-like matrix: 27-point node stencil on an nx*ny*nz grid with `dof` unknowns per
 * node (dense dof x dof coupling blocks), M = K = nx*ny*nz*dof, values U(-1,1).  Stand-in for
 * SuiteSparse FEM inputs that cannot be downloaded here (Boeing/pcrystk02 = 13965 rows, 3 dof,
 * ~69 nnz/row is reproduced by 19 x 15 x 16 x 3 + trimming). */
int sextans_gen_fem3d_host(int nx, int ny, int nz, int dof, uint64_t seed, int r0, int r1, int **row_ptr,
                           int **col_idx, float **val, int64_t *nnz);
int sextans_gen_fem3d_device(int device, int nx, int ny, int nz, int dof, uint64_t seed, int r0, int r1,
                             int **d_row_ptr, int **d_col_idx, float **d_val, int64_t *nnz);
/* Power-law matrix (the sweep harness's load-balancing case): row lengths with
 * P(len >= x) = (xmin / x)^(tail_x100 / 100) on [xmin, min(max_len, K)] -- a few hub rows hundreds of thousands
 * of entries long next to a mass of short rows -- columns one uniform draw per equal stratum of [0, K) (distinct,
 * ascending), values U(-1,1).  Same bits on host and device. */
/* 2-D stencil (points = 5 or 9) on an nx x ny grid with dof unknowns per node; KKT / arrow block structure
 * [[H, A^T, U], [A, 0, U], [V, V, D]] with n variables (pentadiagonal H), n/2 constraints of 3 variables each and `arrow`
 * border rows/columns (border rows hold every 16th column: a few very long rows).  Same bits on host and device. */
int sextans_gen_stencil2d_host(int nx, int ny, int points, int dof, uint64_t seed, int r0, int r1, int **row_ptr, int **col_idx,
                               float **val, int64_t *nnz);
int sextans_gen_stencil2d_device(int device, int nx, int ny, int points, int dof, uint64_t seed, int r0, int r1, int **d_row_ptr,
                                 int **d_col_idx, float **d_val, int64_t *nnz);
/* P A P^T of a device-resident CSR matrix (measurement infrastructure for meshes in arbitrary node orders): row / column i becomes
 * row / column new_of_old[i] (host array, a permutation of 0 .. M-1), columns ascending per row, values travel with their entries.
 * New device arrays (sextans_device_free).  Rows of up to 4096 entries. */
int sextans_csr_permute_symmetric_device(int device, int M, int64_t nnz, const int *d_row_ptr, const int *d_col_idx, const float *d_val,
                                         const int *new_of_old, int **o_row_ptr, int **o_col_idx, float **o_val);
/* Rows [r0, r1) of a device-resident CSR matrix as a matrix of their own (what a rank of the row-partitioned SpMM holds; measurement
 * infrastructure): a new row_ptr of r1 - r0 + 1 entries starting at 0 (sextans_device_free), *first_entry = where the slab's
 * col_idx / val begin inside the original arrays (use d_col_idx + *first_entry, d_val + *first_entry), *nnz = its non-zeros. */
int sextans_csr_slice_rows_device(int device, int r0, int r1, const int *d_row_ptr, int **o_row_ptr, int64_t *first_entry, int64_t *nnz);
/* HOLDOUT class (round 5): kron(T_n, P) -- n copies of a caller-given pm x pk sparsity pattern P (the tests and the bench pass the
 * pattern of a real SuiteSparse matrix, nasa4704) on the block diagonal, each coupled to its neighbours through the same pattern
 * (T_n tridiagonal): M = n * pm rows, row i * pm + p holds the columns j * pk + P[p][*] for j = i - 1, i, i + 1; values U(-1,1) from the
 * counter RNG (halved in the off-diagonal blocks).  `variant` bits: 1 = rectangular (every third column dropped, the rest renumbered:
 * K = 2/3 of n * pk), 2 = unsymmetric pattern (30 % of the strictly lower entries dropped).  *K_out = columns of the result.  The
 * pattern arrays are HOST pointers in both forms.  Same bits on host and device; any row range [r0, r1). */
int sextans_gen_kron_host(int n, int pm, int pk, const int *p_row_ptr, const int *p_col_idx, int variant, uint64_t seed, int r0, int r1,
                          int **row_ptr, int **col_idx, float **val, int64_t *nnz, int *K_out);
int sextans_gen_kron_device(int device, int n, int pm, int pk, const int *p_row_ptr, const int *p_col_idx, int variant, uint64_t seed,
                            int r0, int r1, int **d_row_ptr, int **d_col_idx, float **d_val, int64_t *nnz, int *K_out);
int sextans_gen_kkt_host(int n, int arrow, uint64_t seed, int r0, int r1, int **row_ptr, int **col_idx, float **val, int64_t *nnz);
int sextans_gen_kkt_device(int device, int n, int arrow, uint64_t seed, int r0, int r1, int **d_row_ptr, int **d_col_idx,
                           float **d_val, int64_t *nnz);
int sextans_gen_powerlaw_host(int M, int K, int xmin, int tail_x100, int max_len, uint64_t seed, int r0, int r1,
                              int **row_ptr, int **col_idx, float **val, int64_t *nnz);
int sextans_gen_powerlaw_device(int device, int M, int K, int xmin, int tail_x100, int max_len, uint64_t seed, int r0,
                                int r1, int **d_row_ptr, int **d_col_idx, float **d_val, int64_t *nnz);
/* Blocked-ELL synthetic input (BASELINE config 5): every block row gets `ell_width` distinct sorted
 * block columns, values bf16(U(-1,1)).  Device form allocates (free with sextans_device_free). */
int sextans_gen_bell_host(int M, int K, int ell_width, uint64_t seed, int **block_col, uint16_t **block_val);
int sextans_gen_bell_device(int device, int M, int K, int ell_width, uint64_t seed, int **d_block_col,
                            uint16_t **d_block_val);
/* Block-banded blocked-ELL (ell_width = 2 * half_width + 1 consecutive block columns around the diagonal block, window
 * shifted inwards at the edges): block rows share block columns, the case spmm_bell_mfma_shared is built for. */
int sextans_gen_bell_banded_host(int M, int K, int half_width, uint64_t seed, int **block_col, uint16_t **block_val);
int sextans_gen_bell_banded_device(int device, int M, int K, int half_width, uint64_t seed, int **d_block_col,
                                   uint16_t **d_block_val);
/* bf16(U(-1,1)) fill (same bits on host and device). */
int sextans_gen_uniform_bf16_host(uint16_t *dst, int64_t n, uint64_t seed);
int sextans_gen_uniform_bf16_device(int device, uint16_t *d_dst, int64_t n, uint64_t seed, void *stream);
/* U(-1,1) fp32 fill, element i of stream `seed` (same bits on host and device). */
int sextans_gen_uniform_host(float *dst, int64_t n, uint64_t seed);
int sextans_gen_uniform_device(int device, float *d_dst, int64_t n, uint64_t seed, void *stream);
int sextans_device_free(int device, void *d_ptr);
/* Device memory helpers for callers of the device generators that have no HIP binding of their own (Python tools, the bench): they go
 * through the ONE HIP runtime this library is linked against -- a second dlopen("libamdhip64.so") by another name can map a second
 * runtime whose calls fail on this one's pointers.  Synchronous with respect to the host; kind = SEXTANS_COPY_*. */
#define SEXTANS_COPY_HOST_TO_DEVICE 1
#define SEXTANS_COPY_DEVICE_TO_HOST 2
#define SEXTANS_COPY_DEVICE_TO_DEVICE 3
int sextans_device_alloc(int device, size_t bytes, void **d_ptr);
int sextans_device_copy(int device, void *dst, const void *src, size_t bytes, int kind);

#ifdef __cplusplus
}
#endif
#endif /* SEXTANS_AMD_H */
