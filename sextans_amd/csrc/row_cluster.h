// row_cluster.h -- graph-compact row blocks for the LDS-panel plan (csrc/row_cluster.hip).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace sx {

// Strides of a Cartesian-grid stencil matrix in natural ordering, inferred from the column offsets of a sample of rows:
// s2 = rows per grid line (distance between neighbouring column clusters of a row), s3 = rows per grid plane (distance between groups
// of clusters; 0 for a 2-D grid).  Returns false when the sampled rows do not agree on such a structure.
struct GridStrides { long long s2 = 0, s3 = 0; };
bool detect_grid_strides(int M, const std::vector<int> &sample_rows, const std::vector<std::vector<int>> &sample_cols, GridStrides *out);

// perm[i] = row of the matrix that becomes row i of the clustered order: rows sorted brick by brick, a brick = (a run of <= run_rows
// rows of a grid line) x b2 lines x b3 planes; cut[i] = 1 where a brick starts (the plan builder starts a row block there).
// Device arrays of M ints / bytes (caller frees).  Returns non-zero on a HIP error.
int build_brick_order_device(int M, GridStrides s, int run_rows, int b2, int b3, int super_group, int **d_perm, unsigned char **d_cut,
                             std::string &err);

// Rows of a CSR matrix gathered in the order perm (columns untouched): new device arrays (caller frees).
int permute_csr_rows_device(int M, int64_t nnz, const int *d_rp, const int *d_ci, const float *d_v, const int *d_perm, int **o_rp,
                            int **o_ci, float **o_v, std::string &err);

// slot_row[b * RB + s] = perm[blk_row[b] + s] (slots past the block's last row repeat the last row): what the kernel needs to
// address C, at an address that depends on the block number only.
int build_slot_rows_device(int nblk, int RB, const int *d_blk_row, const int *d_perm, int **d_slot_row, std::string &err);

}  // namespace sx
