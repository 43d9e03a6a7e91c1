// panel_plan.h -- the packed, row-bucketed form of A consumed by the LDS-panel kernel
// (spmm_csr_panel).
//
// Analogue of the reference's host-side non-zero scheduling (generate_edge_list_for_all_PEs,
// sparse_helper.h:345-403) and of its packed stream with window-local 14-bit column indices
// (edge_list_64bit, sparse_helper.h:419-443): Sextans re-indexes every non-zero relative to a
// 4096-column B window that its PEs hold on chip.  Here:
//   * rows are bucketed greedily into blocks of at most `rows_per_block` consecutive rows whose
//     DISTINCT columns fit the LDS panel (max_unique);
//   * every block gets the ascending list of those columns (its dictionary) -- the kernel copies
//     exactly these B rows into LDS once;
//   * every non-zero is re-encoded as (16-bit index into the dictionary, fp32 value) in a packed
//     stream in which each row starts on a 4-entry boundary, so a lane fetches 4 entries with one
//     8-byte + one 16-byte load;
//   * blocks without enough reuse (nnz < min_reuse * distinct) keep an empty dictionary ("direct"):
//     their stream carries the 32-bit column instead and B rows are gathered from global memory.
// Row order and per-row non-zero order are untouched, so results stay bit-identical to
// cpu_spmm_CSR (sparse_helper.h:262-290).
#pragma once
#include <cstdint>
#include <vector>

namespace sx {

constexpr uint16_t kPadIndex = 0xFFFF;   // idx16 of padding entries in dictionary rows (value -0.0f)

struct PanelPlan {
    int rows_per_block = 0;
    int max_unique = 0;
    std::vector<int> blk_row;        // nblk + 1: block b owns rows [blk_row[b], blk_row[b+1])
    std::vector<int> dict_ptr;       // nblk + 1 offsets into dict (empty range = direct block)
    std::vector<int> dict;           // distinct columns per dictionary block, ascending
    std::vector<int> row_off;        // M + 1: first packed entry of each row (multiple of 4); row_off[M] = total
    std::vector<uint16_t> idx16;     // packed local indices (dictionary rows; kPadIndex in their padding; 0 elsewhere)
    std::vector<int> col32;          // packed 32-bit columns (direct rows; 0 elsewhere/padding)
    std::vector<float> val;          // packed values (-0.0f in the padding of dictionary rows, 0 elsewhere)
    int64_t nnz_in_panel_blocks = 0;
    int64_t nnz_total = 0;
    int max_dict = 0;
};

// Build the plan on the host (multi-threaded).  `val` is the CSR value array.
void build_panel_plan(int M, int K, const int *row_ptr, const int *col_idx, const float *val,
                      int rows_per_block, int max_unique, double min_reuse, PanelPlan &out);

}  // namespace sx
