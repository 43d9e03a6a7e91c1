// plan_device.h -- device-side builder of the packed row-bucketed form of A (see plan_device.hip, panel_plan.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace sx {

constexpr int kPlanPartBlocks = 64;   // rows are cut into parts of kPlanPartBlocks * rows_per_block rows; every part starts a
                                      // new block (host and device builders alike, so their block lists are identical)
constexpr int kPlanPadRows = 16;      // +1.0f rows behind the dictionary in the LDS panel when index lists are shared with a shift: a padding entry of
                                      // a shifted list addresses pad row + shift (at most kPlanPadRows - 1 panel rows)
constexpr int kPlanTailPad = 256;     // zero entries behind the last row (the kernels fetch up to 6 batches of 16 ahead)

struct DevicePlan {                   // device arrays in exactly the form the LDS-panel kernels read
    int lpr = 0, rows_per_block = 0, sets = 1;   // rows_per_block = sets * (256 / lpr) row slots
    int nblk = 0;
    int *d_blk_row = nullptr;         // nblk + 1
    int *d_dict_cnt = nullptr;        // nblk: dictionary entries (0 = direct block)
    int *d_dict = nullptr;            // nblk x dict_stride, last column repeated
    int dict_stride = 0;
    int *d_slot_info = nullptr;       // nblk x rows_per_block x {first packed entry, entries}
    unsigned short *d_idx16 = nullptr;   // BYTE offset of the entry's B row in the panel (index * 16 * lpr); padding -> pad row
    int *d_ioff = nullptr;            // nblk x rows_per_block x {start of the slot's index list in d_idx16, shift in bytes to add to every
                                      // offset of the list} when lists are SHARED (null: every row has its own list at its first packed entry):
                                      // consecutive rows of a block whose lists are equal up to a constant shift keep one copy (share_index_lists)
    int64_t idx_len = 0;              // entries of d_idx16 (= stream_len without sharing)
    int64_t shared_rows = 0;          // rows whose index list is another row's
    int *d_col32 = nullptr;           // stream_len when `mixed`, else 1 element
    float *d_val = nullptr;
    int64_t stream_len = 0;
    int max_dict = 0;
    int64_t capacity_cuts = 0;        // blocks that ended because the next row's columns no longer fit the dictionary capacity
    int max_row_len = 0;              // longest row of the matrix the plan was built from
    bool mixed = false;
    int64_t nnz_in_panel_blocks = 0;
    int64_t total_dict = 0;           // sum of the block dictionaries (B rows copied into LDS per N tile)
    std::vector<int> h_blk_row;
};

// *bad: bit 0 = row_ptr not a monotone 0 .. nnz sequence, bit 1 = a column index outside [0, K).  Returns non-zero on a HIP error.
int validate_csr_device(int M, int K, int64_t nnz, const int *d_rp, const int *d_ci, int *bad, std::string &err);
// smallest / largest column index of a device CSR matrix (lo > hi: no entries).  Returns non-zero on a HIP error.
int column_range_device(int64_t nnz, const int *d_ci, int *lo, int *hi, std::string &err);
// flag[k / 64] = 1 where the matrix has an entry in that 64-column segment (the only rows of B a call repacks); ceil(K / 64) + 4 bytes
int column_touch_flags_device(int K, int64_t nnz, const int *d_ci, unsigned char **d_flag, int64_t *touched_segments, std::string &err);
void free_device_plan(DevicePlan &d);
int build_panel_plan_device(int M, int K, const int *d_rp, const int *d_ci, const float *d_v, int lpr, int max_unique,
                            double min_reuse, DevicePlan &out, std::string &err, const unsigned char *d_cut = nullptr,
                            bool share_index_lists = false, int sets = 1);

}  // namespace sx
