// ref_wrap.cpp -- extern "C" shims over the REFERENCE's own host functions.  TEST INFRASTRUCTURE ONLY.
//
// This file contains no reference code.  oracle/Makefile streams the reference header
// /root/reference/src/sparse_helper.h (which #includes its mmio.h) in front of this file
// on g++'s stdin, so the symbols used below are the reference's real implementations:
//   read_suitsparse_matrix  sparse_helper.h:169-259
//   CSC_2_CSR               sparse_helper.h:475-509
//   cpu_spmm_CSR            sparse_helper.h:262-290
//   generate_edge_list_for_all_PEs  sparse_helper.h:345-403  (the FPGA non-zero scheduler)
// (edge_list_64bit, sparse_helper.h:406-473, is NOT built: its signature needs tapa::aligned_allocator
//  from the absent TAPA headers, and oracle/Makefile strips it rather than stand in for TAPA.)
// The result (oracle/_ref/libsextans_ref.so) is used by tests to pin oracle/sextans_oracle.c
// and to generate tests/golden/, and by bench.py as the "reference" CPU baseline.
//
// NOTE: the reference's loader prints to stdout and calls exit(1) on malformed input; callers
// only hand it well-formed files.
#include <chrono>
#include <cstring>

extern "C" {

// Loads `path` exactly as sextans-host.cpp:67-84 does (CSC read, then CSC_2_CSR) and returns
// malloc'ed CSR + CSC arrays.  fmt_first = 1 (CSC, what the reference host uses).
int ref_load_mtx(const char *path, int *M, int *K, int *nnz, int **csc_ptr, int **csc_idx,
                 float **csc_val, int **csr_ptr, int **csr_idx, float **csr_val) {
    vector<int> cp, ci, rp, ri;
    vector<float> cv, rv;
    read_suitsparse_matrix(const_cast<char *>(path), cp, ci, cv, *M, *K, *nnz, CSC);
    CSC_2_CSR(*M, *K, *nnz, cp, ci, cv, rp, ri, rv);
    auto dupi = [](const vector<int> &v) {
        int *p = (int *)malloc(sizeof(int) * (v.size() ? v.size() : 1));
        if (!v.empty()) memcpy(p, v.data(), sizeof(int) * v.size());
        return p;
    };
    auto dupf = [](const vector<float> &v, size_t n) {
        float *p = (float *)malloc(sizeof(float) * (n ? n : 1));
        if (n) memcpy(p, v.data(), sizeof(float) * n);
        return p;
    };
    *csc_ptr = dupi(cp); *csc_idx = dupi(ci); *csc_val = dupf(cv, (size_t)*nnz);
    *csr_ptr = dupi(rp); *csr_idx = dupi(ri); *csr_val = dupf(rv, (size_t)*nnz);
    return 0;
}

// Direct CSR read (read_suitsparse_matrix with mf = CSR, sparse_helper.h:207-208).
int ref_load_mtx_csr(const char *path, int *M, int *K, int *nnz, int **ptr, int **idx,
                     float **val) {
    vector<int> p, i;
    vector<float> v;
    read_suitsparse_matrix(const_cast<char *>(path), p, i, v, *M, *K, *nnz, CSR);
    *ptr = (int *)malloc(sizeof(int) * (p.size() ? p.size() : 1));
    *idx = (int *)malloc(sizeof(int) * (*nnz ? *nnz : 1));
    *val = (float *)malloc(sizeof(float) * (*nnz ? *nnz : 1));
    memcpy(*ptr, p.data(), sizeof(int) * p.size());
    if (*nnz) { memcpy(*idx, i.data(), sizeof(int) * *nnz); memcpy(*val, v.data(), sizeof(float) * *nnz); }
    return 0;
}

void ref_free(void *p) { free(p); }

// cpu_spmm_CSR on caller buffers (copied into the std::vectors the reference signature wants).
// Returns the wall time of the cpu_spmm_CSR call itself in seconds (steady_clock, as
// sextans-host.cpp:207-217 times it).
double ref_cpu_spmm_csr(int M, int N, int K, int nnz, float alpha, const int *row_ptr,
                        const int *col_idx, const float *val, const float *B, float beta,
                        float *C) {
    vector<int> rp(row_ptr, row_ptr + M + 1), ci(col_idx, col_idx + nnz);
    vector<float> v(val, val + nnz), b(B, B + (size_t)K * N), c(C, C + (size_t)M * N);
    auto t0 = std::chrono::steady_clock::now();
    cpu_spmm_CSR(M, N, K, nnz, alpha, rp, ci, v, b, beta, c);
    auto t1 = std::chrono::steady_clock::now();
    memcpy(C, c.data(), sizeof(float) * (size_t)M * N);
    return std::chrono::duration<double>(t1 - t0).count();
}

// The reference scheduler on caller CSC arrays with the host's constants (64 PEs, window 4096,
// distance 10: sextans-host.cpp:119-129, sextans.h:7-12).  Returns malloc'ed ptr[num_windows + 1] and,
// for PE p and slot i < L = ptr[num_windows], row/col/val[p * L + i]; row = -1 marks a bubble.
int ref_generate_edge_list(int M, int K, int nnz, const int *col_ptr, const int *row_idx, const float *val,
                           int *num_windows, int *L, int **ptr, int **row, int **col, float **attr) {
    vector<int> cp(col_ptr, col_ptr + K + 1), ri(row_idx, row_idx + nnz);
    vector<float> cv(val, val + nnz);
    vector<vector<edge> > pes;
    vector<int> p;
    generate_edge_list_for_all_PEs(cp, ri, cv, 64, M, K, 4096, pes, p, 10);
    *num_windows = (int)p.size() - 1;
    *L = p.back();
    const size_t n = (size_t)64 * (size_t)(*L);
    *ptr = (int *)malloc(sizeof(int) * p.size());
    memcpy(*ptr, p.data(), sizeof(int) * p.size());
    *row = (int *)malloc(sizeof(int) * (n ? n : 1));
    *col = (int *)malloc(sizeof(int) * (n ? n : 1));
    *attr = (float *)malloc(sizeof(float) * (n ? n : 1));
    for (int q = 0; q < 64; ++q)
        for (int i = 0; i < *L; ++i) {
            const edge &e = pes[q][i];
            (*row)[(size_t)q * *L + i] = e.row;
            (*col)[(size_t)q * *L + i] = e.col;
            (*attr)[(size_t)q * *L + i] = e.attr;
        }
    return 0;
}

}  // extern "C"
