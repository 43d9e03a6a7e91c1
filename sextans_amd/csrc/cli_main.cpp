// cli_main.cpp -- the `sextans` command of the MI355X engine.  Same call surface and report as the
// reference's main() (sextans-host.cpp:26-292):
//
//     sextans <A.mtx> <N> [rp_time] [alpha] [beta]
//
//   argc 3: N only; argc 4: rp_time; argc 5: alpha beta; argc 6: rp_time alpha beta; anything else
//   prints the usage line and returns EXIT_FAILURE (sextans-host.cpp:33-48).  N is rounded up to a
//   multiple of 8 (:51); alpha = 0.85, beta = -2.06, rp_time = 1 by default (:29-31).
//
// The reference selects its backend with env TAPAB (bitstream path; empty = software simulation,
// :231-238).  This build has exactly one backend, the gfx950 HIP engine; SEXTANS_DEVICE picks the
// device index.  Without a usable device the command fails loudly (exit 1): there is no CPU path
// for the product computation.  The CPU golden below exists only for the built-in self check the
// reference performs (:206-219, :262-289).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "sextans_amd.h"

using std::cout;

static int fail(const char *what, int rc) {
    cout << what << ": " << sextans_error_string(rc);
    const char *detail = sextans_last_error();
    if (detail && *detail) cout << " [" << detail << "]";
    cout << std::endl;
    return 1;
}

int main(int argc, char **argv) {
    printf("start host\n");
    float alpha = 0.85f, beta = -2.06f;
    int rp_time = 1;
    switch (argc) {
        case 6: rp_time = atoi(argv[3]); alpha = (float)atof(argv[4]); beta = (float)atof(argv[5]); break;
        case 5: alpha = (float)atof(argv[3]); beta = (float)atof(argv[4]); break;
        case 4: rp_time = atoi(argv[3]); break;
        case 3: break;
        default:
            cout << "Usage: " << argv[0] << " [matrix A file] [N] [rp_time] [alpha] [beta]" << std::endl;
            return EXIT_FAILURE;
    }
    const char *path = argv[1];
    const int N = sextans_round_up_n(atoi(argv[2]));
    cout << "N = " << N << "\n";
    cout << "alpha = " << alpha << "\n";
    cout << "beta = " << beta << "\n";

    cout << "Reading sparse A matrix...";
    int M = 0, K = 0, nnz = 0;
    int *row_ptr = nullptr, *col_idx = nullptr;
    float *val = nullptr;
    if (int rc = sextans_mtx_read(path, SEXTANS_FMT_CSR, &M, &K, &nnz, &row_ptr, &col_idx, &val)) {
        cout << "\n";
        if (rc == SEXTANS_ERR_OPEN) cout << "Could not open " << path << std::endl;
        else cout << "Could not read " << path << ": " << sextans_error_string(rc) << std::endl;
        return 1;   // the reference exit(1)s here (sparse_helper.h:181-191)
    }
    cout << "done\n";
    cout << "Matrix size: \n";
    cout << "A: sparse matrix, " << M << " x " << K << ". NNZ = " << nnz << "\n";
    cout << "B: dense matrix, " << K << " x " << N << "\n";
    cout << "C: dense matrix, " << M << " x " << N << "\n";

    std::vector<float> B((size_t)K * N), C_cpu((size_t)M * N), C_dev;
    cout << "Generating dense matirx B ...";
    sextans_init_dense_B(K, N, B.data());
    cout << "Generating dense matirx C ...";
    sextans_init_dense_C(M, N, C_cpu.data());
    C_dev = C_cpu;
    cout << "done\n";

    // Device-side preparation (the reference prepares its FPGA streams here, :114-204).
    cout << "Preparing sparse A for MI355X ...";
    int dev = 0;
    if (const char *d = getenv("SEXTANS_DEVICE")) dev = atoi(d);
    sextans_handle_t h = nullptr;
    if (int rc = sextans_create(&h, dev)) { cout << "\n"; return fail("sextans_create", rc); }
    if (int rc = sextans_set_matrix_csr(h, M, K, nnz, row_ptr, col_idx, val)) {
        cout << "\n";
        return fail("sextans_set_matrix_csr", rc);
    }
    cout << "done\n";

    cout << "Run spmm on cpu...";
    auto t0 = std::chrono::steady_clock::now();
    sextans_selfcheck_golden(M, N, K, alpha, row_ptr, col_idx, val, B.data(), beta, C_cpu.data());
    auto t1 = std::chrono::steady_clock::now();
    const double time_cpu = std::chrono::duration<double>(t1 - t0).count();
    cout << "done (" << time_cpu * 1000 << " msec)\n";
    cout << "CPU GFLOPS: " << sextans_gflops(M, N, nnz, time_cpu) << "\n";

    cout << "launch kernel\n";
    double elapsed_ns = 0.0;
    if (int rc = sextans_spmm_host(h, N, alpha, B.data(), beta, C_dev.data(), rp_time, &elapsed_ns))
        return fail("sextans_spmm_host", rc);
    const double time_taken = elapsed_ns * (1e-9 / (rp_time < 1 ? 1 : rp_time));   // :252
    printf("Kernel time is %f ms\n", time_taken * 1000);
    printf("GFLOPS:%f \n", (float)sextans_gflops(M, N, nnz, time_taken));

    float pct = 0.f;
    const int mismatch = sextans_verify(M, N, C_cpu.data(), C_dev.data(), &pct);
    if (pct < 2.0f) cout << "Success!\n";
    else cout << "Failed.\n";
    printf("num_mismatch = %d, percent = %.2f%%\n", mismatch, pct);

    sextans_destroy(h);
    sextans_host_free(row_ptr); sextans_host_free(col_idx); sextans_host_free(val);
    return EXIT_SUCCESS;
}
