// examples/dist_spmm.cpp -- multi-GPU SpMM through the C ABI alone (no Python, no torch): what a maintainer of the
// reference host program would write to shard  C = alpha*A*B + beta*C  over the GPUs of one node.
//
//   dist_spmm <A.mtx> <N> [gpus] [rm] [onedevice]
//       one host thread per GPU; default gpus = all visible gfx950 devices;
//       rm: row-major B and C through sextans_dist_spmm_rm as well (slabs in place, in-place all-gather);
//       onedevice: `gpus` ranks as threads on device 0 -- with a collectives library that supports it (SEXTANS_RCCL_PATH = the
//       loopback communicator of tests/fake_rccl.cpp): how the multi-rank code is exercised on a single-GPU box.
//
// Every rank loads the matrix (sextans_mtx_read), takes an nnz-balanced row range (sextans_partition_rows_by_nnz),
// keeps its rows in its own engine, holds the full B and C_in, and calls sextans_dist_spmm; C_out is complete on
// every rank afterwards.  Rank 0 then recomputes the product on one GPU (sextans_spmm_device) and compares bit for
// bit (rows that the engine re-associates -- hub rows of power-law matrices -- are compared with the reference's
// 1e-4 criterion instead).  Exit code 0 = match.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "sextans_amd.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != SEXTANS_OK) {                                                                     \
            fprintf(stderr, "%s: %s (%s)\n", #call, sextans_error_string(rc_), sextans_last_error()); \
            exit(2);                                                                                 \
        }                                                                                            \
    } while (0)
#define HIP(call) do { if ((call) != hipSuccess) { fprintf(stderr, "%s failed\n", #call); exit(2); } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <A.mtx> <N> [gpus] [rm] [onedevice]\n", argv[0]); return 1; }
    bool rm = false, one_device = false;   // (keywords in any order behind the rank count)
    for (int a = 4; a < argc; ++a) {
        if (!strcmp(argv[a], "rm")) rm = true;
        else if (!strcmp(argv[a], "onedevice")) one_device = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[a]); return 1; }
    }
    const int N = sextans_round_up_n(atoi(argv[2]));
    int world = 0;
    if (sextans_device_count(&world) != SEXTANS_OK || world < 1) { fprintf(stderr, "no gfx950 device\n"); return 2; }
    if (argc > 3) world = one_device ? std::max(1, atoi(argv[3])) : std::min(world, std::max(1, atoi(argv[3])));
    int M, K, nnz, *rp, *ci;
    float *va;
    CHECK(sextans_mtx_read(argv[1], SEXTANS_FMT_CSR, &M, &K, &nnz, &rp, &ci, &va));
    std::vector<int> ranges(2 * (size_t)world);
    CHECK(sextans_partition_rows_by_nnz(M, rp, world, ranges.data()));
    std::vector<float> B((size_t)K * N), Cin((size_t)M * N);
    for (size_t i = 0; i < B.size(); ++i) B[i] = (float)((i * 7 + 3) % 31) / 16.0f - 1.0f;
    for (size_t i = 0; i < Cin.size(); ++i) Cin[i] = (float)((i * 5 + 1) % 29) / 8.0f - 1.5f;
    const float alpha = 0.85f, beta = -2.06f;                        // sextans-host.cpp:29-30
    char id[128];
    CHECK(sextans_dist_unique_id(id));
    std::vector<std::vector<float>> result((size_t)world), result_rm((size_t)world);
    std::vector<float> Br, Cr;   // row-major copies of the same operands
    if (rm) {
        Br.resize(B.size()); Cr.resize(Cin.size());
        for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) Br[(size_t)k * N + n] = B[(size_t)n * K + k];
        for (int r = 0; r < M; ++r) for (int n = 0; n < N; ++n) Cr[(size_t)r * N + n] = Cin[(size_t)n * M + r];
    }
    std::vector<std::thread> ranks;
    for (int g = 0; g < world; ++g)
        ranks.emplace_back([&, g]() {
            const int dev = one_device ? 0 : g;
            HIP(hipSetDevice(dev));
            void *comm = nullptr;
            CHECK(sextans_dist_comm_init(&comm, dev, world, g, id));
            sextans_handle_t h = nullptr;
            CHECK(sextans_create(&h, dev));
            const int r0 = ranges[2 * (size_t)g], r1 = ranges[2 * (size_t)g + 1];
            std::vector<int> lrp((size_t)(r1 - r0) + 1);
            for (int r = r0; r <= r1; ++r) lrp[(size_t)(r - r0)] = rp[r] - rp[r0];
            CHECK(sextans_set_matrix_csr(h, r1 - r0, K, rp[r1] - rp[r0], lrp.data(), ci + rp[r0], va + rp[r0]));
            float *dB, *dCin, *dCout;
            HIP(hipMalloc((void **)&dB, B.size() * 4)); HIP(hipMalloc((void **)&dCin, Cin.size() * 4));
            HIP(hipMalloc((void **)&dCout, Cin.size() * 4));
            HIP(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
            HIP(hipMemcpy(dCin, Cin.data(), Cin.size() * 4, hipMemcpyHostToDevice));
            hipStream_t st;
            HIP(hipStreamCreate(&st));
            // everything collective or slow -- non-zero counts, plan build, cut lists -- outside the call a caller would time
            CHECK(sextans_dist_prepare(h, comm, world, g, ranges.data(), N, 4, SEXTANS_DIST_CSR_COLMAJOR, (void *)st));
            CHECK(sextans_dist_spmm(h, comm, world, g, ranges.data(), N, alpha, dB, K, beta, dCin, M, dCout, M, 4, (void *)st));
            HIP(hipStreamSynchronize(st));
            result[(size_t)g].resize(Cin.size());
            HIP(hipMemcpy(result[(size_t)g].data(), dCout, Cin.size() * 4, hipMemcpyDeviceToHost));
            if (rm) {   // the same product on row-major operands: in place inside C (C_in == C_out), runs exchanged in place
                HIP(hipMemcpy(dB, Br.data(), Br.size() * 4, hipMemcpyHostToDevice));
                HIP(hipMemcpy(dCin, Cr.data(), Cr.size() * 4, hipMemcpyHostToDevice));
                CHECK(sextans_dist_spmm_rm(h, comm, world, g, ranges.data(), N, alpha, dB, N, beta, dCin, N, dCin, N, (void *)st));
                HIP(hipStreamSynchronize(st));
                std::vector<float> t(Cin.size());
                HIP(hipMemcpy(t.data(), dCin, Cin.size() * 4, hipMemcpyDeviceToHost));
                result_rm[(size_t)g].resize(Cin.size());
                for (int r = 0; r < M; ++r) for (int n = 0; n < N; ++n) result_rm[(size_t)g][(size_t)n * M + r] = t[(size_t)r * N + n];
            }
            sextans_destroy(h);
            sextans_dist_comm_destroy(comm);
            (void)hipFree(dB); (void)hipFree(dCin); (void)hipFree(dCout); (void)hipStreamDestroy(st);
        });
    for (auto &t : ranks) t.join();
    // single-GPU result of the same product
    HIP(hipSetDevice(0));
    sextans_handle_t h = nullptr;
    CHECK(sextans_create(&h, 0));
    CHECK(sextans_set_matrix_csr(h, M, K, nnz, rp, ci, va));
    std::vector<float> single = Cin;
    CHECK(sextans_spmm_host(h, N, alpha, B.data(), beta, single.data(), 1, nullptr));
    sextans_destroy(h);
    int bad = 0;
    for (int form = 0; form < (rm ? 2 : 1); ++form)
    for (int g = 0; g < world; ++g) {
        const std::vector<float> &got = form ? result_rm[(size_t)g] : result[(size_t)g];
        if (memcmp(got.data(), single.data(), single.size() * 4) == 0) continue;
        float pct = 0.f;                                             // hub rows are cut differently per rank range
        const int mism = sextans_verify(M, N, single.data(), got.data(), &pct);
        printf("rank %d%s: not bit-identical; reference criterion: %d mismatches (%.4f %%)\n", g, form ? " (row-major)" : "", mism, pct);
        bad += mism;
    }
    printf("dist_spmm: %d rank(s), M=%d K=%d nnz=%d N=%d%s: %s\n", world, M, K, nnz, N, rm ? ", column-major and row-major forms" : "", bad ? "MISMATCH" : "all ranks match the single-GPU result");
    sextans_host_free(rp); sextans_host_free(ci); sextans_host_free(va);
    return bad ? 3 : 0;
}
