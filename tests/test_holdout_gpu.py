"""HOLDOUT inputs (round 5, VERDICT r04 task 1): kron(T_n, nasa4704) -- the pattern of the one real SuiteSparse matrix in this mount
on the block diagonal, coupled tridiagonally -- in three numberings plus a rectangular and an unsymmetric-pattern variant, through
the C ABI, bit-identical to cpu_spmm_CSR (sparse_helper.h:262-290): the whole matrix at 47 k rows, 1 000 sampled rows at 4.0 M rows.
The reference evaluates on SuiteSparse matrices of any shape (README.md:17-18,31; sparse_helper.h:345-403 schedules any M x K)."""
import numpy as np
import pytest

from util import ALPHA, BETA

pytestmark = pytest.mark.gpu


def _pull(ptr, n, dt):
    """Device array -> numpy, through the library's own HIP runtime (no second dlopen of libamdhip64 by another name)."""
    import torch
    from sextans_amd import api
    out = np.empty(n, dt)
    torch.cuda.synchronize()
    api.device_copy(0, out.ctypes.data, ptr, out.nbytes, api.COPY_D2H)
    return out


@pytest.mark.parametrize("variant", ["", "rect", "unsym", "rectunsym"])
def test_device_generator_bit_identical_to_host(sx, variant):
    from sextans_amd import api, holdout
    prp, pci, pm, pk = holdout.nasa_pattern()
    n, r0, r1 = 4, 3000, 17000
    p, i, v, nnz, K = api.gen_kron_device(0, n, prp, pci, pk, holdout.VARIANTS[variant], holdout.SEED, r0, r1)
    try:
        hp, hi, hv, hK = api.gen_kron_host(n, prp, pci, pk, holdout.VARIANTS[variant], holdout.SEED, r0, r1)
        assert (nnz, K) == (len(hi), hK)
        assert np.array_equal(_pull(p, len(hp), np.int32), hp) and np.array_equal(_pull(i, nnz, np.int32), hi)
        assert np.array_equal(_pull(v, nnz, np.float32).view(np.uint32), hv.view(np.uint32))
    finally:
        for q in (p, i, v):
            api.device_free(0, q)


def _numbered(rp, ci, v, M, numbering, n):
    from sextans_amd import holdout, meshgen
    if numbering == "natural":
        return rp, ci, v
    perm = holdout.random_permutation(M) if numbering == "random" else holdout.rcm_permutation(n)
    return meshgen.permute_symmetric(rp, ci, v, M, perm)


CASES = [("", "natural"), ("", "random"), ("", "rcm"), ("rect", "natural"), ("unsym", "natural"), ("unsym", "random"), ("rectunsym", "natural")]


@pytest.mark.parametrize("variant,numbering", CASES)
def test_whole_matrix_against_the_oracle(engine, oracle, variant, numbering):
    """47 040 rows (n = 10), every variant and numbering, N = 16 / 24 / 128, automatic dispatch and the natural-order forms."""
    from sextans_amd import holdout
    n = 10
    rp, ci, v, M, K = holdout.kron_host(n, variant)
    rp, ci, v = _numbered(rp, ci, v, M, numbering, n)
    rs = np.random.RandomState(len(variant) + len(numbering))
    try:
        for N in (16, 24, 128):
            B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
            want = C0.copy()
            oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
            for rc in (-1, 0, 2):
                engine.set_option("row_cluster", rc)
                engine.set_matrix_csr(M, K, rp, ci, v)
                for rp_time in (1, 3):
                    got = C0.copy()
                    engine.spmm(N, ALPHA, B, BETA, got, rp_time=rp_time)
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (variant, numbering, N, rc, rp_time, engine.last_kernel())
                if rc == 2:      # forced: every variant clusters -- rectangular and unsymmetric patterns over the row-similarity graph
                    assert int(engine.get_stat("row_cluster")) == 2 and engine.last_kernel() == "spmm_csr_panel_v2_reordered", (
                        variant, numbering, N, engine.get_stat("cluster_decline"))
                    # graph the rows were clustered over: the matrix itself / row similarity (rectangular) / A + A^T (unsymmetric pattern)
                    # ("": 3 as well -- at this size a few rows sit on the long-row path, their mirrors are missing from the main matrix)
                    assert int(engine.get_stat("cluster_graph_kind")) in {"": (0, 3), "rect": (2,), "unsym": (3,), "rectunsym": (2,)}[variant], (variant, engine.get_stat("pattern_symmetry"))
                    assert engine.get_stat("panel_rows_clustered") < 12 * M      # (natural order: 11.7 dictionary rows per matrix row)
    finally:
        engine.set_option("row_cluster", -1)


@pytest.mark.parametrize("variant,numbering", [("", "natural"), ("", "random"), ("rect", "natural"), ("unsym", "natural")])
def test_full_size_sampled_rows(variant, numbering):
    """4.0 M rows, 267 M non-zeros (n = 850), generated (and renumbered) in HBM, N = 16: 1 000 sampled rows recomputed by the oracle
    from the HOST generator's rows (+ the same renumbering)."""
    import torch
    from oracle.bindings import Oracle
    from sextans_amd import api, holdout
    n, N = 850, 16
    pat = holdout.nasa_pattern()
    M, K, p, i, v, nnz = holdout.kron_device(0, n, variant, numbering, pattern=pat)
    st = torch.cuda.current_stream().cuda_stream
    try:
        assert M == 850 * 4704 and (nnz == (3 * n - 2) * 104756 if variant == "" else nnz < (3 * n - 2) * 104756)
        B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); out = torch.zeros(M * N, device="cuda")
        api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
        with api.Engine(0) as e:
            e.set_matrix_csr_device(M, K, nnz, p, i, v)
            e.spmm_device(N, float(ALPHA), B.data_ptr(), K, float(BETA), Cin.data_ptr(), out.data_ptr(), M, st)
            torch.cuda.synchronize()
            print("holdout", variant or "sym", numbering, e.last_kernel(), "row_cluster", e.get_stat("row_cluster"), "decline", e.get_stat("cluster_decline"),
                  "graph", e.get_stat("cluster_graph_kind"), "symmetry", e.get_stat("pattern_symmetry"))
            assert int(e.get_stat("cluster_decline")) != 1
        o = Oracle()
        Bh = B.cpu().numpy(); Ch = Cin.cpu().numpy().reshape(N, M); got = out.cpu().numpy().reshape(N, M)
        perm = holdout.random_permutation(M) if numbering == "random" else None
        old_of_new = None
        if perm is not None:
            old_of_new = np.empty(M, np.int64); old_of_new[perm] = np.arange(M)
        for r in np.random.RandomState(3).choice(M, 1000, replace=False):
            r_old = int(r if perm is None else old_of_new[r])
            rp1, ci1, v1, _ = api.gen_kron_host(n, pat[0], pat[1], pat[3], holdout.VARIANTS[variant], holdout.SEED, r_old, r_old + 1)
            if perm is not None:
                cols = perm[ci1].astype(np.int32)
                order = np.argsort(cols, kind="stable")
                ci1, v1 = cols[order], v1[order]
            want = np.ascontiguousarray(Ch[:, r]).copy()
            o.spmm(1, N, K, ALPHA, rp1, np.ascontiguousarray(ci1, np.int32), v1, Bh, BETA, want)
            assert np.array_equal(want.view(np.uint32), np.ascontiguousarray(got[:, r]).view(np.uint32)), (variant, numbering, int(r))
    finally:
        for q in (p, i, v):
            api.device_free(0, q)


def test_run_level_clustering_on_the_file_order(engine, oracle):
    """Round 5, a measured negative kept behind option "run_cluster" (default off): a matrix in a numbering WITH locality whose natural
    row blocks are cut short by the panel capacity (the holdout class in its file order) keeps runs of 16 consecutive rows together and
    builds its row blocks from runs chosen over the graph of runs.  Fewer panel rows, no staging passes (the plain panel kernel through
    a slot -> row table), bit-identical -- and no faster, hence off by default."""
    from sextans_amd import holdout
    n = 16                                                  # 75 264 rows (the form needs >= 65 536)
    rp, ci, v, M, K = holdout.kron_host(n)
    rs = np.random.RandomState(4)
    try:
        for N in (16, 40, 128):
            B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
            want = C0.copy()
            oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
            for run_cluster in (2, 0):
                engine.set_option("row_cluster", -1); engine.set_option("run_cluster", run_cluster)
                engine.set_option("fuse_b", 0)              # (B of this size would be staged from column-major: the repacked path is what is tested)
                engine.set_matrix_csr(M, K, rp, ci, v)
                for rp_time in (1, 3):
                    got = C0.copy()
                    engine.spmm(N, ALPHA, B, BETA, got, rp_time=rp_time)
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, run_cluster, rp_time, engine.last_kernel())
                assert engine.last_kernel() == "spmm_csr_panel_v2"
                if run_cluster and int(engine.get_stat("cluster_decline")) == 12:
                    assert int(engine.get_stat("cluster_runs")) == 1 and int(engine.get_stat("row_cluster")) == 1
                    assert engine.get_stat("panel_rows_clustered") < engine.get_stat("panel_rows_natural")
                else:
                    assert int(engine.get_stat("cluster_runs")) == 0
    finally:
        engine.set_option("run_cluster", 0); engine.set_option("row_cluster", -1); engine.set_option("fuse_b", 1)
