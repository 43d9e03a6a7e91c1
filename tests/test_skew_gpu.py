"""Automatic handling of skewed (power-law) matrices, SURVEY.md 8f row 1.  The reference balances by construction
(rows dealt to PEs by row % 64 + bubble padding, sparse_helper.h:345-403); here rows longer than a threshold chosen
from the matrix are summed in parallel pieces and folded in order.  Engine defaults plus the opt-in to re-association
(split_rows = -1, automatic threshold; the engine's own default is 0 = strict order for every row since round 3):
rows that are NOT hubs must stay bit-identical to cpu_spmm_CSR under every kernel; hub rows
(reported by sextans_reassociated_rows) must meet |d| <= 1e-4 * (|alpha| * sum|a*b| + |beta*c|)."""
import time

import numpy as np
import pytest

from util import ALPHA, BETA

pytestmark = pytest.mark.gpu


def _defaults(engine, **opts):
    d = dict(lanes_per_row=4, stage_a=1, xcd_remap=1, exact=1, kernel=0, panel_min_reuse_x100=400, fuse_b=1,
             split_rows=-1, bucket_rows=-1, window_rows=319, window_cols=65536, window_unroll=8)
    d.update(opts)
    for k, v in d.items():
        engine.set_option(k, v)


@pytest.fixture(autouse=True)
def _strict_order_afterwards(engine):
    yield
    engine.set_option("split_rows", 0)   # the shared engine goes back to the library default


def _check(out, want, M, N, rp, ci, v, B, C0, hubs, alpha, beta):
    o2, w2 = out.reshape(N, M), want.reshape(N, M)
    plain = np.ones(M, bool)
    plain[hubs] = False
    assert np.array_equal(o2[:, plain].view(np.uint32), w2[:, plain].view(np.uint32)), "a non-hub row changed"
    K = len(B) // N
    rows = np.repeat(np.arange(M), np.diff(rp))
    for n in range(N):
        bound = np.bincount(rows, weights=np.abs(v).astype(np.float64) * np.abs(B[n * K + ci]), minlength=M)
        bound = 1e-4 * (abs(float(alpha)) * bound + np.abs(float(beta) * C0[n * M:(n + 1) * M]))
        assert np.all(np.abs(o2[n].astype(np.float64) - w2[n]) <= bound + 1e-30), n


@pytest.mark.parametrize("kernel", [0, 1, 2, 3])
@pytest.mark.parametrize("N", [8, 16, 40])
def test_power_law_default_options(engine, oracle, kernel, N):
    from sextans_amd import api
    M = K = 20000
    rp, ci, v = api.gen_powerlaw_host(M, K, 3, 120, 15000, 11)
    lens = np.diff(rp)
    T = max(1024, int(rp[-1]) // 16384)
    expect = np.nonzero(lens > T)[0]
    assert len(expect) >= 3 and lens.max() > 5000
    rs = np.random.RandomState(N)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    _defaults(engine, kernel=kernel)
    engine.set_matrix_csr(M, K, rp, ci, v)
    hubs = engine.reassociated_rows()
    assert np.array_equal(hubs, expect) and engine.get_stat("split_threshold") == T
    L0 = max(32, 2 * (int(rp[-1]) // M))
    assert engine.get_stat("bucket_threshold") == L0 and engine.get_stat("piece_path_rows") == (lens > L0).sum() > 10 * len(hubs)
    out = C0.copy()
    engine.spmm(N, ALPHA, B, BETA, out, rp_time=2)
    assert engine.last_kernel().endswith("+hub_pieces")
    if kernel == 3:
        assert engine.last_kernel() == "spmm_csr_window+hub_pieces"
    _check(out, want, M, N, rp, ci, v, B, C0, hubs, ALPHA, BETA)
    # strict order on request: no row is split, everything bit-identical (and slow for the hubs); the long rows
    # still take the piece path, one piece each
    _defaults(engine, kernel=kernel, split_rows=0)
    out = C0.copy()
    engine.spmm(N, ALPHA, B, BETA, out)
    assert len(engine.reassociated_rows()) == 0 and engine.last_kernel().endswith("+hub_pieces")
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    _defaults(engine, kernel=kernel, split_rows=0, bucket_rows=0)          # ... or nothing leaves the main kernel
    engine.set_option("exact_chain", 0)                                    # (not even as exact chains)
    out = C0.copy()
    engine.spmm(N, ALPHA, B, BETA, out)
    assert "+hub" not in engine.last_kernel() and engine.get_stat("piece_path_rows") == 0
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    engine.set_option("exact_chain", 1)
    _defaults(engine)


def test_fem_with_hub_rows_keeps_the_panel_kernel(engine, oracle):
    """A matrix with B-row reuse plus a few hub rows: the LDS-panel kernel runs on the main matrix, the hubs go
    through the piece path; row-range calls see the hubs of their range only."""
    import torch
    from sextans_amd import api
    frp, fci, fv = api.gen_fem3d_host(20, 20, 12, 3, 5)
    M0 = 20 * 20 * 12 * 3
    K = M0
    rs = np.random.RandomState(17)
    hub_at = [7, 5000, M0 - 1]
    rows = []
    for r in range(M0):
        c, x = fci[frp[r]:frp[r + 1]], fv[frp[r]:frp[r + 1]]
        if r in hub_at:
            c = np.sort(rs.choice(K, size=6000 + r % 100, replace=False)).astype(np.int32)
            x = rs.uniform(-1, 1, len(c)).astype(np.float32)
        rows.append((c, x))
    M = M0
    rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum([len(c) for c, _ in rows])
    ci = np.concatenate([c for c, _ in rows]).astype(np.int32)
    v = np.concatenate([x for _, x in rows]).astype(np.float32)
    N = 16
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    alpha, beta = np.float32(-1.5), np.float32(0.75)
    want = C0.copy()
    oracle.spmm(M, N, K, alpha, rp, ci, v, B, beta, want)
    _defaults(engine, bucket_rows=0)
    engine.set_matrix_csr(M, K, rp, ci, v)
    assert list(engine.reassociated_rows()) == hub_at
    out = C0.copy()
    engine.spmm(N, float(alpha), B, float(beta), out)
    assert engine.last_kernel() in ("spmm_csr_panel+hub_pieces", "spmm_csr_panel_v2+hub_pieces")
    _check(out, want, M, N, rp, ci, v, B, C0, np.array(hub_at), alpha, beta)
    # in place (C_in == C_out) and in row ranges cut at kernel-friendly boundaries
    st = torch.cuda.current_stream().cuda_stream
    dB = torch.from_numpy(B).cuda(); dC = torch.from_numpy(C0).cuda()
    cuts = [0, engine.align_row(N, M // 2), M]
    for i in range(2):
        c0, c1 = cuts[i], cuts[i + 1]
        engine.spmm_device_rows(N, float(alpha), dB.data_ptr(), K, float(beta), dC.data_ptr() + 4 * c0, M,
                                dC.data_ptr() + 4 * c0, M, c0, c1, reuse_b_panels=i > 0, stream=st)
        assert engine.last_kernel() in ("spmm_csr_panel+hub_pieces", "spmm_csr_panel_v2+hub_pieces")
    torch.cuda.synchronize()
    assert np.array_equal(dC.cpu().numpy().view(np.uint32), out.view(np.uint32))     # same bits as the whole-matrix call


def test_power_law_1m_rows_within_1p5x_of_uniform(engine, sx):
    """VERDICT r01 task 5: a 1M-row power-law matrix runs within 1.5x of a uniform matrix with the same number
    of non-zeros, kernel = 0, hub rows re-associated (split_rows = -1: opt-in since round 3, everything else default)."""
    import torch
    from sextans_amd import api
    M = K = 1_000_000
    N = 16
    st = torch.cuda.current_stream().cuda_stream
    B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
    api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st)
    api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)

    def time_it(ptrs, nnz):
        with api.Engine(0) as e:                       # a fresh engine: every other option at its default
            e.set_option("split_rows", -1)
            e.set_matrix_csr_device(M, K, nnz, *ptrs)
            f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 10, e.last_kernel(), int(e.get_stat("reassociated_rows"))

    pl = api.gen_powerlaw_device(0, M, K, 6, 120, 400_000, 7)
    t_pl, k_pl, hubs = time_it(pl[:3], pl[3])
    for q in pl[:3]:
        api.device_free(0, q)
    un = api.gen_csr_device(0, M, K, pl[3] / M, 7)
    t_un, k_un, _ = time_it(un[:3], un[3])
    for q in un[:3]:
        api.device_free(0, q)
    print(f"power-law {pl[3]} nnz: {t_pl * 1e3:.3f} ms ({k_pl}, {hubs} hub rows); uniform {un[3]} nnz: {t_un * 1e3:.3f} ms ({k_un})")
    assert hubs > 100 and k_pl.endswith("+hub_pieces")
    assert abs(un[3] - pl[3]) < 0.02 * pl[3]
    assert t_pl <= 1.5 * t_un, (t_pl, t_un)


def test_cli_on_a_power_law_file(sx, tmp_path):
    """The reference's program flow (`sextans A.mtx N`) on a skewed matrix: hub rows are re-associated by default and
    must still pass the reference's own verification (sextans-host.cpp:262-289: 0 mismatches at 1e-4)."""
    import subprocess
    from sextans_amd import api
    M = K = 6000
    rp, ci, v = api.gen_powerlaw_host(M, K, 3, 120, 5000, 3)
    assert np.diff(rp).max() > 2000
    path = tmp_path / "powerlaw.mtx"
    rows = np.repeat(np.arange(M), np.diff(rp))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write("%d %d %d\n" % (M, K, len(ci)))
        for r, c, x in zip(rows, ci, v):
            f.write("%d %d %.9g\n" % (r + 1, c + 1, x))
    r = subprocess.run([sx.api.CLI_PATH, str(path), "16", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Success!" in r.stdout and "num_mismatch = 0" in r.stdout, r.stdout[-600:]


@pytest.mark.parametrize("kernel", [0, 1, 2, 3])
@pytest.mark.parametrize("N", [8, 16, 40, 96])
def test_exact_chains_keep_strict_order_bit_identical(engine, oracle, kernel, N):
    """Round 3: the engine's DEFAULT is strict CSR order for every row (split_rows = 0).  Rows longer than
    max(1024, nnz / 16384) are then summed as exact chains (chain_fused: producer wavefronts form all rounded products of the
    row into an LDS ring, one lane per output column adds them in order) -- bit-identical to cpu_spmm_CSR like everything else,
    at ~0.2 ns per entry instead of ~80 ns through the piece kernel.  exact_chain = 0 keeps those rows on the piece path:
    same bits."""
    import torch
    from sextans_amd import api
    M = K = 20000
    rp, ci, v = api.gen_powerlaw_host(M, K, 3, 120, 15000, 11)
    lens = np.diff(rp)
    Tc = max(1024, int(rp[-1]) // 16384)
    assert (lens > Tc).sum() >= 3
    rs = np.random.RandomState(N)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    for chain in (1, 0):
        _defaults(engine, kernel=kernel, split_rows=0)
        engine.set_option("exact_chain", chain)
        engine.set_matrix_csr(M, K, rp, ci, v)
        assert engine.get_stat("exact_chain_rows") == ((lens > Tc).sum() if chain else 0)
        assert engine.get_stat("piece_path_rows") == (lens > max(32, 2 * (int(rp[-1]) // int((lens > 0).sum())))).sum()   # (2 x the mean of the non-empty rows)
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out, rp_time=2)          # (hipGraph replay: the fork/join onto the side stream is captured)
        assert len(engine.reassociated_rows()) == 0 and engine.last_kernel().endswith("+hub_pieces")
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (kernel, N, chain)
    # row ranges (multi-GPU chunks): each call sums the chain rows of its range
    engine.set_option("exact_chain", 1)
    st = torch.cuda.current_stream().cuda_stream
    dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
    got = torch.full((M * N,), float("nan"), device="cuda")
    cuts = [0, 3000, 3001, 12000, M]
    for i in range(4):
        c0, c1 = cuts[i], cuts[i + 1]
        slab = torch.full(((c1 - c0) * N,), float("nan"), device="cuda")
        engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * c0, M, slab.data_ptr(), c1 - c0, c0, c1,
                                reuse_b_panels=i > 0, stream=st)
        got.view(N, M)[:, c0:c1] = slab.view(N, c1 - c0)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    engine.set_option("exact_chain", 1)


def test_power_law_1m_rows_strict_order_default(engine, sx):
    """The same 1M-row power-law matrix with NO option set: every row in strict order, the 399 302-entry row included.
    Round 2 needed 31.6 ms for that (one row group walking the row); the exact chains bring it to a small multiple of
    the uniform matrix, bit-identical (sampled hub rows against the oracle in test_exact_chains...)."""
    import torch
    from sextans_amd import api
    M = K = 1_000_000
    N = 16
    st = torch.cuda.current_stream().cuda_stream
    B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
    api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st)
    api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
    pl = api.gen_powerlaw_device(0, M, K, 6, 120, 400_000, 7)
    try:
        with api.Engine(0) as e:                       # a fresh engine: every option at its default
            e.set_matrix_csr_device(M, K, pl[3], *pl[:3])
            f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                f()
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 10
            chains, reass = int(e.get_stat("exact_chain_rows")), int(e.get_stat("reassociated_rows"))
            strict = Cout.clone()
            e.set_option("exact_chain", 0)             # the same rows through the piece kernel: same bits, ~30 ms
            f(); torch.cuda.synchronize()
            same = bool(torch.equal(strict, Cout))
        print(f"power-law {pl[3]} nnz, strict order: {t * 1e3:.3f} ms ({chains} exact chains, {reass} re-associated rows)")
        assert chains > 100 and reass == 0 and same
        assert t < 2.5e-3, t
    finally:
        for q in pl[:3]:
            api.device_free(0, q)


def test_emptied_rows_do_not_make_the_others_long(engine, oracle):
    """A mesh matrix with half of its rows emptied (eliminated unknowns): the automatic bucketing threshold is twice the mean length of
    the NON-EMPTY rows -- with the mean over all rows every remaining row counted as long and went to the piece kernel (2M-row FEM:
    913 us per step instead of 351).  And rows above the threshold that are a quarter of the matrix are no outliers: a bimodal matrix
    (27- and 270-entry rows) is not bucketed either.  Bit-identical, piece_path_rows == 0."""
    from sextans_amd import api
    rs = np.random.RandomState(5)
    rp, ci, v = api.gen_fem3d_host(14, 13, 12, 3, 7)
    rp, ci, v = np.array(rp), np.array(ci), np.array(v)
    M = K = 14 * 13 * 12 * 3
    lens = np.diff(rp)
    keep_row = rs.rand(M) < 0.5
    keep = np.repeat(keep_row, lens)
    lens2 = np.where(keep_row, lens, 0)
    half = (np.concatenate([[0], np.cumsum(lens2)]).astype(np.int32), ci[keep], v[keep])
    # bimodal: 40 % of the rows get 10 x their entries (extra random columns)
    import scipy.sparse as sp
    A = sp.csr_matrix((v, ci, rp), shape=(M, K))
    extra_rows = np.flatnonzero(rs.rand(M) < 0.4)
    E = sp.csr_matrix((rs.uniform(-1, 1, len(extra_rows) * 200).astype(np.float32), (np.repeat(extra_rows, 200), rs.randint(0, K, len(extra_rows) * 200))), shape=(M, K))
    E.sum_duplicates()
    Bm = (A + E).tocsr(); Bm.sort_indices()
    bimodal = (Bm.indptr.astype(np.int32), Bm.indices.astype(np.int32), Bm.data.astype(np.float32))
    N = 16
    try:
        for name, (rp2, ci2, v2) in (("half emptied", half), ("bimodal", bimodal)):
            _defaults(engine)
            engine.set_matrix_csr(M, K, rp2, ci2, v2)
            assert int(engine.get_stat("piece_path_rows")) == 0, (name, engine.get_stat("piece_path_rows"), engine.get_stat("bucket_threshold"))
            B = rs.uniform(-1, 1, K * N).astype(np.float32)
            C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
            want = C0.copy()
            oracle.spmm(M, N, K, ALPHA, rp2, ci2, v2, B, BETA, want)
            out = C0.copy()
            engine.spmm(N, ALPHA, B, BETA, out)
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (name, engine.last_kernel())
    finally:
        _defaults(engine)
