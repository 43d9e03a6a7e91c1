"""MIXED plans (some row blocks reuse B rows and have a dictionary, others -- uniformly random rows -- do not) in their SPLIT form
(round 5): the dictionary blocks on spmm_csr_panel_v2 (it walks the list of those blocks), the rows of the other blocks on the gather
kernel (it walks the groups of 128 rows that hold such a row and skips the others) -- through the column-major and the row-major entry
points, with rows on the piece path in both parts.  Bit-identical to cpu_spmm_CSR (sparse_helper.h:262-290), like the one-launch form
spmm_csr_panel<MIXED> ("split_mixed" = 0) it replaces.  The reference schedules every non-zero for its on-chip window whatever the
row's neighbours look like (sparse_helper.h:345-403); here the rows without reuse simply do not pay for a panel."""
import numpy as np
import pytest

from util import ALPHA, BETA

pytestmark = pytest.mark.gpu


def _mixed_matrix(seed=3):
    """FEM rows with runs of uniformly random rows at the front, in the middle and at the end, a few very long rows in both kinds"""
    from sextans_amd import api
    frp, fci, fv = api.gen_fem3d_host(20, 19, 18, 3, 5)
    Mf = 20 * 19 * 18 * 3
    K = Mf
    rs = np.random.RandomState(seed)
    rows = []

    def random_rows(n):
        for _ in range(n):
            ln = int(rs.poisson(30))
            c = np.sort(rs.choice(K, size=ln, replace=False)).astype(np.int32)
            rows.append((c, rs.uniform(-1, 1, ln).astype(np.float32)))

    random_rows(1500)
    for r in range(Mf):
        if r == Mf // 2:
            random_rows(2100)
        c, x = fci[frp[r]:frp[r + 1]], fv[frp[r]:frp[r + 1]]
        if r in (11, Mf - 5):                                  # hub rows inside the FEM part
            c = np.sort(rs.choice(K, size=5000, replace=False)).astype(np.int32); x = rs.uniform(-1, 1, 5000).astype(np.float32)
        rows.append((c, x))
    random_rows(900)
    c = np.sort(rs.choice(K, size=4000, replace=False)).astype(np.int32)   # ... and one among the random rows
    rows[700] = (c, rs.uniform(-1, 1, 4000).astype(np.float32))
    M = len(rows)
    rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum([len(c) for c, _ in rows])
    return rp, np.concatenate([c for c, _ in rows]).astype(np.int32), np.concatenate([x for _, x in rows]).astype(np.float32), M, K


@pytest.mark.parametrize("N", [8, 16, 24, 40, 64, 128])
def test_mixed_plan_split_form_is_bit_identical(engine, oracle, N):
    import torch
    rp, ci, v, M, K = _mixed_matrix()
    rs = np.random.RandomState(N)
    B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    st = torch.cuda.current_stream().cuda_stream
    try:
        for split in (1, 0):
            engine.set_option("split_mixed", split)
            engine.set_matrix_csr(M, K, rp, ci, v)
            out = C0.copy()
            engine.spmm(N, ALPHA, B, BETA, out, rp_time=2)
            if N >= 16:
                assert engine.get_stat("mixed_plan") == (2 if split else 1), (N, split, engine.get_stat("mixed_plan"), engine.last_kernel())
                assert engine.last_kernel().startswith("spmm_csr_panel_v2" if split else "spmm_csr_panel+"), (split, engine.last_kernel())
            assert engine.get_stat("piece_path_rows") >= 3
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (N, split, engine.last_kernel())
            # row-major operands, padded leading dimensions, in place
            Br = np.ascontiguousarray(B.reshape(N, K).T); Cr = np.ascontiguousarray(C0.reshape(N, M).T)
            for ld in (N, N + 4):
                tb = torch.zeros((K, ld), device="cuda"); tb[:, :N] = torch.from_numpy(Br).cuda()
                tc = torch.zeros((M, ld), device="cuda"); tc[:, :N] = torch.from_numpy(Cr).cuda()
                engine.spmm_device_rm(N, float(ALPHA), tb.data_ptr(), ld, float(BETA), tc.data_ptr(), ld, tc.data_ptr(), ld, st)
                torch.cuda.synchronize()
                got = np.ascontiguousarray(tc[:, :N].cpu().numpy().T).reshape(-1)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, split, ld, engine.last_kernel())
                if split and N >= 16:
                    assert engine.last_kernel() == "spmm_csr_panel_v2_rowmajor+long_rows", engine.last_kernel()
            # a row range (a chunk of the multi-GPU pipeline) keeps the one-launch form or the gather kernel: same bits
            c0, c1 = engine.align_row(N, M // 3), engine.align_row(N, 2 * M // 3)
            dB = torch.from_numpy(B).cuda(); dC = torch.from_numpy(C0).cuda()
            slab = torch.full(((c1 - c0) * N,), float("nan"), device="cuda")
            engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dC.data_ptr() + 4 * c0, M, slab.data_ptr(), c1 - c0, c0, c1, stream=st)
            torch.cuda.synchronize()
            assert np.array_equal(slab.cpu().numpy().reshape(N, c1 - c0).view(np.uint32), want.reshape(N, M)[:, c0:c1].copy().view(np.uint32)), (N, split)
    finally:
        engine.set_option("split_mixed", 1)


def test_split_form_from_a_small_share_of_blocks_with_reuse(engine, oracle, sx):
    """Whole-matrix calls take the split form from 15 % of the non-zeros in blocks with reuse on (everything else keeps the 50 %
    threshold of the LDS-panel plan): FEM rows holding ~30 % of the non-zeros of a matrix of uniformly random rows.  Same bits; with
    "split_mixed" = 0 the gather kernel runs alone."""
    import torch
    from sextans_amd import api
    frp, fci, fv = api.gen_fem3d_host(20, 19, 18, 3, 5)
    Mf = 20 * 19 * 18 * 3
    Mu = 110_000
    urp, uci, uv = api.gen_csr_host(Mu, Mf, 36.0, 4, 0, Mu)
    rp = np.concatenate([frp, frp[-1] + urp[1:]]).astype(np.int32); ci = np.concatenate([fci, uci]); v = np.concatenate([fv, uv])
    M, K, N = Mf + Mu, Mf, 16
    share = float(frp[-1]) / float(rp[-1])
    assert 0.2 < share < 0.45
    rs = np.random.RandomState(8)
    B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    st = torch.cuda.current_stream().cuda_stream
    Br = torch.from_numpy(np.ascontiguousarray(B.reshape(N, K).T)).cuda(); Cr = torch.from_numpy(np.ascontiguousarray(C0.reshape(N, M).T)).cuda()
    try:
        for split, cm_kernel, rm_kernel in ((1, "spmm_csr_panel_v2", "spmm_csr_panel_v2_rowmajor"), (0, "spmm_csr_rowgroup", "spmm_csr_rowgroup_rowmajor")):
            engine.set_option("split_mixed", split)
            engine.set_matrix_csr(M, K, rp, ci, v)
            out = C0.copy()
            engine.spmm(N, ALPHA, B, BETA, out)
            assert engine.last_kernel() == cm_kernel, (split, engine.last_kernel(), engine.get_stat("panel_fraction"))
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), split
            ro = torch.empty_like(Cr)
            engine.spmm_device_rm(N, float(ALPHA), Br.data_ptr(), N, float(BETA), Cr.data_ptr(), N, ro.data_ptr(), N, st)
            torch.cuda.synchronize()
            assert engine.last_kernel() == rm_kernel, (split, engine.last_kernel())
            assert np.array_equal(np.ascontiguousarray(ro.cpu().numpy().T).reshape(-1).view(np.uint32), want.view(np.uint32)), split
    finally:
        engine.set_option("split_mixed", 1)
