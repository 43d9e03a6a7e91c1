"""Clustered-order plan (csrc/row_cluster.hip, engine option row_cluster): for matrices with grid-stencil structure the LDS-panel plan
visits the rows brick by brick (a run of <= 16 rows of a grid line x 2 lines x 2 planes), so that a 64-row block needs ~30 % fewer B
rows in its panel.  Rows are independent: every sum keeps its order, results stay bit-identical to cpu_spmm_CSR.  Row-range calls,
8-column tiles and column-major staging keep the natural-order plan."""
import numpy as np
import pytest

from util import ALPHA, BETA, random_csr

pytestmark = pytest.mark.gpu

OPTS = dict(lanes_per_row=0, kernel=0, fuse_b=0, panel_v2=-1, cols_per_lane=0, tiles_per_wg=0, split_rows=0, bucket_rows=-1,
            panel_min_reuse_x100=200, panel_min_reuse_wide_x100=150, row_cluster=-1, cluster_group=6, cluster_shape=0, row_sets=2)


def _set(engine, **kw):
    d = dict(OPTS)
    d.update(kw)
    for k, v in d.items():
        engine.set_option(k, v)


def _cases():
    from sextans_amd import api
    yield "fem 3 dof", api.gen_fem3d_host(16, 15, 14, 3, 7), 16 * 15 * 14 * 3, (48, 720)
    yield "fem 1 dof", api.gen_fem3d_host(30, 28, 26, 1, 7), 30 * 28 * 26, (30, 840)
    yield "2-D 9-point, 2 dof", api.gen_stencil2d_host(120, 110, 9, 2, 3), 120 * 110 * 2, (240, 0)
    yield "2-D 5-point", api.gen_stencil2d_host(130, 90, 5, 1, 3), 130 * 90, (130, 0)


@pytest.mark.parametrize("N", [16, 40, 128])
def test_clustered_plan_is_bit_identical_and_smaller(engine, oracle, N):
    for name, (rp, ci, v), M, strides in _cases():
        K = M
        rs = np.random.RandomState(N)
        B = rs.uniform(-1, 1, K * N).astype(np.float32)
        C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        try:
            for rc in (1, -1, 0):
                _set(engine, row_cluster=rc)
                engine.set_matrix_csr(M, K, rp, ci, v)
                for rp_time in (1, 4):                      # (4: the hipGraph replay of the repeat loop)
                    out = C0.copy()
                    engine.spmm(N, ALPHA, B, BETA, out, rp_time=rp_time)
                    assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (name, N, rc, rp_time, engine.last_kernel())
                if engine.last_kernel().startswith("spmm_csr_panel"):      # (the 5-point stencil at N = 16 stays on the gather kernel)
                    state = int(engine.get_stat("row_cluster"))
                    if rc == 0:
                        assert state == -1
                    else:
                        assert (int(engine.get_stat("grid_stride_line")), int(engine.get_stat("grid_stride_plane"))) == strides, name
                        nat, clu = engine.get_stat("panel_rows_natural"), engine.get_stat("panel_rows_clustered")
                        assert clu < 0.85 * nat, (name, nat, clu)
                        assert state == 1
        finally:
            _set(engine)


def test_no_structure_no_clustering(engine, oracle, sx):
    """Random columns inside a band (reuse, but no grid strides; its natural-order blocks are full) and a KKT system (different
    stencils per section, border rows on the exact-chain path): under the automatic setting neither the grid detector nor the graph
    clustering (tests/test_graph_cluster_gpu.py) takes them and the natural-order plan runs; forced (row_cluster = 1) they may get
    a graph-clustered plan -- same bits either way."""
    from sextans_amd import api
    rs = np.random.RandomState(3)
    cases = []
    M = 6000
    rp, ci, v = random_csr(rs, M, M, 12)
    band = np.clip(np.repeat(np.arange(M), np.diff(rp)) + rs.randint(-40, 41, size=len(ci)), 0, M - 1).astype(np.int32)
    order = np.lexsort((band, np.repeat(np.arange(M), np.diff(rp))))
    cases.append(("banded random", rp, band[order], v[order], M))
    n = 3000
    krp, kci, kv = api.gen_kkt_host(n, 2, 3)
    cases.append(("kkt", krp, kci, kv, api.kkt_rows(n, 2)))
    try:
        for name, rp, ci, v, M in cases:
            for rc in (-1, 1):
                _set(engine, row_cluster=rc, kernel=2)
                engine.set_matrix_csr(M, M, rp, ci, v)
                N = 32
                B = rs.uniform(-1, 1, M * N).astype(np.float32)
                C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
                want = C0.copy()
                oracle.spmm(M, N, M, ALPHA, rp, ci, v, B, BETA, want)
                out = C0.copy()
                engine.spmm(N, ALPHA, B, BETA, out)
                assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (name, rc)
                state = int(engine.get_stat("row_cluster"))
                assert state != 1, name                                   # never the grid bricks
                if rc == -1:
                    assert state == -1, (name, rc, engine.get_stat("cluster_decline"))
    finally:
        _set(engine)


def test_row_ranges_and_long_rows_with_a_clustered_plan(engine, oracle):
    """Whole-matrix calls use the clustered plan, row-range calls (the chunks of the multi-GPU pipeline) the natural one; rows that
    leave the main matrix (long rows -> piece path / exact chains) are skipped through the slot -> row table.  All bit-identical."""
    import torch
    from sextans_amd import api
    rp, ci, v = api.gen_fem3d_host(16, 15, 14, 3, 7)
    M = K = 16 * 15 * 14 * 3
    # make a few rows long (they leave the main matrix): append random extra entries to rows 100, 2000, 9000
    rs = np.random.RandomState(5)
    rows = np.repeat(np.arange(M), np.diff(rp))
    extra_r, extra_c, extra_v = [], [], []
    for r, n in ((100, 700), (2000, 1500), (9000, 2600)):
        cols = np.setdiff1d(rs.choice(K, size=n, replace=False), ci[rp[r]:rp[r + 1]])
        extra_r.append(np.full(len(cols), r)); extra_c.append(cols); extra_v.append(rs.uniform(-1, 1, len(cols)))
    rows = np.concatenate([rows] + extra_r); cols = np.concatenate([ci] + extra_c); vals = np.concatenate([v] + extra_v).astype(np.float32)
    order = np.lexsort((cols, rows))
    rows, ci2, v2 = rows[order], cols[order].astype(np.int32), vals[order]
    rp2 = np.zeros(M + 1, dtype=np.int32); np.add.at(rp2, rows + 1, 1); rp2 = np.cumsum(rp2).astype(np.int32)
    N = 32
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp2, ci2, v2, B, BETA, want)
    try:
        _set(engine, row_cluster=1)
        engine.set_matrix_csr(M, K, rp2, ci2, v2)
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        assert engine.get_stat("piece_path_rows") >= 3 and int(engine.get_stat("row_cluster")) == 1
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
        st = torch.cuda.current_stream().cuda_stream
        dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
        got = torch.full((M * N,), float("nan"), device="cuda")
        cuts = [0, engine.align_row(N, 3000), engine.align_row(N, 7000), M]
        for i in range(3):
            c0, c1 = cuts[i], cuts[i + 1]
            slab = torch.full(((c1 - c0) * N,), float("nan"), device="cuda")
            engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * c0, M, slab.data_ptr(), c1 - c0, c0, c1,
                                    reuse_b_panels=i > 0, stream=st)
            got.view(N, M)[:, c0:c1] = slab.view(N, c1 - c0)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
        out = C0.copy()                                   # and a whole-matrix call again
        engine.spmm(N, ALPHA, B, BETA, out)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    finally:
        _set(engine)


def test_released_plan_stream_is_rebuilt_for_row_ranges(engine, oracle):
    """A matrix too large for column-major staging (K x 16 floats > 16 MiB): once the clustered plan serves the whole-matrix calls
    the natural-order plan hands its packed stream back (stat "device_bytes" drops by 4.7 - 6 bytes per non-zero); the first row-range
    call rebuilds it -- same bytes, same results."""
    import torch
    from sextans_amd import api
    nx = 45
    M = K = nx * nx * nx * 3
    rp, ci, v = api.gen_fem3d_host(nx, nx, nx, 3, 7)
    rs = np.random.RandomState(8)
    N = 16
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    try:
        _set(engine, row_cluster=0)
        engine.set_matrix_csr(M, K, rp, ci, v)
        out = C0.copy(); engine.spmm(N, ALPHA, B, BETA, out)
        natural_only = engine.get_stat("device_bytes")
        _set(engine, row_cluster=-1)
        engine.set_matrix_csr(M, K, rp, ci, v)
        out = C0.copy(); engine.spmm(N, ALPHA, B, BETA, out)
        assert int(engine.get_stat("row_cluster")) == 1
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
        with_cluster = engine.get_stat("device_bytes")
        assert with_cluster < natural_only + 2.0 * len(ci), (natural_only, with_cluster)     # not two 6-byte streams side by side
        st = torch.cuda.current_stream().cuda_stream
        dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda()
        got = torch.full((M * N,), float("nan"), device="cuda")
        cuts = [0, engine.align_row(N, M // 2), M]
        for i in range(2):
            c0, c1 = cuts[i], cuts[i + 1]
            slab = torch.full(((c1 - c0) * N,), float("nan"), device="cuda")
            engine.spmm_device_rows(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr() + 4 * c0, M, slab.data_ptr(), c1 - c0, c0, c1,
                                    reuse_b_panels=i > 0, stream=st)
            assert engine.last_kernel().startswith("spmm_csr_panel")
            got.view(N, M)[:, c0:c1] = slab.view(N, c1 - c0)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert engine.get_stat("device_bytes") > with_cluster + 4.0 * len(ci)                 # the stream is back (4 B values + shared indices)
    finally:
        _set(engine)


@pytest.mark.parametrize("N", [16, 48, 128])
def test_two_row_sets_per_block(engine, oracle, N):
    """Short-row grid matrices (every row <= 32 entries): the clustered plan uses 128-row bricks as blocks of TWO 64-slot row sets on one
    dictionary / panel (spmm_panel_v2.h: SETS; option row_sets: 2 = 3-D grids, 3 = 2-D grids too, 1 = never).  Bit-identical to
    cpu_spmm_CSR whatever the setting, fewer panel rows than 64-row bricks, partial bricks at the grid edges (30 x 28 x 26 is no
    multiple of 16 x 4 x 2) and rows of 0 entries included; longer rows (3 dof: 81 entries) keep 64-row blocks."""
    from sextans_amd import api
    cases = [("27-point 1 dof", api.gen_fem3d_host(30, 28, 26, 1, 7), 30 * 28 * 26, True),
             ("2-D 9-point 2 dof", api.gen_stencil2d_host(120, 110, 9, 2, 3), 120 * 110 * 2, False),
             ("27-point 3 dof", api.gen_fem3d_host(16, 15, 14, 3, 7), 16 * 15 * 14 * 3, None)]
    rs = np.random.RandomState(N)
    for name, (rp, ci, v), M, auto in cases:
        rp, ci, v = np.array(rp), np.array(ci), np.array(v)
        if auto is not None:                       # a few empty rows: slots of 0 entries inside a set
            keep = np.ones(len(ci), bool)
            for r in (5, 77, M - 3):
                keep[rp[r]:rp[r + 1]] = False
            lens = np.diff(rp); lens[[5, 77, M - 3]] = 0
            ci, v = ci[keep], v[keep]
            rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        K = M
        B = rs.uniform(-1, 1, K * N).astype(np.float32)
        C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        try:
            rows = {}
            for sets in (1, 2, 3):
                _set(engine, row_cluster=1, row_sets=sets)
                engine.set_matrix_csr(M, K, rp, ci, v)
                for rp_time in (1, 3):
                    out = C0.copy()
                    engine.spmm(N, ALPHA, B, BETA, out, rp_time=rp_time)
                    assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (name, N, sets, rp_time, engine.last_kernel())
                assert engine.last_kernel() == "spmm_csr_panel_v2" and int(engine.get_stat("row_cluster")) == 1, (name, sets)
                used = int(engine.get_stat("row_sets"))
                assert used == (1 if auto is None or sets == 1 else 2 if (sets == 3 or auto) else 1), (name, sets, used)
                rows[sets] = engine.get_stat("panel_rows_clustered")
            if auto is not None:
                assert rows[3] < 0.9 * rows[1], (name, rows)
        finally:
            _set(engine)
