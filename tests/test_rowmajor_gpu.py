"""Row-major operand entry point sextans_spmm_device_rm (round 5): B row-major K x N, C row-major M x N -- the layouts the kernels
want, so no layout pass runs on the LDS-panel paths.  The reference lays B and C out for its kernel on the host, outside the timed
call (sextans-host.cpp:150-195, 264-270).  Same per-row order and rounding: BIT-IDENTICAL to cpu_spmm_CSR (sparse_helper.h:262-290)
on the natural-order, grid-brick and graph-clustered plans, with padded leading dimensions, in place, N = 8 / 16 / 24 / 128, and on
the row-group gather kernel, and on everything that falls back to column-major copies (mixed plans, long rows, unaligned operands)."""
import numpy as np
import pytest

from util import ALPHA, BETA, random_csr

pytestmark = pytest.mark.gpu


def _want(oracle, M, K, N, rp, ci, v, B, C0, alpha=ALPHA, beta=BETA):
    """B (K, N), C0 (M, N) row-major numpy -> oracle result as (M, N)"""
    w = np.ascontiguousarray(C0.T).reshape(-1).copy()
    oracle.spmm(M, N, K, alpha, rp, ci, v, np.ascontiguousarray(B.T).reshape(-1), beta, w)
    return np.ascontiguousarray(w.reshape(N, M).T)


def _run(e, M, K, N, B, C0, ldb=None, ldc_in=None, ldc=None, inplace=False, alpha=ALPHA, beta=BETA, offset=0):
    import torch
    ldb, ldc_in, ldc = ldb or N, ldc_in or N, ldc or N
    tb = torch.full((K * ldb + offset,), 7.0, device="cuda"); tci = torch.full((M * ldc_in + offset,), 9.0, device="cuda")
    tb[offset:].view(K, ldb)[:, :N] = torch.from_numpy(B).cuda(); tci[offset:].view(M, ldc_in)[:, :N] = torch.from_numpy(C0).cuda()
    if inplace:
        tco, ldc = tci, ldc_in
    else:
        tco = torch.full((M * ldc + offset,), -5.0, device="cuda")
    e.spmm_device_rm(N, float(alpha), tb.data_ptr() + 4 * offset, ldb, float(beta), tci.data_ptr() + 4 * offset, ldc_in, tco.data_ptr() + 4 * offset, ldc,
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    full = tco[offset:].view(M, ldc).cpu().numpy()
    if not inplace and ldc > N:
        assert np.all(full[:, N:] == -5.0), "columns beyond N were written"
    if ldb > N:
        assert np.all(tb[offset:].view(K, ldb)[:, N:].cpu().numpy() == 7.0)
    return np.ascontiguousarray(full[:, :N])


def _matrices():
    from sextans_amd import api, meshgen
    rp, ci, v = api.gen_fem3d_host(9, 8, 7, 3, 5)                       # 1 512 rows: natural-order plan (no clustering below 4 096 rows)
    yield "fem small, natural plan", (rp, ci, v, 1512, 1512), ("spmm_csr_panel_v2_rowmajor",), {}
    rp, ci, v = api.gen_fem3d_host(20, 19, 18, 3, 5)                    # 20 520 rows: grid bricks
    M = 20 * 19 * 18 * 3
    yield "fem, grid bricks", (rp, ci, v, M, M), ("spmm_csr_panel_v2_rowmajor",), {}
    q = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 2))
    yield "fem, random node order, graph clustering", (*q, M, M), ("spmm_csr_panel_v2_rowmajor_clustered",), {}
    yield "fem, random node order, natural panels of the clustered plan", (*q, M, M), ("spmm_csr_panel_v2_rowmajor_clustered",), {"relabel_columns": 0}
    rp1, ci1, v1 = api.gen_fem3d_host(30, 28, 26, 1, 3)                 # short rows: two row sets per block / 320-row panels
    yield "27-point 1 dof, bricks with two row sets", (rp1, ci1, v1, 30 * 28 * 26, 30 * 28 * 26), ("spmm_csr_panel_v2_rowmajor",), {}


@pytest.mark.parametrize("N", [8, 16, 24, 128])
def test_rowmajor_panel_paths_are_bit_identical(engine, oracle, N):
    for name, (rp, ci, v, M, K), kernels, opts in _matrices():
        rs = np.random.RandomState(N + M % 13)
        B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
        want = _want(oracle, M, K, N, rp, ci, v, B, C0)
        try:
            for k, val in opts.items():
                engine.set_option(k, val)
            engine.set_matrix_csr(M, K, rp, ci, v)
            for kw in ({}, {"ldb": N + 8, "ldc_in": N + 4, "ldc": N + 12}, {"inplace": True}, {"inplace": True, "ldc_in": N + 20}):
                got = _run(engine, M, K, N, B, C0, **kw)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, N, kw, engine.last_kernel())
                assert engine.last_kernel() in kernels, (name, N, engine.last_kernel(), engine.get_stat("row_cluster"), engine.get_stat("cluster_decline"))
            # the column-major entry point on the same engine afterwards: same bits (plans are shared, layouts are per call)
            cm = np.ascontiguousarray(C0.T).reshape(-1).copy()
            engine.spmm(N, ALPHA, np.ascontiguousarray(B.T).reshape(-1), BETA, cm)
            assert np.array_equal(np.ascontiguousarray(cm.reshape(N, M).T).view(np.uint32), want.view(np.uint32)), (name, N)
        finally:
            for k in opts:
                engine.set_option(k, 1)


def test_rowmajor_alpha_beta_special_values(engine, oracle):
    from sextans_amd import api
    rp, ci, v = api.gen_fem3d_host(9, 8, 7, 3, 5)
    M = K = 1512
    engine.set_matrix_csr(M, K, rp, ci, v)
    rs = np.random.RandomState(3)
    B = rs.uniform(-1, 1, (K, 16)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, 16)).astype(np.float32)
    for alpha, beta in ((0.0, 1.0), (1.0, 0.0), (-1.5, 0.25), (0.0, 0.0)):
        want = _want(oracle, M, K, 16, rp, ci, v, B, C0, np.float32(alpha), np.float32(beta))
        got = _run(engine, M, K, 16, B, C0, alpha=alpha, beta=beta)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (alpha, beta)


def test_rowmajor_lane_per_row_kernel(engine, oracle):
    """Very short rows in a local numbering (5-point stencil): spmm_csr_colwise on the row-major operands -- 16-byte accesses; from
    N = 32 on groups of T lanes per row take neighbouring 16-column tiles (T = 2 .. 8 incl. the odd group sizes 3, 5, 7: 85 / 51 / 36
    rows per workgroup) so that a wavefront's loads cover whole lines; a tile count without a divisor up to 8 (11) and the 8-column
    tail keep one lane per row and tile."""
    from sextans_amd import api
    rp, ci, v = api.gen_stencil2d_host(90, 81, 5, 1, 3)
    M = K = 7290
    engine.set_matrix_csr(M, K, rp, ci, v)
    rs = np.random.RandomState(12)
    for N in (8, 16, 24, 32, 48, 64, 80, 96, 112, 128, 136, 176, 256):
        B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
        want = _want(oracle, M, K, N, rp, ci, v, B, C0)
        for kw in ({}, {"ldb": N + 4, "ldc_in": N + 8, "ldc": N + 4}, {"inplace": True}):
            got = _run(engine, M, K, N, B, C0, **kw)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, kw, engine.last_kernel())
            assert engine.last_kernel() == "spmm_csr_colwise_rowmajor", engine.last_kernel()


def test_rowmajor_fallback_classes(engine, oracle):
    """Unaligned operands (whatever the matrix): through column-major copies."""
    from sextans_amd import api
    rs = np.random.RandomState(11)
    cases = []
    rp, ci, v = random_csr(rs, 5000, 7000, 12)                          # no reuse: row-group gather kernel, unaligned here
    cases.append(("random columns, pointers 8 bytes off", rp, ci, v, 5000, 7000, 2))
    rp, ci, v = random_csr(rs, 3000, 3000, 10, long_rows=3)             # rows on the piece path, unaligned
    cases.append(("long rows, pointers 4 bytes off", rp, ci, v, 3000, 3000, 1))
    rp, ci, v = api.gen_fem3d_host(9, 8, 7, 3, 5)
    cases.append(("fem small, pointers 4 bytes off a 16-byte boundary", rp, ci, v, 1512, 1512, 1))
    for name, rp, ci, v, M, K, off in cases:
        engine.set_matrix_csr(M, K, rp, ci, v)
        for N in (8, 24, 64):
            B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
            want = _want(oracle, M, K, N, rp, ci, v, B, C0)
            for kw in ({}, {"ldb": N + 4, "ldc_in": N + 8, "ldc": N + 4}, {"inplace": True}):
                got = _run(engine, M, K, N, B, C0, offset=off, **kw)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, N, kw, engine.last_kernel())
            assert "rowmajor" not in engine.last_kernel(), (name, engine.last_kernel())


@pytest.mark.parametrize("stage", [1, 0])
def test_rowmajor_gather_kernel(engine, oracle, stage):
    """No reuse between rows (random columns: config 4's class): the row-group gather kernel reads the caller's row-major B rows where
    the column-major form reads repacked panel rows, and writes 16 bytes of a C row per lane -- no repack, no transposes; 32-column
    tiles at N >= 32, a 16- / 8-column tail, both A-stream forms (LDS-staged / direct)."""
    rs = np.random.RandomState(12)
    M, K = 5003, 7001
    rp, ci, v = random_csr(rs, M, K, 12)
    try:
        engine.set_option("stage_a", stage)
        engine.set_matrix_csr(M, K, rp, ci, v)
        for N in (8, 16, 24, 32, 40, 56, 64, 128):
            B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
            want = _want(oracle, M, K, N, rp, ci, v, B, C0)
            for kw in ({}, {"ldb": N + 4, "ldc_in": N + 8, "ldc": N + 4}, {"inplace": True}):
                got = _run(engine, M, K, N, B, C0, **kw)
                assert engine.last_kernel() == "spmm_csr_rowgroup_rowmajor", (N, engine.last_kernel())
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, kw)
        # an empty row, a row of one entry, alpha = 0 / beta = 0 with non-finite C_in
        rp2 = rp.copy(); drop = rp2[11] - rp2[10]; rp2[11:] -= drop
        ci2 = np.concatenate([ci[:rp[10]], ci[rp[11]:]]); v2 = np.concatenate([v[:rp[10]], v[rp[11]:]])
        engine.set_matrix_csr(M, K, rp2, ci2, v2)
        N = 16
        B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
        C0[5, 3] = np.inf; C0[6, 0] = np.nan
        for a, b in ((ALPHA, 0.0), (0.0, BETA), (1.0, 1.0)):
            want = _want(oracle, M, K, N, rp2, ci2, v2, B, C0, a, b)
            got = _run(engine, M, K, N, B, C0, alpha=a, beta=b)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (a, b)
    finally:
        engine.set_option("stage_a", 1)


def test_rowmajor_reconsiders_a_declined_clustered_plan(engine, oracle):
    """The holdout class in its natural numbering: the graph-clustered plan copies ~36 % fewer B rows -- not enough for the column-major
    form (two passes over C: decline 12), enough for row-major calls, which pay nothing for it.  Column-major calls on the same
    engine keep the natural-order plan."""
    from sextans_amd import holdout
    rp, ci, v, M, K = holdout.kron_host(10)
    engine.set_matrix_csr(M, K, rp, ci, v)
    rs = np.random.RandomState(5)
    N = 16
    B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
    want = _want(oracle, M, K, N, rp, ci, v, B, C0)
    cm = np.ascontiguousarray(C0.T).reshape(-1).copy()
    engine.spmm(N, ALPHA, np.ascontiguousarray(B.T).reshape(-1), BETA, cm)
    first = (int(engine.get_stat("row_cluster")), int(engine.get_stat("cluster_decline")), engine.last_kernel())
    got = _run(engine, M, K, N, B, C0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if first[:2] == (-1, 12):
        assert engine.last_kernel() == "spmm_csr_panel_v2_rowmajor_clustered" and int(engine.get_stat("row_cluster")) == 2, (first, engine.last_kernel(), engine.get_stat("cluster_decline"))
        cm2 = np.ascontiguousarray(C0.T).reshape(-1).copy()
        engine.spmm(N, ALPHA, np.ascontiguousarray(B.T).reshape(-1), BETA, cm2)
        assert engine.last_kernel() == first[2] and np.array_equal(cm2.view(np.uint32), cm.view(np.uint32))
    assert np.array_equal(np.ascontiguousarray(cm.reshape(N, M).T).view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("N", [16, 40])
def test_rowmajor_c_beyond_4gb_takes_64bit_addresses(engine, oracle, N):
    """M * ldc * 4 bytes >= 4 GB (4M rows x 512 columns in production; here 20 520 rows with a leading dimension of 53 248 floats): the
    kernel's 32-bit byte offsets into C do not reach -- the RM == 2 instantiations (64-bit lane addresses) run instead of the detour
    through column-major copies, on every plan kind."""
    import torch
    for name, (rp, ci, v, M, K), kernels, opts in _matrices():
        ld = -(-((1 << 32) // (4 * M) + 8) // 4) * 4
        assert M * ld * 4 >= 1 << 32
        rs = np.random.RandomState(N + 3)
        B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
        want = _want(oracle, M, K, N, rp, ci, v, B, C0)
        try:
            for k, val in opts.items():
                engine.set_option(k, val)
            engine.set_matrix_csr(M, K, rp, ci, v)
            tb = torch.from_numpy(B).cuda()
            tci = torch.empty(M * ld, device="cuda"); tco = torch.empty(M * ld, device="cuda")
            tci.view(M, ld)[:, :N] = torch.from_numpy(C0).cuda(); tco.view(M, ld)[:, :N + 4] = -5.0
            engine.spmm_device_rm(N, ALPHA, tb.data_ptr(), N, BETA, tci.data_ptr(), ld, tco.data_ptr(), ld, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = tco.view(M, ld)[:, :N + 4].cpu().numpy()
            assert engine.last_kernel() in kernels, (name, engine.last_kernel())
            assert np.all(got[:, N:] == -5.0)
            assert np.array_equal(np.ascontiguousarray(got[:, :N]).view(np.uint32), want.view(np.uint32)), (name, N)
            # in place, too
            engine.spmm_device_rm(N, ALPHA, tb.data_ptr(), N, BETA, tci.data_ptr(), ld, tci.data_ptr(), ld, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(np.ascontiguousarray(tci.view(M, ld)[:, :N].cpu().numpy()).view(np.uint32), want.view(np.uint32)), (name, N)
            del tci, tco
            torch.cuda.empty_cache()
        finally:
            for k in opts:
                engine.set_option(k, 1)


def test_rowmajor_calls_allocate_no_layout_workspaces(sx, oracle):
    """An engine that only ever serves row-major calls on the native paths holds the matrix and its plan, not the B-panel / C-staging
    workspaces of the column-major entry points (K x N and M x N floats each: 8 GB apiece at 4M rows and N = 512)."""
    import torch
    from sextans_amd import api
    rp, ci, v = api.gen_fem3d_host(20, 19, 18, 3, 5)
    M = K = 20 * 19 * 18 * 3
    N = 64
    rs = np.random.RandomState(2)
    B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
    want = _want(oracle, M, K, N, rp, ci, v, B, C0)
    with api.Engine(0) as e:
        e.set_matrix_csr(M, K, rp, ci, v)
        got = _run(e, M, K, N, B, C0)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and e.last_kernel() == "spmm_csr_panel_v2_rowmajor"
        lean = e.get_stat("device_bytes")
        cm = np.ascontiguousarray(C0.T).reshape(-1).copy()
        e.spmm(N, ALPHA, np.ascontiguousarray(B.T).reshape(-1), BETA, cm)
        full = e.get_stat("device_bytes")
        assert np.array_equal(np.ascontiguousarray(cm.reshape(N, M).T).view(np.uint32), want.view(np.uint32))
        assert full - lean >= 4 * K * N, (lean, full)      # the column-major call brought (at least) the B panels
        got = _run(e, M, K, N, B, C0)                       # and the row-major call still works afterwards
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("N", [8, 16, 40, 96])
def test_rowmajor_long_rows_pieces_and_exact_chains(engine, oracle, N):
    """Rows on the long-row paths from row-major operands, without copies: the piece kernel gathers B rows ldb floats apart, the chain
    producers stage them by LDS-DMA, fold and chain consumer write C[r][n].  (i) power-law matrix on the gather kernel: bucketed rows +
    exact chains (strict order: bit-identical); (ii) FEM matrix with hub rows on the LDS-panel kernel, natural and graph-clustered plans;
    padded leading dimensions and in place."""
    from sextans_amd import api, meshgen
    cases = []
    M = K = 20000
    rp, ci, v = api.gen_powerlaw_host(M, K, 3, 120, 15000, 11)
    cases.append(("power law", rp, ci, v, M, K, "spmm_csr_rowgroup_rowmajor+long_rows", True))
    frp, fci, fv = api.gen_fem3d_host(20, 19, 12, 3, 5)
    M0 = 20 * 19 * 12 * 3
    rs = np.random.RandomState(17)
    rows = []
    for r in range(M0):
        c, x = fci[frp[r]:frp[r + 1]], fv[frp[r]:frp[r + 1]]
        if r in (7, 5000, M0 - 1):
            c = np.sort(rs.choice(M0, size=6000 + r % 100, replace=False)).astype(np.int32)
            x = rs.uniform(-1, 1, len(c)).astype(np.float32)
        rows.append((c, x))
    rp2 = np.zeros(M0 + 1, np.int32); rp2[1:] = np.cumsum([len(c) for c, _ in rows])
    ci2 = np.concatenate([c for c, _ in rows]).astype(np.int32); v2 = np.concatenate([x for _, x in rows]).astype(np.float32)
    cases.append(("fem + hub rows", rp2, ci2, v2, M0, M0, "spmm_csr_panel_v2_rowmajor+long_rows", True))
    q = meshgen.permute_symmetric(rp2, ci2, v2, M0, meshgen.node_permutation(M0 // 3, 3, 2))
    cases.append(("fem + hub rows, random node order", *q, M0, M0, ("spmm_csr_panel_v2_rowmajor_clustered+long_rows", "spmm_csr_panel_v2_rowmajor+long_rows"), False))
    for name, rp, ci, v, M, K, kernel, strict in cases:
        rs = np.random.RandomState(N + len(name))
        B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
        want = _want(oracle, M, K, N, rp, ci, v, B, C0)
        engine.set_matrix_csr(M, K, rp, ci, v)
        for kw in ({}, {"ldb": N + 4, "ldc_in": N + 8, "ldc": N + 12}, {"inplace": True}):
            got = _run(engine, M, K, N, B, C0, **kw)
            assert engine.last_kernel() == kernel or engine.last_kernel() in kernel, (name, N, engine.last_kernel())
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, N, kw)
        assert engine.get_stat("piece_path_rows") > 0 and len(engine.reassociated_rows()) == 0
        # the column-major entry point on the same engine afterwards
        cm = np.ascontiguousarray(C0.T).reshape(-1).copy()
        engine.spmm(N, ALPHA, np.ascontiguousarray(B.T).reshape(-1), BETA, cm)
        assert np.array_equal(np.ascontiguousarray(cm.reshape(N, M).T).view(np.uint32), want.view(np.uint32)), (name, N)
    # re-association opted in (split_rows = -1): hub rows cut into pieces, folded in order -- within the stated 1e-4 bound, the rest bit-identical
    name, rp, ci, v, M, K, kernel, _ = cases[0]
    try:
        engine.set_option("split_rows", -1)
        engine.set_matrix_csr(M, K, rp, ci, v)
        rs = np.random.RandomState(3)
        B = rs.uniform(-1, 1, (K, N)).astype(np.float32); C0 = rs.uniform(-1, 1, (M, N)).astype(np.float32)
        want = _want(oracle, M, K, N, rp, ci, v, B, C0)
        got = _run(engine, M, K, N, B, C0)
        hub = engine.reassociated_rows()
        assert len(hub) > 0 and "+long_rows" in engine.last_kernel()
        keep = np.ones(M, bool); keep[hub] = False
        assert np.array_equal(got[keep].view(np.uint32), want[keep].view(np.uint32))
        A = np.zeros(M)
        absB = np.abs(B).max()
        for r in hub:
            A[r] = np.abs(v[rp[r]:rp[r + 1]]).sum() * absB
        assert np.all(np.abs(got[hub].astype(np.float64) - want[hub]) <= 1e-4 * (abs(ALPHA) * A[hub][:, None] + np.abs(BETA * C0[hub])) + 1e-30)
    finally:
        engine.set_option("split_rows", 0)
