"""Graph clustering of the rows (csrc/graph_cluster.hip) and the REORDERED form of the LDS-panel SpMM (csrc/reorder_kernels.h,
spmm_csr_panel_v2<..., CROW>): matrices whose graph has locality but whose numbering does not -- a 3-dof FEM matrix under a random
node permutation, the same under reverse Cuthill-McKee, an unstructured jittered-point mesh -- are aggregated over the matrix graph
on the device, their columns relabelled, B repacked into permuted panels and C staged block-major.  The rows of a matrix are
independent and every row keeps its CSR entry order, so the result stays BIT-IDENTICAL to cpu_spmm_CSR (sparse_helper.h:262-290).
What the reference does for the same purpose: generate_edge_list_for_all_PEs schedules any matrix for its on-chip B window
(sparse_helper.h:345-403)."""
import numpy as np
import pytest

from util import ALPHA, BETA, random_csr

pytestmark = pytest.mark.gpu

OPTS = dict(lanes_per_row=0, kernel=0, fuse_b=1, panel_v2=-1, cols_per_lane=0, tiles_per_wg=0, split_rows=0, bucket_rows=-1,
            panel_min_reuse_x100=200, panel_min_reuse_wide_x100=150, row_cluster=-1, exact=1)


def _set(engine, **kw):
    d = dict(OPTS)
    d.update(kw)
    for k, v in d.items():
        engine.set_option(k, v)


def _fem(nx, ny, nz, dof, seed=7):
    from sextans_amd import api
    rp, ci, v = api.gen_fem3d_host(nx, ny, nz, dof, seed)
    return rp, ci, v, nx * ny * nz * dof


def _classes():
    """name, (rp, ci, v, M), the state "row_cluster" must report under the automatic setting"""
    from sextans_amd import meshgen
    rp, ci, v, M = _fem(18, 17, 16, 3)
    yield "fem 3 dof, random node order", (*meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 11)), M), 2
    yield "fem 3 dof, RCM", (*meshgen.permute_symmetric(rp, ci, v, M, meshgen.rcm_node_permutation(rp, ci, M, 3)), M), None
    rp1, ci1, v1, M1 = _fem(30, 28, 26, 1)
    yield "fem 1 dof, random node order", (*meshgen.permute_symmetric(rp1, ci1, v1, M1, meshgen.node_permutation(M1, 1, 5)), M1), 2
    yield "jittered mesh, 3 dof, random order", meshgen.jittered_mesh3d(18, 16, 15, 3, numbering="random", dof=3), 2
    yield "jittered mesh, 1 dof, sweep order", meshgen.jittered_mesh3d(28, 26, 24, 4, numbering="sweep"), None


def _operands(rs, M, K, N):
    return rs.uniform(-1, 1, K * N).astype(np.float32), rs.uniform(-1, 1, M * N).astype(np.float32)


@pytest.mark.parametrize("N", [16, 24, 128])
def test_reordered_form_is_bit_identical(engine, oracle, N):
    for name, (rp, ci, v, M), auto_state in _classes():
        rs = np.random.RandomState(N + M % 97)
        B, C0 = _operands(rs, M, M, N)
        want = C0.copy()
        oracle.spmm(M, N, M, ALPHA, rp, ci, v, B, BETA, want)
        try:
            for rc in (2, -1, 0):
                _set(engine, row_cluster=rc)
                engine.set_matrix_csr(M, M, rp, ci, v)
                for rp_time in (1, 4):                      # (4: the hipGraph replay of the repeat loop, B panels reused)
                    out = C0.copy()
                    engine.spmm(N, ALPHA, B, BETA, out, rp_time=rp_time)
                    assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (name, N, rc, rp_time, engine.last_kernel())
                state = int(engine.get_stat("row_cluster"))
                if rc == 2:
                    assert state == 2 and engine.last_kernel() == "spmm_csr_panel_v2_reordered", (name, state, engine.last_kernel())
                    assert engine.get_stat("panel_rows_clustered") > 0
                elif rc == 0:
                    assert state == -1 and "reordered" not in engine.last_kernel()
                elif auto_state is not None:
                    assert state == auto_state, (name, state, engine.get_stat("cluster_shared_fraction"), engine.get_stat("cluster_decline"),
                                                 engine.get_stat("panel_blocks"), engine.get_stat("panel_rows_natural"), engine.get_stat("panel_rows_clustered"))
        finally:
            _set(engine)


def test_clustering_quality(engine):
    """What the clustering is for: far fewer B rows copied into LDS, full row blocks.  (Natural order of the randomly renumbered 3-dof
    matrix: ~22 rows fill the 576-row panel; clustered: >= 48 rows per block and < 45 % of the panel rows.)"""
    from sextans_amd import meshgen
    rp, ci, v, M = _fem(20, 20, 20, 3)
    prp, pci, pv = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 1))
    try:
        _set(engine, row_cluster=-1)
        engine.set_matrix_csr(M, M, prp, pci, pv)
        N = 16
        B = np.ones(M * N, np.float32); C = np.zeros(M * N, np.float32)
        engine.spmm(N, ALPHA, B, BETA, C)
        assert int(engine.get_stat("row_cluster")) == 2
        nat, clu = engine.get_stat("panel_rows_natural"), engine.get_stat("panel_rows_clustered")
        blocks = engine.get_stat("panel_blocks_clustered")
        assert clu < 0.45 * nat, (nat, clu)
        assert M / blocks >= 48, (M, blocks)
        assert engine.get_stat("cluster_shared_fraction") > 0.3
        assert engine.get_stat("device_bytes") > 8 * len(pci)
    finally:
        _set(engine)


def test_declined_inputs(engine, oracle):
    """Random columns (nothing to find: the sampled pre-test says so), rectangular matrices, matrices with rows on the long-row
    path: the graph clustering declines, also when forced, and the natural-order kernels run -- same bits."""
    rs = np.random.RandomState(9)
    cases = []
    M = 8192
    rp, ci, v = random_csr(rs, M, M, 24)
    cases.append(("uniform random", M, M, rp, ci, v, -1))
    rp, ci, v = random_csr(rs, M, M + 512, 24)
    cases.append(("rectangular", M, M + 512, rp, ci, v, 2))
    rp, ci, v = random_csr(rs, M, M, 40, long_rows=3)        # 1600-entry rows: exact chains
    cases.append(("long rows", M, M, rp, ci, v, 2))
    try:
        for name, M, K, rp, ci, v, rc in cases:
            _set(engine, row_cluster=rc)
            engine.set_matrix_csr(M, K, rp, ci, v)
            N = 32
            B, C0 = _operands(rs, M, K, N)
            want = C0.copy()
            oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
            out = C0.copy()
            engine.spmm(N, ALPHA, B, BETA, out)
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), name
            assert int(engine.get_stat("row_cluster")) == -1, name
    finally:
        _set(engine)


def test_device_calls_alias_ranges_and_options(engine, oracle):
    """Device-resident calls on the reordered form: C_in == C_out, leading dimensions, alpha / beta special values, exact = 0 within
    the stated bound; row-range calls in between keep the natural-order forms (and never reuse the permuted B panels)."""
    import torch
    from sextans_amd import meshgen
    rp, ci, v, M = _fem(17, 16, 15, 3)
    rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 2))
    rs = np.random.RandomState(1)
    N = 40
    ldb, ldc = M + 24, M + 8
    Bf = rs.uniform(-1, 1, ldb * N).astype(np.float32)
    Cf = rs.uniform(-1, 1, ldc * N).astype(np.float32)
    B = np.ascontiguousarray(Bf.reshape(N, ldb)[:, :M]).reshape(-1)
    C0 = np.ascontiguousarray(Cf.reshape(N, ldc)[:, :M]).reshape(-1)
    st = torch.cuda.current_stream().cuda_stream
    try:
        _set(engine, row_cluster=2)
        engine.set_matrix_csr(M, M, rp, ci, v)
        dB = torch.from_numpy(Bf).cuda()
        for alpha, beta in ((ALPHA, BETA), (np.float32(1), np.float32(0)), (np.float32(0), np.float32(1)), (np.float32(-2.5), np.float32(1))):
            want = C0.copy()
            oracle.spmm(M, N, M, alpha, rp, ci, v, B, beta, want)
            dC = torch.from_numpy(Cf).cuda()
            engine.spmm_device(N, float(alpha), dB.data_ptr(), ldb, float(beta), dC.data_ptr(), dC.data_ptr(), ldc, st)   # in place
            torch.cuda.synchronize()
            got = dC.cpu().numpy().reshape(N, ldc)
            assert engine.last_kernel() == "spmm_csr_panel_v2_reordered"
            assert np.array_equal(np.ascontiguousarray(got[:, :M]).reshape(-1).view(np.uint32), want.view(np.uint32)), (alpha, beta)
            assert np.array_equal(got[:, M:], Cf.reshape(N, ldc)[:, M:])          # the padding rows of C are untouched
        # row ranges between whole-matrix calls
        want = C0.copy()
        oracle.spmm(M, N, M, ALPHA, rp, ci, v, B, BETA, want)
        dCin = torch.from_numpy(C0).cuda()
        dBc = torch.from_numpy(B).cuda()
        got = torch.full((M * N,), float("nan"), device="cuda")
        cuts = [0, engine.align_row(N, M // 3), engine.align_row(N, 2 * M // 3), M]
        for i in range(3):
            c0, c1 = cuts[i], cuts[i + 1]
            slab = torch.full(((c1 - c0) * N,), float("nan"), device="cuda")
            engine.spmm_device_rows(N, ALPHA, dBc.data_ptr(), M, BETA, dCin.data_ptr() + 4 * c0, M, slab.data_ptr(), c1 - c0, c0, c1,
                                    reuse_b_panels=i > 0, stream=st)
            assert "reordered" not in engine.last_kernel()
            got.view(N, M)[:, c0:c1] = slab.view(N, c1 - c0)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        assert engine.last_kernel() == "spmm_csr_panel_v2_reordered"
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
        # natural B panels under the graph-clustered plan (option relabel_columns = 0: measured a wash): same bits
        _set(engine, row_cluster=2)
        engine.set_option("relabel_columns", 0)
        engine.set_matrix_csr(M, M, rp, ci, v)
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        engine.set_option("relabel_columns", 1)
        assert engine.last_kernel() == "spmm_csr_panel_v2_reordered"
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
        engine.set_matrix_csr(M, M, rp, ci, v)
        # exact = 0 (FMA): the stated tolerance |d| <= 1e-4 (|alpha| sum |a b| + |beta c|)
        _set(engine, row_cluster=2, exact=0)
        out = C0.copy()
        engine.spmm(N, ALPHA, B, BETA, out)
        import scipy.sparse as sp
        A = sp.csr_matrix((np.abs(v), ci, rp), shape=(M, M))
        bound = 1e-4 * (abs(ALPHA) * (A @ np.abs(B.reshape(N, M).T)).T.reshape(-1) + np.abs(BETA * C0))
        assert np.all(np.abs(out.astype(np.float64) - want) <= bound + 1e-30)
    finally:
        _set(engine)


def test_device_permutation_tool_matches_the_host_one():
    """sextans_csr_permute_symmetric_device (what tools/ use to renumber the 4M-row matrices in HBM) == meshgen.permute_symmetric."""
    import ctypes
    from sextans_amd import api, meshgen
    nx, ny, nz, dof = 13, 12, 11, 3
    M = nx * ny * nz * dof
    rp, ci, v = api.gen_fem3d_host(nx, ny, nz, dof, 5)
    perm = meshgen.node_permutation(M // dof, dof, 4)
    want = meshgen.permute_symmetric(rp, ci, v, M, perm)
    d = api.gen_fem3d_device(0, nx, ny, nz, dof, 5)
    p = api.permute_symmetric_device(0, M, d[3], *d[:3], perm)
    import torch
    nnz = d[3]

    def fetch(ptr, n, dt):   # (through the library's own HIP runtime: no second dlopen of libamdhip64 by another name)
        out = np.empty(n, np.int32 if dt == torch.int32 else np.float32)
        api.device_copy(0, out.ctypes.data, ptr, out.nbytes, api.COPY_D2H)
        return out
    got = (fetch(p[0], M + 1, torch.int32), fetch(p[1], nnz, torch.int32), fetch(p[2], nnz, torch.float32))
    for a, b in zip(got, want):
        assert np.array_equal(a.view(np.uint32), np.asarray(b).view(np.uint32))
    for q in list(d[:3]) + list(p):
        api.device_free(0, q)


@pytest.mark.parametrize("N", [32, 40, 128])
def test_tile_group_pipelining_is_bit_identical(engine, oracle, N):
    """Option pipeline_tiles = 1 (off by default: measured slower): the layout passes of the second tile group run on
    the engine's side stream under the first group's kernel -- natural-order, grid-brick and reordered forms, eager and inside the
    hipGraph of the repeat loop."""
    from sextans_amd import meshgen
    rp, ci, v, M = _fem(16, 15, 14, 3)
    prp, pci, pv = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 3))
    rs = np.random.RandomState(N)
    try:
        for name, (a, b, c), rc in (("natural", (rp, ci, v), 0), ("bricks", (rp, ci, v), 1), ("reordered", (prp, pci, pv), 2)):
            B, C0 = _operands(rs, M, M, N)
            want = C0.copy()
            oracle.spmm(M, N, M, ALPHA, a, b, c, B, BETA, want)
            _set(engine, row_cluster=rc, fuse_b=0)
            engine.set_option("pipeline_tiles", 1)
            engine.set_matrix_csr(M, M, a, b, c)
            for rp_time in (1, 3):
                out = C0.copy()
                engine.spmm(N, ALPHA, B, BETA, out, rp_time=rp_time)
                assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (name, N, rp_time, engine.last_kernel())
            assert engine.last_kernel().startswith("spmm_csr_panel_v2")
    finally:
        engine.set_option("pipeline_tiles", 0)
        _set(engine)


def test_renumbered_fem_at_full_size():
    """The 4M-row 3-dof FEM matrix (318 M non-zeros) under a random node order, renumbered in HBM: too large for the oracle as a
    whole, so (1) the reordered form must equal the natural-order forms of the same matrix BIT FOR BIT at N = 16 and N = 40 (the
    natural-order forms are pinned to cpu_spmm_CSR by every other test), (2) 600 sampled rows are recomputed by the oracle from the
    host generator + the same renumbering, (3) the plan figures are what the clustering is for."""
    import torch
    from oracle.bindings import Oracle
    from sextans_amd import api, meshgen
    n, dof = 110, 3
    M = n * n * n * dof
    perm = meshgen.node_permutation(M // dof, dof, 1)
    base = api.gen_fem3d_device(0, n, n, n, dof, 3)
    nnz = base[3]
    p = api.permute_symmetric_device(0, M, nnz, *base[:3], perm)
    for q in base[:3]:
        api.device_free(0, q)
    st = torch.cuda.current_stream().cuda_stream
    try:
        for N in (16, 40):
            B = torch.empty(M * N, device="cuda"); Cin = torch.empty(M * N, device="cuda")
            api.gen_uniform_device(0, B.data_ptr(), M * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
            outs = {}
            for rc in (0, -1):
                with api.Engine(0) as e:
                    e.set_option("row_cluster", rc)
                    e.set_matrix_csr_device(M, M, nnz, *p)
                    out = torch.zeros(M * N, device="cuda")
                    e.spmm_device(N, float(ALPHA), B.data_ptr(), M, float(BETA), Cin.data_ptr(), out.data_ptr(), M, st)
                    torch.cuda.synchronize()
                    outs[rc] = out
                    if rc == -1:
                        assert e.last_kernel() == "spmm_csr_panel_v2_reordered" and int(e.get_stat("row_cluster")) == 2
                        assert e.get_stat("panel_rows_clustered") < 0.35 * e.get_stat("panel_rows_natural")
                        assert M / e.get_stat("panel_blocks_clustered") > 58
            assert torch.equal(outs[0].view(torch.int32), outs[-1].view(torch.int32)), N
            if N == 16:      # sampled rows against the oracle: row r of the renumbered matrix = row old(r) of the generator's matrix
                o = Oracle()
                old_of_new = np.empty(M, np.int64); old_of_new[perm] = np.arange(M)
                Bh = B.cpu().numpy(); Ch = Cin.cpu().numpy(); got = outs[-1].cpu().numpy()
                rs = np.random.RandomState(0)
                for r in rs.choice(M, 600, replace=False):
                    r0 = int(old_of_new[r])
                    rp, ci, v = api.gen_fem3d_host(n, n, n, dof, 3, r0, r0 + 1)
                    cols = perm[ci].astype(np.int32)
                    order = np.argsort(cols, kind="stable")
                    rp1 = np.array([0, len(cols)], np.int32)
                    want = np.ascontiguousarray(Ch.reshape(N, M)[:, r]).copy()
                    o.spmm(1, N, M, ALPHA, rp1, cols[order], v[order], Bh, BETA, want)
                    assert np.array_equal(want.view(np.uint32), np.ascontiguousarray(got.reshape(N, M)[:, r]).view(np.uint32)), r
            del B, Cin, outs
            torch.cuda.empty_cache()
    finally:
        for q in p:
            api.device_free(0, q)


def _with_long_rows(rp, ci, v, M, rows, L, seed=1):
    """rows `rows` get ~L more distinct random columns (sorted, values U(-1, 1))"""
    rs = np.random.RandomState(seed)
    rp, ci, v = np.asarray(rp), np.asarray(ci), np.asarray(v)
    lens = np.diff(rp).copy()
    pc, pv, prev = [], [], 0
    for r in sorted(rows):
        pc.append(ci[rp[prev]:rp[r]]); pv.append(v[rp[prev]:rp[r]])
        c = np.union1d(ci[rp[r]:rp[r + 1]], rs.choice(M, L, replace=False)).astype(np.int32)
        pc.append(c); pv.append(rs.uniform(-1, 1, len(c)).astype(np.float32))
        lens[r] = len(c); prev = r + 1
    pc.append(ci[rp[prev]:]); pv.append(v[rp[prev]:])
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), np.concatenate(pc).astype(np.int32), np.concatenate(pv).astype(np.float32)


@pytest.mark.parametrize("N", [16, 40])
def test_a_few_long_rows_do_not_take_the_clustered_plans_away(engine, oracle, N):
    """A mesh matrix with a handful of rows that cannot fit an LDS panel (constraint rows: 600 .. 5000 entries).  Such a row used to
    stay in the main matrix (its share of the non-zeros is far below the bucketing rule's 2 %), made its block a direct block and the
    plan "mixed": every clustered plan declined, the register-resident kernel gone, one workgroup running thousands of entries alone
    (1.5M-row FEM: 267 -> 547 us per step).  Now rows longer than 512 entries always leave for the piece path (one piece each: same
    order), the grid-brick plan serves the rest, and so does the graph-clustered plan with the REORDERED form: the piece kernel reads
    the permuted panels through the column relabelling and its rows are folded into C behind the staging -> C pass.  Bit-identical."""
    from sextans_amd import meshgen
    rp, ci, v, M = _fem(20, 18, 16, 3)
    long_rows = [7, M // 2 + 1, M - 2]
    grid = _with_long_rows(rp, ci, v, M, long_rows, 700)
    perm = meshgen.node_permutation(M // 3, 3, 4)
    rnd = meshgen.permute_symmetric(*_with_long_rows(rp, ci, v, M, long_rows[:1], 3000), M, perm)
    rs = np.random.RandomState(N)
    try:
        for name, (rp2, ci2, v2), state in (("grid order", grid, 1), ("random order", rnd, 2)):
            B, C0 = _operands(rs, M, M, N)
            want = C0.copy()
            oracle.spmm(M, N, M, ALPHA, rp2, ci2, v2, B, BETA, want)
            for rc in (1 if state == 1 else 2, -1, 0):
                _set(engine, row_cluster=rc, fuse_b=0)
                engine.set_matrix_csr(M, M, rp2, ci2, v2)
                for rp_time in (1, 3):
                    out = C0.copy()
                    engine.spmm(N, ALPHA, B, BETA, out, rp_time=rp_time)
                    assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (name, N, rc, rp_time, engine.last_kernel())
                    inplace = C0.copy()                              # C_in == C_out on the device path is the default of spmm(); rows of the piece path
                    engine.spmm(N, ALPHA, B, BETA, inplace, rp_time=1)  # read their C_in AFTER the staging pass wrote the rest
                    assert np.array_equal(inplace.view(np.uint32), want.view(np.uint32))
                assert int(engine.get_stat("piece_path_rows")) == (3 if state == 1 else 1), (name, engine.get_stat("piece_path_rows"))
                assert int(engine.get_stat("reassociated_rows")) == 0
                if rc != 0:
                    assert int(engine.get_stat("row_cluster")) == state, (name, rc, engine.get_stat("row_cluster"), engine.get_stat("cluster_decline"))
                    if state == 2 and N % 16 == 0:
                        assert engine.last_kernel() == "spmm_csr_panel_v2_reordered"
    finally:
        _set(engine)


def test_dof_major_numbering_gets_the_graph_plan(engine, oracle):
    """All x unknowns, then all y, then all z (what block-field FEM codes write): a row's 81 columns sit in three far-apart ranges, the
    grid detector still finds the strides, but a brick holds ONE unknown per node and its dictionary is three times the node-major one
    (blocks cut to ~28 rows).  The automatic choice builds the graph plan as well and keeps it when it copies >= 40 % fewer B rows than
    the bricks (3M-row matrix, N = 16: 860 -> 712 us per step); row_cluster = 1 keeps the bricks.  Bit-identical either way."""
    from sextans_amd import meshgen
    nx, ny, nz, dof = 30, 30, 28, 3
    rp, ci, v, M = _fem(nx, ny, nz, dof)
    nn = nx * ny * nz
    perm = (np.arange(dof)[None, :] * nn + np.arange(nn)[:, None]).reshape(-1)          # row node * dof + d -> d * nn + node
    rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, perm)
    N = 16
    rs = np.random.RandomState(3)
    B, C0 = _operands(rs, M, M, N)
    want = C0.copy()
    oracle.spmm(M, N, M, ALPHA, rp, ci, v, B, BETA, want)
    try:
        rows = {}
        for rc, state in ((-1, 2), (1, 1), (0, -1)):
            _set(engine, row_cluster=rc, fuse_b=0)
            engine.set_matrix_csr(M, M, rp, ci, v)
            out = C0.copy()
            engine.spmm(N, ALPHA, B, BETA, out)
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (rc, engine.last_kernel())
            assert int(engine.get_stat("row_cluster")) == state, (rc, engine.get_stat("row_cluster"), engine.get_stat("cluster_decline"))
            rows[rc] = engine.get_stat("panel_rows_clustered")
        assert rows[-1] < 0.6 * rows[1], rows
    finally:
        _set(engine)


@pytest.mark.parametrize("dims", [(17, 17, 17), (18, 17, 15), (19, 15, 15)])
def test_reordered_form_row_counts_not_a_multiple_of_four(engine, oracle, dims):
    """M % 4 = 1, 2, 3: the staging -> C pass writes 4 consecutive rows of a column per lane where the address is 16-byte aligned and
    falls back to scalar stores elsewhere (and for the last rows of a column); N = 8, 16, 24, 40 cover the half-empty tail tile."""
    from sextans_amd import meshgen
    rp, ci, v, M = _fem(*dims, 1)
    assert M % 4 != 0 and M >= 4096
    rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M, 1, 9))
    rs = np.random.RandomState(M)
    try:
        _set(engine, row_cluster=2, fuse_b=0)
        engine.set_matrix_csr(M, M, rp, ci, v)
        for N in (16, 24, 40, 8):
            B, C0 = _operands(rs, M, M, N)
            want = C0.copy()
            oracle.spmm(M, N, M, ALPHA, rp, ci, v, B, BETA, want)
            out = C0.copy()
            engine.spmm(N, ALPHA, B, BETA, out)
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (dims, N, engine.last_kernel())
            if N >= 16:
                assert engine.last_kernel() == "spmm_csr_panel_v2_reordered" and int(engine.get_stat("row_cluster")) == 2
    finally:
        _set(engine)


def test_row_slab_of_a_renumbered_matrix_clusters_with_its_row_offset(engine, oracle):
    """What a rank of the row-partitioned SpMM holds: rows [r0, r1) of a square matrix, all K columns.  Told where its rows sit (option
    row_offset -- sextans_dist_spmm sets it), the engine clusters the slab over its own square pattern (edges to rows of other ranks
    dropped) and runs the reordered form on the rectangular matrix; without the offset a non-square matrix is clustered over its
    row-similarity graph since round 5 (rows joined through shared columns; until round 4 it kept the natural-order forms).
    Bit-identical to the same rows of cpu_spmm_CSR either way."""
    from sextans_amd import meshgen
    from sextans_amd import dist as sxd
    rp, ci, v, M = _fem(26, 24, 22, 3)
    rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 13))
    N = 16
    rs = np.random.RandomState(17)
    B, C0 = _operands(rs, M, M, N)
    want = C0.copy()
    oracle.spmm(M, N, M, ALPHA, rp, ci, v, B, BETA, want)
    try:
        for r0, r1 in ((0, M // 2), (M // 2, M - 4097), (M - 4097, M)):   # (the last slab holds a tenth of the rows: too few of its neighbours are its own)
            lrp, lci, lv = sxd.slice_csr(rp, ci, v, r0, r1)
            m = r1 - r0
            Cl = np.ascontiguousarray(C0.reshape(N, M)[:, r0:r1]).reshape(-1)
            wl = np.ascontiguousarray(want.reshape(N, M)[:, r0:r1]).reshape(-1)
            for off, kind in ((r0, 1), (-1, 2)):
                _set(engine, row_cluster=-1, fuse_b=0)
                engine.set_option("row_offset", off)
                engine.set_matrix_csr(m, M, lrp, lci, lv)
                out = Cl.copy()
                engine.spmm(N, ALPHA, B, BETA, out)
                assert np.array_equal(out.view(np.uint32), wl.view(np.uint32)), (r0, r1, off, engine.last_kernel())
                if r1 - r0 > M // 3:
                    assert int(engine.get_stat("row_cluster")) == 2 and int(engine.get_stat("cluster_graph_kind")) == kind, (
                        r0, r1, off, engine.get_stat("row_cluster"), engine.get_stat("cluster_decline"), engine.get_stat("cluster_graph_kind"))
    finally:
        engine.set_option("row_offset", -1)
        _set(engine)


def test_exported_row_order_renumbers_a_matrix_into_compact_row_ranges(engine, oracle):
    """sextans_export_row_order: the clustered plan's row order, for callers that can renumber their matrix once (P A P^T, like RCM).
    A randomly numbered mesh renumbered by it has full natural-order row blocks again (so contiguous row ranges -- the partition of
    sextans_dist_spmm -- are compact pieces of the graph), and its SpMM is the same SpMM: bit-identical to cpu_spmm_CSR after the
    rows of B and C are permuted along."""
    from sextans_amd import meshgen
    rp, ci, v, M = _fem(22, 20, 18, 3)
    rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 21))
    try:
        _set(engine)
        engine.set_matrix_csr(M, M, rp, ci, v)
        order, kind = engine.export_row_order()
        assert kind == 2 and np.array_equal(np.sort(order), np.arange(M))
        nat_random = engine.get_stat("panel_rows_natural")
        new_of_old = np.empty(M, np.int64); new_of_old[order] = np.arange(M)
        rp2, ci2, v2 = meshgen.permute_symmetric(rp, ci, v, M, new_of_old)
        N = 16
        rs = np.random.RandomState(1)
        B, C0 = _operands(rs, M, M, N)
        want = C0.copy()
        oracle.spmm(M, N, M, ALPHA, rp, ci, v, B, BETA, want)
        # the renumbered problem: row / column i of the old matrix is row / column new_of_old[i] of the new one
        B2 = np.ascontiguousarray(B.reshape(N, M)[:, order]).reshape(-1); C2 = np.ascontiguousarray(C0.reshape(N, M)[:, order]).reshape(-1)
        engine.set_option("row_cluster", 0)                       # natural-order forms only: the numbering itself now has the locality
        engine.set_matrix_csr(M, M, rp2, ci2, v2)
        out = C2.copy()
        engine.spmm(N, ALPHA, B2, BETA, out)
        # (the new matrix's rows hold the same entries in a different column order: sums may associate differently -- compare against
        # the oracle ON THE RENUMBERED MATRIX for bits, against the original for the value)
        want2 = C2.copy()
        oracle.spmm(M, N, M, ALPHA, rp2, ci2, v2, B2, BETA, want2)
        assert np.array_equal(out.view(np.uint32), want2.view(np.uint32))
        assert np.allclose(out.reshape(N, M), want.reshape(N, M)[:, order], rtol=1e-4, atol=1e-4)
        assert engine.get_stat("panel_rows_natural") < 0.5 * nat_random and M / engine.get_stat("panel_blocks") > 50
        # a matrix without a clustered order: identity
        urp, uci, uv = random_csr(rs, 6000, 6000, 9)
        _set(engine)
        engine.set_matrix_csr(6000, 6000, urp, uci, uv)
        order, kind = engine.export_row_order()
        assert kind == 0 and np.array_equal(order, np.arange(6000))
    finally:
        _set(engine)
