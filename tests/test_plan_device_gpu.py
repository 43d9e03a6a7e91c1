"""The packed row-bucketed form of A is built on the DEVICE since round 3 (csrc/plan_device.hip: block formation with
an LDS hash set per part of 64 row blocks, dictionaries by bitonic sort, stream re-encoding -- the CSR arrays never
leave HBM).  It must be byte-identical to the host builder of the same format, sextans_pack_csr (csrc/panel_plan.cpp),
which follows the same greedy rule over the same parts.  Reference analogue being replaced: the host-side scheduling and
packing of the non-zero stream, sparse_helper.h:345-473 / sextans-host.cpp:114-148."""
import time

import numpy as np
import pytest

from util import NASA, random_csr

pytestmark = pytest.mark.gpu

KEYS = ("blk_row", "dict_ptr", "dict", "row_off", "idx16", "col32", "val")


def _same(dev, host, what):
    for k in ("M", "K", "nnz", "lanes_per_row", "nblk", "stream_len", "max_dict", "nnz_in_panel_blocks"):
        assert dev[k] == host[k], (what, k, dev[k], host[k])
    for k in KEYS:
        a, b = dev[k], host[k]
        assert a.shape == b.shape, (what, k, a.shape, b.shape)
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)      # -0.0f padding must be -0.0f
        assert np.array_equal(a, b), (what, k, int(np.argmax(a != b)))


def _cases():
    from sextans_amd import api
    rs = np.random.RandomState(123)
    yield "fem 3 dof", (*api.gen_fem3d_host(14, 12, 9, 3, 7), 14 * 12 * 9 * 3)
    yield "fem 1 dof", (*api.gen_fem3d_host(30, 20, 10, 1, 5), 6000)
    frp, fci, fv = api.gen_fem3d_host(9, 8, 6, 3, 7)
    rrp, rci, rv = random_csr(rs, 900, 1296, 15, long_rows=3)          # no reuse + long rows: direct blocks (mixed plan)
    yield "fem + random", (np.concatenate([frp, frp[-1] + rrp[1:]]).astype(np.int32), np.concatenate([fci, rci]).astype(np.int32),
                           np.concatenate([fv, rv]).astype(np.float32), 1296)
    # ragged: empty rows, rows with duplicates, one row wider than the whole panel, M not a multiple of anything
    M, K = 5003, 7001
    rows = []
    for r in range(M):
        n = int(rs.choice([0, 0, 1, 2, 5, 17, 40, 90]))
        if r == 777:
            n = 3000
        lo = max(0, min(K - 400, int(r * K / M) - 200))
        c = rs.randint(lo, lo + 400 if r != 777 else K, n)
        rows.append(np.sort(c).astype(np.int32))                          # duplicates allowed
    rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum([len(c) for c in rows])
    ci = np.concatenate(rows).astype(np.int32)
    yield "ragged banded", (rp, ci, rs.uniform(-1, 1, len(ci)).astype(np.float32), K)
    yield "empty", (np.zeros(4, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), 5)


@pytest.mark.parametrize("lanes", [2, 4, 8])
@pytest.mark.parametrize("min_reuse", [0, 200, 400])
def test_device_plan_is_byte_identical_to_the_host_builder(engine, sx, lanes, min_reuse):
    from sextans_amd import api
    for k, v in dict(kernel=2, split_rows=0, bucket_rows=0, exact_chain=0, mfma_dense_tiles=0).items():   # no row leaves the main matrix
        engine.set_option(k, v)
    try:
        for what, (rp, ci, v, K) in _cases():
            M = len(rp) - 1
            engine.set_option("panel_min_reuse_x100", min_reuse)
            engine.set_option("panel_min_reuse_wide_x100", min_reuse)   # (the plan is built for the lower of the two thresholds)
            engine.set_matrix_csr(M, K, rp, ci, v)
            dev = engine.export_plan(lanes)
            host = api.pack_csr(M, K, rp, ci, v, lanes, min_reuse)
            _same(dev, host, (what, lanes, min_reuse))
        rp, ci, v, M, K, nnz = sx.read_suitsparse_matrix(NASA)
        engine.set_matrix_csr(M, K, rp, ci, v)
        _same(engine.export_plan(lanes), api.pack_csr(M, K, rp, ci, v, lanes, min_reuse), ("nasa4704", lanes, min_reuse))
    finally:
        for k, v in dict(kernel=0, bucket_rows=-1, exact_chain=1, panel_min_reuse_x100=200, panel_min_reuse_wide_x100=150).items():
            engine.set_option(k, v)


def test_device_plan_of_a_device_resident_matrix_and_its_build_time(sx):
    """A matrix that only exists in HBM (generated there), 1.3 M rows / 100 M non-zeros: identical to the host builder run
    on the host twin of the generator, and built in well under the 0.3 s the round-2 review asked for at 3x that size."""
    import torch
    from sextans_amd import api
    dims = (75, 75, 75, 3)
    M = K = dims[0] * dims[1] * dims[2] * dims[3]
    p, i, v, nnz = api.gen_fem3d_device(0, *dims, 3)
    try:
        with sx.Engine(0) as e:
            e.set_option("kernel", 2)
            e.set_matrix_csr_device(M, K, nnz, p, i, v)
            t0 = time.perf_counter()
            dev = e.export_plan(4)
            t_total = time.perf_counter() - t0
            built = e.get_stat("plan_build_s")
        hrp, hci, hv = api.gen_fem3d_host(*dims, 3)
        t0 = time.perf_counter()
        host = api.pack_csr(M, K, hrp, hci, hv, 4, 150)   # the engine's default: min(panel_min_reuse_x100, .._wide_x100)
        t_host = time.perf_counter() - t0
        print(f"device plan build {built * 1e3:.1f} ms (export incl. read-back {t_total:.2f} s); host builder {t_host:.2f} s; "
              f"{nnz} nnz, {dev['nblk']} blocks")
        _same(dev, host, "fem 75^3 x 3")
        assert built < 0.15, built
    finally:
        for q in (p, i, v):
            api.device_free(0, q)
        torch.cuda.empty_cache()


def test_device_matrix_is_validated_on_the_device(sx):
    """sextans_set_matrix_csr_device takes pointers nobody has looked at: a column index outside [0, K) or a
    non-monotone row_ptr must be rejected before any kernel gathers with it."""
    import torch
    from sextans_amd import api
    rp = torch.tensor([0, 2, 4, 6], dtype=torch.int32, device="cuda")
    ci = torch.tensor([0, 1, 2, 9, 1, 2], dtype=torch.int32, device="cuda")       # 9 >= K
    v = torch.ones(6, dtype=torch.float32, device="cuda")
    B = torch.ones(4 * 8, device="cuda"); Cm = torch.zeros(3 * 8, device="cuda")
    with sx.Engine(0) as e:
        e.set_option("kernel", 2)
        e.set_matrix_csr_device(3, 4, 6, rp.data_ptr(), ci.data_ptr(), v.data_ptr())
        with pytest.raises(api.SextansError):
            e.spmm_device(8, 1.0, B.data_ptr(), 4, 0.0, Cm.data_ptr(), Cm.data_ptr(), 3)
        rp2 = torch.tensor([0, 4, 2, 6], dtype=torch.int32, device="cuda")
        ci2 = torch.tensor([0, 1, 2, 3, 1, 2], dtype=torch.int32, device="cuda")
        e.set_matrix_csr_device(3, 4, 6, rp2.data_ptr(), ci2.data_ptr(), v.data_ptr())
        with pytest.raises(api.SextansError):
            e.spmm_device(8, 1.0, B.data_ptr(), 4, 0.0, Cm.data_ptr(), Cm.data_ptr(), 3)


def test_shared_index_lists(engine, oracle):
    """Consecutive rows of a block whose 16-bit index lists are equal up to a constant shift -- the dof rows of a mesh node (shift 0),
    the next node along a grid line (shift = its dictionary rows) -- keep ONE copy of the list (plan_device.hip: share_index_lists;
    6 -> ~4.2 bytes per non-zero on a grid-ordered matrix).  The exported plan is byte-identical either way (the public form carries
    every row's own list), results are bit-identical on both panel kernels and on the clustered plan, and the index stream shrinks by
    more than half for 3-dof and 1-dof grid matrices alike."""
    from sextans_amd import api
    import numpy as np
    from util import ALPHA, BETA
    rs = np.random.RandomState(2)
    for dims, dof, shrink in (((12, 11, 10), 3, True), ((20, 18, 16), 1, True)):
        M = K = dims[0] * dims[1] * dims[2] * dof
        rp, ci, v = api.gen_fem3d_host(*dims, dof, 7)
        N = 32
        B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        want = C0.copy()
        oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
        plans = {}
        try:
            for share in (0, 1):
                engine.set_option("share_index", share)
                engine.set_option("kernel", 2); engine.set_option("fuse_b", 0)
                engine.set_matrix_csr(M, K, rp, ci, v)
                for rc in (0, -1):                       # natural-order plan (both panel kernels) and the clustered one
                    engine.set_option("row_cluster", rc)
                    for pv2 in (0, -1):
                        engine.set_option("panel_v2", pv2)
                        out = C0.copy()
                        engine.spmm(N, ALPHA, B, BETA, out)
                        assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (dims, share, rc, pv2, engine.last_kernel())
                engine.set_option("row_cluster", 0)
                out = C0.copy(); engine.spmm(N, ALPHA, B, BETA, out)
                plans[share] = (engine.export_plan(4), engine.get_stat("index_stream_entries"), engine.get_stat("value_stream_entries"))
            p0, p1 = plans[0][0], plans[1][0]
            for name in ("blk_row", "dict_ptr", "dict", "row_off", "idx16", "val"):
                assert np.array_equal(p0[name], p1[name]), name
            assert plans[0][1] == plans[0][2]
            if shrink:
                assert plans[1][1] < 0.5 * plans[1][2], plans[1][1:]
            else:
                assert plans[1][1] == plans[1][2]
        finally:
            for k, val in (("share_index", 1), ("kernel", 0), ("fuse_b", 1), ("row_cluster", -1), ("panel_v2", -1)):
                engine.set_option(k, val)
