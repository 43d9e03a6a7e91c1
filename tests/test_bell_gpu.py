"""Blocked-ELL bf16 MFMA path (BASELINE config 5).  The reference has no such path, so parity here is
UNPINNED by the reference: the checker is our fp32 CPU restatement on the same bf16 inputs plus a
float64 evaluation.  Tolerance (stated): |gpu - f64| <= 4e-6 * sum|a*b| + 1e-6*|beta*c|  -- fp32
accumulation of exact bf16 products; the MFMA's in-instruction summation order is hardware-defined,
so bit equality with a sequential CPU sum is not expected."""
import numpy as np
import pytest


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f64_reference(M, K, N, W, bcol, bval, B16, alpha, beta, C0):
    A = np.zeros((M, K), np.float64)
    blocks = bf16_to_f32(bval).reshape(M // 32, W, 32, 32).astype(np.float64)
    for br in range(M // 32):
        for s in range(W):
            bc = bcol[br * W + s]
            if bc >= 0:
                A[br * 32:(br + 1) * 32, bc * 32:(bc + 1) * 32] += blocks[br, s]
    Bm = bf16_to_f32(B16).reshape(N, K).T.astype(np.float64)
    Cm = C0.reshape(N, M).T.astype(np.float64)
    out = alpha * (A @ Bm) + beta * Cm
    asum = np.abs(A) @ np.abs(Bm)
    return out.T.reshape(-1), asum.T.reshape(-1)


def test_bell_host_generator_properties(sx):
    from sextans_amd import api
    M, K, W = 512, 2048, 7
    c, v = api.gen_bell_host(M, K, W, 5)
    c = c.reshape(M // 32, W)
    assert c.min() >= 0 and c.max() < K // 32 and np.all(np.diff(c, axis=1) > 0)
    f = bf16_to_f32(v)
    assert f.min() >= -1 and f.max() <= 1 and abs(f.mean()) < 0.01
    c2, v2 = api.gen_bell_host(M, K, W, 5)
    assert np.array_equal(c.reshape(-1), c2) and np.array_equal(v, v2)
    c3, _ = api.gen_bell_host(64, 64, 2, 1)          # tiny K/32 = 2 == ell_width: columns must be [0,1]
    assert np.array_equal(c3.reshape(2, 2), [[0, 1], [0, 1]])


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,W", [(256, 512, 128, 5), (64, 64, 32, 2), (512, 1024, 256, 9), (96, 320, 64, 3),
                                     (128, 256, 96, 4)])
def test_bell_mfma_vs_cpu_restatement(engine, oracle, sx, M, K, N, W):
    import torch
    from sextans_amd import api
    rs = np.random.RandomState(M + N)
    bcol, bval = api.gen_bell_host(M, K, W, 5)
    if W >= 4:                                          # some empty slots (-1) in the ELL structure
        bcol = bcol.copy().reshape(M // 32, W)
        bcol[::2, -1] = -1
        bcol[1, 0] = -1
        bcol = bcol.reshape(-1)
    B16 = api.gen_uniform_bf16_host(K * N, 6)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    alpha, beta = np.float32(0.85), np.float32(-2.06)
    want32 = C0.copy()
    oracle.bell_spmm(M, K, N, W, bcol, bval, B16, alpha, beta, want32)
    want64, asum = f64_reference(M, K, N, W, bcol, bval, B16, float(alpha), float(beta), C0)
    engine.set_matrix_bell(M, K, W, bcol, bval)
    dB = torch.from_numpy(B16.view(np.int16)).cuda()
    dCin = torch.from_numpy(C0).cuda()
    dC = torch.zeros(M * N, device="cuda")
    engine.spmm_bell_device(N, alpha, dB.data_ptr(), K, beta, dCin.data_ptr(), dC.data_ptr(), M,
                            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert engine.last_kernel() in ("spmm_bell_mfma", "spmm_bell_mfma_shared")
    got = dC.cpu().numpy().astype(np.float64)
    tol = 4e-6 * asum + 1e-6 * np.abs(float(beta) * C0) + 1e-30
    assert np.all(np.abs(got - want64) <= tol), float(np.max(np.abs(got - want64) / tol))
    assert np.all(np.abs(want32.astype(np.float64) - want64) <= tol)        # the fp32 oracle is inside the same band
    rel = np.linalg.norm(got - want64) / np.linalg.norm(want64)
    assert rel < 1e-6
    # transpose-detecting: asymmetric inputs above; also in-place C (C_in == C_out)
    dC2 = dCin.clone()
    engine.spmm_bell_device(N, alpha, dB.data_ptr(), K, beta, dC2.data_ptr(), dC2.data_ptr(), M,
                            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(dC2, dC)


@pytest.mark.gpu
def test_bell_device_generator_matches_host(engine, sx):
    import ctypes as C
    import torch
    from sextans_amd import api
    M, K, W = 1024, 4096, 11
    hc, hv = api.gen_bell_host(M, K, W, 5)
    dc, dv = api.gen_bell_device(0, M, K, W, 5)
    try:
        hip = C.CDLL("libamdhip64.so.7")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        tc = torch.empty(len(hc), dtype=torch.int32, device="cuda")
        tv = torch.empty(len(hv), dtype=torch.int16, device="cuda")
        assert hip.hipMemcpy(tc.data_ptr(), dc, len(hc) * 4, 3) == 0 and hip.hipMemcpy(tv.data_ptr(), dv, len(hv) * 2, 3) == 0
        assert np.array_equal(tc.cpu().numpy(), hc) and np.array_equal(tv.cpu().numpy().view(np.uint16), hv)
    finally:
        api.device_free(0, dc); api.device_free(0, dv)
    t = torch.empty(10000, dtype=torch.int16, device="cuda")
    api.gen_uniform_bf16_device(0, t.data_ptr(), 10000, 6, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy().view(np.uint16), api.gen_uniform_bf16_host(10000, 6))


def _run_bell(engine, M, K, N, W, bcol, bval, B16, alpha, beta, C0):
    import torch
    engine.set_matrix_bell(M, K, W, bcol, bval)
    dB = torch.from_numpy(B16.view(np.int16)).cuda()
    dCin = torch.from_numpy(C0).cuda()
    dC = torch.zeros(M * N, device="cuda")
    engine.spmm_bell_device(N, alpha, dB.data_ptr(), K, beta, dCin.data_ptr(), dC.data_ptr(), M, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return dC.cpu().numpy().astype(np.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,hw,holes", [(1024, 1024, 2, False), (32 * 6, 32 * 40, 5, True), (32 * 13, 32 * 13, 6, True),
                                          (2048, 4096, 0, False), (32 * 9, 32 * 64, 12, True)])
def test_bell_shared_tile_kernel_on_block_banded_matrices(engine, sx, M, K, hw, holes):
    """spmm_bell_mfma_shared (N = 256): workgroups of 8 block rows walk the union of their block columns, each B tile
    staged once in an LDS ring.  Block-banded inputs (rows share columns), with empty slots, block-row counts that are not
    a multiple of 8, one-block bands; the per-wavefront kernel on the same input must give the same numbers within the
    stated bound, and both sit inside it."""
    from sextans_amd import api
    N, W = 256, 2 * hw + 1
    rs = np.random.RandomState(M + hw)
    bcol, bval = api.gen_bell_banded_host(M, K, hw, 9)
    bc2 = bcol.reshape(M // 32, W)
    assert np.all(np.diff(bc2, axis=1) == 1) and bc2.min() >= 0 and bc2.max() < K // 32
    if M == K:
        assert np.all(bc2[W:-W, hw] == np.arange(M // 32)[W:-W])           # centred on the diagonal block away from the edges
    if holes:
        bc2 = bc2.copy()
        bc2[::3, 0] = -1; bc2[1::4, W // 2] = -1; bc2[2, :] = -1                # empty slots, one completely empty block row
        bcol = bc2.reshape(-1)
    B16 = api.gen_uniform_bf16_host(K * N, 6)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    alpha, beta = np.float32(0.85), np.float32(-2.06)
    want64, asum = f64_reference(M, K, N, W, bcol, bval, B16, float(alpha), float(beta), C0)
    tol = 4e-6 * asum + 1e-6 * np.abs(float(beta) * C0) + 1e-30
    try:
        engine.set_option("bell_shared", 1)
        got = _run_bell(engine, M, K, N, W, bcol, bval, B16, alpha, beta, C0)
        assert engine.last_kernel() == "spmm_bell_mfma_shared"
        assert np.all(np.abs(got - want64) <= tol), float(np.max(np.abs(got - want64) / tol))
        engine.set_option("bell_shared", 0)
        plain = _run_bell(engine, M, K, N, W, bcol, bval, B16, alpha, beta, C0)
        assert engine.last_kernel() == "spmm_bell_mfma"
        assert np.all(np.abs(plain - want64) <= tol)
        engine.set_option("bell_shared", -1)                                    # auto: banded rows share columns
        _run_bell(engine, M, K, N, W, bcol, bval, B16, alpha, beta, C0)
        share = engine.get_stat("bell_share")
        assert engine.last_kernel() == ("spmm_bell_mfma_shared" if share >= 1.5 else "spmm_bell_mfma"), share
        if hw >= 2 and not holes:
            assert share > 2.0
    finally:
        engine.set_option("bell_shared", -1)


@pytest.mark.gpu
def test_bell_shared_tile_kernel_without_sharing_and_auto_dispatch(engine, sx):
    """Uniformly random block columns (the config-5 generator): no sharing -- forced through the shared-tile kernel the
    result is still right (every tile used by one wavefront), and the automatic choice stays on the per-wavefront kernel."""
    from sextans_amd import api
    M, K, N, W = 32 * 10, 32 * 200, 256, 17
    rs = np.random.RandomState(3)
    bcol, bval = api.gen_bell_host(M, K, W, 5)
    B16 = api.gen_uniform_bf16_host(K * N, 6)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    alpha, beta = np.float32(1.25), np.float32(0.5)
    want64, asum = f64_reference(M, K, N, W, bcol, bval, B16, float(alpha), float(beta), C0)
    tol = 4e-6 * asum + 1e-6 * np.abs(float(beta) * C0) + 1e-30
    try:
        engine.set_option("bell_shared", 1)
        got = _run_bell(engine, M, K, N, W, bcol, bval, B16, alpha, beta, C0)
        assert engine.last_kernel() == "spmm_bell_mfma_shared" and np.all(np.abs(got - want64) <= tol)
        engine.set_option("bell_shared", -1)
        got = _run_bell(engine, M, K, N, W, bcol, bval, B16, alpha, beta, C0)
        assert engine.last_kernel() == "spmm_bell_mfma" and engine.get_stat("bell_share") < 1.5
        assert np.all(np.abs(got - want64) <= tol)
    finally:
        engine.set_option("bell_shared", -1)


@pytest.mark.gpu
def test_duplicate_or_unsorted_block_columns_never_take_the_shared_kernel(engine, sx):
    """sextans_set_matrix_bell* does not require distinct, ascending block columns and the per-wavefront kernels sum every slot;
    spmm_bell_mfma_shared keeps one slot per (block row, union position), so a row that lists a block column twice -- or out of
    order -- must keep the matrix off it, also when the caller asks for it (bell_shared = 1).  """
    from sextans_amd import api
    M, K, N, hw = 32 * 16, 32 * 32, 256, 3
    W = 2 * hw + 1
    rs = np.random.RandomState(5)
    alpha, beta = np.float32(0.85), np.float32(-2.06)
    B16 = api.gen_uniform_bf16_host(K * N, 6)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    bcol0, bval = api.gen_bell_banded_host(M, K, hw, 9)
    try:
        for what in ("duplicate", "unsorted"):
            bc = bcol0.copy().reshape(M // 32, W)
            if what == "duplicate":
                bc[3, 4] = bc[3, 1]; bc[9, W - 1] = bc[9, W - 2]              # the same block column in two slots of a row
            else:
                bc[5, [0, 2]] = bc[5, [2, 0]]
            bcol = bc.reshape(-1)
            want64, asum = f64_reference(M, K, N, W, bcol, bval, B16, float(alpha), float(beta), C0)
            tol = 4e-6 * asum + 1e-6 * np.abs(float(beta) * C0) + 1e-30
            for opt in (-1, 1):
                engine.set_option("bell_shared", opt)
                got = _run_bell(engine, M, K, N, W, bcol, bval, B16, alpha, beta, C0)
                assert engine.last_kernel() == "spmm_bell_mfma", (what, opt)
                assert np.all(np.abs(got - want64) <= tol), (what, opt, float(np.max(np.abs(got - want64) / tol)))
    finally:
        engine.set_option("bell_shared", -1)


@pytest.mark.gpu
def test_measurement_options_are_gated(sx):
    """bell_debug / cluster_shape / cluster_group / phase_timing are not part of the drop-in surface: without
    SEXTANS_DEBUG_OPTIONS=1 in the environment a non-default value is refused (the default value itself is accepted)."""
    import os
    import ctypes as C
    from sextans_amd import api
    L = api.lib()
    if api.device_count() < 1:
        pytest.skip("needs a device to create an engine")
    old = os.environ.pop("SEXTANS_DEBUG_OPTIONS", None)
    e = api.Engine(0)
    try:
        for key, default, other in (("bell_debug", 0, 1), ("cluster_shape", 0, 160202), ("cluster_group", 6, 2), ("phase_timing", 0, 1)):
            assert L.sextans_set_option(e._h, key.encode(), C.c_int64(default)) == 0
            assert L.sextans_set_option(e._h, key.encode(), C.c_int64(other)) != 0, key
            assert e.get_option(key) == default
        os.environ["SEXTANS_DEBUG_OPTIONS"] = "1"
        assert L.sextans_set_option(e._h, b"cluster_group", C.c_int64(2)) == 0
        assert L.sextans_set_option(e._h, b"cluster_shape", C.c_int64(160000)) != 0      # a zero factor is refused even then
        assert L.sextans_set_option(e._h, b"cluster_group", C.c_int64(6)) == 0
    finally:
        e.close()
        if old is None:
            os.environ.pop("SEXTANS_DEBUG_OPTIONS", None)
        else:
            os.environ["SEXTANS_DEBUG_OPTIONS"] = old


@pytest.mark.gpu
def test_bell_block_row_ranges_and_the_dist_entry_point(engine, sx):
    """config 5 shards by block-row ranges.  (i) The slabs of a 3-way partition, each computed by an engine that holds only its
    block rows and written PACKED (sextans_spmm_bell_device2: C_in inside the whole matrix, C_out with its own leading dimension), equal
    the whole-matrix result bit for bit -- a wavefront's 32 rows never depend on the others.  (ii) sextans_dist_spmm_bell on a 1-rank
    RCCL communicator and without one: staging, ncclAllGather, unpack -> the same bits, C_in == C_out allowed."""
    import torch
    from sextans_amd import api
    M, K, N, W = 768, 1024, 256, 6
    st = torch.cuda.current_stream().cuda_stream
    bcol, bval = api.gen_bell_host(M, K, W, 9)
    B16 = api.gen_uniform_bf16_host(K * N, 3)
    C0 = np.random.RandomState(1).uniform(-1, 1, M * N).astype(np.float32)
    alpha, beta = np.float32(0.85), np.float32(-2.06)
    dB = torch.from_numpy(B16.view(np.int16)).cuda(); dCin = torch.from_numpy(C0).cuda()
    whole = torch.zeros(M * N, device="cuda")
    engine.set_matrix_bell(M, K, W, bcol, bval)
    engine.spmm_bell_device(N, alpha, dB.data_ptr(), K, beta, dCin.data_ptr(), whole.data_ptr(), M, st)
    torch.cuda.synchronize()
    ranges = [(0, 256), (256, 576), (576, 768)]
    out = torch.full((M * N,), float("nan"), device="cuda")
    with api.Engine(0) as e:
        for r0, r1 in ranges:
            e.set_matrix_bell(r1 - r0, K, W, bcol[r0 // 32 * W:r1 // 32 * W], bval[r0 // 32 * W * 1024:r1 // 32 * W * 1024])
            slab = torch.full(((r1 - r0) * N,), float("nan"), device="cuda")
            e.spmm_bell_device2(N, alpha, dB.data_ptr(), K, beta, dCin.data_ptr() + 4 * r0, M, slab.data_ptr(), r1 - r0, st)
            out.view(N, M)[:, r0:r1] = slab.view(N, r1 - r0)
        torch.cuda.synchronize()
        assert torch.equal(out, whole)
    for with_comm in (True, False):
        comm = api.dist_comm_init(0, 1, 0, api.dist_unique_id()) if with_comm else None
        try:
            out = torch.full((M * N,), float("nan"), device="cuda")
            engine.dist_spmm_bell(comm, 1, 0, [(0, M)], N, alpha, dB.data_ptr(), K, beta, dCin.data_ptr(), M, out.data_ptr(), M, stream=st)
            torch.cuda.synchronize()
            assert torch.equal(out, whole), with_comm
            inpl = dCin.clone()
            engine.dist_spmm_bell(comm, 1, 0, [(0, M)], N, alpha, dB.data_ptr(), K, beta, inpl.data_ptr(), M, inpl.data_ptr(), M, stream=st)
            torch.cuda.synchronize()
            assert torch.equal(inpl, whole), with_comm
            with pytest.raises(Exception):   # ranges are whole block rows
                engine.dist_spmm_bell(comm, 1, 0, [(0, M - 8)], N, alpha, dB.data_ptr(), K, beta, dCin.data_ptr(), M, out.data_ptr(), M, stream=st)
        finally:
            if comm is not None:
                api.dist_comm_destroy(comm)
