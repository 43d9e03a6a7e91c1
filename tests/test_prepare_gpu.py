"""sextans_prepare / sextans_dist_prepare / sextans_dist_bind_library (round 6): everything a first call would build inside itself is built
ahead of it -- so that the call can sit inside a hipGraph capture or a timed region -- and the collectives library can be chosen.
The analogue in the reference: the host prepares its streams before tapa::invoke (/root/reference/src/sextans-host.cpp:114-204)."""
import numpy as np
import pytest

from util import ALPHA, BETA, random_csr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout", ["colmajor", "rowmajor"])
@pytest.mark.parametrize("matrix", ["fem", "random order", "random"])
def test_prepared_call_inside_a_stream_capture(sx, oracle, layout, matrix):
    """After sextans_prepare the compute call allocates nothing and never synchronises with the host: it can be captured into a graph
    (torch.cuda.graph = hipStreamBeginCapture on a side stream) and replayed with new operand contents; bit-identical to the oracle."""
    import torch
    from sextans_amd import api, meshgen
    rs = np.random.RandomState(3)
    if matrix == "random":
        M, K = 5000, 4000
        rp, ci, v = random_csr(rs, M, K, 14, long_rows=1)
    else:
        rp, ci, v = api.gen_fem3d_host(30, 28, 26, 3, 7)
        M = K = 30 * 28 * 26 * 3
        if matrix == "random order":
            rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 9))
    N = 32
    rm = layout == "rowmajor"
    with sx.Engine(0) as e:
        e.set_matrix_csr(M, K, rp, ci, v)
        e.prepare(N, rowmajor=rm)
        build_s = e.get_stat("plan_build_s")
        dB = torch.empty(K * N, device="cuda"); dCin = torch.empty(M * N, device="cuda"); dC = torch.empty(M * N, device="cuda")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st = torch.cuda.current_stream().cuda_stream
            if rm:
                e.spmm_device_rm(N, ALPHA, dB.data_ptr(), N, BETA, dCin.data_ptr(), N, dC.data_ptr(), N, st)
            else:
                e.spmm_device(N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), dC.data_ptr(), M, st)
        assert e.get_stat("plan_build_s") == build_s        # nothing was built inside the captured call
        for trial in range(2):                               # new operands, same graph
            B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
            want = C0.copy()
            oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
            if rm:
                dB.copy_(torch.from_numpy(np.ascontiguousarray(B.reshape(N, K).T).reshape(-1)))
                dCin.copy_(torch.from_numpy(np.ascontiguousarray(C0.reshape(N, M).T).reshape(-1)))
            else:
                dB.copy_(torch.from_numpy(B)); dCin.copy_(torch.from_numpy(C0))
            g.replay()
            torch.cuda.synchronize()
            got = dC.cpu().numpy()
            if rm:
                got = np.ascontiguousarray(got.reshape(M, N).T).reshape(-1)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (layout, matrix, trial, e.last_kernel())


def test_dist_prepare_without_a_communicator_and_bind_library_errors(sx, oracle):
    """world == 1 with comm == NULL: the local half of the preparation for all three forms; an unloadable collectives library is an
    error (not a silent fallback) and the default binding comes back afterwards."""
    import torch
    from sextans_amd import api
    rs = np.random.RandomState(5)
    rp, ci, v = api.gen_fem3d_host(12, 11, 10, 3, 7)
    M = K = 12 * 11 * 10 * 3
    N = 16
    B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    want = C0.copy()
    oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, want)
    st = torch.cuda.current_stream().cuda_stream
    with sx.Engine(0) as e:
        e.set_matrix_csr(M, K, rp, ci, v)
        for form, nchunks in ((0, 3), (1, 0)):
            e.dist_prepare(None, 1, 0, [(0, M)], N, nchunks=nchunks, form=form, stream=st)
        x0 = e.get_stat("dist_setup_exchanges")
        dB = torch.from_numpy(B).cuda(); dCin = torch.from_numpy(C0).cuda(); out = torch.full((M * N,), float("nan"), device="cuda")
        e.dist_spmm(None, 1, 0, [(0, M)], N, ALPHA, dB.data_ptr(), K, BETA, dCin.data_ptr(), M, out.data_ptr(), M, nchunks=3, stream=st)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert e.get_stat("dist_setup_exchanges") == x0
        with pytest.raises(api.SextansError):               # ranges that do not match the engine's matrix
            e.dist_prepare(None, 1, 0, [(0, M - 1)], N, nchunks=3, form=0, stream=st)
        with pytest.raises(api.SextansError):               # blocked-ELL form without a blocked-ELL matrix
            e.dist_prepare(None, 1, 0, [(0, 1024)], 32, form=2, stream=st)
    with pytest.raises(api.SextansError):
        api.dist_bind_library("/nonexistent/librccl_of_nobody.so")
    api.dist_bind_library(None)                              # back to the default search: RCCL loads again
    comm = api.dist_comm_init(0, 1, 0, api.dist_unique_id())
    with pytest.raises(api.SextansError):                    # ... and cannot be replaced while a communicator is alive
        api.dist_bind_library(None)
    api.dist_comm_destroy(comm)
    api.dist_bind_library(None)
