""""the replacement must be re-entrant per handle" (the reference is single-threaded; a handle is used by one thread at a
time, different handles by different threads at once -- the thread-per-GPU model of examples/dist_spmm.cpp).  Four host threads, four
engines on device 0, different matrices, kernels and streams, plans built concurrently inside the threads, 50 interleaved SpMMs each:
every result bit-identical to cpu_spmm_CSR (sparse_helper.h:262-290), and every thread's error text stays its own."""
import threading

import numpy as np
import pytest

from util import ALPHA, BETA, NASA, random_csr

pytestmark = pytest.mark.gpu


def _cases():
    from sextans_amd import api, meshgen
    rp, ci, v, M, K, _ = api.read_suitsparse_matrix(NASA)
    yield "nasa4704 (column-major staging)", (rp, ci, v, M, K), 16, {}
    frp, fci, fv = api.gen_fem3d_host(20, 19, 18, 3, 5)
    Mf = 20 * 19 * 18 * 3
    yield "fem (grid bricks, tile loop)", (frp, fci, fv, Mf, Mf), 32, {}
    q = meshgen.permute_symmetric(frp, fci, fv, Mf, meshgen.node_permutation(Mf // 3, 3, 6))
    yield "fem random order (graph clustering, reordered form)", (*q, Mf, Mf), 16, {}
    rs = np.random.RandomState(4)
    urp, uci, uv = random_csr(rs, 30000, 50000, 14, long_rows=2)
    yield "random columns + long rows (gather kernel, piece path)", (urp, uci, uv, 30000, 50000), 24, {}


def test_four_engines_four_threads_one_device(sx, oracle):
    import torch
    from sextans_amd import api
    cases = list(_cases())
    wants, inputs = [], []
    for name, (rp, ci, v, M, K), N, _ in cases:
        rs = np.random.RandomState(len(name))
        B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
        w = C0.copy()
        oracle.spmm(M, N, K, ALPHA, rp, ci, v, B, BETA, w)
        wants.append(w); inputs.append((B, C0))
    errors, kernels = [], [None] * len(cases)
    start = threading.Barrier(len(cases))

    def worker(i):
        try:
            name, (rp, ci, v, M, K), N, opts = cases[i]
            B, C0 = inputs[i]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                tB = torch.from_numpy(B).cuda(); tC = torch.from_numpy(C0).cuda(); out = torch.empty_like(tC)
            stream.synchronize()
            start.wait()
            with api.Engine(0) as e:                      # created, prepared and used inside the thread
                for k, val in opts.items():
                    e.set_option(k, val)
                e.set_matrix_csr(M, K, rp, ci, v)
                for it in range(50):
                    if it % 10 == 0:
                        with torch.cuda.stream(stream):
                            out.fill_(float("nan"))
                    e.spmm_device(N, float(ALPHA), tB.data_ptr(), K, float(BETA), tC.data_ptr(), out.data_ptr(), M, stream.cuda_stream)
                    if it % 7 == i:                       # the host-buffer entry point (own stream, hipGraph replay) in between
                        h = C0.copy()
                        e.spmm(N, ALPHA, B, BETA, h, rp_time=3)
                        assert np.array_equal(h.view(np.uint32), wants[i].view(np.uint32)), (name, it, "host entry")
                    if it % 5 == 4:
                        stream.synchronize()
                        got = out.cpu().numpy()
                        assert np.array_equal(got.view(np.uint32), wants[i].view(np.uint32)), (name, it, e.last_kernel())
                    if it % 6 == 3:                       # the row-major entry point on the same engine (its first call builds what it needs:
                        with torch.cuda.stream(stream):   # allocations and a host sync while other threads capture and launch)
                            rB = tB.view(N, K).t().contiguous(); rC = tC.view(N, M).t().contiguous(); rO = torch.empty_like(rC)
                        e.spmm_device_rm(N, float(ALPHA), rB.data_ptr(), N, float(BETA), rC.data_ptr(), N, rO.data_ptr(), N, stream.cuda_stream)
                        stream.synchronize()
                        got = rO.t().contiguous().cpu().numpy().reshape(-1)
                        assert np.array_equal(got.view(np.uint32), wants[i].view(np.uint32)), (name, it, "row-major", e.last_kernel())
                # a failing call in this thread: the error text is thread-local
                with pytest.raises(api.SextansError):
                    e.set_option("no_such_option", 1)
                kernels[i] = e.last_kernel()
        except BaseException as ex:   # noqa: reported by the main thread
            errors.append((cases[i][0], repr(ex)))
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    assert all(k is not None for k in kernels) and len(set(kernels)) >= 3, kernels      # really different code paths side by side
