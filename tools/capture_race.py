"""Which concurrent activity of ANOTHER host thread invalidates a hipGraph capture of the host-buffer entry point?  (round 5, found by
tests/test_concurrency_gpu.py)  Thread A loops sextans_spmm_host(rp_time = 3) on its own engine; thread B loops one activity."""
import sys, threading, time
sys.path.insert(0, ".")
import numpy as np
import torch
from sextans_amd import api, meshgen

import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
host = np.zeros(1024, np.float32)
rp, ci, v = api.gen_fem3d_host(20, 19, 18, 3, 5)
M = K = 20 * 19 * 18 * 3
q = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 6))
N = 16
rs = np.random.RandomState(0)
B = rs.uniform(-1, 1, K * N).astype(np.float32); C0 = rs.uniform(-1, 1, M * N).astype(np.float32)


def run(activity, a_matrix):
    stop, errs, fb = threading.Event(), [], [0]

    def A():
        try:
            with api.Engine(0) as e:
                e.set_matrix_csr(M, K, *a_matrix)
                for _ in range(60):
                    e.spmm(N, 0.85, B, -2.06, C0.copy(), rp_time=3)
                fb[0] = e.get_stat("graph_fallbacks")
        except Exception as ex:
            errs.append("A: " + repr(ex)[:200])
        stop.set()

    def Bt():
        try:
            with api.Engine(0) as e:
                e.set_matrix_csr(M, K, rp, ci, v)
                st = torch.cuda.Stream()
                tB = torch.from_numpy(B).cuda(); tC = torch.from_numpy(C0).cuda(); out = torch.empty_like(tC)
                torch.cuda.synchronize()
                while not stop.is_set():
                    if activity == "device_spmm":
                        e.spmm_device(N, 0.85, tB.data_ptr(), K, -2.06, tC.data_ptr(), out.data_ptr(), M, st.cuda_stream)
                    elif activity == "set_matrix":
                        e.set_matrix_csr(M, K, *q); e.spmm_device(N, 0.85, tB.data_ptr(), K, -2.06, tC.data_ptr(), out.data_ptr(), M, st.cuda_stream)
                    elif activity == "host_spmm":
                        e.spmm(N, 0.85, B, -2.06, C0.copy(), rp_time=3)
                    elif activity == "torch_copy":
                        out.cpu()
                    elif activity == "device_sync":
                        hip.hipDeviceSynchronize()
                    elif activity == "legacy_memcpy":      # what a caller's own code might do: breaks A's capture, A must fall back, not fail
                        hip.hipMemcpy(host.ctypes.data, tB.data_ptr(), 4096, 2)
                    elif activity == "idle":
                        time.sleep(0.001)
        except Exception as ex:
            errs.append("B: " + repr(ex)[:200])

    ta, tb = threading.Thread(target=A), threading.Thread(target=Bt)
    ta.start(); tb.start(); ta.join(); tb.join()
    print(f"{activity:12s} A on {'renumbered' if a_matrix is q else 'grid'} matrix: {'ok' if not errs else errs}  (A's graph fallbacks: {fb[0]:.0f} of 60)", flush=True)


for act in ("idle", "device_spmm", "torch_copy", "device_sync", "legacy_memcpy", "host_spmm", "set_matrix"):
    for am in ((rp, ci, v), q):
        run(act, am)
