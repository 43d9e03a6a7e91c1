"""The reference's canonical run (nasa4704.mtx, N = 16; config 3 stand-in, N = 128) on several library builds, ONE box, round-robin,
each build in its own process:   python tools/nasa_ab.py lib1.so lib2.so ...
eager step = 1000 back-to-back sextans_spmm_device calls (wall / 1000); per repeat = sextans_spmm_host(rp_time = 1000) / 1000 (hipGraph replay)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import sextans_amd.api as api
    api.LIB_PATH = os.path.join(ROOT, sys.argv[2])
    import numpy as np
    import torch
    out = []
    for name in ("nasa4704 N=16", "config-3 stand-in N=128"):
        if name.startswith("nasa"):
            rp, ci, v, M, K, nnz = api.read_suitsparse_matrix(os.path.join(ROOT, "matrices", "nasa4704", "nasa4704.mtx")); N = 16
        else:
            rp, ci, v = api.gen_fem3d_host(35, 19, 7, 3, 2); M = K = 13965; nnz = len(ci); N = 128
        e = api.Engine(0)
        e.set_matrix_csr(M, K, rp, ci, v)
        st = torch.cuda.current_stream().cuda_stream
        B = torch.from_numpy(api.init_dense_B(K, N)).cuda(); Cin = torch.from_numpy(api.init_dense_C(M, N)).cuda(); Cout = torch.empty_like(Cin)
        f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
        for _ in range(2000): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(1000): f()
        torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 1000 * 1e6
        Bh, Ch = api.init_dense_B(K, N), api.init_dense_C(M, N)
        e.spmm(N, 0.85, Bh, -2.06, Ch.copy(), rp_time=100)
        ns = min(e.spmm(N, 0.85, Bh, -2.06, Ch.copy(), rp_time=1000) for _ in range(3))
        out.append(f"{name}: eager {eager:.2f} us/step, {ns / 1e6:.3f} us/repeat ({e.last_kernel()})")
        e.close()
    print(sys.argv[2].split("/")[-1].ljust(28), " | ".join(out), flush=True)
else:
    for rnd in range(3):
        for lib in sys.argv[1:]:
            subprocess.run([sys.executable, __file__, "--child", lib])
