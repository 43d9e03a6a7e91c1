"""A/B timing on ONE box: tools/ab.py <fem dims AxBxCxD | synth:spec> <N-list> <iters> lib1.so lib2.so ... -- each library in its own subprocess, round-robin, several rounds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sextans_amd.api as api
    api.LIB_PATH = os.path.join(ROOT, sys.argv[2])
    import torch
    Ns = [int(x) for x in sys.argv[4].split(",")]
    if sys.argv[3].startswith("synth:"):
        from sextans_amd import sweep
        M, K, p0, p1, p2, nnz = sweep._synth(sys.argv[3], 0)
        p = (p0, p1, p2, nnz)
    else:
        dims = [int(x) for x in sys.argv[3].split("x")]
        M = K = dims[0] * dims[1] * dims[2] * dims[3]
        p = api.gen_fem3d_device(0, *dims, 3)
    e = api.Engine(0)
    for kv in [x for x in os.environ.get('SX_AB_OPTS', '').split(',') if x]:
        e.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    e.set_matrix_csr_device(M, K, p[3], *p[:3])
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for N in Ns:
        B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
        api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
        f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
        for _ in range(5): f()
        e.set_option("profile", 1); e.profile_reset()
        for _ in range(int(sys.argv[5])): f()
        torch.cuda.synchronize()
        k_ns, n, r_ns = e.profile_read(); e.set_option("profile", 0)
        res.append(f"N={N}: {k_ns / 1e3:.1f}")
        del B, Cin, Cout
    print(sys.argv[2].split("/")[-1], " ".join(res), flush=True)
else:
    dims, Ns, iters = sys.argv[1], sys.argv[2], sys.argv[3]
    libs = sys.argv[4:]
    for rnd in range(3):
        for lib in libs:
            subprocess.run([sys.executable, __file__, "--child", lib, dims, Ns, iters])
