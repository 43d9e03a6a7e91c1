"""Option sets compared on ONE box, round-robin: tools/ab_opts.py <fem dims AxBxCxD | synth:spec> <N> <iters> "k=v,k=v" "k=v" ...  (kernel us per launch)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
N = int(sys.argv[2]); iters = int(sys.argv[3])
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[4:]]
if sys.argv[1].startswith("synth:"):
    from sextans_amd import sweep
    M, K, p0, p1, p2, nnz = sweep._synth(sys.argv[1], 0)
    p = (p0, p1, p2, nnz)
else:
    dims = [int(x) for x in sys.argv[1].split("x")]
    M = K = dims[0] * dims[1] * dims[2] * dims[3]
    p = api.gen_fem3d_device(0, *dims, 3)
st = torch.cuda.current_stream().cuda_stream
B = torch.empty(K * N, device="cuda"); Cin = torch.empty(M * N, device="cuda"); Cout = torch.empty(M * N, device="cuda")
api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
engines = []
for o in sets:
    e = api.Engine(0)
    for k, v in o.items():
        e.set_option(k, v)
    e.set_matrix_csr_device(M, K, p[3], *p[:3])
    engines.append(e)
for rnd in range(3):
    out = []
    for o, e in zip(sets, engines):
        f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
        for _ in range(3): f()
        torch.cuda.synchronize(); t1 = time.time(); f(); torch.cuda.synchronize()
        for _ in range(min(300, int(0.06 / max(time.time() - t1, 1e-6)))): f()   # ~60 ms of warm-up (clock ramp after an idle phase)
        e.set_option("profile", 1); e.profile_reset()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(iters): f()
        torch.cuda.synchronize(); wall = (time.time() - t0) / iters
        k_ns, n, r_ns = e.profile_read(); p_ns, _ = e.profile_read_post(); e.set_option("profile", 0)
        out.append(f"{o}: kernel {k_ns / 1e3:.1f} pre {r_ns / 1e3:.1f} post {p_ns / 1e3:.1f} wall {wall * 1e6:.1f} ({e.last_kernel()})")
    print(f"N={N} round {rnd}: " + " | ".join(out), flush=True)
