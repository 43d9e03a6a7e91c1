"""nasa4704 N = 16 / config-3 stand-in N = 128, rp_time = 1000 per-repeat time under option sets, one process, round-robin:  python tools/nasa_opts.py "k=v,k=v" ..."""
import os, sys
sys.path.insert(0, ".")
from sextans_amd import api
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
for name in ("nasa", "config3"):
    if name == "nasa":
        rp, ci, v, M, K, nnz = api.read_suitsparse_matrix("matrices/nasa4704/nasa4704.mtx"); N = 16
    else:
        rp, ci, v = api.gen_fem3d_host(35, 19, 7, 3, 2); M = K = 13965; N = 128
    engines = []
    for o in sets:
        e = api.Engine(0)
        for k, val in o.items():
            e.set_option(k, val)
        e.set_matrix_csr(M, K, rp, ci, v)
        engines.append(e)
    Bh, Ch = api.init_dense_B(K, N), api.init_dense_C(M, N)
    for rnd in range(3):
        out = []
        for o, e in zip(sets, engines):
            e.spmm(N, 0.85, Bh, -2.06, Ch.copy(), rp_time=100)
            ns = min(e.spmm(N, 0.85, Bh, -2.06, Ch.copy(), rp_time=1000) for _ in range(3))
            out.append(f"{o}: {ns / 1e6:.3f} us/repeat ({e.last_kernel()}, idx {e.get_stat('index_stream_entries'):.0f} / val {e.get_stat('value_stream_entries'):.0f})")
        print(f"{name} round {rnd}: " + " | ".join(out), flush=True)
