"""One ~470 k-row holdout instance END TO END through the reference's call surface: kron(T_100, nasa4704) written to a Matrix-Market
file (sextans_mtx_write, ~31 M entries) and run through `bin/sextans <A.mtx> <N> [rp_time]` -- loader, CPU golden, device SpMM,
verification (sextans-host.cpp:26-292).  The .mtx files go to /tmp (hundreds of MB); the CLI reports to the file named on the command line.
    python tools/holdout_cli.py gpurun_out/r05_cli_holdout.txt [n]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, ".")
from sextans_amd import holdout

out = open(sys.argv[1], "w")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cli = os.path.join("sextans_amd", "bin", "sextans")
for variant, N, rp in (("", 16, 20), ("", 128, 5), ("rect", 16, 20), ("unsym", 24, 5)):
    path = f"/tmp/holdout_kron{n}_{variant or 'sym'}.mtx"
    if not os.path.exists(path):
        t0 = time.time()
        rp_, ci, v, M, K = holdout.kron_host(n, variant)
        t1 = time.time()
        holdout.write_mtx(path, rp_, ci, v, M, K)
        print(f"# {path}: {M} x {K}, {len(ci)} entries, generated in {t1 - t0:.1f} s, written in {time.time() - t1:.1f} s, "
              f"{os.path.getsize(path) / 1e6:.0f} MB", file=out, flush=True)
        del rp_, ci, v
    t0 = time.time()
    r = subprocess.run([cli, path, str(N), str(rp)], capture_output=True, text=True)
    print(f"$ {cli} {path} {N} {rp}   (exit {r.returncode}, {time.time() - t0:.1f} s wall)", file=out)
    print(r.stdout + r.stderr, file=out, flush=True)
    assert r.returncode == 0 and "Success!" in r.stdout and "num_mismatch = 0," in r.stdout, r.stdout[-400:]
print("all runs: Success, 0 mismatches", file=out)
