"""Holdout classes of round 5 through the dispatcher AS IT IS: kron(T_850, nasa4704) in three numberings + the rectangular and the
unsymmetric-pattern variants, N = 16 (and 128 for the base class); one JSON record per run with the plan figures.
    python tools/holdout_r05.py [out.jsonl] [--n 850] [--skip-rcm]"""
import json
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
from sextans_amd import api, holdout

opts = dict(kv.split("=") for kv in sys.argv if "=" in kv and not kv.startswith("--"))
layout = "rm" if "--rm" in sys.argv else "cm"
out = open(sys.argv[1], "a") if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else sys.stdout
n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 850
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
cases = [("", "natural", 16), ("", "natural", 128), ("", "random", 16), ("rect", "natural", 16), ("unsym", "natural", 16),
         ("unsym", "random", 16)]
if "--skip-rcm" not in sys.argv:
    cases.append(("", "rcm", 16))
for variant, numbering, N in cases:
    t0 = time.time()
    M, K, p, i, v, nnz = holdout.kron_device(0, n, variant, numbering)
    gen_s = time.time() - t0
    e = api.Engine(0)
    for k, val in opts.items():
        e.set_option(k, int(val))
    e.set_matrix_csr_device(M, K, nnz, p, i, v)
    rec = bench._measure(api, torch, e, M, K, N, nnz, dev, st, 30, layout)
    rec.update(matrix=f"kron(T_{n}, nasa4704) {variant or 'sym'} {numbering}", gen_s=round(gen_s, 1), options=opts)
    for k in ("row_cluster", "cluster_decline", "panel_rows_natural", "panel_rows_clustered", "panel_blocks", "panel_blocks_clustered",
              "cluster_shared_fraction", "index_stream_entries", "value_stream_entries", "cluster_graph_kind", "pattern_symmetry", "panel_fraction", "piece_path_rows", "row_coherence", "device_bytes"):
        try:
            rec[k] = round(e.get_stat(k), 4)
        except Exception:
            pass
    e.close()
    for q in (p, i, v):
        api.device_free(0, q)
    torch.cuda.empty_cache()
    print(json.dumps(rec), file=out, flush=True)
