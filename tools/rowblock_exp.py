#!/usr/bin/env python3
"""fp32 matrix-core path for dense row blocks ("mfma_dense_tiles" = 2) against the VALU kernels, same box, same matrices.

    python tools/rowblock_exp.py [fem3|fem6|blocks ...]  > gpurun_out/r06_rowblock_mfma.jsonl

One JSON record per (matrix, N, option set): step time, dominant kernel, share of the non-zeros routed, plan build seconds.
Option sets: strict (default), exact=0 (FMA on the VALU kernels), fast = exact 0 + rows routed to v_mfma_f32_16x16x4_f32 at the
stated fill threshold.  The last two give the same bits (tests/test_rowblock_mfma_gpu.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from sextans_amd import api  # noqa: E402


def dense_pattern(bs):
    rp = (np.arange(bs + 1) * bs).astype(np.int32)
    ci = np.tile(np.arange(bs, dtype=np.int32), bs)
    return rp, ci


def main():
    which = sys.argv[1:] or ["fem3", "fem6", "blocks"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    st = torch.cuda.current_stream().cuda_stream
    sets = [("strict", {}), ("exact0", {"exact": 0}),
            ("fast_thr50", {"exact": 0, "dense_tile_fill_x100": 50, "mfma_dense_tiles": 2}),
            ("fast_thr30", {"exact": 0, "dense_tile_fill_x100": 30, "mfma_dense_tiles": 2})]
    for w in which:
        if w == "fem3":
            dims = (110, 110, 110, 3); M = K = 110 ** 3 * 3
            p, i, v, nnz = api.gen_fem3d_device(0, *dims, 3); name = "fem3d 110^3 x 3 dof"; Ns = (32, 64, 128, 256)
        elif w == "fem6":
            dims = (80, 80, 80, 6); M = K = 80 ** 3 * 6
            p, i, v, nnz = api.gen_fem3d_device(0, *dims, 3); name = "fem3d 80^3 x 6 dof"; Ns = (16, 64, 128, 256)
        else:
            prp, pci = dense_pattern(32)
            n = 32768
            p, i, v, nnz, K = api.gen_kron_device(0, n, prp, pci, 32, 0, 7); M = n * 32
            name = "kron(T_32768, dense 32x32): block tridiagonal, fully dense blocks"; Ns = (16, 64, 128, 256)
        for N in Ns:
            for label, opts in sets:
                if w == "fem3" and label == "fast_thr50":
                    continue          # (fill 0.34: nothing routed)
                e = api.Engine(0)
                for k, val in opts.items():
                    e.set_option(k, val)
                e.set_matrix_csr_device(M, K, nnz, p, i, v)
                try:
                    r = bench._measure(api, torch, e, M, K, N, nnz, dev, st, 10 if N >= 128 else 30)
                    r.update(matrix=name, set=label, options=opts, routed_fraction=round(e.get_stat("dense_tile_fraction"), 4) if opts.get("mfma_dense_tiles") else 0.0,
                             tflops=round(2.0 * N * nnz / (r["us_per_step"] * 1e-6) / 1e12, 2))
                except Exception as ex:   # noqa: BLE001
                    r = {"matrix": name, "N": N, "set": label, "error": str(ex)}
                e.close()
                print(json.dumps(r), flush=True)
        for q in (p, i, v):
            api.device_free(0, q)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
