#!/usr/bin/env python3
"""VERDICT r05 task 2, "measure first": share of a block's dictionary rows that the previous block of the walk holds too (stat
"dict_overlap_consecutive"), per matrix class, for the plan whole-matrix calls use (column-major and row-major prepares)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sextans_amd import api, holdout, meshgen  # noqa: E402

torch.cuda.set_device(0)


def cases():
    M, K, p, i, v, nnz = holdout.kron_device(0, 850, "", "natural"); yield "holdout kron(T_850, nasa4704) natural", M, K, p, i, v, nnz
    M, K, p, i, v, nnz = holdout.kron_device(0, 850, "", "random"); yield "holdout random numbering", M, K, p, i, v, nnz
    M = K = 110 ** 3 * 3; p, i, v, nnz = api.gen_fem3d_device(0, 110, 110, 110, 3, 3); yield "fem3d 110^3 x 3 natural", M, K, p, i, v, nnz
    q = api.permute_symmetric_device(0, M, nnz, p, i, v, meshgen.node_permutation(M // 3, 3, 1)); yield "fem3d 110^3 x 3 random node order", M, K, q[0], q[1], q[2], nnz
    M, K, p, i, v, nnz = holdout.kron_device(0, 850, "", "rcm"); yield "holdout RCM numbering", M, K, p, i, v, nnz


for name, M, K, p, i, v, nnz in cases():
    for rm in (False, True):
        e = api.Engine(0)
        e.set_matrix_csr_device(M, K, nnz, p, i, v)
        e.prepare(16, rowmajor=rm)
        print(json.dumps({"matrix": name, "prepare": "row-major" if rm else "column-major", "row_cluster": int(e.get_stat("row_cluster")),
                          "dict_overlap_consecutive": round(e.get_stat("dict_overlap_consecutive"), 4),
                          "cluster_run16_fraction": round(e.get_stat("cluster_run16_fraction"), 4),
                          "panel_rows_natural": int(e.get_stat("panel_rows_natural")), "panel_rows_clustered": int(e.get_stat("panel_rows_clustered")),
                          "blocks": int(e.get_stat("panel_blocks_clustered") if e.get_stat("row_cluster") > 0 else e.get_stat("panel_blocks"))}), flush=True)
        e.close()
