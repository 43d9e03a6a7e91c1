#!/usr/bin/env python3
"""Generate tests/golden/edges/ from the REFERENCE's own non-zero scheduler.

Runs only in the build container (needs /root/reference through oracle/_ref/libsextans_ref.so).  For
each input it stores the CSC arrays fed to, and the raw output of, generate_edge_list_for_all_PEs
(sparse_helper.h:345-403) called with the host's constants (64 PEs, window 4096, distance 10):

  edges/<name>.npz   M, K, csc_ptr, csc_idx, csc_val   input
                     ptr[num_windows+1]               edge_list_ptr
                     row, col [64, L] int16/int32, val [64, L]   scheduled slots, row == -1 = bubble
  edges/nasa4704.json  edge_list_ptr and sha256 of the 8 channel arrays for the shipped matrix

The 64-bit word packing (edge_list_64bit, sparse_helper.h:406-473) cannot be executed here (its
signature needs TAPA's allocator); tests apply tests/util.py:edge_words, a numpy restatement of that
function, to these scheduler outputs.  Only data is written.  Usage: python tests/golden/make_golden_edges.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.bindings import Ref  # noqa: E402
from util import CASES, NASA, edge_words  # noqa: E402

OUT = os.path.join(HERE, "edges")


def random_csc(seed, M, K, density, long_col=None):
    rs = np.random.RandomState(seed)
    cols = []
    ptr = np.zeros(K + 1, np.int32)
    for c in range(K):
        n = rs.binomial(M, density) if c != long_col else M
        cols.append(np.sort(rs.choice(M, size=n, replace=False)).astype(np.int32))
        ptr[c + 1] = ptr[c] + n
    idx = np.concatenate(cols) if cols else np.zeros(0, np.int32)
    val = rs.uniform(-1, 1, idx.size).astype(np.float32)
    return M, K, ptr, idx.astype(np.int32), val


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = Ref()
    inputs = {}
    for name in ("real_general", "pattern_symmetric", "empty_rows_long_row", "duplicates", "one_by_one",
                 "no_entries"):
        d = ref.load(os.path.join(CASES, name + ".mtx"))
        inputs[name] = (d["M"], d["K"], *d["csc"])
    inputs["two_windows"] = random_csc(11, 150, 4500, 0.004)            # K > 4096: second window
    inputs["dense_column"] = random_csc(12, 200, 70, 0.05, long_col=3)   # one PE row hit every column
    inputs["tall_one_pe"] = random_csc(13, 640, 40, 0.02)
    for name, (M, K, cp, ri, cv) in inputs.items():
        ptr, row, col, val = ref.generate_edge_list(M, K, cp, ri, cv)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), M=M, K=K, csc_ptr=cp, csc_idx=ri, csc_val=cv,
                            ptr=ptr, row=row.astype(np.int32), col=col.astype(np.int32), val=val)
        print(f"{name}: M={M} K={K} nnz={len(ri)} L={ptr[-1]}")
    d = ref.load(NASA)
    ptr, row, col, val = ref.generate_edge_list(d["M"], d["K"], *d["csc"])
    ch = edge_words(ptr, row, col, val)
    known = dict(edge_list_ptr=[int(x) for x in ptr], chan_len=int(ch.shape[1]),
                 channel_sha256=[hashlib.sha256(ch[c].tobytes()).hexdigest() for c in range(8)],
                 bubbles=int((row == -1).sum()))
    with open(os.path.join(OUT, "nasa4704.json"), "w") as f:
        json.dump(known, f, indent=1)
    print("nasa4704:", known["edge_list_ptr"], known["bubbles"])


if __name__ == "__main__":
    main()
