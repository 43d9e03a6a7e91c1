"""Packed-plan figures of the classes of DESIGN 9: blocks, dictionary rows, index / value stream entries with and without shared index lists."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api, sweep
st = torch.cuda.current_stream().cuda_stream
for spec in sys.argv[1:] or ["synth:fem3d:110:110:110:3", "synth:fem3d:160:160:160:1", "synth:stencil2d:1400:1400:9:2", "synth:femperm:110:110:110:3:random"]:
    M, K, p0, p1, p2, nnz = sweep._synth(spec, 0)
    N = 16
    B = torch.empty(K * N, device="cuda"); C = torch.empty(M * N, device="cuda")
    api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, C.data_ptr(), M * N, 42, st)
    for share in (0, 1):
        e = api.Engine(0); e.set_option("share_index", share)
        e.set_matrix_csr_device(M, K, nnz, p0, p1, p2)
        e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, C.data_ptr(), C.data_ptr(), M, st); torch.cuda.synchronize()
        g = e.get_stat
        idx, val = g("index_stream_entries"), g("value_stream_entries")
        print(f"{spec} share_index={share}: {e.last_kernel()} row_cluster={int(g('row_cluster'))} row_sets={int(g('row_sets'))} nnz={nnz} "
              f"value entries {val:.0f} index entries {idx:.0f} bytes/nnz {(4 * val + 2 * idx) / nnz:.2f} "
              f"(+ slot tables) panel rows natural {g('panel_rows_natural'):.0f} clustered {g('panel_rows_clustered'):.0f} device_bytes {g('device_bytes') / 1e9:.2f} GB plan_build_s {g('plan_build_s'):.3f}", flush=True)
        del e
    del B, C
