"""The sweep harness over every .mtx fixture x several N, with the reference's
pass criterion and bit-identity against the host golden."""
import glob
import io
import json
import os

import pytest

from util import CASES, NASA

pytestmark = pytest.mark.gpu


def test_sweep_all_fixtures(sx, oracle):
    import numpy as np
    from sextans_amd import sweep
    paths = sorted(glob.glob(os.path.join(CASES, "*.mtx"))) + [NASA]
    buf = io.StringIO()
    checked = []

    def against_oracle(rec, M, K, N, rp, ci, va, B, C0, C):
        """Every (matrix, N) result against the oracle's cpu_spmm_CSR restatement, bit for bit."""
        want = C0.copy()
        oracle.spmm(M, N, K, np.float32(0.85), rp, ci, va, B, np.float32(-2.06), want)
        nan = np.isnan(want)
        assert np.array_equal(nan, np.isnan(C)), rec
        assert np.array_equal(want.view(np.uint32)[~nan], C.view(np.uint32)[~nan]), rec
        checked.append((rec["matrix"], N))

    recs = sweep.sweep(paths, [8, 16, 40], rp_time=3, check=True, out=buf, inspect=against_oracle)
    assert len(checked) == len(recs)
    lines = [json.loads(l) for l in buf.getvalue().splitlines()]
    assert len(lines) == len(recs) and len(recs) >= 3 * (len(paths) - 1)
    for r in recs:
        assert "error" not in r, r
        assert r["N"] in (8, 16, 40) and r["ms"] > 0 and r["gflops"] >= 0 and 0 <= r["roofline_frac"] < 1
        if r["matrix"] != "number_formats":        # that fixture holds inf: NaN compare as mismatches
            assert r["passed"] and r["mismatch"] == 0 and r["bit_identical"], r
    nasa = [r for r in recs if r["matrix"] == "nasa4704" and r["N"] == 16][0]
    assert nasa["nnz"] == 104756 and nasa["kernel"].startswith("spmm_csr_panel")


def test_sweep_synthetic_classes(sx):
    """The synthetic classes of the sweep harness (uniform / banded / FEM / power-law), small sizes."""
    from sextans_amd import sweep
    buf = io.StringIO()
    recs = sweep.sweep_synthetic(["synth:uniform:20000:20", "synth:banded:20000:20:300", "synth:fem3d:12:12:12:3",
                                  "synth:powerlaw:30000:4:120:20000"], [8, 32], steps=3, out=buf,
                                 options={"split_rows": -1})   # opt in to re-associated hub rows (default: strict order)
    assert len(recs) == 8 and len(buf.getvalue().splitlines()) == 8
    by = {(r["matrix"].split(":")[1], r["N"]): r for r in recs}
    assert by[("fem3d", 32)]["kernel"].startswith("spmm_csr_panel") and by[("uniform", 8)]["kernel"] == "spmm_csr_rowgroup"
    assert by[("powerlaw", 8)]["kernel"].endswith("+hub_pieces") and by[("powerlaw", 8)]["reassociated_rows"] > 0
    assert by[("uniform", 8)]["piece_path_rows"] == 0
    assert all(r["ms"] > 0 and 0 < r["roofline_frac"] < 1 for r in recs)


def test_sweep_renumbered_classes(sx):
    """Round 4: the FEM matrix under random / RCM node orders (renumbered in HBM) and the unstructured meshes, with the row-order
    figures every synthetic record now carries."""
    from sextans_amd import sweep
    buf = io.StringIO()
    recs = sweep.sweep_synthetic(["synth:femperm:16:15:14:3:random", "synth:femperm:16:15:14:3:rcm", "synth:mesh3d:17:3:random",
                                  "synth:stencil2d:120:100:5:1"], [16, 32], steps=3, out=buf, options={"fuse_b": 0})   # (small B would be staged column-major)
    assert len(recs) == 8
    by = {(":".join(r["matrix"].split(":")[1:]), r["N"]): r for r in recs}
    r = by[("femperm:16:15:14:3:random", 16)]
    assert r["kernel"] == "spmm_csr_panel_v2_reordered" and r["row_cluster"] == 2
    assert 0 < r["panel_rows_clustered"] < 0.5 * r["panel_rows_natural"]
    assert by[("mesh3d:17:3:random", 32)]["kernel"] == "spmm_csr_panel_v2_reordered"
    assert by[("femperm:16:15:14:3:rcm", 16)]["kernel"].startswith("spmm_csr_panel")
    assert by[("stencil2d:120:100:5:1", 16)]["kernel"] == "spmm_csr_colwise"
    assert all(r["ms"] > 0 and 0 < r["roofline_frac"] < 1 for r in recs)
