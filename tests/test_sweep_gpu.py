"""The sweep harness over every .mtx fixture x several N, with the reference's
pass criterion and bit-identity against the host golden."""
import glob
import io
import json
import os

import pytest

from util import CASES, NASA

pytestmark = pytest.mark.gpu


def test_sweep_all_fixtures(sx):
    from sextans_amd import sweep
    paths = sorted(glob.glob(os.path.join(CASES, "*.mtx"))) + [NASA]
    buf = io.StringIO()
    recs = sweep.sweep(paths, [8, 16, 40], rp_time=3, check=True, out=buf)
    lines = [json.loads(l) for l in buf.getvalue().splitlines()]
    assert len(lines) == len(recs) and len(recs) >= 3 * (len(paths) - 1)
    for r in recs:
        assert "error" not in r, r
        assert r["N"] in (8, 16, 40) and r["ms"] > 0 and r["gflops"] >= 0 and 0 <= r["roofline_frac"] < 1
        if r["matrix"] != "number_formats":        # that fixture holds inf: NaN compare as mismatches
            assert r["passed"] and r["mismatch"] == 0 and r["bit_identical"], r
    nasa = [r for r in recs if r["matrix"] == "nasa4704" and r["N"] == 16][0]
    assert nasa["nnz"] == 104756 and nasa["kernel"] == "spmm_csr_panel"
