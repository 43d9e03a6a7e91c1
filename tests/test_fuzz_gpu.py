"""Randomised structure sweep: every kernel path (row-group, panel mixed / dictionary-only) against the
oracle's cpu_spmm_CSR restatement, BIT-EXACT, on matrices whose shape parameters are drawn so that the
corner cases of the packed stream are hit: dictionaries smaller and larger than a stride chunk, blocks cut
by the dictionary capacity, row lengths on every residue mod 4 and mod the batch, empty rows and blocks,
duplicate columns, K smaller than a row block, values with signed zeros / infinities."""
import numpy as np
import pytest

from util import bits_equal

pytestmark = pytest.mark.gpu


def make_matrix(rs, kind, M, K, mean):
    lens = rs.poisson(mean, M)
    lens[rs.rand(M) < 0.1] = 0
    if kind == "tails":                     # every residue of the row length mod 16
        lens = (np.arange(M) % 37)
    lens = np.minimum(lens, 4 * K)
    rp = np.zeros(M + 1, np.int32)
    rp[1:] = np.cumsum(lens)
    ci = np.empty(rp[-1], np.int32)
    for i in range(M):
        n = lens[i]
        if not n:
            continue
        if kind in ("banded", "tails"):     # reuse: columns near the diagonal, duplicates allowed
            c = (i * K // max(M, 1) + rs.randint(-12, 13, n)) % K
        elif kind == "blocky":              # a few shared columns per group of rows
            base = (i // 8) * 5 % K
            c = (base + rs.randint(0, 6, n)) % K
        else:                               # uniform: no reuse
            c = rs.randint(0, K, n)
        ci[rp[i]:rp[i + 1]] = np.sort(c)
    v = rs.uniform(-1, 1, rp[-1]).astype(np.float32)
    if rp[-1] > 8:
        v[rs.randint(0, rp[-1], 3)] = [0.0, -0.0, np.float32(1e-40)]
    return rp, ci.astype(np.int32), v


CASES = []
_rs = np.random.RandomState(2024)
for _i in range(int(__import__("os").environ.get("SEXTANS_FUZZ_CASES", "36"))):     # soak: SEXTANS_FUZZ_CASES=1000
    CASES.append((_i, ["banded", "blocky", "uniform", "tails"][_i % 4], int(_rs.choice([1, 7, 63, 64, 65, 200, 777, 2500])),
                  int(_rs.choice([1, 5, 40, 300, 5000])), float(_rs.choice([0.7, 3, 11, 30])),
                  int(_rs.choice([8, 16, 24, 40])), int(_rs.choice([2, 4, 8])), int(_rs.choice([0, 0, 400, 100000]))))


@pytest.mark.parametrize("seed,kind,M,K,mean,N,lpr,min_reuse", CASES)
def test_structure_sweep(engine, oracle, seed, kind, M, K, mean, N, lpr, min_reuse):
    rs = np.random.RandomState(seed)
    rp, ci, v = make_matrix(rs, kind, M, K, mean)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    if K * N > 4:
        B[rs.randint(0, K * N, 2)] = [np.float32(-0.0), np.float32(np.inf)]
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    alpha, beta = np.float32(rs.choice([0.85, -1.5, 1.0])), np.float32(rs.choice([-2.06, 0.0, 1.0]))
    want = C0.copy()
    oracle.spmm(M, N, K, alpha, rp, ci, v, B, beta, want)
    # kernel 1 = row-group gather; kernel 2 = panel, staging B from the repacked panel (fuse_b 0) or straight
    # from the caller's column-major B (fuse_b 1, the default for small matrices)
    for kernel, fuse_b in ((1, 1), (2, 0), (2, 1)):
        for k, val in dict(lanes_per_row=lpr, stage_a=1, xcd_remap=1, exact=1, kernel=kernel, fuse_b=fuse_b,
                           panel_min_reuse_x100=min_reuse, split_rows=0, bucket_rows=int(rs.choice([0, -1]))).items():
            engine.set_option(k, val)
        engine.set_matrix_csr(M, K, rp, ci, v)
        out = C0.copy()
        engine.spmm(N, float(alpha), B, float(beta), out, rp_time=int(rs.choice([1, 2])))
        assert bits_equal(out, want), (kernel, fuse_b, engine.last_kernel())
    engine.set_option("fuse_b", 1)
