"""Randomised combinations of everything round 2 added around the kernels: kernel choice (auto / gather / panel /
window), lane counts, long-row bucketing and hub splitting thresholds, whole-matrix and row-range calls (cuts snapped
with sextans_align_row or deliberately unaligned), in-place C, alpha/beta -- on small matrices of four structures.
Rows that are not re-associated must be BIT-EXACT against the oracle's cpu_spmm_CSR; re-associated hub rows
(sextans_reassociated_rows) must meet |d| <= 1e-4 * (|alpha| * sum|a*b| + |beta*c|)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(rs, kind, M, K):
    if kind == "uniform":
        lens = rs.poisson(9, M)
    elif kind == "banded":
        lens = rs.poisson(14, M)
    elif kind == "hubs":
        lens = rs.poisson(6, M)
        lens[rs.choice(M, 3, replace=False)] = [min(K, 1500), min(K, 700), min(K, 260)]
    else:                                   # "blocky": groups of rows sharing columns (dictionary reuse)
        lens = rs.poisson(20, M)
    lens = np.minimum(lens, K)
    lens[rs.rand(M) < 0.05] = 0
    rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum(lens)
    ci = np.empty(rp[-1], np.int32)
    for i in range(M):
        n = lens[i]
        if not n:
            continue
        if kind == "banded":
            c = np.clip(i * K // M + rs.randint(-60, 61, n), 0, K - 1)
        elif kind == "blocky":
            c = ((i // 16) * 23 + rs.randint(0, 40, n)) % K
        else:
            c = rs.choice(K, size=n, replace=False) if n > 200 else rs.randint(0, K, n)
        ci[rp[i]:rp[i + 1]] = np.sort(c)
    return rp, ci, rs.uniform(-1, 1, rp[-1]).astype(np.float32)


CASES = []
_rs = np.random.RandomState(77)
for _i in range(int(os.environ.get("SEXTANS_COMBO_CASES", "48"))):     # soak: SEXTANS_COMBO_CASES=1000
    CASES.append((_i, ["uniform", "banded", "hubs", "blocky"][_i % 4], int(_rs.choice([700, 1500, 2600])),
                  int(_rs.choice([600, 3000, 9000])), int(_rs.choice([8, 16, 24, 32, 40])), int(_rs.choice([0, 1, 2, 3])),
                  int(_rs.choice([0, 0, 2, 4, 8])), int(_rs.choice([0, -1, 64, 300])), int(_rs.choice([0, -1, 16])),
                  int(_rs.choice([0, 1, 2]))))


@pytest.mark.parametrize("seed,kind,M,K,N,kernel,lpr,split,bucket,mode", CASES)
def test_combination(engine, oracle, seed, kind, M, K, N, kernel, lpr, split, bucket, mode):
    import torch
    rs = np.random.RandomState(1000 + seed)
    rp, ci, v = make(rs, kind, M, K)
    B = rs.uniform(-1, 1, K * N).astype(np.float32)
    C0 = rs.uniform(-1, 1, M * N).astype(np.float32)
    alpha, beta = np.float32(rs.choice([0.85, -1.5, 1.0])), np.float32(rs.choice([-2.06, 0.0, 1.0]))
    want = C0.copy()
    oracle.spmm(M, N, K, alpha, rp, ci, v, B, beta, want)
    for k, val in dict(lanes_per_row=lpr, stage_a=1, xcd_remap=1, exact=1, kernel=kernel, panel_min_reuse_x100=200, fuse_b=1,
                       split_rows=split, bucket_rows=bucket, window_rows=int(rs.choice([319, 100])), window_cols=int(rs.choice([65536, 512])),
                       window_unroll=int(rs.choice([4, 8])), mfma_dense_tiles=0).items():
        engine.set_option(k, val)
    engine.set_matrix_csr(M, K, rp, ci, v)
    hubs = engine.reassociated_rows()
    lens = np.diff(rp)
    if split == 0:
        assert len(hubs) == 0
    elif split > 0:
        assert np.array_equal(hubs, np.nonzero(lens > split)[0])
    st = torch.cuda.current_stream().cuda_stream
    dB = torch.from_numpy(B).cuda()
    dC = torch.from_numpy(C0).cuda()
    if mode == 0:                                        # whole matrix, host buffers
        out = C0.copy()
        engine.spmm(N, float(alpha), B, float(beta), out, rp_time=int(rs.choice([1, 3])))
    else:                                                # row ranges, in place; mode 1: aligned cuts, mode 2: arbitrary
        a, b = sorted(rs.randint(0, M + 1, 2))
        cuts = [0, engine.align_row(N, int(a)), engine.align_row(N, int(b)), M] if mode == 1 else [0, int(a), int(b), M]
        for i in range(3):
            c0, c1 = cuts[i], cuts[i + 1]
            engine.spmm_device_rows(N, float(alpha), dB.data_ptr(), K, float(beta), dC.data_ptr() + 4 * c0, M,
                                    dC.data_ptr() + 4 * c0, M, c0, c1, reuse_b_panels=i > 0, stream=st)
        torch.cuda.synchronize()
        out = dC.cpu().numpy()
    o2, w2 = out.reshape(N, M), want.reshape(N, M)
    plain = np.ones(M, bool)
    plain[hubs] = False
    assert np.array_equal(o2[:, plain].view(np.uint32), w2[:, plain].view(np.uint32)), engine.last_kernel()
    if len(hubs):
        rows = np.repeat(np.arange(M), lens)
        for n in range(N):
            bound = np.bincount(rows, weights=np.abs(v).astype(np.float64) * np.abs(B[n * K + ci]), minlength=M)
            bound = 1e-4 * (abs(float(alpha)) * bound + np.abs(float(beta) * C0[n * M:(n + 1) * M]))
            assert np.all(np.abs(o2[n].astype(np.float64) - w2[n]) <= bound + 1e-30)
    for k, val in dict(kernel=0, lanes_per_row=0, split_rows=0, bucket_rows=-1, window_rows=319, window_cols=65536,
                       window_unroll=8).items():
        engine.set_option(k, val)
