"""sextans_invoke: the tapa::invoke(Sextans, ...) argument list (sextans-host.cpp:237-251) on buffers
prepared exactly as the reference host prepares them (edge stream, B / C channels), checked BIT-EXACT
against cpu_spmm_CSR (the oracle's restatement; committed reference goldens for nasa4704) re-expressed in
the accelerator's C layout (sextans-host.cpp:264-270)."""
import glob
import hashlib
import os
import struct

import numpy as np
import pytest

import sextans_amd.api as api
from util import (ALPHA, BETA, GOLDEN, NASA, bits_equal, chan_b_ref, chan_c_ref, edge_words, formula_B,
                  formula_C)

pytestmark = pytest.mark.gpu
EDGES = os.path.join(GOLDEN, "edges")
FIXTURES = sorted(glob.glob(os.path.join(EDGES, "*.npz")))


def f32_bits(x):
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def expected_channels(oracle, M, K, N, rp, ci, v, alpha, B, beta, C0):
    want = C0.copy()
    oracle.spmm(M, N, K, np.float32(alpha), rp, ci, v, B, np.float32(beta), want)
    ch = chan_c_ref(M, N, want)
    cs = (M + 15) // 16 * 16
    pad = np.float32(alpha) * np.float32(0) + np.float32(beta) * np.float32(0)
    m = np.arange(M, cs)
    for nn in range(N):                                         # rows M .. colsize-1 are written too
        ch[m % 8, cs * (nn // 8) + (m // 8) * 8 + nn % 8] = pad
    return ch, cs * (N // 8)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
@pytest.mark.parametrize("N,num_ch_b", [(8, 4), (24, 8)])
def test_invoke_on_reference_prepared_buffers(engine, oracle, path, N, num_ch_b):
    d = np.load(path)
    M, K = int(d["M"]), int(d["K"])
    ach = edge_words(d["ptr"], d["row"], d["col"], d["val"])    # bytes as the reference host lays them out
    nw = len(d["ptr"]) - 1
    ptr = np.zeros(1024, np.int32)
    ptr[:nw + 1] = d["ptr"]
    B, C0 = formula_B(K, N), formula_C(M, N)
    bch, cin = chan_b_ref(K, N, B, num_ch_b), chan_c_ref(M, N, C0)
    out, ns = engine.invoke(ptr, ach, bch, cin, nw, int(d["ptr"][-1]), M, K, (1 << 16) | N, f32_bits(ALPHA),
                            f32_bits(BETA))
    rp, ci, v = api.CSC_2_CSR(M, K, len(d["csc_idx"]), d["csc_ptr"], d["csc_idx"], d["csc_val"])
    want, used = expected_channels(oracle, M, K, N, rp, ci, v, ALPHA, B, BETA, C0)
    assert ns > 0
    assert bits_equal(out[:, :used], want[:, :used])
    assert not out[:, used:].any()                              # beyond colsize * N/8 nothing is written


def test_invoke_nasa4704_canonical(engine, sx):
    """The reference's shipped run through the accelerator formats: golden C sha256."""
    cp, ri, cv, M, K, nnz = sx.read_suitsparse_matrix(NASA, api.FMT_CSC)
    e = api.edges_pack_csc(M, K, cp, ri, cv)
    N = 16
    bch = api.chan_pack_b(K, N, sx.init_dense_B(K, N), 4)
    cin = api.chan_pack_c(M, N, sx.init_dense_C(M, N))
    for rp_time in (1, 3):
        out, ns = engine.invoke(e["edge_list_ptr"], e["channels"], bch, cin, e["num_windows"], e["num_a_len"],
                                M, K, (rp_time << 16) | N, f32_bits(ALPHA), f32_bits(BETA))
        Cm = api.chan_unpack_c(M, N, out)
        assert hashlib.sha256(Cm.tobytes()).hexdigest() == \
            "988205f823683783aea5cd8c7eb0f88846b59bd957d3429ba63e2110bd3ad88f"
    # matrix stays resident: edge_list_ptr = None reuses it with new dense operands
    g = np.load(os.path.join(GOLDEN, "nasa4704_N16.npz"))
    bch = api.chan_pack_b(K, N, formula_B(K, N), 8)
    cin = api.chan_pack_c(M, N, formula_C(M, N))
    out, _ = engine.invoke(None, None, bch, cin, e["num_windows"], e["num_a_len"], M, K, (1 << 16) | N,
                           f32_bits(ALPHA), f32_bits(BETA))
    assert bits_equal(api.chan_unpack_c(M, N, out), g["C_formula"])
    with pytest.raises(api.SextansError):                       # resident matrix has another shape
        engine.invoke(None, None, bch, cin, e["num_windows"], e["num_a_len"], M + 1, K, (1 << 16) | N,
                      f32_bits(ALPHA), f32_bits(BETA))


@pytest.mark.parametrize("alpha,beta", [(-0.5, -1.25), (0.0, 1.0), (2.0, 0.0)])
def test_invoke_padding_rows_and_signs(engine, oracle, alpha, beta):
    """M = 150 pads to 160 rows: the accelerator writes alpha*0 + beta*0 there (-0.0 when both are
    negative)."""
    d = np.load(os.path.join(EDGES, "two_windows.npz"))
    M, K, N = int(d["M"]), int(d["K"]), 16
    e = api.edges_pack_csc(M, K, d["csc_ptr"], d["csc_idx"], d["csc_val"])
    B, C0 = formula_B(K, N), formula_C(M, N)
    out, _ = engine.invoke(e["edge_list_ptr"], e["channels"], api.chan_pack_b(K, N, B, 4),
                           api.chan_pack_c(M, N, C0), e["num_windows"], e["num_a_len"], M, K, (1 << 16) | N,
                           f32_bits(alpha), f32_bits(beta))
    rp, ci, v = api.CSC_2_CSR(M, K, len(d["csc_idx"]), d["csc_ptr"], d["csc_idx"], d["csc_val"])
    want, used = expected_channels(oracle, M, K, N, rp, ci, v, alpha, B, beta, C0)
    assert bits_equal(out[:, :used], want[:, :used])
    if alpha < 0 and beta < 0:
        assert np.signbit(out[150 % 8, (150 // 8) * 8]) and out[150 % 8, (150 // 8) * 8] == 0


def test_invoke_argument_errors(engine):
    d = np.load(os.path.join(EDGES, "duplicates.npz"))
    M, K, N = int(d["M"]), int(d["K"]), 8
    e = api.edges_pack_csc(M, K, d["csc_ptr"], d["csc_idx"], d["csc_val"])
    bch, cin = api.chan_pack_b(K, N, formula_B(K, N), 4), api.chan_pack_c(M, N, formula_C(M, N))
    args = (e["edge_list_ptr"], e["channels"], bch, cin, e["num_windows"])
    with pytest.raises(api.SextansError):                       # NUM_A_LEN disagrees with edge_list_ptr
        engine.invoke(*args, e["num_a_len"] + 1, M, K, (1 << 16) | N, 0, 0)
    with pytest.raises(api.SextansError):                       # N = 12 is not a multiple of 8
        engine.invoke(*args, e["num_a_len"], M, K, (1 << 16) | 12, 0, 0)
    with pytest.raises(api.SextansError):                       # stream addresses rows >= M
        engine.invoke(*args, e["num_a_len"], M - 40, K, (1 << 16) | N, 0, 0)


def test_cli_through_fpga_buffers(sx):
    """SEXTANS_FPGA_BUFFERS=1: the CLI prepares the edge stream and channel buffers like the reference's
    main() (sextans-host.cpp:114-204), enters through sextans_invoke and reads C back out of the channels."""
    import subprocess
    env = dict(os.environ, SEXTANS_FPGA_BUFFERS="1")
    for argv, needles in ((["16"], ("N = 16", "num_mismatch = 0, percent = 0.00%")),
                          (["100", "3", "1.25", "0.5"], ("N = 104", "num_mismatch = 0"))):
        r = subprocess.run([sx.api.CLI_PATH, NASA] + argv, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        for needle in ("Preparing sparse A for FPGA ...done", "Preparing dense B for FPGA ...",
                       "Preparing dense C for FPGA ...done", "launch kernel", "Success!") + needles:
            assert needle in r.stdout, (needle, r.stdout)
