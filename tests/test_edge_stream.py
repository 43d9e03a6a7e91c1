"""The accelerator's own buffer formats: scheduled non-zero stream, 64-bit words,
dense B / C channel layouts, container file.  CPU only.

Pinned against (a) committed outputs of the reference's generate_edge_list_for_all_PEs
(tests/golden/edges/, sparse_helper.h:345-403), (b) the live reference scheduler when oracle/_ref is
present, (c) numpy restatements of edge_list_64bit (sparse_helper.h:406-473) and of the channel loops
of sextans-host.cpp:152-195 (tests/util.py)."""
import glob
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import sextans_amd.api as sx
from oracle.bindings import Ref
from util import GOLDEN, NASA, bits_equal, chan_b_ref, chan_c_ref, edge_words, formula_B, formula_C

EDGES = os.path.join(GOLDEN, "edges")
FIXTURES = sorted(glob.glob(os.path.join(EDGES, "*.npz")))


def load_fixture(path):
    d = np.load(path)
    return {k: d[k] for k in d.files}


def test_fixtures_present():
    assert len(FIXTURES) >= 9


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_writer_matches_reference_scheduler_fixture(path):
    f = load_fixture(path)
    M, K = int(f["M"]), int(f["K"])
    e = sx.edges_pack_csc(M, K, f["csc_ptr"], f["csc_idx"], f["csc_val"])
    nw = len(f["ptr"]) - 1
    assert e["num_windows"] == nw == (K + 4095) // 4096
    assert e["num_a_len"] == int(f["ptr"][-1])
    assert e["nnz"] == len(f["csc_idx"])
    # edge_list_ptr_fpga: real entries, then zero padding to ((n+15)/16*16 + 1023)/1024*1024 ints
    assert e["edge_list_ptr"].size == ((nw + 1 + 15) // 16 * 16 + 1023) // 1024 * 1024
    assert np.array_equal(e["edge_list_ptr"][:nw + 1], f["ptr"])
    assert not e["edge_list_ptr"][nw + 1:].any()
    want = edge_words(f["ptr"], f["row"], f["col"], f["val"])
    assert e["channels"].shape == want.shape
    assert np.array_equal(e["channels"], want)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_reader_returns_csr_in_stream_order(path):
    f = load_fixture(path)
    M, K = int(f["M"]), int(f["K"])
    ch = edge_words(f["ptr"], f["row"], f["col"], f["val"])          # reference-side bytes
    rp, ci, v = sx.edges_decode_csr(f["ptr"], ch, len(f["ptr"]) - 1, M, K)
    nnz = len(f["csc_idx"])
    rp2, ci2, v2 = sx.CSC_2_CSR(M, K, nnz, f["csc_ptr"], f["csc_idx"], f["csc_val"])
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2) and bits_equal(v, v2)


def test_nasa4704_stream_known_answers():
    with open(os.path.join(EDGES, "nasa4704.json")) as fh:
        known = json.load(fh)
    cp, ri, cv, M, K, nnz = sx.read_suitsparse_matrix(NASA, sx.FMT_CSC)
    e = sx.edges_pack_csc(M, K, cp, ri, cv)
    nw = len(known["edge_list_ptr"]) - 1
    assert list(e["edge_list_ptr"][:nw + 1]) == known["edge_list_ptr"]
    assert e["channels"].shape == (8, known["chan_len"])
    got = [hashlib.sha256(e["channels"][c].tobytes()).hexdigest() for c in range(8)]
    assert got == known["channel_sha256"]
    L = e["num_a_len"]
    row18 = (e["channels"][:, :8 * L] >> np.uint64(32)) & np.uint64(0x3FFFF)
    assert int((row18 == 0x3FFFF).sum()) == known["bubbles"]
    assert 64 * L - known["bubbles"] == nnz == 104756


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("M,K,density,seed", [(1, 1, 1.0, 0), (5, 3, 0.5, 1), (100, 9000, 0.01, 2),
                                              (300, 300, 0.2, 3), (64, 64, 1.0, 4), (1000, 5000, 0.002, 5),
                                              (70, 4097, 0.05, 6), (130, 100, 0.0, 7), (4096, 64, 0.03, 8)])
def test_writer_matches_live_reference_scheduler(M, K, density, seed):
    rs = np.random.RandomState(seed)
    mask = rs.rand(K, M) < density                                     # [col, row]
    cp = np.zeros(K + 1, np.int32)
    cp[1:] = np.cumsum(mask.sum(1))
    ri = np.nonzero(mask)[1].astype(np.int32)
    cv = rs.uniform(-1, 1, ri.size).astype(np.float32)
    ptr, row, col, val = Ref().generate_edge_list(M, K, cp, ri, cv)
    e = sx.edges_pack_csc(M, K, cp, ri, cv)
    assert np.array_equal(e["edge_list_ptr"][:len(ptr)], ptr)
    assert np.array_equal(e["channels"], edge_words(ptr, row, col, val))


def test_word_layout_by_hand():
    """2 x 2, A = [[1.5, 0], [0, -2]] in CSC.  Row 0 -> PE 0 (channel 0, slot 0), row 1 -> PE 1
    (channel 1, slot 0); both PEs schedule their entry at slot 0; every other PE holds a bubble."""
    e = sx.edges_pack_csc(2, 2, np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32),
                          np.array([1.5, -2.0], np.float32))
    assert e["num_a_len"] == 1 and e["channels"].shape == (8, 512)
    bubble = 0x3FFFF << 32
    w0 = struct.unpack("<I", struct.pack("<f", 1.5))[0]                    # col 0, row 0
    w1 = (1 << 50) | struct.unpack("<I", struct.pack("<f", -2.0))[0]       # col 1, row 1 // 64 = 0
    ch = e["channels"]
    assert int(ch[0, 0]) == w0 and int(ch[1, 0]) == w1
    for c in range(8):
        for s in range(8):
            if (c, s) not in ((0, 0), (1, 0)):
                assert int(ch[c, s]) == bubble
    assert not ch[:, 8:].any()                                             # tail of the 512-word chunk


def test_pe_slot_is_bit_reversed():
    """Row r = PE r (r < 64): channel r % 8, slot bitrev3(r / 8): PE 8 -> slot 4, PE 24 -> slot 6."""
    M = 64
    cp = np.array([0, M], np.int32)
    ri = np.arange(M, dtype=np.int32)
    cv = np.arange(1, M + 1, dtype=np.float32)
    ch = sx.edges_pack_csc(M, 1, cp, ri, cv)["channels"]
    vals = (ch[:, :8] & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32)
    for r in range(M):
        slot = ((r // 8 & 1) << 2) | (r // 8 & 2) | (r // 8 >> 2 & 1)
        assert vals[r % 8, slot] == r + 1
    assert vals[0, 4] == 9 and vals[0, 6] == 25


def test_raw_distance_between_entries_of_one_row():
    """One row, 5 columns: slots 0, 10, 20, 30, 40 (DEP_DIST_LOAD_STORE = 10); a second row of the
    same PE fills the gaps in arrival order."""
    cp = np.array([0, 2, 3, 4, 5, 6], np.int32)
    ri = np.array([0, 64, 0, 0, 0, 0], np.int32)                           # row 64 is PE 0 too
    cv = np.array([1, 7, 2, 3, 4, 5], np.float32)
    e = sx.edges_pack_csc(65, 5, cp, ri, cv)
    assert e["num_a_len"] == 41
    pe0 = e["channels"][0, 0:8 * 41:8]
    vals = (pe0 & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32)
    rows = ((pe0 >> np.uint64(32)) & np.uint64(0x3FFFF)).astype(np.int64)
    assert [int(i) for i in np.nonzero(rows != 0x3FFFF)[0]] == [0, 1, 10, 20, 30, 40]
    assert list(vals[[0, 1, 10, 20, 30, 40]]) == [1, 7, 2, 3, 4, 5]
    assert rows[1] == 1


def test_reader_skips_every_word_with_row_bit_17():
    """The kernel tests row bit 17, not the full 0x3FFFF pattern (sextans.cpp:407)."""
    e = sx.edges_pack_csc(2, 2, np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32),
                          np.array([1.5, -2.0], np.float32))
    ch = e["channels"].copy()
    ch[2, 0] = np.uint64((0x20000 << 32) | 0x3F800000)                     # bit 17 set, nonzero payload
    rp, ci, v = sx.edges_decode_csr(e["edge_list_ptr"], ch, 1, 2, 2)
    assert list(rp) == [0, 1, 2] and list(ci) == [0, 1] and list(v) == [1.5, -2.0]


def test_reader_rejects_out_of_range_words():
    e = sx.edges_pack_csc(2, 2, np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32),
                          np.array([1.5, -2.0], np.float32))
    ch = e["channels"].copy()
    ch[2, 0] = np.uint64(0x3F800000)                                       # PE 2, row 2 >= M
    with pytest.raises(sx.SextansError) as ei:
        sx.edges_decode_csr(e["edge_list_ptr"], ch, 1, 2, 2)
    assert ei.value.code == 6
    ch = e["channels"].copy()
    ch[0, 0] = np.uint64((5 << 50) | 0x3F800000)                           # column 5 >= K
    with pytest.raises(sx.SextansError):
        sx.edges_decode_csr(e["edge_list_ptr"], ch, 1, 2, 2)
    with pytest.raises(sx.SextansError):                                   # window count must match K
        sx.edges_decode_csr(e["edge_list_ptr"], e["channels"], 2, 2, 2)
    bad_ptr = e["edge_list_ptr"].copy()
    bad_ptr[0] = 1
    with pytest.raises(sx.SextansError):
        sx.edges_decode_csr(bad_ptr, e["channels"], 1, 2, 2)


def test_writer_rejects_bad_input():
    with pytest.raises(sx.SextansError):                                   # row index out of range
        sx.edges_pack_csc(2, 1, np.array([0, 1], np.int32), np.array([2], np.int32), np.array([1], np.float32))
    with pytest.raises(sx.SextansError):                                   # col_ptr[K] != nnz
        sx.edges_pack_csc(2, 1, np.array([0, 2], np.int32), np.array([0], np.int32), np.array([1], np.float32))
    with pytest.raises(sx.SextansError):                                   # row field would alias the bubble bit
        sx.edges_pack_csc(64 * (1 << 17) + 1, 1, np.array([0, 0], np.int32), np.zeros(0, np.int32),
                          np.zeros(0, np.float32))


def test_container_round_trip(tmp_path):
    f = load_fixture(os.path.join(EDGES, "two_windows.npz"))
    e = sx.edges_pack_csc(int(f["M"]), int(f["K"]), f["csc_ptr"], f["csc_idx"], f["csc_val"])
    p = str(tmp_path / "a.sxe")
    sx.edges_save(p, e)
    assert os.path.getsize(p) == 48 + 4 * e["edge_list_ptr"].size + 8 * e["channels"].size
    with open(p, "rb") as fh:
        assert fh.read(8) == b"SXTEDGE1"
    g = sx.edges_load(p)
    for k in ("M", "K", "num_windows", "num_a_len", "nnz"):
        assert g[k] == e[k]
    assert np.array_equal(g["edge_list_ptr"], e["edge_list_ptr"]) and np.array_equal(g["channels"], e["channels"])
    with open(p, "r+b") as fh:
        fh.write(b"NOTEDGES")
    with pytest.raises(sx.SextansError) as ei:
        sx.edges_load(p)
    assert ei.value.code == 8
    with open(p, "wb") as fh:                                              # truncated
        fh.write(b"SXTEDGE1" + b"\0" * 10)
    with pytest.raises(sx.SextansError):
        sx.edges_load(p)
    with pytest.raises(sx.SextansError) as ei:
        sx.edges_load(str(tmp_path / "missing.sxe"))
    assert ei.value.code == 1


@pytest.mark.parametrize("K,N", [(1, 8), (17, 8), (100, 16), (129, 24), (4704, 16)])
@pytest.mark.parametrize("num_ch_b", [4, 8])
def test_b_channel_layout(K, N, num_ch_b):
    B = formula_B(K, N)
    ch = sx.chan_pack_b(K, N, B, num_ch_b)
    want = chan_b_ref(K, N, B, num_ch_b)
    assert ch.shape == want.shape and bits_equal(ch, want)
    assert bits_equal(sx.chan_unpack_b(K, N, ch), B)


@pytest.mark.parametrize("M,N", [(1, 8), (7, 8), (16, 16), (33, 24), (4704, 16)])
def test_c_channel_layout(M, N):
    Cm = formula_C(M, N)
    ch = sx.chan_pack_c(M, N, Cm)
    want = chan_c_ref(M, N, Cm)
    assert ch.shape == want.shape and bits_equal(ch, want)
    assert bits_equal(sx.chan_unpack_c(M, N, ch), Cm)


def test_channel_lengths_follow_the_host():
    L = sx.lib()
    assert L.sextans_chan_b_colsize(4704, 4) == 4704 * 2 and L.sextans_chan_b_colsize(4705, 4) == 4712 * 2
    assert L.sextans_chan_b_colsize(4705, 8) == 4720
    assert L.sextans_chan_b_len(4704, 16, 4) == (9408 * 2 + 1023) // 1024 * 1024
    assert L.sextans_chan_c_colsize(4705) == 4720
    assert L.sextans_chan_c_len(4704, 16) == (4704 * 2 + 1023) // 1024 * 1024
    with pytest.raises(sx.SextansError):
        sx.chan_pack_b(4, 12, np.zeros(48, np.float32), 4)                 # N not a multiple of 8


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_restatement_matches_reference_scheduler_fixture(oracle, path):
    """oracle/sextans_oracle.c:orc_edge_stream (linear probing, as the reference) against the committed outputs
    of the reference's scheduler; the product's writer (union-find) against the oracle."""
    f = load_fixture(path)
    M, K = int(f["M"]), int(f["K"])
    ptr, ch = oracle.edge_stream(M, K, f["csc_ptr"], f["csc_idx"], f["csc_val"])
    assert np.array_equal(ptr, f["ptr"])
    assert np.array_equal(ch, edge_words(f["ptr"], f["row"], f["col"], f["val"]))
    e = sx.edges_pack_csc(M, K, f["csc_ptr"], f["csc_idx"], f["csc_val"])
    assert np.array_equal(e["channels"], ch)


@pytest.mark.parametrize("seed", range(6))
def test_writer_matches_oracle_on_random_matrices(oracle, seed):
    rs = np.random.RandomState(100 + seed)
    M, K = int(rs.choice([3, 64, 200, 1500])), int(rs.choice([1, 70, 4096, 9000]))
    mask = rs.rand(K, M) < rs.choice([0.002, 0.02, 0.3])
    cp = np.zeros(K + 1, np.int32)
    cp[1:] = np.cumsum(mask.sum(1))
    ri = np.nonzero(mask)[1].astype(np.int32)
    cv = rs.uniform(-1, 1, ri.size).astype(np.float32)
    ptr, ch = oracle.edge_stream(M, K, cp, ri, cv)
    e = sx.edges_pack_csc(M, K, cp, ri, cv)
    assert np.array_equal(e["edge_list_ptr"][:len(ptr)], ptr) and np.array_equal(e["channels"], ch)
