"""Host logic of the packed row-bucketed form of A (panel_plan.cpp behind sextans_pack_csr): the
engine's analogue of the reference's packed non-zero stream.  CPU only."""
import numpy as np
import pytest

from util import NASA, random_csr


def check_invariants(P, M, K, rp, ci, v, lanes, min_reuse):
    RB = 256 // lanes
    cap = 36 * 1024 // (16 * lanes)
    br, dp, ro = P["blk_row"], P["dict_ptr"], P["row_off"]
    assert br[0] == 0 and br[-1] == M and np.all(np.diff(br) >= 1 if M else True)
    assert np.all(np.diff(br) <= RB)
    assert ro[0] == 0 and np.all(ro % 4 == 0)
    lens = np.diff(rp)
    assert np.array_equal(np.diff(ro), (lens + 3) // 4 * 4)                 # rows padded to 4 entries
    assert P["stream_len"] >= ro[-1] + 32                                   # tail padding for prefetch
    assert np.all(np.diff(dp) >= 0) and np.all(np.diff(dp) <= cap) and P["max_dict"] == (np.diff(dp).max() if len(dp) > 1 else 0)
    covered = 0
    for b in range(len(br) - 1):
        d = P["dict"][dp[b]:dp[b + 1]]
        j0, j1 = rp[br[b]], rp[br[b + 1]]
        cols = ci[j0:j1]
        if len(d):
            assert np.all(np.diff(d) > 0)                                   # ascending, distinct
            assert np.array_equal(d, np.unique(cols))                       # exactly the block's columns
            # (a block cut short by the end of its part of 64 row blocks keeps its dictionary whatever its reuse)
            remnant = br[b + 1] - br[b] < RB and (br[b + 1] % (64 * RB) == 0 or br[b + 1] == M)
            assert (j1 - j0) >= min_reuse * len(d) or remnant
            covered += j1 - j0
    assert covered == P["nnz_in_panel_blocks"]
    # the decoder gives back the CSR arrays bit for bit
    assert np.array_equal(P["decoded_col_idx"], ci) and np.array_equal(P["decoded_val"].view(np.uint32), v.view(np.uint32))
    # padding: value zero everywhere; inside dictionary rows it is -0.0f with index 0xFFFF (the kernel
    # multiplies it with a +1.0f panel row: x + (-0.0f) == x bit for bit), elsewhere +0.0f / index 0
    mask = np.ones(P["stream_len"], bool)
    dict_pad = np.zeros(P["stream_len"], bool)
    for b in range(len(br) - 1):
        for r in range(br[b], br[b + 1]):
            mask[ro[r]:ro[r] + lens[r]] = False
            if dp[b + 1] > dp[b]:
                dict_pad[ro[r] + lens[r]:ro[r + 1]] = True
    assert np.all(P["val"][mask] == 0)
    assert np.all(np.signbit(P["val"][dict_pad])) and np.all(P["idx16"][dict_pad] == 0xFFFF)
    assert not np.any(np.signbit(P["val"][mask & ~dict_pad])) and not np.any(P["idx16"][mask & ~dict_pad])


@pytest.mark.parametrize("lanes", [2, 4, 8])
@pytest.mark.parametrize("min_reuse", [0, 150, 400])
def test_pack_roundtrip_and_invariants(sx, lanes, min_reuse):
    from sextans_amd import api
    rs = np.random.RandomState(lanes * 10 + min_reuse)
    frp, fci, fv = api.gen_fem3d_host(9, 8, 6, 3, 7)
    rrp, rci, rv = random_csr(rs, 500, 1296, 15, long_rows=2)
    M, K = 1296 + 500, 1296
    rp = np.concatenate([frp, frp[-1] + rrp[1:]]).astype(np.int32)
    ci = np.concatenate([fci, rci]).astype(np.int32)
    v = np.concatenate([fv, rv]).astype(np.float32)
    P = api.pack_csr(M, K, rp, ci, v, lanes, min_reuse)
    check_invariants(P, M, K, rp, ci, v, lanes, min_reuse / 100.0)
    assert 0 < P["nnz_in_panel_blocks"] <= P["nnz"]


def test_pack_nasa4704_uses_dictionaries(sx):
    from sextans_amd import api
    rp, ci, v, M, K, nnz = sx.read_suitsparse_matrix(NASA)
    P = api.pack_csr(M, K, rp, ci, v, 4, 400)
    check_invariants(P, M, K, rp, ci, v, 4, 4.0)
    assert P["nnz_in_panel_blocks"] > 0.9 * nnz          # FEM structure: nearly every block has >= 4x reuse
    P0 = api.pack_csr(M, K, rp, ci, v, 4, 100000)        # impossible reuse threshold: all direct -- except the remnant
    remn = [b for b in range(P0["nblk"]) if P0["dict_ptr"][b + 1] > P0["dict_ptr"][b]]      # blocks at the ends of the parts
    assert all(P0["blk_row"][b + 1] % 4096 == 0 or P0["blk_row"][b + 1] == M for b in remn) and len(remn) <= 2
    assert P0["nnz_in_panel_blocks"] == sum(rp[P0["blk_row"][b + 1]] - rp[P0["blk_row"][b]] for b in remn) < 0.02 * nnz
    check_invariants(P0, M, K, rp, ci, v, 4, 1000.0)


def test_pack_degenerate(sx):
    from sextans_amd import api
    P = api.pack_csr(0, 5, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
    assert P["nblk"] in (0, 1) and P["nnz"] == 0
    rp = np.zeros(71, np.int32)                           # 70 empty rows
    P = api.pack_csr(70, 9, rp, np.zeros(0, np.int32), np.zeros(0, np.float32))
    assert P["blk_row"][-1] == 70 and P["nnz_in_panel_blocks"] == 0
    # a single row with more distinct columns than the panel holds -> its own direct block
    cols = np.arange(0, 4000, 2, dtype=np.int32)
    rp = np.array([0, 3, 3 + len(cols), 3 + len(cols) + 2], np.int32)
    ci = np.concatenate([[1, 5, 9], cols, [0, 7]]).astype(np.int32)
    v = np.arange(len(ci), dtype=np.float32) + 1
    P = api.pack_csr(3, 4000, rp, ci, v, 4, 0)
    check_invariants(P, 3, 4000, rp, ci, v, 4, 0.0)
    with pytest.raises(sx.SextansError):
        api.pack_csr(2, 3, np.array([0, 1, 2], np.int32), np.array([0, 3], np.int32), np.ones(2, np.float32))
    with pytest.raises(sx.SextansError):
        api.pack_csr(2, 3, np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32), np.ones(2, np.float32), lanes_per_row=3)
