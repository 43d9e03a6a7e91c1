"""The native multi-GPU entry points with world > 1 on ONE GPU ("Next round" 1b/1c).

BASELINE config 4 is stated as "A row-split across 8 x MI355X with RCCL all-gather(C)"; the pool gives single-GPU boxes, so until round 5
sextans_dist_spmm / _rm / _bell had only ever executed with world == 1.  Here every rank is a host thread with its own engine and stream
on device 0 and the collectives come from a loopback communicator (tests/fake_rccl.cpp: RCCL's entry points built from events and
device-to-device copies) bound through sextans_dist_bind_library: cut-list exchange, padding of unequal ranges, position -> row tables of
clustered-order chunks, grouped broadcasts, block-row ranges and the all-or-nothing status protocol of sextans_dist_prepare run exactly
as they would over xGMI.  Every rank's WHOLE C is compared bit for bit with the CPU oracle (cpu_spmm_CSR, sparse_helper.h:262-290).
The sharding matched: rows -> PEs with B broadcast (/root/reference/src/sparse_helper.h:370, src/sextans.cpp:916-927).

One subprocess per case (tests/loopback_worker.py): a rank that dies in front of a collective leaves its peers in a host barrier, and
only a process can be killed by a timeout."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(scenario, world, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(HERE, "loopback_worker.py"), scenario, str(world)], capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0 and f"LOOPBACK OK {scenario} world={world}" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]
    return r.stdout


@pytest.mark.parametrize("world", [2, 3, 8])
def test_dist_spmm_column_major_chunks(sx, world):
    """sextans_dist_spmm: 1 and 4 chunks, even and nnz-balanced ranges (unequal slabs padded to the longest), gather-kernel and LDS-panel
    matrices, lazily prepared and through sextans_dist_prepare (after which the calls exchange and synchronise nothing)."""
    _run("colmajor", world)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_dist_spmm_clustered_order_chunks(sx, world):
    """A mesh in a random node order: every rank's slab runs on its own graph-clustered plan, chunks are ranges of the plan's row blocks,
    slabs travel in clustered order and every rank scatters them through the SENDERS' position -> row tables (exchanged once)."""
    out = _run("colmajor_clustered", world, timeout=1500)
    assert "spmm_csr_panel_v2_reordered" in out


@pytest.mark.parametrize("world", [2, 3, 8])
def test_dist_spmm_row_major_in_place(sx, world):
    """sextans_dist_spmm_rm: equal ranges -> one in-place ncclAllGather, nnz-balanced ranges -> a group of ncclBroadcast; ldc == N and
    ldc > N (packed copy); C_in == C_out."""
    _run("rowmajor", world)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_dist_spmm_blocked_ell(sx, world):
    """sextans_dist_spmm_bell over unequal block-row ranges: bit-identical to the single-GPU call on the whole matrix."""
    _run("bell", world)


@pytest.mark.parametrize("world", [2, 4])
def test_dist_prepare_fails_on_every_rank_or_on_none(sx, world):
    """One rank's engine holds the wrong matrix: sextans_dist_prepare returns SEXTANS_ERR_INVALID there and SEXTANS_ERR_PEER on all the
    others -- nobody is left waiting in a collective -- and the same communicator prepares and runs correctly afterwards."""
    _run("errors", world)


@pytest.mark.parametrize("world", [3, 8])
def test_ranks_without_rows(sx, world):
    """Ragged partitions (the reference's tests hold empty and ragged inputs too): a 37-row matrix, mostly empty rows, over 3 and 8 ranks --
    nnz-balanced ranges with empty members, and everything on rank 0 -- through sextans_dist_prepare, sextans_dist_spmm (1 / 3 chunks) and
    sextans_dist_spmm_rm."""
    _run("empty_ranges", world)


def test_hub_rows_are_cut_at_the_global_threshold_on_every_rank(sx):
    """SEXTANS_MODE_FAST across ranks: the hub-split threshold follows the whole matrix's non-zero count, known only through the exchange
    of the ranks' counts -- 3 ranks of a power-law matrix give the single-GPU fast-mode result bit for bit, lazily and prepared."""
    _run("hub_rows", 3)


def test_config4_as_stated_row_split_over_8_ranks_at_full_size(sx):
    """BASELINE.json config 4: "Synthetic 4Mx4M CSR, ~0.001% density, N=16, A row-split across 8xMI355X with RCCL all-gather(C)" -- the full-size
    matrix, 8 ranks with nnz-balanced ranges, sextans_dist_prepare + sextans_dist_spmm (4 chunks) and sextans_dist_spmm_rm: every rank ends
    with the complete C, bit-identical to one engine holding all rows.  Ranks are threads on ONE GPU (loopback communicator): the
    configuration as stated, executed -- its timing needs the 8 GPUs."""
    _run("config4_full", 8, timeout=1800)


def test_config5_block_row_ranges_over_8_ranks_at_full_size(sx):
    """BASELINE config 5 (blocked-ELL 1M x 1M, 1 % block fill, N = 256, bf16 MFMA path) in 8 unequal block-row ranges: every
    rank's complete fp32 C bit-identical to the single-engine call -- same wavefront code on the same operands."""
    _run("config5_full", 8, timeout=1800)


def test_cpp_example_with_ranks_on_one_device(sx):
    """examples/dist_spmm.cpp (no Python, no torch in the data path) with 3 and 8 ranks as threads on device 0 over the loopback
    communicator: column-major and row-major forms against the single-GPU result."""
    sys.path.insert(0, HERE)
    from loopback_worker import fake_rccl_path
    exe = os.path.join(os.path.dirname(sx.api.CLI_PATH), "dist_spmm")
    nasa = os.path.join(os.path.dirname(HERE), "matrices", "nasa4704", "nasa4704.mtx")
    env = dict(os.environ, SEXTANS_RCCL_PATH=fake_rccl_path())
    for world in ("3", "8"):
        r = subprocess.run([exe, nasa, "24", world, "rm", "onedevice"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and f"{world} rank(s)" in r.stdout and "all ranks match the single-GPU result" in r.stdout, r.stdout + r.stderr
