"""Multi-GPU form of the path: A row-range partitioned, B replicated, all-gather of C panels.

The reference is single-device; its only sharding is static (rows -> PEs by row % 64, B broadcast
down a daisy chain: sparse_helper.h:370, sextans.cpp:916-927).  The same independence of output
rows is what this module uses across GPUs: every rank owns a contiguous, nnz-balanced row range
of A, computes its M_g x N slab of C in place inside the full column-major C buffer, and one
all-gather (RCCL over xGMI when the backend is "nccl", gloo in the CPU tests) completes C on
every rank.  No reduction collective is needed (no K split).

Host-side logic here is pure numpy/torch.distributed and runs on CPU; the compute itself is the
HIP engine (sextans_amd.api.Engine) -- this module never computes SpMM.
"""
import numpy as np


def partition_rows_by_nnz(row_ptr, world):
    """Contiguous row ranges with (nearly) equal non-zero counts: split points by binary search in
    row_ptr.  Returns [(r0, r1)] * world, covering [0, M) in order; ranges may be empty."""
    row_ptr = np.asarray(row_ptr)
    M = len(row_ptr) - 1
    nnz = int(row_ptr[-1])
    cuts = [0]
    for g in range(1, world):
        target = (nnz * g) // world
        r = int(np.searchsorted(row_ptr, target, side="left"))
        r = min(max(r, cuts[-1]), M)
        cuts.append(r)
    cuts.append(M)
    return [(cuts[g], cuts[g + 1]) for g in range(world)]


def partition_rows_even(M, world):
    """Equal row counts (M must divide evenly for the in-place single-buffer all-gather)."""
    base, rem = divmod(M, world)
    out, r = [], 0
    for g in range(world):
        n = base + (1 if g < rem else 0)
        out.append((r, r + n))
        r += n
    return out


def balanced_ranges_from_even_slices(local_row_ptr, M, rank, group=None):
    """NNZ-balanced contiguous row ranges of a matrix NO rank holds whole: rank g passes the row_ptr (rebased to 0,
    int32 torch tensor, CPU or GPU) of the rows partition_rows_even(M, world)[g] -- with a counter-based generator or a
    row-sliced file every rank can produce exactly those -- the row lengths are all-gathered (4 bytes per row) and
    every rank finds the same cuts by binary search in the whole matrix's row_ptr (partition_rows_by_nnz).
    bench.py's default at N > 1."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    even = partition_rows_even(M, world)
    e0, e1 = even[rank]
    if local_row_ptr.numel() != e1 - e0 + 1:
        raise ValueError(f"rank {rank}: row_ptr of {local_row_ptr.numel() - 1} rows, the even slice has {e1 - e0}")
    width = max(b - a for a, b in even) + 1
    lens = torch.zeros(width, dtype=torch.int32, device=local_row_ptr.device)
    lens[:e1 - e0 + 1] = local_row_ptr.to(torch.int32)
    if world > 1:
        allp = [torch.empty_like(lens) for _ in range(world)]
        dist.all_gather(allp, lens, group=group)
    else:
        allp = [lens]
    row_ptr = np.zeros(M + 1, np.int64)
    for g, (a, b) in enumerate(even):
        seg = allp[g][:b - a + 1].cpu().numpy().astype(np.int64)
        row_ptr[a + 1:b + 1] = row_ptr[a] + seg[1:] - seg[0]
    return partition_rows_by_nnz(row_ptr, world)


def global_nnz(local_nnz, device="cpu", group=None):
    """Non-zeros of the whole matrix = sum of the ranks' local counts.  Handed to every rank's engine as option
    "global_nnz", it makes the automatic hub-split threshold ("split_rows" = -1) the one a single GPU holding all rows
    would choose, so the row-partitioned result equals the single-GPU result bit for bit also on power-law inputs."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(local_nnz)], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return int(t.item())


def slice_csr(row_ptr, col_idx, val, r0, r1):
    """Local CSR of rows [r0, r1): row_ptr rebased to 0, column indices unchanged (B is replicated)."""
    row_ptr = np.asarray(row_ptr)
    a, b = int(row_ptr[r0]), int(row_ptr[r1])
    return (row_ptr[r0:r1 + 1] - row_ptr[r0]).astype(np.int32), col_idx[a:b], val[a:b]


def all_gather_c(C_full, M, N, ranges, rank, group=None, _force=False):
    """(In-place per-column form; see SlabGather for the single-collective form bench.py uses.)
    Complete the column-major M x N matrix `C_full` (flat torch tensor, CPU or GPU) on every rank.

    On entry rank g has written rows ranges[g] of every column; on return all rows are present.
    C is column major, so a row slab is strided: the gather is done per column, in place (each
    rank's send buffer is a view of its own rows inside the receive column), and on the nccl
    backend the N per-column collectives are coalesced into one RCCL group launch.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1 and not _force:
        return
    lens = [r1 - r0 for r0, r1 in ranges]
    even = len(set(lens)) == 1 and lens[0] * world == M
    backend = dist.get_backend(group)
    cols = C_full.view(N, M)

    if even:
        r0, r1 = ranges[rank]

        def one_column(n):
            dist.all_gather_into_tensor(cols[n], cols[n][r0:r1], group=group)

        if backend == "nccl":
            try:
                with dist._coalescing_manager(group=group, device=C_full.device, async_ops=False):
                    for n in range(N):
                        one_column(n)
            except (AttributeError, TypeError, RuntimeError, NotImplementedError):
                for n in range(N):   # coalescing unavailable in this torch build: plain per-column calls
                    one_column(n)
        else:
            for n in range(N):
                one_column(n)
        return

    # Uneven (nnz-balanced) ranges: one packed collective of padded slabs, then scatter into place.
    import torch
    lmax = max(lens)
    r0, r1 = ranges[rank]
    send = torch.zeros((N, lmax), dtype=C_full.dtype, device=C_full.device)
    send[:, :r1 - r0] = cols[:, r0:r1]
    recv = torch.empty((world, N, lmax), dtype=C_full.dtype, device=C_full.device)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    for g, (a, b) in enumerate(ranges):
        if g != rank and b > a:
            cols[:, a:b] = recv[g, :, :b - a]


def all_gather_rows(C_rm, ranges, rank, group=None, _force=False):
    """Row-major form: complete the M x N tensor `C_rm` (contiguous, CPU or GPU) on every rank IN PLACE.  On entry rank g has written
    rows ranges[g] -- one contiguous run of a row-major matrix -- on return all rows are present.  Equal ranges: one
    all_gather_into_tensor whose send buffer is the rank's own run inside the receive tensor; nnz-balanced ranges: one broadcast per
    run (root = its owner).  The torch.distributed twin of sextans_dist_spmm_rm's exchange: nothing is packed or unpacked."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1 and not _force:
        return
    if not C_rm.is_contiguous():
        raise ValueError("all_gather_rows: C must be a contiguous row-major tensor (a padded leading dimension needs a packed copy)")
    M = C_rm.shape[0]
    lens = [r1 - r0 for r0, r1 in ranges]
    r0, r1 = ranges[rank]
    if len(set(lens)) == 1 and lens[0] * world == M:
        dist.all_gather_into_tensor(C_rm.view(-1), C_rm[r0:r1].reshape(-1), group=group)
        return
    for g, (a, b) in enumerate(ranges):
        if b > a:
            dist.broadcast(C_rm[a:b], src=dist.get_global_rank(group, g) if group is not None else g, group=group)


class SlabGather:
    """All-gather of C as ONE large collective (the form bench.py uses at N > 1).

    Column-major C makes a rank's row slab strided, and N per-column collectives of a few MB each
    are latency-bound on xGMI.  Instead every rank's SpMM writes a PACKED column-major slab
    (`ldc_out = lmax`, pointer `local_ptr()`) into its slot of a staging buffer S[world][N][lmax];
    a single in-place all_gather_into_tensor moves 4*N*lmax bytes per rank (32 MB for BASELINE
    config 4 at 8 GPUs); `unpack_into` then writes the full column-major C with one strided copy
    (HBM-local, ~0.1 ms for 256 MB).  Uneven (nnz-balanced) ranges are padded to the longest.
    """

    def __init__(self, M, N, ranges, rank, device, dtype=None):
        import torch
        self.M, self.N, self.ranges, self.rank = M, N, list(ranges), rank
        self.world = len(self.ranges)
        self.lens = [b - a for a, b in self.ranges]
        self.lmax = max(max(self.lens), 1)
        self.even = len(set(self.lens)) == 1 and self.lens[0] * self.world == M
        self.S = torch.zeros((self.world, N, self.lmax), dtype=dtype or torch.float32, device=device)

    def local_slab(self):
        """(N, lmax) view = column-major lmax x N slab of this rank (leading dimension lmax)."""
        return self.S[self.rank]

    def local_ptr(self):
        return self.S[self.rank].data_ptr()

    def gather(self, group=None, _force=False):
        import torch.distributed as dist
        if self.world > 1 or _force:
            dist.all_gather_into_tensor(self.S.view(-1), self.S[self.rank].reshape(-1), group=group)

    def unpack_into(self, C_full):
        """C_full: flat column-major M x N tensor on the same device."""
        cols = C_full.view(self.N, self.M)
        if self.even:
            cols.view(self.N, self.world, self.lmax).copy_(self.S.permute(1, 0, 2))
        else:
            for g, (a, b) in enumerate(self.ranges):
                if b > a:
                    cols[:, a:b] = self.S[g, :, :b - a]


class PipelinedSlabGather:
    """SlabGather with compute/communication overlap: the rank's slab is produced in `nchunks` row
    chunks; as soon as chunk c is computed its all-gather starts (async, on RCCL's stream) while the
    SpMM of chunk c+1 runs.  Per chunk one in-place all_gather_into_tensor of 4*N*(longest chunk c of any
    rank) bytes per rank.  Row ranges may be unequal (nnz-balanced): chunk c of every rank is padded to the
    longest chunk c.  `align(row) -> row' <= row` lets the caller snap this rank's interior cut positions to
    boundaries its kernels like (Engine.align_row: wavefront / row-block boundaries, so every chunk keeps the
    best kernel); the cut positions of all ranks are then exchanged once, here.

        pg = PipelinedSlabGather(M, N, ranges, rank, device, nchunks=4, align=lambda r: eng.align_row(N, r))
        pg.run(lambda c0, c1, out_ptr, ld_out, first: engine.spmm_device_rows(..., row_begin=c0, row_end=c1,
                                                                            reuse_b_panels=not first, ...))
        pg.finish(C_full)      # waits for the collectives, writes column-major C
    """

    def __init__(self, M, N, ranges, rank, device, nchunks=4, dtype=None, align=None, group=None):
        import torch
        self.M, self.N, self.ranges, self.rank = M, N, list(ranges), rank
        self.world = len(self.ranges)
        lens = [b - a for a, b in self.ranges]
        if sum(lens) != M or any(self.ranges[g][1] != self.ranges[g + 1][0] for g in range(self.world - 1)):
            raise ValueError("ranges must tile [0, M) in rank order")
        nchunks = max(1, min(nchunks, max(max(lens), 1)))

        def cuts_of(L, snap=None):
            c = [L * i // nchunks for i in range(nchunks + 1)]
            if snap is not None:
                for i in range(1, nchunks):
                    c[i] = min(max(int(snap(c[i])), c[i - 1]), L)
            return c
        all_cuts = [cuts_of(L) for L in lens]
        if align is not None:
            mine = cuts_of(lens[rank], align)
            if self.world > 1:
                import torch.distributed as dist
                t = torch.tensor(mine, dtype=torch.int64, device=device)
                buf = torch.empty(self.world * (nchunks + 1), dtype=torch.int64, device=device)
                dist.all_gather_into_tensor(buf, t, group=group)
                all_cuts = buf.view(self.world, nchunks + 1).cpu().tolist()
            else:
                all_cuts[rank] = mine
        self.cuts = all_cuts                                     # [rank][chunk] local row offsets
        self.chunks = [(all_cuts[rank][c], all_cuts[rank][c + 1]) for c in range(nchunks)]
        self.lmax = [max(max(all_cuts[g][c + 1] - all_cuts[g][c] for g in range(self.world)), 1)
                     for c in range(nchunks)]
        self.even = len(set(lens)) == 1 and all(all_cuts[g] == all_cuts[0] for g in range(self.world))
        self.L = lens[0] if self.even else None
        self.S = [torch.zeros((self.world, N, self.lmax[c]), dtype=dtype or torch.float32, device=device)
                  for c in range(nchunks)]
        self.works = []

    def run(self, compute_chunk, group=None, _force=False):
        """compute_chunk(c0, c1, out_ptr, ld_out, first) must enqueue the SpMM of local rows [c0, c1) writing a
        column-major (c1-c0) x N slab with leading dimension ld_out at out_ptr."""
        import torch.distributed as dist
        self.works = []
        first = True
        for (c0, c1), S, lmax in zip(self.chunks, self.S, self.lmax):
            if c1 > c0:
                compute_chunk(c0, c1, S[self.rank].data_ptr(), lmax, first)
                first = False
            if self.world > 1 or _force:
                self.works.append(dist.all_gather_into_tensor(S.view(-1), S[self.rank].reshape(-1),
                                                              group=group, async_op=True))

    def finish(self, C_full):
        """Chunk c is unpacked as soon as ITS all-gather has landed (Work.wait() only orders the current stream behind
        that collective), so the copies of chunks 0..n-2 run under the all-gathers still in flight and only the last
        chunk's copy is exposed."""
        works = self.works if self.works else [None] * len(self.S)
        self.works = []
        cols_even = C_full.view(self.N, self.world, self.L) if self.even else None
        cols = C_full.view(self.N, self.M)
        for c, (S, w) in enumerate(zip(self.S, works)):
            if w is not None:
                w.wait()
            if self.even:
                c0, c1 = self.chunks[c]
                if c1 > c0:
                    cols_even[:, :, c0:c1].copy_(S.permute(1, 0, 2))
                continue
            for g, (a, _) in enumerate(self.ranges):
                c0, c1 = self.cuts[g][c], self.cuts[g][c + 1]
                if c1 > c0:
                    cols[:, a + c0:a + c1] = S[g, :, :c1 - c0]
