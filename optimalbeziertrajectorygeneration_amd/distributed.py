"""Multi-GPU execution of the constraint sweeps: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU rehearsal in tests).

Two independent ways to shard:

  * batch sharding (default, NO collective): the B = n_x + 1 evaluation rows of one SLSQP
    Jacobian are independent; rank r takes a contiguous block of rows.  Results go back to
    the host as Jacobian column blocks.  `shard_rows`; `RowShardedFdStep` runs a rank's block of one
    finite-difference step (structured or brute force) on the GPU.
  * pair partitioning (one evaluation across GPUs, the 256-vehicle case): every rank holds
    all control points (65 KB), owns a contiguous balanced block of the lexicographic pair
    list -- and of the gjkNew hull pair list, "GJK pairs partition the same way" -- and ONE
    all-gather assembles either the full constraint vector or only the per-pair minima
    (separation minima of the Bernstein family, gjkNew's distance and flag of the hull family).
    On a fully connected 8-GPU xGMI node each rank pushes its shard to 7 peers concurrently, so
    the collective is latency-bound at these sizes (1 MB shards).  `PairPartitionedSweep`,
    `all_gather_pair_blocks`.

The compute itself is injected (`evaluate(pair_begin, pair_count) -> tensor[B, count*width]`)
so that the orchestration can be rehearsed on CPU with gloo; on GPUs the evaluators are
closures over `Context.temporal_sep[_min]_dev` / `Context.gjk_swarm_dev`.
"""
import torch
import torch.distributed as dist


def partition(n_items, world):
    """Contiguous balanced blocks: [(begin, count)] * world; the first n_items % world blocks
    get one extra item."""
    base, extra = divmod(n_items, world)
    out, b = [], 0
    for r in range(world):
        c = base + (1 if r < extra else 0)
        out.append((b, c))
        b += c
    return out


def shard_rows(B, world, rank):
    """Batch sharding: the (begin, count) of evaluation rows owned by `rank`."""
    return partition(B, world)[rank]


def _world_rank(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def all_gather_pair_blocks(parts, group=None, force=False):
    """ONE all-gather for several partitioned results.

    parts: list of (mine, blocks, width): `mine` is this rank's tensor [B, count_r*width] (any
    dtype), `blocks` the [(begin, count)] partition of that result's pair list over the ranks.
    The blocks of all parts are padded to their largest count, byte-packed (8-byte items first so
    that every segment stays aligned) and exchanged with a single `all_gather_into_tensor`;
    returns the assembled [B, n_pairs*width] tensors, identical on every rank."""
    world, rank = _world_rank(group)
    if world == 1 and not (force and dist.is_initialized()):      # force: run the collective even alone (rehearsals)
        return [p[0] for p in parts]
    order = sorted(range(len(parts)), key=lambda i: -parts[i][0].element_size())
    segs, meta = [], []
    for i in order:
        mine, blocks, w = parts[i]
        B = mine.shape[0]
        cmax = max(c for _, c in blocks)
        pad = torch.zeros((B, cmax * w), dtype=mine.dtype, device=mine.device)
        pad[:, :mine.shape[1]] = mine
        segs.append(pad.view(torch.uint8).reshape(-1))
        meta.append((i, B, cmax, w, mine.dtype, blocks))
    send = torch.cat(segs)
    recv = torch.empty((world, send.numel()), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv.view(-1), send, group=group)
    out = [None] * len(parts)
    off = 0
    for (i, B, cmax, w, dtype, blocks) in meta:
        nbytes = B * cmax * w * torch.empty((), dtype=dtype).element_size()
        seg = recv[:, off:off + nbytes].contiguous().view(dtype).view(world, B, cmax * w)
        out[i] = torch.cat([seg[r, :, :c * w] for r, (_, c) in enumerate(blocks)], dim=1).contiguous()
        off += nbytes
    return out


def fd_row_vehicle(rows, n_veh, dim, n_free_cols):
    """Which vehicle batch row b of a finite-difference batch advances (obtg_fd_view_begin: row 0 is x, row b >= 1 advances free
    control point b - 1, numbered row-major over (coordinate row, free column) as x.reshape(numRows, numCols) in
    optimization.py:283): -> int64 tensor, -1 for row 0."""
    rows = torch.as_tensor(rows, dtype=torch.int64)
    k = (rows - 1) % (n_veh * dim * n_free_cols)
    veh = k // (dim * n_free_cols)
    return torch.where(rows > 0, veh, torch.full_like(veh, -1))


def pairs_of_vehicle(n_obj):
    """[n_obj][n_obj - 1] lexicographic pair indices of the pairs that contain object v (partners ascending) -- the only
    separation pairs a control point of vehicle v moves."""
    v = torch.arange(n_obj)[:, None]
    q = torch.arange(n_obj - 1)[None, :]
    partner = q + (q >= v).long()
    lo, hi = torch.minimum(v, partner), torch.maximum(v, partner)
    return lo * n_obj - lo * (lo + 1) // 2 + (hi - lo - 1)


class SparseMinimaGather(object):
    """What the ranks of a row-sharded finite-difference step exchange when rank 0's SLSQP wants the per-pair separation
    minima of EVERY row (north_star: "an all-gather of inter-vehicle separation minima"): batch row b differs from row 0 only
    in the n_obj - 1 pairs of the vehicle it advances, so a rank sends, per owned row, those n_obj - 1 minima -- B x (n_obj - 1)
    doubles in all (C4: 7169 x 255 x 8 B = 14.6 MB; the dense [B][P] block would be 1.87 GB) -- and row 0's P minima travel in
    the tail of every rank's block (meaningful in rank 0's: C4 261 KB per rank).  ONE `all_gather_into_tensor` of persistent
    buffers, nothing else: at C3 a second collective and an assembly copy cost more than the step they followed.
    `dense_rows()` rebuilds any rows of the dense block on demand."""

    def __init__(self, B, n_veh, n_obj, dim, n_free_cols, group=None, world=None, rank=None):
        w, r = _world_rank(group)
        self.group = group
        self.world = w if world is None else int(world)
        self.rank = r if rank is None else int(rank)
        self.B, self.n_obj = int(B), int(n_obj)
        self.P = self.n_obj * (self.n_obj - 1) // 2
        self.blocks = partition(self.B, self.world)
        self.begin, self.count = self.blocks[self.rank]
        self.max_count = max(c for _, c in self.blocks)
        self.veh = fd_row_vehicle(torch.arange(self.B), n_veh, dim, n_free_cols)         # [B]
        self.pair_idx = pairs_of_vehicle(self.n_obj)                                      # [n_obj][n_obj - 1]
        self.block_len = self.max_count * (self.n_obj - 1) + self.P                       # doubles a rank sends
        self.bytes_per_step = 8 * self.world * self.block_len
        self._send = self._recv = None

    def _buffers(self, device, dtype=torch.float64):
        if self._send is None or self._send.device != torch.device(device) or self._send.dtype != dtype:
            self._send = torch.zeros(self.block_len, dtype=dtype, device=device)
            self._recv = torch.zeros((self.world, self.block_len), dtype=dtype, device=device)
        return self._send, self._recv

    def send_rows(self, device, dtype=torch.float64):
        """[max_count][n_obj - 1], a view of the persistent send block: a kernel may write this rank's rows straight into its
        first `count` rows (obtg_temporal_sep_fd_min_rows_dev) and pass `send_rows()[:count]` to exchange_compact."""
        return self._buffers(device, dtype)[0][:self.max_count * (self.n_obj - 1)].view(self.max_count, self.n_obj - 1)

    def send_row0(self, device, dtype=torch.float64):
        """[P], the tail of the send block: row 0's minima on the rank that owns the batch's row 0."""
        return self._buffers(device, dtype)[0][self.max_count * (self.n_obj - 1):]

    def compact(self, minima_rows):
        """minima_rows: this rank's [count][P] per-pair minima -> [count][n_obj - 1]: per row the pairs of its vehicle
        (row 0, which advances nothing, carries its first n_obj - 1 pairs: never read back)."""
        veh = self.veh[self.begin:self.begin + self.count].clamp(min=0).to(minima_rows.device)
        idx = self.pair_idx.to(minima_rows.device)[veh]                                   # [count][n_obj - 1]
        return torch.gather(minima_rows, 1, idx)

    def exchange(self, minima_rows, force=False):
        """-> (sparse[B][n_obj - 1], row0[P]) identical on every rank, from this rank's DENSE rows [count][P]."""
        row0 = minima_rows[0] if self.begin == 0 and self.count else None
        return self.exchange_compact(self.compact(minima_rows), row0, force)

    def exchange_compact(self, mine, row0, force=False):
        """The same from rows that are compact already -- what `obtg_temporal_sep_fd_min_rows_dev` writes: mine[count][n_obj - 1]
        (the owner of the batch's row 0 passes anything in ITS row 0 slot), row0[P] (rank 0; None elsewhere)."""
        rows, tail = self.send_rows(mine.device, mine.dtype), self.send_row0(mine.device, mine.dtype)
        if mine.data_ptr() != rows.data_ptr():                 # (a caller that wrote into send_rows() itself skips the copy)
            rows[:self.count] = mine
        if row0 is not None and row0.data_ptr() != tail.data_ptr():
            tail.copy_(row0)
        send, recv = self._buffers(mine.device, mine.dtype)
        live = dist.is_available() and dist.is_initialized() and (self.world > 1 or force)
        if live:
            dist.all_gather_into_tensor(recv.view(-1), send, group=self.group)
        else:
            recv[self.rank].copy_(send)
        w = self.n_obj - 1
        if self.world == 1:
            sparse = recv[0, :self.count * w].view(self.count, w)
        else:
            sparse = torch.cat([recv[r, :c * w].view(c, w) for r, (_, c) in enumerate(self.blocks)], dim=0)
        return sparse, recv[0, self.max_count * w:]

    def dense_rows(self, sparse, row0, rows):
        """Rows `rows` of the dense [B][P] minima block, rebuilt from what `exchange` returned."""
        rows = torch.as_tensor(rows, dtype=torch.int64)
        out = row0[None, :].repeat(len(rows), 1)
        for i, b in enumerate(rows.tolist()):
            v = int(self.veh[b])
            if v >= 0:
                out[i, self.pair_idx[v].to(out.device)] = sparse[b]
        return out


class PairPartitionedSweep(object):
    """All-gather of per-rank pair blocks into the full [B, n_pairs*width] result."""

    def __init__(self, n_pairs, width, group=None):
        self.group = group
        self.world, self.rank = _world_rank(group)
        self.n_pairs, self.width = n_pairs, width
        self.blocks = partition(n_pairs, self.world)
        self.max_count = max(c for _, c in self.blocks)

    @property
    def my_block(self):
        return self.blocks[self.rank]

    def run(self, evaluate, B, device, dtype=torch.float64):
        """evaluate(pair_begin, pair_count) -> tensor [B, pair_count*width] on `device`.
        Returns the assembled [B, n_pairs*width] tensor (identical on every rank)."""
        begin, count = self.my_block
        mine = evaluate(begin, count)
        if self.world == 1:
            return mine
        return all_gather_pair_blocks([(mine, self.blocks, self.width)], group=self.group)[0]


def gpu_temporal_sep_evaluator(ctx, dY, B, max_sep, min_only=False):
    """Evaluator over the HIP path for PairPartitionedSweep.run (device tensors, ctx's stream must
    be torch's current stream)."""
    L = 1 if min_only else 2 * ctx.deg + ctx.deg_elev + 1

    def evaluate(begin, count):
        out = torch.empty((B, count * L), dtype=torch.float64, device=dY.device)
        if count:
            fn = ctx.temporal_sep_min_dev if min_only else ctx.temporal_sep_dev
            fn(dY.data_ptr(), B, max_sep, out.data_ptr(), begin, count)
        return out
    return evaluate


class GpuHullPairSweep(object):
    """The gjkNew hull sweep with its pair list partitioned over the ranks ("GJK pairs
    partition the same way").  Every rank holds all control points and static objects, registers ITS
    contiguous block of the hull pair list with the context (`obtg_ctx_set_hull_pairs`), sweeps it, and
    one all-gather returns (dist[B, P_s] float64, flag[B, P_s] int32) on every rank -- gjkNew's separation
    distance (NaN where flag != 1) and flag per pair."""

    def __init__(self, ctx, pair_a, pair_b, group=None, max_iter=128, md_cap=256):
        self.ctx, self.group = ctx, group
        self.world, self.rank = _world_rank(group)
        self.n_pairs = len(pair_a)
        self.blocks = partition(self.n_pairs, self.world)
        b, c = self.blocks[self.rank]
        ctx.set_hull_pairs(pair_a[b:b + c], pair_b[b:b + c])
        self.count = c
        self.max_iter, self.md_cap = max_iter, md_cap
        self._buf = None

    def evaluate(self, dY, B):
        """-> (dist[B, count], flag[B, count]) of this rank's block (device tensors)."""
        c = self.count
        if self._buf is None or self._buf[0].shape[0] != B:
            dev = dY.device
            self._buf = (torch.empty((B, c), dtype=torch.float64, device=dev),
                         torch.empty((B, c), dtype=torch.int32, device=dev),
                         torch.empty((B, c, 3), dtype=torch.float64, device=dev),
                         torch.empty((B, c, 3), dtype=torch.float64, device=dev))
        d_dist, d_flag, d_p1, d_p2 = self._buf
        if c:
            self.ctx.gjk_swarm_dev(dY.data_ptr(), B, d_flag.data_ptr(), d_p1.data_ptr(), d_p2.data_ptr(),
                                   d_dist.data_ptr(), None, None, self.max_iter, self.md_cap)
        return d_dist, d_flag

    def parts(self, dY, B):
        d_dist, d_flag = self.evaluate(dY, B)
        return [(d_dist, self.blocks, 1), (d_flag, self.blocks, 1)]

    def run(self, dY, B):
        return all_gather_pair_blocks(self.parts(dY, B), group=self.group)


class RowShardedFdStep(object):
    """Batch sharding of ONE finite-difference step (optimization.py:83-187 under SciPy's
    approx_derivative): the B = n_x + 1 rows formed from x's control points are split into contiguous balanced blocks,
    rank r evaluates rows [begin, begin + count) -- through the structured step (one launch: the unperturbed row, the source
    of the streams, is evaluated by every rank; per row only what its vehicle touches) or through the brute-force sweep on a
    row-range view -- and keeps its rows: Jacobian row blocks, no collective on the data path.  world / rank default to the
    process group's (1 / 0 without one); pass them to lay out a step by hand."""

    def __init__(self, B, group=None, world=None, rank=None):
        w, r = _world_rank(group)
        self.world = w if world is None else int(world)
        self.rank = r if rank is None else int(rank)
        self.B = int(B)
        self.begin, self.count = shard_rows(self.B, self.world, self.rank)

    def buffers(self, ctx, n_hull_pairs, device):
        """Output tensors of this rank's rows (what `run` fills)."""
        f64, i32 = torch.float64, torch.int32
        c = self.count
        return dict(sep=torch.empty((c, ctx.len_temporal_sep), dtype=f64, device=device),
                    speed=torch.empty((c, ctx.len_speed), dtype=f64, device=device),
                    ang=torch.empty((c, ctx.len_ang_rate), dtype=f64, device=device) if ctx.dim == 2 else None,   # (optimization.py:171-187: planar only)
                    flag=torch.empty((c, n_hull_pairs), dtype=i32, device=device),
                    p1=torch.empty((c, n_hull_pairs, 3), dtype=f64, device=device),
                    p2=torch.empty((c, n_hull_pairs, 3), dtype=f64, device=device),
                    dist=torch.empty((c, n_hull_pairs), dtype=f64, device=device),
                    status=torch.empty((c, n_hull_pairs), dtype=i32, device=device))

    def run(self, ctx, dY0, n_fixed_cols, h, d_tf_rows, max_sep, speed_bound, speed_is_max, max_rate, out, structured="auto",
            max_iter=128, md_cap=256):
        """dY0: device pointer of x's control points; d_tf_rows: device pointer of THIS rank's `count` final times;
        out: `buffers(...)`.  structured: True (raises ObtgError where the shape has no structured step: 3-D rows, degree 20,
        rows beyond 158 KB), False (the brute-force sweep on obtg_fd_view_begin_rows: the same numbers), "auto" (the
        structured step where there is one, else brute force).  Sets `self.strategy` to what ran."""
        from . import _capi
        self.strategy = None
        if self.count == 0:
            return out
        if structured and out["ang"] is not None:
            try:
                ctx.constraint_sweep_fd_structured_dev(dY0, n_fixed_cols, h, d_tf_rows, self.count, max_sep, out["sep"].data_ptr(),
                                                       speed_bound, speed_is_max, max_rate, out["speed"].data_ptr(),
                                                       out["ang"].data_ptr(), out["flag"].data_ptr(), out["p1"].data_ptr(),
                                                       out["p2"].data_ptr(), out["dist"].data_ptr(), None, out["status"].data_ptr(),
                                                       max_iter, md_cap, row_begin=self.begin)
                self.strategy = "structured"
                return out
            except _capi.ObtgError as e:
                if structured != "auto" or e.code != _capi.ERR_UNSUPPORTED:
                    raise
        elif structured is True:
            raise _capi.ObtgError("the structured step covers planar rows only", _capi.ERR_UNSUPPORTED)
        ctx.fd_view_begin(dY0, n_fixed_cols, h, self.count, row_begin=self.begin)
        try:
            ctx.constraint_sweep_dev(None, d_tf_rows, self.count, max_sep, out["sep"].data_ptr(), speed_bound, speed_is_max,
                                     max_rate, out["speed"].data_ptr(), out["ang"].data_ptr() if out["ang"] is not None else None,
                                     out["flag"].data_ptr(), out["p1"].data_ptr(), out["p2"].data_ptr(), out["dist"].data_ptr(),
                                     None, out["status"].data_ptr(), max_iter, md_cap)
        finally:
            ctx.fd_view_end()
        self.strategy = "brute force"
        return out
