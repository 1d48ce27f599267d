"""Multi-GPU execution of the constraint sweeps: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU rehearsal in tests).

Two independent ways to shard:

  * batch sharding (default, NO collective): the B = n_x + 1 evaluation rows of one SLSQP
    Jacobian are independent; rank r takes a contiguous block of rows.  Results go back to
    the host as Jacobian column blocks.  `shard_rows`.
  * pair partitioning (one evaluation across GPUs, the 256-vehicle case): every rank holds
    all control points (65 KB), owns a contiguous balanced block of the lexicographic pair
    list, and ONE all-gather assembles either the full constraint vector or only the
    per-pair minima.  On a fully connected 8-GPU xGMI node each rank pushes its shard to 7
    peers concurrently, so the collective is latency-bound at these sizes (1 MB shards).
    `PairPartitionedSweep`.

The compute itself is injected (`evaluate(pair_begin, pair_count) -> tensor[B, count*width]`)
so that the orchestration can be rehearsed on CPU with gloo; on GPUs the evaluator is a
closure over `Context.temporal_sep_dev`.
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


class PairPartitionedSweep(object):
    """All-gather of per-rank pair blocks into the full [B, n_pairs*width] result."""

    def __init__(self, n_pairs, width, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
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
        w = self.width
        # equal-sized shards for all_gather_into_tensor: pad the short blocks by one pair
        send = torch.zeros((self.max_count * w, B), dtype=dtype, device=device)
        send[:count * w] = mine.t()
        recv = torch.empty((self.world * self.max_count * w, B), dtype=dtype, device=device)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        recv = recv.view(self.world, self.max_count * w, B)
        parts = [recv[r, :c * w] for r, (_, c) in enumerate(self.blocks)]
        return torch.cat(parts, dim=0).t().contiguous()


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
