"""Multi-GPU layer: one process per GPU, samples sharded by contiguous blocks, and ONE
collective -- an all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests) that reassembles
the stabilised sequence.  The reference has no distributed code at all (single tf.Session,
main:489); samples are independent at inference (BatchNorm uses moving statistics), so there is
no other exchange step on this path.

Training is the one place with a real exchange: data-parallel replicas average their gradients --
`GradBucket` keeps every parameter gradient as a view into ONE flat buffer so that the step needs a single all-reduce
(155 MB for the 38.7 M parameters; on xGMI's point-to-point links one large collective beats many small ones)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition of range(n_items): the first n_items % world ranks get one extra."""
    if world < 1 or not (0 <= rank < world) or n_items < 0:
        raise ValueError("bad shard arguments")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def gather_sequence(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather the per-rank shards (leading dim = this rank's shard_range size) back into the
    full [n_total, ...] sequence on every rank, in the original order."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: shard has {local.shape[0]} items, expected {sizes[rank]}")
    mx = max(sizes) if sizes else 0
    if mx == 0:
        return local.new_empty((0,) + tuple(local.shape[1:]))
    padded = local
    if local.shape[0] < mx:
        pad = local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))
        padded = torch.cat([local, pad])
    out = local.new_empty((world, mx) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out.view((world * mx,) + tuple(local.shape[1:])), padded.contiguous(), group=group)
    return torch.cat([out[r, :sizes[r]] for r in range(world)])


class SequenceGatherer:
    """Overlapped reassembly of a block-partitioned sequence (BASELINE configs[3]: a clip sharded by `shard_range`): while a rank
    works through its shard in micro-batches, every finished micro-batch is all-gathered straight into its final position of the
    full [n_total, ...] sequence, asynchronously, beside the next micro-batch's kernels -- instead of one collective of the whole
    shard after the last kernel (0.78 GB per rank for cfg3's uint8 frames: ~36 ms on one xGMI ring link, exposed).  Every rank must
    call `submit` with the same chunk lengths in the same order; shards may differ in length by one (block partition), so the common
    part (min shard length) goes through `submit` and each rank's last item, if it has one more, through `finish`."""

    def __init__(self, n_total: int, item_shape, dtype, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.sizes = shard_sizes(n_total, self.world)
        self.offsets = [shard_range(n_total, r, self.world)[0] for r in range(self.world)]
        self.common = min(self.sizes) if self.sizes else 0
        self.full = torch.empty((n_total,) + tuple(item_shape), dtype=dtype, device=device)
        self.pending = []

    def submit(self, chunk: torch.Tensor, local_start: int):
        """chunk = items [local_start, local_start + len(chunk)) of this rank's shard, inside the common part."""
        n = chunk.shape[0]
        if local_start < 0 or local_start + n > self.common:
            raise ValueError("submit() covers the common part of the shards only; the ragged remainder goes through finish()")
        chunk = chunk.contiguous()
        outs = [self.full[o + local_start:o + local_start + n] for o in self.offsets]
        self.pending.append((dist.all_gather(outs, chunk, group=self.group, async_op=True), chunk))

    def finish(self, tail: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tail = this rank's items beyond the common part (0 or 1 of them; None / empty if it has none).  Returns the full sequence."""
        extra = [s - self.common for s in self.sizes]
        if max(extra, default=0) > 0:
            mx = max(extra)
            mine = extra[self.rank]
            pad = self.full.new_zeros((mx,) + tuple(self.full.shape[1:]))
            if mine:
                if tail is None or tail.shape[0] != mine:
                    raise ValueError(f"rank {self.rank}: finish() needs its {mine} item(s) beyond the common part")
                pad[:mine].copy_(tail)
            got = self.full.new_empty((self.world, mx) + tuple(self.full.shape[1:]))
            dist.all_gather_into_tensor(got.view((self.world * mx,) + tuple(self.full.shape[1:])), pad, group=self.group)
            for r in range(self.world):
                if extra[r]:
                    self.full[self.offsets[r] + self.common:self.offsets[r] + self.sizes[r]].copy_(got[r, :extra[r]])
        for w, _keep in self.pending:
            w.wait()
        self.pending.clear()
        return self.full


class _Works:
    """A list of point-to-point requests waited for as one."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()


class FrameGatherer:
    """Overlapped reassembly for a stream of equally sized per-rank batches: submit() starts an
    asynchronous all-gather of this step's frames into one of `depth` rotating buffers and
    returns at once; the collective runs beside the next step's kernels.  result(k) / drain()
    wait.  Buffers: [world, *shape]; rank r's frames land at index r (block partition)."""

    def __init__(self, shape, world: int, device, dtype=torch.float32, depth: int = 2, group=None, schedule: str = "allgather"):
        """schedule = "allgather": one `all_gather_into_tensor` per step (RCCL picks ring / direct itself).
        schedule = "direct": every rank pushes its frames straight to each of its world-1 peers with point-to-point sends and
        posts the matching receives into its own gather buffer (`batch_isend_irecv`) -- on a fully connected xGMI hive the
        world-1 transfers of a rank use world-1 different links at once, where a ring moves (world-1)/world of the data over ONE
        link per rank (~140 ms vs ~20 ms for cfg3's 3.1 GB per rank).  Same result, bit for bit."""
        if schedule not in ("allgather", "direct"):
            raise ValueError("schedule must be 'allgather' or 'direct'")
        self.schedule = schedule
        self.world, self.depth, self.group = world, depth, group
        self.bufs = [torch.empty((world,) + tuple(shape), dtype=dtype, device=device) for _ in range(depth)]
        self.pending: List[Optional[tuple]] = [None] * depth
        self.count = 0

    def reserve(self) -> int:
        """Slot the next submit() will use, after waiting for the collective that last used it: a caller that
        stages its frames in a per-slot buffer of its own may overwrite that buffer once this returns."""
        slot = self.count % self.depth
        self._wait(slot)
        return slot

    def submit(self, frames: torch.Tensor) -> int:
        slot = self.count % self.depth
        self._wait(slot)
        buf = self.bufs[slot]
        frames = frames.contiguous()
        if self.schedule == "direct" and self.world > 1:
            rank = dist.get_rank(self.group)
            buf[rank].copy_(frames)                  # own shard: a local copy on the current stream
            ops = []
            for d in range(1, self.world):           # peer order staggered by rank so that no two ranks start on the same target
                dst, src = (rank + d) % self.world, (rank - d) % self.world
                ops.append(dist.P2POp(dist.isend, frames, dst, group=self.group))
                ops.append(dist.P2POp(dist.irecv, buf[src], src, group=self.group))
            work = _Works(dist.batch_isend_irecv(ops))
        else:
            work = dist.all_gather_into_tensor(buf.view((-1,) + tuple(buf.shape[2:])), frames, group=self.group, async_op=True)
        self.pending[slot] = (work, frames)          # keep `frames` alive until the collective is done
        self.count += 1
        return slot

    def _wait(self, slot: int):
        if self.pending[slot] is not None:
            self.pending[slot][0].wait()
            self.pending[slot] = None

    def result(self, slot: int) -> torch.Tensor:
        self._wait(slot)
        b = self.bufs[slot]
        return b.view((-1,) + tuple(b.shape[2:]))

    def drain(self):
        for s in range(self.depth):
            self._wait(s)


class StepGroupGatherer:
    """bench.py's reassembly of a STREAM of steps (weak scaling: every rank produces `B` uint8 frames per step): `every` steps' frames
    are staged in one of two rotating buffers and reassembled with ONE collective (fewer, larger collectives: xGMI rings are
    per-link bound), overlapped with the following steps.  A run whose step count is not a multiple of `every` ends in a shorter
    group: its gatherer is allocated up front (`tail_steps`: the step counts the caller will flush at -- warm-up and timed region),
    so nothing allocates between the benchmark's opening barrier and its closing synchronise.

    Per step: `dst = stage()` -> write the step's frames into `dst` ([B, ...] view of the staging buffer; on the stream the
    collective will be ordered after) -> `commit()`.  `flush()` gathers a group the step count left unfinished and waits for
    everything in flight.  `to_comm(t)` maps a staging tensor to what the process group can move (gloo: a host copy)."""

    def __init__(self, every: int, item_shape, world: int, stage_device, comm_device, dtype=torch.uint8, schedule: str = "allgather",
                 tail_steps=(), group=None, to_comm=None):
        if every < 1:
            raise ValueError("every must be >= 1")
        self.every, self.item_shape, self.world = every, tuple(item_shape), world
        self.to_comm = to_comm or (lambda t: t)
        self.main = FrameGatherer((every * self.item_shape[0],) + self.item_shape[1:], world, comm_device, dtype=dtype, schedule=schedule, group=group)
        self.stagebufs = [torch.empty((every,) + self.item_shape, dtype=dtype, device=stage_device) for _ in range(2)]
        self.tails = {}
        for n in tail_steps:
            r = n % every
            if r and r not in self.tails:
                self.tails[r] = FrameGatherer((r * self.item_shape[0],) + self.item_shape[1:], world, comm_device, dtype=dtype, schedule=schedule,
                                              group=group)
        self.g, self.slot = 0, 0
        self.groups_submitted, self.tail_groups_submitted = 0, 0
        self.last = None                    # (gatherer, slot) of the newest collective: result() reads it

    def stage(self) -> torch.Tensor:
        if self.g == 0:
            self.slot = self.main.reserve()      # the collective that last read this staging buffer has completed
        return self.stagebufs[self.slot][self.g]

    def commit(self):
        self.g += 1
        if self.g == self.every:
            buf = self.stagebufs[self.slot].view((self.every * self.item_shape[0],) + self.item_shape[1:])
            self.last = (self.main, self.main.submit(self.to_comm(buf)))
            self.groups_submitted += 1
            self.g = 0

    def flush(self):
        """Gather a group the step count left unfinished (one smaller collective), then wait for everything in flight."""
        if self.g > 0:
            if self.g not in self.tails:
                raise ValueError(f"a group of {self.g} steps was not announced in tail_steps")
            tail = self.tails[self.g]
            buf = self.stagebufs[self.slot][:self.g].reshape((self.g * self.item_shape[0],) + self.item_shape[1:])
            self.last = (tail, tail.submit(self.to_comm(buf)))
            tail.drain()
            self.tail_groups_submitted += 1
            self.g = 0
        self.main.drain()

    def result(self) -> torch.Tensor:
        """Frames of the newest reassembled group: [world * steps_in_group * B, ...], rank-major."""
        g, slot = self.last
        return g.result(slot)


class GradBucket:
    """One flat fp32 buffer; `views[name]` are tensors of the given shapes that alias consecutive pieces of it (each piece
    16-byte aligned).  `allreduce_mean()` averages the whole buffer over the process group in a single collective."""

    def __init__(self, shapes, device):
        self.offsets, n = {}, 0
        for name, shape in shapes.items():
            self.offsets[name] = n
            numel = 1
            for s in shape:
                numel *= int(s)
            n += (numel + 3) // 4 * 4
        self.flat = torch.zeros(max(n, 4), dtype=torch.float32, device=device)
        self.views = {}
        for name, shape in shapes.items():
            numel = 1
            for s in shape:
                numel *= int(s)
            self.views[name] = self.flat[self.offsets[name]:self.offsets[name] + numel].view(tuple(shape))

    def span(self, first: str, last: str):
        """[lo, hi) element range of the flat buffer that holds tensors `first` .. `last` (in the bucket's own order)."""
        names = list(self.offsets)
        i, j = names.index(first), names.index(last)
        if j < i:
            raise ValueError("span: `last` comes before `first`")
        hi = self.offsets[names[j + 1]] if j + 1 < len(names) else self.flat.numel()
        return self.offsets[first], hi

    def allreduce_range_start(self, lo: int, hi: int, group=None, single_rank_ok: bool = False):
        """Start averaging flat[lo:hi] over the process group WITHOUT waiting for it: the collective is ordered after the work
        already queued on the current stream (the kernels that produced those gradients) and runs beside whatever is queued
        next -- the way a backward pass overlaps the exchange of finished gradient buckets with the layers still to come.
        Returns a handle for `allreduce_finish` (None without a process group or with a single rank -- unless
        `single_rank_ok`, which issues the collective anyway so that the path can be exercised on one GPU)."""
        if not dist.is_available() or not dist.is_initialized():
            return None
        world = dist.get_world_size(group)
        if (world == 1 and not single_rank_ok) or hi <= lo:
            return None
        seg = self.flat[lo:hi]
        if hasattr(dist.ReduceOp, "AVG") and seg.is_cuda:
            return (dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=group, async_op=True), None, 1.0)
        return (dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=group, async_op=True), seg, 1.0 / world)      # gloo: no AVG

    @staticmethod
    def allreduce_finish(handles):
        """Wait for the collectives started by `allreduce_range_start` (on CUDA: makes the current stream wait for them)."""
        for h in handles:
            if h is None:
                continue
            work, seg, scale = h
            work.wait()
            if seg is not None:
                seg.mul_(scale)

    def allreduce_mean(self, group=None, async_op: bool = False):
        """flat <- mean over ranks.  No-op without an initialised process group or with a single rank."""
        if not dist.is_available() or not dist.is_initialized():
            return None
        world = dist.get_world_size(group)
        if world == 1:
            return None
        if hasattr(dist.ReduceOp, "AVG") and self.flat.is_cuda:
            return dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=False)
        self.flat.mul_(1.0 / world)           # gloo has no AVG: a scale on the flat buffer (CPU tests only)
        return work
