"""The timed region of the benches, shared by bench.py and testable without a GPU: W untimed warm-up steps, then EXACTLY K steps
bracketed by a barrier + device synchronisation on both sides, elapsed time = the MAX over ranks (the driver's contract).
No torch.cuda calls here: the caller passes `sync` (torch.cuda.synchronize on a GPU, a no-op in the gloo CPU tests)."""
from __future__ import annotations

import time
from typing import Callable, Optional


def timed_region(step: Callable[[int], object], steps: int, warmup: int, sync: Callable[[], None], dist=None,
                 drain: Optional[Callable[[], None]] = None, before_timed: Optional[Callable[[], None]] = None,
                 device: str = "cuda", prime: Optional[Callable[[], None]] = None):
    """Returns (elapsed seconds, max over ranks; the last step's return value).  `step(k)` gets the index of the timed step
    (-1 during warm-up); `drain` waits for asynchronous work the steps started (the overlapped all-gather); `before_timed`
    runs after the warm-up has drained and before the opening barrier (profilers are switched on there); `prime` runs in front of the
    LAST warm-up step (bench.py switches its per-launch event recording on there, so that the first use of that launch path -- a
    one-off cost of about a millisecond -- falls into the warm-up and not into the first timed step: a 20-step run read 1 % slow)."""
    import torch
    out = None
    for i in range(warmup):
        if prime is not None and i == warmup - 1:
            prime()
        out = step(-1)
    if drain is not None:
        drain()
    sync()
    if before_timed is not None:
        before_timed()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        out = step(k)
    t1 = time.perf_counter()
    if drain is not None:
        drain()
    t2 = time.perf_counter()
    sync()
    t3 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    import os, sys
    if os.environ.get("VSTAB_BENCH_DEBUG"):
        print(f"timed region: issue {1e3 * (t1 - t0):.2f} ms, drain {1e3 * (t2 - t1):.2f}, sync {1e3 * (t3 - t2):.2f}, barrier {1e3 * (elapsed - (t3 - t0)):.2f}", file=sys.stderr)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def aggregate_value(units_per_rank_step: int, world: int, steps: int, elapsed: float) -> float:
    """Whole-job throughput: the units ALL ranks processed / the slowest rank's time."""
    return world * units_per_rank_step * steps / elapsed
