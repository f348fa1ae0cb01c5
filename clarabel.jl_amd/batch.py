"""Multi-GPU driver pieces: independent problems shard one-per-rank, there is no
data-path collective; the process group is used only for the start/stop barrier and the
max-over-ranks wall time.  Backend "nccl" (= RCCL over xGMI) on the GPU box, "gloo" in CPU tests."""
from __future__ import annotations

import time


def shard(n_items: int, rank: int, world: int):
    """round-robin: problem k -> rank k mod world (every item exactly once, sizes differ by <= 1)"""
    return list(range(rank, n_items, world))


def timed_steps(step, steps: int, warmup: int, dist=None, device_sync=None, reduce_device=None):
    """W untimed warm-up steps, then EXACTLY `steps` timed steps bracketed by barrier + device sync
    on both sides; returns the MAX elapsed seconds over ranks (bench.py contract)."""

    def barrier():
        if dist is not None:
            dist.barrier()
        if device_sync is not None:
            device_sync()

    for i in range(warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device or "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def gather_counts(local_count: int, dist=None, reduce_device=None) -> int:
    """sum over ranks of an integer (units processed) -- for whole-job throughput"""
    if dist is None:
        return int(local_count)
    import torch

    tt = torch.tensor([local_count], dtype=torch.int64, device=reduce_device or "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.SUM)
    return int(tt.item())


def run_concurrent(solve_one, items, in_flight: int):
    """several independent problems in flight on ONE GPU -- one host thread and one handle each (the
    C ABI is per-handle thread-safe and releases the GIL while a call runs), so that one problem's host-side work (symbolic
    analysis, cone algebra, PCIe) overlaps the others' device work.  Returns the results in item order."""
    if in_flight <= 1:
        return [solve_one(it) for it in items]
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=in_flight) as ex:
        return list(ex.map(solve_one, items))
