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


# ---- host placement of a rank's worker processes (bench.py --config 4 --gpus 8: 8 ranks x `--workers` processes on one node) ----------

def cap_host_threads(n: int = 1):
    """One BLAS / OpenMP thread per worker process: with 6 workers per GPU and 8 GPUs a node runs 48 problem-solving processes next to
    8 rank processes; thread teams per process (OpenBLAS spins up one per 20k-element dot) would oversubscribe every core.  Must
    run before the worker processes are spawned (they inherit the environment)."""
    import os

    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[k] = str(n)


def gpu_numa_cpus(device: int, local_world: int = 1, pci_bus_id: str | None = None):
    """CPUs a rank's processes should run on: the cores of the NUMA node its GPU hangs off (sysfs), else an even slice of the
    cores this process may use (one slice per local rank).  Returns a sorted list of CPU ids (never empty)."""
    import os

    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    node = None
    if pci_bus_id:
        try:
            with open(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/numa_node") as f:
                node = int(f.read().strip())
        except (OSError, ValueError):
            node = None
    if node is not None and node >= 0:
        try:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                cpus = []
                for part in f.read().strip().split(","):
                    lo, _, hi = part.partition("-")
                    cpus += list(range(int(lo), int(hi or lo) + 1))
            cpus = [c for c in cpus if c in set(allowed)]
            if cpus:
                # (GPUs that share a NUMA node share its cores: the scheduler balances inside the node)
                return cpus
        except (OSError, ValueError):
            pass
    w = max(1, local_world)
    per = max(1, len(allowed) // w)
    lo = (device % w) * per
    return allowed[lo:lo + per] or allowed


def pin_process(cpus):
    """restrict the calling process (and the threads it starts) to `cpus`; a no-op where the platform cannot"""
    import os

    if cpus and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, set(cpus))
            return True
        except OSError:
            return False
    return False


def host_core_budget(world: int, workers: int, in_flight: int = 1):
    """host cores the batch driver keeps busy: per rank one driver process + `workers` solver processes (`in_flight` threads each hold
    the GIL in turn, so a process is ~1 core); the analysis thread of a handle is short-lived and not counted"""
    return world * (1 + max(1, workers))
