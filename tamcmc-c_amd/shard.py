"""Multi-GPU sharding of the path: independent stars, one star (all its tempered chains) per GPU / rank,
no data-path collective -- the reference's own outer loop over stars (main.cpp:176; scripts/slurm/job_array.sh:8,28).
torch.distributed is used for the launch barrier and the max-over-ranks of the elapsed time only."""
import os
import time


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def stars_of_rank(n_stars, rank, world):
    """Star k -> rank k mod world (round-robin, like one SLURM array task per star)."""
    return [k for k in range(n_stars) if k % world == rank]


def timed_region(fn, dist=None, sync=None):
    """barrier + sync, run fn(), sync + barrier; returns the MAX elapsed time over ranks."""
    def fence():
        if sync is not None:
            sync()
        if dist is not None:
            dist.barrier()
            if sync is not None:
                sync()
    fence()
    t0 = time.perf_counter()
    out = fn()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        dev = "cuda" if (sync is not None and torch.cuda.is_available()) else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def aggregate_rate(units_per_rank, world, elapsed):
    """Whole-job throughput: every rank processed `units_per_rank` units in the (max) elapsed time."""
    return world * units_per_rank / elapsed
