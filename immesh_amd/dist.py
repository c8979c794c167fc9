"""Multi-GPU plumbing of bench.py (one process per GPU under torch.distributed.run; backend "nccl" = RCCL on ROCm, "gloo"
in the CPU tests).  Two multi-GPU modes: replicas (every rank runs its own scan stream against its own
map: no data-path collective, only the timing barrier and the max-over-ranks elapsed time) and the sharded map / mesher
(one stream; the collectives live in the C++ layer -- RCCL -- or behind the host callbacks, see immesh_c_api.h)."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return dist


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """max of a python float over all ranks (identity when not initialised)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def stream_of_rank(rank, n_scans):
    """Scan indices of rank's stream: every rank replays the same synthetic trajectory, phase-shifted by `rank` scans so that the
    replicas do not process byte-identical inputs.  Returns n_scans + 1 consecutive indices (the first seeds the stream)."""
    return list(range(rank, rank + n_scans + 1))


def aggregate_throughput(steps_per_rank, world, elapsed_max_s):
    """whole-job scans/s: all ranks' scans over the slowest rank's time"""
    return steps_per_rank * world / elapsed_max_s
