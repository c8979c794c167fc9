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


def attach_collectives(h, hip, dist, backend, device, world, mesh):
    """Data-path collectives of a sharded context.  backend "nccl": the library calls RCCL itself on its own streams
    (immesh_rccl_init: ncclAllReduce of the 46 sums between a residual pass and the 18-state update, ncclAllGather of the mesher's exchange
    records); the unique id travels through torch.distributed.  Any other backend (gloo, functional tests): host callbacks.
    Returns a short description for the bench line."""
    import numpy as np
    import torch
    if backend == "nccl" and hasattr(h, "rccl_init"):
        uid = torch.zeros(128, dtype=torch.uint8, device=device)
        if dist.get_rank() == 0:
            uid.copy_(torch.from_numpy(h.rccl_unique_id()))
        dist.broadcast(uid, src=0)
        h.rccl_init(uid.cpu().numpy())
        return "RCCL inside the library (ncclAllReduce / ncclAllGather on its own streams)"

    def _allreduce(buf):                      # 46 doubles: H^T R^-1 H, H^T R^-1 z, counters
        if backend == "nccl":
            t = torch.from_numpy(buf).to(device); dist.all_reduce(t); buf[:] = t.cpu().numpy()
        else:
            dist.all_reduce(torch.from_numpy(buf))
    h.set_allreduce(_allreduce)
    if mesh:   # sharded mesher: all-gather of this scan's smoothed vertices and triangle marks (two exchanges per scan)
        def _allgather(send, recv):
            if backend == "nccl":
                src = torch.from_numpy(send).to(device)
                dst = torch.empty(len(send) * world, dtype=torch.uint8, device=device)
                dist.all_gather_into_tensor(dst, src)
                recv[:] = dst.cpu().numpy()
            else:
                parts = [torch.empty(len(send), dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, torch.from_numpy(send))
                for r_ in range(world):
                    recv[r_ * len(send):(r_ + 1) * len(send)] = parts[r_].numpy()
        h.set_allgather(_allgather)
    return f"host callbacks through torch.distributed ({backend})"


# ---- host mirror of the kernels' brick ownership (regmap.hpp shard_owner / c_api.cpp immesh_shard_owner), vectorised ----------------------------------
def _hash64(k):
    import numpy as np
    k = k.astype(np.uint64)
    with np.errstate(over="ignore"):
        k ^= k >> np.uint64(30); k *= np.uint64(0xbf58476d1ce4e5b9)
        k ^= k >> np.uint64(27); k *= np.uint64(0x94d049bb133111eb)
        k ^= k >> np.uint64(31)
    return k


def owners_of_root_voxels(keys, brick_log2, world, scheme=0):
    """keys: n x 3 int64 root-voxel keys -> owning rank per key (same function as the device / immesh_shard_owner).  scheme 0 = lattice colouring
    (bx + 3 by + 5 bz) mod world, 1 = hash(brick) mod world (immesh_config::shard_scheme)"""
    import numpy as np
    b = np.int64(brick_log2)
    k = np.asarray(keys, np.int64) >> b                         # arithmetic shift: bricks tile negative keys too
    if scheme != 1 and world % 3 != 0 and world % 5 != 0:   # (a world divisible by 3 or 5 uses the hash whatever the scheme: 3 / 5 would not be units mod it)
        return np.mod(k[:, 0] + 3 * k[:, 1] + 5 * k[:, 2], np.int64(world)).astype(np.int64)   # (numpy's mod is non-negative for a positive modulus)
    B, M = np.uint64(1 << 20), np.uint64((1 << 21) - 1)
    ku = (k.astype(np.uint64) + B) & M
    packed = ku[:, 0] | (ku[:, 1] << np.uint64(21)) | (ku[:, 2] << np.uint64(42))
    return (_hash64(packed) % np.uint64(world)).astype(np.int64)


def root_voxel_keys(world_xyz, voxel_size):
    """VOXEL_LOC quantisation of world-frame points (voxel_mapping.cpp:122-128: p / voxel_size, minus one below zero, truncated)"""
    import numpy as np
    q = np.asarray(world_xyz, np.float64) / float(voxel_size)   # (the device divides the float coordinate in double as well; a point exactly on a face is measure zero here)
    q = np.where(q < 0, q - 1.0, q)
    return np.trunc(q).astype(np.int64)


def load_balance(world_clouds, voxel_size, world, brick_log2s=(3, 4, 5), scheme=0):
    """Share of the down-sampled scan points (= the matcher's and the map update's work) each rank owns, per scan, for every brick size: the slowest
    rank of a sharded job is the one with the largest share.  Returns {brick_voxels: {"max_share_mean": ..., "max_share_worst_scan": ..., "min_share_mean": ...,
    "busiest_rank": r}}; the fair share is 1 / world."""
    import numpy as np
    out = {}
    for b in brick_log2s:
        mx, mn, tot = [], [], np.zeros(world)
        for pts in world_clouds:
            own = owners_of_root_voxels(root_voxel_keys(pts, voxel_size), b, world, scheme)
            sh = np.bincount(own, minlength=world) / max(1, len(own))
            mx.append(sh.max()); mn.append(sh.min()); tot += sh
        out[int(1 << b)] = {"max_share_mean": round(float(np.mean(mx)), 4), "max_share_worst_scan": round(float(np.max(mx)), 4), "min_share_mean": round(float(np.mean(mn)), 4),
                            "busiest_rank": int(np.argmax(tot)), "fair_share": round(1.0 / world, 4)}
    return out
