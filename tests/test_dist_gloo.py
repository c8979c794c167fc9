"""The N>1 path of bench.py on CPU: world_size-2 gloo process group, barrier, max-over-ranks timing, per-rank stream assignment,
whole-job aggregation.  (The data-path collectives of the sharded mode are exercised by tests/test_gpu_sharded.py and tests/test_bench_plumbing.py.)"""
import os
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from immesh_amd import dist as D
    assert D.env_rank() == (rank, world, rank)
    dist = D.init("gloo")
    D.barrier()
    elapsed = 0.5 + 0.25 * rank            # rank 1 is the slow one
    mx = D.max_over_ranks(elapsed)
    stream = D.stream_of_rank(rank, 4)
    out[rank] = (mx, stream, D.aggregate_throughput(20, world, mx))
    D.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_streams():
    world = 2
    port = 29500 + (os.getpid() % 400)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert set(out.keys()) == {0, 1}
    assert out[0][0] == out[1][0] == 0.75                       # both ranks agree on the max
    assert out[0][1] == [0, 1, 2, 3, 4] and out[1][1] == [1, 2, 3, 4, 5]
    assert abs(out[0][2] - 40 / 0.75) < 1e-9                    # whole-job scans/s = all ranks' scans / slowest rank


def test_single_process_is_identity():
    sys.path.insert(0, ROOT)
    from immesh_amd import dist as D
    assert D.max_over_ranks(1.25) == 1.25
    assert D.aggregate_throughput(10, 1, 2.0) == 5.0


def test_shard_owner_partitions_bricks():
    """Host mirror of the kernels' ownership function (no GPU needed): every key has exactly one owner, all voxels of a brick share
    it (negative keys included), and the bricks spread evenly over the ranks."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from immesh_amd import capi
    lib = capi.load_hip_library()
    rng = np.random.default_rng(1)
    for world in (2, 4, 8):
        cfg = capi.avia_config(shard_rank=0, shard_world=world, shard_brick_log2=5)
        keys = rng.integers(-4000, 4000, size=(4000, 3))
        owners = np.array([capi.shard_owner(lib, cfg, k) for k in keys])
        assert owners.min() >= 0 and owners.max() < world
        counts = np.bincount(owners, minlength=world)
        assert counts.min() > 0.7 * len(keys) / world             # balanced
        for k, o in zip(keys[:200], owners[:200]):                 # constant inside a 32^3 brick, also across the origin
            base = (k >> 5) << 5
            assert capi.shard_owner(lib, cfg, base) == o and capi.shard_owner(lib, cfg, base + 31) == o
    assert capi.shard_owner(lib, capi.avia_config(), np.array([5, -7, 9])) == 0     # sharding off
    # both ownership schemes (immesh_config::shard_scheme): the numpy mirror bench.py's load-balance report uses (immesh_amd/dist.py) is the library's function;
    # lattice colouring (scheme 0, the default): along every axis consecutive bricks cycle through all ranks, neighbours never share an owner
    from immesh_amd import dist as D
    for scheme in (0, 1):
        for world in (2, 4, 8):
            cfg = capi.avia_config(shard_rank=0, shard_world=world, shard_brick_log2=3, shard_scheme=scheme)
            keys = rng.integers(-3000, 3000, size=(1500, 3))
            lib_owner = np.array([capi.shard_owner(lib, cfg, k) for k in keys])
            np.testing.assert_array_equal(D.owners_of_root_voxels(keys, 3, world, scheme), lib_owner)
            if scheme == 0:
                for axis in range(3):
                    k = np.zeros((world, 3), np.int64); k[:, axis] = np.arange(world) * 8 - 24
                    assert sorted(D.owners_of_root_voxels(k, 3, world, 0).tolist()) == list(range(world))
                    step = np.zeros(3, np.int64); step[axis] = 8
                    assert (D.owners_of_root_voxels(keys, 3, world, 0) != D.owners_of_root_voxels(keys + step, 3, world, 0)).all()
