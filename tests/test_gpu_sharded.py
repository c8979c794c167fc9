"""Voxel-hash sharded registration map, GPU tests.
1. one process, two sharded contexts + one unsharded: the ranks' partial normal equations add up to the unsharded ones, their
   match sets partition the unsharded match set, and the owned parts of their plane tables tile the unsharded table;
2. two processes (gloo, both on cuda:0 -- the pool has one GPU per box; production uses backend "nccl" = RCCL over xGMI with one
   GPU per rank): the iterated EKF with the all-reduce callback reproduces the unsharded poses."""
import os
import sys

import numpy as np
import pytest

from immesh_amd import capi, synth
from parity_utils import clouds_within_rounding
from conftest import make_hip
from parity_utils import compare_plane_tables

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(rank=0, world=0, scheme=0):
    return capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 16, cap_triangles=1 << 18,
                            shard_rank=rank, shard_world=world, shard_brick_log2=3, shard_scheme=scheme)   # 8-voxel (4 m) bricks: many bricks in a 100 m scene


def _scans(n, npts=30000):
    cfg = _cfg()
    extT = np.array(list(cfg.extT))
    out = []
    for k in range(n):
        R, t = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, R, t, n_pts=npts, extT=extT)
        out.append((R, t, raw, synth.voxel_grid_downsample(raw, 0.4)))
    return out


@pytest.mark.parametrize("scheme", [0, 1])   # brick ownership: lattice colouring (the default) / hash of the brick (rounds 1-4), immesh_config::shard_scheme
def test_partial_sums_and_plane_tables_tile(hip_lib, scheme):
    P = 2
    scans = _scans(3)
    ref = make_hip(hip_lib, _cfg())
    shards = [make_hip(hip_lib, _cfg(r, P, scheme)) for r in range(P)]
    R0, t0, raw0, _ = scans[0]
    st0 = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    for h in [ref] + shards:
        h.map_build(p0, st0)
    R1, t1, _, down1 = scans[1]
    st = capi.make_state(R=R1 @ synth.so3_exp(np.array([1e-3, -2e-3, 1.5e-3])), t=t1 + np.array([0.02, -0.01, 0.01]), cov_diag=1e-4)
    rr = ref.residuals(down1, st)
    rs = [h.residuals(down1, st) for h in shards]
    assert all(r["n_match"] > 100 for r in rs)                       # both ranks really own part of the scene
    np.testing.assert_array_equal(np.sort(np.concatenate([r["match_idx"] for r in rs])), rr["match_idx"])
    np.testing.assert_allclose(sum(r["HTH"] for r in rs), rr["HTH"], rtol=1e-9, atol=1e-9 * np.abs(rr["HTH"]).max())
    np.testing.assert_allclose(sum(r["HTz"] for r in rs), rr["HTz"], rtol=1e-9, atol=1e-9 * np.abs(rr["HTz"]).max())
    # map growth: every rank replays only its voxels (+ halo); owned parts tile the unsharded table exactly
    for h in [ref] + shards:
        h.map_update(down1, st)
        h.map_update(scans[2][3], capi.make_state(R=scans[2][0], t=scans[2][1]))
    full = ref.dump_planes()
    owned = []
    for r, h in enumerate(shards):
        d = h.dump_planes()
        mine = np.array([capi.shard_owner(hip_lib, _cfg(r, P, scheme), k) == r for k in d["key"]])
        assert mine.sum() > 100 and (~mine).sum() > 0                # has a halo
        owned.append(d[mine])
    tiled = np.concatenate(owned)
    assert len(tiled) == len(full)
    compare_plane_tables(full, tiled, 1e-12)
    assert sum(h.counters()["n_root_voxels"] for h in shards) > ref.counters()["n_root_voxels"]   # halos are replicated


def _rank_main(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    lib = capi.load_hip_library()
    h = capi.HotPath(lib, _cfg(rank, world), "immesh_")
    h.set_allreduce(lambda buf: dist.all_reduce(torch.from_numpy(buf)))          # in place on the library's buffer

    def allgather(send, recv):
        parts = [torch.empty(len(send), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(send))
        for r in range(world):
            recv[r * len(send):(r + 1) * len(send)] = parts[r].numpy()
    h.set_allgather(allgather)                                                    # (carries immesh_broadcast_scan here; RCCL: ncclBroadcast)
    ref = capi.HotPath(lib, _cfg(), "immesh_") if rank == 0 else None
    scans = _scans(5)
    R0, t0, raw0, _ = scans[0]
    st = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    h.map_build(p0, st)
    if ref: ref.map_build(p0, st)
    st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
    sr = st.copy()
    err = 0.0
    from conftest import fetch_device
    for k in range(1, 5):
        down = scans[k][3]
        # the scan is DISTRIBUTED BY THE LIBRARY -- only the root holds it (every other rank passes nothing), every rank registers
        # and grows its shard from the copy the broadcast left in its own device memory
        d_down, n_down = h.broadcast_scan(down if rank == 0 else None, root=0)
        assert n_down == len(down)
        np.testing.assert_array_equal(fetch_device(d_down, (n_down, 3)), down)      # (the test generates the stream on every rank: it can look)
        prior = synth.forward_without_imu(st)
        st, info = h.register(d_down, prior, prior, n=n_down)
        h.map_update(d_down, st, n=n_down)
        if ref:
            pr = synth.forward_without_imu(sr)
            sr, ir = ref.register(down, pr, pr)
            ref.map_update(down, sr)
            assert info["n_iter"] == ir["n_iter"] and info["n_match"] == ir["n_match"]
            err = max(err, float(np.abs(st[:24] - sr[:24]).max()))
    out[rank] = (st[:24].copy(), err)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_register_matches_unsharded(world):
    """2 ranks, and configs[4]'s 8 (eight processes on the one GPU of the box; gloo carries the all-reduce and the scan broadcast)"""
    import torch.multiprocessing as mp
    port = 29600 + (os.getpid() % 300) + world
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_main, args=(world, port, out), nprocs=world, join=True)
    for r in range(1, world):
        np.testing.assert_array_equal(out[0][0], out[r][0])   # the ranks stay in lock step (identical all-reduced sums -> identical states)
    assert out[0][1] < 1e-9                                    # and agree with the unsharded run up to summation order


# ---------------------------------------------------------------------------------------------------------------------
# sharded mesher: owner-computed admission + kNN + Delaunay per mesh-voxel brick; only the boundary band travels (band survivors /
# decisions of the admission, smoothed positions and triangle marks within reach of another rank's brick); every rank reports the triangles whose
# smallest vertex lies in its bricks -- the UNION of the ranks' result lists must be the unsharded lists, every entry exactly once
# ---------------------------------------------------------------------------------------------------------------------
MESH_KEYS = ("new_vtx", "tri_add", "flip_add", "tri_rem", "tri_upd", "flip_upd", "smooth_ids", "smooth_xyz")


def _mesh_cfg(rank=0, world=0, brick_log2=2):
    return capi.avia_config(cap_root_voxels=1 << 14, cap_scan_points=200000, cap_vertices=1 << 17, cap_triangles=1 << 19,
                            shard_rank=rank, shard_world=world, shard_brick_log2=brick_log2, shard_mesh=1 if world > 1 else 0)   # 4-voxel (1.6 m) bricks by default


def _mesh_rank_main(rank, world, port, out, brick_log2, n_scans):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    lib = capi.load_hip_library()
    h = capi.HotPath(lib, _mesh_cfg(rank, world, brick_log2), "immesh_")

    def allgather(send, recv):
        parts = [torch.empty(len(send), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(send))
        for r in range(world):
            recv[r * len(send):(r + 1) * len(send)] = parts[r].numpy()
    h.set_allgather(allgather)
    ref = capi.HotPath(lib, _mesh_cfg(), "immesh_") if rank == 0 else None
    cfg = _mesh_cfg()
    extT = np.array(list(cfg.extT)); extR = np.array(list(cfg.extR)).reshape(3, 3)
    for k in range(n_scans):
        R, t = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, R, t, n_pts=40000, extT=extT)
        world_pts = raw.copy()
        world_pts[:, :3] = ((raw[:, :3].astype(np.float64) @ extR.T + extT) @ R.T + t).astype(np.float32)
        m = h.mesh_scan(world_pts, t, frame_idx=k)
        out[(rank, k)] = {key: np.array(m[key]) for key in MESH_KEYS}
        if ref:
            mr = ref.mesh_scan(world_pts, t, frame_idx=k)
            out[("ref", k)] = {key: np.array(mr[key]) for key in MESH_KEYS}
    cnt = h.counters()
    out[(rank, "cnt")] = (cnt["n_u"], cnt["n_vertices"], cnt["n_triangles_live"])
    if ref:
        rc = ref.counters()
        out[("ref", "cnt")] = (rc["n_u"], rc["n_vertices"], rc["n_triangles_live"])
    out[(rank, "traffic")] = h.shard_traffic()
    dist.barrier()
    dist.destroy_process_group()


def _lex(tri, flip=None):
    """rows of tri (n x 3) in lexicographic order, with their flips"""
    tri = np.asarray(tri).reshape(-1, 3)
    order = np.lexsort((tri[:, 2], tri[:, 1], tri[:, 0])) if len(tri) else np.zeros(0, int)
    return tri[order], (np.asarray(flip)[order] if flip is not None else None)


@pytest.mark.parametrize("world,brick_log2", [(2, 2), (4, 2), (2, 3), (8, 3)])   # (8, 3): configs[4]'s rank count and brick size, eight processes on the one GPU
def test_sharded_mesher_union_of_rank_lists_is_the_unsharded_result(world, brick_log2):
    import torch.multiprocessing as mp
    n_scans = 4
    port = 29900 + (os.getpid() % 90) + world
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_mesh_rank_main, args=(world, port, out, brick_log2, n_scans), nprocs=world, join=True)
    n_tri = 0
    shares = np.zeros(world)
    for k in range(n_scans):
        ref = out[("ref", k)]
        parts = [out[(r, k)] for r in range(world)]
        for p_ in parts:                                           # the vertex commit is replicated: same ids, same positions on every rank
            np.testing.assert_array_equal(p_["new_vtx"], ref["new_vtx"], err_msg=f"scan {k} new_vtx")
        for tri_key, flip_key in (("tri_add", "flip_add"), ("tri_rem", None), ("tri_upd", "flip_upd")):
            tri = np.concatenate([np.asarray(p_[tri_key]).reshape(-1, 3) for p_ in parts])
            flip = np.concatenate([np.asarray(p_[flip_key]).reshape(-1) for p_ in parts]) if flip_key else None
            tri_s, flip_s = _lex(tri, flip)
            np.testing.assert_array_equal(tri_s, np.asarray(ref[tri_key]).reshape(-1, 3), err_msg=f"scan {k} {tri_key}")   # same triangles, none twice
            if flip_key:
                np.testing.assert_array_equal(flip_s, np.asarray(ref[flip_key]).reshape(-1), err_msg=f"scan {k} {flip_key}")
        ids = np.concatenate([np.asarray(p_["smooth_ids"]).reshape(-1) for p_ in parts])
        xyz = np.concatenate([np.asarray(p_["smooth_xyz"]).reshape(-1, 3) for p_ in parts])
        order = np.argsort(ids, kind="stable")
        np.testing.assert_array_equal(ids[order], ref["smooth_ids"], err_msg=f"scan {k} smooth_ids")
        np.testing.assert_array_equal(xyz[order], np.asarray(ref["smooth_xyz"]).reshape(-1, 3), err_msg=f"scan {k} smooth_xyz")
        n_tri += len(ref["tri_add"])
        shares += [len(p_["tri_add"]) for p_ in parts]
    assert n_tri > 5000
    assert (shares > 0).all()                                      # every rank reported part of the mesh
    nu_ref, nv_ref, nl_ref = out[("ref", "cnt")]
    cnts = [out[(r, "cnt")] for r in range(world)]
    assert all(c[1] == nv_ref for c in cnts)                       # every rank knows every vertex (replicated 16-byte commit)
    assert sum(c[0] for c in cnts) == nu_ref                       # each voxel searched / triangulated on exactly one rank
    assert sum(c[2] for c in cnts) == nl_ref                       # live triangles: the ranks' reported parts add up
    for r in range(world):
        tr = out[(r, "traffic")]
        assert tr["bytes"] > 0 and 4 * n_scans <= tr["calls"] <= 8 * n_scans   # one all-gather per exchange: >= 2 admission rounds + smoothed band + mark band per scan


def test_rccl_inside_the_library_world_size_one(hip_lib):
    """RCCL called by the C++ layer itself (immesh_rccl_init: librccl.so opened at run time, ncclCommInitRank, then per residual pass
    residual_kernel -> ncclAllReduce on the device-resident 46 sums -> ekf_step_kernel, all enqueued on the context's stream without a host round
    trip).  One rank is all a one-GPU box can run: the collective degenerates to a copy, so the result must equal the unsharded path -- to rounding:
    the unsharded path is the resident-grid kernel (residual_persistent_kernel: block partials of 256 points, regrouped gain algebra), this one the
    per-pass chain.  What it exercises is the RCCL symbols, the communicator life cycle and the in-stream pass / reduce / update chain."""
    scans = _scans(4)
    cfg1 = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 16, cap_triangles=1 << 18, shard_rank=0, shard_world=1)
    h, ref, shadow = make_hip(hip_lib, cfg1), make_hip(hip_lib, _cfg()), make_hip(hip_lib, _cfg())
    h.rccl_init(h.rccl_unique_id())
    R0, t0, raw0, _ = scans[0]
    st = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    h.map_build(p0, st); ref.map_build(p0, st)
    st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
    sr = st.copy()
    for k in range(1, 4):
        down, raw = scans[k][3], scans[k][2]
        prior, pr = synth.forward_without_imu(st), synth.forward_without_imu(sr)
        d_down, n_down = h.broadcast_scan(down, root=0)            # ncclBroadcast (header + points) inside the library; one rank: a copy
        assert n_down == len(down)
        st, info = h.process_scan(d_down, raw, prior, prior, frame_idx=k, do_mesh=1, n_ds=n_down)
        sr, ir = ref.process_scan(down, raw, pr, pr, frame_idx=k, do_mesh=1)
        assert info == ir
        np.testing.assert_allclose(st[:24], sr[:24], rtol=0, atol=1e-9)
        np.testing.assert_allclose(st[24:], sr[24:], rtol=0, atol=1e-12)
        mh, mr = h.mesh_fetch(), ref.mesh_fetch()
        assert len(mh["tri_add"]) > 0
        # the two poses agree to ~1e-12, so a world-frame f32 coordinate may round the other way: at most one ulp apart, and the RCCL context's mesh
        # lists are compared EXACTLY on its own world-frame cloud (a second unsharded mesher is fed that cloud -- never a conditional assert)
        wh, wr = h.mesh_world_scan(), ref.mesh_world_scan()
        assert clouds_within_rounding(wh[:, :3], wr[:, :3])
        ms = shadow.mesh_scan(wh, st[9:12], frame_idx=k)
        for key in ("new_vtx", "tri_add", "tri_rem", "tri_upd", "flip_add", "flip_upd", "smooth_ids"):
            np.testing.assert_array_equal(mh[key], ms[key], err_msg=f"scan {k} {key}")
    h.close(); ref.close(); shadow.close()


def test_bench_gpus_flag_spawns_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 --backend gloo` as ONE command (not under torch.distributed.run): it re-executes itself as two ranks -- both on
    cuda:0 here, the pool has one GPU per box -- and the line's headline is the sharded job (n_gpus 2, strong), the replica leg rides along (weak)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2", "--map-voxels", "150000",
                        "--pts", "30000", "--profile-scans", "0", "--cpu-seconds", "0", "--extra-configs", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        print(r.stderr[-8000:])   # (pytest shows captured output in full; its assertion repr truncates)
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    # N > 1: the headline is the sharded job (ONE stream, voxel bricks over the ranks, strong scaling); the replica leg is reported beside it
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] == d["sharded"]["value"] > 0 and d["replicas"]["scaling"] == "weak" and d["replicas"]["value"] > 0
    assert d["sharded"]["scaling"] == "strong" and "gloo" in d["sharded"]["collectives"] and d["sharded"]["pose_err_m"] < 0.1
    lb = d["sharded"]["load_balance_point_share_per_brick_size"]
    assert set(lb) == {"8", "16", "32"} and all(0.5 <= v["max_share_mean"] <= 1.0 and v["fair_share"] == 0.5 for v in lb.values())
    assert 4 <= d["sharded"]["exchange_calls_per_scan"] <= 8      # one all-gather per exchange: >= 2 admission rounds + the two band exchanges
