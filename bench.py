#!/usr/bin/env python3
"""bench.py -- scans/s of the MI355X-native ImMesh hot path (BASELINE.json metric).

One "step" = one LiDAR scan through the whole per-scan hot path behind the C ABI (immesh_process_scan):
iterated-EKF point-to-plane registration against the HBM-resident voxel/plane map, map growth, and (``--mesh 1``)
incremental voxel-wise meshing.  Workload at N=1: the synthetic Livox-Avia-shaped 100k-pt/scan stream of SURVEY.md
8(d) C2/C3 into a pre-built ~10 M-root-voxel map; scans and the down-sampled clouds are resident in HBM before the
timed region starts.  N>1 (torchrun, one rank per GPU): every rank runs the same stream against its own map shard
replica-free -- see DESIGN.md "Multi-GPU" -- weak scaling, max-over-ranks timing.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from immesh_amd import capi, synth  # noqa: E402
from immesh_amd import dist as D  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_big_map(h, cfg, torch, dev, target_voxels, side_m, seed=20260924):
    """Pre-build the registration map by streaming a dense survey of the procedural world (synth.py geometry) through
    immesh_map_update, strip by strip, generated on the GPU (harness only; the product never sees torch)."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    extT = torch.tensor(list(cfg.extT), device=dev, dtype=torch.float32)
    st = capi.make_state()  # identity pose: p_world = extR p + extT  ->  p_body = p_world - extT
    L = synth.LATTICE
    x0 = -side_m / 2 + 50.0
    nstrip = int(np.ceil(side_m / L))
    gs, ws = 1.0 / 6.0, 1.0 / 8.0   # survey lattice spacing: ground 36 pts/m^2, walls 64 pts/m^2
    t0 = time.time()
    nv = 0
    for si in range(nstrip):
        xs0 = (np.floor(x0 / L) + si) * L
        # ground
        nx, ny = int(L / gs), int(side_m / gs)
        ix = torch.arange(nx, device=dev, dtype=torch.float32)
        iy = torch.arange(ny, device=dev, dtype=torch.float32)
        X = (xs0 + (ix[:, None] + 0.5) * gs).expand(nx, ny).reshape(-1)
        Y = (-side_m / 2 + (iy[None, :] + 0.5) * gs).expand(nx, ny).reshape(-1)
        X = X + (torch.rand(X.shape, device=dev, generator=g) - 0.5) * 0.8 * gs
        Y = Y + (torch.rand(Y.shape, device=dev, generator=g) - 0.5) * 0.8 * gs
        fx = torch.remainder(X, L); fy = torch.remainder(Y, L)
        keep = ~((fx > synth.BOX_LO) & (fx < synth.BOX_HI) & (fy > synth.BOX_LO) & (fy < synth.BOX_HI))
        X, Y = X[keep], Y[keep]
        Z = synth.GROUND_Z + torch.randn(X.shape, device=dev, generator=g) * 0.02
        parts = [torch.stack([X, Y, Z], dim=1)]
        # walls of the boxes in this strip
        nb = int(np.ceil(side_m / L))
        jy = torch.arange(nb, device=dev, dtype=torch.float32) + np.floor(-side_m / 2 / L)
        nu = int((synth.BOX_HI - synth.BOX_LO) / ws); nz = int((synth.BOX_TOP - synth.GROUND_Z) / ws)
        iu = torch.arange(nu, device=dev, dtype=torch.float32); iz = torch.arange(nz, device=dev, dtype=torch.float32)
        U = (synth.BOX_LO + (iu[:, None] + 0.5) * ws).expand(nu, nz).reshape(-1)
        Zw = (synth.GROUND_Z + (iz[None, :] + 0.5) * ws).expand(nu, nz).reshape(-1)
        for axis, val in ((0, synth.BOX_LO), (0, synth.BOX_HI), (1, synth.BOX_LO), (1, synth.BOX_HI)):
            m = len(U) * nb
            u = U[None, :].expand(nb, -1).reshape(-1) + (torch.rand(m, device=dev, generator=g) - 0.5) * 0.8 * ws
            z = Zw[None, :].expand(nb, -1).reshape(-1) + (torch.rand(m, device=dev, generator=g) - 0.5) * 0.8 * ws
            w = val + torch.randn(m, device=dev, generator=g) * 0.02
            by = (jy[:, None] * L).expand(nb, len(U)).reshape(-1)
            if axis == 0:
                parts.append(torch.stack([xs0 + w, by + u, z], dim=1))
            else:
                parts.append(torch.stack([xs0 + u, by + w, z], dim=1))
        P = (torch.cat(parts, dim=0) - extT[None, :]).contiguous()
        torch.cuda.synchronize()
        cap = int(cfg.cap_scan_points)
        for a in range(0, P.shape[0], cap):
            chunk = P[a:a + cap]
            h.map_update(chunk.data_ptr(), st, n=chunk.shape[0])
        nv = h.counters()["n_root_voxels"]
        if nv >= target_voxels:
            break
    log(f"[bench] map pre-build: {nv} root voxels, {si + 1} strips, {time.time() - t0:.1f} s")
    return nv


def make_scans(n_scans, n_pts, cfg, cache_dir, kitti=False):
    """Synthetic Livox-shaped stream: raw scans (lidar frame, xyzI) + the VoxelGrid-downsampled clouds."""
    os.makedirs(cache_dir, exist_ok=True)
    extT = np.array(list(cfg.extT))
    raws, downs = [], []
    for k in range(n_scans):
        f = os.path.join(cache_dir, f"hdl64_{k}.npy" if kitti else f"livox_{n_pts}_{k}.npy")
        if os.path.exists(f):
            raw = np.load(f)
        else:
            R, t = synth.trajectory_pose(k)
            raw = synth.hdl64_scan(k, R, t) if kitti else synth.livox_scan(k, R, t, n_pts=n_pts, extT=extT)
            tmp = f + f".{os.getpid()}.tmp.npy"     # ranks generate the same scans concurrently: publish atomically
            np.save(tmp, raw)
            os.replace(tmp, f)
        raws.append(raw)
        downs.append(synth.voxel_grid_downsample(raw, 0.5 if kitti else 0.4))   # filter_size_surf: avia.yaml:5 / velodyne.yaml:5
    return raws, downs


# algorithmic bytes of one launch of each kernel (SURVEY.md 8(d) record sizes; DESIGN.md "Kernels")
def algorithmic_bytes(kname, c, n_scans, n_raw):
    n_ds = c["_n_ds_mean"]; iters = max(1, c["n_iter"]) / n_scans
    kname = kname.split("<")[0]
    if kname == "residual_kernel":      # per EKF iteration: point 12 B + key/slot 12 B per point, 12 B per extra probe, 229 B per plane test
        launches = c["n_iter"]
        return (n_ds * 24 * launches + c["n_extra_probe"] * 12 + c["n_plane_tests"] * 229) / launches
    if kname == "point_var_kernel":     # point 12 B in, Point_with_var 96 B + sort key 8 + slot 12 out
        return n_ds * (12 + 96 + 8 + 12)
    if kname == "replay_kernel":        # per scan: every point record once (96 B) + every refit re-reads its retained points (96 B each) and writes a plane (229 B)
        return n_ds * 96 + (c["n_refit_pts"] * 96 + c["n_refits"] * 229) / n_scans
    if kname == "mesh_knn_kernel":      # per scan: C20 inspected vertices x 12 B + query 12 B + 20 ids out
        return (c["c20"] * 12 + c["n_v"] * (12 + 80 + 24)) / n_scans
    if kname == "mesh_delaunay_kernel":  # per scan: n_u x (12 B pos + 6 x 12 B incident triangles) + T_v x 12 B
        return (c["n_u"] * 84 + c["t_v"] * 12) / n_scans
    if kname == "mesh_transform_kernel":
        return n_raw * 32
    return None


def roofline_from_committed_profile(kernel, cnt, args, note):
    """Fallback when the live HIP-event leg could not run: average launch time of `kernel` from profiles/r01_full_kernel_stats.csv (rocprofv3
    --kernel-trace --stats of this script), algorithmic bytes from THIS run's counters.  Clearly labelled in `source`."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01_full_kernel_stats.csv")
    try:
        avg_ns = None
        for r in csv.DictReader(open(path)):
            n = r["Name"].strip('"')
            n = n[5:] if n.startswith("void ") else n
            if n.startswith(kernel + "("):
                avg_ns = float(r["AverageNs"])
                break
        if avg_ns is None:
            return None
        c = dict(cnt)
        by = algorithmic_bytes(kernel, c, args.steps, args.pts)
        if by is None:
            return None
        avg_ms = avg_ns * 1e-6
        ach = by / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": int(by),
                "source": "profiles/r01_full_kernel_stats.csv (committed rocprofv3 average) -- live HIP-event leg unavailable: " + (note or "unknown")}
    except Exception:   # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pts", type=int, default=100000, help="raw points per scan")
    ap.add_argument("--map-voxels", type=float, default=10e6, help="root voxels of the pre-built registration map")
    ap.add_argument("--mesh", type=int, default=1, help="1 = full pipeline (configs[2]); 0 = registration + map update only (configs[1])")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU budget of the oracle baseline leg (0 = skip)")
    ap.add_argument("--profile-scans", type=int, default=5)
    ap.add_argument("--config", choices=["avia", "velodyne"], default="avia", help="avia = BASELINE configs[1]/[2] (the metric's workload); velodyne = configs[3], KITTI-shaped HDL-64 scans with velodyne.yaml parameters")
    ap.add_argument("--shard", type=int, default=0, help="N>1 only. 0 = replicas (every rank its own stream + map, weak scaling); 1 = ONE stream, registration map sharded by "
                    "root-voxel bricks over the ranks, 46-double all-reduce per EKF iteration, meshing on rank 0 (strong scaling; the capacity mode of configs[4])")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo only for single-GPU functional tests)")
    ap.add_argument("--device-downsample", type=int, default=0, help="1 = the VoxelGrid down-sampling of every raw scan also runs on the device inside the timed region (SURVEY 8(f) rank 1)")
    ap.add_argument("--async-mesh", type=int, default=1, help="1 = meshing of scan k overlaps registration of scan k+1 (the reference's mesh service thread); 0 = strictly serial per scan")
    ap.add_argument("--profile-child", type=int, default=0, help="internal (spawned by the parent run): 1 = serial stage timing, 2 = HIP-event profile, 3 = both; prints their JSON only")
    ap.add_argument("--profile-inproc", type=int, default=0, help="1 = run the profile legs inside this process instead of a child process")
    ap.add_argument("--profile-timeout", type=float, default=90.0, help="seconds the parent waits for the profile child")
    args = ap.parse_args()

    import torch
    rank, world, local = D.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    local = local % torch.cuda.device_count()   # one GPU per rank on a real node; functional tests may stack ranks on one device (gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = D.init(args.backend, dev if args.backend == "nccl" else None) if world > 1 else None
    sharded = bool(args.shard) and world > 1

    hip = capi.load_hip_library()
    side = float(np.sqrt(args.map_voxels / 8.8)) + 40.0     # ~8.8 root voxels per m^2 of this world (ground + walls)
    kitti = args.config == "velodyne"
    if kitti:
        cfg = capi.velodyne_config(device=local, cap_root_voxels=1 << 18, cap_scan_points=400_000, cap_vertices=1 << 24, cap_triangles=1 << 25)
    else:
        # sharded map: a rank keeps its bricks plus the one-voxel halo (~20 % at 32^3-voxel bricks) -> capacity per rank, not per job
        share = (1.5 / world) if (bool(args.shard) and world > 1) else 1.0
        cfg = capi.avia_config(device=local, cap_root_voxels=int(args.map_voxels * 1.3 * share) + (1 << 16), cap_scan_points=2_500_000,
                               cap_vertices=1 << 24, cap_triangles=1 << 25)
    if sharded:
        cfg.shard_rank, cfg.shard_world, cfg.shard_brick_log2, cfg.shard_mesh = rank, world, 5, 1 if args.mesh else 0
    h = capi.HotPath(hip, cfg, "immesh_")
    if sharded:
        if args.backend == "nccl":
            red = torch.zeros(46, dtype=torch.float64, device=dev)

            def _allreduce(buf):                      # 46 doubles: H^T R^-1 H, H^T R^-1 z, counters -- RCCL all-reduce over xGMI
                red.copy_(torch.from_numpy(buf)); dist.all_reduce(red); buf[:] = red.cpu().numpy()
        else:
            def _allreduce(buf):
                dist.all_reduce(torch.from_numpy(buf))
        h.set_allreduce(_allreduce)
        if args.mesh:   # sharded mesher: all-gather of this scan's smoothed vertices and triangle marks (two exchanges per scan)
            def _allgather(send, recv):
                if args.backend == "nccl":
                    src = torch.from_numpy(send).to(dev)
                    dst = torch.empty(len(send) * world, dtype=torch.uint8, device=dev)
                    dist.all_gather_into_tensor(dst, src)
                    recv[:] = dst.cpu().numpy()
                else:
                    parts = [torch.empty(len(send), dtype=torch.uint8) for _ in range(world)]
                    dist.all_gather(parts, torch.from_numpy(send))
                    for r_ in range(world):
                        recv[r_ * len(send):(r_ + 1) * len(send)] = parts[r_].numpy()
            h.set_allgather(_allgather)
    n_total = args.warmup + args.steps + 2 * args.profile_scans
    raws, downs = make_scans(n_total + 1 + (world - 1), args.pts, cfg, os.path.join(os.environ.get("TMPDIR", "/tmp"), "immesh_scan_cache"), kitti)
    if kitti:   # the map grows from the stream itself (3 m root voxels, max_layer 4)
        R0_, t0_ = synth.trajectory_pose(0)
        h.map_build(np.ascontiguousarray(raws[0][:, :3]), capi.make_state(R=R0_, t=t0_))
        n_map = h.counters()["n_root_voxels"]
    else:
        n_map = build_big_map(h, cfg, torch, dev, args.map_voxels, side)
    idx = D.stream_of_rank(0 if sharded else rank, n_total)   # replicas: rank r replays the stream phase-shifted by r scans; sharded: one stream
    raws, downs = [raws[i] for i in idx], [downs[i] for i in idx]
    d_raw = [torch.from_numpy(r).to(dev) for r in raws]
    d_down = [torch.from_numpy(d).to(dev) for d in downs]
    n_ds_mean = float(np.mean([len(d) for d in downs[1:1 + args.warmup + args.steps]]))

    # scan 0 seeds the stream state; constant-velocity prior (Forward_without_imu) between scans
    R0, t0 = synth.trajectory_pose(idx[0])
    st = capi.make_state(R=R0, t=t0)
    st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
    NOWAIT = 0x10   # IMMESH_SCAN_NOWAIT: return once the pose is final, map growth finishes on the stream ahead of the next scan's work
    mesh_mode = (2 if (args.async_mesh and not sharded) else 1) if args.mesh else (NOWAIT if args.async_mesh else 0)   # sharded mesher: serial per scan (its collectives must not interleave)
    if mesh_mode & 3:
        # mesh map is seeded by scan 0 (the registration map is the pre-built survey); sharded: every rank takes part in the scan's all-reduces
        h.process_scan(d_down[0].data_ptr(), d_raw[0].data_ptr(), st, st, frame_idx=0, do_mesh=1 if (mesh_mode & 3) else 0, n_ds=len(downs[0]), n_raw=len(raws[0]))


    def run(k, state, mode=None):
        prior = capi.forward_without_imu_native(hip, state)     # constant-velocity prior (Forward_without_imu), host side of the library
        down_ptr, n_ds = d_down[k].data_ptr(), len(downs[k])
        if args.device_downsample:
            _, n_ds = h.downsample(d_raw[k].data_ptr(), 0.5 if kitti else 0.4, n=len(raws[k]), stride=4, to_host=False)
            down_ptr = h.downsample_result_ptr()
        out, info = h.process_scan(down_ptr, d_raw[k].data_ptr(), prior, prior, frame_idx=k, do_mesh=mesh_mode if mode is None else mode,
                                   n_ds=n_ds, n_raw=len(raws[k]))
        return out, info

    k = 1
    for _ in range(args.warmup):
        st, _ = run(k, st); k += 1
    h.counters(reset=True)
    stage = np.zeros(4)
    torch.cuda.synchronize()
    D.barrier()
    t_begin = time.perf_counter()
    t_marks = [t_begin]
    for _ in range(args.steps):
        st, info = run(k, st); k += 1
        t_marks.append(time.perf_counter())
    if mesh_mode == 2:
        h.mesh_wait()          # drain the mesher: every scan of the timed region is fully meshed before the clock stops
    h.last_timing()            # waits for the last scan's map update (the library's own stream)
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t_begin, dev)
    cnt = h.counters()
    cnt["_n_ds_mean"] = n_ds_mean
    pose_err = float(np.linalg.norm(st[9:12] - synth.trajectory_pose(idx[k - 1])[1]))

    # ---- roofline leg: serial per-stage times + per-kernel HIP-event timing (events recorded on the library's own streams) over extra scans.
    # It runs in a CHILD PROCESS of this script (same workload, its own context on the same GPU) with a timeout: the headline number above is
    # complete before it starts, and a failure of the instrumented legs cannot take the bench line with it.  The child uses the serial launch
    # order and unmasked mesher streams (IMMESH_SERIAL_ORDER / IMMESH_MESH_CUS=0): the configuration the per-kernel numbers describe best.
    def profile_legs(k, st, what=3):   # bit 0: serial stage timing, bit 1: HIP-event profile
        res = {}
        if what & 1:
            pstage = np.zeros(4)
            for _ in range(args.profile_scans):
                st, _ = run(k, st, mode=1 if (mesh_mode & 3) else 0); k += 1    # serial mode, profiler off: per-stage times of one scan
                tm = h.last_timing()
                pstage += [tm["total"], tm["register"], tm["map_update"], tm["mesh"]]
            res["stage_per_scan"] = list(map(float, pstage / max(1, args.profile_scans)))
        if what & 2:
            h.counters(reset=True)
            h.profile_enable(True)
            k0 = k
            for _ in range(args.profile_scans):
                st, _ = run(k, st, mode=1 if (mesh_mode & 3) else 0); k += 1    # serial mode, HIP events around every launch
            ks = h.profile_read()
            h.profile_enable(False)
            pc_ = h.counters(); pc_["_n_ds_mean"] = float(np.mean([len(d) for d in downs[k0:k]]))
            res["kstats"] = ks
            res["pc"] = {kk_: float(v) for kk_, v in pc_.items()}
        return res

    if args.profile_child:
        print(json.dumps(profile_legs(k, st, args.profile_child)), flush=True)
        return

    roofline = None
    kstats = {}
    prof = None
    prof_note = None
    if (rank == 0 or sharded) and args.profile_scans > 0:   # sharded: every rank takes part in the all-reduces of the extra scans
        if sharded or args.profile_inproc:
            # sharded: every rank takes part in the extra scans' collectives, so they run in-process -- uninstrumented stage timing only unless
            # --profile-inproc asks for the HIP-event leg too (open issue)
            prof = profile_legs(k, st, 3 if args.profile_inproc else 1)
            k += 2 * args.profile_scans
        else:
            # two children, so that a failure of the profiler leg does not cost the (uninstrumented) stage timing
            env = {kk_: v for kk_, v in os.environ.items() if kk_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            env.update({"IMMESH_SERIAL_ORDER": "1", "IMMESH_MESH_CUS": "0"})
            if world > 1 and "HIP_VISIBLE_DEVICES" not in env:
                env["HIP_VISIBLE_DEVICES"] = str(local)
            prof = {}
            notes = []
            for what in (1, 2):
                cmd = [sys.executable, os.path.abspath(sys.argv[0]), "--profile-child", str(what), "--gpus", "1", "--steps", "0", "--warmup", str(min(args.warmup, 5)),
                       "--pts", str(args.pts), "--map-voxels", str(args.map_voxels), "--mesh", str(args.mesh), "--cpu-seconds", "0", "--profile-scans", str(args.profile_scans),
                       "--config", args.config, "--device-downsample", str(args.device_downsample), "--async-mesh", str(args.async_mesh)]
                label = "stage-timing child" if what == 1 else "profiler child"
                try:
                    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.profile_timeout, text=True)
                    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                    if r.returncode == 0 and lines:
                        prof.update(json.loads(lines[-1]))
                    else:
                        notes.append(f"{label} failed (rc {r.returncode}): {(r.stderr or '').strip().splitlines()[-1][:200] if (r.stderr or '').strip() else 'no output'}")
                except subprocess.TimeoutExpired:
                    notes.append(f"{label} timed out after {args.profile_timeout:.0f} s")
                except Exception as e:   # noqa: BLE001
                    notes.append(f"{label} could not run: {e}")
            prof_note = "; ".join(notes) if notes else None
    if prof and "stage_per_scan" in prof:
        stage = np.array(prof["stage_per_scan"]) * args.steps
    if prof and "kstats" in prof:
        kstats = prof["kstats"]
        pc = prof["pc"]
        best = None
        for name, s_ in kstats.items():
            if s_["launches"] and algorithmic_bytes(name, pc, args.profile_scans, args.pts) is not None:
                if best is None or s_["total_ms"] > kstats[best]["total_ms"]:
                    best = name
        if best:
            per_scan_launches = kstats[best]["launches"] / args.profile_scans
            by = algorithmic_bytes(best, pc, args.profile_scans, args.pts)
            if best.split("<")[0] not in ("residual_kernel",):
                by = by / max(1.0, per_scan_launches)
            avg_ms = kstats[best]["total_ms"] / kstats[best]["launches"]
            ach = by / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": best, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None, "avg_launch_ms": round(avg_ms, 5),
                        "algorithmic_bytes_per_launch": int(by), "source": "HIP events, live (instrumented child run of this script: serial launch order, unmasked mesher streams)"}
    if roofline is None and rank == 0 and args.profile_scans > 0:
        # the live leg is unavailable: fall back to the committed rocprofv3 average of the dominant kernel, with this run's own counters
        roofline = roofline_from_committed_profile("mesh_delaunay_kernel<256>" if args.mesh else "residual_kernel", cnt, args, prof_note)
    if roofline is not None:
        tr = os.path.join(ROOT, "profiles", "traffic_r01.json")   # PMC-derived HBM bytes/launch from the committed rocprofv3 --pmc passes
        if os.path.exists(tr):
            try:
                roofline["traffic"] = json.load(open(tr)).get(roofline["kernel"])
            except Exception:
                pass

    # ---- CPU baseline leg: the oracle (CPU restatement, "port") on the host cores, bounded sample of the same stream.  Two passes over the same
    # scans, half the budget each: single-threaded, and with the reference's own threading (a 12-thread pool over mesh voxels,
    # maximum_thread_for_rec_mesh; 4 OpenMP threads in the matcher, MP_PROC_NUM) -- the results are identical, the faster pass is reported.
    cpu = None
    if rank == 0 and args.cpu_seconds > 0:
        orc_so = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(orc_so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        orc_lib = ctypes.CDLL(orc_so)

        def cpu_pass(mesher_threads, matcher_threads, budget):
            o = capi.HotPath(orc_lib, cfg, "orc_")
            o.set_threads(mesher_threads, matcher_threads)
            so = capi.make_state(R=R0, t=t0)
            o.map_build(np.ascontiguousarray(raws[0][:, :3]), so)   # (kitti: exactly the GPU leg's map; avia: a local map instead of the 10M-voxel survey)
            so[12:15] = [1.0, 0, 0]; so[15:18] = [0, 0, np.deg2rad(2.0)]
            if args.mesh:
                o.process_scan(downs[0], raws[0], so, so, frame_idx=0, do_mesh=True)
            tc, nc, kk = 0.0, 0, 1
            while tc < budget and kk < len(raws):
                prior = synth.forward_without_imu(so)
                a = time.perf_counter()
                so, _ = o.process_scan(downs[kk], raws[kk], prior, prior, frame_idx=kk, do_mesh=bool(args.mesh))
                tc += time.perf_counter() - a
                nc += 1; kk += 1
            o.close() if hasattr(o, "close") else None
            return nc, tc

        ncores = os.cpu_count() or 1
        n1, t1 = cpu_pass(1, 1, args.cpu_seconds / 2)
        mt = (min(12, ncores), min(4, ncores))
        n2, t2 = cpu_pass(mt[0], mt[1], args.cpu_seconds / 2) if ncores > 1 else (n1, t1)
        single, threaded = n1 / t1, n2 / t2
        use_threads = threaded > single
        nc, tc = (n2, t2) if use_threads else (n1, t1)
        cpu = {"value": round(nc / tc, 4), "unit": "scans/s", "cores": mt[0] if use_threads else 1, "kind": "port",
               "sample": f"{nc} scans of the same stream through oracle/liboracle.so (" +
                         (f"{mt[0]} threads over mesh voxels, {mt[1]} in the matcher -- the reference's own threading" if use_threads else "single thread") +
                         "), map = scan 0 + growth (not the 10M-voxel map)",
               "ms_per_scan": round(1e3 * tc / nc, 3), "single_thread_value": round(single, 4), "reference_threading_value": round(threaded, 4)}

    if rank == 0:
        out = {
            "metric": ("scans/sec (reg+mesh), KITTI-shaped 130k-ray scans" if kitti else "scans/sec (reg+mesh), 100k-pt scan into 10M-voxel map") if args.mesh else
                      "scans/sec (registration + map update, meshing off), 100k-pt scan into 10M-voxel map",
            "value": round(D.aggregate_throughput(args.steps, 1 if sharded else world, elapsed), 4), "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (("synthetic KITTI-shaped HDL-64 scan stream (velodyne.yaml), " if kitti else "synthetic Livox-Avia 100k-pt/scan stream, ") +
                                    ("full pipeline (registration + map update + voxel meshing)" if args.mesh else "registration + map update, meshing off")),
                       "n_raw": int(np.mean([len(r) for r in raws])), "n_ds_mean": round(n_ds_mean, 1), "map_root_voxels": int(n_map), "params": "config/velodyne.yaml" if kitti else "config/avia.yaml",
                       "parallelism": (f"one stream; registration map sharded over {world} GPUs (brick ownership + 1-voxel halo, all-reduce of 46 doubles per EKF iteration); mesher sharded by mesh-voxel bricks (replicated vertex admission, owner-computes kNN + Delaunay, all-gather of smoothed vertices and triangle marks)" if sharded
                                       else f"{world} independent scan streams, one per GPU" if world > 1 else "1 GPU"),
                       "mesh_mode": {0: "off", NOWAIT: "off (map update of scan k overlaps the host side of scan k+1)", 1: "serial", 2: "async (mesh of scan k overlaps registration of scan k+1; vertex admission + kNN of scan k+1 overlap triangulation of scan k)"}[mesh_mode],
                       "downsample": "device (inside the timed region)" if args.device_downsample else "host (before the timed region; the hot path starts at lio_state_estimation)"},
            "stages_ms_serial": {"gpu_total": round(stage[0] / max(1, args.steps), 4), "register": round(stage[1] / max(1, args.steps), 4),
                          "map_update": round(stage[2] / max(1, args.steps), 4), "mesh": round(stage[3] / max(1, args.steps), 4)},
            "counters_per_scan": {kk_: round(v / args.steps, 1) for kk_, v in cnt.items() if kk_ in ("n_iter", "n_match", "n_plane_tests", "n_extra_probe", "n_refits", "n_new", "v_act", "n_u", "t_add", "t_rem")},
            "scan_thread_ms": ({"p50": round(float(np.percentile(np.diff(t_marks) * 1e3, 50)), 4), "p95": round(float(np.percentile(np.diff(t_marks) * 1e3, 95)), 4)}
                               if len(t_marks) > 2 else None),   # host time per immesh_process_scan call (asynchronous mode: until the pose is final)
            "pose_err_m": round(pose_err, 4),
            "roofline": roofline, "cpu_baseline": cpu,
            "kernels_ms_per_scan": {n: round(s["total_ms"] / max(1, args.profile_scans), 4) for n, s in sorted(kstats.items(), key=lambda kv: -kv[1]["total_ms"])},
        }
        if prof_note:
            out["profile_leg_note"] = prof_note
        print(json.dumps(out), flush=True)
    if world > 1:
        D.barrier()          # rank 0 may still be in its roofline / CPU-baseline legs: leave together
        dist.destroy_process_group()
    if prof_note:            # the instrumented child was killed: do not risk the runtime teardown of this process on a possibly disturbed device
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
