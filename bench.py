#!/usr/bin/env python3
"""bench.py -- scans/s of the MI355X-native ImMesh hot path (BASELINE.json metric).

One "step" = one LiDAR scan through the whole per-scan hot path behind the C ABI (immesh_process_scan):
iterated-EKF point-to-plane registration against the HBM-resident voxel/plane map, map growth, and (``--mesh 1``)
incremental voxel-wise meshing.  Workload at N=1: the synthetic Livox-Avia-shaped 100k-pt/scan stream of SURVEY.md
8(d) C2/C3 into a pre-built ~10 M-root-voxel map; scans and the down-sampled clouds are resident in HBM before the
timed region starts.  N>1: one rank per GPU (``python bench.py --gpus N`` re-executes itself under torch.distributed.run
when it was not launched by it).  Headline = N independent scan streams, one per GPU (weak scaling, max-over-ranks timing);
the ``sharded`` object of the same line is the north-star split measured right after it: ONE stream, registration map and
mesher sharded by voxel bricks over the ranks (strong scaling) -- see DESIGN.md "Multi-GPU".

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from immesh_amd import capi, synth  # noqa: E402
from immesh_amd import dist as D  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def survey_strips(cfg, torch, dev, side_m, seed=20260924):
    """Dense survey of the procedural world (synth.py geometry), strip by strip, generated on the GPU: yields (n, 3) float32 device tensors of
    body-frame points for the identity pose (harness only; the product never sees torch)."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    extT = torch.tensor(list(cfg.extT), device=dev, dtype=torch.float32)   # identity pose: p_world = extR p + extT  ->  p_body = p_world - extT
    L = synth.LATTICE
    x0 = -side_m / 2 + 50.0
    nstrip = int(np.ceil(side_m / L))
    gs, ws = 1.0 / 6.0, 1.0 / 8.0   # survey lattice spacing: ground 36 pts/m^2, walls 64 pts/m^2
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
        if P.is_cuda:
            torch.cuda.synchronize()
        yield P


def build_big_map(h, cfg, torch, dev, target_voxels, side_m, seed=20260924, also=None):
    """Pre-build the registration map by streaming the survey through immesh_map_update.  `also`: a second context (the CPU oracle in the
    full-size parity tests) that is fed the very same chunks from host copies."""
    st = capi.make_state()
    t0 = time.time()
    nv, si = 0, 0
    cap = int(cfg.cap_scan_points)
    for si, P in enumerate(survey_strips(cfg, torch, dev, side_m, seed)):
        for a in range(0, P.shape[0], cap):
            chunk = P[a:a + cap]
            h.map_update(chunk.data_ptr(), st, n=chunk.shape[0])
            if also is not None:
                also.map_update(np.ascontiguousarray(chunk.cpu().numpy()), st)
        nv = h.counters()["n_root_voxels"]
        if nv >= target_voxels:
            break
    log(f"[bench] map pre-build: {nv} root voxels, {si + 1} strips, {time.time() - t0:.1f} s")
    return nv


def livox_scan_torch(torch, dev, k, R, t, n_pts, extT, seed=20260924, range_sigma=0.02, bearing_sigma_deg=0.05, blind=1.0):
    """synth.livox_scan on the GPU (harness only: the long steady-state leg needs hundreds of scans, the numpy ray caster takes seconds per scan).
    Same scan model -- Halton(2,3) directions offset per scan, the same procedural world, range / bearing noise -- with torch's generator."""
    m = int(n_pts * 2.2) + 64
    start = 1 + k * m
    az = (synth.halton(m, 2, start) - 0.5) * np.deg2rad(70.4)
    el = (synth.halton(m, 3, start) - 0.5) * np.deg2rad(77.2)
    dirs = torch.from_numpy(np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)).to(dev)
    Rt = torch.from_numpy(np.asarray(R, dtype=np.float64)).to(dev)
    o = np.asarray(R, dtype=np.float64) @ np.asarray(extT, dtype=np.float64) + np.asarray(t, dtype=np.float64)
    d = dirs @ Rt.T
    L = synth.LATTICE
    inf = float("inf")
    tb = torch.full((m,), inf, dtype=torch.float64, device=dev)
    tg = (synth.GROUND_Z - o[2]) / d[:, 2]
    ok = (d[:, 2] < 0) & (tg > 0)
    tb = torch.where(ok, tg, tb)
    inv = 1.0 / torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
    ci, cj, rc = int(np.floor(o[0] / L)), int(np.floor(o[1] / L)), int(np.ceil(100.0 / L)) + 1
    ot = torch.from_numpy(o).to(dev)
    for i in range(ci - rc, ci + rc + 1):
        for j in range(cj - rc, cj + rc + 1):
            lo = np.array([i * L + synth.BOX_LO, j * L + synth.BOX_LO, synth.GROUND_Z]); hi = np.array([i * L + synth.BOX_HI, j * L + synth.BOX_HI, synth.BOX_TOP])
            if np.linalg.norm(0.5 * (lo + hi)[:2] - o[:2]) > 108.0:
                continue
            t1 = (torch.from_numpy(lo).to(dev) - ot) * inv; t2 = (torch.from_numpy(hi).to(dev) - ot) * inv
            tn = torch.minimum(t1, t2).max(dim=1).values; tf = torch.maximum(t1, t2).min(dim=1).values
            hit = (tn <= tf) & (tn > 0) & (tn < tb)
            tb = torch.where(hit, tn, tb)
    keep = torch.isfinite(tb) & (tb <= 100.0) & (tb > blind)
    dirs, rg = dirs[keep][:n_pts], tb[keep][:n_pts]
    g = torch.Generator(device=dev); g.manual_seed(seed + 7919 * k)
    n = rg.shape[0]
    pert = torch.randn((n, 3), device=dev, dtype=torch.float64, generator=g) * np.deg2rad(bearing_sigma_deg)
    dn = dirs + torch.linalg.cross(pert, dirs)
    dn = dn / dn.norm(dim=1, keepdim=True)
    rn = rg + torch.randn(n, device=dev, dtype=torch.float64, generator=g) * range_sigma
    inten = 10.0 + 80.0 * torch.rand(n, device=dev, dtype=torch.float64, generator=g)
    return torch.cat([dn * rn[:, None], inten[:, None]], dim=1).to(torch.float32).contiguous()


def corridor_cloud(torch, dev, n_scans, spacing=0.07, reach=45.0, seed=20260926):
    """Dense cloud (world frame, (n, 4) float32 xyzI) of the world surfaces around the stream's trajectory: what seeds the MESH map to the density a
    surveyed area has ("mesh map pre-seeded from the same survey, capped at the corridor").  Jittered lattice of `spacing` on the
    ground and on the walls of the boxes inside the corridor: finer than the 0.1 m minimum vertex spacing, so the vertex lattice saturates."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    ts = np.array([synth.trajectory_pose(k)[1] for k in range(n_scans + 1)])
    x0, x1, y0, y1 = ts[:, 0].min() - 5.0, ts[:, 0].max() + reach, ts[:, 1].min() - reach, ts[:, 1].max() + reach
    L = synth.LATTICE
    xs = torch.arange(x0, x1, spacing, device=dev, dtype=torch.float32); ys = torch.arange(y0, y1, spacing, device=dev, dtype=torch.float32)
    X = xs[:, None].expand(len(xs), len(ys)).reshape(-1); Y = ys[None, :].expand(len(xs), len(ys)).reshape(-1)
    X = X + (torch.rand(X.shape, device=dev, generator=g) - 0.5) * 0.8 * spacing; Y = Y + (torch.rand(Y.shape, device=dev, generator=g) - 0.5) * 0.8 * spacing
    fx, fy = torch.remainder(X, L), torch.remainder(Y, L)
    keep = ~((fx > synth.BOX_LO) & (fx < synth.BOX_HI) & (fy > synth.BOX_LO) & (fy < synth.BOX_HI))
    X, Y = X[keep], Y[keep]
    parts = [torch.stack([X, Y, synth.GROUND_Z + torch.randn(X.shape, device=dev, generator=g) * 0.02], dim=1)]
    us = torch.arange(synth.BOX_LO, synth.BOX_HI, spacing, device=dev, dtype=torch.float32); zs = torch.arange(synth.GROUND_Z, synth.BOX_TOP, spacing, device=dev, dtype=torch.float32)
    U = us[:, None].expand(len(us), len(zs)).reshape(-1); Z = zs[None, :].expand(len(us), len(zs)).reshape(-1)
    for i in range(int(np.floor(x0 / L)), int(np.ceil(x1 / L))):
        for j in range(int(np.floor(y0 / L)), int(np.ceil(y1 / L))):
            for axis, val in ((0, synth.BOX_LO), (0, synth.BOX_HI), (1, synth.BOX_LO), (1, synth.BOX_HI)):
                u = U + (torch.rand(U.shape, device=dev, generator=g) - 0.5) * 0.8 * spacing; z = Z + (torch.rand(U.shape, device=dev, generator=g) - 0.5) * 0.8 * spacing
                w = val + torch.randn(U.shape, device=dev, generator=g) * 0.02
                parts.append(torch.stack([i * L + w, j * L + u, z], dim=1) if axis == 0 else torch.stack([i * L + u, j * L + w, z], dim=1))
    P = torch.cat(parts, dim=0)
    P = P[(P[:, 0] >= x0) & (P[:, 0] <= x1) & (P[:, 1] >= y0) & (P[:, 1] <= y1)]
    return torch.cat([P, torch.full((P.shape[0], 1), 50.0, device=dev)], dim=1).contiguous()


def make_scans(n_scans, n_pts, cfg, cache_dir, kitti=False):
    """Synthetic Livox-shaped stream: raw scans (lidar frame, xyzI) + the VoxelGrid-downsampled clouds.  The numpy ray caster takes
    1-2 s per scan, and the CPU-baseline leg wants >= 220 scans: missing scans are generated by worker PROCESSES (immesh_amd/scan_gen.py, started
    with subprocess -- this process holds a live HIP runtime) into the cache directory."""
    from immesh_amd import scan_gen
    os.makedirs(cache_dir, exist_ok=True)
    extT = [float(v) for v in cfg.extT]
    missing = [k for k in range(n_scans) if not os.path.exists(scan_gen.scan_path(cache_dir, n_pts, kitti, k)[:-4] + "_down.npy")]
    nproc = max(1, min(64, (os.cpu_count() or 1) - 2, len(missing)))
    if nproc > 1 and len(missing) > 4:
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, "-m", "immesh_amd.scan_gen", cache_dir, str(n_pts), str(int(kitti))] + [repr(v) for v in extT] + [str(k) for k in missing[w::nproc]],
                                  env=env, cwd=ROOT, stdout=subprocess.DEVNULL) for w in range(nproc)]
        for pr in procs:
            pr.wait()
    raws, downs = [], []
    for k in range(n_scans):
        raw, down = scan_gen.generate(cache_dir, n_pts, kitti, extT, k)      # (loads what the workers delivered, generates what they did not)
        raws.append(raw)
        downs.append(down)
    return raws, downs


# algorithmic bytes of one launch of each kernel (SURVEY.md 8(d) record sizes; DESIGN.md "Kernels")
def algorithmic_bytes(kname, c, n_scans, n_raw):
    n_ds = c["_n_ds_mean"]; iters = max(1, c["n_iter"]) / n_scans
    kname = kname.split("<")[0]
    if kname == "residual_kernel":      # per EKF iteration: point 12 B + key/slot 12 B per point, 12 B per extra probe, 229 B per plane test
        launches = c["n_iter"]
        return (n_ds * 24 * launches + c["n_extra_probe"] * 12 + c["n_plane_tests"] * 229) / launches
    if kname == "residual_persistent_kernel":
        # ONE launch per scan runs every EKF iteration (the same per-iteration bytes, summed over the scan's iterations) and, as its epilogue, the map
        # update's per-point preparation (point 12 B in, Point_with_var 96 B + sort key 8 + slot 12 out) and the full scan's transform (16 B in, 16 B out)
        return (n_ds * 24 * c["n_iter"] + c["n_extra_probe"] * 12 + c["n_plane_tests"] * 229) / n_scans + n_ds * (12 + 96 + 8 + 12) + n_raw * 32
    if kname == "point_var_kernel":     # point 12 B in, Point_with_var 96 B + sort key 8 + slot 12 out
        return n_ds * (12 + 96 + 8 + 12)
    if kname == "replay_kernel":        # per scan: every point record once (96 B) + every refit re-reads its retained points (96 B each) and writes a plane (229 B)
        return n_ds * 96 + (c["n_refit_pts"] * 96 + c["n_refits"] * 229) / n_scans
    if kname == "mesh_knn_kernel":      # per scan: C20 inspected vertices x 12 B + query 12 B + 20 ids out
        return (c["c20"] * 12 + c["n_v"] * (12 + 80 + 24)) / n_scans
    if kname in ("mesh_delaunay_kernel", "mesh_delaunay64_kernel", "mesh_tri64_kernel"):  # per scan: n_u x (12 B pos + 6 x 12 B incident triangles) + T_v x 12 B
        return (c["n_u"] * 84 + c["t_v"] * 12) / n_scans
    if kname == "mesh_transform_kernel":
        return n_raw * 32
    return None


COMMITTED_STATS = "r06_full_kernel_stats.csv"   # profiles/: rocprofv3 --kernel-trace --stats of the default command on HEAD
COMMITTED_TRAFFIC = "traffic_r06.json"          # profiles/: PMC-derived HBM bytes per launch (tools/pmc_traffic.py), same command


def kernel_sources_sha():
    """fingerprint of everything that decides which kernels run on what: the kernel sources AND the host layer that picks and sequences them (c_api.cpp's
    hashed-vs-radix VoxelGrid choice, mesh_host.cpp's launch / exchange sequence change per-launch HBM traffic just as a kernel edit does).
    A PMC traffic file measured on other sources is stale by construction.  Comments and layout are not compiled: every line is hashed without comments
    and with its blanks collapsed, line ends kept (a #define must not merge with the code that follows) -- 'c3:' scheme; a file stamped by an older
    scheme simply does not match."""
    import hashlib
    import re
    hsh = hashlib.sha256()
    d = os.path.join(ROOT, "immesh_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".inc", ".hpp", ".cpp")):
            text = open(os.path.join(d, name), "r", errors="replace").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)      # block comments
            text = re.sub(r"//[^\n]*", " ", text)                   # line comments (a '//' inside a string literal goes too: the hash only has to be stable)
            lines = [" ".join(ln.split()) for ln in text.split("\n")]
            text = "\n".join(ln for ln in lines if ln)
            hsh.update(name.encode()); hsh.update(text.encode())
    return "c3:" + hsh.hexdigest()[:16]


def roofline_from_committed_profile(mesh, cnt, n_scans, n_raw, note):
    """Fallback when the live HIP-event leg could not run: the kernel with the largest total time in the committed rocprofv3 --kernel-trace --stats
    summary of this script (among those with an algorithmic-bytes model), its average launch time from there, algorithmic bytes from THIS
    run's counters.  Clearly labelled in `source`."""
    import csv
    for name in (COMMITTED_STATS, "r01_full_kernel_stats.csv"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            for r in csv.DictReader(open(path)):   # rows are sorted by total duration
                n = r["Name"].strip('"')
                n = n[5:] if n.startswith("void ") else n
                kernel = n.split("(")[0]
                if (not mesh) and kernel.startswith("mesh_"):
                    continue
                by = algorithmic_bytes(kernel, dict(cnt), n_scans, n_raw)
                if by is None or kernel.split("<")[0] in ("point_var_kernel", "replay_kernel"):   # (their rows include the map pre-build launches)
                    continue
                avg_ms = float(r["AverageNs"]) * 1e-6
                ach = by / (avg_ms * 1e-3) / 1e9
                return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                        "avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": int(by),
                        "source": f"profiles/{name} (committed rocprofv3 average) -- live HIP-event leg unavailable: " + (note or "unknown")}
        except Exception:   # noqa: BLE001
            continue
    return None


COUNTER_KEYS = ("n_iter", "n_match", "n_plane_tests", "n_extra_probe", "n_refits", "n_refit_pts", "n_app", "n_new", "c1", "v_act", "n_v", "c20", "n_u", "t_v", "t_add", "t_rem")


def measure(args, torch, D, dist, hip, rank, world, local, dev, sharded, full):
    """One measured pass of the stream: context + map, warm-up, the timed region (barrier / synchronize on both sides, max over ranks) and -- on the
    rank(s) that report, when `full` -- the instrumented legs on the SAME context, continuing the SAME stream."""
    kitti = args.config == "velodyne"
    side = float(np.sqrt(args.map_voxels / 8.8)) + 40.0     # ~8.8 root voxels per m^2 of this world (ground + walls)
    if kitti:
        cfg = capi.velodyne_config(device=local, cap_root_voxels=1 << 18, cap_scan_points=400_000, cap_vertices=1 << 24, cap_triangles=1 << 25)
    else:
        # sharded map: a rank keeps its bricks plus the one-voxel halo (~20 % at 32^3-voxel bricks) -> capacity per rank, not per job
        bv = float(1 << args.brick_log2)
        share = (max(1.5, 1.25 * ((bv + 2.0) / bv) ** 2) / world) if sharded else 1.0   # owned bricks + their one-voxel halo (surfaces: ~((B + 2) / B)^2)
        cfg = capi.avia_config(device=local, cap_root_voxels=int(args.map_voxels * 1.3 * share) + (1 << 16), cap_scan_points=args.cap_scan_points,
                               cap_vertices=1 << args.mesh_cap_log2, cap_triangles=1 << (args.mesh_cap_log2 + 1))
    if sharded:
        cfg.shard_rank, cfg.shard_world, cfg.shard_brick_log2, cfg.shard_mesh = rank, world, args.brick_log2, 1 if args.mesh else 0
        cfg.shard_scheme = args.shard_scheme
    h = capi.HotPath(hip, cfg, "immesh_")
    comm = "none"
    if sharded:
        comm = D.attach_collectives(h, hip, dist, args.backend, dev, world, bool(args.mesh))
    n_extra = (2 * args.profile_scans if full else 0) + (args.nu_scans if args.mesh else 0)
    n_pre = max(0, args.map_scans - 1) if kitti else 0      # C4: scans 1 .. map_scans-1 build the map before the warm-up
    n_total = n_pre + args.warmup + args.steps + n_extra
    n_cpu = int(min(260, 24 + args.cpu_seconds * 12)) if (full and args.cpu_seconds > 0 and rank == 0 and not args.gpu_scans) else 0   # the CPU-baseline leg replays the stream from scan 1: 20 + 200 + the all-cores sample
    n_cpu += n_pre if n_cpu else 0
    extT_np = np.array(list(cfg.extT))
    if args.gpu_scans and not kitti:
        # scans ray-cast on the GPU (harness), down-sampled by the library's own VoxelGrid BEFORE the timed region: the long steady-state leg
        raws, downs, d_raw, d_down = [], [], [], []
        for kk in range(n_total + 1):
            Rk, tk = synth.trajectory_pose(kk)
            r = livox_scan_torch(torch, dev, kk, Rk, tk, args.pts, extT_np)
            dn, _ = h.downsample(r.data_ptr(), 0.4, n=r.shape[0], stride=4, to_host=True)
            d_raw.append(r); d_down.append(torch.from_numpy(dn).to(dev)); raws.append(np.empty((r.shape[0], 0))); downs.append(dn)
    else:
        raws, downs = make_scans(max(n_total + 1 + (world - 1), n_cpu), args.pts, cfg, os.path.join(os.environ.get("TMPDIR", "/tmp"), "immesh_scan_cache"), kitti)
        cpu_raws, cpu_downs = raws, downs
    if kitti:   # the map grows from the stream itself (3 m root voxels, max_layer 4)
        R0_, t0_ = synth.trajectory_pose(0)
        h.map_build(np.ascontiguousarray(raws[0][:, :3]), capi.make_state(R=R0_, t=t0_))
        n_map = h.counters()["n_root_voxels"]
    else:
        n_map = build_big_map(h, cfg, torch, dev, args.map_voxels, side)
    idx = D.stream_of_rank(0 if sharded else rank, n_total)   # replicas: rank r replays the stream phase-shifted by r scans; sharded: one stream
    if not (args.gpu_scans and not kitti):
        raws, downs = [raws[i] for i in idx], [downs[i] for i in idx]
        d_raw = [torch.from_numpy(r).to(dev) for r in raws]
        d_down = [torch.from_numpy(d).to(dev) for d in downs]
    n_ds_mean = float(np.mean([len(d) for d in downs[1 + n_pre:1 + n_pre + args.warmup + args.steps]]))
    mesh_seed, seed_cloud = None, None
    if args.dense_mesh and args.mesh and not kitti and not sharded:
        # the mesh map pre-seeded from the survey, capped at the stream's corridor.  The cloud goes through the mesher in
        # packages of mesh_append_budget points (every point is offered: step 1), before the stream starts
        t_seed = time.time()
        P = corridor_cloud(torch, dev, n_total)
        cam0 = synth.trajectory_pose(0)[1] + np.array([0.0, 0.0, 1.0])
        pkg = int(cfg.mesh_append_budget)
        for a in range(0, P.shape[0], pkg):
            ch = P[a:a + pkg].contiguous()
            h.mesh_scan(ch.data_ptr(), cam0, frame_idx=0, n=ch.shape[0], fetch=False)
        cs = h.counters()
        mesh_seed = {"cloud_points": int(P.shape[0]), "vertices": int(cs["n_vertices"]), "triangles_live": int(cs["n_triangles_live"]), "seconds": round(time.time() - t_seed, 1)}
        log(f"[bench] mesh map pre-seeded from the corridor survey: {mesh_seed}")
        seed_cloud = P.cpu().numpy() if (full and args.cpu_seconds > 0 and rank == 0) else None   # the CPU-baseline leg seeds the oracle's mesh map with the very same cloud
        del P

    # scan 0 seeds the stream state; constant-velocity prior (Forward_without_imu) between scans
    R0, t0 = synth.trajectory_pose(idx[0])
    st = capi.make_state(R=R0, t=t0)
    st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
    NOWAIT = 0x10   # IMMESH_SCAN_NOWAIT: return once the pose is final, map growth finishes on the stream ahead of the next scan's work
    mesh_mode = (2 if (args.async_mesh and not sharded) else 1) if args.mesh else (NOWAIT if args.async_mesh else 0)   # sharded mesher: serial per scan (its collectives must not interleave)
    if mesh_mode & 3:
        # mesh map is seeded by scan 0 (the registration map is the pre-built survey); sharded: every rank takes part in the scan's all-reduces
        h.process_scan(d_down[0].data_ptr(), d_raw[0].data_ptr(), st, st, frame_idx=0, do_mesh=1, n_ds=len(downs[0]), n_raw=len(raws[0]))

    ds_state = {"pending": None}

    def run(k, state, mode=None):
        prior = capi.forward_without_imu_native(hip, state)     # constant-velocity prior (Forward_without_imu), host side of the library
        down_ptr, n_ds = d_down[k].data_ptr(), len(downs[k])
        if args.host_inputs:
            return h.process_scan(downs[k], raws[k], prior, prior, frame_idx=k, do_mesh=mesh_mode if mode is None else mode)
        if args.device_downsample:
            # pipelined: scan k's VoxelGrid was enqueued (pre-processing stream) before scan k-1 was registered and is collected here; scan k+1's goes
            # in now, beside this scan's registration
            leaf = 0.5 if kitti else 0.4
            if ds_state["pending"] != k:
                if ds_state["pending"] is not None:
                    h.downsample_end()
                h.downsample_begin(d_raw[k].data_ptr(), leaf, n=len(raws[k]), stride=4)
            t_a = time.perf_counter()
            n_ds, down_ptr = h.downsample_end()
            t_b = time.perf_counter()
            ds_state["pending"] = None
            if k + 1 < len(d_raw):
                h.downsample_begin(d_raw[k + 1].data_ptr(), leaf, n=len(raws[k + 1]), stride=4)
                ds_state["pending"] = k + 1
            ds_state["end_s"] = ds_state.get("end_s", 0.0) + (t_b - t_a); ds_state["begin_s"] = ds_state.get("begin_s", 0.0) + (time.perf_counter() - t_b)
            ds_state["calls"] = ds_state.get("calls", 0) + 1
        return h.process_scan(down_ptr, d_raw[k].data_ptr(), prior, prior, frame_idx=k, do_mesh=mesh_mode if mode is None else mode, n_ds=n_ds, n_raw=len(raws[k]))

    k = 1
    import gc
    shim_info = None
    if n_pre:
        # SURVEY 8(d) C4 "map built from first 50 scans": the stream's own first scans through the full pipeline, untimed
        t_pre = time.time()
        for _ in range(n_pre):
            st, _ = run(k, st); k += 1
        if mesh_mode == 2:
            h.mesh_wait()
        n_map = h.counters()["n_root_voxels"]
        log(f"[bench] C4 map: {n_pre + 1} scans -> {n_map} root voxels, {h.counters()['n_vertices']} mesh vertices ({time.time() - t_pre:.1f} s)")
    if args.dropin_shim:
        # ---- the same stream THROUGH THE DROP-IN: two threads as the reference runs them, host clouds, lists fetched, mirrors applied
        # the host mirror behind the shim: the REFERENCE'S OWN Triangle_manager (drop_in/_ref/libimmesh_dropin_async_refmirror.so: triangle.hpp / .cpp compiled from
        # where they lie, built where /root/reference exists and carried to the GPU box) when it is there, else the hash-map stand-in of drop_in/stubs
        ref_so = os.path.join(ROOT, "drop_in", "_ref", "libimmesh_dropin_async_refmirror.so")
        use_ref = args.dropin_mirror == "ref" or (args.dropin_mirror == "auto" and os.path.exists(ref_so))
        if use_ref and not os.path.exists(ref_so):
            raise SystemExit("--dropin-mirror ref: drop_in/_ref/libimmesh_dropin_async_refmirror.so is not built (make -C drop_in refmirror, needs /root/reference)")
        if not use_ref:
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "drop_in"), "libimmesh_dropin_async.so"], stdout=subprocess.DEVNULL)
        sl = ctypes.CDLL(ref_so if use_ref else os.path.join(ROOT, "drop_in", "libimmesh_dropin_async.so"))
        sl.dropin_create.restype = ctypes.c_void_p; sl.dropin_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        sl.dropin_destroy.argtypes = [ctypes.c_void_p, ctypes.c_int]; sl.dropin_seed_mirror.argtypes = [ctypes.c_void_p]
        sl.dropin_mirror_sizes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        sl.dropin_run_stream.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_double] * 3 + [ctypes.c_void_p] * 3
        drv = sl.dropin_create(h.ctx, extT_np.ctypes.data_as(ctypes.c_void_p), 0)
        if not drv or sl.dropin_seed_mirror(drv) != 0:
            raise SystemExit("drop-in driver: create / mirror seeding failed")

        def shim_stream(k0, n, state):
            rr = [np.ascontiguousarray(raws[i], np.float32) for i in range(k0, k0 + n)]; dd = [np.ascontiguousarray(downs[i], np.float32) for i in range(k0, k0 + n)]
            pr = (ctypes.c_void_p * n)(*[a.ctypes.data for a in rr]); pd = (ctypes.c_void_p * n)(*[a.ctypes.data for a in dd])
            nr = np.array([len(a) for a in rr], np.int32); nd = np.array([len(a) for a in dd], np.int32)
            so = np.array(state, np.float64); ms = np.zeros(2)
            rc = sl.dropin_run_stream(drv, n, pr, nr.ctypes.data_as(ctypes.c_void_p), pd, nd.ctypes.data_as(ctypes.c_void_p), so.ctypes.data_as(ctypes.c_void_p), 0.1, 0.3, 0.5, None, None,
                                      ms.ctypes.data_as(ctypes.c_void_p))
            if rc != 0:
                raise SystemExit(f"dropin_run_stream failed ({rc})")
            return so, ms
        st, _ = shim_stream(k, args.warmup, st); k += args.warmup
        stage5 = np.zeros(5); sl.dropin_stage_ms.argtypes = [ctypes.c_void_p, ctypes.c_void_p]; sl.dropin_stage_ms(drv, stage5.ctypes.data_as(ctypes.c_void_p))
        h.counters(reset=True)
        torch.cuda.synchronize()
        gc.collect(); gc.disable()
        D.barrier()
        st, ms = shim_stream(k, args.steps, st); k += args.steps
        gc.enable()
        sl.dropin_stage_ms(drv, stage5.ctypes.data_as(ctypes.c_void_p))
        nvm, nlm = ctypes.c_int64(0), ctypes.c_int64(0)
        sl.dropin_mirror_sizes(drv, ctypes.byref(nvm), ctypes.byref(nlm))
        sl.dropin_destroy(drv, 0)
        cm = h.counters()
        shim_info = {"host_mirror": ("the reference's own Triangle_manager (src/meshing/r3live/triangle.hpp / triangle.cpp compiled from where they lie: per-vertex adjacency sets, region buckets, a mutex per operation)"
                                     if use_ref else "stand-in of drop_in/stubs (a hash map + a hash set)"),
                     "threads": "scan thread (immesh_process_scan per scan) / service thread (wait for the frame's job, fetch its lists, enqueue) / mirror thread (apply the lists; host queue 256 frames deep)",
                     "scans_per_s_until_last_pose": round(1e3 * args.steps / float(ms[0]), 1), "scans_per_s_until_mirrors_current": round(1e3 * args.steps / float(ms[1]), 1),
                     "ms_until_last_pose": round(float(ms[0]), 3), "ms_until_mirrors_current": round(float(ms[1]), 3), "mirror_vertices": int(nvm.value), "mirror_live_triangles": int(nlm.value),
                     "host_ms_per_scan": {"scan_thread_before_the_call (round 5: packing the pcl clouds; round 6: consumed in place by immesh_process_scan_strided)": round(stage5[0] / args.steps, 4), "scan_thread_immesh_process_scan": round(stage5[1] / args.steps, 4),
                                          "service_thread_wait_for_job": round(stage5[2] / args.steps, 4), "service_thread_fetch": round(stage5[3] / args.steps, 4),
                                          "mirror_thread_update": round(stage5[4] / args.steps, 4)},
                     "mirror_equals_device": bool(nvm.value == cm["n_vertices"] and nlm.value == cm["n_triangles_live"])}
        log(f"[bench] through the drop-in shim: {shim_info}")
        D.barrier()
        elapsed = D.max_over_ranks(float(ms[1]) * 1e-3, dev)
        t_marks = []
    else:
        # (before the warm-up, not between it and the timed region: a generation-2 collection of the harness's objects inside the loop is a 30 ms stall
        #  -- seen: one call of 32 ms among 50 of 0.25 ms --, and tens of milliseconds of idle device in front of the first timed scan cost that scan
        #  0.2 ms of clock ramp: its call took 0.36 ms against 0.17 for the others)
        gc.collect(); gc.disable()
        for _ in range(args.warmup):
            st, _ = run(k, st); k += 1
        h.counters(reset=True)
        torch.cuda.synchronize()
        for key in ("end_s", "begin_s", "calls"):
            ds_state.pop(key, None)
        D.barrier()
        t_begin = time.perf_counter()
        t_marks = [t_begin]
        for _ in range(args.steps):
            st, info = run(k, st); k += 1
            t_marks.append(time.perf_counter())
        t_d0 = time.perf_counter()
        if mesh_mode == 2:
            h.mesh_wait()          # drain the mesher: every scan of the timed region is fully meshed before the clock stops
        t_d1 = time.perf_counter()
        h.last_timing()            # waits for the last scan's map update (the library's own stream)
        t_d2 = time.perf_counter()
        torch.cuda.synchronize()
        gc.enable()
        log(f"[bench] drain after the last scan: mesher {1e3 * (t_d1 - t_d0):.2f} ms, map update {1e3 * (t_d2 - t_d1):.2f} ms, device {1e3 * (time.perf_counter() - t_d2):.2f} ms; "
            f"slowest calls of the timed loop (ms): {np.round(np.sort(np.diff(t_marks))[-3:] * 1e3, 2).tolist()} at scans {np.argsort(np.diff(t_marks))[-3:].tolist()}")
        if ds_state.get("calls"):
            log(f"[bench] VoxelGrid on the scan thread, ms per scan: collect (immesh_downsample_end) {1e3 * ds_state['end_s'] / ds_state['calls']:.4f}, "
                f"enqueue of the next (immesh_downsample_begin) {1e3 * ds_state['begin_s'] / ds_state['calls']:.4f}")
        D.barrier()
        elapsed = D.max_over_ranks(time.perf_counter() - t_begin, dev)
    cnt = h.counters()
    nu_hist = None
    if (mesh_mode & 3) and args.nu_scans > 0 and not sharded:
        # neighbourhood sizes of the scans right after the timed region (serial): which triangulation kernel takes which share of the voxels
        sizes = []
        for _ in range(args.nu_scans):
            if k >= len(d_down):
                break
            st, _ = run(k, st, mode=1); k += 1
            sizes.append(h.mesh_neighbourhood_sizes())
        if sizes:
            a_ = np.concatenate(sizes)
            edges = [0, 16, 32, 48, 64, 96, 128, 256, 1 << 30]
            nu_hist = {"scans": len(sizes), "voxels": int(len(a_)), "mean": round(float(a_.mean()), 1) if len(a_) else 0.0, "max": int(a_.max()) if len(a_) else 0,
                       "histogram": {f"{edges[i] + 1}-{edges[i + 1] if edges[i + 1] < (1 << 30) else 'inf'}": int(((a_ > edges[i]) & (a_ <= edges[i + 1])).sum()) for i in range(len(edges) - 1)},
                       "share_delaunay64_kernel": round(float((a_ <= 64).mean()), 4) if len(a_) else None,
                       "share_general_kernel_65_256": round(float(((a_ > 64) & (a_ <= 256)).mean()), 4) if len(a_) else None,
                       "share_big_path_above_256": round(float((a_ > 256).mean()), 4) if len(a_) else None}
    res = {"elapsed": elapsed, "shim": shim_info, "cnt": cnt, "n_ds_mean": n_ds_mean, "nu_hist": nu_hist, "mesh_seed": mesh_seed, "n_map": int(n_map), "mesh_mode": mesh_mode, "n_raw": int(np.mean([len(r) for r in raws])), "comm": comm,
           "pose_err": float(np.linalg.norm(st[9:12] - synth.trajectory_pose(idx[k - 1])[1])),
           "scan_thread_ms": ({"p50": round(float(np.percentile(np.diff(t_marks) * 1e3, 50)), 4), "p95": round(float(np.percentile(np.diff(t_marks) * 1e3, 95)), 4)}
                              if len(t_marks) > 2 else None),   # host time per immesh_process_scan call (asynchronous mode: until the pose is final)
           "cpu_inputs": (cfg, cpu_raws, cpu_downs, R0, t0, seed_cloud) if not (args.gpu_scans and not kitti) else None}
    if sharded:
        res["shard_traffic"] = h.shard_traffic()
        if rank == 0 and not kitti:
            # who does how much: the share of every scan's down-sampled points (the matcher's and the map update's work) per rank, for 8^3 / 16^3 / 32^3
            # voxel bricks -- the job runs at the pace of the rank with the LARGEST share (rank 0's share says nothing)
            extR_np = np.array(list(cfg.extR)).reshape(3, 3)
            clouds = []
            for kk in range(1 + args.warmup, 1 + args.warmup + args.steps):
                Rk, tk = synth.trajectory_pose(idx[kk])
                clouds.append((downs[kk][:, :3].astype(np.float64) @ extR_np.T + extT_np) @ Rk.T + tk)
            res["load_balance"] = D.load_balance(clouds, cfg.voxel_size, world, scheme=args.shard_scheme)

    # ---- instrumented legs (roofline): the SAME context continues the SAME stream -- same map, same mesh map, scans right after the timed
    # ones.  (a) serial per-stage times, profiler off; (b) HIP events around every launch on the library's own streams.  Run by main() under a
    # watchdog: the headline number is complete at this point and must not be lost if they hang.
    def legs():
        nonlocal st, k
        stage_, kstats_, pc_out, note = None, None, None, None
        try:
            if full and args.profile_scans > 0 and (rank == 0 or sharded):
                serial = 1 if (mesh_mode & 3) else 0
                pstage = np.zeros(4)
                for _ in range(args.profile_scans):
                    st, _ = run(k, st, mode=serial); k += 1
                    tm = h.last_timing()
                    pstage += [tm["total"], tm["register"], tm["map_update"], tm["mesh"]]
                stage_ = list(map(float, pstage / args.profile_scans))
                if not sharded or args.profile_inproc:
                    h.counters(reset=True)
                    h.profile_enable(True)
                    k0 = k
                    for _ in range(args.profile_scans):
                        st, _ = run(k, st, mode=serial); k += 1
                    kstats_ = h.profile_read()
                    h.profile_enable(False)
                    pc_ = h.counters(); pc_["_n_ds_mean"] = float(np.mean([len(d) for d in downs[k0:k]]))
                    pc_out = {kk_: float(v) for kk_, v in pc_.items()}
        except Exception as e:   # noqa: BLE001
            note = f"instrumented leg failed: {str(e)[:200]}"
        h.close()
        return stage_, kstats_, pc_out, note
    res["legs"] = legs
    return res


def cpu_baseline_leg(args, hip_cfg_inputs, budget_s):
    """The oracle (CPU restatement, kind "port") on the host cores: a bounded sample of the same stream.  Two variants per SURVEY 8(d): (i) the
    reference's own threading (12-thread pool over mesh voxels, maximum_thread_for_rec_mesh; 4 OpenMP threads in the matcher, MP_PROC_NUM) and
    (ii) every parallelisable loop on all cores; per-stage p50 / p95 after warm-up.  Results of the variants are identical."""
    cfg, raws, downs, R0, t0, seed_cloud = hip_cfg_inputs
    orc_so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(orc_so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    orc_lib = ctypes.CDLL(orc_so)
    ncores = os.cpu_count() or 1

    cpu_map_voxels = int(min(args.map_voxels, args.cpu_map_voxels))
    o = capi.HotPath(orc_lib, cfg, "orc_")
    o.set_threads(ncores, ncores)
    so = capi.make_state(R=R0, t=t0)
    if args.config == "velodyne" or cpu_map_voxels <= 0:
        o.map_build(np.ascontiguousarray(raws[0][:, :3]), so)   # kitti: exactly the GPU leg's map
    else:
        # the corridor of the GPU leg's survey around the trajectory (the same strips, generated on the host): >= cpu_map_voxels root voxels
        # the SAME survey the GPU leg ingested (same generator, same device, same seed), strip by strip, until the map holds cpu_map_voxels root voxels
        import torch
        side = float(np.sqrt(cpu_map_voxels / 8.8)) + 40.0
        ident = capi.make_state()
        cap = int(cfg.cap_scan_points)
        for P in survey_strips(cfg, torch, torch.device("cuda", int(cfg.device)) if torch.cuda.is_available() else torch.device("cpu"), side):
            Pn = P.cpu().numpy()
            for a in range(0, len(Pn), cap):
                o.map_update(np.ascontiguousarray(Pn[a:a + cap]), ident)
            if o.counters()["n_root_voxels"] >= cpu_map_voxels:
                break
    n_map = int(o.counters()["n_root_voxels"])
    log(f"[bench] CPU baseline: the oracle's map holds {n_map} root voxels")
    so[12:15] = [1.0, 0, 0]; so[15:18] = [0, 0, np.deg2rad(2.0)]
    if args.mesh and seed_cloud is not None:   # the same pre-seeded mesh map as the GPU leg: the corridor cloud in packages of mesh_append_budget points
        o.set_threads(ncores if ncores < 16 else ncores // 2, min(4, ncores))
        cam0 = synth.trajectory_pose(0)[1] + np.array([0.0, 0.0, 1.0])
        pkg = int(cfg.mesh_append_budget)
        for a in range(0, len(seed_cloud), pkg):
            o.mesh_scan(np.ascontiguousarray(seed_cloud[a:a + pkg]), cam0, frame_idx=0, fetch=False)
    if args.mesh:
        o.process_scan(downs[0], raws[0], so, so, frame_idx=0, do_mesh=True)
    kk0 = 1
    if args.config == "velodyne" and args.map_scans > 1:   # the map (registration + mesh) from the first map_scans scans, as on the GPU leg
        o.set_threads(ncores if ncores < 16 else ncores // 2, min(4, ncores))
        for kk0 in range(1, min(args.map_scans, len(raws) - 8)):
            prior = synth.forward_without_imu(so)
            so, _ = o.process_scan(downs[kk0], raws[kk0], prior, prior, frame_idx=kk0, do_mesh=bool(args.mesh))
        kk0 += 1
        n_map = int(o.counters()["n_root_voxels"])
        log(f"[bench] CPU baseline: C4 map from the first {kk0} scans: {n_map} root voxels")
    cursor = {"kk": kk0, "so": so}

    def cpu_pass(mesher_threads, matcher_threads, budget, warm, max_scans):
        """one variant of the threading on the next scans of the stream (the context, its map and its mesh map carry on: one map build for both)"""
        o.set_threads(mesher_threads, matcher_threads)
        so, kk = cursor["so"], cursor["kk"]
        tc, n_done, per, stages = 0.0, 0, [], []
        while (tc < budget or len(per) < 3) and kk < len(raws) and n_done < max_scans:
            prior = synth.forward_without_imu(so)
            a = time.perf_counter()
            so, _ = o.process_scan(downs[kk], raws[kk], prior, prior, frame_idx=kk, do_mesh=bool(args.mesh))
            dt = time.perf_counter() - a
            if n_done >= warm:
                tc += dt; per.append(dt)
                tm = o.last_timing(); stages.append([tm["register"], tm["map_update"], tm["mesh"]])
            kk += 1; n_done += 1
        cursor["so"], cursor["kk"] = so, kk
        per = np.array(per) * 1e3; stages = np.array(stages)
        if len(per) == 0:
            raise RuntimeError("CPU baseline: the stream is too short for the two threading variants")
        return {"scans": int(len(per)), "scans_per_s": round(len(per) / tc, 4), "ms_p50": round(float(np.percentile(per, 50)), 3), "ms_p95": round(float(np.percentile(per, 95)), 3),
                "stages_ms_p50": {"register": round(float(np.percentile(stages[:, 0], 50)), 3), "map_update": round(float(np.percentile(stages[:, 1], 50)), 3),
                                  "mesh": round(float(np.percentile(stages[:, 2], 50)), 3)},
                "threads": {"mesher": mesher_threads, "matcher": matcher_threads}, "map_root_voxels": n_map, "warmup_scans": warm}

    # reference threading = 200 scans after 20 warm-up scans (or what the CPU budget allows); every-loop-parallel = the physical cores
    # (on 2 x SMT the logical count only adds contention to these short loops), a shorter sample
    warm = 2 if budget_s < 5 else 20
    ref_thr = (min(12, ncores), min(4, ncores))
    phys = max(1, ncores // 2) if ncores >= 16 else ncores
    avail = len(raws) - kk0                    # the stream (what the map build left of it) is shared by the two variants
    n_all = min(24, max(4, avail // 8)) if ncores > 1 else 0
    v_ref = cpu_pass(ref_thr[0], ref_thr[1], budget_s * 0.75, min(warm, max(0, avail - n_all - 3)), min(avail - n_all, warm + 200))
    v_all = cpu_pass(phys, phys, budget_s * 0.25, 1, n_all) if ncores > 1 else v_ref
    o.close()
    best, cores = (v_all, ncores) if v_all["scans_per_s"] > v_ref["scans_per_s"] else (v_ref, ref_thr[0])
    return {"value": best["scans_per_s"], "unit": "scans/s", "cores": cores, "kind": "port",
            "sample": f"{best['scans']} scans of the same stream (after {best['warmup_scans']} warm-up scans) through oracle/liboracle.so against a map of {best['map_root_voxels']} root voxels built from the "
                      "same survey strips as the GPU leg's map (--cpu-map-voxels; default = the metric's 10 M); the same pre-seeded mesh map; "
                      "the faster of the two threading variants is `value` (reference threading: 12-thread mesher pool + 4 matcher threads; all-cores: every parallel loop on the physical cores, per-thread counters)",
            "ms_per_scan": best["ms_p50"], "reference_threading": v_ref, "all_cores": v_all, "host_cores": ncores}


def build_roofline(res, args):
    kstats, pc = res["kstats"], res["pc"]
    best = None
    for name, s_ in kstats.items():
        if s_["launches"] and algorithmic_bytes(name, pc, args.profile_scans, args.pts if args.mesh else 0) is not None:
            if best is None or s_["total_ms"] > kstats[best]["total_ms"]:
                best = name
    if not best:
        return None
    # The dominant kernel of the METRIC is the largest one on the chain that paces the pipeline.  Since round 6 that is measured, not assumed (DESIGN 4.2:
    # the scan thread's wait for the mesher): the mesher runs beside the pose chain with jobs to spare, the period is the pose chain's, and its largest
    # launch is the registration.  The mesher's largest launch (one wavefront per voxel; longer than the registration by a few microseconds since the
    # registration lost its hash probe) is reported next to it as `largest_launch_off_the_pacing_chain`.
    off_chain = None
    pose = [n_ for n_ in kstats if n_.split("<")[0] in ("residual_persistent_kernel", "residual_kernel") and kstats[n_]["launches"]]
    if pose and best.split("<")[0].startswith("mesh_"):
        ob = algorithmic_bytes(best, pc, args.profile_scans, args.pts if args.mesh else 0) / max(1.0, kstats[best]["launches"] / args.profile_scans)
        oms = kstats[best]["total_ms"] / kstats[best]["launches"]
        off_chain = {"kernel": best, "avg_launch_ms": round(oms, 5), "algorithmic_bytes_per_launch": int(ob), "achieved": round(ob / (oms * 1e-3) / 1e9, 3), "unit": "GB/s",
                     "frac": round(ob / (oms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}
        best = max(pose, key=lambda n_: kstats[n_]["total_ms"])
    per_scan_launches = kstats[best]["launches"] / args.profile_scans
    by = algorithmic_bytes(best, pc, args.profile_scans, args.pts if args.mesh else 0)
    if best.split("<")[0] not in ("residual_kernel", "residual_persistent_kernel"):
        by = by / max(1.0, per_scan_launches)
    avg_ms = kstats[best]["total_ms"] / kstats[best]["launches"]
    ach = by / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": best, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None, "avg_launch_ms": round(avg_ms, 5),
            "algorithmic_bytes_per_launch": int(by), "launches": int(kstats[best]["launches"]),
            "counters_of_the_profiled_scans": {kk_: round(pc[kk_] / args.profile_scans, 1) for kk_ in COUNTER_KEYS if kk_ in pc},
            "source": f"HIP events, live: {args.profile_scans} scans of the same stream on the same context, right after the timed region (serial per scan, events on the library's own streams)",
            **({"largest_launch_off_the_pacing_chain": off_chain} if off_chain else {})}


def dry_run_leg(args, torch, hip, dev, local):
    """configs[4] on ONE GPU: rank r of W of the sharded job, alone.  Its context is configured exactly as in the W-rank job (brick ownership + halo, per-rank
    capacities, sharded mesher) and is fed everything the job would feed it -- the whole survey of the 50 M-voxel square, every 500 000-pt scan -- with
    the data-path collectives stubbed (immesh_stub_collectives).  Reported: the root voxels / HBM bytes of the share and the rank's time per scan: the
    W-rank rate is bounded by the slowest rank's time plus the (latency-bound, < 0.1 ms) collectives -- a prediction until a node exists."""
    W, r = args.dry_run_world, args.dry_run_rank
    bv = float(1 << args.brick_log2)
    share = max(1.5, 1.25 * ((bv + 2.0) / bv) ** 2) / W
    mk = lambda rr_: capi.avia_config(device=local, cap_root_voxels=int(args.map_voxels * 1.3 * share) + (1 << 16), cap_scan_points=2_500_000, cap_vertices=1 << 24, cap_triangles=1 << 25,
                                      shard_rank=max(rr_, 0), shard_world=W, shard_brick_log2=args.brick_log2, shard_mesh=1 if args.mesh else 0)
    extT = np.array(list(mk(0).extT)); extR = np.array(list(mk(0).extR)).reshape(3, 3)
    n_total = args.warmup + args.steps
    # the scan stream first (down-sampled by a plain context): the per-rank shares of its points decide WHICH rank is the slowest one of the job
    h0 = capi.HotPath(hip, capi.avia_config(device=local, cap_root_voxels=1 << 12, cap_scan_points=2_500_000, cap_vertices=1 << 12, cap_triangles=1 << 12), "immesh_")
    d_raw, downs_h, n_ds, clouds = [], [], [], []
    for kk in range(n_total + 1):
        Rk, tk = synth.trajectory_pose(kk)
        rr = livox_scan_torch(torch, dev, kk, Rk, tk, args.pts, extT)
        dn, _ = h0.downsample(rr.data_ptr(), 0.4, n=rr.shape[0], stride=4, to_host=True)
        d_raw.append(rr); downs_h.append(dn); n_ds.append(len(dn))
        if kk > args.warmup:
            clouds.append((dn[:, :3].astype(np.float64) @ extR.T + extT) @ Rk.T + tk)
    h0.close()
    balance = {"lattice colouring (bx + 3 by + 5 bz) mod W [scheme 0, the default]": D.load_balance(clouds, mk(0).voxel_size, W, scheme=0),
               "hash(brick) mod W [scheme 1, rounds 1-4]": D.load_balance(clouds, mk(0).voxel_size, W, scheme=1)}
    bal_now = balance[[k_ for k_ in balance if f"scheme {args.shard_scheme}" in k_][0]]
    ranks = list(range(W)) if r < 0 else [r]     # --dry-run-rank -2 (default): EVERY rank's share in turn; the job waits for the slowest one
    side = float(np.sqrt(args.map_voxels / 8.8)) + 40.0
    d_down = [torch.from_numpy(dn).to(dev) for dn in downs_h]
    per_rank = []
    by_calls = {}     # exchanges a scan needed (>= 2 admission rounds + the two bands) -> that scan's time, over all ranks
    for r in ranks:
        cfg = mk(r)
        cfg.shard_scheme = args.shard_scheme
        h = capi.HotPath(hip, cfg, "immesh_")
        h.stub_collectives()
        t0 = time.time()
        n_map = build_big_map(h, cfg, torch, dev, args.map_voxels, side)      # (a rank keeps its bricks + halo: the loop runs through every strip of the square)
        t_map = time.time() - t0
        R0, t0_ = synth.trajectory_pose(0)
        st = capi.make_state(R=R0, t=t0_)
        st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
        mode = 1 if args.mesh else 0          # the sharded mesher runs serial per scan (its exchanges must not interleave with the next scan's all-reduces)
        if mode:
            h.process_scan(d_down[0].data_ptr(), d_raw[0].data_ptr(), st, st, frame_idx=0, do_mesh=1, n_ds=n_ds[0], n_raw=d_raw[0].shape[0])
        k = 1
        for _ in range(args.warmup):
            prior = capi.forward_without_imu_native(hip, st)
            st, _ = h.process_scan(d_down[k].data_ptr(), d_raw[k].data_ptr(), prior, prior, frame_idx=k, do_mesh=mode, n_ds=n_ds[k], n_raw=d_raw[k].shape[0]); k += 1
        h.counters(reset=True)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        marks = [tb]
        xc = [h.shard_traffic()["calls"]]
        for _ in range(args.steps):
            prior = capi.forward_without_imu_native(hip, st)
            st, info = h.process_scan(d_down[k].data_ptr(), d_raw[k].data_ptr(), prior, prior, frame_idx=k, do_mesh=mode, n_ds=n_ds[k], n_raw=d_raw[k].shape[0]); k += 1
            h.last_timing()                 # (serial per scan in this leg: the scan's map update has finished)
            marks.append(time.perf_counter())
            xc.append(h.shard_traffic()["calls"])   # (a host counter: which scans needed a third admission round is what makes the scan times bimodal)
        torch.cuda.synchronize()
        el = time.perf_counter() - tb
        cnt = h.counters()
        per = np.diff(marks) * 1e3
        for ms_, nx_ in zip(per, np.diff(xc)):
            by_calls.setdefault(int(nx_), []).append(float(ms_))
        # the median scan, not the mean: a rank's ten scans follow the allocation of its 90 GB context, and one stalled call (a 6 ms hiccup seen on one rank
        # of one run) would otherwise decide which rank "the job waits for"
        per_rank.append({"rank": r, "ms_per_scan": round(float(np.median(per)), 4), "ms_per_scan_mean": round(1e3 * el / args.steps, 4), "ms_per_scan_max": round(float(per.max()), 4), "root_voxels_kept": int(n_map), "device_bytes_allocated": int(h.device_bytes()), "map_build_seconds": round(t_map, 1),
                         "matches_per_scan": round(cnt["n_match"] / max(1, cnt["n_iter"]), 1), "new_vertices_per_scan": round(cnt["n_new"] / args.steps, 1),
                         "pose_err_m": float(np.linalg.norm(st[9:12] - synth.trajectory_pose(k - 1)[1]))})
        log(f"[bench] dry run, rank {r} of {W}: {per_rank[-1]}")
        h.close()
        del h
        import gc as _gc
        _gc.collect()
    # the SAME scans through ONE unsharded GPU (VERDICT r05 next #3d): what a rank of the sharded job has to beat.  The metric's 10 M-voxel map (50 M root
    # voxels do not fit one GPU: that is what the job is sharded for), serial per scan exactly like the ranks above, then asynchronous as the headline runs
    single = None
    if len(ranks) > 1 and args.mesh:
        try:
            cfg1 = capi.avia_config(device=local, cap_root_voxels=int(10e6 * 1.3) + (1 << 16), cap_scan_points=2_500_000, cap_vertices=1 << 24, cap_triangles=1 << 25)
            h = capi.HotPath(hip, cfg1, "immesh_")
            n_map1 = build_big_map(h, cfg1, torch, dev, 10e6, float(np.sqrt(10e6 / 8.8)) + 40.0)
            single = {"map_root_voxels": int(n_map1)}
            for mode_, label_ in ((1, "ms_per_scan_serial"), (2, "ms_per_scan_async")):
                R0, t0_ = synth.trajectory_pose(0)
                st = capi.make_state(R=R0, t=t0_)
                st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
                k = 1
                for _ in range(args.warmup):
                    prior = capi.forward_without_imu_native(hip, st)
                    st, _ = h.process_scan(d_down[k].data_ptr(), d_raw[k].data_ptr(), prior, prior, frame_idx=k, do_mesh=mode_, n_ds=n_ds[k], n_raw=d_raw[k].shape[0]); k += 1
                h.mesh_wait(); h.counters(reset=True)
                tb = time.perf_counter(); marks = [tb]
                for _ in range(args.steps):
                    prior = capi.forward_without_imu_native(hip, st)
                    st, _ = h.process_scan(d_down[k].data_ptr(), d_raw[k].data_ptr(), prior, prior, frame_idx=k, do_mesh=mode_, n_ds=n_ds[k], n_raw=d_raw[k].shape[0]); k += 1
                    if mode_ == 1:
                        h.last_timing()
                    marks.append(time.perf_counter())
                h.mesh_wait(); h.counters()
                single[label_] = round(float(np.median(np.diff(marks))) * 1e3, 4) if mode_ == 1 else round(1e3 * (time.perf_counter() - tb) / args.steps, 4)
            h.close()
            del h
        except Exception as e:   # noqa: BLE001
            single = {"error": str(e)[:160]}
    slow = max(per_rank, key=lambda q_: q_["ms_per_scan"])
    el_ms = slow["ms_per_scan"]
    who = f"rank {slow['rank']} of {W}, the slowest of {'all ' + str(W) + ' ranks run in turn' if len(ranks) > 1 else 'the one rank run'}"
    out = {"metric": f"scans/sec bound of the {W}-rank job = the slowest rank's share ({who}; collectives stubbed), 500k-pt scan into 50M-voxel map" if args.pts >= 400000 else f"scans/sec of {who} alone (collectives stubbed)",
           "value": round(1e3 / el_ms, 4), "unit": "scans/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el_ms,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"every rank's share of the {W}-rank sharded job in turn on one GPU, collectives stubbed: {args.pts}-pt scans, {int(args.map_voxels)}-root-voxel survey",
                      "n_raw": int(d_raw[1].shape[0]), "n_ds_mean": round(float(np.mean(n_ds[1:])), 1), "map_root_voxels": int(slow["root_voxels_kept"]), "params": "config/avia.yaml",
                      "parallelism": f"dry run: {int(bv)}^3-voxel bricks (ownership scheme {args.shard_scheme}) + 1-voxel halo, sharded mesher; all-reduce / all-gather replaced by local no-ops"},
           "load_balance_point_share_per_brick_size": balance,
           "share": {"per_rank": per_rank, "slowest_rank": slow["rank"], "fastest_ms_per_scan": min(q_["ms_per_scan"] for q_ in per_rank),
                     "root_voxels_kept_by_this_rank": int(slow["root_voxels_kept"]), "of_total_surveyed": int(args.map_voxels), "device_bytes_allocated": int(slow["device_bytes_allocated"]),
                     "busiest_point_share_mean": bal_now[int(bv)]["max_share_mean"], "fair_share": round(1.0 / W, 4),
                     "ms_per_scan_by_exchanges_of_the_scan": {str(k_): {"scans": len(v_), "median_ms": round(float(np.median(v_)), 4)} for k_, v_ in sorted(by_calls.items())},
                     "one_unsharded_gpu_on_the_same_scans": single},
           "pose_err_m": slow["pose_err_m"]}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pts", type=int, default=100000, help="raw points per scan")
    ap.add_argument("--map-voxels", type=float, default=10e6, help="root voxels of the pre-built registration map")
    ap.add_argument("--mesh", type=int, default=1, help="1 = full pipeline (configs[2]); 0 = registration + map update only (configs[1])")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="CPU budget of the oracle baseline leg (0 = skip)")
    ap.add_argument("--profile-scans", type=int, default=5)
    ap.add_argument("--config", choices=["avia", "velodyne"], default="avia", help="avia = BASELINE configs[1]/[2] (the metric's workload); velodyne = configs[3], KITTI-shaped HDL-64 scans with velodyne.yaml parameters")
    ap.add_argument("--shard", type=int, default=0, help="N>1 only. 0 = headline is N replicas (every rank its own stream + map, weak scaling) with the sharded split as a second leg; "
                    "1 = only the sharded split: ONE stream, registration map and mesher sharded by voxel bricks over the ranks (strong scaling; the capacity mode of configs[4])")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo only for single-GPU functional tests)")
    ap.add_argument("--device-downsample", type=int, default=0, help="1 = the VoxelGrid down-sampling of every raw scan also runs on the device inside the timed region (SURVEY 8(f) rank 1)")
    ap.add_argument("--cpu-map-voxels", type=float, default=10.0e6, help="root voxels of the map the CPU-baseline leg builds for the oracle from the same survey strips (default: the metric's 10 M, ~40 s of host time; 0 = scan 0 only)")
    ap.add_argument("--host-inputs", type=int, default=0, help="1 = every scan is handed over as HOST buffers (the library stages them over PCIe inside the timed region): the PCIe-inclusive rate")
    ap.add_argument("--dropin-shim", type=int, default=0, help="1 = the timed region runs THROUGH THE DROP-IN (drop_in/immesh_shim_async.cpp behind drop_in/dropin_driver.cpp): pcl-shaped host clouds in, "
                    "one immesh_process_scan(ASYNC) per scan on the scan thread, every frame's result lists fetched and applied to the Global_map / Triangle_manager mirrors by the service thread")
    ap.add_argument("--async-mesh", type=int, default=1, help="1 = meshing of scan k overlaps registration of scan k+1 (the reference's mesh service thread); 0 = strictly serial per scan")
    ap.add_argument("--gpu-scans", type=int, default=0, help="1 = the scan stream is ray-cast on the GPU by the harness (torch) and down-sampled by the library before the timed region: "
                    "hundreds of scans in seconds (the steady-state leg); the CPU-baseline leg needs the default host-generated stream")
    ap.add_argument("--dense-mesh", type=int, default=1, help="1 (default, SURVEY 8(d) C3: \"mesh map pre-seeded from the same survey, capped at the corridor\") = the MESH map is pre-seeded from a dense "
                    "survey of the stream's corridor before the stream starts; 0 = seeded by scan 0 only (the stream meshes unexplored ground: the headline of rounds 1-2)")
    ap.add_argument("--nu-scans", type=int, default=5, help="scans after the timed region whose per-voxel neighbourhood sizes n_u are collected (histogram + kernel shares)")
    ap.add_argument("--dry-run-rank", type=int, default=-1, help=">= 0: run ONE rank of a --dry-run-world job alone on this GPU with stubbed collectives (configs[4] capacity / per-rank time)")
    ap.add_argument("--dry-run-world", type=int, default=8)
    ap.add_argument("--mesh-cap-log2", type=int, default=24, help="capacity of the mesh map: 2^k vertices, 2^(k+1) triangles (the hash tables are sized from it)")
    ap.add_argument("--cap-scan-points", type=int, default=2_500_000, help="largest scan the context accepts (sizes the per-scan scratch and the VoxelGrid's leaf table)")
    ap.add_argument("--brick-log2", type=int, default=3, help="sharded runs: voxel bricks of 2^k voxels per axis are the unit of ownership (registration map and mesher); 3 = 8^3 (SURVEY 8(e)): "
                    "a scan's footprint spans hundreds of bricks, so the ranks' shares of a scan stay close to 1/N; 5 = 32^3 (rounds 1-3: a handful of bricks per scan)")
    ap.add_argument("--profile-inproc", type=int, default=0, help="sharded runs only: 1 = also run the HIP-event leg (every rank takes part)")
    ap.add_argument("--profile-timeout", type=float, default=120.0, help="watchdog of the instrumented legs + extra configurations (seconds)")
    ap.add_argument("--sharded-leg", type=int, default=1, help="N>1: after the replica headline also measure the sharded split (ONE stream over N ranks) and report it as `sharded`")
    ap.add_argument("--shard-scheme", type=int, default=0, choices=[0, 1], help="brick ownership of the sharded map / mesher: 0 = lattice colouring (bx + 3 by + 5 bz) mod W (default), 1 = hash(brick) mod W (rounds 1-4)")
    ap.add_argument("--dropin-mirror", choices=["auto", "ref", "stub"], default="auto", help="--dropin-shim 1: the host mirror the shim applies the lists to: ref = the reference's own Triangle_manager "
                    "(drop_in/_ref/libimmesh_dropin_async_refmirror.so), stub = the stand-in of drop_in/stubs, auto = ref when it is built")
    ap.add_argument("--map-scans", type=int, default=50, help="--config velodyne (BASELINE configs[3] = SURVEY 8(d) C4): the map -- registration map AND mesh map -- is built from the first "
                    "this many scans of the stream (scan 0 through immesh_map_build, the rest through the full pipeline, untimed), then the stream is timed; 1 = scan 0 only (the variant of rounds 1-4)")
    ap.add_argument("--extra-configs", type=int, default=-1, help="1 = also run BASELINE configs[1] (--mesh 0) and configs[3] (--config velodyne) as short child runs and report them under "
                    "`extra`; default: on for the plain N=1 headline run")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # launched as `python bench.py --gpus N`: become the launcher of N ranks (one per GPU) of this very command line
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(sys.argv[0])] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    rank, world, local = D.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    local = local % torch.cuda.device_count()   # one GPU per rank on a real node; functional tests may stack ranks on one device (gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = D.init(args.backend, dev if args.backend == "nccl" else None) if world > 1 else None
    only_sharded = bool(args.shard) and world > 1
    kitti = args.config == "velodyne"
    hip = capi.load_hip_library()
    if args.dry_run_rank >= 0 or args.dry_run_rank == -2:
        return dry_run_leg(args, torch, hip, dev, local)

    res = measure(args, torch, D, dist, hip, rank, world, local, dev, sharded=only_sharded, full=True)
    value = D.aggregate_throughput(args.steps, 1 if only_sharded else world, res["elapsed"])
    mesh_mode, cnt = res["mesh_mode"], res["cnt"]
    cnt["_n_ds_mean"] = res["n_ds_mean"]
    NOWAIT = 0x10

    # The line as it stands after the timed region; the legs below fill it in.  A watchdog prints it -- with whatever has been filled in -- if one
    # of them hangs: the headline number must not be lost.
    out = None
    printed = threading.Event()

    def emit():
        if rank == 0 and out is not None and not printed.is_set():
            printed.set()
            print(json.dumps(out), flush=True)

    def watchdog():
        # every rank runs one: when a leg after the timed region hangs (a collective of the sharded leg with a dead peer, say), rank 0 prints the line
        # as it stands and every rank leaves -- the launcher must not be left waiting for the others
        if not printed.wait(3.0 * args.profile_timeout + args.cpu_seconds + (30.0 if rank == 0 else 40.0)):
            if out is not None:
                out["profile_leg_note"] = ((out.get("profile_leg_note") or "") + " watchdog: a leg after the timed region did not finish; the line was printed without it").strip()
            emit()
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)

    def add_traffic(rf):
        tr = os.path.join(ROOT, "profiles", COMMITTED_TRAFFIC)   # PMC-derived HBM bytes/launch from the committed rocprofv3 --pmc passes
        if rf is not None and os.path.exists(tr):
            try:
                tj = json.load(open(tr))
                meta = tj.get("_meta", {})
                if meta.get("kernel_sources_sha16") != kernel_sources_sha():
                    # refuse a file measured on other kernel sources: PMC counters cannot be read in-process, and a number from another build would be passed off as this one's
                    rf["traffic"] = None
                    rf["traffic_source"] = (f"profiles/{COMMITTED_TRAFFIC} was measured on kernel sources {meta.get('kernel_sources_sha16')} (commit {meta.get('commit')}), this build is "
                                            f"{kernel_sources_sha()}: stale, not reported (tools/refresh_profiles.sh regenerates it)")
                else:
                    rec = tj.get(rf["kernel"])
                    if rec is None:   # (rocprofv3 names template instantiations -- residual_persistent_kernel<true> --, the in-library profiler the template)
                        rec = next((v_ for k_, v_ in tj.items() if k_ != "_meta" and k_.split("<")[0] == rf["kernel"].split("<")[0]), None)
                    rf["traffic"] = rec.get("hbm_bytes_per_launch") if isinstance(rec, dict) else None   # bytes per launch, like `algorithmic_bytes_per_launch`
                    rf["traffic_unit"] = "bytes per launch: (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024, the guide's gfx950 correction"
                    rf["traffic_detail"] = rec
                    rf["traffic_source"] = (f"profiles/{COMMITTED_TRAFFIC}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on these kernel sources ({meta.get('kernel_sources_sha16')}, "
                                            f"commit {meta.get('commit')}); PMC counters cannot be read in-process, so not measured in this run")
            except Exception:   # noqa: BLE001
                pass
        return rf

    if rank == 0:
        out = {
            "metric": ("scans/sec (reg+mesh), KITTI-shaped 130k-ray scans" if kitti else "scans/sec (reg+mesh), 100k-pt scan into 10M-voxel map") if args.mesh else
                      "scans/sec (registration + map update, meshing off), 100k-pt scan into 10M-voxel map",
            "value": round(value, 4), "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * res["elapsed"] / args.steps, 4), "higher_is_better": True, "scaling": "strong" if only_sharded else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (("synthetic KITTI-shaped HDL-64 scan stream (velodyne.yaml), " if kitti else "synthetic Livox-Avia 100k-pt/scan stream, ") +
                                    ("full pipeline (registration + map update + voxel meshing)" if args.mesh else "registration + map update, meshing off")),
                       "n_raw": res["n_raw"], "n_ds_mean": round(res["n_ds_mean"], 1), "map_root_voxels": res["n_map"], "params": "config/velodyne.yaml" if kitti else "config/avia.yaml",
                       "parallelism": (f"one stream; registration map sharded over {world} GPUs in {1 << args.brick_log2}^3-voxel bricks (ownership + 1-voxel halo, all-reduce of 46 doubles per EKF iteration); mesher sharded by mesh-voxel bricks (owner-computed vertex admission + kNN + Delaunay; all-gathers of the boundary band only: band candidates / decisions, smoothed positions and triangle marks within reach of another rank's brick; every rank reports its own part of the result lists); collectives: {res['comm']}" if only_sharded
                                       else f"{world} independent scan streams, one per GPU" if world > 1 else "1 GPU"),
                       "mesh_map": ("pre-seeded from a dense survey of the stream's corridor (SURVEY 8(d) C3)" if (args.dense_mesh and args.mesh and not kitti and not only_sharded) else (f"built by the first {args.map_scans} scans of the stream, like the registration map (SURVEY 8(d) C4)" if (kitti and args.map_scans > 1) else "seeded by scan 0 only")) if args.mesh else "none",
                       "registration_map": (f"built from the first {args.map_scans} scans of the stream (SURVEY 8(d) C4)" if (kitti and args.map_scans > 1) else "scan 0 only") if kitti else "dense survey (SURVEY 8(d) C2/C3)",
                       "mesh_mode": {0: "off", NOWAIT: "off (map update of scan k overlaps the host side of scan k+1)", 1: "serial", 2: "async (mesh of scan k overlaps registration of scan k+1; vertex admission + kNN of scan k+1 overlap triangulation of scan k)"}[mesh_mode],
                       "downsample": "device, inside the timed region (asynchronous: scan k+1's VoxelGrid on the pre-processing stream beside scan k's registration)" if args.device_downsample else "host (before the timed region; the hot path starts at lio_state_estimation)",
                       "inputs": ("pcl-shaped host clouds through the drop-in shim, consumed in place by immesh_process_scan_strided (packed into pinned staging + copied over PCIe inside the timed region; result lists fetched, host mirrors applied)" if args.dropin_shim else
                                  "host buffers, staged over PCIe inside the timed region" if args.host_inputs else "resident in HBM before the timed region")},
            "stages_ms_serial": None,
            "counters_per_scan": {kk_: round(v / args.steps, 1) for kk_, v in cnt.items() if kk_ in COUNTER_KEYS},
            "scan_thread_ms": res["scan_thread_ms"], "pose_err_m": round(res["pose_err"], 4),
            "n_u": res["nu_hist"], "mesh_seed": res["mesh_seed"], "drop_in_shim": res.get("shim"),
            "roofline": None, "cpu_baseline": None, "kernels_ms_per_scan": {},
        }
        if only_sharded:
            out["collectives"] = res["comm"]
        if res.get("shard_traffic"):
            out["exchange_bytes_per_scan_rank0"] = round(res["shard_traffic"]["bytes"] / max(1, args.steps + args.warmup + 1), 1)
            out["exchange_calls_per_scan"] = round(res["shard_traffic"]["calls"] / max(1, args.steps + args.warmup + 1), 2)
        if res.get("load_balance"):
            out["load_balance_point_share_per_brick_size"] = res["load_balance"]
        if args.profile_scans > 0:   # until the live leg has delivered: the committed rocprofv3 average of the dominant kernel with this run's own counters
            out["roofline"] = add_traffic(roofline_from_committed_profile(args.mesh, cnt, args.steps, args.pts, "not run yet"))
    threading.Thread(target=watchdog, daemon=True).start()

    stage, kstats, pc, prof_note = res["legs"]()
    if rank == 0:
        if stage:
            out["stages_ms_serial"] = {"gpu_total": round(stage[0], 4), "register": round(stage[1], 4), "map_update": round(stage[2], 4), "mesh": round(stage[3], 4)}
        if kstats:
            res["kstats"], res["pc"] = kstats, pc
            out["kernels_ms_per_scan"] = {n: round(s_["total_ms"] / max(1, args.profile_scans), 4) for n, s_ in sorted(kstats.items(), key=lambda kv: -kv[1]["total_ms"])}
            live = build_roofline(res, args)
            if live is not None:
                out["roofline"] = add_traffic(live)
        elif args.profile_scans > 0:
            out["roofline"] = add_traffic(roofline_from_committed_profile(args.mesh, cnt, args.steps, args.pts, prof_note or "the HIP-event leg did not run"))
        if prof_note:
            out["profile_leg_note"] = prof_note

    # ---- N > 1: the north-star split as a second leg -- ONE stream, voxel bricks sharded over the ranks.  It runs as a CHILD job (this script with
    # --shard 1, the same N ranks) that rank 0 launches after every rank of this job has finished the replica headline and the other ranks have
    # left: a crash or a hung collective in the sharded path (RCCL issued by the library on its own streams) must not take the headline with it.
    if world > 1:
        D.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
    if world > 1 and not only_sharded and args.sharded_leg:
        env = {kk_: v for kk_, v in os.environ.items() if kk_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
                                                                       "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
                                                                       "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")}
        cmd = [sys.executable, os.path.abspath(sys.argv[0]), "--gpus", str(world), "--shard", "1", "--backend", args.backend, "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--pts", str(args.pts), "--map-voxels", str(args.map_voxels), "--mesh", str(args.mesh), "--config", args.config, "--cpu-seconds", "0", "--profile-scans", "0",
               "--extra-configs", "0", "--brick-log2", str(args.brick_log2)]
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1.5 * args.profile_timeout, text=True)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                d = json.loads(lines[-1])
                out["sharded"] = {"value": d["value"], "unit": d["unit"], "scaling": d["scaling"], "ms_per_step": d["ms_per_step"], "parallelism": d["config"]["parallelism"],
                                  "collectives": d.get("collectives"), "map_root_voxels_rank0": d["config"]["map_root_voxels"], "exchange_bytes_per_scan_rank0": d.get("exchange_bytes_per_scan_rank0"),
                                  "exchange_calls_per_scan": d.get("exchange_calls_per_scan"), "load_balance_point_share_per_brick_size": d.get("load_balance_point_share_per_brick_size"),
                                  "pose_err_m": d["pose_err_m"], "how": "child job of the same N ranks, launched by rank 0 after the replica leg"}
                # THE HEADLINE FOR N > 1 IS THE SHARDED JOB (north star: ONE scan stream, voxels sharded by spatial hash over the GPUs: strong scaling of the
                # metric's workload); the independent replicas measured above move to `replicas`.  Only if the sharded job fails does the line keep the
                # replica numbers (then it says so: "scaling": "weak" + `sharded.error`).
                out["replicas"] = {"value": out["value"], "unit": out["unit"], "scaling": "weak", "ms_per_step": out["ms_per_step"], "parallelism": out["config"]["parallelism"],
                                   "scan_thread_ms": out["scan_thread_ms"], "pose_err_m": out["pose_err_m"], "mesh_mode": out["config"]["mesh_mode"]}
                out["value"], out["ms_per_step"], out["scaling"], out["pose_err_m"] = d["value"], d["ms_per_step"], "strong", d["pose_err_m"]
                out["scan_thread_ms"] = d.get("scan_thread_ms")
                out["counters_per_scan"] = d.get("counters_per_scan", out["counters_per_scan"])
                for kk_ in ("parallelism", "mesh_mode", "mesh_map", "map_root_voxels"):
                    out["config"][kk_] = d["config"][kk_]
                out["config"]["map_root_voxels_note"] = "rank 0's share (owned bricks + halo) of the 10 M-voxel survey"
            else:
                out["sharded"] = {"error": f"rc {r.returncode}: " + (r.stderr or "")[-300:]}
        except Exception as e:   # noqa: BLE001
            out["sharded"] = {"error": str(e)[:200]}

    # ---- CPU baseline leg (rank 0, N = 1 semantics: the oracle on this box's host cores)
    if rank == 0 and args.cpu_seconds > 0 and not args.gpu_scans and world == 1:
        out["cpu_baseline"] = cpu_baseline_leg(args, res["cpu_inputs"], args.cpu_seconds)
    elif rank == 0 and world > 1:
        out["cpu_baseline_note"] = "timed at N = 1 only (the same host cores, the same oracle: nothing about it changes with the number of GPUs)"

    # ---- BASELINE configs[1] and configs[3] as short child runs of this script (N = 1 headline runs only)
    want_extra = args.extra_configs if args.extra_configs >= 0 else int(world == 1 and args.mesh == 1 and args.config == "avia" and not args.device_downsample and args.pts == 100000 and args.map_voxels >= 10e6)
    if rank == 0 and want_extra:
        extra = {}
        env = {kk_: v for kk_, v in os.environ.items() if kk_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        for label, flags in (("configs[1] registration only", ["--mesh", "0"]), ("configs[3] velodyne.yaml, KITTI-shaped, map built from the first 50 scans (SURVEY 8(d) C4)", ["--config", "velodyne", "--steps", str(min(args.steps, 20)), "--cpu-seconds", "8"]),
                             ("configs[3] variant of rounds 1-4: map from scan 0 only (every timed scan creates root voxels)", ["--config", "velodyne", "--map-scans", "1", "--steps", str(min(args.steps, 20))]),
                             ("full pipeline, VoxelGrid of the raw scan on the device inside the timed region", ["--device-downsample", "1"]),
                             ("full pipeline, scans handed over as host buffers (PCIe-inclusive)", ["--host-inputs", "1"]),
                             ("full pipeline THROUGH THE DROP-IN SHIM (what an unchanged ImMesh_node.cpp sees: pcl host clouds in, lists fetched, Triangle_manager / Global_map mirrors applied)", ["--dropin-shim", "1"]),
                             ("through the drop-in shim, host mirror = the stand-in of drop_in/stubs (round 4's leg)", ["--dropin-shim", "1", "--dropin-mirror", "stub"]),
                             ("full pipeline, mesh map seeded by scan 0 only (the stream meshes unexplored ground: the headline of rounds 1-2)", ["--dense-mesh", "0"]),
                             ("full pipeline, steady state: 500 scans after 20 warm-up scans", ["--gpu-scans", "1", "--steps", "500", "--warmup", "20", "--nu-scans", "0"]),
                             ("configs[4] dry run: every rank of 8 in turn, alone, 500k-pt scans, 50 M-voxel survey, collectives stubbed (the job waits for the slowest)", ["--dry-run-rank", "-2", "--pts", "500000", "--map-voxels", "50e6", "--steps", "10", "--warmup", "3"])):
            cmd = [sys.executable, os.path.abspath(sys.argv[0]), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--cpu-seconds", "0", "--profile-scans", "0", "--extra-configs", "0"] + flags   # (later flags win)
            try:
                # (IMMESH_DEBUG_WAITS: the library prints, when the context is destroyed, the kernel-entry times of the mesher's last jobs, the job-to-job period
                #  and how long the scan thread stood waiting for a world buffer -- who paces the pipeline; two getenv calls and one clock read per scan)
                r = subprocess.run(cmd, env=dict(env, IMMESH_DEBUG_WAITS="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.profile_timeout, text=True)
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if r.returncode == 0 and lines:
                    d = json.loads(lines[-1])
                    extra[label] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "metric": d["metric"], "n_ds_mean": d["config"]["n_ds_mean"],
                                    "map_root_voxels": d["config"]["map_root_voxels"], "scan_thread_ms": d.get("scan_thread_ms")}
                    marks = [ln for ln in (r.stderr or "").splitlines() if ln.startswith("[mesh marks]")]
                    if marks and "steady state" in label:
                        extra[label]["mesher_pipeline"] = marks[-1][len("[mesh marks] "):]
                    if d.get("cpu_baseline"):
                        cb = d["cpu_baseline"]
                        extra[label]["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "ms_per_scan": cb["ms_per_scan"],
                                                        "reference_threading": cb["reference_threading"], "sample": cb["sample"]}
                    if d.get("share"):
                        extra[label]["share"] = d["share"]
                        extra[label]["load_balance_point_share_per_brick_size"] = d.get("load_balance_point_share_per_brick_size")
                    if d.get("drop_in_shim"):
                        extra[label]["drop_in_shim"] = d["drop_in_shim"]
                    for kk_ in ("n_u", "mesh_seed", "counters_per_scan"):
                        if d.get(kk_) and (kk_ != "counters_per_scan" or "dense" in label or "MESH" in label):
                            extra[label][kk_] = d[kk_]
                else:
                    extra[label] = {"error": f"rc {r.returncode}"}
            except Exception as e:   # noqa: BLE001
                extra[label] = {"error": str(e)[:120]}
        out["extra"] = extra

    emit()


if __name__ == "__main__":
    main()
