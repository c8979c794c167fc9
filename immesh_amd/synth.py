"""Synthetic scan streams for tests and bench: a procedural world (ground plane + box
buildings on a 20 m lattice), a Livox-Avia-shaped and an HDL-64-shaped ray caster, the PCL-VoxelGrid-style
down-sampling that precedes the hot path (src/voxel_mapping.cpp:1888-1891) and the constant-velocity
prior of ImuProcess::Forward_without_imu (src/IMU_Processing.cpp:486-553).  Harness only -- both the HIP path and
the oracle consume the arrays produced here, so nothing in this file can create a parity difference.
"""
import numpy as np

GROUND_Z = -1.62
LATTICE = 20.0
BOX_LO, BOX_HI = 6.0, 14.0
BOX_TOP = 6.41


def halton(n, base, start=1):
    i = np.arange(start, start + n, dtype=np.int64)
    f = np.ones(n)
    r = np.zeros(n)
    while np.any(i > 0):
        f = f / base
        r = r + f * (i % base)
        i = i // base
    return r


def yaw_R(yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def so3_exp(v):
    th = np.linalg.norm(v)
    if th < 1e-5:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def raycast(o, d, max_range=100.0):
    """o: (3,), d: (n,3) unit.  Returns range (inf when nothing is hit)."""
    n = len(d)
    t_best = np.full(n, np.inf)
    dz = d[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (GROUND_Z - o[2]) / dz
    ok = (dz < 0) & (tg > 0)
    t_best[ok] = tg[ok]
    # boxes whose lattice cell is within max_range of the origin
    ci = int(np.floor(o[0] / LATTICE)); cj = int(np.floor(o[1] / LATTICE))
    rc = int(np.ceil(max_range / LATTICE)) + 1
    inv = 1.0 / np.where(np.abs(d) < 1e-12, 1e-12, d)
    for i in range(ci - rc, ci + rc + 1):
        for j in range(cj - rc, cj + rc + 1):
            lo = np.array([i * LATTICE + BOX_LO, j * LATTICE + BOX_LO, GROUND_Z])
            hi = np.array([i * LATTICE + BOX_HI, j * LATTICE + BOX_HI, BOX_TOP])
            c = 0.5 * (lo + hi)
            if np.linalg.norm(c[:2] - o[:2]) > max_range + 8:
                continue
            t1 = (lo - o) * inv
            t2 = (hi - o) * inv
            tn = np.minimum(t1, t2).max(axis=1)
            tf = np.maximum(t1, t2).min(axis=1)
            hit = (tn <= tf) & (tn > 0) & (tn < t_best)
            t_best[hit] = tn[hit]
    t_best[t_best > max_range] = np.inf
    return t_best


def _scan_from_dirs(dirs_body, R, t, extR, extT, rng, range_sigma, bearing_sigma_deg, blind, n_keep):
    R_wl = R @ extR
    t_wl = R @ extT + t
    d_w = dirs_body @ R_wl.T
    rg = raycast(t_wl, d_w)
    ok = np.isfinite(rg) & (rg > blind)
    dirs_body, rg = dirs_body[ok], rg[ok]
    if n_keep is not None:
        dirs_body, rg = dirs_body[:n_keep], rg[:n_keep]
    n = len(rg)
    # bearing noise: small rotation of the direction; range noise along the ray
    sig = np.deg2rad(bearing_sigma_deg)
    pert = rng.normal(0.0, sig, (n, 3))
    dn = dirs_body + np.cross(pert, dirs_body)
    dn /= np.linalg.norm(dn, axis=1, keepdims=True)
    rn = rg + rng.normal(0.0, range_sigma, n)
    pts = (dn * rn[:, None]).astype(np.float32)
    inten = (10.0 + 80.0 * rng.random(n)).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([pts, inten[:, None]], axis=1))


def livox_scan(k, R, t, n_pts=100000, seed=20260924, extR=None, extT=None, range_sigma=0.02, bearing_sigma_deg=0.05, blind=1.0):
    """Livox-Avia-shaped scan k (non-repetitive: Halton(2,3) offset per scan), lidar frame, (n,4) float32 xyzI."""
    extR = np.eye(3) if extR is None else extR
    extT = np.zeros(3) if extT is None else extT
    rng = np.random.default_rng(seed + 7919 * k)
    m = int(n_pts * 2.2) + 64
    start = 1 + k * m
    az = (halton(m, 2, start) - 0.5) * np.deg2rad(70.4)
    el = (halton(m, 3, start) - 0.5) * np.deg2rad(77.2)
    dirs = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
    out = _scan_from_dirs(dirs, R, t, extR, extT, rng, range_sigma, bearing_sigma_deg, blind, n_pts)
    return out


def hdl64_scan(k, R, t, seed=20260925, n_az=2032, range_sigma=0.04, bearing_sigma_deg=0.1, blind=1.0):
    """HDL-64-shaped scan: 64 rings, elevation +2 .. -24.33 deg, n_az azimuth steps (130 048 rays at 2032)."""
    rng = np.random.default_rng(seed + 104729 * k)
    el = np.deg2rad(np.linspace(2.0, -24.33, 64))
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False) + rng.uniform(0, 2 * np.pi / n_az)
    A, E = np.meshgrid(az, el, indexing="ij")
    dirs = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=-1).reshape(-1, 3)
    return _scan_from_dirs(dirs, R, t, np.eye(3), np.zeros(3), rng, range_sigma, bearing_sigma_deg, blind, None)


def trajectory_pose(k, speed=1.0, yaw_rate_deg=2.0, dt=0.1):
    """Constant speed along +x (world), constant yaw rate; pose of scan k."""
    return yaw_R(np.deg2rad(yaw_rate_deg) * dt * k), np.array([speed * dt * k, 0.0, 0.0])


def voxel_grid_downsample(pts_xyz, leaf):
    """pcl::VoxelGrid semantics: centroid per occupied leaf, output ordered by linear leaf index."""
    p = np.asarray(pts_xyz[:, :3], dtype=np.float32)
    inv = np.float32(1.0 / leaf)
    mn = np.floor(p.min(axis=0) * inv).astype(np.int64)
    mx = np.floor(p.max(axis=0) * inv).astype(np.int64)
    dims = mx - mn + 1
    ijk = np.floor(p * inv).astype(np.int64) - mn
    idx = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    order = np.argsort(idx, kind="stable")
    sidx = idx[order]
    starts = np.flatnonzero(np.concatenate([[True], sidx[1:] != sidx[:-1]]))
    cnt = np.diff(np.concatenate([starts, [len(sidx)]]))
    # centroid accumulated sequentially in float32, point by point in (stable) index order, as PCL's `centroid += pt` does (np.add.reduceat
    # would sum long runs pairwise): the k-th point of every leaf is added in lock step
    ps = p[order]
    sums = np.zeros((len(starts), 3), np.float32)
    for k in range(int(cnt.max())):
        live = cnt > k
        sums[live] = sums[live] + ps[starts[live] + k]
    return np.ascontiguousarray((sums / cnt.astype(np.float32)[:, None]).astype(np.float32))


def forward_without_imu(state, dt=0.1, cov_gyr=0.3, cov_acc=0.5):
    """ImuProcess::Forward_without_imu (src/IMU_Processing.cpp:486-553): constant-velocity prior; bias_g plays the
    role of the angular rate ('omega in constant model')."""
    s = np.array(state, dtype=np.float64, copy=True)
    R = s[0:9].reshape(3, 3); t = s[9:12]; vel = s[12:15]; bg = s[15:18]
    cov = s[24:].reshape(18, 18)
    F = np.eye(18)
    F[0:3, 0:3] = so3_exp(bg * (-dt))
    F[0:3, 9:12] = np.eye(3) * dt
    F[3:6, 6:9] = np.eye(3) * dt
    W = np.zeros((18, 18))
    W[9:12, 9:12] = np.eye(3) * cov_gyr * dt * dt
    W[6:9, 6:9] = np.eye(3) * cov_acc * dt * dt
    cov = F @ cov @ F.T + W
    R = R @ so3_exp(bg * dt)
    t = t + vel * dt
    s[0:9] = R.reshape(-1); s[9:12] = t; s[24:] = cov.reshape(-1)
    return s


def survey_points(x_range, y_range, density_per_m2, rng, range_sigma=0.02):
    """Dense survey of the world surfaces inside a rectangle (ground + box walls/roofs), world frame, (n,3) float64.
    Used to pre-build large registration maps (the 10 M-voxel map of BASELINE.json's metric)."""
    x0, x1 = x_range; y0, y1 = y_range
    area = (x1 - x0) * (y1 - y0)
    n = int(area * density_per_m2)
    g = np.stack([rng.uniform(x0, x1, n), rng.uniform(y0, y1, n), np.full(n, GROUND_Z)], axis=1)
    # drop ground samples under a box footprint
    fx = np.mod(g[:, 0], LATTICE); fy = np.mod(g[:, 1], LATTICE)
    under = (fx > BOX_LO) & (fx < BOX_HI) & (fy > BOX_LO) & (fy < BOX_HI)
    g = g[~under]
    g[:, 2] += rng.normal(0, range_sigma, len(g))
    parts = [g]
    for i in range(int(np.floor(x0 / LATTICE)), int(np.ceil(x1 / LATTICE))):
        for j in range(int(np.floor(y0 / LATTICE)), int(np.ceil(y1 / LATTICE))):
            bx0, bx1 = i * LATTICE + BOX_LO, i * LATTICE + BOX_HI
            by0, by1 = j * LATTICE + BOX_LO, j * LATTICE + BOX_HI
            if bx1 < x0 or bx0 > x1 or by1 < y0 or by0 > y1:
                continue
            h = BOX_TOP - GROUND_Z
            nw = int((bx1 - bx0) * h * density_per_m2)
            for axis, val in ((0, bx0), (0, bx1), (1, by0), (1, by1)):
                u = rng.uniform(bx0 if axis == 1 else by0, bx1 if axis == 1 else by1, nw)
                z = rng.uniform(GROUND_Z, BOX_TOP, nw)
                w = np.full(nw, val) + rng.normal(0, range_sigma, nw)
                parts.append(np.stack([w, u, z], axis=1) if axis == 0 else np.stack([u, w, z], axis=1))
    return np.concatenate(parts, axis=0)
