"""Pins the CPU oracle (oracle/) -- the reference has NO tests or golden vectors, so these analytic
known-answer tests + independent numpy/scipy cross-checks are what anchors the restatement."""
import ctypes as C

import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- a1 key quantisation (voxel_mapping.cpp:118-127): truncation after -1 for negatives, NOT floor ----------------
@pytest.mark.parametrize("p,vs,key", [
    ((0.1, 0.6, 1.2), 0.5, (0, 1, 2)),
    ((-0.1, -0.6, -1.2), 0.5, (-1, -2, -3)),
    ((-1.0, -0.5, 0.0), 0.5, (-3, -2, 0)),     # exact negative integers land one voxel lower (-2.0 -> -3)
    ((2.999, -2.999, 3.0), 3.0, (0, -1, 1)),
])
def test_key_quantisation(oracle_lib, p, vs, key):
    out = np.zeros(3, np.int64)
    oracle_lib.orc_key(_dp(np.array(p, float)), C.c_double(vs), _dp(out))
    assert tuple(out) == key


# ---- a7 calcBodyVar (voxel_mapping.cpp:1221-1241) ---------------------------------------------------------------
def _body_var_numpy(p, range_inc, degree_inc):
    p = np.array(p, float)
    if p[2] == 0:
        p[2] = 1e-4
    rng = np.float32(np.sqrt(p @ p))
    rv = np.float32(range_inc) * np.float32(range_inc)
    dv = np.sin(float(np.float32(degree_inc)) * 0.017453293) ** 2
    d = p / np.linalg.norm(p)
    dh = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
    b1 = np.array([1, 1, -(d[0] + d[1]) / d[2]]); b1 /= np.linalg.norm(b1)
    b2 = np.cross(b1, d); b2 /= np.linalg.norm(b2)
    N = np.stack([b1, b2], axis=1)
    A = float(rng) * dh @ N
    return np.outer(d, d) * float(rv) + A @ (np.eye(2) * dv) @ A.T


@pytest.mark.parametrize("p", [(10.0, 0.0, 0.0), (3.0, -4.0, 1.5), (0.5, 0.2, -7.0), (1.0, 2.0, 0.0)])
def test_calc_body_var(oracle_lib, p):
    var = np.zeros(9)
    oracle_lib.orc_calc_body_var(_dp(np.array(p, float)), C.c_float(0.02), C.c_float(0.05), _dp(var))
    ref = _body_var_numpy(p, 0.02, 0.05)
    np.testing.assert_allclose(var.reshape(3, 3), ref, rtol=1e-10, atol=1e-18)
    # closed form: along-ray variance = range_inc^2, cross-ray variance = (range*sin(beam))^2
    d = np.array(p, float); d[2] = d[2] if d[2] != 0 else 1e-4
    r = np.linalg.norm(d); d /= r
    assert d @ var.reshape(3, 3) @ d == pytest.approx(np.float32(0.02) ** 2, rel=1e-6)
    e = np.linalg.eigvalsh(var.reshape(3, 3))
    assert e[-1] == pytest.approx((r * np.sin(0.05 * 0.017453293)) ** 2, rel=1e-5) or e[-1] == pytest.approx(0.0004, rel=1e-5)


# ---- symmetric eigen-solver & 18x18 inverse vs numpy -----------------------------------------------------------
def test_sym3_eigen_vs_numpy(oracle_lib):
    rng = np.random.default_rng(1)
    for _ in range(200):
        B = rng.normal(size=(3, 3)) * rng.choice([1e-3, 1.0, 50.0])
        A = B @ B.T
        ev, V = np.zeros(3), np.zeros(9)
        oracle_lib.orc_sym3_eigen(_dp(A.copy()), _dp(ev), _dp(V))
        V = V.reshape(3, 3)
        w = np.linalg.eigvalsh(A)
        np.testing.assert_allclose(np.sort(ev), w, rtol=1e-11, atol=1e-13 * w[-1])
        np.testing.assert_allclose(V @ np.diag(ev) @ V.T, A, rtol=0, atol=1e-12 * max(1, w[-1]))
        np.testing.assert_allclose(V.T @ V, np.eye(3), atol=1e-13)


def test_inverse_18(oracle_lib):
    rng = np.random.default_rng(2)
    B = rng.normal(size=(18, 18))
    A = B @ B.T + np.eye(18) * 1e-3
    Ai = np.zeros(324)
    assert oracle_lib.orc_inv(_dp(A.copy()), _dp(Ai), 18) == 0
    np.testing.assert_allclose(Ai.reshape(18, 18), np.linalg.inv(A), rtol=1e-8, atol=1e-10)


# ---- a5 init_plane: analytic plane + independent finite-difference check of plane_var ----------------------------
def _fit_numpy(P):
    c = P.mean(axis=0)
    Cm = (P.T @ P) / len(P) - np.outer(c, c)
    w, V = np.linalg.eigh(Cm)
    return c, V[:, 0], w


def test_init_plane_known_plane(oracle_lib):
    cfg = capi.avia_config()
    for i in range(3):
        cfg.extT[i] = 0.0
    hp = make_oracle(oracle_lib, cfg)
    rng = np.random.default_rng(3)
    n_true = np.array([0.3, -0.5, 0.8]); n_true /= np.linalg.norm(n_true)
    # 40 points on the plane n.x = 2.0 inside one 0.5 m voxel around (1.2, 0.7, c), lidar at origin, identity pose
    u = np.cross(n_true, [1, 0, 0]); u /= np.linalg.norm(u); v = np.cross(n_true, u)
    centre = n_true * 2.0 + 0.4 * u
    key0 = np.floor(centre / 0.5)
    lo, hi = key0 * 0.5 + 0.02, key0 * 0.5 + 0.48
    P = []
    while len(P) < 40:
        q = centre + rng.uniform(-0.25, 0.25) * u + rng.uniform(-0.25, 0.25) * v + n_true * rng.normal(0, 0.002)
        if np.all(q > lo) and np.all(q < hi):
            P.append(q)
    P = np.array(P, dtype=np.float32)
    state = capi.make_state()
    hp.map_build(P, state)
    recs = hp.dump_planes()
    assert len(recs) == 1 and recs[0]["is_plane"] == 1 and recs[0]["layer"] == 0
    r = recs[0]
    Pd = P.astype(np.float64)
    c, nrm, w = _fit_numpy(Pd)
    s = np.sign(nrm @ r["normal"])
    np.testing.assert_allclose(r["normal"] * s, nrm, atol=1e-9)
    np.testing.assert_allclose(r["center"], c, atol=1e-12)
    assert abs(abs(r["normal"] @ n_true) - 1) < 1e-3
    assert r["d"] == pytest.approx(np.float32(-(r["normal"] @ r["center"])), rel=1e-6)
    assert abs(abs(r["d"]) - 2.0) < 5e-3
    assert r["radius"] == pytest.approx(np.sqrt(w[2]), rel=1e-6)
    assert r["min_eig"] == pytest.approx(w[0], rel=1e-4, abs=1e-9)
    # plane_var = sum_i J_i Sigma_i J_i^T with J_i = d(normal,center)/d p_i : check J numerically via numpy refits
    pv = r["plane_var"].reshape(6, 6)
    np.testing.assert_allclose(pv, pv.T, atol=1e-18)
    Sig = []
    for q in Pd:
        var = np.zeros(9)
        oracle_lib.orc_calc_body_var(_dp(q.copy()), C.c_float(0.02), C.c_float(0.05), _dp(var))
        V = var.reshape(3, 3)
        K = np.array([[0, -q[2], q[1]], [q[2], 0, -q[0]], [-q[1], q[0], 0]])
        Sig.append(V + K @ (np.eye(3) * 1e-7) @ K.T + np.eye(3) * 1e-7)   # voxel_map_init :1260-1263 with cov = 1e-7 I
    acc = np.zeros((6, 6))
    eps = 1e-6
    for i in range(len(Pd)):
        J = np.zeros((6, 3))
        for k in range(3):
            Pp = Pd.copy(); Pp[i, k] += eps
            Pm = Pd.copy(); Pm[i, k] -= eps
            cp, np_, _ = _fit_numpy(Pp); cm, nm_, _ = _fit_numpy(Pm)
            np_ *= np.sign(np_ @ r["normal"]); nm_ *= np.sign(nm_ @ r["normal"])
            J[0:3, k] = (np_ - nm_) / (2 * eps); J[3:6, k] = (cp - cm) / (2 * eps)
        acc += J @ Sig[i] @ J.T
    np.testing.assert_allclose(pv, acc, rtol=2e-4, atol=1e-12)


# ---- a4 octree split: two orthogonal planes in one root voxel => root non-planar, children planar -----------------
def test_octree_split(oracle_lib):
    # 0.01 m^2 is a generous planarity threshold: an L-shaped corner inside a 0.5 m voxel still passes as a plane, so the
    # split is exercised with the KITTI parameters (3 m roots, 4 layers).
    cfg = capi.velodyne_config()
    hp = make_oracle(oracle_lib, cfg)
    rng = np.random.default_rng(4)
    n = 400
    floor_pts = np.stack([rng.uniform(3.1, 5.9, n), rng.uniform(0.1, 2.9, n), 0.2 + rng.normal(0, 0.003, n)], axis=1)
    wall_pts = np.stack([3.2 + rng.normal(0, 0.003, n), rng.uniform(0.1, 2.9, n), rng.uniform(0.3, 2.9, n)], axis=1)
    P = np.concatenate([floor_pts, wall_pts]).astype(np.float32)
    hp.map_build(P, capi.make_state())
    recs = hp.dump_planes()
    root = recs[recs["layer"] == 0]
    assert len(root) == 1 and root[0]["is_plane"] == 0 and tuple(root[0]["key"]) == (1, 0, 0)
    kids = recs[(recs["layer"] >= 1) & (recs["is_plane"] == 1)]
    assert len(kids) >= 2 and recs["layer"].max() <= 4
    assert np.sum(np.abs(kids["normal"][:, 2]) > 0.99) >= 1   # a floor child
    assert np.sum(np.abs(kids["normal"][:, 0]) > 0.99) >= 1   # a wall child
    # child geometry: path bits say which octant; a layer-1 child with x-bit set lies in x > 4.5
    for k in kids[kids["layer"] == 1]:
        xbit = (k["path"] >> 2) & 1
        assert (k["center"][0] > 4.5) == bool(xbit)


# ---- a10-a13 residual KAT: scan point at a known offset from a fitted plane --------------------------------------
def test_residual_known_offset(oracle_lib):
    cfg = capi.avia_config()
    for i in range(3):
        cfg.extT[i] = 0.0
    hp = make_oracle(oracle_lib, cfg)
    rng = np.random.default_rng(5)
    # horizontal plane z = -1.3 (inside voxel z-key -3: [-1.5,-1.0)), 5 x 5 voxels, 30 pts each
    xs = rng.uniform(3.0, 5.5, 3000); ys = rng.uniform(-1.0, 1.5, 3000)
    P = np.stack([xs, ys, -1.3 + rng.normal(0, 0.003, 3000)], axis=1).astype(np.float32)
    hp.map_build(P, capi.make_state())
    delta = 0.01
    Q = np.stack([rng.uniform(3.2, 5.3, 50), rng.uniform(-0.8, 1.3, 50), np.full(50, -1.3 + delta)], axis=1).astype(np.float32)
    r = hp.residuals(Q, capi.make_state())
    assert r["n_match"] >= 45
    np.testing.assert_allclose(np.abs(r["dis"]), delta, atol=2.5e-3)
    assert np.all(np.abs(np.abs(r["normals"][:, 2]) - 1) < 1e-3)
    # HTH/HTz consistent with the per-match rows: H = [ (p_imu x) R^T n , n ]
    H = []
    for j, i in enumerate(r["match_idx"]):
        nrm = r["normals"][j].astype(np.float32).astype(np.float64)
        p = Q[i].astype(np.float64)
        K = np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]])
        H.append(np.concatenate([K @ nrm, nrm]))
    H = np.array(H)
    HTH = (H * r["r_inv"][:, None]).T @ H
    HTz = (H * r["r_inv"][:, None]).T @ (-r["dis"].astype(np.float64))
    np.testing.assert_allclose(r["HTH"], HTH, rtol=1e-9, atol=1e-9 * np.abs(HTH).max())
    np.testing.assert_allclose(r["HTz"], HTz, rtol=1e-9, atol=1e-9 * np.abs(HTz).max())


# ---- a14 + A.13: the iterated update recovers an injected pose perturbation on synthetic scans -------------------
def test_register_recovers_pose(oracle_lib):
    cfg = capi.avia_config()
    hp = make_oracle(oracle_lib, cfg)
    extT = np.array(list(cfg.extT))
    R0, t0 = synth.trajectory_pose(0)
    raw = synth.livox_scan(0, R0, t0, n_pts=30000, extT=extT)
    hp.map_build(np.ascontiguousarray(raw[:, :3]), capi.make_state(R=R0, t=t0))
    R1, t1 = synth.trajectory_pose(1)
    raw1 = synth.livox_scan(1, R1, t1, n_pts=30000, extT=extT)
    down = synth.voxel_grid_downsample(raw1, 0.4)
    prior = capi.make_state(R=R1 @ synth.so3_exp(np.array([0.002, -0.003, 0.004])), t=t1 + np.array([0.03, -0.02, 0.015]), cov_diag=1e-3)
    cov0 = prior[24:].reshape(18, 18)
    cov0[0:3, 0:3] = np.eye(3) * 2e-5   # realistic attitude prior (sigma 0.26 deg); the matcher gate scales with range^2 * this
    post, info = hp.register(down, prior, prior)
    assert 2 <= info["n_iter"] <= 4 and info["n_match"] > 500
    err = post[9:12] - t1
    # x (along the street) is only weakly observed from far facades in the 70-deg FoV; y/z are well observed
    assert abs(err[0]) < 0.03 and abs(err[1]) < 0.01 and abs(err[2]) < 0.012
    dR = post[0:9].reshape(3, 3).T @ R1
    assert np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))) < 0.06
    cov = post[24:].reshape(18, 18)
    assert np.all(np.diag(cov)[:6] < 1e-3) and np.all(np.linalg.eigvalsh((cov + cov.T) / 2) > -1e-12)
