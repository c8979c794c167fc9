"""GPU parity tests proper (rows a1-a16): the HIP path through the C ABI vs the CPU oracle on identical seeded inputs.
Bars (BASELINE.json north_star): match index sets identical; plane normals / pose within 1e-5."""
import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle, make_hip
from parity_utils import compare_plane_tables

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _scan(k, n, cfg, kind="livox"):
    R, t = synth.trajectory_pose(k)
    extT = np.array(list(cfg.extT))
    raw = synth.livox_scan(k, R, t, n_pts=n, extT=extT) if kind == "livox" else synth.hdl64_scan(k, R, t, n_az=n)
    return R, t, raw


def _both(oracle_lib, hip_lib, cfg):
    return make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)


@pytest.mark.parametrize("n", [6, 300, 30000])
def test_map_build_parity(oracle_lib, hip_lib, n):
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000)
    o, h = _both(oracle_lib, hip_lib, cfg)
    R, t, raw = _scan(0, max(n, 100), cfg)
    pts = np.ascontiguousarray(raw[:n, :3])
    st = capi.make_state(R=R, t=t)
    o.map_build(pts, st); h.map_build(pts, st)
    a, b = o.dump_planes(), h.dump_planes()
    npl = compare_plane_tables(a, b, TOL)
    if n >= 30000:
        assert npl > 500
    assert h.counters()["n_root_voxels"] == o.counters()["n_root_voxels"]


def test_map_build_kitti_deep_octree(oracle_lib, hip_lib):
    cfg = capi.velodyne_config(cap_root_voxels=1 << 14, cap_scan_points=200000)
    o, h = _both(oracle_lib, hip_lib, cfg)
    R, t, raw = _scan(0, 1024, cfg, kind="hdl64")
    pts = np.ascontiguousarray(raw[:, :3])
    st = capi.make_state(R=R, t=t)
    o.map_build(pts, st); h.map_build(pts, st)
    a, b = o.dump_planes(), h.dump_planes()
    assert a["layer"].max() >= 2           # the 3 m roots really split
    assert compare_plane_tables(a, b, TOL) > 100


def test_residuals_parity(oracle_lib, hip_lib):
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000)
    o, h = _both(oracle_lib, hip_lib, cfg)
    R0, t0, raw0 = _scan(0, 40000, cfg)
    st0 = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    o.map_build(p0, st0); h.map_build(p0, st0)
    R1, t1, raw1 = _scan(1, 40000, cfg)
    down = synth.voxel_grid_downsample(raw1, 0.4)
    st = capi.make_state(R=R1 @ synth.so3_exp(np.array([1e-3, -2e-3, 1.5e-3])), t=t1 + np.array([0.02, -0.01, 0.01]), cov_diag=1e-4)
    st[24:].reshape(18, 18)[0:3, 0:3] = np.eye(3) * 1e-5
    ro, rh = o.residuals(down, st), h.residuals(down, st)
    assert ro["n_match"] > 1000
    np.testing.assert_array_equal(rh["match_idx"], ro["match_idx"])           # identical match sets
    s = np.sign(np.sum(rh["normals"] * ro["normals"], axis=1))
    np.testing.assert_allclose(rh["normals"] * s[:, None], ro["normals"], atol=TOL)
    np.testing.assert_allclose(rh["dis"] * s, ro["dis"], atol=TOL)
    np.testing.assert_allclose(rh["r_inv"], ro["r_inv"], rtol=1e-6)
    np.testing.assert_allclose(rh["HTH"], ro["HTH"], rtol=1e-7, atol=1e-7 * np.abs(ro["HTH"]).max())
    np.testing.assert_allclose(rh["HTz"], ro["HTz"], rtol=1e-7, atol=1e-7 * np.abs(ro["HTz"]).max())
    co, ch = o.counters(), h.counters()
    assert ch["n_plane_tests"] == co["n_plane_tests"] and ch["n_extra_probe"] == co["n_extra_probe"]


def test_register_and_update_stream_parity(oracle_lib, hip_lib):
    """A 6-scan stream: iterated EKF update + map growth each scan; pose and the whole plane table must agree."""
    cfg = capi.avia_config(cap_root_voxels=1 << 17, cap_scan_points=200000)
    o, h = _both(oracle_lib, hip_lib, cfg)
    R0, t0, raw0 = _scan(0, 30000, cfg)
    st0 = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    o.map_build(p0, st0); h.map_build(p0, st0)
    so = st0.copy(); so[12:15] = [1.0, 0, 0]; so[15:18] = [0, 0, np.deg2rad(2.0)]
    sh = so.copy()
    for k in range(1, 7):
        _, tk, raw = _scan(k, 30000, cfg)
        down = synth.voxel_grid_downsample(raw, 0.4)
        po, ph = synth.forward_without_imu(so), synth.forward_without_imu(sh)
        so, io = o.register(down, po, po)
        sh, ih = h.register(down, ph, ph, want_eff=(k == 1))
        assert ih["n_iter"] == io["n_iter"] and ih["n_match"] == io["n_match"]
        np.testing.assert_allclose(sh[:24], so[:24], rtol=0, atol=TOL)        # pose / velocity / biases
        np.testing.assert_allclose(sh[24:], so[24:], rtol=0, atol=TOL * np.abs(so[24:]).max())
        assert np.linalg.norm(sh[9:12] - tk) < 0.05
        if k == 1:
            assert len(ih["eff_pts"]) == ih["n_match"] and np.all(np.isfinite(ih["eff_norm_dis"]))
        o.map_update(down, so); h.map_update(down, sh)
    a, b = o.dump_planes(), h.dump_planes()
    assert compare_plane_tables(a, b, TOL) > 1000
    co, ch = o.counters(), h.counters()
    assert ch["n_refits"] == co["n_refits"] and ch["n_refit_pts"] == co["n_refit_pts"]
    assert ch["n_root_voxels"] == co["n_root_voxels"]


def test_update_state_machine_freeze(oracle_lib, hip_lib):
    """Many scans of the same wall patch: voxels reach max_points_size (100) and freeze (update_enable=0, points freed)."""
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=100000)
    for i in range(3):
        cfg.extT[i] = 0.0
    o, h = _both(oracle_lib, hip_lib, cfg)
    rng = np.random.default_rng(11)
    st = capi.make_state()
    for it in range(12):
        n = 400
        P = np.stack([rng.uniform(5.0, 6.0, n), rng.uniform(-0.5, 0.5, n), -1.3 + rng.normal(0, 0.004, n)], axis=1).astype(np.float32)
        if it == 0:
            o.map_build(P, st); h.map_build(P, st)
        else:
            o.map_update(P, st); h.map_update(P, st)
    a, b = o.dump_planes(), h.dump_planes()
    assert (a["update_enable"] == 0).sum() > 0
    compare_plane_tables(a, b, TOL)


@pytest.mark.parametrize("max_layer", [0, 2, 4])
def test_node_that_turns_nonplanar_exactly_when_it_fills(oracle_lib, hip_lib, max_layer):
    """OctoTree::UpdateOctoTree's planar branch (voxel_loc.cpp:233-251) runs the fullness test in the SAME call in which a refit turned the
    node non-planar: the node stops updating and drops its points.  One root voxel, max_points_size = 12: six coplanar points (planar at the 6th),
    then six points off the plane -- the 12th point triggers the refit (non-planar) and fills the node --, then more points (a max-layer node must
    ignore them; a shallower one routes them to children)."""
    cfg = capi.avia_config(cap_root_voxels=1 << 10, cap_scan_points=10000, max_layer=max_layer, max_points_size=12)
    for i in range(3):
        cfg.extT[i] = 0.0
    o, h = _both(oracle_lib, hip_lib, cfg)
    rng = np.random.default_rng(7)
    st = capi.make_state(cov_diag=1e-8)
    def pts(n, z_sigma, z0=0.2):
        return np.stack([10.0 + rng.uniform(0.03, 0.47, n), 10.0 + rng.uniform(0.03, 0.47, n), z0 + rng.normal(0, z_sigma, n)], axis=1).astype(np.float32)
    first = pts(6, 0.002, z0=0.04)
    o.map_build(first, st); h.map_build(first, st)
    a = o.dump_planes()
    assert len(a) == 1 and a[0]["is_plane"] == 1 and a[0]["n_points"] == 6
    second = pts(6, 0.002, z0=0.46)                      # a second sheet 42 cm above the first: lambda_min ~ 0.016 > 0.01 in every direction
    o.map_update(second, st); h.map_update(second, st)
    a, b = o.dump_planes(), h.dump_planes()
    root = a[[i for i, r in enumerate(a) if r["layer"] == 0][0]]
    assert root["is_plane"] == 0 and root["update_enable"] == 0, (root["is_plane"], root["update_enable"])   # non-planar AND frozen in one call
    compare_plane_tables(a, b, TOL)
    for _ in range(3):
        more = pts(7, 0.003, z0=0.04)
        o.map_update(more, st); h.map_update(more, st)
        a, b = o.dump_planes(), h.dump_planes()
        compare_plane_tables(a, b, TOL)
    if max_layer == 0:
        assert len(a) == 1                              # a max-layer node that stopped updating ignores everything
    else:
        assert len(a) > 1                               # a shallower one routed the later points to children, which initialised
    co, ch = o.counters(), h.counters()
    assert ch["n_refits"] == co["n_refits"] and ch["n_refit_pts"] == co["n_refit_pts"]


def test_edge_cases(oracle_lib, hip_lib):
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=100000)
    o, h = _both(oracle_lib, hip_lib, cfg)
    # negative coordinates, exact voxel-boundary values, z == 0 exactly, a single point
    P = np.array([[-1.0, -0.5, 0.0], [-1.0, -0.5, 0.0], [2.0, 3.0, 0.0], [-2.25, 4.5, -0.75], [1e-3, -1e-3, 0.0], [7.5, -7.5, 0.0]], np.float32)
    # small non-collinear in-plane offsets: exactly collinear clusters have a rank-1 covariance whose minimum eigenvector is arbitrary
    # (in the reference too), which no implementation can be compared on
    offs = np.array([[0, 0, 0], [0.01, 0, 0], [0, -0.02, 0], [0.013, 0.017, 0], [-0.021, 0.009, 0.001], [0.004, -0.015, -0.001]], np.float32)
    P = np.concatenate([P + o for o in offs] * 2)
    st = capi.make_state(t=np.array([0.0, 0.0, 0.5]))
    o.map_build(P, st); h.map_build(P, st)
    compare_plane_tables(o.dump_planes(), h.dump_planes(), TOL)
    one = np.array([[3.0, 1.0, 0.0]], np.float32)
    ro, rh = o.residuals(one, st), h.residuals(one, st)
    assert rh["n_match"] == ro["n_match"]
    o.map_update(one, st); h.map_update(one, st)
    compare_plane_tables(o.dump_planes(), h.dump_planes(), TOL)
    # bad arguments are errors, not crashes
    with pytest.raises(RuntimeError):
        h.map_update(one, st, n=0)


def test_register_and_update_stream_parity_kitti(oracle_lib, hip_lib):
    """velodyne.yaml (3 m root voxels, max_layer 4, max_points_size 1000): deep octrees.  The matcher scans the per-root flat lists of planar
    descendants, which the map update keeps in step with every plane-flag change -- match sets, poses and the whole plane table must agree."""
    cfg = capi.velodyne_config(cap_root_voxels=1 << 14, cap_scan_points=200000)
    o, h = _both(oracle_lib, hip_lib, cfg)
    R0, t0, raw0 = _scan(0, 700, cfg, kind="hdl64")
    st0 = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    o.map_build(p0, st0); h.map_build(p0, st0)
    so = st0.copy(); so[12:15] = [1.0, 0, 0]; so[15:18] = [0, 0, np.deg2rad(2.0)]
    sh = so.copy()
    for k in range(1, 6):
        _, tk, raw = _scan(k, 700, cfg, kind="hdl64")
        down = synth.voxel_grid_downsample(raw, 0.5)
        if k == 1:
            ro, rh = o.residuals(down, synth.forward_without_imu(so)), h.residuals(down, synth.forward_without_imu(sh))
            assert ro["n_match"] > 1000
            np.testing.assert_array_equal(rh["match_idx"], ro["match_idx"])
            co, ch = o.counters(), h.counters()
            assert ch["n_plane_tests"] == co["n_plane_tests"] and co["n_plane_tests"] > 2 * ro["n_match"]   # several leaves tested per point
        po, ph = synth.forward_without_imu(so), synth.forward_without_imu(sh)
        so, io = o.register(down, po, po)
        sh, ih = h.register(down, ph, ph)
        assert ih["n_iter"] == io["n_iter"] and ih["n_match"] == io["n_match"]
        np.testing.assert_allclose(sh[:24], so[:24], rtol=0, atol=TOL)
        o.map_update(down, so); h.map_update(down, sh)
    a, b = o.dump_planes(), h.dump_planes()
    assert a["layer"].max() >= 2 and compare_plane_tables(a, b, TOL) > 300


def test_register_without_overlap(oracle_lib, hip_lib):
    """A scan that hits no mapped voxel: zero matches; the update degenerates to the prior (state_propagat) on both sides."""
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=100000)
    o, h = _both(oracle_lib, hip_lib, cfg)
    R0, t0, raw0 = _scan(0, 5000, cfg)
    st0 = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    o.map_build(p0, st0); h.map_build(p0, st0)
    far = (np.random.default_rng(3).uniform(-1, 1, (500, 3)) + [0, 0, 500.0]).astype(np.float32)
    prior = capi.make_state(R=R0, t=t0 + np.array([0.1, 0.0, 0.0]), cov_diag=1e-4)
    so, io = o.register(far, prior, prior)
    sh, ih = h.register(far, prior, prior)
    assert io["n_match"] == 0 and ih["n_match"] == 0 and ih["n_iter"] == io["n_iter"]
    np.testing.assert_allclose(sh, so, rtol=0, atol=1e-12)


def test_plane_decisions_next_to_the_planarity_threshold(oracle_lib, hip_lib, record_property):
    """the device's plane fit uses cheap reciprocals / reciprocal square roots + Newton steps and regrouped sums where the oracle (and Eigen)
    divide and take IEEE roots -- differences of ~1e-13 relative, which could flip `min eigenvalue < planer_threshold` for a voxel whose smallest
    eigenvalue sits on the threshold.  3 000 root voxels whose point clouds are slabs with a thickness variance drawn around the threshold (hundreds of
    nodes end up with lambda_min within 2 % of it): map build + two updates (refits with the deferred / bounded decisions of the replay kernels), and every
    is_plane / update_enable / point count must equal the oracle's, the values to 1e-5."""
    cfg = capi.avia_config(cap_root_voxels=1 << 14, cap_scan_points=400000)
    o, h = _both(oracle_lib, hip_lib, cfg)
    rng = np.random.default_rng(20260925)
    thr = float(cfg.planer_threshold)
    n_vox, per = 3000, 40
    centres = np.stack(np.meshgrid(np.arange(60), np.arange(50), indexing="ij"), -1).reshape(-1, 2)[:n_vox] * 1.0   # one slab per 0.5 m voxel, 1 m apart
    clouds = []
    for rep in range(3):
        per = 40 if rep == 0 else 20     # (80 retained points at the end: below max_points_size, every node keeps refitting)
        pts = np.zeros((n_vox, per, 3))
        pts[:, :, 0] = centres[:, None, 0] + rng.uniform(0.02, 0.48, (n_vox, per))     # (in-plane variance 0.0176: the smallest eigenvalue is the thickness')
        pts[:, :, 1] = centres[:, None, 1] + rng.uniform(0.02, 0.48, (n_vox, per))
        # thickness: uniform in [-w, w] has variance w^2 / 3 -> w = sqrt(3 * thr * (1 + eps)), eps within +-2 %
        w = np.sqrt(3.0 * thr * (1.0 + rng.uniform(0.03, 0.09, n_vox)))     # (the 1 / n sample variance and the clipping below take ~6 % off)
        pts[:, :, 2] = 0.25 + np.clip(rng.uniform(-1.0, 1.0, (n_vox, per)) * w[:, None], -0.24, 0.24)
        clouds.append(np.ascontiguousarray(pts.reshape(-1, 3).astype(np.float32)))
    st = capi.make_state(R=np.eye(3), t=np.zeros(3))
    o.map_build(clouds[0], st); h.map_build(clouds[0], st)
    a, b = o.dump_planes(), h.dump_planes()
    compare_plane_tables(a, b, TOL)
    for rep in (1, 2):
        o.map_update(clouds[rep], st); h.map_update(clouds[rep], st)
    a, b = o.dump_planes(), h.dump_planes()
    compare_plane_tables(a, b, TOL)
    rel = np.abs(a["min_eig"].astype(np.float64) / thr - 1.0)
    record_property("nodes_within_1e-3_of_threshold", int((rel < 1e-3).sum()))
    assert (rel < 2e-2).sum() > 100 and a["is_plane"].sum() > 100 and (a["is_plane"] == 0).sum() > 100    # both outcomes, many of them close
    co, ch = o.counters(), h.counters()
    assert ch["n_refits"] == co["n_refits"] and ch["n_refit_pts"] == co["n_refit_pts"]
