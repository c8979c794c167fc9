"""Legacy registration path (`voxel_map_en = false`): the point map that stands in for the ikd-Tree and the "Old map ICP"
matcher.  tests/golden/ikdmap_r01.npz was produced by the reference's OWN ikd-Tree (tools/make_golden_ikdmap.py): Build + three
Add_Points(.., true) batches, the surviving point set, 5-NN answers.  The oracle (CPU) and the HIP path (GPU) must reproduce it bit for bit;
the matcher / EKF on top is compared HIP-vs-oracle (the plane fit restates Eigen's colPivHouseholderQr: unpinned, see oracle/orc_ikdmap.hpp)."""
import os

import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle, make_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "ikdmap_r01.npz")


def _cfg():
    return capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=100000, cap_vertices=1 << 12, cap_triangles=1 << 14)


def _check_against_golden(h):
    g = np.load(GOLD)
    h.ikd_build(g["scan0"], float(g["ds"]))
    sizes = [h.ikd_size()]
    for k in (1, 2, 3):
        h.ikd_add_points(g[f"scan{k}"])
        sizes.append(h.ikd_size())
    assert sizes == list(g["sizes"])                                   # KD_TREE::validnum() after Build and after every Add_Points
    np.testing.assert_array_equal(h.ikd_dump(), g["points"])           # the surviving points, bit for bit
    nn, d2, nf = h.ikd_knn(g["queries"])
    assert np.all(nf == 5)
    np.testing.assert_array_equal(d2, g["d2"])                         # float squared distances, ascending
    np.testing.assert_array_equal(nn, g["nn"])                         # and the neighbours themselves
    # note: the reference counts a point once per box that covers it; the two slabs overlap in a corner, the survivors are what is pinned
    h.ikd_delete_boxes(g["boxes"])                                     # Delete_Point_Boxes with two slabs
    np.testing.assert_array_equal(h.ikd_dump(), g["points_after_delete"])


def test_oracle_map_matches_reference_ikdtree_golden(oracle_lib):
    _check_against_golden(make_oracle(oracle_lib, _cfg()))


def test_oracle_map_matches_reference_ikdtree_live(oracle_lib, ref_ikd_lib):
    """fresh random batches through the reference's ikd-Tree, right now (only where /root/reference exists)"""
    import ctypes as C
    L = ref_ikd_lib
    L.ref_ikd_create.restype = C.c_void_p
    tree = C.c_void_p(L.ref_ikd_create())
    L.ref_ikd_set_downsample(tree, C.c_float(0.5))
    rng = np.random.default_rng(11)
    o = make_oracle(oracle_lib, _cfg())
    first = rng.uniform(-8, 8, (3000, 3)).astype(np.float32); first[:, 2] *= 0.1
    L.ref_ikd_build(tree, first.ctypes.data_as(C.c_void_p), len(first))
    o.ikd_build(first, 0.5)
    for _ in range(4):
        b = rng.uniform(-9, 9, (2500, 3)).astype(np.float32); b[:, 2] *= 0.1
        L.ref_ikd_add_points_ds(tree, b.ctypes.data_as(C.c_void_p), len(b))
        o.ikd_add_points(b)
        assert L.ref_ikd_validnum(tree) == o.ikd_size()
    n = o.ikd_size()
    pts = np.zeros((n + 8, 3), np.float32)
    assert L.ref_ikd_flatten(tree, pts.ctypes.data_as(C.c_void_p), n + 8) == n
    pts = pts[:n]
    np.testing.assert_array_equal(pts[np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0]))], o.ikd_dump())


def test_oracle_legacy_registration_recovers_pose(oracle_lib):
    """KAT: a map of the scene, a scan from a known pose, a perturbed prior -> the iterated update pulls the pose back"""
    cfg = _cfg()
    o = make_oracle(oracle_lib, cfg)
    extT = np.array(list(cfg.extT))
    maps = []
    for k in range(3):
        R, t = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, R, t, n_pts=15000, extT=extT)
        down = synth.voxel_grid_downsample(raw, 0.4)
        maps.append((((down.astype(np.float64) + extT) @ R.T) + t).astype(np.float32))
    o.ikd_build(maps[0], 0.4)
    o.ikd_add_points(maps[1]); o.ikd_add_points(maps[2])
    R, t = synth.trajectory_pose(1)
    down = synth.voxel_grid_downsample(synth.livox_scan(7, R, t, n_pts=15000, extT=extT), 0.4)
    truth = capi.make_state(R=R, t=t, cov_diag=1e-3)
    prior = capi.make_state(R=R @ synth.so3_exp(np.array([2e-3, -1e-3, 3e-3])), t=t + np.array([0.03, -0.02, 0.01]), cov_diag=1e-3)
    st, info = o.ikd_register(down, prior, prior)
    assert info["n_match"] > 500 and 2 <= info["n_iter"] <= 4
    assert np.linalg.norm(st[9:12] - t) < 0.5 * np.linalg.norm(prior[9:12] - t)
    nv = info["normals_pd2"]
    np.testing.assert_allclose(np.linalg.norm(nv[:, :3], axis=1), 1.0, atol=1e-5)
    assert np.all(np.abs(nv[:, 3]) <= 2.0) and np.all(np.diff(info["match_idx"]) > 0)
    assert truth is not None


def _fov_run(h, pts):
    """laser_map_fov_segment: a 120 m cube, 20 m detection range -> the cube shifts when the sensor is within 30 m of a face"""
    h.ikd_build(pts, 0.5)
    out = []
    for pos in ([0, 0, 0], [10, 0, 0], [31, 0, 0], [31, -32, 0], [80, -32, 2]):
        n_del = h.ikd_fov_segment(np.array(pos, np.float64), 120.0, 20.0)
        out.append((n_del, h.ikd_size()))
    return out, h.ikd_dump()


def test_oracle_fov_segment_known_answers(oracle_lib):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-70, 70, (20000, 3)).astype(np.float32); pts[:, 2] *= 0.05
    o = make_oracle(oracle_lib, _cfg())
    hist, left = _fov_run(o, pts)
    # by hand: the first call initialises the cube [-60, 60)^3; at x = 10 no face is within 1.5 * 20 m; at x = 31 the +x face is 29 m away ->
    # shift by mov_dist = max((120 - 60) * 0.45, 20 * 0.5) = 27 m and delete the slab x in [-60, -33); then the -y face (y = -32: 28 m) ->
    # slab y in [33, 60) of the shifted cube; then +x again (x = 80: 7 m) -> slab x in [-33, -6)
    def slab(p, lo, hi):
        return np.all((p >= np.array(lo, np.float32)) & (p < np.array(hi, np.float32)), axis=1)
    keep = np.ones(len(pts), bool)
    expect = [(0, len(pts)), (0, len(pts))]
    for lo, hi in (([-60, -60, -60], [-33, 60, 60]), ([-33, 33, -60], [87, 60, 60]), ([-33, -87, -60], [-6, 33, 60])):
        gone = keep & slab(pts, lo, hi)
        keep &= ~gone
        expect.append((int(gone.sum()), int(keep.sum())))
    assert hist == expect and expect[2][0] > 1000 and expect[3][0] > 500 and expect[4][0] > 1000
    rest = pts[keep]
    np.testing.assert_array_equal(left, rest[np.lexsort((rest[:, 2], rest[:, 1], rest[:, 0]))])


@pytest.mark.gpu
def test_hip_fov_segment_matches_oracle(oracle_lib, hip_lib):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-70, 70, (20000, 3)).astype(np.float32); pts[:, 2] *= 0.05
    ho, lo = _fov_run(make_oracle(oracle_lib, _cfg()), pts)
    hh, lh = _fov_run(make_hip(hip_lib, _cfg()), pts)
    assert hh == ho
    np.testing.assert_array_equal(lh, lo)


@pytest.mark.gpu
def test_hip_map_matches_reference_ikdtree_golden(hip_lib):
    _check_against_golden(make_hip(hip_lib, _cfg()))


@pytest.mark.gpu
def test_hip_legacy_registration_matches_oracle(oracle_lib, hip_lib):
    cfg = _cfg()
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    extT = np.array(list(cfg.extT))
    st_o = st_h = None
    for k in range(4):
        R, t = synth.trajectory_pose(k)
        down = synth.voxel_grid_downsample(synth.livox_scan(k, R, t, n_pts=20000, extT=extT), 0.4)
        if k == 0:
            w = (((down.astype(np.float64) + extT) @ R.T) + t).astype(np.float32)
            o.ikd_build(w, 0.4); h.ikd_build(w, 0.4)
            st_o = st_h = capi.make_state(R=R, t=t, cov_diag=1e-3)
            st_o[12:15] = [1.0, 0, 0]; st_o[15:18] = [0, 0, np.deg2rad(2.0)]
            st_h = st_o.copy()
            continue
        pr_o, pr_h = synth.forward_without_imu(st_o), synth.forward_without_imu(st_h)
        st_o, io = o.ikd_register(down, pr_o, pr_o)
        st_h, ih = h.ikd_register(down, pr_h, pr_h)
        assert ih["n_iter"] == io["n_iter"] and ih["n_match"] == io["n_match"] > 800
        np.testing.assert_array_equal(ih["match_idx"], io["match_idx"])
        np.testing.assert_array_equal(ih["normals_pd2"], io["normals_pd2"])      # float plane fit + float residual: same operations, same order
        np.testing.assert_allclose(st_h[:24], st_o[:24], rtol=0, atol=1e-9)
        np.testing.assert_allclose(st_h[24:], st_o[24:], rtol=1e-7, atol=1e-12)
        assert np.linalg.norm(st_h[9:12] - t) < 0.05
        # map growth: m_ikdtree.Add_Points(feats_down_world, true) with the registered pose
        for hh, st in ((o, st_o), (h, st_h)):
            Rk, tk = st[:9].reshape(3, 3), st[9:12]
            hh.ikd_add_points((((down.astype(np.float64) + extT) @ Rk.T) + tk).astype(np.float32))
        assert h.ikd_size() == o.ikd_size()
    np.testing.assert_array_equal(h.ikd_dump(), o.ikd_dump())
