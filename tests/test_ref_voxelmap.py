"""The oracle's registration-map rows pinned to the REFERENCE'S OWN code: oracle/_ref/libref_voxelmap.so is compiled from /root/reference/src/voxel_loc.hpp,
voxel_loc.cpp (whole) and buildVoxelMap / BuildResidualListOMP / build_single_residual / updateVoxelMap / calcBodyVar / var_contrast cut out of
voxel_mapping.cpp by line range, behind an Eigen-shaped stub (oracle/ref_voxelmap/stubs: Eigen / PCL / ROS are not in the image).  Pinned: the
reference's logic -- key quantisation (A.1), the octree state machine (A.3: init at the 6th point, refit every 6 points over ALL retained points,
freeze at max_points_size, cut into octants, non-planar routing), which points a fit sees, the plane-covariance formula as written, the matcher's
float gates and its near-voxel retry with the unit mismatch (A.2), calcBodyVar.  NOT pinned: Eigen's arithmetic (the stub multiplies with plain sums;
EigenSolver is the oracle's Jacobi) -- values are compared to 1e-9 relative, every discrete outcome (node sets, is_plane, point counts, match sets,
layers) exactly.  The lists fed to the reference functions are the ones the oracle's callers build (orc_debug_tap)."""
import ctypes as C
import os

import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle, ROOT
from parity_utils import compare_plane_tables_fast


@pytest.fixture(scope="module")
def rv():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_voxelmap.so")
    if not os.path.exists(so):
        if os.path.exists("/root/reference/src/voxel_loc.cpp"):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref/libref_voxelmap.so not built and /root/reference absent")
    lib = C.CDLL(so)
    lib.rv_create.restype = C.c_void_p; lib.rv_create.argtypes = [C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_double]
    lib.rv_destroy.argtypes = [C.c_void_p]
    lib.rv_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.rv_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.rv_dump.restype = C.c_int64; lib.rv_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.rv_root_voxels.restype = C.c_int64; lib.rv_root_voxels.argtypes = [C.c_void_p]
    lib.rv_residual_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 6
    lib.rv_calc_body_var.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    lib.rv_key.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
    return lib


def _ref_map(rv, cfg):
    li = (C.c_int * 5)(*list(cfg.layer_init))
    return rv.rv_create(cfg.voxel_size, cfg.max_layer, li, cfg.max_points_size, cfg.planer_threshold)


def _ref_dump(rv, m):
    n = rv.rv_dump(m, None, 0)
    recs = np.zeros(n, capi.PLANE_DTYPE)
    if n:
        rv.rv_dump(m, recs.ctypes.data_as(C.c_void_p), n)
    return recs


def _tap(oracle_lib, o, which):
    f = oracle_lib.orc_debug_pv; f.restype = C.c_int64; f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    n = f(o.ctx, which, None, None, None, 0)
    p, pw, var = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 9))
    f(o.ctx, which, p.ctypes.data_as(C.c_void_p), pw.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), n)
    return p, pw, var


def _tap_on(oracle_lib, o):
    oracle_lib.orc_debug_tap.argtypes = [C.c_void_p, C.c_int32]
    oracle_lib.orc_debug_tap(o.ctx, 1)


def test_calc_body_var_and_key_against_the_reference_functions(oracle_lib, rv):
    rng = np.random.default_rng(11)
    oracle_lib.orc_calc_body_var.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    pts = np.concatenate([rng.normal(0, 20, (200, 3)), [[3.0, -4.0, 0.0], [0.5, 0.25, 0.0], [1e-3, 2e-3, 50.0]]])
    for p in pts:
        for (ri, di) in ((0.02, 0.05), (0.04, 0.1), (0.04, 0.01)):
            a, b = np.zeros(9), np.zeros(9)
            pa, pb = p.copy(), p.copy()
            oracle_lib.orc_calc_body_var(pa.ctypes.data_as(C.c_void_p), ri, di, a.ctypes.data_as(C.c_void_p))
            rv.rv_calc_body_var(pb.ctypes.data_as(C.c_void_p), ri, di, b.ctypes.data_as(C.c_void_p))
            np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-18)
    # VOXEL_LOC: quotient narrowed to float, -1 when negative, truncated -- not floor; exact negative integers land one voxel lower
    oracle_lib.orc_key.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
    pts = np.concatenate([rng.normal(0, 30, (500, 3)), [[-1.0, -0.5, 0.0], [-2.0, 1.0, -1.5], [0.49999999, -0.0000001, 2.9999999], [-3.0, -6.0, 3.0]]])
    for vs in (0.5, 3.0):
        for p in pts:
            ka, kb = np.zeros(3, np.int64), np.zeros(3, np.int64)
            oracle_lib.orc_key(p.ctypes.data_as(C.c_void_p), vs, ka.ctypes.data_as(C.c_void_p))
            rv.rv_key(p.ctypes.data_as(C.c_void_p), vs, kb.ctypes.data_as(C.c_void_p))
            assert np.array_equal(ka, kb), (p, vs)
    k = np.zeros(3, np.int64)
    rv.rv_key(np.array([-1.0, -0.5, 0.25]).ctypes.data_as(C.c_void_p), 0.5, k.ctypes.data_as(C.c_void_p))
    assert k.tolist() == [-3, -2, 0]


@pytest.mark.parametrize("kind", ["avia", "velodyne"])
def test_octree_states_of_the_reference_code_equal_the_oracles(oracle_lib, rv, kind):
    """buildVoxelMap on a first scan, then updateVoxelMap (behind the var_contrast sort) scan after scan: after every step the two maps hold the same
    node set with the same is_plane / update_enable / point counts, and the fitted planes agree to rounding."""
    if kind == "avia":
        cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000)
        scan = lambda k: synth.livox_scan(k, *synth.trajectory_pose(k), n_pts=30000, extT=np.array(list(cfg.extT)))
        leaf, n_scans = 0.4, 7
    else:
        cfg = capi.velodyne_config(cap_root_voxels=1 << 14, cap_scan_points=200000)
        scan = lambda k: synth.hdl64_scan(k, *synth.trajectory_pose(k), n_az=512)
        leaf, n_scans = 0.5, 5
    o = make_oracle(oracle_lib, cfg)
    _tap_on(oracle_lib, o)
    m = _ref_map(rv, cfg)
    R0, t0 = synth.trajectory_pose(0)
    raw0 = scan(0)
    st = capi.make_state(R=R0, t=t0)
    o.map_build(np.ascontiguousarray(raw0[:, :3]), st)
    p, _, var = _tap(oracle_lib, o, 0)
    rv.rv_build(m, p.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), len(p))
    assert rv.rv_root_voxels(m) == o.counters()["n_root_voxels"]
    n_pl = compare_plane_tables_fast(o.dump_planes(), _ref_dump(rv, m), 1e-9)
    assert n_pl > 300
    seen_layers, seen_frozen = set(), 0
    for k in range(1, n_scans):
        Rk, tk = synth.trajectory_pose(k)
        down = synth.voxel_grid_downsample(scan(k), leaf)
        sk = capi.make_state(R=Rk, t=tk, cov_diag=1e-6)
        o.map_update(down, sk)
        p, _, var = _tap(oracle_lib, o, 1)
        assert len(p) == len(down)
        rv.rv_update(m, p.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), len(p), 1)
        a, b = o.dump_planes(), _ref_dump(rv, m)
        n_pl = compare_plane_tables_fast(a, b, 1e-9)    # same node set / is_plane / update_enable / n_points / new_points are part of the comparison
        seen_layers |= set(np.unique(a["layer"]).tolist()); seen_frozen += int((a["update_enable"] == 0).sum())
    assert n_pl > 300
    if kind == "velodyne":
        assert max(seen_layers) >= 2      # the 3 m roots really subdivided (initialised nodes two layers down)
    rv.rv_destroy(m)


def test_octree_freeze_and_refit_cadence_on_one_voxel(oracle_lib, rv):
    """A.3 on a single root voxel, point by point: init at the 6th point, a refit on every 6th new point over ALL retained points, update_enable drops at
    max_points_size (100) and the points are freed -- identical states after every batch on both sides."""
    cfg = capi.avia_config(cap_root_voxels=1 << 10, cap_scan_points=10000)
    o = make_oracle(oracle_lib, cfg)
    _tap_on(oracle_lib, o)
    m = _ref_map(rv, cfg)
    rng = np.random.default_rng(4)
    ident = capi.make_state(cov_diag=1e-6)
    extT = np.array(list(cfg.extT))
    states = []
    for batch in range(30):
        n = int(rng.integers(1, 9))
        # points of the plane z = 0.2 inside the voxel [10, 10.5)^2 x [0, 0.5) (world = body + extT for the identity pose)
        w = np.stack([10.0 + rng.uniform(0.02, 0.48, n), 10.0 + rng.uniform(0.02, 0.48, n), 0.2 + rng.normal(0, 0.004, n)], axis=1)
        body = (w - extT).astype(np.float32)
        o.map_update(body, ident)
        p, _, var = _tap(oracle_lib, o, 1)
        rv.rv_update(m, p.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), len(p), 1)
        a, b = o.dump_planes(), _ref_dump(rv, m)
        assert len(a) == len(b)
        if len(a):
            compare_plane_tables_fast(a, b, 1e-9)
            states.append((int(a[0]["n_points"]), int(a[0]["new_points"]), int(a[0]["update_enable"])))
    assert any(s[2] == 0 and s[0] == 0 for s in states), states       # frozen: points freed, no more updates
    assert any(s[2] == 1 and s[0] > 6 for s in states)
    rv.rv_destroy(m)


@pytest.mark.parametrize("kind", ["avia", "velodyne"])
def test_matcher_of_the_reference_code_equals_the_oracles(oracle_lib, rv, kind):
    """BuildResidualListOMP + build_single_residual (incl. the near-voxel retry and the recursion into all eight children) on the map both sides built:
    the match index sets are identical; normal, centre, d, layer and plane_var of every match agree."""
    if kind == "avia":
        cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000)
        scan = lambda k: synth.livox_scan(k, *synth.trajectory_pose(k), n_pts=40000, extT=np.array(list(cfg.extT)))
        leaf = 0.4
    else:
        cfg = capi.velodyne_config(cap_root_voxels=1 << 14, cap_scan_points=200000)
        scan = lambda k: synth.hdl64_scan(k, *synth.trajectory_pose(k), n_az=512)
        leaf = 0.5
    o = make_oracle(oracle_lib, cfg)
    _tap_on(oracle_lib, o)
    m = _ref_map(rv, cfg)
    R0, t0 = synth.trajectory_pose(0)
    st = capi.make_state(R=R0, t=t0)
    o.map_build(np.ascontiguousarray(scan(0)[:, :3]), st)
    p, _, var = _tap(oracle_lib, o, 0)
    rv.rv_build(m, p.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), len(p))
    for k in (1, 2):
        Rk, tk = synth.trajectory_pose(k)
        down = synth.voxel_grid_downsample(scan(k), leaf)
        st = capi.make_state(R=Rk @ synth.so3_exp(np.array([1e-3, -2e-3, 1.5e-3])), t=tk + np.array([0.02, -0.01, 0.01]), cov_diag=1e-4)
        ro = o.residuals(down, st)
        _, pw, var = _tap(oracle_lib, o, 2)
        n = len(pw)
        idx, nrm, cen, d, lay = np.zeros(n, np.int32), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n), np.zeros(n, np.int32)
        M = rv.rv_residual_list(m, pw.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), n, cfg.voxel_size, cfg.sigma_num, idx.ctypes.data_as(C.c_void_p),
                                nrm.ctypes.data_as(C.c_void_p), cen.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), lay.ctypes.data_as(C.c_void_p), None)
        assert M == ro["n_match"] and M > 1000
        np.testing.assert_array_equal(idx[:M], ro["match_idx"])               # identical match sets, same (ascending) order
        np.testing.assert_allclose(nrm[:M], ro["normals"], rtol=0, atol=1e-9)
        if kind == "velodyne":
            assert lay[:M].max() >= 1                                          # matches found below the root: the recursion into the children ran
        # grow the map for the next round on both sides
        o.map_update(down, st)
        p, _, var2 = _tap(oracle_lib, o, 1)
        rv.rv_update(m, p.ctypes.data_as(C.c_void_p), var2.ctypes.data_as(C.c_void_p), len(p), 1)
    rv.rv_destroy(m)
