"""The oracle's mesher pinned END TO END to the REFERENCE'S OWN code: oracle/_ref/libref_globalmap.so is incremental_mesh_reconstruction (whole),
Global_map::append_points_to_global_map, retrieve_neighbor_pts_kdtree, delaunay_triangulation (everything around the CGAL call), triangle_compare,
correct_triangle_index, class RGB_pts / RGB_Voxel, Hash_map_3d, the ikd-Tree and the Triangle_manager compiled from where they lie under /root/reference
(oracle/Makefile, oracle/ref_globalmap/ref_globalmap_wrap.cpp: whole files symlinked, excerpts cut by line range at build time; Eigen / PCL / CGAL / TBB
shaped stubs).  Rows a17, a19, a25 (the three the round-4 review named as restatement-only) and, once more, a18, a20-a24, a26 of SURVEY section 8(a).

Frames of the synthetic stream go through the reference's frame function and through the oracle's mesh_scan.  After EVERY frame: the same vertex ids
and positions (bit for bit), the same smoothed positions, the same neighbourhood-union size for every triangulated voxel in the same (ascending key)
order, the same visited-voxel set with the same (m_meshing_times, m_new_added_pts_count, points) per voxel, the same live triangle set, the same
m_index_flip -- except on triangles two voxels added in the same frame with DIFFERENT orientations: there the reference itself keeps whichever voxel
comes last in an unordered_map< shared_ptr > (pointer-hash order, ImMesh_mesh_reconstruction.cpp:234-244); the checker's rule is "the voxel with the
larger key" (SURVEY 8(c) determinism caveats), and the test requires the oracle's flip to be exactly that voxel's (taken from the reference's own
correct_triangle_index calls)."""
import ctypes as C
import os
import shutil

import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle, ROOT

VP = C.c_void_p


def _load_fresh(tmp_path):
    """The reference keeps the mesher's state in globals (g_map_rgb_pts_mesh, g_triangles_manager): every test loads its own copy of the library."""
    so = os.path.join(ROOT, "oracle", "_ref", "libref_globalmap.so")
    if not os.path.exists(so):
        if os.path.exists("/root/reference/src/ImMesh_mesh_reconstruction.cpp"):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref/libref_globalmap.so not built and /root/reference absent")
    mine = os.path.join(str(tmp_path), "libref_globalmap_%d.so" % os.getpid())
    shutil.copy(so, mine)
    lib = C.CDLL(mine)
    lib.rg_init.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    lib.rg_frame.argtypes = [VP, C.c_int, VP, VP, C.c_int]
    lib.rg_n_vertices.restype = C.c_int64
    lib.rg_vertices.argtypes = [VP, VP, C.c_int64]
    lib.rg_live.restype = C.c_int64; lib.rg_live.argtypes = [VP, VP, C.c_int64]
    lib.rg_n_u.argtypes = [VP, C.c_int]
    lib.rg_flip_calls.restype = C.c_int64; lib.rg_flip_calls.argtypes = [VP, VP, VP, C.c_int64]
    lib.rg_recent_voxels.restype = C.c_int64; lib.rg_recent_voxels.argtypes = [VP, VP, C.c_int64]
    lib.rg_n_voxels.restype = C.c_int64
    return lib


def _p(a):
    return a.ctypes.data_as(VP)


def _ref_live(lib):
    n = lib.rg_live(None, None, 0)
    tri, fl = np.zeros((max(n, 1), 3), np.int32), np.zeros(max(n, 1), np.uint8)
    lib.rg_live(_p(tri), _p(fl), n)
    return {tuple(t): int(f) for t, f in zip(tri[:n].tolist(), fl[:n].tolist())}


def _ref_vertices(lib):
    n = lib.rg_n_vertices()
    pos, sm = np.zeros((n, 3)), np.zeros((n, 3))
    lib.rg_vertices(_p(pos), _p(sm), n)
    return pos, sm


def _orc_vertices(oracle_lib, hp):
    f = oracle_lib.orc_mesh_vertices; f.restype = C.c_int; f.argtypes = [VP, VP, VP, C.c_int64]
    n = f(hp.ctx, None, None, 0)
    pos, sm = np.zeros((n, 3)), np.zeros((n, 3))
    f(hp.ctx, _p(pos), _p(sm), n)
    return pos, sm


def _scan_world(kind, k, n):
    R, t = synth.trajectory_pose(k)
    raw = synth.livox_scan(k, R, t, n_pts=n) if kind == "avia" else synth.hdl64_scan(k, R, t, n_az=n // 64)
    w = raw.copy()
    w[:, :3] = (raw[:, :3].astype(np.float64) @ R.T + t).astype(np.float32)
    return np.ascontiguousarray(w[:, :4], dtype=np.float32), R, t


@pytest.mark.parametrize("kind,n_pts,n_frames", [("avia", 20000, 7), ("avia", 100000, 3), ("velodyne", 64 * 512, 4)])
def test_frames_through_the_reference_mesher_equal_the_oracle(oracle_lib, tmp_path, kind, n_pts, n_frames):
    cfg = capi.avia_config() if kind == "avia" else capi.velodyne_config()
    hp = make_oracle(oracle_lib, cfg)
    lib = _load_fresh(tmp_path)
    lib.rg_init(cfg.mesh_min_spacing, cfg.mesh_voxel, cfg.mesh_region, cfg.mesh_append_budget, n_frames)
    flips = {}                 # the oracle's m_index_flip per live triangle, from its per-frame lists
    tolerated = set()          # triangles two voxels added with different orientations in one frame (see the module docstring)
    n_rem_total = n_multi = 0
    for k in range(n_frames):
        w, R, t = _scan_world(kind, k, n_pts)
        m = hp.mesh_scan(w, t, frame_idx=k)
        n_vox = lib.rg_frame(_p(w), len(w), _p(np.ascontiguousarray(R)), _p(np.ascontiguousarray(t)), k)
        # -- a17: vertex ids + positions (the id of a vertex is its index) ----------------------------------------------------------------------
        pos_r, sm_r = _ref_vertices(lib)
        pos_o, sm_o = _orc_vertices(oracle_lib, hp)
        assert len(pos_r) == len(pos_o) and len(pos_r) > 0, (k, len(pos_r), len(pos_o))
        np.testing.assert_array_equal(pos_r, pos_o)
        # -- a19: smoothed positions of every vertex, neighbourhood unions ------------------------------------------------------------------------
        np.testing.assert_allclose(sm_r, sm_o, rtol=0, atol=1e-12)
        nu_r = np.zeros(max(n_vox, 1), np.int32)
        assert lib.rg_n_u(_p(nu_r), n_vox) == n_vox
        nu_o = hp.mesh_neighbourhood_sizes()
        # the reference enters delaunay_triangulation for every selected voxel with >= 3 points; so does the oracle, in ascending key order
        np.testing.assert_array_equal(nu_r[:n_vox], nu_o)
        # -- a25: the visited-voxel set and what the selection left in every voxel -----------------------------------------------------------------
        nrv = lib.rg_recent_voxels(None, None, 0)
        keys, st = np.zeros((nrv, 3), np.int64), np.zeros((nrv, 3), np.int32)
        lib.rg_recent_voxels(_p(keys), _p(st), nrv)
        assert int((st[:, 2] >= 3).sum()) >= n_vox                      # (selected = visited, not yet meshed since its last new point, >= 3 points)
        assert (st[:, 0] == 1).all() and (st[:, 1] == 0).all()          # every visited voxel ends the frame meshed once, counters reset (:132-138)
        # -- a21-a24: live set + flips ---------------------------------------------------------------------------------------------------------------
        for tri, f in zip(map(tuple, m["tri_upd"].tolist()), m["flip_upd"].tolist()):   # (a kept triangle of one voxel may be another voxel's removal)
            flips[tri] = f
        for tri in map(tuple, m["tri_rem"].tolist()):
            flips.pop(tri, None); tolerated.discard(tri)
        n_rem_total += len(m["tri_rem"])
        for tri, f in zip(map(tuple, m["tri_add"].tolist()), m["flip_add"].tolist()):
            flips[tri] = f
        live_r = _ref_live(lib)
        assert set(live_r) == set(flips), (k, len(live_r), len(flips))
        # the reference's own correct_triangle_index calls of this frame, in voxel order: the last call on a triplet is the larger-key voxel's
        nc = lib.rg_flip_calls(None, None, None, 0)
        ctri, cfl, crk = np.zeros((nc, 3), np.int32), np.zeros(nc, np.int32), np.zeros(nc, np.int32)
        lib.rg_flip_calls(_p(ctri), _p(cfl), _p(crk), nc)
        assert (np.diff(crk) >= 0).all()
        added = set(map(tuple, m["tri_add"].tolist()))
        last, seen = {}, {}
        for tri, f in zip(map(tuple, np.sort(ctri, axis=1).tolist()), cfl.tolist()):
            last[tri] = f; seen.setdefault(tri, set()).add(f)
        for tri in added:
            assert flips[tri] == last[tri], (k, tri)                     # the oracle's rule, checked against the reference's own arithmetic
            if len(seen[tri]) > 1:
                tolerated.add(tri); n_multi += 1
        bad = [tri for tri in live_r if live_r[tri] != flips[tri] and tri not in tolerated]
        assert not bad, (k, len(bad), bad[:5])
    assert len(live_r) > 3000 and n_rem_total > 50
    assert len(tolerated) < 0.05 * len(live_r)                           # and the order-dependent ones are a sliver
    print(f"{kind}: {len(pos_r)} vertices, {len(live_r)} live triangles, {n_rem_total} removals, {n_multi} order-dependent flips")


def test_smooth_pts_and_save_to_ply_file_of_the_reference_equal_the_oracle(oracle_lib, tmp_path):
    """SURVEY 8(f) rank 3, pinned (round 6): Global_map::smooth_pts (pointcloud_rgbd.cpp:932-958, on the REAL ikd-Tree) and save_to_ply_file
    (mesh_rec_geometry.cpp:71-131, whole; PCL's file writer is a recorder) against the oracle's smooth_pts / export_mesh after the same frames."""
    cfg = capi.avia_config()
    hp = make_oracle(oracle_lib, cfg)
    lib = _load_fresh(tmp_path)
    lib.rg_smooth_pts.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, VP]
    lib.rg_save_ply.restype = C.c_int64; lib.rg_save_ply.argtypes = [C.c_double, C.c_double]
    lib.rg_ply_fetch.argtypes = [VP, VP]
    n_frames = 4
    lib.rg_init(cfg.mesh_min_spacing, cfg.mesh_voxel, cfg.mesh_region, cfg.mesh_append_budget, n_frames)
    for k in range(n_frames):
        w, R, t = _scan_world("avia", k, 12000)      # a thin stream: plenty of frontier voxels with one or two vertices (never smoothed by the mesher)
        hp.mesh_scan(w, t, frame_idx=k)
        lib.rg_frame(_p(w), len(w), _p(np.ascontiguousarray(R)), _p(np.ascontiguousarray(t)), k)
    nv = lib.rg_n_vertices()
    assert nv == hp.counters()["n_vertices"] > 5000
    accept = 1.25 * cfg.mesh_voxel
    # -- smooth_pts: the renderer's parameters, a partial factor, the "<= 0 -> 0.8 x voxel" default, a wide radius; every vertex of the map
    ids = np.arange(nv, dtype=np.int32)
    for factor, max_dis in ((1.0, accept), (0.3, accept), (1.0, 0.0), (1.0, 2.0 * accept)):
        so = hp.smooth_pts(ids, factor, 20, max_dis)
        sr = np.zeros((nv, 3))
        one = np.zeros(3)
        for i in range(nv):
            lib.rg_smooth_pts(i, factor, 20.0, max_dis, _p(one)); sr[i] = one
        assert np.array_equal(np.isnan(so), np.isnan(sr))                  # nobody within reach -> 0/0 on both sides
        assert np.isnan(sr).any(axis=1).sum() < nv // 2
        np.testing.assert_allclose(np.nan_to_num(so), np.nan_to_num(sr), rtol=0, atol=1e-12)
    # ... and the call left nothing behind (the wrapper puts the stored value back): the maps still agree
    pos_r, sm_r = _ref_vertices(lib); pos_o, sm_o = _orc_vertices(oracle_lib, hp)
    np.testing.assert_array_equal(pos_r, pos_o)
    np.testing.assert_allclose(sm_r, sm_o, rtol=0, atol=1e-12)
    # -- save_to_ply_file: vertices (float, smoothed with g_kd_tree_accept_pt_dis) and faces with their winding; the reference walks its region
    #    buckets, the oracle orders by sorted triplet: compared as sets of rows
    for factor in (0.0, 1.0):                                               # (1.0 last: the reference's smooth_pts stores its result in every point)
        vo, fo = hp.mesh_export(factor, 20)
        nf = lib.rg_save_ply(factor, 20.0)
        vr, fr = np.zeros((nv, 3), np.float32), np.zeros((max(nf, 1), 3), np.int32)
        lib.rg_ply_fetch(_p(vr), _p(fr))
        assert nf == len(fo) > 5000
        assert np.array_equal(np.isnan(vo), np.isnan(vr))
        np.testing.assert_array_equal(np.nan_to_num(vo), np.nan_to_num(vr))  # the same doubles cast to float
        # the same triangles; the winding of a face is its m_index_flip, equal except where the reference itself is order-dependent (a triangle two voxels
        # add in one frame with different orientations -- the module docstring; test_frames_through_the_reference_mesher_equal_the_oracle checks those
        # against the reference's own correct_triangle_index calls): a handful, never the rule
        wr = {tuple(sorted(f)): tuple(f) for f in fr[:nf].tolist()}; wo = {tuple(sorted(f)): tuple(f) for f in fo.tolist()}
        assert wr.keys() == wo.keys() and len(wr) == nf
        even = lambda a, b: (a.index(b[0]) - 0) % 3 == 0 and a[(a.index(b[0]) + 1) % 3] == b[1]   # same cyclic order
        n_other = sum(0 if even(wr[k], wo[k]) else 1 for k in wr)
        assert n_other <= nf // 200, (n_other, nf)           # (61 of 28 124 on this stream)
