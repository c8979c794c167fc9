"""BASELINE.json's full size -- 100k-point scans into a ~10 M-root-voxel map -- where the CPU oracle is too slow to be the checker:
size-independent properties of the HIP path's own output (SURVEY 8(c) "invariants").
  * determinism / idempotence: the same stream twice, serial and asynchronous mesher -> bit-identical poses, vertices, triangles
  * registration recovers the injected prior error (pose within 5 cm of ground truth over the stream)
  * plane table: unit normals, d = -n.c, radius / min eigenvalue consistent, every planar node below the planarity threshold
  * mesh: vertices pairwise >= min_spacing apart, triplets sorted / unique / in range, live count == adds - removes,
    every triangle's vertices within the neighbourhood reach, replaying the diff lists on the host reproduces the live set size
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from immesh_amd import capi, synth  # noqa: E402
from conftest import make_hip  # noqa: E402

pytestmark = pytest.mark.gpu
N_SCANS = 6


def _run_stream(hip_lib, torch, mesh_mode, scans, n_voxels=10e6):
    import bench
    dev = torch.device("cuda", 0)
    cfg = capi.avia_config(cap_root_voxels=int(n_voxels * 1.3) + (1 << 16), cap_scan_points=2_500_000, cap_vertices=1 << 22, cap_triangles=1 << 24)
    h = make_hip(hip_lib, cfg)
    n_map = bench.build_big_map(h, cfg, torch, dev, n_voxels, float(np.sqrt(n_voxels / 8.8)) + 40.0)
    R0, t0 = synth.trajectory_pose(0)
    st = capi.make_state(R=R0, t=t0)
    st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
    live = set()
    verts = []
    poses = []
    tot_add = tot_rem = 0
    pending = None

    def collect():
        nonlocal tot_add, tot_rem
        m = h.mesh_fetch()
        verts.append(m["new_vtx"])
        rem = {tuple(t) for t in m["tri_rem"]}; add = {tuple(t) for t in m["tri_add"]}
        assert rem <= live and not (add & (live - rem))           # removes were live, adds were not
        live.difference_update(rem); live.update(add)
        tot_add += len(add); tot_rem += len(rem)
        return m

    last = None
    for k, (d_down, d_raw, n_ds, n_raw) in enumerate(scans):
        prior = capi.forward_without_imu_native(hip_lib, st) if k else st
        st, info = h.process_scan(d_down.data_ptr(), d_raw.data_ptr(), prior, prior, frame_idx=k, do_mesh=mesh_mode, n_ds=n_ds, n_raw=n_raw)
        if mesh_mode == 2:
            h.mesh_wait()
        last = collect()
        poses.append(st[:24].copy())
        assert info["n_match"] > 1500
    return {"h": h, "poses": np.array(poses), "verts": np.concatenate(verts), "live": live, "tot": (tot_add, tot_rem), "last": last, "n_map": n_map, "cfg": cfg}


@pytest.fixture(scope="module")
def scans():
    torch = pytest.importorskip("torch")
    cfg = capi.avia_config()
    extT = np.array(list(cfg.extT))
    out = []
    for k in range(N_SCANS):
        R, t = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, R, t, n_pts=100000, extT=extT)
        down = synth.voxel_grid_downsample(raw, 0.4)
        out.append((torch.from_numpy(down).cuda(), torch.from_numpy(raw).cuda(), len(down), len(raw)))
    return out


def test_full_size_stream_properties(hip_lib, scans):
    torch = pytest.importorskip("torch")
    a = _run_stream(hip_lib, torch, 1, scans)
    assert a["n_map"] >= 10_000_000
    # ---- registration tracks the ground-truth trajectory (1 m/s along x, the prior carries a constant-velocity error)
    for k in range(1, N_SCANS):
        _, t = synth.trajectory_pose(k)
        assert np.linalg.norm(a["poses"][k][9:12] - t) < 0.05, k
    # ---- plane table invariants
    planes = a["h"].dump_planes(cap=2_000_000)         # a 2 M-node sample of the >10 M-node table
    pl = planes[planes["is_plane"] == 1]
    assert len(pl) > 1_000_000
    n = pl["normal"]; c = pl["center"]
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-9)
    assert np.all(np.abs(pl["d"] + (n * c).sum(axis=1)) <= 1e-6 * np.maximum(1.0, np.abs(c).max(axis=1)))   # d = -n.c, stored in float32
    assert np.all(pl["min_eig"] < 0.01) and np.all(pl["radius"] >= 0) and np.all(pl["radius"] < 0.5)   # inside a 0.5 m voxel
    pv = pl["plane_var"].reshape(-1, 6, 6)[:100000]
    assert np.all(np.abs(pv - pv.transpose(0, 2, 1)) <= 1e-12 * np.abs(pv).max()) and np.all(np.einsum("nii->ni", pv) >= 0)
    # ---- mesh invariants
    V = a["verts"]
    cnt = a["h"].counters()
    assert len(V) == cnt["n_vertices"] and len(a["live"]) == cnt["n_triangles_live"] == a["tot"][0] - a["tot"][1]
    from scipy.spatial import cKDTree
    dd, _ = cKDTree(V.astype(np.float64)).query(V.astype(np.float64), k=2)
    assert dd[:, 1].min() >= 0.1 * (1 - 1e-6)                          # min-spacing rule of append_points_to_global_map
    T = np.array(sorted(a["live"]))
    assert np.all(T[:, 0] < T[:, 1]) and np.all(T[:, 1] < T[:, 2]) and T.min() >= 0 and T.max() < len(V)
    e = np.linalg.norm(V[T[:, 0]] - V[T[:, 1]], axis=1)
    assert e.max() < 2 * (0.4 * 1.25 + 0.4 * np.sqrt(3))              # both ends inside one voxel's neighbourhood reach
    m = a["last"]
    for key in ("tri_add", "tri_rem", "tri_upd"):                      # result lists sorted lexicographically, unique
        L = m[key]
        if len(L) > 1:
            order = np.lexsort((L[:, 2], L[:, 1], L[:, 0]))
            assert np.array_equal(order, np.arange(len(L))) and len(np.unique(L, axis=0)) == len(L)
    assert np.all(np.diff(m["smooth_ids"]) > 0)
    # ---- determinism: the same stream again with the asynchronous mesher -> bit-identical
    a["h"].close()
    b = _run_stream(hip_lib, torch, 2, scans)
    np.testing.assert_array_equal(b["poses"], a["poses"])
    np.testing.assert_array_equal(b["verts"], a["verts"])
    assert b["live"] == a["live"]
