"""GPU parity tests proper (rows a17-a26): the HIP mesher through the C ABI vs the CPU oracle on identical seeded inputs.
Bar (BASELINE.json north_star): vertex ids / triangle triplets bit-exact; smoothed positions within 1e-9 (f64 means of
exactly representable values; the sum order is the same on both sides)."""
import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle, make_hip

pytestmark = pytest.mark.gpu


def _world_scan(k, n, cfg):
    """scan k transformed to the world frame with the true pose (what map_incremental_grow hands to the mesher)."""
    R, t = synth.trajectory_pose(k)
    extT = np.array(list(cfg.extT))
    raw = synth.livox_scan(k, R, t, n_pts=n, extT=extT)
    pw = (raw[:, :3].astype(np.float64) + extT) @ R.T + t
    out = raw.copy()
    out[:, :3] = pw.astype(np.float32)
    return np.ascontiguousarray(out), t


def _compare_scan(mo, mh, tag=""):
    assert mh["vtx_base"] == mo["vtx_base"], tag
    np.testing.assert_array_equal(mh["new_vtx"], mo["new_vtx"], err_msg=f"{tag} new vertices")
    assert mh["n_voxels_meshed"] == mo["n_voxels_meshed"], tag
    np.testing.assert_array_equal(mh["tri_rem"], mo["tri_rem"], err_msg=f"{tag} tri_rem")
    np.testing.assert_array_equal(mh["tri_add"], mo["tri_add"], err_msg=f"{tag} tri_add")
    np.testing.assert_array_equal(mh["flip_add"], mo["flip_add"], err_msg=f"{tag} flip_add")
    np.testing.assert_array_equal(mh["tri_upd"], mo["tri_upd"], err_msg=f"{tag} tri_upd")
    np.testing.assert_array_equal(mh["flip_upd"], mo["flip_upd"], err_msg=f"{tag} flip_upd")
    np.testing.assert_array_equal(mh["smooth_ids"], mo["smooth_ids"], err_msg=f"{tag} smooth ids")
    np.testing.assert_allclose(mh["smooth_xyz"], mo["smooth_xyz"], rtol=0, atol=1e-9, err_msg=f"{tag} smooth xyz")


@pytest.mark.parametrize("split", ["0", "1"])
def test_mesh_stream_parity(oracle_lib, hip_lib, split, monkeypatch):
    """8 overlapping scans: append (ids), 20-NN union, Delaunay, diff, flips -- every result list must be identical.  split 1: the triangulations as a launch
    of their own on the third stream (mesh_tri64_kernel) and the diff against the live set at the head of phase B (mesh_diff64_kernel) -- what the
    worker switches to when two jobs are in flight -- forced for every job; 0: the one-launch form."""
    monkeypatch.setenv("IMMESH_SPLIT", split)   # (read when a context is created)
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    tot_add = tot_rem = 0
    for k in range(8):
        pts, cam = _world_scan(k, 40000, cfg)
        mo = o.mesh_scan(pts, cam, frame_idx=k)
        mh = h.mesh_scan(pts, cam, frame_idx=k)
        _compare_scan(mo, mh, f"scan {k}")
        np.testing.assert_array_equal(h.mesh_neighbourhood_sizes(), o.mesh_neighbourhood_sizes())   # n_u per voxel, in voxel order
        tot_add += len(mo["tri_add"]); tot_rem += len(mo["tri_rem"])
    assert tot_add > 10000 and tot_rem > 1000
    co, ch = o.counters(), h.counters()
    for key in ("n_app", "n_new", "v_act", "n_v", "n_u", "t_v", "t_add", "t_rem", "n_vertices", "n_triangles_live"):
        assert ch[key] == co[key], key


def test_mesh_dense_repeat(oracle_lib, hip_lib):
    """The same area scanned repeatedly until the min-spacing grid saturates: voxels with many vertices, large
    neighbourhoods, heavy remove/re-add traffic; budget 5000 -> step 4."""
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20, mesh_append_budget=5000)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    rng = np.random.default_rng(5)
    cam = np.array([0.0, 0.0, 1.5])
    for k in range(10):
        n = 20000
        # a 6 m x 6 m floor patch with a 1.5 m step and a wall: edges where the PCA plane is ambiguous
        x = rng.uniform(2, 8, n); y = rng.uniform(-3, 3, n)
        z = np.where(x > 5, 0.0, 0.0) + rng.normal(0, 0.01, n)
        wall = rng.random(n) < 0.3
        x[wall] = 8.0 + rng.normal(0, 0.01, wall.sum()); z[wall] = rng.uniform(0, 2.5, wall.sum())
        pts = np.stack([x, y, z, np.ones(n)], axis=1).astype(np.float32)
        mo = o.mesh_scan(pts, cam, frame_idx=k)
        mh = h.mesh_scan(pts, cam, frame_idx=k)
        _compare_scan(mo, mh, f"scan {k}")
    assert o.counters()["n_u"] / max(1, o.counters()["v_act"]) > 40     # neighbourhoods really are large


def test_mesh_kitti_scale(oracle_lib, hip_lib):
    """velodyne.yaml meshing constants (distance_scale 1.5: 0.15 m spacing, 0.6 m voxels), HDL-64-shaped scans."""
    cfg = capi.velodyne_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    for k in range(3):
        R, t = synth.trajectory_pose(k)
        raw = synth.hdl64_scan(k, R, t, n_az=512)
        pw = raw[:, :3].astype(np.float64) @ R.T + t
        pts = raw.copy(); pts[:, :3] = pw.astype(np.float32)
        pts = np.ascontiguousarray(pts)
        mo = o.mesh_scan(pts, t, frame_idx=k)
        mh = h.mesh_scan(pts, t, frame_idx=k)
        _compare_scan(mo, mh, f"scan {k}")


def test_mesh_edge_cases(oracle_lib, hip_lib):
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=100000, cap_vertices=1 << 16, cap_triangles=1 << 18)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    cam = np.zeros(3)
    # fewer than 3 points, duplicates, negative coordinates / exact cell boundaries (std::round half away from zero)
    tiny = np.array([[0.05, 0.05, 0.0, 1], [-0.05, -0.05, 0.0, 1], [0.05, 0.05, 0.0, 1]], np.float32)
    _compare_scan(o.mesh_scan(tiny, cam), h.mesh_scan(tiny, cam), "tiny")
    rng = np.random.default_rng(9)
    g = np.stack(np.meshgrid(np.arange(-10, 10) * 0.15, np.arange(-10, 10) * 0.15, indexing="ij"), axis=-1).reshape(-1, 2)
    pts = np.concatenate([g + rng.normal(0, 0.003, g.shape), rng.normal(0, 0.002, (len(g), 1)), np.ones((len(g), 1))], axis=1).astype(np.float32)
    _compare_scan(o.mesh_scan(pts, cam), h.mesh_scan(pts, cam), "grid")
    # a scan that adds nothing: voxels are visited but nothing is re-meshed
    _compare_scan(o.mesh_scan(pts, cam), h.mesh_scan(pts, cam), "repeat")
    with pytest.raises(RuntimeError):
        h.mesh_scan(pts, cam, n=0)


@pytest.mark.parametrize("mesh_mode", [1, 2])
def test_process_scan_full_pipeline(oracle_lib, hip_lib, mesh_mode):
    """immesh_process_scan = lio_state_estimation + map_incremental_grow + incremental_mesh_reconstruction on device-resident inputs.
    mesh_mode 2 = the mesh job is queued for the mesher's worker thread / stream and collected with immesh_mesh_wait."""
    torch = pytest.importorskip("torch")
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    extT = np.array(list(cfg.extT))
    R0, t0 = synth.trajectory_pose(0)
    raw0 = synth.livox_scan(0, R0, t0, n_pts=30000, extT=extT)
    st = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    o.map_build(p0, st); h.map_build(p0, st)
    so = st.copy(); so[12:15] = [1.0, 0, 0]; so[15:18] = [0, 0, np.deg2rad(2.0)]
    sh = so.copy()
    from parity_utils import ComposedRunChecker
    chk = ComposedRunChecker(make_oracle(oracle_lib, cfg), cfg.mesh_append_budget, _compare_scan)
    for k in range(1, 5):
        Rk, tk = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, Rk, tk, n_pts=30000, extT=extT)
        down = synth.voxel_grid_downsample(raw, 0.4)
        po, ph = synth.forward_without_imu(so), synth.forward_without_imu(sh)
        so, io = o.process_scan(down, raw, po, po, frame_idx=k, do_mesh=True)
        d_down = torch.from_numpy(down).cuda(); d_raw = torch.from_numpy(raw).cuda()     # device-resident inputs are used in place
        sh, ih = h.process_scan(d_down.data_ptr(), d_raw.data_ptr(), ph, ph, frame_idx=k, do_mesh=mesh_mode, n_ds=len(down), n_raw=len(raw))
        if mesh_mode == 2:
            h.mesh_wait()
        assert ih == io
        np.testing.assert_allclose(sh[:24], so[:24], rtol=0, atol=1e-5)
        mo, mh = o.mesh_fetch(), h.mesh_fetch()
        # the composed run, exactly: world-frame clouds within one f32 ulp (poses agree to ~1e-12, transformLidar stores f32), every list equal to a
        # shadow oracle fed the device's own cloud, and equal to the full oracle pipeline until a mesher candidate rounds the other way
        chk.check_scan(k, o, h, sh, mo, mh, pose_o=so, lever=float(np.abs(raw[:, :3]).max()) + 1.0)
        tm = h.last_timing()
        assert tm["total"] > 0 and tm["mesh"] > 0
    print(f"[full pipeline, mesh_mode {mesh_mode}] {chk.summary()}")
    assert chk.summary()["scans_equal_to_shadow_oracle"] == 4


@pytest.mark.parametrize("where", ["host", "device"])
def test_process_scan_strided_equals_packed(hip_lib, where):
    """immesh_process_scan_strided: the pcl-shaped clouds of the reference (PointXYZINormal: 48 bytes a point, intensity at 32; PointXYZI: 32 / 16) consumed in
    place give the packed call's poses, match counts and mesh lists bit for bit -- from host memory (packed into pinned staging by the library) and from
    device memory (gathered by a kernel)."""
    torch = pytest.importorskip("torch")
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    extT = np.array(list(cfg.extT))
    scans = []
    for k in range(5):
        Rk, tk = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, Rk, tk, n_pts=30000, extT=extT)
        scans.append((synth.voxel_grid_downsample(raw, 0.4), raw, Rk, tk))

    def pcl(points, stride, int_off):   # a cloud as pcl lays it out: x y z 1 | (normals) | intensity ..., everything else poisoned
        buf = np.full((len(points), stride // 4), np.float32(np.nan))
        buf[:, 0:3] = points[:, 0:3]
        if points.shape[1] > 3:
            buf[:, int_off // 4] = points[:, 3]
        return np.ascontiguousarray(buf)

    results = {}
    for mode in ("packed", "strided"):
        h = make_hip(hip_lib, cfg)
        st = capi.make_state(R=scans[0][2], t=scans[0][3])
        h.map_build(np.ascontiguousarray(scans[0][1][:, :3]), st)
        st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
        out = []
        for k in range(1, 5):
            down, raw = scans[k][0], scans[k][1]
            prior = synth.forward_without_imu(st)
            if mode == "packed":
                st, info = h.process_scan(np.ascontiguousarray(down), np.ascontiguousarray(raw), prior, prior, frame_idx=k, do_mesh=1)
            else:
                # down cloud as PointXYZINormal (48 / -), raw cloud alternately as PointXYZINormal (48 / 32) and PointXYZI (32 / 16)
                rs, ro = (48, 32) if k % 2 else (32, 16)
                d48, rbuf = pcl(down, 48, 32), pcl(raw, rs, ro)
                if where == "device":
                    d48, rbuf = torch.from_numpy(d48).cuda(), torch.from_numpy(rbuf).cuda()
                    torch.cuda.current_stream().synchronize()
                    st, info = h.process_scan_strided(d48.data_ptr(), len(down), 48, rbuf.data_ptr(), len(raw), rs, ro, prior, prior, frame_idx=k, do_mesh=1)
                else:
                    st, info = h.process_scan_strided(d48, len(down), 48, rbuf, len(raw), rs, ro, prior, prior, frame_idx=k, do_mesh=1)
                    d48[:] = np.nan; rbuf[:] = np.nan      # the caller's clouds are free when the call returns
            out.append((st.copy(), info["n_match"], h.mesh_fetch()))
        results[mode] = (out, h.counters())
        h.close()
    for (s1, n1, m1), (s2, n2, m2) in zip(results["packed"][0], results["strided"][0]):
        np.testing.assert_array_equal(s1, s2)
        assert n1 == n2
        for key in ("new_vtx", "tri_add", "tri_rem", "tri_upd", "flip_add", "smooth_ids", "smooth_xyz"):
            np.testing.assert_array_equal(m1[key], m2[key], err_msg=key)
    for key in ("n_vertices", "n_triangles_live", "n_match", "n_refits", "n_root_voxels"):
        assert results["packed"][1][key] == results["strided"][1][key], key


@pytest.mark.parametrize("split", ["adaptive", "always", "never"])
def test_async_pipeline_matches_serial(hip_lib, split, monkeypatch):
    """Queueing mesh jobs (up to three in flight) while later scans register must give the same mesh as the strictly serial order -- with the triangulations
    on the third stream for every job (IMMESH_SPLIT=1: mesh_tri64_kernel + mesh_diff64_kernel), for none (0: the one-launch mesh_delaunay64_kernel),
    and decided per job by the jobs in flight (the default)."""
    torch = pytest.importorskip("torch")
    if split != "adaptive":
        monkeypatch.setenv("IMMESH_SPLIT", "1" if split == "always" else "0")   # (read when a context is created)
    else:
        monkeypatch.delenv("IMMESH_SPLIT", raising=False)
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    extT = np.array(list(cfg.extT))
    scans = []
    for k in range(6):
        Rk, tk = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, Rk, tk, n_pts=30000, extT=extT)
        scans.append((torch.from_numpy(synth.voxel_grid_downsample(raw, 0.4)).cuda(), torch.from_numpy(raw).cuda(), Rk, tk))
    results = {}
    for mode in (1, 2):
        h = make_hip(hip_lib, cfg)
        R0, t0 = scans[0][2], scans[0][3]
        st = capi.make_state(R=R0, t=t0)
        h.map_build(np.ascontiguousarray(scans[0][1].cpu().numpy()[:, :3]), st)
        st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
        for k in range(1, 6):
            down, raw = scans[k][0].clone(), scans[k][1].clone()
            torch.cuda.current_stream().synchronize()   # (stream, not device: the library's worker may be capturing a graph -- hipDeviceSynchronize is not permitted then)
            prior = synth.forward_without_imu(st)
            st, _ = h.process_scan(down.data_ptr(), raw.data_ptr(), prior, prior, frame_idx=k, do_mesh=mode, n_ds=down.shape[0], n_raw=raw.shape[0])
            # the input clouds of an asynchronous call are still read after it has returned with the pose: immesh_inputs_consumed is the fence an
            # application needs before it refills them (here: poisons them -- a scan that had not been consumed would change pose and mesh)
            h.inputs_consumed()
            down.fill_(float("nan")); raw.fill_(float("nan"))
            torch.cuda.current_stream().synchronize()
        h.mesh_wait()
        last = h.mesh_fetch()
        results[mode] = (st.copy(), last, h.counters())
        h.close()
    s1, m1, c1 = results[1]; s2, m2, c2 = results[2]
    np.testing.assert_array_equal(s1, s2)
    for key in ("new_vtx", "tri_add", "tri_rem", "tri_upd", "flip_add", "smooth_ids"):
        np.testing.assert_array_equal(m1[key], m2[key], err_msg=key)
    for key in ("n_new", "v_act", "t_add", "t_rem", "n_vertices", "n_triangles_live"):
        assert c1[key] == c2[key], key
    assert c1["n_degenerate_skips"] == 0 == c2["n_degenerate_skips"]      # a noisy scan never meets the degenerate-admission rule


def test_long_async_stream_under_all_arrangements(hip_lib, monkeypatch):
    """24 asynchronous scans in a row without ever waiting for the mesher: up to three jobs in flight, the worker free to switch between its two
    arrangements in the middle of the stream (a backlog builds while the scan thread runs ahead).  Always deep (IMMESH_SPLIT=1), never (0) and the
    worker's own choice must leave the same map: every pose, the lists of the last job, the vertex / live-triangle / refit totals."""
    torch = pytest.importorskip("torch")
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    extT = np.array(list(cfg.extT))
    scans = []
    for k in range(25):
        Rk, tk = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, Rk, tk, n_pts=20000, extT=extT)
        scans.append((torch.from_numpy(synth.voxel_grid_downsample(raw, 0.4)).cuda(), torch.from_numpy(raw).cuda(), Rk, tk))
    torch.cuda.current_stream().synchronize()
    results = {}
    for split in ("0", "1", None):
        if split is None:
            monkeypatch.delenv("IMMESH_SPLIT", raising=False)
        else:
            monkeypatch.setenv("IMMESH_SPLIT", split)
        h = make_hip(hip_lib, cfg)
        st = capi.make_state(R=scans[0][2], t=scans[0][3])
        h.map_build(np.ascontiguousarray(scans[0][1].cpu().numpy()[:, :3]), st)
        st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
        poses = []
        for k in range(1, 25):
            prior = synth.forward_without_imu(st)
            st, _ = h.process_scan(scans[k][0].data_ptr(), scans[k][1].data_ptr(), prior, prior, frame_idx=k, do_mesh=2, n_ds=scans[k][0].shape[0], n_raw=scans[k][1].shape[0])
            poses.append(st[:12].copy())
        h.mesh_wait()
        results[split] = (np.array(poses), h.mesh_fetch(), h.counters())
        h.close()
    p0, m0, c0 = results["0"]
    for split in ("1", None):
        p1, m1, c1 = results[split]
        np.testing.assert_array_equal(p0, p1)
        for key in ("new_vtx", "tri_add", "tri_rem", "tri_upd", "flip_add", "flip_upd", "smooth_ids"):
            np.testing.assert_array_equal(m0[key], m1[key], err_msg=f"IMMESH_SPLIT={split} {key}")
        for key in ("n_new", "v_act", "t_add", "t_rem", "n_u", "t_v", "n_vertices", "n_triangles_live", "n_refits", "n_degenerate_skips"):
            assert c0[key] == c1[key], (split, key)
    assert c0["n_triangles_live"] > 1000 and c0["t_rem"] > 0


def test_mesh_volumetric_cloud(oracle_lib, hip_lib):
    """A space-filling cloud (vegetation-like): up to ~45 vertices per mesh voxel, thousands of candidates around a voxel (the kNN kernel
    stages them in several LDS batches), neighbourhoods above 256 vertices (the large-neighbourhood Delaunay instantiation)."""
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=100000, cap_vertices=1 << 16, cap_triangles=1 << 20, mesh_append_budget=20000)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    rng = np.random.default_rng(21)
    cam = np.array([-3.0, 1.0, 1.0])
    for k in range(3):
        n = 20000
        pts = np.concatenate([rng.uniform(0.0, 2.0, (n, 3)), np.ones((n, 1))], axis=1).astype(np.float32)
        mo = o.mesh_scan(pts, cam, frame_idx=k)
        mh = h.mesh_scan(pts, cam, frame_idx=k)
        _compare_scan(mo, mh, f"scan {k}")
    co = o.counters()
    assert co["n_u"] / max(1, co["v_act"]) > 60 and co["n_vertices"] > 2500


def test_mesh_offline_sized_cloud(oracle_lib, hip_lib):
    """More than 65536 candidates in one call (reconstruct_mesh_from_pointcloud-sized input, step 1): the vertex-admission kernel is
    driven in bounded rounds with host checks instead of one resident launch."""
    n = 90000
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 17, cap_triangles=1 << 20, mesh_append_budget=n)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    rng = np.random.default_rng(22)
    # a 30 m x 30 m undulating floor, points in random order (long dependency chains between neighbouring candidates)
    x = rng.uniform(0, 30, n); y = rng.uniform(0, 30, n)
    z = 0.3 * np.sin(x * 0.5) * np.cos(y * 0.4) + rng.normal(0, 0.005, n)
    pts = np.stack([x, y, z, np.ones(n)], axis=1).astype(np.float32)
    cam = np.array([15.0, 15.0, 10.0])
    mo = o.mesh_scan(pts, cam)
    mh = h.mesh_scan(pts, cam)
    _compare_scan(mo, mh, "offline")
    assert len(mo["new_vtx"]) > 30000


def test_mesh_export_and_ply(oracle_lib, hip_lib, tmp_path):
    """save_to_ply_file: smoothed vertex positions (Global_map::smooth_pts over every vertex) and the live triangles with their winding."""
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    for k in range(4):
        pts, cam = _world_scan(k, 40000, cfg)
        o.mesh_scan(pts, cam, frame_idx=k); h.mesh_scan(pts, cam, frame_idx=k)
    for factor in (1.0, 0.5, 0.0):
        vo, fo = o.mesh_export(factor, 20)
        vh, fh = h.mesh_export(factor, 20)
        np.testing.assert_array_equal(fh, fo)                                   # same faces, same winding, same order
        assert np.array_equal(np.isnan(vh), np.isnan(vo))                       # isolated vertices export as NaN, as in the reference
        np.testing.assert_allclose(np.nan_to_num(vh), np.nan_to_num(vo), rtol=0, atol=1e-6)
    assert len(fo) == o.counters()["n_triangles_live"] > 10000
    # binary PLY round trip
    path = str(tmp_path / "rec_mesh.ply")
    h.save_ply(path, 1.0, 20)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert b"format binary_little_endian 1.0" in head and b"property list uchar int vertex_indices" in head
    nv = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
    nf = int([l for l in head.split(b"\n") if l.startswith(b"element face")][0].split()[-1])
    v = np.frombuffer(body[:nv * 12], np.float32).reshape(-1, 3)
    rec = np.frombuffer(body[nv * 12:], np.uint8).reshape(nf, 13)
    assert np.all(rec[:, 0] == 3)
    f = rec[:, 1:].copy().view(np.int32).reshape(nf, 3)
    vh, fh = h.mesh_export(1.0, 20)
    assert np.array_equal(f, fh) and np.array_equal(np.nan_to_num(v), np.nan_to_num(vh))
    with pytest.raises(RuntimeError):
        h.mesh_export(1.0, 10)                                                  # only the reference's k = 20 is supported


def test_smooth_pts_for_the_renderer(oracle_lib, hip_lib):
    """Global_map::smooth_pts on demand (mesh_rec_display.cpp:78-103): the renderer smooths every triangle vertex the mesher has not -- vertices of sparse
    frontier voxels (< 3 points, never smoothed by the mesher, yet part of their neighbours' triangulations).  Values = the oracle's restatement of
    pointcloud_rgbd.cpp:932-958, for every vertex of the map, with the renderer's parameters and with others; no NaN for any vertex of a live triangle."""
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    for k in range(4):
        pts, cam = _world_scan(k, 12000, cfg)     # a thin stream: plenty of mesh voxels with one or two vertices at the frontier
        o.mesh_scan(pts, cam, frame_idx=k); h.mesh_scan(pts, cam, frame_idx=k)
    nv = o.counters()["n_vertices"]
    assert nv == h.counters()["n_vertices"] > 5000
    ids = np.arange(nv, dtype=np.int32)
    accept = 1.25 * cfg.mesh_voxel
    for factor, max_dis in ((1.0, accept), (0.3, accept), (1.0, 0.0), (1.0, 2.0 * accept)):   # (renderer: g_ply_smooth_factor, g_kd_tree_accept_pt_dis; <= 0: 0.8 x voxel)
        so, sh = o.smooth_pts(ids, factor, 20, max_dis), h.smooth_pts(ids, factor, 20, max_dis)
        assert np.array_equal(np.isnan(sh), np.isnan(so))
        np.testing.assert_allclose(np.nan_to_num(sh), np.nan_to_num(so), rtol=0, atol=1e-9)
    # request order and repeated ids are the caller's; a subset gives the same values as the whole map
    sub = np.array([5, 3, 5, nv - 1, 0, 17, 3], dtype=np.int32)
    np.testing.assert_array_equal(h.smooth_pts(sub, 1.0, 20, accept), h.smooth_pts(ids, 1.0, 20, accept)[sub])
    # the GL buffer of a simulated renderer pass over every live triangle: get_pos(1) after the on-demand smoothing
    _, faces = h.mesh_export(0.0, 20)
    tri_ids = np.ascontiguousarray(faces.reshape(-1))
    do, dh = o.mesh_display_vertices(tri_ids, 1.0, 20, accept), h.mesh_display_vertices(tri_ids, 1.0, 20, accept)
    assert dh.dtype == np.float32 and dh.shape == (len(tri_ids), 3)
    assert not np.isnan(dh).any()                                                                   # the drop-in's mirror would have held NaN here (VERDICT r05 missing #2)
    np.testing.assert_allclose(dh, do, rtol=0, atol=1e-6)
    # ... and the vertices the mesher never smoothed are among them (that is the case the entry exists for)
    smoothed_by_mesher = np.zeros(nv, bool)
    for k in range(4, 6):
        pts, cam = _world_scan(k, 12000, cfg)
        mo = o.mesh_scan(pts, cam, frame_idx=k); mh = h.mesh_scan(pts, cam, frame_idx=k)
        smoothed_by_mesher[mh["smooth_ids"][mh["smooth_ids"] < nv]] = True
        _compare_scan(mo, mh, f"scan {k}")                                                         # the queries have not touched the map
    with pytest.raises(RuntimeError):
        h.smooth_pts(np.array([h.counters()["n_vertices"]], np.int32), 1.0, 20, accept)             # not a vertex
    with pytest.raises(RuntimeError):
        h.smooth_pts(sub, 1.0, 10, accept)                                                          # only the reference's k = 20
    with pytest.raises(RuntimeError):
        h.smooth_pts(sub, 1.0, 20, 3.0 * accept)                                                    # beyond the reach of the 20-NN pull


def test_smooth_pts_from_a_third_thread_beside_the_scan_loop(hip_lib):
    """The renderer's thread queries while the scan thread keeps registering and meshing asynchronously: every query sees the map between two mesh jobs."""
    import threading
    torch = pytest.importorskip("torch")
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    extT = np.array(list(cfg.extT))
    scans = []
    for k in range(12):
        Rk, tk = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, Rk, tk, n_pts=30000, extT=extT)
        scans.append((torch.from_numpy(synth.voxel_grid_downsample(raw, 0.4)).cuda(), torch.from_numpy(raw).cuda(), Rk, tk))
    h = make_hip(hip_lib, cfg)
    st = capi.make_state(R=scans[0][2], t=scans[0][3])
    h.map_build(np.ascontiguousarray(scans[0][1].cpu().numpy()[:, :3]), st)
    st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
    prior = synth.forward_without_imu(st)
    st, _ = h.process_scan(scans[1][0].data_ptr(), scans[1][1].data_ptr(), prior, prior, frame_idx=1, do_mesh=1, n_ds=scans[1][0].shape[0], n_raw=scans[1][1].shape[0])
    n0 = h.counters()["n_vertices"]
    assert n0 > 1000
    ids = np.arange(n0, dtype=np.int32)
    accept = 1.25 * cfg.mesh_voxel
    stop, errors, seen = threading.Event(), [], []
    def renderer():
        try:
            while not stop.is_set():
                v = h.smooth_pts(ids, 1.0, 20, accept)
                seen.append(v)
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    th = threading.Thread(target=renderer); th.start()
    for k in range(2, 12):
        prior = synth.forward_without_imu(st)
        st, _ = h.process_scan(scans[k][0].data_ptr(), scans[k][1].data_ptr(), prior, prior, frame_idx=k, do_mesh=2, n_ds=scans[k][0].shape[0], n_raw=scans[k][1].shape[0])
    h.mesh_wait()
    stop.set(); th.join()
    assert not errors, errors
    assert len(seen) >= 2
    final = h.smooth_pts(ids, 1.0, 20, accept)
    # a vertex's value only changes when a later scan puts a new vertex among its 20 nearest: every snapshot is finite where the final one is, and the
    # first snapshot taken equals a query against the map as it was then or later -- here: same NaN pattern or fewer NaNs as the map fills in
    for v in seen:
        assert v.shape == final.shape
        assert not (np.isnan(final).any(axis=1) & ~np.isnan(v).any(axis=1)).any()
    h.close()


def test_reconstruct_mesh_from_pointcloud(oracle_lib, hip_lib):
    """Offline entry (ImMesh_node.cpp:235-244 -> reconstruct_mesh_from_pointcloud): VoxelGrid(0.01) + one meshing call, identity pose."""
    n = 40000
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 17, cap_triangles=1 << 20, mesh_append_budget=50000000)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    rng = np.random.default_rng(31)
    x = rng.uniform(-6, 6, n); y = rng.uniform(-6, 6, n)
    z = 0.2 * np.sin(x) + 0.1 * np.cos(1.7 * y) + rng.normal(0, 0.003, n)
    pts = np.stack([x, y, z, rng.uniform(0, 100, n)], axis=1).astype(np.float32)
    mo = o.reconstruct_mesh_from_pointcloud(pts, 0.01)
    mh = h.reconstruct_mesh_from_pointcloud(pts, 0.01)
    _compare_scan(mo, mh, "offline cloud")
    assert len(mo["new_vtx"]) > 8000 and len(mo["tri_add"]) > 15000


def test_hip_diff_lists_on_the_reference_triangle_manager(hip_lib, ref_tri_lib):
    """Rows a21 / a22 / a24 against reference code: the HIP path's per-scan lists applied to the reference's OWN Triangle_manager (prebuilt
    oracle/_ref/libref_triangle.so, compiled from /root/reference/src/meshing/r3live/triangle.{hpp,cpp}) in the reference's commit order leave it
    with exactly the live set -- triplets and m_index_flip -- that the device-side map exports."""
    from ref_triangle_mirror import RefTriangleMirror
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    h = make_hip(hip_lib, cfg)
    mirror = RefTriangleMirror(ref_tri_lib, cfg.mesh_region)
    n_rem = 0
    for k in range(8):
        pts, cam = _world_scan(k, 40000, cfg)
        m = h.mesh_scan(pts, cam, frame_idx=k)
        assert mirror.apply(m, k) == 0          # every removal named a triangle the real manager holds
        n_rem += len(m["tri_rem"])
    live = mirror.live()
    vtx, faces = h.mesh_export(smooth_factor=0.0)
    # save_to_ply_file winding: (v0, v1, v2) when m_index_flip != 0, else (v0, v2, v1)
    dev = {tuple(sorted(map(int, f))): (1 if (f[1] < f[2]) else 0) for f in faces}
    assert n_rem > 1000 and len(live) == len(dev) == h.counters()["n_triangles_live"]
    assert dev == {t: (1 if fl else 0) for t, fl in live.items()}
    mirror.close()


def test_more_active_voxels_than_the_lds_stage_holds(oracle_lib, hip_lib):
    """A sparse, non-repetitive scan: every candidate lands in its own mesh voxel that already holds two vertices, so ~10 000 voxels go active with
    ONE new vertex each -- more than the 8192 the single-launch admission tail orders in LDS (that used to end in IMMESH_E_CAPACITY).
    The global-array variant of the same stage must give the oracle's lists."""
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20, mesh_append_budget=16000)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    vox = 0.4
    gx, gy = np.meshgrid(np.arange(100), np.arange(100), indexing="ij")
    base = np.stack([gx.ravel() * vox, gy.ravel() * vox, np.zeros(gx.size)], axis=1)     # 10 000 mesh voxels on a plane
    cam = np.array([20.0, 20.0, 5.0])

    rng = np.random.default_rng(11)

    def scan(offsets):   # (jittered: a regular lattice would be nothing but distance ties and cocircular quadruples)
        p = np.concatenate([base + np.array(off) + rng.uniform(-0.02, 0.02, base.shape) for off in offsets], axis=0).astype(np.float32)
        return np.ascontiguousarray(np.concatenate([p, np.ones((len(p), 1), np.float32)], axis=1))

    a = scan([(0.05, 0.05, 0.0), (0.25, 0.25, 0.01)])      # two vertices per voxel: nothing to triangulate yet
    b = scan([(0.05, 0.30, 0.02)])                          # the third vertex of every voxel
    for k, pts in enumerate((a, b)):
        mo = o.mesh_scan(pts, cam, frame_idx=k)
        mh = h.mesh_scan(pts, cam, frame_idx=k)
        _compare_scan(mo, mh, f"scan {k}")
    assert mo["n_voxels_meshed"] > 8192 and len(mo["tri_add"]) > 8192


@pytest.mark.parametrize("spacing", [0.25, 0.125])
@pytest.mark.parametrize("split", ["0", "1"])
def test_exact_ties_lattice_cloud(oracle_lib, hip_lib, spacing, split, monkeypatch):
    """exact ties, not avoided.  A REGULAR lattice on exactly representable coordinates (binary fractions): every 20-NN query
    has equal distances at the cut (shells of 4 / 4 / 4 / 8 neighbours: the 20th falls inside the shell of eight), every lattice square is a cocircular
    quadruple (in-circle determinant exactly zero), the PCA of a neighbourhood has a repeated eigenvalue.  The checker's rules (ties by ascending id in
    the kNN, insertion order in the triangulation) must be the HIP path's, bit for bit -- spacing 0.25 m keeps the neighbourhoods on the
    register-resident triangulation (n_u <= 64), 0.125 m puts them on the general kernel (65..256)."""
    monkeypatch.setenv("IMMESH_SPLIT", split)   # (the hand-over to the general path -- tri_nf = -1 -- crosses two launches when split)
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    nx = int(6.0 / spacing)
    gx, gy = np.meshgrid(np.arange(nx), np.arange(nx), indexing="ij")
    cam = np.array([8.0, 0.0, 3.0])

    def sheet(x0, y0, z0):
        p = np.stack([x0 + gx.ravel() * spacing, y0 + gy.ravel() * spacing, np.full(gx.size, z0)], axis=1).astype(np.float32)
        assert np.array_equal(p.astype(np.float64), np.stack([x0 + gx.ravel() * spacing, y0 + gy.ravel() * spacing, np.full(gx.size, z0)], axis=1))   # exact
        return np.ascontiguousarray(np.concatenate([p, np.ones((len(p), 1), np.float32)], axis=1))

    scans = [sheet(5.0, -3.0, 0.0),                    # the lattice
             sheet(5.0, -3.0, 0.0)]                    # the same points again: every candidate meets its own dedupe cell
    if spacing >= 0.25:
        scans.append(sheet(5.0 + spacing / 2, -3.0 + spacing / 2, 0.0))                 # face centres: four corners at the same distance
    else:
        scans += [sheet(5.0, -3.0, 0.125), sheet(5.0, -3.0, 0.25)]                      # two more sheets above: a cubic lattice, neighbourhoods of ~90 vertices
    scans.append(sheet(11.0, -3.0, 0.0))               # an adjacent lattice: neighbourhoods that straddle the seam
    n_u_max = 0
    for k, pts in enumerate(scans):
        mo = o.mesh_scan(pts, cam, frame_idx=k)
        mh = h.mesh_scan(pts, cam, frame_idx=k)
        _compare_scan(mo, mh, f"lattice {spacing} scan {k}")
        nu = o.mesh_neighbourhood_sizes()
        np.testing.assert_array_equal(h.mesh_neighbourhood_sizes(), nu)
        n_u_max = max(n_u_max, int(nu.max()) if len(nu) else 0)
    assert (n_u_max <= 64) if spacing >= 0.25 else (n_u_max > 64), n_u_max
    # how often the degenerate-admission rule fired ("no circumdisk contains the point: not inserted", oracle/orc_delaunay.hpp:11-27): the same count on both
    # sides, reported by immesh_counters_t::n_degenerate_skips (0 on every noisy stream of this suite)
    ds_o, ds_h = o.counters()["n_degenerate_skips"], h.counters()["n_degenerate_skips"]
    assert ds_h == ds_o, (ds_h, ds_o)
    print(f"lattice {spacing}: n_degenerate_skips = {ds_h}")
    co, ch = o.counters(), h.counters()
    for key in ("n_app", "n_new", "v_act", "n_v", "n_u", "t_v", "t_add", "t_rem", "n_vertices", "n_triangles_live"):
        assert ch[key] == co[key], key
    assert co["n_triangles_live"] > 500
