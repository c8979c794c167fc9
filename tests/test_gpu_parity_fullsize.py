"""HIP path vs the CPU oracle AT BASELINE.json's sizes -- not properties, the checker itself:
  (i)   configs[1]/[2]: >= 10 scans of the 100 000-pt Livox-Avia stream, mesher on, against a >= 1 M-root-voxel map that BOTH sides build
        from the same survey strips (the trajectory corridor of the bench's 10 M-voxel survey, so that the oracle finishes in seconds);
  (ii)  configs[3]: full-width HDL-64 scans (2032 azimuth steps x 64 rings = 130 048 rays, velodyne.yaml, max_layer 4, 3 m roots);
  (iii) configs[4]'s scan size: one 500 000-pt scan.
Bars (north_star): match-index sets identical, plane tables / pose within 1e-5, vertex ids + positions and every triangle list bit-exact.
The COMPOSED run (registration -> map growth -> meshing in one call) is compared exactly for every scan (parity_utils.ComposedRunChecker):
the device's world-frame cloud may differ from the oracle's by at most one f32 ulp per coordinate (poses agree to ~1e-12; transformLidar stores
f32), a shadow oracle mesher fed the device's own cloud must reproduce every list of every scan bit for bit, and until a mesher candidate has
rounded the other way the lists must equal the full oracle pipeline's.  The mesher is ALSO compared on the oracle's world-frame clouds."""
import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle, make_hip
from parity_utils import compare_plane_tables_fast, ComposedRunChecker
from test_gpu_mesher import _compare_scan

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
TOL = 1e-5


def _world(raw, st, cfg):
    """transformLidar of the full scan with pose `st` (f64 compute, f32 store) -- the SAME array is handed to both meshers"""
    R = np.asarray(st[0:9]).reshape(3, 3); t = np.asarray(st[9:12])
    extR = np.array(list(cfg.extR)).reshape(3, 3); extT = np.array(list(cfg.extT))
    p = raw[:, :3].astype(np.float64) @ extR.T + extT
    out = raw.copy()
    out[:, :3] = (p @ R.T + t).astype(np.float32)
    return np.ascontiguousarray(out)


def _exact(mo, mh):
    return all(np.array_equal(mh[k], mo[k]) for k in ("new_vtx", "tri_add", "tri_rem", "tri_upd", "flip_add", "flip_upd", "smooth_ids")) and mh["vtx_base"] == mo["vtx_base"]


def test_avia_100k_stream_into_1m_voxel_map(oracle_lib, hip_lib, record_property):
    torch = pytest.importorskip("torch")
    import bench
    dev = torch.device("cuda", 0)
    n_vox = 1.0e6
    caps = dict(cap_root_voxels=int(n_vox * 1.6), cap_scan_points=2_500_000, cap_vertices=1 << 22, cap_triangles=1 << 24)
    cfg = capi.avia_config(**caps)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    o2, h2 = make_oracle(oracle_lib, capi.avia_config(cap_root_voxels=1 << 12, **{k: v for k, v in caps.items() if k != "cap_root_voxels"})), \
        make_hip(hip_lib, capi.avia_config(cap_root_voxels=1 << 12, **{k: v for k, v in caps.items() if k != "cap_root_voxels"}))   # mesher-only pair
    nv = bench.build_big_map(h, cfg, torch, dev, n_vox, float(np.sqrt(n_vox / 8.8)) + 40.0, also=o)
    assert nv >= 1_000_000 and o.counters()["n_root_voxels"] == nv
    co, ch = o.counters(), h.counters()
    assert ch["n_refits"] == co["n_refits"] and ch["n_refit_pts"] == co["n_refit_pts"]
    n_planar = compare_plane_tables_fast(o.dump_planes(), h.dump_planes(), TOL)
    assert n_planar > 800_000
    record_property("planar_nodes_compared", n_planar)

    extT = np.array(list(cfg.extT))
    R0, t0 = synth.trajectory_pose(0)
    so = capi.make_state(R=R0, t=t0)
    so[12:15] = [1.0, 0, 0]; so[15:18] = [0, 0, np.deg2rad(2.0)]
    sh = so.copy()
    o3 = make_oracle(oracle_lib, capi.avia_config(cap_root_voxels=1 << 12, **{k: v for k, v in caps.items() if k != "cap_root_voxels"}))   # shadow mesher
    chk = ComposedRunChecker(o3, cfg.mesh_append_budget, _compare_scan)
    for k in range(0, 11):
        Rk, tk = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, Rk, tk, n_pts=100000, extT=extT)
        down = synth.voxel_grid_downsample(raw, 0.4)
        po, ph = (synth.forward_without_imu(so), synth.forward_without_imu(sh)) if k else (so, sh)
        if k == 3:   # one matcher pass at a fixed state: match index sets / normals / residuals at full size
            ro, rh = o.residuals(down, po), h.residuals(down, po)
            assert np.array_equal(rh["match_idx"], ro["match_idx"]) and len(ro["match_idx"]) > 5000
            np.testing.assert_allclose(rh["HTH"], ro["HTH"], rtol=1e-7, atol=1e-6)
            np.testing.assert_allclose(rh["HTz"], ro["HTz"], rtol=1e-7, atol=1e-6)
            np.testing.assert_array_equal(rh["dis"], ro["dis"])
        so, io = o.process_scan(down, raw, po, po, frame_idx=k, do_mesh=True)
        d_down, d_raw = torch.from_numpy(down).cuda(), torch.from_numpy(raw).cuda()
        sh, ih = h.process_scan(d_down.data_ptr(), d_raw.data_ptr(), ph, ph, frame_idx=k, do_mesh=1, n_ds=len(down), n_raw=len(raw))
        assert ih == io, (k, ih, io)
        np.testing.assert_allclose(sh[:24], so[:24], rtol=0, atol=TOL)
        np.testing.assert_allclose(sh[24:], so[24:], rtol=0, atol=1e-9)      # posterior covariance
        mo, mh = o.mesh_fetch(), h.mesh_fetch()
        chk.check_scan(k, o, h, sh, mo, mh, pose_o=so, lever=float(np.abs(raw[:, :3]).max()) + 1.0)   # the composed run, exactly (<= 1 ulp clouds, shadow oracle on the device's cloud, full oracle until a candidate flips)
        # the mesher on the ORACLE's world-frame cloud: every list of every scan
        w = o.mesh_world_scan()
        _compare_scan(o2.mesh_scan(w, so[9:12], frame_idx=k), h2.mesh_scan(w, so[9:12], frame_idx=k), f"scan {k} (identical world-frame input)")
    sm = chk.summary()
    record_property("composed_run", str(sm))
    print(f"[parity] composed run, 11 scans: {sm}")
    assert sm["scans_equal_to_shadow_oracle"] == 11
    co, ch = o.counters(), h.counters()
    for key in ("n_match", "n_plane_tests", "n_extra_probe", "n_refits", "n_refit_pts", "n_root_voxels"):   # (n_iter per scan is compared above; the oracle also counts the stand-alone matcher pass)
        assert ch[key] == co[key], key
    c2o, c2h = o2.counters(), h2.counters()
    for key in ("n_app", "n_new", "v_act", "n_v", "n_u", "t_v", "t_add", "t_rem", "n_vertices", "n_triangles_live"):
        assert c2h[key] == c2o[key], key
    assert c2o["n_vertices"] > 20000
    assert compare_plane_tables_fast(o.dump_planes(), h.dump_planes(), TOL) >= n_planar   # the grown map, again in full


@pytest.mark.timeout(900)
def test_plane_table_parity_on_the_10m_voxel_map(oracle_lib, hip_lib, record_property):
    """The checker itself at the size of BASELINE.json's metric: BOTH sides ingest the bench's survey strips until the map holds
    10 M root voxels (38 s on the GPU box's host in round 3; a time budget stops a slower host earlier and the size reached is recorded); the plane
    tables, the refit counters, one full-size matcher pass and three stream scans are then compared."""
    torch = pytest.importorskip("torch")
    import time
    import bench
    dev = torch.device("cuda", 0)
    n_vox = 10.0e6
    cfg = capi.avia_config(cap_root_voxels=int(n_vox * 1.3) + (1 << 16), cap_scan_points=2_500_000, cap_vertices=1 << 20, cap_triangles=1 << 22)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    side = float(np.sqrt(n_vox / 8.8)) + 40.0
    ident = capi.make_state()
    cap = int(cfg.cap_scan_points)
    t0 = time.time()
    for P in bench.survey_strips(cfg, torch, dev, side):
        for a in range(0, P.shape[0], cap):
            chunk = P[a:a + cap]
            h.map_update(chunk.data_ptr(), ident, n=chunk.shape[0])
            o.map_update(np.ascontiguousarray(chunk.cpu().numpy()), ident)
        nv = h.counters()["n_root_voxels"]
        if nv >= n_vox or time.time() - t0 > 240.0:
            break
    record_property("root_voxels_compared", int(nv))
    print(f"[parity] 10 M map: {nv} root voxels on both sides after {time.time() - t0:.0f} s")
    assert nv >= 2_000_000 and o.counters()["n_root_voxels"] == nv
    co, ch = o.counters(), h.counters()
    assert ch["n_refits"] == co["n_refits"] and ch["n_refit_pts"] == co["n_refit_pts"]
    n_planar = compare_plane_tables_fast(o.dump_planes(), h.dump_planes(), TOL)   # every initialised node of both maps (same node sets is part of the comparison)
    assert n_planar > 0.8 * nv
    record_property("planar_nodes_compared", n_planar)
    extT = np.array(list(cfg.extT))
    so = capi.make_state(R=synth.trajectory_pose(0)[0], t=synth.trajectory_pose(0)[1])
    so[12:15] = [1.0, 0, 0]; so[15:18] = [0, 0, np.deg2rad(2.0)]
    sh = so.copy()
    for k in range(0, 3):
        Rk, tk = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, Rk, tk, n_pts=100000, extT=extT)
        down = synth.voxel_grid_downsample(raw, 0.4)
        po, ph = (synth.forward_without_imu(so), synth.forward_without_imu(sh)) if k else (so, sh)
        if k == 1:
            ro, rh = o.residuals(down, po), h.residuals(down, po)
            assert np.array_equal(rh["match_idx"], ro["match_idx"]) and len(ro["match_idx"]) > 5000
            np.testing.assert_array_equal(rh["dis"], ro["dis"])
        so, io = o.process_scan(down, raw, po, po, frame_idx=k, do_mesh=False)
        sh, ih = h.process_scan(down, raw, ph, ph, frame_idx=k, do_mesh=0)
        assert ih == io, (k, ih, io)
        np.testing.assert_allclose(sh[:24], so[:24], rtol=0, atol=TOL)
    co, ch = o.counters(), h.counters()
    assert ch["n_refits"] == co["n_refits"] and ch["n_refit_pts"] == co["n_refit_pts"]


def test_hdl64_full_width_scans(oracle_lib, hip_lib, record_property):
    """configs[3] at its real width: 64 rings x 2032 azimuth steps, velodyne.yaml (3 m root voxels, max_layer 4, 3 EKF iterations, mesh scale 1.5)"""
    caps = dict(cap_root_voxels=1 << 16, cap_scan_points=400_000, cap_vertices=1 << 21, cap_triangles=1 << 23)
    cfg = capi.velodyne_config(**caps)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    o2, h2 = make_oracle(oracle_lib, capi.velodyne_config(**caps)), make_hip(hip_lib, capi.velodyne_config(**caps))
    R0, t0 = synth.trajectory_pose(0)
    raw0 = synth.hdl64_scan(0, R0, t0, n_az=2032)
    assert len(raw0) > 100_000
    st = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    o.map_build(p0, st); h.map_build(p0, st)
    assert compare_plane_tables_fast(o.dump_planes(), h.dump_planes(), TOL) > 500
    so = st.copy(); so[12:15] = [1.0, 0, 0]; so[15:18] = [0, 0, np.deg2rad(2.0)]
    sh = so.copy()
    chk = ComposedRunChecker(make_oracle(oracle_lib, capi.velodyne_config(**caps)), cfg.mesh_append_budget, _compare_scan)
    for k in range(1, 5):
        Rk, tk = synth.trajectory_pose(k)
        raw = synth.hdl64_scan(k, Rk, tk, n_az=2032)
        down = synth.voxel_grid_downsample(raw, 0.5)
        po, ph = synth.forward_without_imu(so), synth.forward_without_imu(sh)
        if k == 2:
            ro, rh = o.residuals(down, po), h.residuals(down, po)
            assert np.array_equal(rh["match_idx"], ro["match_idx"]) and len(ro["match_idx"]) > 3000
            np.testing.assert_allclose(rh["HTH"], ro["HTH"], rtol=1e-7, atol=1e-6)
        # the FULL pipeline (registration + map growth + meshing in one call, as service_LiDAR_update runs it) ...
        so, io = o.process_scan(down, raw, po, po, frame_idx=k, do_mesh=True)
        sh, ih = h.process_scan(down, raw, ph, ph, frame_idx=k, do_mesh=1)
        assert ih == io, (k, ih, io)
        np.testing.assert_allclose(sh[:24], so[:24], rtol=0, atol=TOL)
        np.testing.assert_allclose(sh[24:], so[24:], rtol=0, atol=1e-9)
        mo, mh = o.mesh_fetch(), h.mesh_fetch()
        chk.check_scan(k, o, h, sh, mo, mh, pose_o=so, lever=float(np.abs(raw[:, :3]).max()) + 1.0)
        # ... and the mesher alone on the oracle's world-frame cloud: every list of every scan
        w = o.mesh_world_scan()
        _compare_scan(o2.mesh_scan(w, so[9:12], frame_idx=k), h2.mesh_scan(w, so[9:12], frame_idx=k), f"hdl64 scan {k}")
    record_property("hdl64_composed_run", str(chk.summary()))
    print(f"[parity] hdl64 composed run, 4 scans: {chk.summary()}")
    assert chk.summary()["scans_equal_to_shadow_oracle"] == 4
    assert compare_plane_tables_fast(o.dump_planes(), h.dump_planes(), TOL) > 500
    assert o2.counters()["n_vertices"] > 5000


def test_hdl64_map_built_from_the_first_50_scans(oracle_lib, hip_lib, record_property, tmp_path):
    """BASELINE configs[3] as SURVEY 8(d) C4 words it: the map -- registration map AND mesh map -- is what the first 50 full-width HDL-64 scans of the
    stream leave behind (scan 0 through map_build, scans 1..49 through the full pipeline, each side on its own poses).  Then: every initialised node of
    the two plane tables, and four more composed scans (poses <= 1e-5, lists against the shadow oracle bit for bit).  This is the map bench.py's
    configs[3] leg times its stream on (velodyne.yaml:40-50: 3 m roots, max_layer 4, 1000 points per node)."""
    import bench                                          # (the scan cache: worker processes ray-cast the 54 scans in parallel)
    caps = dict(cap_root_voxels=1 << 16, cap_scan_points=400_000, cap_vertices=1 << 22, cap_triangles=1 << 24)
    cfg = capi.velodyne_config(**caps)
    raws, downs = bench.make_scans(54, 0, cfg, str(tmp_path / "scans"), kitti=True)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    o.set_threads(12, 4)
    # the registration map's shadow: an oracle map that is grown with the DEVICE'S OWN posteriors (as the shadow mesher is fed the device's own cloud).
    # In a composed run each side keeps its own state; the two agree to ~1e-8, but the world points are stored as f32, so now and then a coordinate
    # rounds the other way (4e-6 m at 50 m) -- over 49 scans that moves an ill-conditioned plane's normal by more than 1e-5 although both sides are
    # right.  The bar "plane normals within 1e-5 on identical inputs" is therefore checked on identical inputs: device map vs shadow map.
    om = make_oracle(oracle_lib, capi.velodyne_config(**caps))
    R0, t0 = synth.trajectory_pose(0)
    st = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raws[0][:, :3])
    o.map_build(p0, st); h.map_build(p0, st); om.map_build(p0, st)
    so = st.copy(); so[12:15] = [1.0, 0, 0]; so[15:18] = [0, 0, np.deg2rad(2.0)]
    sh = so.copy()
    chk = ComposedRunChecker(make_oracle(oracle_lib, capi.velodyne_config(**caps)), cfg.mesh_append_budget, _compare_scan)
    chk.shadow.set_threads(12, 4)
    worst = 0.0
    for k in range(1, 54):
        po, ph = synth.forward_without_imu(so), synth.forward_without_imu(sh)
        so, io = o.process_scan(downs[k], raws[k], po, po, frame_idx=k, do_mesh=True)
        sh, ih = h.process_scan(downs[k], raws[k], ph, ph, frame_idx=k, do_mesh=1)
        assert ih == io, (k, ih, io)
        worst = max(worst, float(np.abs(sh[:24] - so[:24]).max()))
        np.testing.assert_allclose(sh[:24], so[:24], rtol=0, atol=TOL)
        om.map_update(downs[k], sh)                                   # (map_incremental_grow at the device's posterior: pose AND covariance blocks)
        mo, mh = o.mesh_fetch(), h.mesh_fetch()
        chk.check_scan(k, o, h, sh, mo, mh, pose_o=so, lever=float(np.abs(raws[k][:, :3]).max()) + 1.0, pose_gap_bound=1e-6)   # every scan of the build, too: the shadow follows the device (53 composed scans: the two states drift to ~1e-8)
        if k == 49:   # the C4 map
            a, b = om.dump_planes(), h.dump_planes()
            n_pl = compare_plane_tables_fast(a, b, TOL)               # every initialised node: same set, is_plane, update_enable, counts; planes within 1e-5
            co, ch, cm_ = o.counters(), h.counters(), om.counters()
            assert ch["n_root_voxels"] == cm_["n_root_voxels"]
            assert abs(ch["n_root_voxels"] - co["n_root_voxels"]) <= 2 and abs(ch["n_refits"] - co["n_refits"]) <= 0.001 * co["n_refits"]   # the full oracle pipeline: the same map up to the rounding flips
            assert n_pl > 2000 and int(a["layer"].max()) >= 2 and int((a["update_enable"] == 0).sum()) > 0       # deep, with frozen nodes: a map that has lived
            record_property("c4_map", str({"root_voxels": co["n_root_voxels"], "initialised_nodes": len(a), "planar": n_pl, "frozen": int((a["update_enable"] == 0).sum()),
                                            "mesh_vertices": co["n_vertices"], "live_triangles": co["n_triangles_live"]}))
    assert chk.summary()["scans_equal_to_shadow_oracle"] == 53
    assert compare_plane_tables_fast(om.dump_planes(), h.dump_planes(), TOL) > 2000
    print(f"[parity] C4 map from 50 scans + 4 composed scans: worst pose / state difference {worst:.2e}; {chk.summary()['first_divergence']=}")


def test_one_500k_point_scan(oracle_lib, hip_lib):
    """configs[4]'s scan size on one GPU: a 500 000-pt scan registered against the map of a first 500 000-pt scan, then meshed"""
    caps = dict(cap_root_voxels=1 << 18, cap_scan_points=600_000, cap_vertices=1 << 21, cap_triangles=1 << 23)
    cfg = capi.avia_config(**caps)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    extT = np.array(list(cfg.extT))
    R0, t0 = synth.trajectory_pose(0)
    raw0 = synth.livox_scan(0, R0, t0, n_pts=500000, extT=extT)
    st = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    o.map_build(p0, st); h.map_build(p0, st)
    assert compare_plane_tables_fast(o.dump_planes(), h.dump_planes(), TOL) > 2000
    R1, t1 = synth.trajectory_pose(1)
    raw = synth.livox_scan(1, R1, t1, n_pts=500000, extT=extT)
    down = synth.voxel_grid_downsample(raw, 0.4)
    prior = capi.make_state(R=R1, t=t1 + np.array([0.02, -0.01, 0.01]), cov_diag=1e-5)
    ro, rh = o.residuals(down, prior), h.residuals(down, prior)
    assert np.array_equal(rh["match_idx"], ro["match_idx"]) and len(ro["match_idx"]) > 5000
    np.testing.assert_allclose(rh["HTH"], ro["HTH"], rtol=1e-7, atol=1e-6)
    so, io = o.process_scan(down, raw, prior, prior, frame_idx=1, do_mesh=True)
    sh, ih = h.process_scan(down, raw, prior, prior, frame_idx=1, do_mesh=1)
    assert ih == io
    np.testing.assert_allclose(sh[:24], so[:24], rtol=0, atol=TOL)
    # the mesher on the identical 500k-pt world-frame scan (step = round(500000 / 10000) = 50 -> 10 000 candidates), two scans so that the second diffs
    o2, h2 = make_oracle(oracle_lib, capi.avia_config(**caps)), make_hip(hip_lib, capi.avia_config(**caps))
    for k, (r_, s_) in enumerate(((raw0, st), (raw, so))):
        w = _world(r_, s_, cfg)
        _compare_scan(o2.mesh_scan(w, s_[9:12], frame_idx=k), h2.mesh_scan(w, s_[9:12], frame_idx=k), f"500k scan {k}")
    assert compare_plane_tables_fast(o.dump_planes(), h.dump_planes(), TOL) > 2000
