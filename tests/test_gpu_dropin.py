"""The compiled drop-in (drop_in/immesh_shim.cpp: the replaced bodies of Voxel_mapping::voxel_map_init / lio_state_estimation / map_incremental_grow,
incremental_mesh_reconstruction and the host-mirror update, behind the reference's own signatures) driven the way service_LiDAR_update drives
them (drop_in/shim_main.cpp) and checked against the CPU oracle: pose, m_effct_feat_num, Global_map size and the live set of the Triangle_manager
mirror (order-independent hash over (triplet, m_index_flip))."""
import os
import struct
import subprocess

import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle, make_hip, ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
M64 = (1 << 64) - 1


def _live_hash(live):
    h = 0
    for (a, b, c), f in live.items():
        x = ((a * 0x9E3779B97F4A7C15) & M64) ^ ((b << 21) & M64) ^ ((c << 42) & M64) ^ (f & 1)
        x ^= x >> 30; x = (x * 0xbf58476d1ce4e5b9) & M64; x ^= x >> 27; x = (x * 0x94d049bb133111eb) & M64; x ^= x >> 31
        h = (h + x) & M64
    return h


def _world_like_the_shim(raw, st, cfg):
    """transformLidar as drop_in/immesh_shim.cpp::map_incremental_grow writes it (same operation order, f64 -> f32)"""
    R = np.asarray(st[0:9]).reshape(3, 3); t = np.asarray(st[9:12])
    eR = np.array(list(cfg.extR)).reshape(3, 3); eT = np.array(list(cfg.extT))
    p = raw[:, :3].astype(np.float64)
    b = [eR[r, 0] * p[:, 0] + eR[r, 1] * p[:, 1] + eR[r, 2] * p[:, 2] + eT[r] for r in range(3)]
    out = raw.copy()
    for r in range(3):
        out[:, r] = (R[r, 0] * b[0] + R[r, 1] * b[1] + R[r, 2] * b[2] + t[r]).astype(np.float32)
    return np.ascontiguousarray(out)


def test_drop_in_shim_matches_the_oracle(oracle_lib, hip_lib, tmp_path):
    run_drop_in(oracle_lib, lambda cfg: make_hip(hip_lib, cfg), "shim_main", tmp_path)


def run_drop_in(oracle_lib, make_direct, target, tmp_path, expect_ply=True):
    exe = os.path.join(ROOT, "drop_in", target)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "drop_in"), target])
    cfg = capi.avia_config()
    extT = np.array(list(cfg.extT))
    o = make_oracle(oracle_lib, cfg)
    n_scans = 5
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    scans = []
    with open(fin, "wb") as f:
        f.write(struct.pack("<i", n_scans))
        for k in range(n_scans):
            R, t = synth.trajectory_pose(k)
            raw = synth.livox_scan(k, R, t, n_pts=30000, extT=extT)
            down = synth.voxel_grid_downsample(raw, 0.4)
            prior = capi.make_state(R=R, t=t + (np.array([0.01, -0.01, 0.005]) if k else 0.0), cov_diag=1e-5)
            scans.append((raw, down, prior))
            f.write(struct.pack("<ii", len(raw), len(down))); f.write(prior.astype("<f8").tobytes()); f.write(raw.astype("<f4").tobytes()); f.write(down.astype("<f4").tobytes())
    r = subprocess.run([exe, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    rec = np.dtype([("state", "<f8", 348), ("eff", "<i4"), ("nv", "<i4"), ("nl", "<i4"), ("hash", "<u8")])
    got = np.fromfile(fout, dtype=rec)
    assert len(got) == n_scans
    # (1) bit for bit against the same C-ABI calls issued directly (the shim adds marshalling and the host mirrors, nothing else)
    h = make_direct(cfg)
    shadow = make_oracle(oracle_lib, capi.avia_config())   # the oracle's mesher re-based on the shim's own world-frame clouds: exact for every scan (parity_utils.ComposedRunChecker's part 2)
    hlive, slive, worlds = {}, {}, {}
    for k, (raw, down, prior) in enumerate(scans):
        if k == 0:
            h.map_build(np.ascontiguousarray(raw[:, :3]), prior)
            continue
        sh, ih = h.register(down, prior, prior)
        h.map_update(down, sh)
        worlds[k] = _world_like_the_shim(raw, sh, cfg)
        m = h.mesh_scan(worlds[k], sh[9:12], frame_idx=k - 1)
        ms = shadow.mesh_scan(worlds[k], sh[9:12], frame_idx=k - 1)
        _apply(slive, ms)
        for tri in map(tuple, m["tri_rem"].tolist()):
            hlive.pop(tri, None)
        for tri, fl in zip(map(tuple, m["tri_add"].tolist()), m["flip_add"].tolist()):
            hlive[tri] = fl
        for tri, fl in zip(map(tuple, m["tri_upd"].tolist()), m["flip_upd"].tolist()):
            if tri in hlive:          # (a flip update may name a triangle another voxel removed in the same scan: it stays removed)
                hlive[tri] = fl
        np.testing.assert_array_equal(got[k]["state"], sh)
        assert got[k]["eff"] == ih["n_match"] and got[k]["nv"] == m["vtx_base"] + len(m["new_vtx"])
        assert got[k]["nl"] == len(hlive) and int(got[k]["hash"]) == _live_hash(hlive)
        # ... and against the shadow oracle, exactly, every scan (no "<= 5 vertices / <= 1 %" any more)
        assert got[k]["nv"] == ms["vtx_base"] + len(ms["new_vtx"]) and got[k]["nl"] == len(slive) and int(got[k]["hash"]) == _live_hash(slive), k
    h.close()
    # (2) against the oracle
    live, in_sync = {}, True
    for k, (raw, down, prior) in enumerate(scans):
        if k == 0:
            o.map_build(np.ascontiguousarray(raw[:, :3]), prior)
            assert got[0]["nv"] == 0 and got[0]["nl"] == 0
            continue
        so, io = o.process_scan(down, raw, prior, prior, frame_idx=k - 1, do_mesh=True)
        np.testing.assert_allclose(got[k]["state"][:24], so[:24], rtol=0, atol=1e-5)
        np.testing.assert_allclose(got[k]["state"][24:], so[24:], rtol=0, atol=1e-9)
        assert got[k]["eff"] == io["n_match"]
        m = o.mesh_fetch()
        for tri in map(tuple, m["tri_rem"].tolist()):
            live.pop(tri, None)
        for tri, fl in zip(map(tuple, m["tri_add"].tolist()), m["flip_add"].tolist()):
            live[tri] = fl
        for tri, fl in zip(map(tuple, m["tri_upd"].tolist()), m["flip_upd"].tolist()):
            if tri in live:
                live[tri] = fl
        nv = m["vtx_base"] + len(m["new_vtx"])
        # the FULL oracle pipeline: exact as long as no mesher candidate of the two world-frame clouds has rounded the other way (poses agree to ~1e-12,
        # the clouds are f32); from the first flip on only the shadow comparison above is meaningful
        wo = o.mesh_world_scan()
        step = max(1, int(round(len(wo) // cfg.mesh_append_budget)))
        in_sync = in_sync and np.array_equal(wo[::step, :3], worlds[k][::step, :3])
        if in_sync:
            assert got[k]["nv"] == nv and got[k]["nl"] == len(live) and int(got[k]["hash"]) == _live_hash(live), k
    assert got[1]["nv"] > 500 and got[-1]["nl"] > 3000
    if expect_ply:
        assert os.path.getsize("/tmp/immesh_dropin_test.ply") > 10000       # save_to_ply_file through the shim


# ---- the SERVICE-LEVEL (asynchronous) drop-in: drop_in/immesh_shim_async.cpp behind drop_in/dropin_driver.cpp (two threads, as the reference runs them) ----------
def _dropin_lib(name):
    import ctypes as C
    if name.startswith("_ref/"):   # the variants with the reference's own Triangle_manager as the host mirror: built only where /root/reference exists
        if os.path.exists("/root/reference/src/meshing/r3live/triangle.hpp"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "drop_in"), "refmirror"])
        elif not os.path.exists(os.path.join(ROOT, "drop_in", name)):
            pytest.skip(f"drop_in/{name} not built and /root/reference absent")
    else:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "drop_in"), name])
    lib = C.CDLL(os.path.join(ROOT, "drop_in", name))
    lib.dropin_create.restype = C.c_void_p; lib.dropin_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.dropin_destroy.argtypes = [C.c_void_p, C.c_int]
    lib.dropin_first_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.dropin_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dropin_wait_meshed.argtypes = [C.c_void_p, C.c_long, C.c_int]
    lib.dropin_frame_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dropin_effect_features.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.dropin_run_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def _apply(live, m):
    for tri in map(tuple, m["tri_rem"].tolist()):
        live.pop(tri, None)
    for tri, fl in zip(map(tuple, m["tri_add"].tolist()), m["flip_add"].tolist()):
        live[tri] = fl
    for tri, fl in zip(map(tuple, m["tri_upd"].tolist()), m["flip_upd"].tolist()):
        if tri in live:
            live[tri] = fl


def run_drop_in_async(make_direct, libname, lockstep, shadow=None, queue_depth=None):
    """Scans through the asynchronous shim (scan thread = this thread, service thread inside the driver) vs the SAME C-ABI calls issued directly:
    immesh_process_scan(IMMESH_MESH_ASYNC) + immesh_mesh_wait + immesh_mesh_fetch.  States, m_effct_feat_num and, per frame, the Global_map size and the
    live set of the Triangle_manager mirror (order-independent hash over (triplet, m_index_flip)) must be equal bit for bit."""
    import ctypes as C
    lib = _dropin_lib(libname)
    depth = C.c_int.in_dll(lib, "g_immesh_mirror_queue_depth")   # frames between the service thread and the mirror thread (read when the service starts)
    depth_default = depth.value
    if queue_depth is not None:
        depth.value = queue_depth
    cfg = capi.avia_config()
    extT = np.array(list(cfg.extT))
    n_scans = 7
    scans = []
    for k in range(n_scans):
        R, t = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, R, t, n_pts=30000, extT=extT)
        down = synth.voxel_grid_downsample(raw, 0.4)
        prior = capi.make_state(R=R, t=t + (np.array([0.01, -0.01, 0.005]) if k else 0.0), cov_diag=1e-5)
        scans.append((np.ascontiguousarray(raw), np.ascontiguousarray(down), prior))
    d = lib.dropin_create(None, extT.ctypes.data_as(C.c_void_p), 1)
    assert d
    got = []
    try:
        assert lib.dropin_first_scan(d, scans[0][0].ctypes.data_as(C.c_void_p), len(scans[0][0]), scans[0][2].ctypes.data_as(C.c_void_p)) == 0
        for k in range(1, n_scans):
            raw, down, prior = scans[k]
            st, eff = np.zeros(348), C.c_int32(0)
            assert lib.dropin_scan(d, raw.ctypes.data_as(C.c_void_p), len(raw), down.ctypes.data_as(C.c_void_p), len(down), prior.ctypes.data_as(C.c_void_p),
                                   st.ctypes.data_as(C.c_void_p), C.byref(eff)) == 0
            if lockstep:
                assert lib.dropin_wait_meshed(d, k, 60000) == 0
            got.append((st, eff.value))
        assert lib.dropin_wait_meshed(d, n_scans - 1, 60000) == 0
        stats = []
        for f in range(n_scans - 1):
            nv, nl, hh = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
            assert lib.dropin_frame_stats(d, f, C.byref(nv), C.byref(nl), C.byref(hh)) == 0
            stats.append((nv.value, nl.value, hh.value))
        n_last = len(scans[-1][1])
        ep, en = np.zeros((n_last, 3), np.float32), np.zeros((n_last, 4), np.float32)
        n_eff = lib.dropin_effect_features(d, ep.ctypes.data_as(C.c_void_p), en.ctypes.data_as(C.c_void_p), n_last)
        # a simulated renderer pass over the mirror (unparse_triangle_set_to_vector, mesh_rec_display.cpp:78-103): every unsmoothed vertex of a live triangle
        # through the shim's Global_map::smooth_pts (one vertex per call, as the unchanged renderer would) -- no NaN in the GL buffer (the reference's body would
        # search a host ikd-Tree the drop-in never feeds), and the buffer equals what ONE immesh_mesh_display_vertices call returns
        lib.dropin_render_pass.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int64]
        rp = np.zeros(4, np.int64)
        assert lib.dropin_render_pass(d, 1.0, 20.0, 1.25 * cfg.mesh_voxel, rp.ctypes.data_as(C.c_void_p), None, 0) == 0
        assert rp[0] > 3000 and rp[1] > 0 and rp[2] == 0 and rp[3] == 0, rp
        rp2 = np.zeros(4, np.int64)   # a second pass finds every vertex smoothed (m_smoothed is set by the first, as in the reference) and the same buffer
        assert lib.dropin_render_pass(d, 1.0, 20.0, 1.25 * cfg.mesh_voxel, rp2.ctypes.data_as(C.c_void_p), None, 0) == 0
        assert rp2[0] == rp[0] and rp2[1] == 0 and rp2[2] == 0, rp2
    finally:
        lib.dropin_destroy(d, 1)
        depth.value = depth_default
    h = make_direct(cfg)
    live, slive = {}, {}
    h.map_build(np.ascontiguousarray(scans[0][0][:, :3]), scans[0][2])
    for k in range(1, n_scans):
        raw, down, prior = scans[k]
        sh, ih = h.process_scan(down, raw, prior, prior, frame_idx=k - 1, do_mesh=2)
        h.mesh_wait()
        m = h.mesh_fetch()
        _apply(live, m)
        np.testing.assert_array_equal(got[k - 1][0], sh)
        assert got[k - 1][1] == ih["n_match"]
        nv, nl, hh = stats[k - 1]
        assert nv == m["vtx_base"] + len(m["new_vtx"]) and nl == len(live) and hh == _live_hash(live), k
        if shadow is not None:   # the direct calls' lists are the oracle's on the device's own world-frame cloud (composed-run check): so is the mirror
            ms = shadow.mesh_scan(h.mesh_world_scan(), sh[9:12], frame_idx=k - 1)
            _apply(slive, ms)
            assert nl == len(slive) and hh == _live_hash(slive), k
    p2, q2 = h.last_matches()
    assert n_eff == len(p2) and np.array_equal(ep[:n_eff], p2) and np.array_equal(en[:n_eff], q2)
    assert stats[0][0] > 500 and stats[-1][1] > 3000
    h.close()


def test_async_drop_in_matches_direct_calls_and_the_oracle(oracle_lib, hip_lib):
    run_drop_in_async(lambda cfg: make_hip(hip_lib, cfg), "libimmesh_dropin_async.so", lockstep=False, shadow=make_oracle(oracle_lib, capi.avia_config()))


def test_async_drop_in_with_the_reference_triangle_manager_as_the_mirror(oracle_lib, hip_lib):
    """The same asynchronous shim compiled against THE REFERENCE'S OWN Triangle_manager (triangle.hpp / triangle.cpp / tools_kd_hash.hpp from where they lie,
    drop_in/Makefile `refmirror`): per frame the real manager's live set -- walked through its region buckets, m_triangle_set_vector -- and its flips hash
    to the direct calls' and the shadow oracle's.  This is the mirror bench.py's drop-in leg pays for."""
    run_drop_in_async(lambda cfg: make_hip(hip_lib, cfg), "_ref/libimmesh_dropin_async_refmirror.so", lockstep=False, shadow=make_oracle(oracle_lib, capi.avia_config()))


@pytest.mark.parametrize("queue_depth", [1, 0])
def test_async_drop_in_with_a_full_mirror_queue_and_without_a_mirror_thread(hip_lib, queue_depth):
    """The host queue between the service thread and the mirror thread one frame deep (the service thread blocks on a full queue -- the scan thread is
    not in lock-step here, so it does fill) and depth 0 (no mirror thread: the service thread applies the lists itself, round 4's arrangement): nothing
    is dropped or reordered, every frame's mirror state equals the direct calls'."""
    run_drop_in_async(lambda cfg: make_hip(hip_lib, cfg), "libimmesh_dropin_async.so", lockstep=False, queue_depth=queue_depth)


def test_async_drop_in_stream_runner(hip_lib):
    """dropin_run_stream (what bench.py times): priors by Forward_without_imu, every frame's lists fetched and applied -- equal to the direct calls"""
    import ctypes as C
    lib = _dropin_lib("libimmesh_dropin_async.so")
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    extT = np.array(list(cfg.extT))
    scans = []
    for k in range(6):
        R, t = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, R, t, n_pts=30000, extT=extT)
        scans.append((np.ascontiguousarray(raw), np.ascontiguousarray(synth.voxel_grid_downsample(raw, 0.4))))
    R0, t0 = synth.trajectory_pose(0)
    st0 = capi.make_state(R=R0, t=t0)
    results = []
    for via_shim in (True, False):
        h = make_hip(hip_lib, cfg)
        h.map_build(np.ascontiguousarray(scans[0][0][:, :3]), st0)
        st = st0.copy(); st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
        if via_shim:
            d = lib.dropin_create(h.ctx, extT.ctypes.data_as(C.c_void_p), 0)     # the driver adopts the context (as bench.py hands over the 10 M-voxel map)
            assert d
            n = len(scans) - 1
            raws = (C.c_void_p * n)(*[s[0].ctypes.data for s in scans[1:]]); downs = (C.c_void_p * n)(*[s[1].ctypes.data for s in scans[1:]])
            n_raw = np.array([len(s[0]) for s in scans[1:]], np.int32); n_ds = np.array([len(s[1]) for s in scans[1:]], np.int32)
            states, eff, ms = np.zeros((n, 348)), np.zeros(n, np.int32), np.zeros(2)
            rc = lib.dropin_run_stream(d, n, raws, n_raw.ctypes.data_as(C.c_void_p), downs, n_ds.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), 0.1, 0.3, 0.5,
                                       states.ctypes.data_as(C.c_void_p), eff.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p))
            lib.dropin_destroy(d, 0)
            assert rc == 0 and ms[1] >= ms[0] > 0
            results.append((states, h.counters()))
        else:
            out = []
            for raw, down in scans[1:]:
                prior = capi.forward_without_imu_native(hip_lib, st)
                st, _ = h.process_scan(down, raw, prior, prior, frame_idx=0, do_mesh=2)
                out.append(st.copy())
            h.mesh_wait()
            results.append((np.array(out), h.counters()))
        h.close()
    np.testing.assert_array_equal(results[0][0], results[1][0])
    for key in ("n_vertices", "n_triangles_live", "t_add", "t_rem", "n_match", "n_refits"):
        assert results[0][1][key] == results[1][1][key], key
