"""The one-launch registration (residual_persistent_kernel) is a RESIDENT grid: its workgroups gather each other's partial sums, so all of them have
to be on the device at once.  VERDICT r03 / ADVICE r03: the gather was unbounded and the grid was not capped against other contexts.  Now
  * the grid is capped at half of the workgroups the device holds resident (occupancy query at create): two contexts always fit;
  * the gather is bounded; a grid that gives up writes nothing, and the host registers the scan with the per-pass launch chain instead.
Both are exercised here on one GPU: two contexts registering >= 40 000-pt clouds at the same time from two threads, and the fallback itself through the
test hook IMMESH_RP_FORCE_ABORT (every gather gives up at once, as if the grid had not become resident)."""
import os
import threading

import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_hip
from parity_utils import compare_plane_tables_fast, clouds_within_rounding

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _stream(n_scans, n_pts):
    cfg = capi.avia_config()
    extT = np.array(list(cfg.extT))
    out = []
    for k in range(n_scans):
        R, t = synth.trajectory_pose(k)
        out.append((R, t, synth.livox_scan(k, R, t, n_pts=n_pts, extT=extT)))
    return out


def test_two_contexts_register_large_scans_concurrently(hip_lib):
    """129 + 129 workgroups of one CU each used to be launched by two contexts with >= 32 k-pt clouds (> 256 CUs: each could be partially placed and wait
    for the other forever).  Two threads, two contexts, 45 000-pt clouds, started together scan by scan; each must reproduce, bit for bit, what a
    third context computes alone -- and neither may have needed the fallback."""
    scans = _stream(6, 120000)
    caps = dict(cap_root_voxels=1 << 17, cap_scan_points=200000, cap_vertices=1 << 16, cap_triangles=1 << 18)
    n_big = 45000

    def run(h, out, barrier=None):
        R0, t0, raw0 = scans[0]
        st = capi.make_state(R=R0, t=t0)
        h.map_build(np.ascontiguousarray(raw0[:, :3]), st)
        st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
        for k in range(1, len(scans)):
            raw = scans[k][2]
            big = np.ascontiguousarray(raw[:n_big, :3])      # a >= 40 000-pt cloud in the role of the down-sampled scan (127 + 1 workgroups)
            prior = synth.forward_without_imu(st)
            if barrier is not None:
                barrier.wait()
            st, info = h.process_scan(big, raw, prior, prior, frame_idx=k, do_mesh=0)
            out.append((st.copy(), info))
        out.append(h.registration_fallbacks())

    ref_out, a_out, b_out = [], [], []
    run(make_hip(hip_lib, capi.avia_config(**caps)), ref_out)
    ha, hb = make_hip(hip_lib, capi.avia_config(**caps)), make_hip(hip_lib, capi.avia_config(**caps))
    bar = threading.Barrier(2)
    errs = []

    def guarded(h, out):
        try:
            run(h, out, bar)
        except Exception as e:   # noqa: BLE001 -- reported below; a failing thread must not leave the other at the barrier
            errs.append(e)
            bar.abort()

    ta, tb = threading.Thread(target=guarded, args=(ha, a_out)), threading.Thread(target=guarded, args=(hb, b_out))
    ta.start(); tb.start()
    ta.join(300); tb.join(300)
    assert not ta.is_alive() and not tb.is_alive(), "two contexts registering concurrently did not finish"
    assert not errs, errs
    assert ref_out[-1] == 0 and a_out[-1] == 0 and b_out[-1] == 0       # nobody needed the fallback
    assert len(a_out) == len(ref_out) == len(b_out)
    for (sr, ir), (sa, ia), (sb, ib) in zip(ref_out[:-1], a_out[:-1], b_out[:-1]):
        assert ir == ia == ib and ir["n_match"] > 5000
        assert np.array_equal(sa, sr) and np.array_equal(sb, sr)           # the resident grid is deterministic: same grid, same bits
    assert compare_plane_tables_fast(ha.dump_planes(), hb.dump_planes(), 0.0) > 1000




def test_a_grid_that_gives_up_falls_back_to_the_per_pass_chain(hip_lib):
    """IMMESH_RP_FORCE_ABORT: every workgroup of every resident-grid registration gives up in its first gather.  The scan must then be registered by the
    per-pass chain (residual_kernel -> ekf_step_kernel) and its map update prepared by point_var_kernel: same poses to rounding (the two updates group
    the same algebra differently), same match counts, same map, and a mesh that is exactly the mesh of the world-frame cloud it produced."""
    scans = _stream(5, 40000)
    caps = dict(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    ref, shadow = make_hip(hip_lib, capi.avia_config(**caps)), make_hip(hip_lib, capi.avia_config(**caps))
    os.environ["IMMESH_RP_FORCE_ABORT"] = "1"
    try:
        h = make_hip(hip_lib, capi.avia_config(**caps))
    finally:
        del os.environ["IMMESH_RP_FORCE_ABORT"]
    R0, t0, raw0 = scans[0]
    st = capi.make_state(R=R0, t=t0)
    p0 = np.ascontiguousarray(raw0[:, :3])
    h.map_build(p0, st); ref.map_build(p0, st)
    st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
    sr = st.copy()
    for k in range(1, 5):
        raw = scans[k][2]
        down = synth.voxel_grid_downsample(raw, 0.4)
        prior, pr = synth.forward_without_imu(st), synth.forward_without_imu(sr)
        mode = 1 if k < 4 else 2           # the asynchronous call takes the same fallback
        st, info = h.process_scan(down, raw, prior, prior, frame_idx=k, do_mesh=mode)
        sr, ir = ref.process_scan(down, raw, pr, pr, frame_idx=k, do_mesh=1)
        if mode == 2:
            h.mesh_wait()
        assert info == ir and info["n_match"] > 500
        np.testing.assert_allclose(st[:24], sr[:24], rtol=0, atol=1e-9)
        np.testing.assert_allclose(st[24:], sr[24:], rtol=0, atol=1e-12)
        wh, wr = h.mesh_world_scan(), ref.mesh_world_scan()
        assert clouds_within_rounding(wh[:, :3], wr[:, :3])
        mh, ms = h.mesh_fetch(), shadow.mesh_scan(wh, st[9:12], frame_idx=k)
        for key in ("new_vtx", "tri_add", "tri_rem", "tri_upd", "flip_add", "flip_upd", "smooth_ids"):
            np.testing.assert_array_equal(mh[key], ms[key], err_msg=f"scan {k} {key}")
        assert len(mh["tri_add"]) > 100
    # immesh_register alone takes the fallback too
    down = synth.voxel_grid_downsample(scans[4][2], 0.4)
    prior = synth.forward_without_imu(st)
    s1, i1 = h.register(down, prior, prior)
    s2, i2 = ref.register(down, prior, prior)
    assert i1["n_iter"] == i2["n_iter"] and i1["n_match"] == i2["n_match"]
    np.testing.assert_allclose(s1[:24], s2[:24], rtol=0, atol=1e-9)
    assert h.registration_fallbacks() == 5 and ref.registration_fallbacks() == 0
    assert compare_plane_tables_fast(ref.dump_planes(), h.dump_planes(), 1e-7) > 500
    hc, rc = h.counters(), ref.counters()
    for key in ("n_refits", "n_refit_pts", "n_root_voxels", "n_match", "n_plane_tests"):
        assert hc[key] == rc[key], key
