"""ImuProcess::UndistortPcl (src/IMU_Processing.cpp:755-958).
CPU part: closed-form known answers for the oracle restatement.  GPU part: the HIP path (host IMU propagation + device per-point
compensation) against the oracle through the C ABI."""
import numpy as np
import pytest

from immesh_amd import capi
from conftest import make_oracle, make_hip


def _package(n=2000, t_scan=0.1, seed=0, n_imu=20, gyr=(0, 0, 0), acc=(0, 0, 9.81), noise=0.0):
    rng = np.random.default_rng(seed)
    pts = np.zeros((n, 5), np.float32)
    pts[:, :3] = rng.uniform(-30, 30, (n, 3))
    pts[:, 3] = rng.uniform(0, 255, n)
    pts[:, 4] = rng.permutation(np.linspace(0.0, t_scan * 1000.0, n)).astype(np.float32)   # arrival order != time order
    pts[-1, 4] = t_scan * 1000.0                                                             # the package ends with its latest point
    imu = np.zeros((n_imu, 7))
    imu[:, 0] = np.linspace(t_scan / n_imu, t_scan, n_imu)
    imu[:, 1:4] = np.asarray(gyr) + noise * rng.normal(size=(n_imu, 3))
    imu[:, 4:7] = np.asarray(acc) + 10 * noise * rng.normal(size=(n_imu, 3))
    return pts, imu


def _state(vel=(0, 0, 0)):
    st = capi.make_state(cov_diag=1e-4)
    st[12:15] = vel
    st[21:24] = [0, 0, -9.81]
    return st


def _cfg_identity():
    c = capi.avia_config(cap_root_voxels=1 << 10, cap_scan_points=200000, cap_vertices=1 << 12, cap_triangles=1 << 14)
    for i in range(3):
        c.extT[i] = 0.0
    return c


def test_static_sensor_leaves_points_alone(oracle_lib):
    cfg = _cfg_identity()
    o = make_oracle(oracle_lib, cfg)
    pts, imu = _package()
    ic = capi.make_imu_ctx(cfg)
    out, st, lut = o.undistort(pts, imu, 0.0, 0.0, ic, _state())
    order = np.argsort(pts[:, 4], kind="stable")
    np.testing.assert_array_equal(out[:, 3], pts[order, 3])                # sorted by offset time, other fields carried along
    np.testing.assert_allclose(out[:, :3], pts[order, :3], atol=1e-5)
    np.testing.assert_allclose(st[:21], _state()[:21], atol=1e-12)         # no motion
    assert lut == pytest.approx(0.1) and ic.last_lidar_end_time == pytest.approx(0.1) and ic.last_imu.t == pytest.approx(0.1)
    cov = st[24:].reshape(18, 18)
    assert np.all(np.diag(cov)[:3] > 1e-4) and np.allclose(cov, cov.T, atol=1e-15)   # process noise accumulated, still symmetric


def test_constant_yaw_rate_closed_form(oracle_lib):
    cfg = _cfg_identity()
    o = make_oracle(oracle_lib, cfg)
    w, T = 0.8, 0.1
    pts, imu = _package(gyr=(0, 0, w))
    ic = capi.make_imu_ctx(cfg, gyr0=(0, 0, w))
    ic.angvel_last[2] = w
    out, st, _ = o.undistort(pts, imu, 0.0, 0.0, ic, _state())
    order = np.argsort(pts[:, 4], kind="stable")
    t = pts[order, 4].astype(np.float64) / 1000.0
    ang = w * (t - T)                                                      # P_end = Rz(w (t - T)) P
    ang[0] = ang[0] * 1.0                                                  # (t[0] == 0: never compensated, see below)
    P = pts[order, :3].astype(np.float64)
    exp = np.stack([np.cos(ang) * P[:, 0] - np.sin(ang) * P[:, 1], np.sin(ang) * P[:, 0] + np.cos(ang) * P[:, 1], P[:, 2]], axis=1)
    assert t[0] == 0.0
    np.testing.assert_allclose(out[1:, :3], exp[1:], atol=2e-5)
    np.testing.assert_array_equal(out[0, :3], pts[order[0], :3])          # curvature 0 is not > the first pose's offset 0.0: left as it is
    c, s = np.cos(w * T), np.sin(w * T)
    np.testing.assert_allclose(st[:9].reshape(3, 3), [[c, -s, 0], [s, c, 0], [0, 0, 1]], atol=1e-12)


def test_constant_velocity_and_the_earliest_point_quirk(oracle_lib):
    cfg = _cfg_identity()
    o = make_oracle(oracle_lib, cfg)
    v, T = np.array([2.0, -1.0, 0.5]), 0.1
    pts, imu = _package()
    pts[:, 4] = (37.0 + pts[:, 4] * 0.63).astype(np.float32)               # stamps 37 .. 100 ms: the earliest one lies in the 8th of 20 IMU intervals
    ic = capi.make_imu_ctx(cfg)
    out, st, _ = o.undistort(pts, imu, 0.0, 0.0, ic, _state(vel=v))
    order = np.argsort(pts[:, 4], kind="stable")
    t = pts[order, 4].astype(np.float64) / 1000.0
    exp = pts[order, :3].astype(np.float64) + np.outer(t - T, v)           # R = I: P_end = P + v (t - T)
    np.testing.assert_allclose(out[1:, :3], exp[1:], atol=2e-5)
    # the reference's backward loop re-enters with it_pcl == begin for every earlier IMU interval: the earliest point is shifted once per interval
    n_applied = int(np.floor(t[0] / (T / 20))) + 1
    assert n_applied == 8
    np.testing.assert_allclose(out[0, :3], pts[order[0], :3].astype(np.float64) + n_applied * (t[0] - T) * v, atol=5e-5)
    np.testing.assert_allclose(st[9:12], v * T, atol=1e-12)


@pytest.mark.gpu
def test_hip_undistort_matches_oracle(oracle_lib, hip_lib):
    cfg = capi.avia_config(cap_root_voxels=1 << 10, cap_scan_points=200000, cap_vertices=1 << 12, cap_triangles=1 << 14)
    o, h = make_oracle(oracle_lib, cfg), make_hip(hip_lib, cfg)
    st_o = st_h = _state(vel=(1.5, 0.2, -0.1))
    ic_o, ic_h = capi.make_imu_ctx(cfg), capi.make_imu_ctx(cfg)
    lut_o = lut_h = 0.0
    for k in range(3):                                                      # three consecutive packages: the carried members matter
        pts, imu = _package(n=100000, seed=k, gyr=(0.1, -0.2, 0.6), acc=(0.3, -0.2, 9.9), noise=0.02)
        imu[:, 0] += 0.1 * k
        out_o, st_o, lut_o = o.undistort(pts, imu, 0.1 * k, lut_o, ic_o, st_o)
        out_h, st_h, lut_h = h.undistort(pts, imu, 0.1 * k, lut_h, ic_h, st_h)
        np.testing.assert_array_equal(out_h[:, 3], out_o[:, 3])            # identical time order
        np.testing.assert_allclose(out_h[:, :3], out_o[:, :3], atol=2e-5, rtol=0)   # f64 math, f32 store: one float ulp at 50 m is 4e-6
        assert np.mean(out_h[:, :3] == out_o[:, :3]) > 0.99
        np.testing.assert_allclose(st_h, st_o, rtol=1e-11, atol=1e-14)
        assert lut_h == lut_o
        for f in ("last_lidar_end_time", "mean_acc_norm"):
            assert getattr(ic_h, f) == getattr(ic_o, f)
        np.testing.assert_allclose(list(ic_h.acc_s_last) + list(ic_h.angvel_last), list(ic_o.acc_s_last) + list(ic_o.angvel_last), rtol=1e-12, atol=1e-15)
        assert np.abs(out_o[:, :3] - pts[np.argsort(pts[:, 4], kind="stable"), :3]).max() > 0.05   # the motion really moved points


@pytest.mark.gpu
def test_undistorted_cloud_feeds_the_path_on_the_device(hip_lib):
    """raw package -> immesh_undistort -> immesh_downsample -> immesh_map_build / immesh_register, all on device pointers."""
    from immesh_amd import synth
    cfg = capi.avia_config(cap_root_voxels=1 << 14, cap_scan_points=200000, cap_vertices=1 << 14, cap_triangles=1 << 16)
    h = make_hip(hip_lib, cfg)
    extT = np.array(list(cfg.extT))
    R, t = synth.trajectory_pose(0)
    raw = synth.livox_scan(0, R, t, n_pts=40000, extT=extT)
    pts = np.concatenate([raw, np.linspace(0, 100.0, len(raw), dtype=np.float32)[:, None]], axis=1)
    _, imu = _package(n=8)
    ic = capi.make_imu_ctx(cfg)
    st0 = capi.make_state(R=R, t=t); st0[21:24] = [0, 0, -9.81]
    _, st, _ = h.undistort(pts, imu, 0.0, 0.0, ic, st0, to_host=False)
    und = h.undistort_result_ptr()
    np.testing.assert_allclose(st[:12], st0[:12], atol=1e-9)               # static sensor: the state stays
    import ctypes as C
    _, n_dense = h.downsample(und, 0.02, n=len(pts), stride=4, to_host=False)     # (xyz view of nearly the whole cloud for the first-scan map)
    f = h._f("map_build"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]; f.restype = C.c_int
    assert f(h.ctx, C.c_void_p(h.downsample_result_ptr()), n_dense, st.ctypes.data_as(C.c_void_p)) == 0
    _, n_ds = h.downsample(und, 0.4, n=len(pts), stride=4, to_host=False)
    assert 2000 < n_ds < n_dense <= len(pts)
    out, info = h.register(h.downsample_result_ptr(), st, st, n=n_ds)
    assert h.counters()["n_root_voxels"] > 500 and info["n_match"] > 1000
    assert np.linalg.norm(out[9:12] - t) < 0.02
