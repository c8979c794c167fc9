"""Sensor decode: Preprocess::avia_handler (feature extraction off) and Preprocess::velodyne_handler
(src/preprocess.cpp:139-232, 497-526) on the wire formats (livox_ros_driver/CustomMsg points, sensor_msgs/PointCloud2 data)."""
import numpy as np
import pytest

from immesh_amd import capi
from conftest import make_oracle, make_hip

LIVOX = np.dtype([("offset_time", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("reflectivity", "u1"), ("tag", "u1"), ("line", "u1")])   # 19 bytes, packed
VELO = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4"), ("intensity", "<f4"), ("time", "<f4"), ("ring", "<u2"), ("pad2", "V6")])  # 32 bytes (PCL layout)


def _cfg():
    return capi.avia_config(cap_root_voxels=1 << 10, cap_scan_points=200000, cap_vertices=1 << 12, cap_triangles=1 << 14)


def _livox_msg(n, seed=0):
    rng = np.random.default_rng(seed)
    m = np.zeros(n, LIVOX)
    assert LIVOX.itemsize == 19
    m["offset_time"] = np.sort(rng.integers(0, 100_000_000, n)).astype(np.uint32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    r = rng.uniform(0.2, 60.0, n)
    m["x"], m["y"], m["z"] = (d * r[:, None]).astype(np.float32).T
    m["reflectivity"] = rng.integers(0, 256, n)
    m["tag"] = rng.integers(0, 256, n)
    m["line"] = rng.integers(0, 8, n)          # lines 6 and 7 are not below N_SCANS = 6
    return m


def _livox_expected(m, n_scans, filt, blind):
    out, valid = [], 0
    for i in range(1, len(m)):
        if m["line"][i] < n_scans:
            valid += 1
            if valid % filt == 0:
                x, y, z = np.float32(m["x"][i]), np.float32(m["y"][i]), np.float32(m["z"][i])
                if m["reflectivity"][i] > 4 and float(np.float32(np.float32(x * x + y * y) + z * z)) > blind * blind:
                    out.append([x, y, z, np.float32(m["reflectivity"][i]), np.float32(m["offset_time"][i]) / np.float32(1000000)])
    return np.array(out, np.float32).reshape(-1, 5)


def test_oracle_avia_handler_known_answers(oracle_lib):
    o = make_oracle(oracle_lib, _cfg())
    m = _livox_msg(5000)
    for filt, blind in ((1, 1.0), (3, 4.0)):
        out, n = o.decode_livox(m.view(np.uint8).reshape(-1, 19), 6, filt, blind)
        exp = _livox_expected(m, 6, filt, blind)
        assert n == len(exp) and 500 < n < 5000
        np.testing.assert_array_equal(out, exp)
    assert m["line"][0] < 6 and not np.any(np.all(out[:, :3] == [m["x"][0], m["y"][0], m["z"][0]], axis=1))   # the loop starts at point 1


def _velo_msg(n, seed=1):
    rng = np.random.default_rng(seed)
    m = np.zeros(n, VELO)
    assert VELO.itemsize == 32
    az = rng.uniform(-np.pi, np.pi, n)
    el = np.deg2rad(rng.uniform(-30.0, 6.0, n))       # beyond the HDL-64 fan on both sides
    r = rng.uniform(1.0, 80.0, n)
    m["x"], m["y"], m["z"] = (r * np.cos(el) * np.cos(az)).astype(np.float32), (r * np.cos(el) * np.sin(az)).astype(np.float32), (r * np.sin(el)).astype(np.float32)
    m["intensity"] = rng.uniform(0, 255, n).astype(np.float32)
    return m, np.rad2deg(el)


def test_oracle_velodyne_handler_known_answers(oracle_lib):
    o = make_oracle(oracle_lib, _cfg())
    m, el = _velo_msg(20000)
    out, n = o.decode_velodyne(m.view(np.uint8).reshape(-1, 32), 32, (0, 4, 8, 16), 64)
    safe = np.abs(el - 2.0) > 1e-3
    safe &= np.abs(el + 24.33) > 1e-3
    keep = (el <= 2.0) & (el >= -24.33)
    # scanID > 50 cuts the lowest rows of the fan: below -8.83 deg the id is 32 + int((-8.83 - angle) * 2 + 0.5) -> angle < -18.08 gives 51
    keep &= ~(el < -18.08 - 1e-3) | (el > -18.08 + 1e-3)
    keep &= el > -18.08
    assert abs(n - int(keep.sum())) <= int((~safe).sum()) + 40          # (float rounding right at the gates)
    assert np.all(out[:, 4] == 0.0) and 0.3 * len(m) < n < 0.8 * len(m)
    ang = np.rad2deg(np.arctan(out[:, 2] / np.hypot(out[:, 0], out[:, 1])))
    assert ang.max() <= 2.0 + 1e-3 and ang.min() >= -18.58 - 1e-3


@pytest.mark.gpu
def test_hip_decode_matches_oracle(oracle_lib, hip_lib):
    o, h = make_oracle(oracle_lib, _cfg()), make_hip(hip_lib, _cfg())
    m = _livox_msg(120000, seed=5)
    w = m.view(np.uint8).reshape(-1, 19)
    for filt, blind in ((1, 1.0), (3, 4.0), (7, 0.5)):
        oo, no = o.decode_livox(w, 6, filt, blind)
        oh, nh = h.decode_livox(w, 6, filt, blind)
        assert nh == no
        np.testing.assert_array_equal(oh, oo)
    v, _ = _velo_msg(130000, seed=6)
    d = v.view(np.uint8).reshape(-1, 32)
    oo, no = o.decode_velodyne(d, 32, (0, 4, 8, 16), 64)
    oh, nh = h.decode_velodyne(d, 32, (0, 4, 8, 16), 64)
    # the elevation gate uses float atan: libm and the device differ by an ulp, which can move a point that sits exactly on a gate
    assert abs(nh - no) <= 4
    if nh == no:
        np.testing.assert_array_equal(oh, oo)
    # decode -> undistort -> downsample stays on the device
    _, n = h.decode_livox(w, 6, 1, 1.0, to_host=False)
    imu = np.zeros((8, 7)); imu[:, 0] = np.linspace(0.0125, 0.1, 8); imu[:, 6] = 9.81
    st = capi.make_state(); st[21:24] = [0, 0, -9.81]
    ic = capi.make_imu_ctx(_cfg())
    import ctypes as C
    f = h._f("undistort"); f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lut = C.c_double(0.0)
    assert f(h.ctx, C.c_void_p(h.decode_result_ptr()), n, imu.ctypes.data_as(C.c_void_p), len(imu), 0.0, C.byref(lut), C.byref(ic), st.ctypes.data_as(C.c_void_p), None) == 0
    _, n_ds = h.downsample(h.undistort_result_ptr(), 0.4, n=n, stride=4, to_host=False)
    assert 1000 < n_ds < n
