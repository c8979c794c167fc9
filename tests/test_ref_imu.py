"""The steps before the path pinned to the REFERENCE'S OWN code: oracle/_ref/libref_imu.so is ImuProcess::UndistortPcl and
ImuProcess::Forward_without_imu (src/IMU_Processing.cpp:755-958, 486-553) compiled from where they lie behind Eigen / PCL / ROS shaped stubs
(oracle/Makefile, oracle/ref_imu/ref_imu_wrap.cpp: excerpts cut by line range at build time, nothing copied).  Checked against them: the oracle's
undistort_pcl (which the HIP path is compared with on the GPU, tests/test_undistort.py), the harness's prior (synth.forward_without_imu) and the
PRODUCT's host function immesh_forward_without_imu (imu_host.hpp: no device needed).  Stamps are distinct: the reference's std::sort is unstable."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle, ROOT
from test_undistort import _package, _state, _cfg_identity

VP = C.c_void_p


@pytest.fixture(scope="module")
def ref_imu():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_imu.so")
    if os.path.exists("/root/reference/src/IMU_Processing.cpp"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    elif not os.path.exists(so):
        pytest.skip("oracle/_ref/libref_imu.so not built and /root/reference absent")
    lib = C.CDLL(so)
    lib.ri_forward_without_imu.argtypes = [VP, C.c_double, C.c_int, VP, VP]
    lib.ri_undistort.argtypes = [VP, C.c_int, VP, C.c_int, C.c_double, VP, VP, VP, VP]
    return lib


def _ref_undistort(lib, pts, imu, beg, lut, ic, st):
    st = np.array(st, dtype=np.float64, copy=True)
    out = np.zeros((len(pts), 4), np.float32)
    samples = np.ascontiguousarray(imu, dtype=np.float64)
    l = C.c_double(lut)
    n = lib.ri_undistort(np.ascontiguousarray(pts).ctypes.data_as(VP), len(pts), samples.ctypes.data_as(VP), len(samples), beg, C.byref(l), C.byref(ic), st.ctypes.data_as(VP),
                         out.ctypes.data_as(VP))
    assert n == len(pts)
    return out, st, l.value


def _copy_ic(ic):
    c = capi.ImuCtx()
    C.memmove(C.byref(c), C.byref(ic), C.sizeof(capi.ImuCtx))
    return c


@pytest.mark.parametrize("case", ["static", "yaw", "tumble", "two packages", "stamps start late"])
def test_undistort_pcl_of_the_reference_equals_the_oracle(oracle_lib, ref_imu, case):
    cfg = _cfg_identity() if case in ("static", "yaw") else capi.avia_config(cap_root_voxels=1 << 10, cap_scan_points=200000, cap_vertices=1 << 12, cap_triangles=1 << 14)
    o = make_oracle(oracle_lib, cfg)
    kw = {"static": {}, "yaw": {"gyr": (0, 0, 0.8)}, "tumble": {"gyr": (0.4, -0.7, 1.1), "acc": (0.8, -0.3, 9.6), "noise": 0.02},
          "two packages": {"gyr": (0.2, 0.1, -0.5), "noise": 0.01}, "stamps start late": {"gyr": (0.3, 0.2, 0.9), "noise": 0.01}}[case]
    def package(seed):
        pts, imu = _package(n=3000, seed=seed, **kw)
        dup = np.flatnonzero(pts[:-1, 4] == pts[-1, 4])                    # (_package stamps its last point with the scan end, which the linspace holds already)
        pts[dup, 4] = np.float32(99.99)
        return pts, imu
    pts, imu = package(3)
    if case == "stamps start late":
        pts[:, 4] = (37.0 + pts[:, 4] * 0.63).astype(np.float32)       # the earliest point lies in a later IMU interval: the "compensated again" quirk (:950-953)
    assert len(np.unique(pts[:, 4])) == len(pts)
    st0 = _state(vel=(1.5, -0.4, 0.2))
    st0[15:18] = [0.01, -0.02, 0.005]; st0[18:21] = [0.05, 0.02, -0.03]
    ic_o = capi.make_imu_ctx(cfg, gyr0=kw.get("gyr", (0, 0, 0)))
    ic_r = _copy_ic(ic_o)
    beg, lut_o, lut_r, st_o, st_r = 0.0, 0.0, 0.0, st0.copy(), st0.copy()
    for k_pkg in range(2 if case == "two packages" else 1):
        if k_pkg == 1:
            pts, imu = package(4)
            imu[:, 0] += 0.1; beg = 0.1
        out_o, st_o, lut_o = o.undistort(pts, imu, beg, lut_o, ic_o, st_o)
        out_r, st_r, lut_r = _ref_undistort(ref_imu, pts, imu, beg, lut_r, ic_r, st_r)
        np.testing.assert_array_equal(out_r[:, 3], out_o[:, 3])                       # the same points in the same (time) order
        np.testing.assert_allclose(out_r[:, :3], out_o[:, :3], rtol=0, atol=2e-6)     # float32 coordinates of 30 m: one spacing
        assert np.mean(out_r[:, :3] == out_o[:, :3]) > 0.999                          # ... and all but a handful bit-equal
        np.testing.assert_allclose(st_r[:24], st_o[:24], rtol=0, atol=1e-12)          # propagated state
        np.testing.assert_allclose(st_r[24:], st_o[24:], rtol=1e-12, atol=1e-18)      # propagated covariance
        assert lut_r == lut_o
        assert ic_r.last_lidar_end_time == ic_o.last_lidar_end_time and ic_r.last_imu.t == ic_o.last_imu.t
        np.testing.assert_allclose(list(ic_r.acc_s_last) + list(ic_r.angvel_last), list(ic_o.acc_s_last) + list(ic_o.angvel_last), rtol=0, atol=1e-12)


def test_forward_without_imu_of_the_reference_equals_the_product_and_the_harness(ref_imu):
    """immesh_forward_without_imu is host code of the product library (no device): compared here, on CPU, with the reference's own body."""
    lib = capi.load_hip_library()
    rng = np.random.default_rng(2)
    for trial in range(6):
        R, _ = synth.trajectory_pose(trial)
        A = rng.normal(size=(18, 18)) * 1e-3
        st = capi.make_state(R=R, t=rng.normal(size=3) * 5, cov_diag=1e-4)
        st[24:] = (A @ A.T + np.eye(18) * 1e-6).reshape(-1)
        st[12:15] = rng.normal(size=3); st[15:18] = rng.normal(size=3) * 0.2; st[18:21] = rng.normal(size=3) * 0.01; st[21:24] = [0, 0, -9.81]
        for dt, first in ((0.1, 0), (0.05, 0), (0.237, 0), (123.0, 1)):
            ref = st.copy()
            cg, ca = np.full(3, 0.3), np.full(3, 0.5)
            assert ref_imu.ri_forward_without_imu(ref.ctypes.data_as(VP), dt, first, cg.ctypes.data_as(VP), ca.ctypes.data_as(VP)) == 1
            eff = 0.1 if first else dt                                                  # (b_first_frame_: the reference propagates over 0.1 s, :501-505)
            mine = capi.forward_without_imu_native(lib, st, dt=eff, cov_gyr=0.3, cov_acc=0.5)
            np.testing.assert_allclose(mine[:24], ref[:24], rtol=0, atol=1e-13)
            np.testing.assert_allclose(mine[24:], ref[24:], rtol=1e-10, atol=1e-18)      # (sums of 18 products: a cancelling entry differs in its last digits)
            np.testing.assert_allclose(synth.forward_without_imu(st, dt=eff), ref, rtol=1e-11, atol=1e-13)
