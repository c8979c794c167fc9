"""The oracle's per-scan registration body pinned to the REFERENCE'S OWN code: oracle/_ref/libref_lio.so is Voxel_mapping::voxel_map_init,
Voxel_mapping::lio_state_estimation (minus the legacy ikd-Tree branch), Voxel_mapping::map_incremental_grow, transformLidar / pointBodyToWorld,
StatesGroup and so3_math.h compiled from where they lie under /root/reference (oracle/Makefile, oracle/ref_lio/ref_lio_wrap.cpp: line ranges cut at
build time, Eigen / PCL / ROS shaped stubs).  Rows a6 (caller), a8, a9, a12, a13, a14, a15 of SURVEY section 8(a).

Pinned: the reference's logic -- the body covariance / cross matrix lists, the covariance propagation, which world point a residual is taken at and
where it is narrowed to float, the H / R^-1 build (CALIB_ANGLE_COV under calib_laser), the order of the sums over the matches, the 18-state update as
written (K_1, G, solution), boxplus / boxminus, the convergence test, the rematch and stop decisions, the covariance update, the map growth's sort +
update, the full-scan transform handed to the mesher.  NOT pinned: Eigen's arithmetic (stub products are plain k-ascending sums, inverse() is
Gauss-Jordan) -- values are compared to rounding (1e-9 relative on sums that cancel, tighter elsewhere), every discrete outcome (iteration counts,
match counts and sets, converged / stop flags, node sets, is_plane, point counts) exactly."""
import ctypes as C
import os

import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle, ROOT
from parity_utils import compare_plane_tables_fast

VP = C.c_void_p


@pytest.fixture(scope="module")
def rl():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_lio.so")
    if not os.path.exists(so):
        if os.path.exists("/root/reference/src/voxel_mapping.cpp"):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref/libref_lio.so not built and /root/reference absent")
    lib = C.CDLL(so)
    lib.rl_create.restype = VP; lib.rl_create.argtypes = [VP]
    lib.rl_destroy.argtypes = [VP]
    lib.rl_set_state.argtypes = [VP, VP]; lib.rl_get_state.argtypes = [VP, VP]
    lib.rl_map_init.argtypes = [VP, VP, C.c_int]
    lib.rl_lio.argtypes = [VP, VP, C.c_int, VP]
    lib.rl_iter.argtypes = [C.c_int] + [VP] * 9
    lib.rl_last_matches.argtypes = [VP, VP, VP, C.c_int]
    lib.rl_grow.argtypes = [VP, VP, C.c_int, VP]
    lib.rl_dump.restype = C.c_int64; lib.rl_dump.argtypes = [VP, VP, C.c_int64]
    lib.rl_root_voxels.restype = C.c_int64; lib.rl_root_voxels.argtypes = [VP]
    lib.rl_state_minus.argtypes = [VP, VP, VP]; lib.rl_state_plus.argtypes = [VP, VP]
    return lib


def _p(a):
    return a.ctypes.data_as(VP)


def _ref_dump(rl, v):
    n = rl.rl_dump(v, None, 0)
    recs = np.zeros(n, capi.PLANE_DTYPE)
    if n:
        rl.rl_dump(v, _p(recs), n)
    return recs


def _ref_lio(rl, v, down, prior, state):
    """lio_state_estimation of the reference: returns (posterior state 348, per-iteration records)."""
    rl.rl_set_state(v, _p(np.ascontiguousarray(state)))
    n_it = rl.rl_lio(v, _p(down), len(down), _p(np.ascontiguousarray(prior)))
    its = []
    for k in range(n_it):
        r = dict(HTH=np.zeros(36), HTz=np.zeros(6), sol=np.zeros(18), G=np.zeros(324), state=np.zeros(24), cov=np.zeros(324))
        nm, rm, fl = C.c_int32(0), C.c_double(0), np.zeros(2, np.int32)
        assert rl.rl_iter(k, _p(r["HTH"]), _p(r["HTz"]), _p(r["sol"]), _p(r["G"]), _p(r["state"]), _p(r["cov"]), C.byref(nm), C.byref(rm), _p(fl)) == 0
        r.update(n_match=nm.value, res_mean=rm.value, converged=int(fl[0]), stop=int(fl[1]))
        its.append(r)
    out = np.zeros(capi.STATE_DOUBLES)
    rl.rl_get_state(v, _p(out))
    return out, its


def _orc_trace(oracle_lib, o, down, prior, state, cap=8):
    f = oracle_lib.orc_register_trace; f.restype = C.c_int
    f.argtypes = [VP, VP, C.c_int32, VP, VP, C.c_int32] + [VP] * 12
    n = len(down)
    out = np.array(state, dtype=np.float64, copy=True)
    HTH, HTz, sol, G = np.zeros((cap, 36)), np.zeros((cap, 6)), np.zeros((cap, 18)), np.zeros((cap, 324))
    st, cov, nm, rm, fl = np.zeros((cap, 24)), np.zeros((cap, 324)), np.zeros(cap, np.int32), np.zeros(cap), np.zeros((cap, 2), np.int32)
    eff_p, eff_n, rinv = np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32), np.zeros(n)
    it = f(o.ctx, _p(down), n, _p(np.ascontiguousarray(prior)), _p(out), cap, _p(HTH), _p(HTz), _p(sol), _p(G), _p(st), _p(cov), _p(nm), _p(rm), _p(fl),
           _p(eff_p), _p(eff_n), _p(rinv))
    assert 0 < it <= cap
    its = [dict(HTH=HTH[k], HTz=HTz[k], sol=sol[k], G=G[k], state=st[k], cov=cov[k], n_match=int(nm[k]), res_mean=rm[k], converged=int(fl[k, 0]),
                stop=int(fl[k, 1])) for k in range(it)]
    M = its[-1]["n_match"]
    return out, its, eff_p[:M], eff_n[:M], rinv[:M]


def _close(a, b, rel, what):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.abs(a).max(), np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max() / scale
    assert err <= rel, f"{what}: relative difference {err:.3e} > {rel:.1e}"


def _setup(kind):
    if kind == "avia":
        cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000)
        scan = lambda k: synth.livox_scan(k, *synth.trajectory_pose(k), n_pts=40000, extT=np.array(list(cfg.extT)))
        leaf = 0.4
    else:
        cfg = capi.velodyne_config(cap_root_voxels=1 << 14, cap_scan_points=200000)
        scan = lambda k: synth.hdl64_scan(k, *synth.trajectory_pose(k), n_az=512)
        leaf = 0.5
    return cfg, scan, leaf


@pytest.mark.parametrize("kind", ["avia", "velodyne"])
def test_scan_stream_through_the_reference_lio_equals_the_oracle(oracle_lib, rl, kind):
    """voxel_map_init on a first scan, then per scan: Forward_without_imu prior -> lio_state_estimation -> map_incremental_grow, on the reference's code
    and on the oracle, EACH FED ITS OWN STATE (a composed run: differences would accumulate).  Every iteration of every scan: same match count, same
    converged / stop flags; HTH, HTz, solution, G, the iterate and (at the stop) the posterior covariance agree to rounding.  After every scan the two
    maps hold the same nodes / is_plane / counts, and the full scan handed to the mesher is the same float cloud."""
    cfg, scan, leaf = _setup(kind)
    o = make_oracle(oracle_lib, cfg)
    v = rl.rl_create(C.byref(cfg))
    R0, t0 = synth.trajectory_pose(0)
    st0 = capi.make_state(R=R0, t=t0)
    raw0 = np.ascontiguousarray(scan(0)[:, :3])
    o.map_build(raw0, st0)
    rl.rl_set_state(v, _p(st0))
    assert rl.rl_map_init(v, _p(raw0), len(raw0)) == 0
    assert rl.rl_root_voxels(v) == o.counters()["n_root_voxels"]
    assert compare_plane_tables_fast(o.dump_planes(), _ref_dump(rl, v), 1e-9) > 300
    s_o, s_r = st0.copy(), st0.copy()
    # the constant-velocity prior needs a velocity / rate: start both from the truth of scan 1
    n_iters_seen, rematch_seen = set(), 0
    for k in range(1, 6):
        Rk, tk = synth.trajectory_pose(k)
        raw = scan(k)
        down = synth.voxel_grid_downsample(raw, leaf)
        # a perturbed prior so that the update has work to do (several iterations, the rematch branch)
        pert_R = synth.so3_exp(np.array([2e-3, -3e-3, 2.5e-3]) * (1 + 0.2 * k)); pert_t = np.array([0.03, -0.02, 0.015])
        pri_o = capi.make_state(R=Rk @ pert_R, t=tk + pert_t, cov_diag=1e-4); pri_o[24:] = synth.forward_without_imu(s_o)[24:]
        pri_r = capi.make_state(R=Rk @ pert_R, t=tk + pert_t, cov_diag=1e-4); pri_r[24:] = synth.forward_without_imu(s_r)[24:]
        post_o, it_o, eff_p, eff_n, rinv = _orc_trace(oracle_lib, o, down, pri_o, pri_o)
        post_r, it_r = _ref_lio(rl, v, down, pri_r, pri_r)
        assert len(it_o) == len(it_r), (k, len(it_o), len(it_r))
        n_iters_seen.add(len(it_r))
        for j, (a, b) in enumerate(zip(it_o, it_r)):
            assert a["n_match"] == b["n_match"] and a["n_match"] > 1000, (k, j)
            assert (a["converged"], a["stop"]) == (b["converged"], b["stop"]), (k, j)
            rematch_seen += a["converged"]
            _close(a["HTH"], b["HTH"], 1e-9, f"scan {k} iteration {j} HTH")
            _close(a["HTz"], b["HTz"], 1e-7, f"scan {k} iteration {j} HTz")       # a sum of signed residuals: cancellation
            _close(a["sol"][:6], b["sol"][:6], 1e-6, f"scan {k} iteration {j} solution")
            np.testing.assert_allclose(a["sol"], b["sol"], rtol=0, atol=1e-10)
            np.testing.assert_allclose(a["G"], b["G"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(a["state"], b["state"], rtol=0, atol=1e-10)
            np.testing.assert_allclose(a["res_mean"], b["res_mean"], rtol=1e-6)
        assert it_r[-1]["stop"] == 1
        np.testing.assert_allclose(post_o[:24], post_r[:24], rtol=0, atol=1e-10)
        _close(post_o[24:], post_r[24:], 1e-7, f"scan {k} posterior covariance")
        assert np.linalg.norm(post_r[9:12] - tk) < np.linalg.norm(pri_r[9:12] - tk)       # and the update moved the pose towards the truth (tight prior)
        # m_laserCloudOri / m_corr_normvect of the last iteration: body points, float normals, float residuals, sqrt(R_inv) in the intensity
        M = it_r[-1]["n_match"]
        lp, ln = np.zeros((M, 4), np.float32), np.zeros((M, 4), np.float32)
        assert rl.rl_last_matches(v, _p(lp), _p(ln), M) == M
        np.testing.assert_array_equal(lp[:, :3], eff_p)                                   # the same points matched, in the same order
        np.testing.assert_array_equal(ln[:, :3], eff_n[:, :3])                            # normals narrowed to float: bit-equal
        np.testing.assert_allclose(ln[:, 3], eff_n[:, 3], rtol=0, atol=2e-7)              # residual (float): the iterates differ by ~1e-12
        np.testing.assert_allclose(lp[:, 3], np.sqrt(rinv).astype(np.float32), rtol=1e-6)
        # map growth + the hand-over to the mesher
        o.map_update(down, post_o)
        world = np.zeros((len(raw), 4), np.float32)
        raw4 = np.ascontiguousarray(raw[:, :4], dtype=np.float32)
        assert rl.rl_grow(v, _p(raw4), len(raw4), _p(world)) == 0
        a, b = o.dump_planes(), _ref_dump(rl, v)
        assert compare_plane_tables_fast(a, b, 1e-8) > 300
        # transformLidar of the full scan (f64 compute, f32 store) against the formula, at the reference's own posterior
        Rp, tp = post_r[:9].reshape(3, 3), post_r[9:12]
        extR, extT = np.array(list(cfg.extR)).reshape(3, 3), np.array(list(cfg.extT))
        want = ((raw4[:, :3].astype(np.float64) @ extR.T + extT) @ Rp.T + tp).astype(np.float32)
        assert (world[:, :3] != want).mean() < 1e-3                                       # (numpy's matmul sums in another order: an ulp now and then)
        np.testing.assert_allclose(world[:, :3], want, rtol=0, atol=2e-5)
        np.testing.assert_array_equal(world[:, 3], raw4[:, 3])
        s_o, s_r = post_o, post_r
    assert len(n_iters_seen) >= 1 and max(n_iters_seen) >= 3
    rl.rl_destroy(v)


def test_stop_and_rematch_decisions_cover_every_branch(oracle_lib, rl):
    """voxel_mapping.cpp:1618-1650: converged -> rematch; the forced rematch at iteration max-2; stop at rematch_num >= 2 or at the last iteration.
    Priors from 'already there' (converges at once: two iterations) to 'far' (runs to max_iteration) must take the same path on both sides."""
    cfg, scan, leaf = _setup("avia")
    o = make_oracle(oracle_lib, cfg)
    v = rl.rl_create(C.byref(cfg))
    R0, t0 = synth.trajectory_pose(0)
    st0 = capi.make_state(R=R0, t=t0)
    raw0 = np.ascontiguousarray(scan(0)[:, :3])
    o.map_build(raw0, st0)
    rl.rl_set_state(v, _p(st0)); rl.rl_map_init(v, _p(raw0), len(raw0))
    down = synth.voxel_grid_downsample(scan(1), leaf)
    R1, t1 = synth.trajectory_pose(1)
    paths = set()
    # (the convergence test is 0.01 deg / 0.15 mm: with sensor noise only a prior that is held tight converges before the last iteration)
    for scale, cov_diag in ((0.0, 1e-13), (0.0, 1e-10), (0.0, 1e-8), (0.02, 1e-6), (0.3, 1e-4), (1.0, 1e-4), (3.0, 1e-4)):
        pri = capi.make_state(R=R1 @ synth.so3_exp(np.array([2e-3, -3e-3, 2.5e-3]) * scale), t=t1 + np.array([0.03, -0.02, 0.015]) * scale, cov_diag=cov_diag)
        _, it_o, _, _, _ = _orc_trace(oracle_lib, o, down, pri, pri)
        _, it_r = _ref_lio(rl, v, down, pri, pri)
        path_o = tuple((a["converged"], a["stop"]) for a in it_o)
        path_r = tuple((a["converged"], a["stop"]) for a in it_r)
        assert path_o == path_r, (scale, path_o, path_r)
        paths.add(path_r)
    assert any(len(p) == 2 for p in paths) and any(len(p) == cfg.max_iter for p in paths), paths
    rl.rl_destroy(v)


def test_boxplus_boxminus_of_the_reference_states_group(oracle_lib, rl):
    """StatesGroup::operator- / operator+= (common_lib.h:249-273) with the reference's own Exp / Log (so3_math.h), against closed forms: Log(Exp(w)) = w,
    (s [+] d) [-] s = d, the small-angle branches (|w| < 1e-5 in Exp(v1,v2,v3): identity; theta < 1e-3 in Log: 0.5 K)."""
    rng = np.random.default_rng(5)
    for mag in (0.0, 1e-7, 5e-6, 1e-4, 5e-4, 2e-3, 0.1, 1.0):
        s = capi.make_state(R=synth.so3_exp(rng.normal(0, 0.5, 3)), t=rng.normal(0, 10, 3))
        d = rng.normal(0, 1, 18); d[:3] = d[:3] / np.linalg.norm(d[:3]) * mag
        s2 = s.copy()
        rl.rl_state_plus(_p(s2), _p(d))
        back = np.zeros(18)
        rl.rl_state_minus(_p(s2), _p(s), _p(back))
        np.testing.assert_allclose(back[3:], d[3:], rtol=0, atol=1e-12)
        if mag > 1e-5:
            np.testing.assert_allclose(back[:3], d[:3], rtol=0, atol=max(2e-9, 2e-7 * mag))   # (Log below 1e-3 is first order: error ~ theta^3 / 12)
            want = s[:9].reshape(3, 3) @ synth.so3_exp(d[:3])
            np.testing.assert_allclose(s2[:9].reshape(3, 3), want, rtol=0, atol=1e-12)
        else:
            np.testing.assert_array_equal(s2[:9], s[:9])                                       # Exp below its threshold is the identity
