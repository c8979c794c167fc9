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
    hlive = {}
    for k, (raw, down, prior) in enumerate(scans):
        if k == 0:
            h.map_build(np.ascontiguousarray(raw[:, :3]), prior)
            continue
        sh, ih = h.register(down, prior, prior)
        h.map_update(down, sh)
        m = h.mesh_scan(_world_like_the_shim(raw, sh, cfg), sh[9:12], frame_idx=k - 1)
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
    h.close()
    # (2) against the oracle
    live, exact = {}, True
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
        exact = exact and got[k]["nv"] == nv and got[k]["nl"] == len(live) and int(got[k]["hash"]) == _live_hash(live)
        if not exact:   # a world point rounded differently (poses agree to ~1e-12, not bit for bit): the maps differ by single vertices
            assert abs(int(got[k]["nv"]) - nv) <= 5 and abs(int(got[k]["nl"]) - len(live)) <= 0.01 * len(live)
    assert got[1]["nv"] > 500 and got[-1]["nl"] > 3000
    if expect_ply:
        assert os.path.getsize("/tmp/immesh_dropin_test.ply") > 10000       # save_to_ply_file through the shim
