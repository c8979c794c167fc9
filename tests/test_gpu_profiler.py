"""In-library HIP-event profiler (immesh_profile_enable / immesh_profile_read) with the mesher on -- the configuration whose bench legs failed at
the end of round 1.  Round-2 bisect (tools/debug_profiler.sh, tools/debug_masked.sh): the failures needed the CU-masked (blocking) mesher
streams; the mask is opt-in now (IMMESH_MESH_CUS) and these tests are part of the GPU tier."""

import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_hip

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


@pytest.mark.parametrize("mode", [1, 2])
def test_profiled_scans_with_mesher(hip_lib, mode):
    torch = pytest.importorskip("torch")
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    h = make_hip(hip_lib, cfg)
    extT = np.array(list(cfg.extT))
    scans = []
    for k in range(10):
        R, t = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, R, t, n_pts=100000, extT=extT)
        scans.append((R, t, torch.from_numpy(raw).cuda(), torch.from_numpy(synth.voxel_grid_downsample(raw, 0.4)).cuda()))
    R0, t0, raw0, _ = scans[0]
    st = capi.make_state(R=R0, t=t0)
    h.map_build(np.ascontiguousarray(raw0.cpu().numpy()[:, :3]), st)
    st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
    for k in range(1, 10):
        if k == 5:
            if mode == 2:
                h.mesh_wait()
            h.profile_enable(True)                      # scans 5.. run with HIP events (and the spin prelude) around every launch
        prior = capi.forward_without_imu_native(hip_lib, st)
        _, _, raw, down = scans[k]
        st, info = h.process_scan(down.data_ptr(), raw.data_ptr(), prior, prior, frame_idx=k, do_mesh=1 if k >= 5 else mode, n_ds=len(down), n_raw=len(raw))
        assert info["n_match"] > 1000
    ks = h.profile_read()
    h.profile_enable(False)
    # one resident launch per scan for all EKF passes.  The triangulations: the one-launch mesh_delaunay64_kernel, or -- when the asynchronous scans before
    # the profiled ones left the worker in its three-jobs arrangement (mode 2) -- mesh_tri64_kernel + mesh_diff64_kernel; five scans either way
    assert ks["residual_persistent_kernel"]["launches"] == 5
    tri = {k_: ks.get(k_, {"launches": 0, "total_ms": 0.0}) for k_ in ("mesh_delaunay64_kernel", "mesh_tri64_kernel", "mesh_diff64_kernel")}
    assert tri["mesh_delaunay64_kernel"]["launches"] + tri["mesh_tri64_kernel"]["launches"] == 5 and tri["mesh_tri64_kernel"]["launches"] == tri["mesh_diff64_kernel"]["launches"]
    assert 0.005 < (tri["mesh_delaunay64_kernel"]["total_ms"] + tri["mesh_tri64_kernel"]["total_ms"]) / 5 < 5.0
