"""VoxelGrid down-sampling: the CPU checker's restatement against the harness spec (numpy), and -- on the GPU --
the device kernels against both, bit-exact (float32 centroids accumulated in stable point order)."""
import numpy as np
import pytest

from immesh_amd import capi, synth
from conftest import make_oracle, make_hip, fetch_device


def _clouds():
    cfg = capi.avia_config()
    extT = np.array(list(cfg.extT))
    R, t = synth.trajectory_pose(2)
    yield synth.livox_scan(2, R, t, n_pts=60000, extT=extT), 0.4
    yield synth.hdl64_scan(1, R, t, n_az=600), 0.5
    rng = np.random.default_rng(4)
    yield (rng.normal(0, 3, (5000, 4))).astype(np.float32), 0.25                       # negative coordinates, dense leaves
    yield np.array([[0.1, 0.2, 0.3, 1.0]], np.float32), 0.4                           # single point
    yield np.repeat(np.array([[1.0, -2.0, 0.5, 0.0]], np.float32), 50, axis=0), 0.4   # all in one leaf
    # leaves with hundreds of points (the float32 sums must be formed in scan order whatever the device does in between) and a cloud with far
    # more leaves than points per leaf
    blob = (rng.uniform(0.02, 0.38, (300, 4)) + np.array([2.0, 0.4, -0.8, 0.0])).astype(np.float32)
    yield np.concatenate([rng.normal(0, 4, (3000, 4)).astype(np.float32), blob, rng.normal(0, 4, (3000, 4)).astype(np.float32)]), 0.4
    yield np.concatenate([rng.normal(0, 2, (500, 4)).astype(np.float32), (rng.uniform(0.05, 0.35, (700, 4)) + np.array([0.4, 0.4, 0.4, 0.0])).astype(np.float32)]), 0.4
    yield rng.uniform(-150, 150, (40000, 4)).astype(np.float32), 0.5


def test_oracle_downsample_matches_harness(oracle_lib):
    o = make_oracle(oracle_lib, capi.avia_config())
    for pts, leaf in _clouds():
        ref = synth.voxel_grid_downsample(pts, leaf)
        got, n = o.downsample(np.ascontiguousarray(pts), leaf)
        assert n == len(ref)
        np.testing.assert_array_equal(got, ref)
        got3, n3 = o.downsample(np.ascontiguousarray(pts[:, :3]), leaf)
        np.testing.assert_array_equal(got3, ref)


@pytest.mark.gpu
def test_device_downsample_bit_exact(oracle_lib, hip_lib):
    import torch
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 16, cap_triangles=1 << 18)
    h = make_hip(hip_lib, cfg)
    for pts, leaf in _clouds():
        ref = synth.voxel_grid_downsample(pts, leaf)
        got, n = h.downsample(np.ascontiguousarray(pts), leaf)
        assert n == len(ref)
        np.testing.assert_array_equal(got, ref)
        d = torch.from_numpy(np.ascontiguousarray(pts)).cuda()                        # device-resident input, result kept on the device
        _, n2 = h.downsample(d.data_ptr(), leaf, n=len(pts), stride=4, to_host=False)
        assert n2 == n
    # the device-resident result feeds the registration without leaving HBM
    extT = np.array(list(cfg.extT))
    R0, t0 = synth.trajectory_pose(0)
    raw0 = synth.livox_scan(0, R0, t0, n_pts=30000, extT=extT)
    st = capi.make_state(R=R0, t=t0)
    h.map_build(np.ascontiguousarray(raw0[:, :3]), st)
    R1, t1 = synth.trajectory_pose(1)
    raw1 = synth.livox_scan(1, R1, t1, n_pts=30000, extT=extT)
    down_host = synth.voxel_grid_downsample(raw1, 0.4)
    _, n = h.downsample(np.ascontiguousarray(raw1), 0.4, to_host=False)
    st1 = capi.make_state(R=R1, t=t1, cov_diag=1e-5)
    a = h.residuals(h.downsample_result_ptr(), st1, n=n)
    b = h.residuals(down_host, st1)
    np.testing.assert_array_equal(a["match_idx"], b["match_idx"])
    np.testing.assert_array_equal(a["HTH"], b["HTH"])


@pytest.mark.gpu
def test_device_downsample_three_launch_form_and_its_fall_backs(hip_lib):
    """The hashed VoxelGrid (leaf table, leaf sort, point scatter, per-leaf ordered sums) against the spec, on the shapes that stress it: more leaves than one
    sort chunk holds (the merge across chunks), leaves of 65..2048 points (the in-LDS ordering), a leaf above 2048 points and a cell outside the packed key's
    range (both: the radix pipeline takes over, table handed back clean) -- and an ordinary cloud right after each, through the synchronous entry and the
    asynchronous pair."""
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 16, cap_triangles=1 << 18)
    h = make_hip(hip_lib, cfg)
    rng = np.random.default_rng(77)
    extT = np.array(list(cfg.extT))
    R0, t0 = synth.trajectory_pose(0)
    ordinary = np.ascontiguousarray(synth.livox_scan(3, R0, t0, n_pts=50000, extT=extT))
    many_leaves = np.zeros((90000, 4), np.float32); many_leaves[:, :3] = rng.uniform(-60, 60, size=(90000, 3))          # ~90 k leaves of 0.4 m: 11 chunks
    medium = np.zeros((40000, 4), np.float32); medium[:, :3] = rng.uniform(0, 2.0, size=(40000, 3))                       # 125 leaves of ~320 points
    crowded = np.zeros((30000, 4), np.float32); crowded[:, :3] = rng.uniform(0.01, 0.39, size=(30000, 3)); crowded[:50, :3] += 5.0   # one leaf holds ~30 k points
    far = ordinary.copy(); far[7, 0] = np.float32(9.0e5)                                                                 # cell 2.25 M: outside +-2^20
    for name, cloud in (("ordinary", ordinary), ("many leaves", many_leaves), ("medium", medium), ("crowded", crowded), ("ordinary", ordinary), ("far", far), ("ordinary", ordinary)):
        ref = synth.voxel_grid_downsample(cloud, 0.4)
        got, n = h.downsample(cloud, 0.4)
        assert n == len(ref), name
        np.testing.assert_array_equal(got, ref, err_msg=name)
        h.downsample_begin(cloud, 0.4)
        n_got, ptr = h.downsample_end()
        assert n_got == len(ref), name
        np.testing.assert_array_equal(fetch_device(ptr, (n_got, 3)), ref, err_msg=name + " (async pair)")
    h.close()


@pytest.mark.gpu
def test_async_pair_gives_the_synchronous_result(hip_lib):
    """immesh_downsample_begin / _end: the VoxelGrid of scan k+1 enqueued ahead of time (radix width predicted from the previous cloud's extents) must be
    bit for bit what immesh_downsample returns -- for the first cloud (no prediction), for consecutive scans, and after a jump of the extents (fallback)."""
    torch = pytest.importorskip("torch")
    cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000)
    h = make_hip(hip_lib, cfg)
    extT = np.array(list(cfg.extT))
    clouds = []
    for k in range(4):
        R, t = synth.trajectory_pose(k)
        clouds.append(synth.livox_scan(k, R, t, n_pts=60000, extT=extT))
    clouds.append(np.ascontiguousarray(clouds[0][:500] * np.float32(0.05)))                 # extents collapse ...
    far = clouds[1].copy(); far[:, :3] *= np.float32(40.0); clouds.append(far)               # ... then jump: more key bits than predicted
    for raw in clouds:
        d = torch.from_numpy(raw).cuda()
        want, n_want = h.downsample(d.data_ptr(), 0.4, n=len(raw), stride=4, to_host=True)
        h.downsample_begin(d.data_ptr(), 0.4, n=len(raw), stride=4)
        with pytest.raises(RuntimeError):           # the synchronous call shares the job's table / parameter block: refused while a job is in flight
            h.downsample(d.data_ptr(), 0.4, n=len(raw), stride=4, to_host=True)
        n_got, ptr = h.downsample_end()
        assert n_got == n_want
        got = fetch_device(ptr, (n_got, 3))         # the result stays in HBM
        np.testing.assert_array_equal(got, want)
    h.close()


@pytest.mark.gpu
def test_async_pair_with_host_clouds_beside_host_input_scans(hip_lib):
    """immesh_downsample_begin staged a HOST cloud into the context's staging buffer, which the next immesh_process_scan with host inputs
    overwrites on another stream.  The job now stages into a buffer of its own: the VoxelGrid of host cloud k+1, begun BEFORE scan k is processed from
    host buffers and collected after it, must still be the synchronous result."""
    torch = pytest.importorskip("torch")
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000, cap_vertices=1 << 18, cap_triangles=1 << 20)
    h, ref = make_hip(hip_lib, cfg), make_hip(hip_lib, cfg)
    extT = np.array(list(cfg.extT))
    scans = []
    for k in range(6):
        R, t = synth.trajectory_pose(k)
        scans.append(np.ascontiguousarray(synth.livox_scan(k, R, t, n_pts=100000, extT=extT)))
    R0, t0 = synth.trajectory_pose(0)
    st = capi.make_state(R=R0, t=t0)
    h.map_build(np.ascontiguousarray(scans[0][:, :3]), st)
    st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
    downs = [synth.voxel_grid_downsample(s, 0.4) for s in scans]
    h.downsample_begin(scans[1], 0.4)
    n1, _ = h.downsample_end()
    assert n1 == len(downs[1])
    for k in range(1, 5):
        h.downsample_begin(scans[k + 1], 0.4)                       # host cloud k+1: staged and down-sampled on the pre-processing stream ...
        prior = synth.forward_without_imu(st)
        st, _ = h.process_scan(downs[k], scans[k], prior, prior, frame_idx=k, do_mesh=2)   # ... while scan k goes in as HOST buffers (the context's staging)
        n_got, ptr = h.downsample_end()
        want, n_want = ref.downsample(scans[k + 1], 0.4)
        assert n_got == n_want == len(downs[k + 1])
        h.mesh_wait()    # (the mesher's worker thread captures its launch graphs during the first scans: a plain hipMemcpy from this thread meanwhile is refused -- hipErrorStreamCaptureImplicit)
        got = fetch_device(ptr, (n_got, 3))
        np.testing.assert_array_equal(got, want)
    h.mesh_wait()
    h.close(); ref.close()


@pytest.mark.gpu
def test_async_pair_behind_an_asynchronous_scan_with_and_without_a_next_registration(hip_lib):
    """Behind an asynchronous immesh_process_scan the library takes the caller for a scan loop and holds the next VoxelGrid job at a gate until the NEXT
    registration launch is running (ds_gate_kernel: a scheduling hint).  The job must give the synchronous result both when that registration comes
    (begin before immesh_process_scan, the loop's order) and when it never does (the gate's own time-out lets the sequence through)."""
    torch = pytest.importorskip("torch")
    import time
    cfg = capi.avia_config(cap_root_voxels=1 << 16, cap_scan_points=200000)
    h, ref = make_hip(hip_lib, cfg), make_hip(hip_lib, cfg)
    extT = np.array(list(cfg.extT))
    scans = []
    for k in range(5):
        R, t = synth.trajectory_pose(k)
        scans.append(np.ascontiguousarray(synth.livox_scan(k, R, t, n_pts=100000, extT=extT)))
    d_raw = [torch.from_numpy(s).cuda() for s in scans]
    downs = [synth.voxel_grid_downsample(s, 0.4) for s in scans]
    d_down = [torch.from_numpy(np.ascontiguousarray(d[:, :3], np.float32)).cuda() for d in downs]
    R0, t0 = synth.trajectory_pose(0)
    st = capi.make_state(R=R0, t=t0)
    h.map_build(np.ascontiguousarray(scans[0][:, :3]), st)
    st[12:15] = [1.0, 0, 0]; st[15:18] = [0, 0, np.deg2rad(2.0)]
    NOWAIT = 0x10

    def check(k, n_got, ptr):
        want, n_want = ref.downsample(scans[k], 0.4)
        assert n_got == n_want == len(downs[k])
        np.testing.assert_array_equal(fetch_device(ptr, (n_got, 3)), want)

    for k in range(1, 4):
        h.downsample_begin(d_raw[k + 1].data_ptr(), 0.4, n=len(scans[k + 1]), stride=4)      # (from the second round on: held until scan k's registration runs)
        prior = synth.forward_without_imu(st)
        st, _ = h.process_scan(d_down[k].data_ptr(), d_raw[k].data_ptr(), prior, prior, frame_idx=k, do_mesh=NOWAIT, n_ds=len(downs[k]), n_raw=len(scans[k]))
        n_got, ptr = h.downsample_end()
        check(k + 1, n_got, ptr)
    # the loop stops: one more job, no registration behind it -- the gate gives up after its bound (150 us), the result is the same
    t0_ = time.perf_counter()
    h.downsample_begin(d_raw[2].data_ptr(), 0.4, n=len(scans[2]), stride=4)
    n_got, ptr = h.downsample_end()
    assert time.perf_counter() - t0_ < 0.05
    check(2, n_got, ptr)
    h.close(); ref.close()
