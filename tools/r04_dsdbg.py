import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from immesh_amd import capi, synth
from conftest import fetch_device
lib = capi.load_hip_library()
cfg = capi.avia_config(cap_root_voxels=1 << 12, cap_scan_points=200000, cap_vertices=1 << 16, cap_triangles=1 << 18)
h = capi.HotPath(lib, cfg, "immesh_")
rng = np.random.default_rng(77)
extT = np.array(list(cfg.extT)); R0, t0 = synth.trajectory_pose(0)
ordinary = np.ascontiguousarray(synth.livox_scan(3, R0, t0, n_pts=50000, extT=extT))
many_leaves = np.zeros((90000, 4), np.float32); many_leaves[:, :3] = rng.uniform(-60, 60, size=(90000, 3))
medium = np.zeros((40000, 4), np.float32); medium[:, :3] = rng.uniform(0, 2.0, size=(40000, 3))
crowded = np.zeros((30000, 4), np.float32); crowded[:, :3] = rng.uniform(0.01, 0.39, size=(30000, 3)); crowded[:50, :3] += 5.0
far = ordinary.copy(); far[7, 0] = np.float32(9.0e5)
for name, cloud in (("ordinary", ordinary), ("many leaves", many_leaves), ("medium", medium), ("crowded", crowded), ("ordinary", ordinary), ("far", far), ("ordinary", ordinary)):
    ref = synth.voxel_grid_downsample(cloud, 0.4)
    print(name, "sync...", flush=True)
    got, n = h.downsample(cloud, 0.4); print("  sync", n, len(ref), np.array_equal(got, ref), flush=True)
    print("  begin", flush=True); h.downsample_begin(cloud, 0.4)
    print("  end", flush=True); n_got, ptr = h.downsample_end()
    print("  async", n_got, np.array_equal(fetch_device(ptr, (n_got, 3)), ref), flush=True)
print("close", flush=True)
h.close()
print("closed", flush=True)
