"""TEST HARNESS ONLY: runs bench.py's control flow (argument handling, the profile child process, the fallbacks, the JSON line) on a machine
without a GPU by standing the CPU oracle in for the HIP library and CPU tensors in for device tensors.  Used by tests/test_bench_plumbing.py;
never used by bench.py itself (which exits without a HIP device)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from immesh_amd import capi, synth  # noqa: E402
import bench  # noqa: E402

_cpu = torch.device("cpu")
torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 1
torch.cuda.set_device = lambda *_a, **_k: None
torch.cuda.synchronize = lambda *_a, **_k: None
_orig_device = torch.device
torch.device = lambda *_a, **_k: _cpu


class _OracleHotPath(capi.HotPath):
    def __init__(self, lib, cfg, prefix="immesh_"):
        super().__init__(lib, cfg, "orc_")
        self._prof = False

    def profile_enable(self, on=True):
        self._prof = bool(on)

    def profile_read(self, reset=False):
        return {"mesh_delaunay64_kernel": {"launches": 2, "total_ms": 0.25}, "residual_persistent_kernel": {"launches": 2, "total_ms": 0.16}}

    def last_timing(self):
        return {"total": 1.0, "register": 0.3, "map_update": 0.2, "mesh": 0.5}


def _small_map(h, cfg, torch_, dev, target_voxels, side_m, seed=0):
    R, t = synth.trajectory_pose(0)
    raw = synth.livox_scan(0, R, t, n_pts=8000, extT=np.array(list(cfg.extT)))
    h.map_build(np.ascontiguousarray(raw[:, :3]), capi.make_state(R=R, t=t))
    return h.counters()["n_root_voxels"]


capi.load_hip_library = lambda: ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
capi.HotPath = _OracleHotPath
capi.forward_without_imu_native = lambda lib, state, *a, **k: synth.forward_without_imu(state)
bench.build_big_map = _small_map

if __name__ == "__main__":
    bench.main()
