"""Worker of bench.py's scan cache: `python -m immesh_amd.scan_gen <cache_dir> <n_pts> <kitti 0|1> <extT x y z> <k> [<k> ...]` ray-casts the synthetic
scans k (synth.livox_scan / hdl64_scan, SURVEY 8(d) seeds) into <cache_dir>.  Separate PROCESSES, started with subprocess: the bench process
holds a live HIP runtime, which neither survives a fork nor should be re-imported by a spawned copy of the bench script."""
import os
import sys

import numpy as np

from immesh_amd import synth


def scan_path(cache_dir, n_pts, kitti, k):
    return os.path.join(cache_dir, f"hdl64_{k}.npy" if kitti else f"livox_{n_pts}_{k}.npy")


def _publish(f, arr):
    tmp = f + f".{os.getpid()}.tmp.npy"     # several processes may generate the same scan: publish atomically
    np.save(tmp, arr)
    os.replace(tmp, f)


def generate(cache_dir, n_pts, kitti, extT, k):
    """scan k and its VoxelGrid-downsampled cloud (filter_size_surf: avia.yaml:5 / velodyne.yaml:5) into the cache; returns (raw, down)"""
    f = scan_path(cache_dir, n_pts, kitti, k)
    fd = f[:-4] + "_down.npy"
    if os.path.exists(f):
        raw = np.load(f)
    else:
        R, t = synth.trajectory_pose(k)
        raw = synth.hdl64_scan(k, R, t) if kitti else synth.livox_scan(k, R, t, n_pts=n_pts, extT=np.array(extT))
        _publish(f, raw)
    if os.path.exists(fd):
        down = np.load(fd)
    else:
        down = synth.voxel_grid_downsample(raw, 0.5 if kitti else 0.4)
        _publish(fd, down)
    return raw, down


if __name__ == "__main__":
    cache_dir, n_pts, kitti = sys.argv[1], int(sys.argv[2]), bool(int(sys.argv[3]))
    extT = [float(v) for v in sys.argv[4:7]]
    for k in sys.argv[7:]:
        generate(cache_dir, n_pts, kitti, extT, int(k))
