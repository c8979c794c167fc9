#!/usr/bin/env python3
"""Golden vectors for the legacy registration map produced by the REFERENCE'S OWN ikd-Tree (include/ikd-Tree/ikd_Tree.cpp,
compiled from where it lies into oracle/_ref/libref_ikdtree.so): set_downsample_param + Build of a first down-sampled scan
(src/voxel_mapping.cpp:1906-1914), three Add_Points(.., true) batches (src/ImMesh_mesh_reconstruction.cpp:439), the surviving point set
(flatten) and 5-NN answers (Nearest_Search, as the "Old map ICP" matcher asks at voxel_mapping.cpp:1428).  Only runs where /root/reference
exists; tests/golden/ikdmap_r01.npz is committed and travels to the GPU box, where the oracle AND the HIP path are compared against it.

usage: python tools/make_golden_ikdmap.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from immesh_amd import synth  # noqa: E402

DS = 0.4   # config/avia.yaml:6 filter_size_map


def dp(a):
    return a.ctypes.data_as(C.c_void_p)


def world_scan(k, n_pts=20000, leaf=0.4):
    R, t = synth.trajectory_pose(k)
    raw = synth.livox_scan(k, R, t, n_pts=n_pts)
    down = synth.voxel_grid_downsample(raw, leaf)
    return ((down.astype(np.float64) @ R.T) + t).astype(np.float32)


def main():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_ikdtree.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    L = C.CDLL(so)
    L.ref_ikd_create.restype = C.c_void_p
    tree = C.c_void_p(L.ref_ikd_create())
    L.ref_ikd_set_downsample(tree, C.c_float(DS))
    scans = [world_scan(k) for k in range(4)]
    L.ref_ikd_build(tree, dp(scans[0]), len(scans[0]))
    sizes = [L.ref_ikd_validnum(tree)]
    for s in scans[1:]:
        L.ref_ikd_add_points_ds(tree, dp(s), len(s))
        sizes.append(L.ref_ikd_validnum(tree))
    cap = sizes[-1] + 16
    pts = np.zeros((cap, 3), np.float32)
    n = L.ref_ikd_flatten(tree, dp(pts), cap)
    assert n == sizes[-1], (n, sizes)
    pts = pts[:n]
    pts = pts[np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0]))]
    rng = np.random.default_rng(7)
    q = world_scan(4)[rng.permutation(2000)[:400]] + rng.normal(0, 0.02, (400, 3)).astype(np.float32)
    nn = np.zeros((len(q), 5, 3), np.float32); d2 = np.zeros((len(q), 5), np.float32)
    for i in range(len(q)):
        k = L.ref_ikd_knn_xyz(tree, dp(np.ascontiguousarray(q[i])), 5, dp(nn[i]), dp(d2[i]))
        assert k == 5
    # Delete_Point_Boxes with two slabs, as laser_map_fov_segment produces them (voxel_mapping_common.cpp:258-276)
    lo, hi = pts.min(axis=0) - 1.0, pts.max(axis=0) + 1.0
    boxes = np.array([[lo[0], lo[1], lo[2], lo[0] + 0.3 * (hi[0] - lo[0]), hi[1], hi[2]],
                      [lo[0], hi[1] - 0.25 * (hi[1] - lo[1]), lo[2], hi[0], hi[1], hi[2]]], np.float32)
    n_del = L.ref_ikd_delete_boxes(tree, dp(boxes), 2)
    left = np.zeros((cap, 3), np.float32)
    n_left = L.ref_ikd_flatten(tree, dp(left), cap)
    assert n_left == L.ref_ikd_validnum(tree)
    left = left[:n_left]
    left = left[np.lexsort((left[:, 2], left[:, 1], left[:, 0]))]
    out = os.path.join(ROOT, "tests", "golden", "ikdmap_r01.npz")
    np.savez_compressed(out, ds=np.float32(DS), scan0=scans[0], scan1=scans[1], scan2=scans[2], scan3=scans[3], sizes=np.array(sizes), points=pts, queries=q, nn=nn, d2=d2,
                        boxes=boxes, n_deleted=np.int64(n_del), points_after_delete=left)
    print("deleted", n_del, "left", n_left)
    print("wrote", out, "sizes", sizes)


if __name__ == "__main__":
    main()
