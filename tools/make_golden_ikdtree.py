#!/usr/bin/env python3
"""Golden vectors produced by the REFERENCE'S OWN code: its in-tree ikd-Tree (include/ikd-Tree/ikd_Tree.cpp, compiled from where it lies
into oracle/_ref/libref_ikdtree.so by oracle/Makefile) drives the vertex admission of Global_map::append_points_to_global_map
(pointcloud_rgbd.cpp:411-552: dedupe cell + 1-NN < min_spacing) and answers 20-NN queries (retrieve_neighbor_pts_kdtree,
mesh_rec_geometry.cpp:350).  Only runs where /root/reference exists; the output tests/golden/ikdtree_r01.npz is committed and travels
to the GPU box, where the tests compare the oracle AND the HIP path against it.

usage: python tools/make_golden_ikdtree.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from immesh_amd import synth  # noqa: E402

MIN_SPACING = 0.1   # mapping_avia.launch: points_minimum_scale * distance_scale


def dp(a):
    return a.ctypes.data_as(C.c_void_p)


def main():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_ikdtree.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    L = C.CDLL(so)
    L.ref_ikd_create.restype = C.c_void_p
    tree = C.c_void_p(L.ref_ikd_create())
    # three overlapping world-frame scans of the procedural scene (8000 candidates each, every point offered: budget >= n -> step 1)
    scans = []
    for k in range(3):
        R, t = synth.trajectory_pose(k)
        raw = synth.livox_scan(k, R, t, n_pts=8000)
        w = raw.copy()
        w[:, :3] = (raw[:, :3].astype(np.float64) @ R.T + t).astype(np.float32)
        scans.append(np.ascontiguousarray(w))
    grid, verts, per_scan = {}, [], []
    idx1, d1 = np.zeros(1, np.int64), np.zeros(1, np.float32)
    for w in scans:
        base = len(verts)
        for p in w:
            g = tuple(int(np.round(float(p[a]) / MIN_SPACING)) for a in range(3))
            if g in grid:
                continue
            if L.ref_ikd_has_root(tree):
                n = L.ref_ikd_knn(tree, dp(p[:3].copy()), 1, dp(idx1), dp(d1))
                if n and float(np.sqrt(d1[0])) < MIN_SPACING:
                    continue
            grid[g] = len(verts)
            L.ref_ikd_add(tree, dp(p[:3].copy()), C.c_long(len(verts)))
            verts.append(p[:3].copy())
        per_scan.append(len(verts) - base)
    V = np.array(verts, np.float32)
    # 20-NN of 256 vertices against the final tree
    rng = np.random.default_rng(0)
    q_ids = np.sort(rng.choice(len(V), 256, replace=False)).astype(np.int32)
    nn_ids = np.full((256, 20), -1, np.int64); nn_d2 = np.zeros((256, 20), np.float32); nn_cnt = np.zeros(256, np.int32)
    ids_r, d_r = np.zeros(20, np.int64), np.zeros(20, np.float32)
    for i, q in enumerate(q_ids):
        n = L.ref_ikd_knn(tree, dp(V[q].copy()), 20, dp(ids_r), dp(d_r))
        nn_cnt[i] = n; nn_ids[i, :n] = ids_r[:n]; nn_d2[i, :n] = d_r[:n]
    out = os.path.join(ROOT, "tests", "golden", "ikdtree_r01.npz")
    np.savez_compressed(out, scans=np.stack(scans), accepted=V, accepted_per_scan=np.array(per_scan, np.int32), q_ids=q_ids, nn_ids=nn_ids.astype(np.int32), nn_d2=nn_d2,
                        nn_cnt=nn_cnt, min_spacing=np.float64(MIN_SPACING))
    print(out, "vertices", len(V), "per scan", per_scan, "bytes", os.path.getsize(out))
    L.ref_ikd_destroy(tree)


if __name__ == "__main__":
    main()
