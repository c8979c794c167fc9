#!/usr/bin/env python3
"""Golden vectors WITH DELIBERATE DISTANCE TIES from the reference's own ikd-Tree (oracle/_ref/libref_ikdtree.so, include/ikd-Tree/ikd_Tree.cpp compiled
from where it lies): a cubic lattice of exactly representable points (spacing 0.125 m: binary fractions, so equal distances are equal floats) admitted
scan by scan as Global_map::append_points_to_global_map does (dedupe cell + 1-NN < min_spacing on the real tree), then 20-NN queries of 256 vertices.
Every query has equal distances at the cut: once its heap is full the tree accepts a candidate only if `dist < top.dist` (strict,
ikd_Tree.cpp:1096-1279; heap order ikd_Tree.h:152-158), so WHICH of the equidistant points it returns depends on the traversal order -- the 20
DISTANCES are well defined, the ids at the cut distance are not.  The committed file (tests/golden/ikdtree_ties_r05.npz) pins what is defined: the
checker (and through its admission the HIP path) must reproduce the accepted vertices bit for bit, every query's 20 distances exactly, every id
below the cut distance, and at the cut the tree's choice must be a subset of the points at that distance.

usage: python tools/make_golden_ikdtree_ties.py     (only where /root/reference exists)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MIN_SPACING = 0.1
SPACING = 0.125


def dp(a):
    return a.ctypes.data_as(C.c_void_p)


def sheet(x0, y0, z0, nx=40):
    gx, gy = np.meshgrid(np.arange(nx), np.arange(nx), indexing="ij")
    p = np.stack([x0 + gx.ravel() * SPACING, y0 + gy.ravel() * SPACING, np.full(gx.size, z0)], axis=1).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([p, np.ones((len(p), 1), np.float32)], axis=1))


def main():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_ikdtree.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    L = C.CDLL(so)
    L.ref_ikd_create.restype = C.c_void_p
    tree = C.c_void_p(L.ref_ikd_create())
    # three sheets of a cubic lattice, the middle one offered twice (every candidate of the repeat meets its own dedupe cell), and a sheet shifted by
    # half a spacing in z: every one of its points is 0.0625 m from two vertices (equal 1-NN distances, both below min_spacing: rejected)
    scans = [sheet(5.0, -2.5, 0.0), sheet(5.0, -2.5, 0.125), sheet(5.0, -2.5, 0.125), sheet(5.0, -2.5, 0.0625), sheet(5.0, -2.5, 0.25)]
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
    rng = np.random.default_rng(1)
    q_ids = np.sort(rng.choice(len(V), 256, replace=False)).astype(np.int32)
    nn_ids = np.full((256, 20), -1, np.int64); nn_d2 = np.zeros((256, 20), np.float32); nn_cnt = np.zeros(256, np.int32)
    ids_r, d_r = np.zeros(20, np.int64), np.zeros(20, np.float32)
    ties_at_cut = 0
    for i, q in enumerate(q_ids):
        n = L.ref_ikd_knn(tree, dp(V[q].copy()), 20, dp(ids_r), dp(d_r))
        nn_cnt[i] = n; nn_ids[i, :n] = ids_r[:n]; nn_d2[i, :n] = d_r[:n]
        d_all = ((V.astype(np.float32) - V[q]) ** 2)
        d_all = (d_all[:, 0] + d_all[:, 1]) + d_all[:, 2]
        ties_at_cut += int((d_all == d_r[n - 1]).sum() > (d_r[:n] == d_r[n - 1]).sum())
    out = os.path.join(ROOT, "tests", "golden", "ikdtree_ties_r05.npz")
    np.savez_compressed(out, scans=np.stack(scans), accepted=V, accepted_per_scan=np.array(per_scan, np.int32), q_ids=q_ids, nn_ids=nn_ids.astype(np.int32), nn_d2=nn_d2,
                        nn_cnt=nn_cnt, min_spacing=np.float64(MIN_SPACING))
    print(out, "vertices", len(V), "per scan", per_scan, "queries with more equidistant points than the cut admits:", ties_at_cut, "bytes", os.path.getsize(out))
    L.ref_ikd_destroy(tree)


if __name__ == "__main__":
    main()
