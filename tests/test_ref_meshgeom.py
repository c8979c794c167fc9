"""The oracle's per-voxel mesh geometry pinned to the REFERENCE'S OWN code: oracle/_ref/libref_meshgeom.so holds compute_angle / is_face_is_ok,
triangle_compare, delaunay_triangulation and correct_triangle_index cut out of /root/reference/src/meshing/mesh_rec_geometry.cpp by line range and compiled
next to the reference's real triangle.hpp / tools_kd_hash.hpp behind Eigen / CGAL shaped stubs (oracle/ref_meshgeom/).  Pinned: everything the reference
does AROUND the CGAL call in delaunay_triangulation -- centre and covariance of the vertex set, the choice of the short / mid axes and their sign flips
(rows 0 and 1 of the centred points), long = short x mid, the 2-D projection, the forced 150-degree filter with its `* 57.3`, the emitted id triples --
plus the add / remove / existing split of triangle_compare and m_index_flip of correct_triangle_index.  NOT pinned: CGAL's triangulation (the
CGAL-shaped class forwards to the oracle's Bowyer-Watson) and Eigen's arithmetic (plain sums; SelfAdjointEigenSolver = Jacobi + ascending sort)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def rg():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_meshgeom.so")
    if not os.path.exists(so):
        if os.path.exists("/root/reference/src/meshing/mesh_rec_geometry.cpp"):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref/libref_meshgeom.so not built and /root/reference absent")
    lib = C.CDLL(so)
    lib.rg_compute_angle.restype = C.c_double; lib.rg_compute_angle.argtypes = [C.c_void_p] * 3
    lib.rg_delaunay.restype = C.c_int64; lib.rg_delaunay.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
    lib.rg_flip.argtypes = [C.c_void_p] * 6
    lib.rg_triangle_compare.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _patch(rng, n, kind):
    """a noisy surface patch the size of a mesh-voxel neighbourhood (0.1 m lattice spacing, 1.2 m across)"""
    u, v = rng.uniform(-0.6, 0.6, n), rng.uniform(-0.6, 0.6, n)
    if kind == "ground":
        P = np.stack([u, v, rng.normal(0, 0.01, n)], axis=1)
    elif kind == "wall":
        P = np.stack([rng.normal(0, 0.01, n), u, v], axis=1)
    else:   # an edge: two planes meeting
        P = np.stack([u, v, np.where(u > 0, u, 0.0) + rng.normal(0, 0.01, n)], axis=1)
    Rz = np.array([[np.cos(0.3), -np.sin(0.3), 0], [np.sin(0.3), np.cos(0.3), 0], [0, 0, 1]])
    return np.ascontiguousarray((P @ Rz.T + np.array([12.0, -7.0, 1.5])).astype(np.float32).astype(np.float64))   # vertex positions are float values


def test_delaunay_triangulation_of_the_reference_code_equals_the_oracles(oracle_lib, rg):
    oracle_lib.orc_voxel_delaunay.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(21)
    n_faces = 0
    for trial in range(60):
        n = int(rng.integers(3, 120))
        P = _patch(rng, n, ("ground", "wall", "edge")[trial % 3])
        ids = np.arange(n, dtype=np.int64) * 7 + 100          # m_pt_index: arbitrary ascending ids
        axes = np.zeros(9)
        out = np.zeros(6 * n + 12, np.int64)
        m = rg.rg_delaunay(_p(P), _p(ids), n, _p(axes), _p(out), len(out))
        sa = np.zeros(3)
        t32 = np.zeros(6 * n + 12, np.int32)
        mo = oracle_lib.orc_voxel_delaunay(_p(P), n, _p(sa), _p(t32), len(t32))
        assert m == mo and m % 3 == 0
        a = np.sort(((out[:m] - 100) // 7).reshape(-1, 3), axis=1); b = np.sort(t32[:mo].reshape(-1, 3), axis=1)
        assert sorted(map(tuple, a.tolist())) == sorted(map(tuple, b.tolist())), trial      # the same faces
        np.testing.assert_array_equal((out[:m] - 100) // 7, t32[:mo])                         # ... emitted in the same order with the same vertex order
        np.testing.assert_allclose(axes[6:9], sa, rtol=0, atol=1e-12)                          # the short axis incl. its sign
        assert abs(np.linalg.norm(axes[6:9]) - 1) < 1e-9 and abs(np.dot(axes[0:3], axes[6:9])) < 1e-9
        n_faces += m // 3
    assert n_faces > 2000


def test_angle_filter(rg):
    # compute_angle uses `* 57.3`: a right angle reads 90.0064; the filter limit of 150 is an effective 149.96 degrees
    a, b, c = np.array([0.0, 0.0]), np.array([1.0, 0.0]), np.array([0.0, 2.0])
    assert abs(rg.rg_compute_angle(_p(a), _p(b), _p(c)) - np.arccos(0.0) * 57.3) < 1e-12
    # a sliver with a 170-degree corner is dropped by delaunay_triangulation, its neighbours are kept
    P = np.array([[0, 0, 0], [1, 0, 0], [0.5, 0.04, 0], [0.5, 1.0, 0], [0.5, -1.0, 0.0]], dtype=np.float64) + np.array([3.0, 4.0, 0.5])
    axes, out = np.zeros(9), np.zeros(64, np.int64)
    m = rg.rg_delaunay(_p(P), _p(np.arange(5, dtype=np.int64)), 5, _p(axes), _p(out), 64)
    tris = {tuple(sorted(t)) for t in out[:m].reshape(-1, 3).tolist()}
    assert (0, 1, 2) not in tris and len(tris) >= 3


def test_triangle_compare_of_the_reference_code(rg):
    rng = np.random.default_rng(5)
    for _ in range(50):
        pool = [tuple(sorted(rng.choice(40, 3, replace=False).tolist())) for _ in range(60)]
        old = sorted(set(pool[:35])); fresh = sorted(set(pool[20:]))
        o = np.array(old, np.int32).reshape(-1, 3)
        f = np.array([rng.permutation(t).tolist() for t in fresh], np.int64).reshape(-1, 3)   # vertex order inside a face is arbitrary: Triangle's constructor sorts
        rem, add, ex = np.zeros((80, 3), np.int32), np.zeros((80, 3), np.int32), np.zeros((80, 3), np.int32)
        n = np.zeros(3, np.int32)
        rg.rg_triangle_compare(_p(o), len(o), _p(f), len(f), _p(rem), _p(add), _p(ex), _p(n))
        assert sorted(map(tuple, rem[:n[0]].tolist())) == sorted(set(old) - set(fresh))
        assert sorted(map(tuple, add[:n[1]].tolist())) == sorted(set(fresh) - set(old))
        assert sorted(map(tuple, ex[:n[2]].tolist())) == sorted(set(old) & set(fresh))


def test_correct_triangle_index_of_the_reference_code_equals_the_oracles(oracle_lib, rg):
    oracle_lib.orc_flip_of.argtypes = [C.c_void_p] * 5
    rng = np.random.default_rng(8)
    seen = set()
    for k in range(400):
        a, b, c = (rng.normal(0, 1, 3) + np.array([5.0, 2.0, 1.0]) for _ in range(3))
        if k % 50 == 0:
            c = a + 2.0 * (b - a)          # degenerate (collinear): normal = (0, 0, 1)
        cam = rng.normal(0, 3, 3)
        sa = rng.normal(0, 1, 3); sa /= np.linalg.norm(sa)
        nrm = np.zeros(3)
        f1 = rg.rg_flip(_p(a), _p(b), _p(c), _p(cam), _p(sa), _p(nrm))
        f2 = oracle_lib.orc_flip_of(_p(a), _p(b), _p(c), _p(cam), _p(sa))
        assert f1 == f2
        assert nrm[2] >= 0                    # the stored normal is forced to z >= 0 (:430-433)
        seen.add(f1)
    assert seen == {0, 1}
