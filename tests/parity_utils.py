"""Helpers shared by the GPU parity tests: canonical comparison of plane tables (normals up to the joint sign flip)."""
import numpy as np


def plane_index(recs):
    return {(tuple(r["key"]), int(r["layer"]), int(r["path"])): i for i, r in enumerate(recs)}


def compare_plane_tables(a, b, tol=1e-5, check_counts=True):
    """a: oracle dump, b: HIP dump.  Returns number of planar nodes compared."""
    ia, ib = plane_index(a), plane_index(b)
    assert set(ia) == set(ib), f"node sets differ: only-oracle {len(set(ia) - set(ib))}, only-hip {len(set(ib) - set(ia))}"
    n_planar = 0
    for k, i in ia.items():
        ra, rb = a[i], b[ib[k]]
        assert ra["is_plane"] == rb["is_plane"], f"is_plane differs at {k}"
        assert ra["update_enable"] == rb["update_enable"], f"update_enable differs at {k}"
        if check_counts and (ra["is_plane"] or ra["layer"] == 4):
            assert ra["n_points"] == rb["n_points"] and ra["new_points"] == rb["new_points"], f"point counts differ at {k}: {ra['n_points']},{ra['new_points']} vs {rb['n_points']},{rb['new_points']}"
        if not ra["is_plane"]:
            continue
        n_planar += 1
        s = 1.0 if np.dot(ra["normal"], rb["normal"]) >= 0 else -1.0
        np.testing.assert_allclose(rb["normal"] * s, ra["normal"], rtol=0, atol=tol)
        np.testing.assert_allclose(rb["center"], ra["center"], rtol=0, atol=tol * max(1.0, np.abs(ra["center"]).max()))
        assert abs(rb["d"] * s - ra["d"]) <= tol * max(1.0, abs(ra["d"]))
        assert abs(rb["radius"] - ra["radius"]) <= tol and abs(rb["min_eig"] - ra["min_eig"]) <= tol
        pa = ra["plane_var"].reshape(6, 6)
        pb = rb["plane_var"].reshape(6, 6).copy()
        pb[0:3, 3:6] *= s; pb[3:6, 0:3] *= s   # normal-centre cross block flips with the normal
        scale = np.abs(pa).max()
        np.testing.assert_allclose(pb, pa, rtol=0, atol=tol * scale + 1e-18)
    return n_planar


def compare_plane_tables_fast(a, b, tol=1e-5):
    """Vectorised compare_plane_tables for million-node tables (same bars).  Returns the number of planar nodes compared."""
    def order(r):
        return np.lexsort((r["path"], r["layer"], r["key"][:, 2], r["key"][:, 1], r["key"][:, 0]))
    assert len(a) == len(b), f"node counts differ: oracle {len(a)}, hip {len(b)}"
    a, b = a[order(a)], b[order(b)]
    assert np.array_equal(a["key"], b["key"]) and np.array_equal(a["layer"], b["layer"]) and np.array_equal(a["path"], b["path"]), "node sets differ"
    assert np.array_equal(a["is_plane"], b["is_plane"]), f"is_plane differs at {int((a['is_plane'] != b['is_plane']).sum())} nodes"
    assert np.array_equal(a["update_enable"], b["update_enable"])
    cnt = (a["is_plane"] == 1) | (a["layer"] == 4)
    assert np.array_equal(a["n_points"][cnt], b["n_points"][cnt]) and np.array_equal(a["new_points"][cnt], b["new_points"][cnt])
    pl = a["is_plane"] == 1
    ra, rb = a[pl], b[pl]
    s = np.where((ra["normal"] * rb["normal"]).sum(axis=1) >= 0, 1.0, -1.0)
    assert np.abs(rb["normal"] * s[:, None] - ra["normal"]).max() <= tol
    assert np.all(np.abs(rb["center"] - ra["center"]).max(axis=1) <= tol * np.maximum(1.0, np.abs(ra["center"]).max(axis=1)))
    assert np.all(np.abs(rb["d"] * s - ra["d"]) <= tol * np.maximum(1.0, np.abs(ra["d"])))
    assert np.abs(rb["radius"] - ra["radius"]).max() <= tol and np.abs(rb["min_eig"] - ra["min_eig"]).max() <= tol
    pa = ra["plane_var"].reshape(-1, 6, 6)
    pb = rb["plane_var"].reshape(-1, 6, 6).copy()
    pb[:, 0:3, 3:6] *= s[:, None, None]; pb[:, 3:6, 0:3] *= s[:, None, None]
    scale = np.abs(pa).max(axis=(1, 2))
    assert np.all(np.abs(pb - pa).max(axis=(1, 2)) <= tol * scale + 1e-18)
    return int(pl.sum())
