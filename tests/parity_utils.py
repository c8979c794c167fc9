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


# ---------------------------------------------------------------------------------------------------------------------------------------------------
# Strict comparison of a COMPOSED run (registration -> map growth -> meshing in one call per scan) against the oracle.
#
# north_star: "vertex indices bit-exact on identical input scans, pose within 1e-5".  The two pipelines estimate poses that agree to ~1e-12 (different
# summation order of H^T R^-1 H over the points, a regrouped but algebraically identical 18-state update).  transformLidar
# (src/voxel_mapping_common.cpp:709-726) computes the world frame in f64 and STORES f32, so a pose difference of 1e-12 occasionally rounds one
# coordinate of a 100 000-pt scan to the neighbouring float.  From then on the two mesh maps legitimately differ by what that candidate changed.
# A bound like "at most 5 vertices apart" would let a real mesher bug in a composed run through, so the check is exact, in three parts:
#   (1) every scan: the device's world-frame cloud and the oracle's differ by AT MOST ONE f32 SPACING + the pose difference d = |dt| + |dR| |p|
#       (itself asserted below 1e-8; measured ~1e-12) per coordinate -- attribution: nothing but the rounding of a pose that agrees far inside the
#       registration bar -- and the count of differing mesher candidates is recorded;
#   (2) every scan, also after a divergence: a SHADOW oracle mesher that is fed the device's own world-frame cloud (immesh_mesh_world_scan) must
#       reproduce every list of the device pipeline bit for bit -- the oracle "re-based" on the device's inputs, so scan 11 of a composed run is
#       compared as exactly as scan 1;
#   (3) as long as no candidate coordinate has differed yet, the device's lists must equal the FULL oracle pipeline's lists bit for bit; the first
#       scan whose lists differ must be one whose candidate clouds differ (otherwise the divergence is a bug, not a rounding flip).
def f32_ulp_distance(a, b):
    """element-wise distance in units in the last place between two float32 arrays"""
    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def clouds_within_rounding(a, b, d=1e-9):
    """two f32 clouds that are roundings of f64 clouds at most d apart: every coordinate within one float spacing + d"""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    spacing = np.spacing(np.maximum(np.abs(a), np.abs(b))).astype(np.float64)
    return bool((np.abs(a.astype(np.float64) - b.astype(np.float64)) <= spacing + d).all())


LIST_KEYS = ("new_vtx", "tri_add", "tri_rem", "tri_upd", "flip_add", "flip_upd", "smooth_ids")


def lists_equal(mo, mh):
    return mh["vtx_base"] == mo["vtx_base"] and all(np.array_equal(mh[k], mo[k]) for k in LIST_KEYS)


class ComposedRunChecker:
    def __init__(self, shadow_oracle, append_budget, compare_scan):
        self.shadow, self.budget, self.compare_scan = shadow_oracle, int(append_budget), compare_scan
        self.in_sync = True          # no mesher candidate has differed between the two pipelines so far
        self.n_exact = 0             # scans whose lists equal the FULL oracle pipeline's
        self.n_shadow_exact = 0      # scans whose lists equal the shadow oracle's (must be all of them)
        self.first_divergence = None
        self.flips = []              # per scan: (coordinates that differ by one ulp, candidates among them)

    def seed(self, world_xyzi, sensor_pos, frame_idx=0):
        """a scan both pipelines were handed as the SAME world-frame cloud (map seeding): the shadow follows"""
        self.shadow.mesh_scan(world_xyzi, sensor_pos, frame_idx=frame_idx)

    def check_scan(self, k, o, h, pose_h, mo, mh, pose_o=None, lever=None, pose_gap_bound=1e-8):
        wo, wh = o.mesh_world_scan(), h.mesh_world_scan()
        assert wo.shape == wh.shape and len(wh) > 0, k
        assert np.array_equal(wo[:, 3], wh[:, 3]), f"scan {k}: intensity channel differs"
        ulp = f32_ulp_distance(wh[:, :3], wo[:, :3])
        # (1) attribution.  Both clouds are f32(R p + t) of the SAME body points with two poses; two reals that are d apart round to floats at most
        # d + one spacing apart, and d <= |dt| + |dR| |p|.  (A plain "<= 1 ulp" is too strict only next to zero, where a float spacing is below d.)
        d = 0.0
        if pose_o is not None:
            dR = np.abs(np.asarray(pose_h[:9]) - np.asarray(pose_o[:9])).max()
            dt = np.abs(np.asarray(pose_h[9:12]) - np.asarray(pose_o[9:12])).max()
            d = dt + 3.0 * dR * (lever if lever is not None else 500.0)
            # (1e-8 for the short runs; a run of dozens of scans in which each side keeps its own state may drift further apart -- still three orders
            #  of magnitude inside the 1e-5 bar the caller asserts, and below one float spacing of the coordinates, which is what this bound is for)
            assert d < pose_gap_bound, f"scan {k}: poses differ by more than rounding ({d})"
        a64, b64 = wh[:, :3].astype(np.float64), wo[:, :3].astype(np.float64)
        spacing = np.spacing(np.maximum(np.abs(wh[:, :3]), np.abs(wo[:, :3]))).astype(np.float64)
        bad = np.abs(a64 - b64) > spacing + d
        assert not bad.any(), f"scan {k}: {int(bad.sum())} world-frame coordinates differ by more than one float spacing + the pose difference {d}"
        if pose_o is None:
            assert ulp.max() <= 1, f"scan {k}: a world-frame coordinate differs by {int(ulp.max())} ulp"
        step = max(1, int(round(len(wh) // self.budget)))     # ImMesh_mesh_reconstruction.cpp:111 (integer division first)
        pt_differs = (ulp > 0).any(axis=1)
        n_cand_flips = int(pt_differs[::step].sum())
        self.flips.append((int((ulp > 0).sum()), n_cand_flips))
        # (2) the oracle re-based on the device's own world-frame cloud: exact for every scan of the composed run
        ms = self.shadow.mesh_scan(wh, np.asarray(pose_h[9:12], np.float64), frame_idx=k)
        self.compare_scan(ms, mh, f"scan {k} (shadow oracle on the device's world-frame cloud)")
        self.n_shadow_exact += 1
        # (3) against the full oracle pipeline: exact until a candidate coordinate rounds the other way
        same = lists_equal(mo, mh)
        if self.in_sync and n_cand_flips == 0:
            assert same, f"scan {k}: lists differ although every candidate of every scan so far was bit-identical"
        if self.in_sync and not same:
            assert n_cand_flips > 0
            self.first_divergence = k
        if n_cand_flips > 0:
            self.in_sync = False
        if same and self.first_divergence is None:
            self.n_exact += 1
        return same

    def summary(self):
        return {"scans_equal_to_full_oracle": self.n_exact, "scans_equal_to_shadow_oracle": self.n_shadow_exact, "first_divergence": self.first_divergence,
                "ulp_flips_per_scan(coords,candidates)": self.flips}
