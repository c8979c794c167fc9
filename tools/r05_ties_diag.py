"""round 5 diagnostic (GPU): small lattice patches through the HIP mesher and the checker, one context pair per patch; prints the first patches whose
triangle lists differ, with the points and both lists (exact cocircular / collinear input)."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from immesh_amd import capi

hip = capi.load_hip_library()
orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
rng = np.random.default_rng(3)
cam = np.array([8.0, 0.0, 3.0])
shown = 0
n_bad = 0
for trial in range(120):
    spacing = [0.25, 0.125][trial % 2]
    nx, ny = rng.integers(3, 7, 2)
    nz = int(rng.integers(1, 3)) if spacing == 0.125 else 1
    gx, gy, gz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    p = np.stack([6.0 + gx.ravel() * spacing, -1.0 + gy.ravel() * spacing, gz.ravel() * spacing], axis=1)
    keep = rng.random(len(p)) < rng.choice([1.0, 0.85])
    p = p[keep].astype(np.float32)
    if len(p) < 4:
        continue
    pts = np.ascontiguousarray(np.concatenate([p, np.ones((len(p), 1), np.float32)], axis=1))
    cfg = capi.avia_config(cap_root_voxels=1 << 10, cap_scan_points=4096, cap_vertices=1 << 12, cap_triangles=1 << 14)
    o, h = capi.HotPath(orc, cfg, "orc_"), capi.HotPath(hip, cfg, "immesh_")
    mo = o.mesh_scan(pts, cam, frame_idx=0); mh = h.mesh_scan(pts, cam, frame_idx=0)
    same = np.array_equal(mo["tri_add"], mh["tri_add"]) and np.array_equal(mo["new_vtx"], mh["new_vtx"])
    nuo, nuh = o.mesh_neighbourhood_sizes(), h.mesh_neighbourhood_sizes()
    if not same:
        n_bad += 1
        if shown < 4:
            shown += 1
            so, sh = set(map(tuple, mo["tri_add"].tolist())), set(map(tuple, mh["tri_add"].tolist()))
            print(f"=== trial {trial} spacing {spacing} grid {nx}x{ny}x{nz} n={len(p)} new_vtx {len(mo['new_vtx'])}/{len(mh['new_vtx'])} n_u oracle {nuo.tolist()} hip {nuh.tolist()}")
            print("points:", p.tolist())
            print("only oracle:", sorted(so - sh))
            print("only hip   :", sorted(sh - so))
    o.close(); h.close()
print("patches that differ:", n_bad)
