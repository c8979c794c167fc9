"""Host mirror built on the REFERENCE'S OWN Triangle_manager (oracle/_ref/libref_triangle.so, compiled from /root/reference/src/meshing/r3live/
triangle.{hpp,cpp} + src/tools/tools_kd_hash.hpp): applies a path's per-scan diff lists exactly as incremental_mesh_reconstruction's
"Voxel-wise mesh push" does (src/ImMesh_mesh_reconstruction.cpp:228-244) and reads the live set back.  Test infrastructure."""
import ctypes as C

import numpy as np


class RefTriangleMirror:
    def __init__(self, lib, region_size=10.0):
        self.lib = lib
        self.ctx = C.c_void_p(lib.rt_create(float(region_size)))

    def close(self):
        if self.ctx:
            self.lib.rt_destroy(self.ctx); self.ctx = None

    def apply(self, m, frame_idx):
        """m = HotPath.mesh_fetch() of one scan.  Returns the number of removals that named an unknown triangle (must be 0)."""
        L = self.lib
        nv = np.ascontiguousarray(m["new_vtx"], np.float32)
        if len(nv):
            L.rt_append_vertices(self.ctx, nv.ctypes.data_as(C.c_void_p), len(nv))
        rem = np.ascontiguousarray(m["tri_rem"], np.int32); add = np.ascontiguousarray(m["tri_add"], np.int32)
        fa = np.ascontiguousarray(m["flip_add"], np.uint8)
        unknown = L.rt_commit(self.ctx, rem.ctypes.data_as(C.c_void_p), len(rem), add.ctypes.data_as(C.c_void_p), fa.ctypes.data_as(C.c_void_p), len(add), frame_idx)
        upd = np.ascontiguousarray(m["tri_upd"], np.int32); fu = np.ascontiguousarray(m["flip_upd"], np.uint8)
        if len(upd):
            L.rt_set_flips(self.ctx, upd.ctypes.data_as(C.c_void_p), fu.ctypes.data_as(C.c_void_p), len(upd))
        return int(unknown)

    def live(self):
        n = int(self.lib.rt_live_size(self.ctx))
        tri = np.zeros((max(n, 1), 3), np.int32); flip = np.zeros(max(n, 1), np.uint8)
        n2 = int(self.lib.rt_live(self.ctx, tri.ctypes.data_as(C.c_void_p), flip.ctypes.data_as(C.c_void_p), n))
        assert n2 == n
        return {(int(a), int(b), int(c)): int(f) for (a, b, c), f in zip(tri[:n], flip[:n])}

    def find_relative(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        out = np.zeros((max(8 * len(ids), 64), 3), np.int32)
        k = int(self.lib.rt_find_relative(self.ctx, ids.ctypes.data_as(C.c_void_p), len(ids), out.ctypes.data_as(C.c_void_p), len(out)))
        assert k <= len(out)
        return set(map(tuple, out[:k].tolist()))
