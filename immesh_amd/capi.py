"""ctypes binding of include/immesh_c_api.h.

``HotPath(lib, prefix)`` wraps one context.  The same wrapper drives the product (``prefix='immesh_'``,
libimmesh_hip.so) and -- from tests/bench only -- the CPU oracle (``prefix='orc_'``, oracle/liboracle.so), which
exports the identical entry points.  Nothing here computes anything.
"""
import ctypes as C
import os
import sys
import numpy as np

STATE_DOUBLES = 348
_HERE = os.path.dirname(os.path.abspath(__file__))


RUNTIME_CHOICE = None


def one_hip_runtime():
    """ONE HIP / HSA runtime in a test or bench process.

    The torch wheel bundles its own copies of libamdhip64 / libhsa-runtime64 and its libraries ask for them by the bare file names
    ``libamdhip64.so`` / ``libhsa-runtime64.so``; the product library is linked against the system copies (/opt/rocm) by soname.  The loader reuses
    an object that is already mapped under the name asked for, so WHO IS MAPPED FIRST decides, and this function fixes it:

    * single-GPU processes (the GPU test tier, smoke, ``bench.py --gpus 1``): the SYSTEM copies are mapped under exactly the names torch asks for
      BEFORE torch is imported, so torch runs on them too -- one runtime, and it is the one the reference's process would have (there is no torch
      there), with torch as the guest (measured on an MI355X box: device ops, a GEMM, a world-1 RCCL all-reduce -- tools/r05_one_runtime.py);
    * ranks of a multi-GPU job (WORLD_SIZE > 1): torch -- whose bundle of runtime + RCCL is the combination that is exercised together everywhere
      -- is imported first, and the library binds to ITS copy by soname: still one runtime, no mix of an RCCL built for one runtime on another
      where it cannot be tried beforehand (this container has no multi-GPU box);
    * torch already imported by the caller: its copy is the process's runtime; nothing to do.

    IMMESH_RUNTIME = system | torch | none overrides the choice: ``none`` leaves the loader alone -- for an application that embeds this
    binding next to its own torch and does not want the system ROCm mapped under torch's file names; ``torch`` / ``system`` force one of the two
    arrangements above.  This is loader policy of the TEST / BENCH binding only: the product (libimmesh_hip.so behind include/immesh_c_api.h, loaded by a
    C++ node) does none of it.

    Returns the runtime files mapped."""
    global RUNTIME_CHOICE
    forced = os.environ.get("IMMESH_RUNTIME", "").strip().lower()
    if forced == "none":
        RUNTIME_CHOICE = "untouched (IMMESH_RUNTIME=none)"
        return mapped_hip_runtimes()
    if forced == "torch" and "torch" not in sys.modules:
        RUNTIME_CHOICE = "torch (IMMESH_RUNTIME=torch)"
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        return mapped_hip_runtimes()
    if "torch" in sys.modules:
        RUNTIME_CHOICE = "torch (imported before the binding)"
    else:
        if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 and os.environ.get("IMMESH_SYSTEM_RUNTIME", "") != "1" and forced != "system":
            RUNTIME_CHOICE = "torch (rank of a multi-GPU job)"
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        else:
            RUNTIME_CHOICE = "system"
            for name in ("libhsa-runtime64.so", "libamdhip64.so"):
                try:
                    C.CDLL(name, mode=C.RTLD_GLOBAL)
                except OSError:
                    pass                  # no ROCm in the loader's cache (a CPU-only box): the library load below fails loudly on its own
    return mapped_hip_runtimes()


def mapped_hip_runtimes():
    out = set()
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                if "libamdhip64" in ln or "libhsa-runtime64" in ln:
                    out.add(ln.split()[-1])
    except OSError:
        pass
    return sorted(out)


one_hip_runtime()


class Config(C.Structure):
    _fields_ = [
        ("voxel_size", C.c_double), ("max_layer", C.c_int32), ("layer_init", C.c_int32 * 5), ("max_points_size", C.c_int32),
        ("planer_threshold", C.c_double), ("dept_err", C.c_double), ("beam_err", C.c_double), ("calib_laser", C.c_int32),
        ("sigma_num", C.c_double), ("max_iter", C.c_int32), ("extR", C.c_double * 9), ("extT", C.c_double * 3),
        ("mesh_min_spacing", C.c_double), ("mesh_voxel", C.c_double), ("mesh_region", C.c_double), ("mesh_append_budget", C.c_int32),
        ("device", C.c_int32), ("cap_root_voxels", C.c_int64), ("cap_nodes", C.c_int64), ("cap_point_chunks", C.c_int64),
        ("cap_vertices", C.c_int64), ("cap_triangles", C.c_int64), ("cap_scan_points", C.c_int64),
        ("shard_rank", C.c_int32), ("shard_world", C.c_int32), ("shard_brick_log2", C.c_int32), ("shard_mesh", C.c_int32),
        ("shard_scheme", C.c_int32), ("reserved_", C.c_int32),
    ]


class MeshSizes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vtx_base", "n_new_vtx", "n_add", "n_rem", "n_upd", "n_smooth", "n_voxels_meshed", "reserved")]


class PlaneRec(C.Structure):
    _fields_ = [("key", C.c_int64 * 3), ("layer", C.c_int32), ("path", C.c_int32), ("is_plane", C.c_int32), ("n_points", C.c_int32),
                ("update_enable", C.c_int32), ("new_points", C.c_int32), ("radius", C.c_float), ("min_eig", C.c_float), ("d", C.c_float),
                ("pad", C.c_float), ("center", C.c_double * 3), ("normal", C.c_double * 3), ("plane_var", C.c_double * 36)]


PLANE_DTYPE = np.dtype([("key", "<i8", 3), ("layer", "<i4"), ("path", "<i4"), ("is_plane", "<i4"), ("n_points", "<i4"),
                        ("update_enable", "<i4"), ("new_points", "<i4"), ("radius", "<f4"), ("min_eig", "<f4"), ("d", "<f4"),
                        ("pad", "<f4"), ("center", "<f8", 3), ("normal", "<f8", 3), ("plane_var", "<f8", 36)])
assert PLANE_DTYPE.itemsize == C.sizeof(PlaneRec)

COUNTER_FIELDS = ("n_ds", "n_iter", "n_match", "n_plane_tests", "n_extra_probe", "n_refits", "n_refit_pts", "n_app", "n_new", "v_act",
                  "n_v", "n_u", "t_v", "t_add", "t_rem", "c1", "c20", "n_root_voxels", "n_nodes", "n_vertices", "n_triangles_live", "n_degenerate_skips")


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 56), ("launches", C.c_int64), ("total_ms", C.c_double)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in COUNTER_FIELDS]


def avia_config(**over):
    """config/avia.yaml + launch/mapping_avia.launch (SURVEY.md section 8 constants table)."""
    c = Config()
    c.voxel_size = 0.5; c.max_layer = 2
    for i in range(5):
        c.layer_init[i] = 5
    c.max_points_size = 100; c.planer_threshold = 0.01; c.dept_err = 0.02; c.beam_err = 0.05; c.calib_laser = 0
    c.sigma_num = 3.0; c.max_iter = 4
    for i, v in enumerate([1, 0, 0, 0, 1, 0, 0, 0, 1]):
        c.extR[i] = v
    for i, v in enumerate([0.04165, 0.02326, -0.0284]):
        c.extT[i] = v
    c.mesh_min_spacing = 0.1; c.mesh_voxel = 0.4; c.mesh_region = 10.0; c.mesh_append_budget = 10000
    c.device = 0
    for k, v in over.items():
        setattr(c, k, v)
    return c


def velodyne_config(**over):
    """config/velodyne.yaml + launch/mapping_velody64.launch (KITTI)."""
    c = avia_config()
    c.voxel_size = 3.0; c.max_layer = 4; c.max_points_size = 1000; c.dept_err = 0.04; c.beam_err = 0.1; c.calib_laser = 1
    c.max_iter = 3
    for i in range(3):
        c.extT[i] = 0.0
    c.mesh_min_spacing = 0.1 * 1.5; c.mesh_voxel = 0.4 * 1.5; c.mesh_region = 10.0 * 1.5; c.mesh_append_budget = 10000
    for k, v in over.items():
        setattr(c, k, v)
    return c


def make_state(R=None, t=None, cov_diag=1e-7, vel=None, gravity=None):
    """StatesGroup() defaults: identity pose, cov = I * INIT_COV (include/common_lib.h:201-210)."""
    s = np.zeros(STATE_DOUBLES)
    s[0:9] = (np.eye(3) if R is None else np.asarray(R, float)).reshape(-1)
    if t is not None:
        s[9:12] = t
    if vel is not None:
        s[12:15] = vel
    if gravity is not None:
        s[21:24] = gravity
    s[24:] = (np.eye(18) * cov_diag).reshape(-1)
    return s


class ImuSample(C.Structure):
    """immesh_imu_sample"""
    _fields_ = [("t", C.c_double), ("gyr", C.c_double * 3), ("acc", C.c_double * 3)]


class ImuCtx(C.Structure):
    """immesh_imu_ctx: the ImuProcess members UndistortPcl carries from scan to scan"""
    _fields_ = [("last_lidar_end_time", C.c_double), ("acc_s_last", C.c_double * 3), ("angvel_last", C.c_double * 3), ("last_imu", ImuSample),
                ("mean_acc_norm", C.c_double), ("cov_gyr", C.c_double * 3), ("cov_acc", C.c_double * 3), ("cov_bias_gyr", C.c_double * 3),
                ("cov_bias_acc", C.c_double * 3), ("lid_rot_to_imu", C.c_double * 9), ("lid_offset_to_imu", C.c_double * 3)]


def make_imu_ctx(cfg, t0=0.0, gyr0=(0, 0, 0), acc0=(0, 0, 9.81)):
    """ImuProcess after IMU_init (src/IMU_Processing.cpp:56-80, 186-230): noise parameters of config/avia.yaml, extrinsics of cfg."""
    ic = ImuCtx()
    ic.last_lidar_end_time = t0
    ic.last_imu.t = t0
    for a in range(3):
        ic.last_imu.gyr[a] = gyr0[a]; ic.last_imu.acc[a] = acc0[a]
        ic.cov_gyr[a] = 0.3; ic.cov_acc[a] = 0.5; ic.cov_bias_gyr[a] = 0.0001; ic.cov_bias_acc[a] = 0.0001
        ic.lid_offset_to_imu[a] = cfg.extT[a]
    ic.mean_acc_norm = float(np.linalg.norm(acc0))
    for i in range(9):
        ic.lid_rot_to_imu[i] = cfg.extR[i]
    return ic


def forward_without_imu_native(lib, state, dt=0.1, cov_gyr=0.3, cov_acc=0.5):
    """immesh_forward_without_imu (C++ host code in the product library); same arithmetic as synth.forward_without_imu."""
    f = lib.immesh_forward_without_imu
    f.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]; f.restype = C.c_int
    out = np.empty(STATE_DOUBLES)
    s = np.ascontiguousarray(state, dtype=np.float64)
    if f(s.ctypes.data_as(C.c_void_p), dt, cov_gyr, cov_acc, out.ctypes.data_as(C.c_void_p)) != 0:
        raise RuntimeError("immesh_forward_without_imu failed")
    return out


def shard_owner(lib, cfg, key3):
    f = lib.immesh_shard_owner; f.argtypes = [C.POINTER(Config), C.c_void_p]; f.restype = C.c_int
    k = np.ascontiguousarray(key3, dtype=np.int64)
    return f(C.byref(cfg), k.ctypes.data_as(C.c_void_p))


def hip_library_path():
    # IMMESH_HIP_LIBRARY: an A/B measurement against a library built from another commit (tools/r06_ab.sh); never set by tests or the default bench
    return os.environ.get("IMMESH_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libimmesh_hip.so")


def load_hip_library():
    """Load the product library.  Raises (never falls back) when it has not been built."""
    p = hip_library_path()
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C immesh_amd/csrc). "
                           "There is no CPU fallback for the hot path.")
    return C.CDLL(p)


def _ptr(a):
    """host ndarray or raw device pointer (int) -> c_void_p"""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    return a.ctypes.data_as(C.c_void_p)


class HotPath:
    def __init__(self, lib, cfg, prefix="immesh_"):
        self.lib, self.prefix, self.cfg = lib, prefix, cfg
        self._f = lambda name: getattr(lib, prefix + name)
        cr = self._f("create"); cr.restype = C.c_void_p; cr.argtypes = [C.POINTER(Config)]
        self.ctx = cr(C.byref(cfg))
        if not self.ctx:
            msg = ""
            if prefix == "immesh_":
                lib.immesh_create_error.restype = C.c_char_p
                msg = lib.immesh_create_error().decode()
            raise RuntimeError(f"{prefix}create failed: {msg}")
        self.ctx = C.c_void_p(self.ctx)

    def close(self):
        if self.ctx:
            d = self._f("destroy"); d.argtypes = [C.c_void_p]; d.restype = None
            d(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = ""
            if self.prefix == "immesh_":
                f = self.lib.immesh_last_error; f.restype = C.c_char_p; f.argtypes = [C.c_void_p]
                msg = f(self.ctx).decode()
            raise RuntimeError(f"{self.prefix}{what} failed rc={rc}: {msg}")

    # -- registration ------------------------------------------------------------------------------------------
    def map_build(self, pts_body_xyz, state, n=None):
        f = self._f("map_build"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]; f.restype = C.c_int
        n = len(pts_body_xyz) if n is None else n
        self._check(f(self.ctx, _ptr(pts_body_xyz), n, _ptr(state)), "map_build")

    def register(self, pts_down, state_prior, state, n=None, want_eff=False):
        f = self._f("register"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        n = len(pts_down) if n is None else n
        out = np.array(state, dtype=np.float64, copy=True)
        n_iter, n_match, res = C.c_int32(0), C.c_int32(0), C.c_double(0)
        eff_p = np.zeros((n, 3), np.float32) if want_eff else None
        eff_n = np.zeros((n, 4), np.float32) if want_eff else None
        self._check(f(self.ctx, _ptr(pts_down), n, _ptr(np.ascontiguousarray(state_prior, dtype=np.float64)), _ptr(out), C.byref(n_iter),
                      C.byref(n_match), C.byref(res), _ptr(eff_p), _ptr(eff_n)), "register")
        info = {"n_iter": n_iter.value, "n_match": n_match.value, "res_mean": res.value}
        if want_eff:
            info["eff_pts"] = eff_p[:n_match.value]; info["eff_norm_dis"] = eff_n[:n_match.value]
        return out, info

    def residuals(self, pts_down, state, n=None):
        f = self._f("residuals"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p] + [C.c_void_p] * 7
        n = len(pts_down) if n is None else n
        HTH, HTz, nm = np.zeros(36), np.zeros(6), C.c_int32(0)
        idx, nrm, dis, rinv = np.zeros(n, np.int32), np.zeros((n, 3)), np.zeros(n, np.float32), np.zeros(n)
        self._check(f(self.ctx, _ptr(pts_down), n, _ptr(np.ascontiguousarray(state, dtype=np.float64)), _ptr(HTH), _ptr(HTz), C.byref(nm),
                      _ptr(idx), _ptr(nrm), _ptr(dis), _ptr(rinv)), "residuals")
        m = nm.value
        return {"HTH": HTH.reshape(6, 6), "HTz": HTz, "n_match": m, "match_idx": idx[:m], "normals": nrm[:m], "dis": dis[:m], "r_inv": rinv[:m]}

    def map_update(self, pts_down, state, n=None):
        f = self._f("map_update"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]; f.restype = C.c_int
        n = len(pts_down) if n is None else n
        self._check(f(self.ctx, _ptr(pts_down), n, _ptr(np.ascontiguousarray(state, dtype=np.float64))), "map_update")

    # -- meshing -----------------------------------------------------------------------------------------------
    def mesh_scan(self, pts_world_xyzi, sensor_pos, frame_idx=0, n=None, fetch=True):
        f = self._f("mesh_scan"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]; f.restype = C.c_int
        n = len(pts_world_xyzi) if n is None else n
        self._check(f(self.ctx, _ptr(pts_world_xyzi), n, _ptr(np.ascontiguousarray(sensor_pos, dtype=np.float64)), frame_idx), "mesh_scan")
        return self.mesh_fetch() if fetch else None

    def reconstruct_mesh_from_pointcloud(self, pts_xyzi, leaf=0.01):
        f = self._f("reconstruct_mesh_from_pointcloud"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double]; f.restype = C.c_int
        self._check(f(self.ctx, _ptr(pts_xyzi), len(pts_xyzi), leaf), "reconstruct_mesh_from_pointcloud")
        return self.mesh_fetch()

    def reconstruct_mesh_from_pointcloud_dev(self, dev_ptr, n, leaf=0.01):
        """the offline entry on a device-resident (n x 4 float32) cloud; the result lists are not fetched"""
        f = self._f("reconstruct_mesh_from_pointcloud"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double]; f.restype = C.c_int
        self._check(f(self.ctx, dev_ptr, n, leaf), "reconstruct_mesh_from_pointcloud")

    def mesh_neighbourhood_sizes(self):
        """n_u of every voxel the newest finished mesh job triangulated (diagnostics; HIP library only)."""
        f = self._f("mesh_neighbourhood_sizes"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]; f.restype = C.c_int
        n = C.c_int32(0)
        self._check(f(self.ctx, None, 0, C.byref(n)), "mesh_neighbourhood_sizes")
        out = np.zeros(max(1, n.value), np.int32)
        self._check(f(self.ctx, _ptr(out), len(out), C.byref(n)), "mesh_neighbourhood_sizes")
        return out[:n.value]

    def mesh_world_scan(self):
        """the world-frame scan (n x 4 float32) the newest finished mesh job was handed (parity diagnostics)"""
        f = self._f("mesh_world_scan"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]; f.restype = C.c_int
        n = C.c_int32(0)
        self._check(f(self.ctx, None, 0, C.byref(n)), "mesh_world_scan")
        out = np.zeros((n.value, 4), np.float32)
        if n.value:
            self._check(f(self.ctx, _ptr(out), n.value, C.byref(n)), "mesh_world_scan")
        return out

    def mesh_wait(self):
        f = self._f("mesh_wait"); f.argtypes = [C.c_void_p]; f.restype = C.c_int
        self._check(f(self.ctx), "mesh_wait")

    def mesh_fetch(self):
        fs = self._f("mesh_sizes"); fs.argtypes = [C.c_void_p, C.POINTER(MeshSizes)]; fs.restype = C.c_int
        s = MeshSizes()
        self._check(fs(self.ctx, C.byref(s)), "mesh_sizes")
        r = {"vtx_base": s.vtx_base, "n_voxels_meshed": s.n_voxels_meshed,
             "new_vtx": np.zeros((s.n_new_vtx, 3), np.float32), "tri_add": np.zeros((s.n_add, 3), np.int32), "flip_add": np.zeros(s.n_add, np.uint8),
             "tri_rem": np.zeros((s.n_rem, 3), np.int32), "tri_upd": np.zeros((s.n_upd, 3), np.int32), "flip_upd": np.zeros(s.n_upd, np.uint8),
             "smooth_ids": np.zeros(s.n_smooth, np.int32), "smooth_xyz": np.zeros((s.n_smooth, 3), np.float64)}
        ff = self._f("mesh_fetch"); ff.argtypes = [C.c_void_p] + [C.c_void_p] * 8; ff.restype = C.c_int
        self._check(ff(self.ctx, _ptr(r["new_vtx"]), _ptr(r["tri_add"]), _ptr(r["flip_add"]), _ptr(r["tri_rem"]), _ptr(r["tri_upd"]),
                       _ptr(r["flip_upd"]), _ptr(r["smooth_ids"]), _ptr(r["smooth_xyz"])), "mesh_fetch")
        return r

    # -- mesh export ------------------------------------------------------------------------------------------
    def mesh_export(self, smooth_factor=1.0, knn=20):
        f = self._f("mesh_export"); f.argtypes = [C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]; f.restype = C.c_int
        nv, nf = C.c_int64(0), C.c_int64(0)
        self._check(f(self.ctx, smooth_factor, knn, C.byref(nv), C.byref(nf)), "mesh_export")
        vtx, faces = np.zeros((nv.value, 3), np.float32), np.zeros((nf.value, 3), np.int32)
        g = self._f("mesh_export_fetch"); g.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; g.restype = C.c_int
        self._check(g(self.ctx, _ptr(vtx), _ptr(faces)), "mesh_export_fetch")
        return vtx, faces

    def smooth_pts(self, ids, smooth_factor=1.0, knn=20, maximum_smooth_dis=0.0):
        """Global_map::smooth_pts for a batch of vertex ids -> (n, 3) float64"""
        f = self._f("smooth_pts"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_double, C.c_void_p]; f.restype = C.c_int
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        out = np.zeros((len(ids), 3), np.float64)
        self._check(f(self.ctx, _ptr(ids), len(ids), smooth_factor, knn, maximum_smooth_dis, _ptr(out)), "smooth_pts")
        return out

    def mesh_display_vertices(self, ids, smooth_factor=1.0, knn=20, maximum_smooth_dis=0.0):
        """get_pos(1) after the renderer's on-demand smoothing, as floats -> (n, 3) float32"""
        f = self._f("mesh_display_vertices"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_double, C.c_void_p]; f.restype = C.c_int
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        out = np.zeros((len(ids), 3), np.float32)
        self._check(f(self.ctx, _ptr(ids), len(ids), smooth_factor, knn, maximum_smooth_dis, _ptr(out)), "mesh_display_vertices")
        return out

    def save_ply(self, path, smooth_factor=1.0, knn=20):
        f = self._f("save_ply"); f.argtypes = [C.c_void_p, C.c_char_p, C.c_double, C.c_int32]; f.restype = C.c_int
        self._check(f(self.ctx, path.encode(), smooth_factor, knn), "save_ply")

    def process_scan_strided(self, down_bytes, n_ds, down_stride, raw_bytes, n_raw, raw_stride, raw_int_off, state_prior, state, frame_idx=0, do_mesh=True):
        """immesh_process_scan_strided: down_bytes / raw_bytes = numpy arrays (any dtype) or device pointers holding the pcl-shaped clouds"""
        f = self._f("process_scan_strided"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        out = np.array(state, dtype=np.float64, copy=True)
        n_iter, n_match = C.c_int32(0), C.c_int32(0)
        self._check(f(self.ctx, _ptr(down_bytes), n_ds, down_stride, _ptr(raw_bytes), n_raw, raw_stride, raw_int_off, _ptr(np.ascontiguousarray(state_prior, dtype=np.float64)),
                      _ptr(out), frame_idx, int(do_mesh), C.byref(n_iter), C.byref(n_match)), "process_scan_strided")
        return out, {"n_iter": n_iter.value, "n_match": n_match.value}

    # -- whole scan -------------------------------------------------------------------------------------------
    def process_scan(self, pts_down, pts_raw_xyzi, state_prior, state, frame_idx=0, do_mesh=True, n_ds=None, n_raw=None):
        f = self._f("process_scan"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        n_ds = len(pts_down) if n_ds is None else n_ds
        n_raw = len(pts_raw_xyzi) if n_raw is None else n_raw
        out = np.array(state, dtype=np.float64, copy=True)
        n_iter, n_match = C.c_int32(0), C.c_int32(0)
        self._check(f(self.ctx, _ptr(pts_down), n_ds, _ptr(pts_raw_xyzi), n_raw, _ptr(np.ascontiguousarray(state_prior, dtype=np.float64)),
                      _ptr(out), frame_idx, int(do_mesh), C.byref(n_iter), C.byref(n_match)), "process_scan")
        return out, {"n_iter": n_iter.value, "n_match": n_match.value}

    def last_timing(self):
        f = self._f("last_timing"); f.argtypes = [C.c_void_p, C.c_void_p]; f.restype = C.c_int
        ms = np.zeros(4, np.float32)
        self._check(f(self.ctx, _ptr(ms)), "last_timing")
        return {"total": float(ms[0]), "register": float(ms[1]), "map_update": float(ms[2]), "mesh": float(ms[3])}

    # -- the stage before the path: VoxelGrid down-sampling ----------------------------------------------------
    def downsample(self, pts, leaf, n=None, stride=None, to_host=True):
        """pts: host ndarray (n x 3 or n x 4 float32) or device pointer (then pass n and stride).  Returns (host array | None, n_out)."""
        f = self._f("downsample"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_void_p]
        if n is None:
            n, stride = pts.shape[0], pts.shape[1]
        out = np.zeros((n, 3), np.float32) if to_host else None
        n_out = C.c_int32(0)
        self._check(f(self.ctx, _ptr(pts), n, stride, leaf, _ptr(out), n, C.byref(n_out)), "downsample")
        return (out[:n_out.value] if to_host else None), n_out.value

    def decode_livox(self, wire_points, n_scans=6, point_filter_num=1, blind=1.0, to_host=True):
        """avia_handler: wire_points = n x 19 uint8 (serialised livox CustomPoint).  Returns (n_out x 5 float32 | None, n_out)."""
        f = self._f("decode_livox"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p]
        w = np.ascontiguousarray(wire_points, dtype=np.uint8).reshape(-1, 19)
        out = np.zeros((len(w), 5), np.float32) if to_host else None
        n_out = C.c_int32(0)
        self._check(f(self.ctx, _ptr(w), len(w), n_scans, point_filter_num, blind, _ptr(out), C.byref(n_out)), "decode_livox")
        return (out[:n_out.value] if to_host else None), n_out.value

    def decode_velodyne(self, data, point_step, offsets, n_scans=64, to_host=True):
        """velodyne_handler: data = n x point_step uint8 (PointCloud2.data); offsets = byte offsets of x, y, z, intensity."""
        f = self._f("decode_velodyne"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p, C.c_void_p]
        d = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, point_step)
        out = np.zeros((len(d), 5), np.float32) if to_host else None
        n_out = C.c_int32(0)
        self._check(f(self.ctx, _ptr(d), len(d), point_step, offsets[0], offsets[1], offsets[2], offsets[3], n_scans, _ptr(out), C.byref(n_out)), "decode_velodyne")
        return (out[:n_out.value] if to_host else None), n_out.value

    def decode_result_ptr(self):
        f = self._f("decode_result"); f.argtypes = [C.c_void_p]; f.restype = C.c_void_p
        return f(self.ctx)

    def undistort(self, pts_xyzit, imu, lidar_beg_time, last_update_time, imu_ctx, state, to_host=True):
        """UndistortPcl.  pts_xyzit: n x 5 float32 (x y z intensity offset_ms); imu: m x 7 float64 (t, gyr, acc).
        Returns (cloud n x 4 in time order | None, propagated state, last_update_time); imu_ctx is updated in place."""
        f = self._f("undistort"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        imu = np.ascontiguousarray(imu, dtype=np.float64).reshape(-1, 7)
        out = np.zeros((len(pts_xyzit), 4), np.float32) if to_host else None
        st = np.array(state, dtype=np.float64, copy=True)
        lut = C.c_double(last_update_time)
        self._check(f(self.ctx, _ptr(np.ascontiguousarray(pts_xyzit, dtype=np.float32)), len(pts_xyzit), _ptr(imu), len(imu), lidar_beg_time,
                      C.byref(lut), C.byref(imu_ctx), _ptr(st), _ptr(out)), "undistort")
        return out, st, lut.value

    def undistort_result_ptr(self):
        f = self._f("undistort_result"); f.argtypes = [C.c_void_p]; f.restype = C.c_void_p
        return f(self.ctx)

    # -- legacy registration path ------------------------------------------------------------
    def ikd_build(self, pts_world_xyz, downsample_size):
        f = self._f("ikd_build"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double]; f.restype = C.c_int
        p = np.ascontiguousarray(pts_world_xyz, dtype=np.float32)
        self._check(f(self.ctx, _ptr(p), len(p), downsample_size), "ikd_build")

    def ikd_add_points(self, pts_world_xyz):
        f = self._f("ikd_add_points"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]; f.restype = C.c_int
        p = np.ascontiguousarray(pts_world_xyz, dtype=np.float32)
        self._check(f(self.ctx, _ptr(p), len(p)), "ikd_add_points")

    def ikd_delete_boxes(self, boxes):
        f = self._f("ikd_delete_boxes"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]; f.restype = C.c_int
        b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
        n = C.c_int32(0)
        self._check(f(self.ctx, _ptr(b), len(b), C.byref(n)), "ikd_delete_boxes")
        return n.value

    def ikd_fov_segment(self, pos_lid, cube_len, detection_range):
        f = self._f("ikd_fov_segment"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]; f.restype = C.c_int
        n = C.c_int32(0)
        self._check(f(self.ctx, _ptr(np.ascontiguousarray(pos_lid, dtype=np.float64)), cube_len, detection_range, C.byref(n)), "ikd_fov_segment")
        return n.value

    def ikd_size(self):
        f = self._f("ikd_size"); f.argtypes = [C.c_void_p, C.c_void_p]; f.restype = C.c_int
        n = C.c_int64(0)
        self._check(f(self.ctx, C.byref(n)), "ikd_size")
        return n.value

    def ikd_dump(self):
        """all map points, sorted lexicographically by (x, y, z)"""
        f = self._f("ikd_dump"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]; f.restype = C.c_int
        cap = self.ikd_size()
        out = np.zeros((max(cap, 1), 3), np.float32)
        n = C.c_int64(0)
        self._check(f(self.ctx, _ptr(out), cap, C.byref(n)), "ikd_dump")
        out = out[:min(cap, n.value)]
        return out[np.lexsort((out[:, 2], out[:, 1], out[:, 0]))]

    def ikd_knn(self, q_xyz):
        f = self._f("ikd_knn"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
        q = np.ascontiguousarray(q_xyz, dtype=np.float32)
        nn = np.zeros((len(q), 5, 3), np.float32); d2 = np.zeros((len(q), 5), np.float32); nf = np.zeros(len(q), np.int32)
        self._check(f(self.ctx, _ptr(q), len(q), _ptr(nn), _ptr(d2), _ptr(nf)), "ikd_knn")
        return nn, d2, nf

    def ikd_register(self, pts_down_body, state_prior, state, laser_point_cov=0.001):
        f = self._f("ikd_register"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        p = np.ascontiguousarray(pts_down_body, dtype=np.float32)
        n = len(p)
        out = np.array(state, dtype=np.float64, copy=True)
        n_iter, n_match, res = C.c_int32(0), C.c_int32(0), C.c_double(0)
        idx = np.zeros(n, np.int32); nv = np.zeros((n, 4), np.float32)
        self._check(f(self.ctx, _ptr(p), n, _ptr(np.ascontiguousarray(state_prior, dtype=np.float64)), _ptr(out), laser_point_cov, C.byref(n_iter),
                      C.byref(n_match), C.byref(res), _ptr(idx), _ptr(nv)), "ikd_register")
        m = n_match.value
        return out, {"n_iter": n_iter.value, "n_match": m, "res_mean": res.value, "match_idx": idx[:m].copy(), "normals_pd2": nv[:m].copy()}

    def downsample_begin(self, pts, leaf, n=None, stride=None):
        """enqueue the VoxelGrid of a (host array | device pointer) cloud on the pre-processing stream; collect with downsample_end()"""
        f = self._f("downsample_begin"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double]; f.restype = C.c_int
        if n is None:
            n, stride = pts.shape[0], pts.shape[1]
        self._ds_keep = pts       # (a host array has to outlive the staging copy)
        self._check(f(self.ctx, _ptr(pts), n, stride, leaf), "downsample_begin")

    def downsample_end(self):
        """-> (n_out, pointer of the n_out x 3 float32 result: device memory for the HIP library, host memory for the oracle)"""
        f = self._f("downsample_end"); f.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]; f.restype = C.c_int
        n = C.c_int32(0); ptr = C.c_void_p(0)
        self._check(f(self.ctx, C.byref(n), C.byref(ptr)), "downsample_end")
        return n.value, ptr.value

    def inputs_consumed(self):
        """block until the last asynchronous process_scan has consumed its (device-resident) input clouds"""
        f = self._f("inputs_consumed"); f.argtypes = [C.c_void_p]; f.restype = C.c_int
        self._check(f(self.ctx), "inputs_consumed")

    def last_matches(self):
        """(matched body points n x 3, normal + residual n x 4) of the last registration on the context"""
        f = self._f("last_matches"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]; f.restype = C.c_int
        n = C.c_int32(0)
        self._check(f(self.ctx, None, None, 0, C.byref(n)), "last_matches")
        p, q = np.zeros((n.value, 3), np.float32), np.zeros((n.value, 4), np.float32)
        if n.value:
            self._check(f(self.ctx, _ptr(p), _ptr(q), n.value, C.byref(n)), "last_matches")
        return p, q

    def mesh_collect_enable(self, on=True):
        f = self._f("mesh_collect_enable"); f.argtypes = [C.c_void_p, C.c_int32]; f.restype = C.c_int
        self._check(f(self.ctx, 1 if on else 0), "mesh_collect_enable")

    def mesh_collect_begin(self, timeout_ms=1000):
        """-> job ordinal, or None when no job finished within the timeout"""
        f = self._f("mesh_collect_begin"); f.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]; f.restype = C.c_int
        o = C.c_int64(0)
        rc = f(self.ctx, timeout_ms, C.byref(o))
        if rc == 1:
            return None
        self._check(rc, "mesh_collect_begin")
        return o.value

    def mesh_collect_end(self):
        f = self._f("mesh_collect_end"); f.argtypes = [C.c_void_p]; f.restype = C.c_int
        self._check(f(self.ctx), "mesh_collect_end")

    def registration_fallbacks(self):
        f = self._f("registration_fallbacks"); f.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]; f.restype = C.c_int
        n = C.c_int64(0)
        self._check(f(self.ctx, C.byref(n)), "registration_fallbacks")
        return n.value

    def downsample_result_ptr(self):
        f = self._f("downsample_result"); f.argtypes = [C.c_void_p]; f.restype = C.c_void_p
        return f(self.ctx)

    # -- multi-GPU sharding ------------------------------------------------------------------------------------
    def set_allreduce(self, fn):
        """fn(np.ndarray float64 view of the library's buffer) must sum it over all ranks IN PLACE (e.g. torch.distributed.all_reduce)."""
        proto = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int32, C.c_void_p)

        def _cb(buf, n, user):
            try:
                fn(np.ctypeslib.as_array(buf, shape=(n,)))
                return 0
            except Exception:   # never raise through the C frame
                return -1
        self._allreduce_cb = proto(_cb)   # keep alive
        f = self._f("set_allreduce"); f.argtypes = [C.c_void_p, proto, C.c_void_p]; f.restype = C.c_int
        self._check(f(self.ctx, self._allreduce_cb, None), "set_allreduce")

    def stub_collectives(self):
        f = self._f("stub_collectives"); f.argtypes = [C.c_void_p]; f.restype = C.c_int
        self._check(f(self.ctx), "stub_collectives")

    def device_bytes(self):
        f = self._f("device_bytes"); f.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]; f.restype = C.c_int
        b = C.c_int64(0)
        self._check(f(self.ctx, C.byref(b)), "device_bytes")
        return b.value

    # -- RCCL inside the library (sharded contexts) ----------------------------------------------------------------
    def rccl_unique_id(self):
        f = self._f("rccl_unique_id"); f.argtypes = [C.c_void_p]; f.restype = C.c_int
        uid = np.zeros(128, np.uint8)
        rc = f(_ptr(uid))
        if rc != 0:
            e = self._f("rccl_error"); e.restype = C.c_char_p
            raise RuntimeError(f"immesh_rccl_unique_id failed rc={rc}: {e().decode()}")
        return uid

    def rccl_init(self, uid):
        f = self._f("rccl_init"); f.argtypes = [C.c_void_p, C.c_void_p]; f.restype = C.c_int
        self._check(f(self.ctx, _ptr(np.ascontiguousarray(uid, np.uint8))), "rccl_init")

    def set_threads(self, mesher_threads, matcher_threads):
        """oracle only (CPU baseline leg): the reference's own threading -- a 12-thread pool over mesh voxels, 4 OpenMP threads in the matcher"""
        f = self._f("set_threads"); f.argtypes = [C.c_void_p, C.c_int32, C.c_int32]; f.restype = C.c_int
        self._check(f(self.ctx, mesher_threads, matcher_threads), "set_threads")

    def set_allgather(self, fn):
        """Sharded mesher.  fn(send: uint8 view [nbytes], recv: uint8 view [world * nbytes]) gathers every rank's `send` into `recv` in
        rank order (e.g. torch.distributed.all_gather_into_tensor)."""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
        world = max(1, int(self.cfg.shard_world))

        def _cb(send, nbytes, recv, user):
            try:
                s = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                r = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
                fn(s, r)
                return 0
            except Exception:   # never raise through the C frame
                import traceback
                traceback.print_exc()
                return -1
        self._allgather_cb = proto(_cb)   # keep alive
        f = self._f("set_allgather"); f.argtypes = [C.c_void_p, proto, C.c_void_p]; f.restype = C.c_int
        self._check(f(self.ctx, self._allgather_cb, None), "set_allgather")

    def broadcast_scan(self, pts, root=0, stride=None, n=None):
        """immesh_broadcast_scan: collective; the root passes the scan (ndarray n x 3 / n x 4 or a device pointer with n and stride), the others None.
        Returns (device pointer, points) of the scan in this context's memory."""
        f = self._f("broadcast_scan"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        if pts is not None and not isinstance(pts, (int, np.integer)):
            pts = np.ascontiguousarray(pts, np.float32)
            n, stride = len(pts), pts.shape[1]
        out, n_out = C.c_void_p(0), C.c_int32(0)
        self._check(f(self.ctx, _ptr(pts), int(n or 0), int(stride or 0), int(root), C.byref(out), C.byref(n_out)), "broadcast_scan")
        return int(out.value), int(n_out.value)

    def shard_traffic(self):
        f = self._f("shard_traffic"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
        b, n = C.c_int64(0), C.c_int64(0)
        self._check(f(self.ctx, C.byref(b), C.byref(n)), "shard_traffic")
        return {"bytes": b.value, "calls": n.value}

    # -- per-kernel timing (product library only) --------------------------------------------------------------
    def profile_enable(self, on=True):
        f = self._f("profile_enable"); f.argtypes = [C.c_void_p, C.c_int32]; f.restype = C.c_int
        self._check(f(self.ctx, 1 if on else 0), "profile_enable")

    def profile_read(self, reset=False):
        f = self._f("profile_read"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]; f.restype = C.c_int
        buf = (KernelStat * 128)()
        n = C.c_int32(0)
        self._check(f(self.ctx, C.byref(buf), 128, C.byref(n), 1 if reset else 0), "profile_read")
        return {buf[i].name.decode(): {"launches": buf[i].launches, "total_ms": buf[i].total_ms} for i in range(min(n.value, 128))}

    # -- introspection ----------------------------------------------------------------------------------------
    def dump_planes(self, cap=None):
        """all initialised octree nodes (unordered); cap = at most that many records (a sample of a big map)"""
        f = self._f("dump_planes"); f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]; f.restype = C.c_int
        n = C.c_int64(0)
        self._check(f(self.ctx, None, 0, C.byref(n)), "dump_planes")
        m = n.value if cap is None else min(n.value, int(cap))
        recs = np.zeros(m, PLANE_DTYPE)
        if m:
            self._check(f(self.ctx, _ptr(recs), m, C.byref(n)), "dump_planes")
        return recs

    def counters(self, reset=False):
        f = self._f("counters"); f.argtypes = [C.c_void_p, C.POINTER(Counters), C.c_int32]; f.restype = C.c_int
        c = Counters()
        self._check(f(self.ctx, C.byref(c), 1 if reset else 0), "counters")
        return {n: getattr(c, n) for n in COUNTER_FIELDS}
