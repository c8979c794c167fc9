import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from immesh_amd import capi  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    """CPU oracle (test infrastructure).  Built on demand; g++ only."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".hpp", ".cpp"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return C.CDLL(so)


@pytest.fixture(scope="session")
def ref_ikd_lib():
    """The reference's own ikd-Tree compiled from /root/reference (oracle/_ref).  Absent on the GPU box unless prebuilt."""
    so = os.path.join(ROOT, "oracle", "_ref", "libref_ikdtree.so")
    if not os.path.exists(so):
        if os.path.exists("/root/reference/include/ikd-Tree/ikd_Tree.cpp"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref not built and /root/reference absent")
    return C.CDLL(so)


@pytest.fixture(scope="session")
def ref_tri_lib():
    """The reference's own Triangle_manager (triangle.hpp / triangle.cpp / tools_kd_hash.hpp) compiled from /root/reference (oracle/_ref).
    On the GPU box only the prebuilt .so exists (oracle/_ref travels with the snapshot)."""
    so = os.path.join(ROOT, "oracle", "_ref", "libref_triangle.so")
    if not os.path.exists(so):
        if os.path.exists("/root/reference/src/meshing/r3live/triangle.hpp"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref/libref_triangle.so not built and /root/reference absent")
    lib = C.CDLL(so)
    lib.rt_create.restype = C.c_void_p; lib.rt_create.argtypes = [C.c_double]
    lib.rt_destroy.argtypes = [C.c_void_p]
    lib.rt_append_vertices.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.rt_commit.restype = C.c_int64; lib.rt_commit.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    lib.rt_set_flips.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.rt_live.restype = C.c_int64; lib.rt_live.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.rt_live_size.restype = C.c_int64; lib.rt_live_size.argtypes = [C.c_void_p]
    lib.rt_find_relative.restype = C.c_int64; lib.rt_find_relative.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    return lib


@pytest.fixture(scope="session")
def hip_lib():
    return capi.load_hip_library()


@pytest.fixture()
def avia():
    return capi.avia_config()


def make_oracle(oracle_lib, cfg):
    return capi.HotPath(oracle_lib, cfg, prefix="orc_")


def make_hip(hip_lib, cfg):
    return capi.HotPath(hip_lib, cfg, prefix="immesh_")


def fetch_device(ptr, shape, dtype=np.float32):
    """Copy a device buffer the HIP library handed out (e.g. immesh_downsample_end) into a host array.  The process holds ONE HIP runtime
    (capi.one_hip_runtime: the system copy the library is linked against, mapped before torch), so the soname resolves to it."""
    import ctypes
    rts = [r for r in capi.mapped_hip_runtimes() if "libamdhip64" in r]
    assert rts, "no HIP runtime mapped"
    # (tests/test_one_runtime.py is where "exactly one" is asserted; should a harness ever map a second one, the library's is the system copy)
    rt = ctypes.CDLL(([r for r in rts if "/torch/" not in r] or rts)[0])
    rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rt.hipMemcpy.restype = ctypes.c_int
    out = np.zeros(shape, dtype)
    rc = rt.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), out.nbytes, 2)   # hipMemcpyDeviceToHost
    assert rc == 0, f"hipMemcpy(device -> host) failed: {rc}"
    return out
