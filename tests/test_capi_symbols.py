"""CPU-side checks of the drop-in boundary: the product library loads and exports every symbol include/immesh_c_api.h declares,
struct layouts of the ctypes binding match the header, and a context cannot be created without a HIP device (no CPU fallback).
No compute calls are made here."""
import ctypes as C
import os
import re

import pytest

from immesh_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "immesh_c_api.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(immesh_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_boundary():
    fns = _declared_functions()
    for must in ("immesh_create", "immesh_destroy", "immesh_map_build", "immesh_register", "immesh_residuals", "immesh_map_update",
                 "immesh_mesh_scan", "immesh_mesh_sizes", "immesh_mesh_fetch", "immesh_process_scan", "immesh_dump_planes", "immesh_counters"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    if not os.path.exists(capi.hip_library_path()):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "immesh_amd", "csrc"), "-j8"])
    lib = capi.load_hip_library()
    missing = [f for f in _declared_functions() if not hasattr(lib, f)]
    assert not missing, missing


def test_oracle_mirrors_the_boundary(oracle_lib):
    """The checker exports the same entry points under the orc_ prefix (so parity tests drive both with identical calls)."""
    for f in _declared_functions():
        if f in ("immesh_default_config", "immesh_create_error", "immesh_last_error", "immesh_profile_enable", "immesh_profile_read",
                 "immesh_rccl_unique_id", "immesh_rccl_init", "immesh_rccl_error"):   # (the checker is single-process: no collectives)
            continue
        assert hasattr(oracle_lib, f.replace("immesh_", "orc_", 1)), f


def test_struct_layouts_match_header():
    assert C.sizeof(capi.PlaneRec) == 8 * 3 + 4 * 6 + 4 * 4 + 8 * 3 + 8 * 3 + 8 * 36
    assert C.sizeof(capi.MeshSizes) == 32
    assert C.sizeof(capi.Counters) == 8 * len(capi.COUNTER_FIELDS)
    assert C.sizeof(capi.KernelStat) == 72
    # immesh_config: field order / padding as the C compiler lays it out
    src = open(HEADER).read()
    body = src[src.index("typedef struct immesh_config {"):src.index("} immesh_config;")]
    names = re.findall(r"\b(?:double|int32_t|int64_t)\s+([A-Za-z_0-9]+)(?:\[\d+\])?;", body)
    assert names == [n for n, _ in capi.Config._fields_]


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    lib = capi.load_hip_library()
    with pytest.raises(RuntimeError, match="no usable HIP device|fallback"):
        capi.HotPath(lib, capi.avia_config(), "immesh_")


def test_forward_without_imu_matches_harness():
    """immesh_forward_without_imu (host-side C++ in the product library) against the numpy harness version used by the oracle legs."""
    import numpy as np
    from immesh_amd import synth
    lib = capi.load_hip_library()
    rng = np.random.default_rng(3)
    st = capi.make_state(R=synth.so3_exp(rng.normal(0, 0.3, 3)), t=rng.normal(0, 5, 3), cov_diag=1e-4, vel=rng.normal(0, 1, 3))
    st[15:18] = rng.normal(0, 0.05, 3)
    c = rng.normal(0, 1e-3, (18, 18)); st[24:] = (c @ c.T + np.eye(18) * 1e-5).reshape(-1)
    a = capi.forward_without_imu_native(lib, st)
    b = synth.forward_without_imu(st)
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-15)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/immesh_c_api.h compiled as C99 with -Wall -Werror, linked against the product library, host-only entry points exercised."""
    import subprocess
    exe = str(tmp_path / "c_abi_smoke")
    libdir = os.path.dirname(capi.hip_library_path())
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "c_abi_smoke.c"),
                           "-o", exe, "-L", libdir, "-limmesh_hip", "-lm", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c abi ok" in out.stdout
