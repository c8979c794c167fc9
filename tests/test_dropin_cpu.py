"""The drop-in shim's marshalling, call order and host-mirror update on a machine without a GPU: drop_in/immesh_shim.cpp + shim_main.cpp linked
against the CPU oracle (entry points renamed by macros, drop_in/Makefile target shim_main_oracle) must reproduce, bit for bit, the same calls
issued directly -- and `make -C drop_in shim_main` (the product link against libimmesh_hip.so) must build."""
import os
import subprocess

from conftest import make_oracle, ROOT
from test_gpu_dropin import run_drop_in, run_drop_in_async


def test_shim_against_the_oracle_build(oracle_lib, tmp_path):
    run_drop_in(oracle_lib, lambda cfg: make_oracle(oracle_lib, cfg), "shim_main_oracle", tmp_path, expect_ply=False)


def test_shim_links_against_the_product_library():
    if not os.path.exists(os.path.join(ROOT, "immesh_amd", "csrc", "libimmesh_hip.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "immesh_amd", "csrc"), "-j8"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "drop_in"), "shim_main"])
    assert os.path.exists(os.path.join(ROOT, "drop_in", "shim_main"))


def test_async_shim_against_the_oracle_build(oracle_lib):
    """the service-level shim (two threads: scan thread + service_reconstruct_mesh) linked against the oracle, scan by scan in lock-step (the checker is
    synchronous: a frame's lists are the current ones until the next call)"""
    run_drop_in_async(lambda cfg: make_oracle(oracle_lib, cfg), "libimmesh_dropin_async_oracle.so", lockstep=True)


def test_async_shim_links_against_the_product_library():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "drop_in"), "libimmesh_dropin_async.so"])
    assert os.path.exists(os.path.join(ROOT, "drop_in", "libimmesh_dropin_async.so"))


def test_async_shim_with_the_reference_triangle_manager_against_the_oracle_build(oracle_lib):
    """the same shim with THE REFERENCE'S OWN Triangle_manager as the host mirror (drop_in/Makefile `refmirror`: triangle.hpp / triangle.cpp /
    tools_kd_hash.hpp compiled from where they lie), linked against the oracle: the real manager's live set and flips per frame equal the direct calls'"""
    run_drop_in_async(lambda cfg: make_oracle(oracle_lib, cfg), "_ref/libimmesh_dropin_async_refmirror_oracle.so", lockstep=True)


def test_async_shim_without_a_mirror_thread_against_the_oracle_build(oracle_lib):
    """queue depth 0: the service thread applies the lists itself (no mirror thread) -- the same frames, the same mirror states"""
    run_drop_in_async(lambda cfg: make_oracle(oracle_lib, cfg), "libimmesh_dropin_async_oracle.so", lockstep=True, queue_depth=0)


def test_both_shims_compile_against_the_reference_s_own_class_declaration():
    """`make -C drop_in realclass`: immesh_shim.cpp and immesh_shim_async.cpp compiled with class Voxel_mapping = src/voxel_mapping.hpp:132-414 of the
    reference (cut out by line range at build time) instead of drop_in/stubs' re-declaration: the replaced bodies name the reference's members with the
    reference's types (VERDICT r05 missing #5).  Only where the reference tree exists (this container; the GPU box has no /root/reference)."""
    if not os.path.exists("/root/reference/src/voxel_mapping.hpp"):
        pytest.skip("/root/reference absent")
    for f in ("immesh_shim_realclass.o", "immesh_shim_async_realclass.o"):
        p = os.path.join(ROOT, "drop_in", "_ref", f)
        if os.path.exists(p):
            os.remove(p)
    out = subprocess.check_output(["make", "-C", os.path.join(ROOT, "drop_in"), "realclass"], stderr=subprocess.STDOUT).decode()
    assert "compiled immesh_shim.cpp + immesh_shim_async.cpp against" in out, out
    for f in ("immesh_shim_realclass.o", "immesh_shim_async_realclass.o"):
        assert os.path.getsize(os.path.join(ROOT, "drop_in", "_ref", f)) > 10000
