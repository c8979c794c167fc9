"""ONE HIP / HSA runtime in a test or bench process: the system copy the product library is linked against, with torch as
its guest -- not the copies the torch wheel bundles, and never both.  Loading libraries needs no GPU, so this runs in the CPU tier; the device
side of the claim (torch ops, a GEMM and an RCCL all-reduce on the system runtime) is tools/r05_one_runtime.py and the GPU test below."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import sys
sys.path.insert(0, {root!r})
{first}
{second}
from immesh_amd import capi
lib = capi.load_hip_library()
import torch
rts = capi.mapped_hip_runtimes()
print("RUNTIMES", "|".join(rts))
"""


def _probe(first, second="", env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    out = subprocess.run([sys.executable, "-c", _PROBE.format(root=ROOT, first=first, second=second)], capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RUNTIMES")][-1]
    return [r for r in line.split(" ", 1)[1].split("|") if r]


def _need_lib():
    from immesh_amd import capi
    if not os.path.exists(capi.hip_library_path()):
        pytest.skip("libimmesh_hip.so not built")


def test_binding_first_means_the_system_runtime_for_everyone():
    """bench.py's order (the binding, then torch): exactly one libamdhip64 and one libhsa-runtime64, neither from the torch wheel."""
    _need_lib()
    rts = _probe("from immesh_amd import capi")
    hip = [r for r in rts if "libamdhip64" in r]
    hsa = [r for r in rts if "libhsa-runtime64" in r]
    assert len(hip) == 1 and len(hsa) == 1, rts
    assert "/torch/" not in hip[0] and "/torch/" not in hsa[0], rts


def test_torch_first_is_still_one_runtime():
    """A process that imported torch before the binding keeps torch's copy, and the library binds to it by soname: one runtime, never two."""
    _need_lib()
    rts = _probe("import torch")
    assert len([r for r in rts if "libamdhip64" in r]) == 1, rts
    assert len([r for r in rts if "libhsa-runtime64" in r]) == 1, rts


def test_a_rank_of_a_multi_gpu_job_runs_on_torchs_bundle():
    """WORLD_SIZE > 1: torch's runtime + RCCL bundle serves the rank and the library binds to it (capi.one_hip_runtime) -- one runtime there too."""
    _need_lib()
    rts = _probe("from immesh_amd import capi", env={"WORLD_SIZE": "8"})
    hip = [r for r in rts if "libamdhip64" in r]
    hsa = [r for r in rts if "libhsa-runtime64" in r]
    assert len(hip) == 1 and len(hsa) == 1, rts
    assert "/torch/" in hip[0] and "/torch/" in hsa[0], rts


@pytest.mark.gpu
def test_this_process_runs_on_one_runtime(hip_lib):
    """The GPU tier itself: conftest imports the binding before torch, so by now torch has initialised the device on the system runtime."""
    import torch
    from immesh_amd import capi
    assert torch.cuda.is_available()
    x = torch.arange(1 << 16, device="cuda", dtype=torch.float64)
    assert float(x.sum()) == (1 << 16) * ((1 << 16) - 1) / 2
    rts = capi.mapped_hip_runtimes()
    hip = [r for r in rts if "libamdhip64" in r]
    assert len(hip) == 1, rts
    assert len([r for r in rts if "libhsa-runtime64" in r]) == 1, rts
    # which one: the system copy unless something imported torch before the binding (then torch's own -- still one)
    assert ("/torch/" not in hip[0]) == (capi.RUNTIME_CHOICE == "system"), (capi.RUNTIME_CHOICE, rts)
