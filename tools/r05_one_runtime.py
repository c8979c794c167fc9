"""One-off experiment (round 5): can the whole test / bench process -- torch included -- run on the HIP + HSA runtime the product
library is linked against (/opt/rocm), instead of the copies the torch wheel bundles?  Preload both by the bare names torch's libraries ask for,
then import torch, run a device op, a world-1 RCCL all-reduce, and the library's smoke; print which runtime files the process mapped."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def maps(tag):
    s = set()
    for ln in open("/proc/self/maps"):
        if "libamdhip64" in ln or "libhsa-runtime" in ln or "librccl" in ln:
            s.add(ln.split()[-1])
    print(tag, sorted(s), flush=True)


if os.environ.get("PRELOAD", "1") == "1":
    ctypes.CDLL("libhsa-runtime64.so", mode=ctypes.RTLD_GLOBAL)
    ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
maps("after preload")
import torch
maps("after torch")
x = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
print("torch op", float((x * 2).sum()), torch.version.hip, flush=True)
y = torch.randn(512, 512, device="cuda"); print("matmul", float((y @ y).abs().mean()), flush=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(1024, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize(); print("all_reduce", float(t.sum()), flush=True)
dist.destroy_process_group()
import __graft_entry__ as g
g.smoke()
maps("after smoke")
print("ONE_RUNTIME_OK", flush=True)
