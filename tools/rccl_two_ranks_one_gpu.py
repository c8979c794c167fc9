"""Does RCCL accept two ranks on ONE device?  (the multi-rank data path is exercised on one GPU through the host callbacks; this shows why
the RCCL transport itself cannot be.)  usage: python tools/rccl_two_ranks_one_gpu.py"""
import os
import sys


def main(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        t = torch.ones(4, device="cuda:0") * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce over two ranks on cuda:0 -> {t.tolist()}", flush=True)
    except Exception as e:   # noqa: BLE001
        print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:300]}", flush=True)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    mp.spawn(main, args=(2, 29511), nprocs=2, join=True)
