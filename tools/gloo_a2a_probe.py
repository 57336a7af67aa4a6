"""Probe: repeated all_to_all_single on CUDA tensors over gloo, two ranks sharing one GPU, with no
library kernels involved.  Tells a backend problem from one in the sharded path."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(40, exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
shape = (world, 9, int(sys.argv[1]), 4)
heavy = torch.ones(1 << 26, dtype=torch.int64, device="cuda") if len(sys.argv) > 2 else None
for it in range(6):
    send = torch.full(shape, rank * 100 + it, dtype=torch.int64, device="cuda")
    if heavy is not None:          # several milliseconds of kernels still in flight when the exchange starts
        for _ in range(8):
            heavy = torch.cumsum(heavy, 0) % 1000003
        send = send + (heavy[:1] * 0)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    torch.cuda.synchronize()
    assert [int(recv[g, 0, 0, 0]) for g in range(world)] == [g * 100 + it for g in range(world)]
    print(f"rank {rank} iteration {it} ok", flush=True)
dist.barrier()
dist.destroy_process_group()
