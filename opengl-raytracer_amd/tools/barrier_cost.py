#!/usr/bin/env python3
"""What the closing barrier of a timed region costs under torch.distributed (run with torch.distributed.run): dist.barrier()
of the nccl backend against a one-element all_reduce, and against a barrier of a gloo group after a device synchronise."""
import os
import time

import torch
import torch.distributed as dist

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
cpu = dist.new_group(backend="gloo")
x = torch.zeros(1, device=dev)
big = torch.zeros((1080, 1920, 4), device=dev)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def nccl_barrier():
    dist.barrier()
    torch.cuda.synchronize()


def gloo_barrier():
    torch.cuda.synchronize()
    dist.barrier(group=cpu)


def tiny_allreduce():
    dist.all_reduce(x)
    torch.cuda.synchronize()


def reduce_image():
    dist.reduce(big, dst=0)
    torch.cuda.synchronize()


if rank == 0:
    print(f"world {world}: nccl barrier {timed(nccl_barrier):.0f} us, gloo barrier after synchronize {timed(gloo_barrier):.0f} us, "
          f"one-element all_reduce {timed(tiny_allreduce):.0f} us, reduce of a 1080p fp32 image {timed(reduce_image):.0f} us", flush=True)
else:
    for fn in (nccl_barrier, gloo_barrier, tiny_allreduce, reduce_image):
        timed(fn)
dist.destroy_process_group()
