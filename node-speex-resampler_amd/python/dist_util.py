"""Multi-GPU plumbing for the resampler (one process per GPU, torch.distributed).

The path shards by independent streams: stream s of the job lives on rank
s % world, no data-path collective exists.  Collectives are used only for control: a barrier
around the timed region, MAX over ranks of the elapsed time, SUM of produced-sample counts and
an XOR-free additive checksum of the outputs.  Backend "nccl" is RCCL on ROCm; the CPU tests run
the same code over "gloo".
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(
        os.environ.get("LOCAL_RANK", "0"))


def init(backend):
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def shard_streams(total_streams, world, rank):
    """Global stream ids owned by `rank` (round-robin: stream s -> rank s % world)."""
    return [s for s in range(total_streams) if s % world == rank]


def barrier(device=None):
    if dist.is_initialized():
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def reduce_scalar(value, op, device):
    """All-reduce one float64 (op: 'max' or 'sum'); identity when not distributed."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def reduce_int(value, device):
    if not dist.is_initialized():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def finish():
    if dist.is_initialized():
        dist.destroy_process_group()
