"""Data-parallel sharding of frames over the GPUs of one node.

The reference runs one process per GPU with ``samples_per_gpu=1`` (CFG:188,
tools/dist_train.sh:7-9): frames are independent, every rank holds a full
replica of the head, and inference needs no collective on the data path --
per-frame results are only gathered on the host at the end
(tools/test.py:218-223).  ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm, "gloo" on CPU) is used for the rendezvous / barrier and for the
result gather, never inside a frame.
"""
import os

import torch
import torch.distributed as dist


def gather_floats(value, device=None):
    """A python float of every rank -> list ordered by rank (on every rank)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else 'cpu')
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def env_rank_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init_process_group(backend=None):
    """Initialise from the torchrun environment (no-op for world size 1)."""
    rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard_frames(num_frames, rank, world):
    """Frame indices of this rank: r, r+W, r+2W, ... (the DistributedSampler
    order without shuffling or padding)."""
    return list(range(rank, num_frames, world))


def gather_results(local_results, num_frames):
    """{frame_index: result} of every rank -> list ordered by frame index on
    every rank (host-side object gather, as multi_gpu_test does)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [local_results[i] for i in range(num_frames)]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local_results)
    merged = {}
    for p in parts:
        merged.update(p)
    missing = [i for i in range(num_frames) if i not in merged]
    if missing:
        raise RuntimeError('frames missing after gather: %r' % missing[:8])
    return [merged[i] for i in range(num_frames)]


def max_over_ranks(value, device=None):
    """MAX of a python float over ranks (the bench's timing reduction)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64,
                     device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
