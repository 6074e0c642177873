"""Multi-GPU = independent clips, one process per GPU (the reference's own multi-GPU mode is one OS
process per GPU over disjoint episode ranges, inference_unity_curve_multi_gpu.sh:41-69).  No collective sits inside
the denoise loop.  RCCL (torch.distributed backend "nccl") is used for the two real exchange steps around it:
a one-time weight broadcast from the rank that read the checkpoint, and the gather of finished latents/frames."""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("EW_FORCE_DIST") == "1" and "MASTER_PORT" in os.environ   # test hook: exercise RCCL with one rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_clips(n_clips, rank, world):
    """clip k -> rank k mod world.  Returns the clip indices this rank owns."""
    return list(range(rank, n_clips, world))


def _single():
    return not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and os.environ.get("EW_FORCE_DIST") != "1")


def broadcast_tensors(tensors, src=0):
    """One-time weight broadcast (packed fp16 weights, 3.04 GB for the full U-Net) from `src` to all ranks."""
    if _single():
        return tensors
    for t in tensors:
        dist.broadcast(t, src=src)
    return tensors


def gather_results(x):
    """all_gather of a per-rank result tensor (final latents [1,T,4,h,w] fp32 = 3.7 MB per clip) -> list, rank order."""
    if _single():
        return [x]
    out = [torch.empty_like(x) for _ in range(dist.get_world_size())]
    dist.all_gather(out, x.contiguous())
    return out


def barrier():
    if not _single():
        dist.barrier()


def max_over_ranks(value, device):
    if _single():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
