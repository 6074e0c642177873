"""Multi-GPU, one process per GPU, two axes (BASELINE.json north_star "denoising-step batch / multi-clip batch"):

* multi-clip (default): independent clips, clip k -> rank k mod N (the reference's own multi-GPU mode is one OS process per GPU
  over disjoint episode ranges, inference_unity_curve_multi_gpu.sh:41-69).  No collective inside the denoise loop; RCCL
  (torch.distributed backend "nccl") carries the one-time weight broadcast and the gather of finished latents.
* CFG pair (`CfgGroup`, round 4): the two rows of the classifier-free-guidance batch the reference concatenates for every step
  (pipeline_evoworld.py:691-711) are independent until the combine at :709-711.  Ranks 2p and 2p+1 form pair p; each runs the
  U-Net on ONE row (B = 1), the two eps rows ([T*h*w, 4] fp16 = 1.84 MB at 72x128x25) meet in ONE all_gather per step over
  xGMI, and the CFG combine + Euler step is replicated (every rank keeps the full latents and both rows of the next model
  input, so nothing else is exchanged).  This is the axis that moves frames/s PER CLIP: projected about 1.84x at 2 GPUs from one-GPU
  B = 1 forward timings (strong scaling; UNMEASURED on multi-GPU hardware)."""
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
            # EW_DIST_BACKEND=gloo: test hook -- several ranks on ONE GPU (RCCL refuses duplicate devices; gloo stages through the host)
            backend = os.environ.get("EW_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


def gather_floats(value, device):
    """all_gather of one float per rank -> list in rank order (bench.py: per-rank seconds next to the max the metric is computed from)."""
    if _single():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def world_info():
    """(process-group world size, backend name) as torch.distributed reports them -- (1, None) without a process group.  A multi-GPU bench
    line carries these so that a scaling record can show RCCL ("nccl") really saw N ranks."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), str(dist.get_backend())
    return 1, None


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


class CfgGroup:
    """The CFG-pair axis.  `size` ranks (1 or 2) share one clip: row r of the CFG batch (0 = unconditional, 1 = conditional) runs
    on the pair member r mod size.  size 1 = both rows on this rank as two B=1 forwards (the degenerate group: same code path,
    exchange through a 1-rank process group when one exists).  Ranks [p*size, (p+1)*size) form pair p."""

    def __init__(self, rank=0, world=1, size=2):
        if size not in (1, 2) or world % size:
            raise ValueError(f"CFG group size must be 1 or 2 and divide the world size (size={size}, world={world})")
        self.size, self.member, self.pair, self.n_pairs = size, rank % size, rank // size, world // size
        self.group, self._flat = None, False
        if dist.is_available() and dist.is_initialized():
            for p in range(self.n_pairs):               # new_group is collective: every rank creates every pair's group
                g = dist.new_group(ranks=list(range(p * size, (p + 1) * size)))
                if p == self.pair:
                    self.group = g
            # RCCL ("nccl") has the flat all_gather_into_tensor; gloo takes the list form
            self._flat = self.group is not None and dist.get_backend(self.group) == "nccl"

    def rows(self):
        """CFG rows this rank runs the U-Net for"""
        return [r for r in range(2) if r % self.size == self.member]

    def all_gather_rows(self, eps_all, mine):
        """eps_all [2, n, c] (row r filled in by its owner), mine = this rank's row tensor [n, c] (size 2) -> eps_all complete on
        every member.  One collective per denoise step."""
        if self.size == 1:
            if self.group is not None:                   # 1-rank group: exercises the RCCL path, result unchanged
                out = [torch.empty_like(mine)]
                dist.all_gather(out, mine.contiguous(), group=self.group)
                eps_all[self.rows()[-1]].copy_(out[0])
            return eps_all
        if self.group is None:
            raise RuntimeError("CfgGroup(size=2) needs an initialised process group")
        # the form is chosen ONCE from the backend (no try / except: a failing collective must propagate, not be retried as another one)
        if self._flat:
            dist.all_gather_into_tensor(eps_all.view(-1), mine.contiguous().view(-1), group=self.group)
        else:
            dist.all_gather([eps_all[0], eps_all[1]], mine.contiguous(), group=self.group)
        return eps_all
