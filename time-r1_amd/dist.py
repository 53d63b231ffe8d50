"""Data parallelism: one process per GPU, gradients of the flat arena averaged with RCCL (torch.distributed backend "nccl" IS
RCCL on ROCm) in a few large buckets over xGMI; metrics gathered like accelerator.gather_for_metrics
(reference src/time_r1/rl/timer1_trainer.py:741-777).  Each prompt group (G completions, rewards, group statistics) stays on
one rank, exactly as in the reference (:703-712): the only exchange is the gradient average at the optimizer step.
"""
import os

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, bucket_bytes=1 << 30):
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.enabled else 1
        self.rank = dist.get_rank() if self.enabled else 0
        self.bucket_bytes = bucket_bytes

    def all_reduce_mean_(self, flat):
        """In-place mean over ranks of a flat fp32 tensor, in buckets of `bucket_bytes` (large enough to saturate the 7 xGMI
        links per GPU, small enough to pipeline)."""
        if not self.enabled:
            return flat
        n = flat.numel()
        per = max(1, self.bucket_bytes // flat.element_size())
        works = []
        for a in range(0, n, per):
            works.append(dist.all_reduce(flat[a: min(n, a + per)], op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()
        flat.mul_(1.0 / self.world)
        return flat

    def gather(self, t):
        """all_gather along dim 0 (metrics); returns the input when not distributed."""
        if not self.enabled:
            return t
        t = t.contiguous()
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return torch.cat([o.reshape(-1, *t.shape[1:]) if t.dim() else o.reshape(1) for o in out], 0)

    def barrier(self):
        if self.enabled:
            dist.barrier()


def init_from_env(device_type="cuda"):
    """torchrun-style env:// rendezvous (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT). Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("TR1_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")   # "nccl" IS RCCL on ROCm
        if device_type == "cuda":
            torch.cuda.set_device(int(os.environ.get("TR1_FORCE_DEVICE", local)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
