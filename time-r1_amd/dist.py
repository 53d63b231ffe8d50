"""Data parallelism: one process per GPU, gradients of the flat arena averaged with RCCL (torch.distributed backend "nccl" IS
RCCL on ROCm) over xGMI; metrics gathered like accelerator.gather_for_metrics (reference src/time_r1/rl/timer1_trainer.py:741-777).
Each prompt group (G completions, rewards, group statistics) stays on one rank, exactly as in the reference (:703-712): the only
exchange is the gradient average at the optimizer step.

MI355X-first gradient exchange (`GradSync`): the arena is laid out layer by layer, and the hand-written backward finishes layers
from the last to the first, so during the LAST micro-step of an accumulation window each layer's gradient range is handed to RCCL
as soon as that layer's backward kernels are enqueued - the all-reduce runs on RCCL's stream over the 7 xGMI links while the
remaining layers are still computing.  The wire format is bf16 (half the bytes; fp32 accumulation stays local), staged through a
bf16 twin of the arena; the 1/world average is folded into the fused AdamW kernel (grad_mult).
"""
import os

import torch
import torch.distributed as dist


def force_single_rank():
    """TR1_DIST_FORCE=1: build the process group and run every collective of the data-parallel path with world size 1 too - the way to execute the
    RCCL calls of this module (all-reduce, reduce-scatter, all-gather on torch.distributed "nccl") on a ONE-GPU box (tests/test_bench_gpu.py)."""
    return os.environ.get("TR1_DIST_FORCE", "0") == "1"


def group_active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_single_rank())


SHARD_WORLDS = (2, 4, 8)


def shard_world_ok(world):
    return world in SHARD_WORLDS or (world == 1 and force_single_rank())


class DataParallel:
    def __init__(self, bucket_bytes=1 << 30):
        self.enabled = group_active()
        self.world = dist.get_world_size() if self.enabled else 1
        self.rank = dist.get_rank() if self.enabled else 0
        self.bucket_bytes = bucket_bytes

    def all_reduce_mean_(self, flat):
        """In-place mean over ranks of a flat tensor, in buckets of `bucket_bytes`."""
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


class GradSync:
    """Gradient SUM over ranks for the flat fp32 gradient arena, range by range, overlapped with backward.

    usage per optimizer step:   sync.begin()  ->  [backward of the last micro-step calls sync.ready(a, b) as ranges become final]
                                ->  sync.finish() (reduces whatever was not announced, waits, writes the sums back to fp32)
    The caller divides by `world` (AdamWFlat passes grad_mult = 1/world to the fused kernel)."""

    def __init__(self, grad_flat, dp: DataParallel, wire_dtype=torch.bfloat16, bucket_elems=1 << 28):
        self.g, self.dp = grad_flat, dp
        self.wire_dtype = wire_dtype
        self.bucket = bucket_elems
        self.stage = None
        self.pending, self.done = [], []
        self.active = False

    def begin(self):
        self.pending, self.done = [], []
        self.wired = []                       # [lo, hi) element ranges whose wire-format copy the producing kernel already wrote (mark_wire)
        self.active = self.dp.enabled
        if self.active and self.wire_dtype != self.g.dtype and self.stage is None:
            self.stage = torch.empty(self.g.numel(), dtype=self.wire_dtype, device=self.g.device)

    def wire_view(self, off, shape):
        """The wire-format twin of the gradient tensor at arena offset `off` (None without a staging arena): a weight-gradient GEMM whose epilogue
        produces the FINAL fp32 value writes its bf16 rounding there as well (Engine._wgrad in the window's last micro-step) and calls mark_wire."""
        if not self.active or self.stage is None:
            return None
        n = 1
        for s in shape:
            n *= int(s)
        return self.stage[off:off + n].view(*shape)

    def mark_wire(self, lo, hi):
        self.wired.append((int(lo), int(hi)))

    def _stage_gaps(self, x, y):
        """fp32 -> bf16 staging copy of [x, y) minus the ranges the producers already wrote in wire format (45 GB of traffic per window at 7B when
        everything is copied; the four large matrices of a decoder layer are 99 % of it)."""
        pos = x
        for lo, hi in sorted(w for w in self.wired if w[1] > x and w[0] < y):
            if lo > pos:
                self.stage[pos:lo].copy_(self.g[pos:lo])
            pos = max(pos, min(hi, y))
        if pos < y:
            self.stage[pos:y].copy_(self.g[pos:y])

    def ready(self, a, b):
        """Elements [a, b) of the gradient arena are final on this rank: start their all-reduce (asynchronous)."""
        if not self.active or b <= a:
            return
        for x in range(a, b, self.bucket):
            y = min(b, x + self.bucket)
            if self.stage is not None:
                buf = self.stage[x:y]
                self._stage_gaps(x, y)            # fp32 -> bf16 wire format (staging copy on the compute stream) of what no epilogue wrote
            else:
                buf = self.g[x:y]
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
            self.pending.append((x, y, work))
        self.done.append((a, b))

    def finish(self, copy_back=True):
        """copy_back=False (bf16 wire): the sums stay in `self.stage` - the optimizer reads them from there (AdamWFlat.step), which saves one
        read of the bf16 arena and one write of the fp32 arena per step (45 GB at 7B)."""
        if not self.active:
            return
        n = self.g.numel()
        covered = sorted(self.done)
        pos = 0
        for a, b in covered + [(n, n)]:           # reduce every range nobody announced
            if a > pos:
                self.ready(pos, a)
            pos = max(pos, b)
        ev = _exposed_begin(self.g)
        for x, y, work in self.pending:
            work.wait()
            if self.stage is not None and copy_back:
                self.g[x:y].copy_(self.stage[x:y])
        _exposed_end(self, ev)
        self.pending, self.done = [], []
        self.active = False


def _exposed_begin(t):
    """HIP event on the compute stream in front of the waits on the exchange: the time until the matching end event is what the compute
    stream spent WAITING for the gradient exchange (the part the backward did not hide)."""
    if t.device.type != "cuda":
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _exposed_end(sync, e0):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    sync.exposed_events = (getattr(sync, "exposed_events", []) + [(e0, e1)])[-256:]


def exposed_ms(sync, clear=True):
    """Sum of the waits recorded by GradSync / ShardSync.finish since the last call (synchronises on the last event)."""
    evs = getattr(sync, "exposed_events", [])
    if clear:
        sync.exposed_events = []
    if not evs:
        return 0.0
    evs[-1][1].synchronize()
    return float(sum(a.elapsed_time(b) for a, b in evs))


class ShardSync:
    """Sharded-optimizer gradient exchange: per arena SEGMENT (embed | decoder layer | norm | lm_head | merger) a REDUCE-SCATTER that
    leaves rank r with the sum of chunk r of that segment - the part of the gradient its 1/world shard of master / m / v needs -
    instead of an all-reduce that would hand every rank everything (reference: DeepSpeed ZeRO reduce_scatter, scripts/zero3.json:22-33).
    Same begin / ready / finish protocol as GradSync, so the backward's per-layer overlap hook drives either.  After finish() the
    summed local gradient shard is in `self.gshard` (fp32 [numel / world], chunk order = Arena.chunks()).  On the xGMI full mesh a
    reduce-scatter moves (world-1)/world of the bytes once over all 7 links in parallel; the matching all-gather of the updated bf16
    weights is issued by the optimizer (AdamWFlat._step_sharded)."""

    def __init__(self, arena, dp: DataParallel, wire_dtype=torch.bfloat16):
        self.arena, self.g, self.dp = arena, arena.grad, dp
        self.wire_dtype = wire_dtype
        self.stage = None
        self.recv = None
        self.gshard = None
        self.pending, self.done = [], set()
        self.active = False
        self._seg_at = {a: (a, b) for _, a, b in arena.segments}

    wire_view, mark_wire, _stage_gaps = GradSync.wire_view, GradSync.mark_wire, GradSync._stage_gaps

    def begin(self):
        self.pending, self.done = [], set()
        self.wired = []
        self.active = self.dp.enabled
        if not self.active:
            return
        W = self.dp.world
        if self.gshard is None:
            self.gshard = torch.empty(self.arena.numel // W, dtype=torch.float32, device=self.g.device)
        if self.wire_dtype != self.g.dtype and self.stage is None:
            self.stage = torch.empty(self.g.numel(), dtype=self.wire_dtype, device=self.g.device)
            self.recv = torch.empty(self.arena.numel // W, dtype=self.wire_dtype, device=self.g.device)

    def ready(self, a, b):
        """Elements [a, b) of the gradient arena (whole segments) are final on this rank: start their reduce-scatter (asynchronous)."""
        if not self.active or b <= a:
            return
        W = self.dp.world
        x = a
        while x < b:
            assert x in self._seg_at, "sharded gradient exchange works on whole arena segments (offset %d is not a segment start)" % x
            _, y = self._seg_at[x]
            assert y <= b, (a, b, x, y)
            if x not in self.done:
                self.done.add(x)
                if self.stage is not None:
                    src = self.stage[x:y]
                    self._stage_gaps(x, y)            # fp32 -> bf16 wire format (staging copy on the compute stream) of what no epilogue wrote
                    dst = self.recv[x // W: y // W]
                else:
                    src = self.g[x:y]
                    dst = self.gshard[x // W: y // W]
                work = dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, async_op=True)
                self.pending.append((x // W, y // W, work))
            x = y

    def finish(self):
        if not self.active:
            return
        for _, a, b in self.arena.segments:        # every segment nobody announced
            self.ready(a, b)
        ev = _exposed_begin(self.g)
        for x, y, work in self.pending:
            work.wait()
            if self.stage is not None:
                self.gshard[x:y].copy_(self.recv[x:y])
        _exposed_end(self, ev)
        self.pending, self.done = [], set()
        self.active = False


def init_from_env(device_type="cuda"):
    """torchrun-style env:// rendezvous (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT). Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_single_rank()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("TR1_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")   # "nccl" IS RCCL on ROCm
        if device_type == "cuda":
            torch.cuda.set_device(int(os.environ.get("TR1_FORCE_DEVICE", local)))
        # explicit timeout: a rank that never arrives (wrong device mapping, a dead peer, an xGMI link that does not train) becomes an error line
        # after TR1_DIST_TIMEOUT_S seconds instead of a job that hangs until the node lease kills it
        import datetime
        timeout = datetime.timedelta(seconds=float(os.environ.get("TR1_DIST_TIMEOUT_S", "600")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local, world


def dist_diagnostics(device):
    """What a first multi-GPU run needs to be debuggable from its one JSON line: which ranks actually took part in a collective on this
    backend (sum / count of rank ids through an all-reduce, device names through an all-gather), the RCCL version torch was built against, the
    backend name.  Every rank must call it (collectives); returns a dict (identical on all ranks)."""
    out = {"backend": None, "world": 1, "ranks_seen": [0], "rccl_version": None, "devices": None}
    try:
        v = torch.cuda.nccl.version()
        out["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:      # informative only
        out["rccl_version"] = "unavailable: %r" % (e,)
    dev = torch.device(device)
    name = torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu"
    if not group_active():
        out["devices"] = [name]
        return out
    world, rank = dist.get_world_size(), dist.get_rank()
    out["backend"], out["world"] = dist.get_backend(), world
    onehot = torch.zeros(world, dtype=torch.int32, device=dev)
    onehot[rank] = 1
    dist.all_reduce(onehot)                                   # rank r contributed iff entry r == 1
    out["ranks_seen"] = [i for i, x in enumerate(onehot.tolist()) if x == 1]
    out["ranks_seen_ok"] = onehot.tolist() == [1] * world
    names = [None] * world
    dist.all_gather_object(names, "%s (cuda:%s)" % (name, dev.index) if dev.type == "cuda" else name)
    out["devices"] = names
    return out
