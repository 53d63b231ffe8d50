"""Optimizer step over the flat trainable arena: global grad-norm clip + fused AdamW (one HIP kernel each), preceded by the
data-parallel gradient exchange.  Replaces HF Trainer.training_step's clip + DeepSpeed engine.step()
(reference: TF trainer.py:1785, scripts/zero3.json:13-21, :35 gradient_clipping auto = max_grad_norm 1.0)."""
import torch

import torch.distributed as dist

from .dist import DataParallel, GradSync, ShardSync


class AdamWFlat:
    def __init__(self, params, ops, lr=1e-6, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, dp: DataParallel = None,
                 grad_wire_dtype=torch.bfloat16, shard_optimizer=False):
        self.params, self.ops = params, ops
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        self.dp = dp or DataParallel()
        # ZeRO-style sharding (config "ZeRO-3 + DP=8", reference scripts/zero3.json): master / m / v live on 1/world of every arena segment;
        # gradients arrive by reduce-scatter, the updated bf16 weights leave by all-gather.  A single process keeps the plain path.
        self.sharded = bool(shard_optimizer) and self.dp.enabled
        if self.sharded:
            if params.train.shard is None:
                params.train.set_shard(self.dp.rank, self.dp.world)
            assert params.train.shard == (self.dp.rank, self.dp.world), params.train.shard
            self.sync = ShardSync(params.train, self.dp, wire_dtype=grad_wire_dtype)
        else:
            self.sync = GradSync(params.train.grad, self.dp, wire_dtype=grad_wire_dtype)
        self._sumsq = ops.zeros(1, dtype=torch.float32)
        self._send = None
        # Engine.lazy_zero_plan(): the decoder layers' large matrices are overwritten by the next window's first weight gradients, so this step does
        # not zero them (set by the owner of the engine; None = zero the whole gradient arena as DeepSpeed's engine.step / optimizer.zero_grad do)
        self.lazy_zero = None
        self.lazy_zero_ok = None      # callable -> bool, re-checked at every step (the trainer passes `lambda: engine.wgrad_overwrite_first`): the plan is only
                                      # sound while the engine's first weight gradient of a window still OVERWRITES the matrices this step leaves un-zeroed

    def _lz(self):
        ok = self.lazy_zero_ok
        return self.lazy_zero if (self.lazy_zero and (ok is None or ok())) else None

    def norm_sink_begin(self, engine):
        """Call before the backward of a window's LAST micro-step (single process): returns the sink to hang on `engine.norm_sink` for that backward, or
        None.  The weight-gradient epilogues of the decoder layers' large matrices then leave the squared norm of the final gradient values, and step()
        reads only the rest of the arena (embedding, norms, biases, lm_head, merger) for the global norm: 30 GB less to stream per step at 7B."""
        self._sink = None
        lz = self._lz()
        if not lz or engine is None:
            return None
        # under a process group the epilogues still serve the exchange: they write the bf16 wire copy of the final gradient (no staging pass for the large
        # matrices).  Their sums of squares are the LOCAL gradient's, which is the clipped norm only for a group of one rank (norm of the sum != sum of norms).
        sync = self.sync if (self.dp.enabled and getattr(self.sync, "active", False) and getattr(self.sync, "stage", None) is not None) else None
        if self.dp.enabled and sync is None:
            return None
        if getattr(self, "_partials", None) is None:
            # one slot per wave of every tile block of the covered matrices (7B: 0.72 M floats), sized from the layout instead of a fixed guess
            a, bound = self.params.train, getattr(self.ops, "wgrad_sumsq_partials", None)
            need = sum(bound(*a.offsets["l%d.%s" % (i, nm)][1]) for i in range(lz["count"]) for nm in type(engine).OVERWRITTEN) if bound else (1 << 21)
            self._partials = torch.empty(need + 512, dtype=torch.float32, device=a.grad.device)
        self._sink = dict(partials=self._partials, n=0, covered=set(),
                          want={"l%d.%s" % (i, nm) for i in range(lz["count"]) for nm in type(engine).OVERWRITTEN}, gver=getattr(self.params.train, "version", 0),
                          engine=engine, bwd=getattr(engine, "bwd_count", 0) + 1,      # valid only if the armed backward is the engine's LAST one before step()
                          sync=sync, norm_ok=(not self.dp.enabled) or self.dp.world == 1)
        return self._sink

    def _norm_from_sink(self):
        """True when the global squared norm could be assembled from the sink + the uncovered parts of the arena (self._sumsq then holds it)."""
        sink, a, lz = getattr(self, "_sink", None), self.params.train, self._lz()
        self._sink = None
        if not sink or not lz or not sink.get("norm_ok", True) or sink["covered"] != sink["want"] or sink["gver"] != getattr(a, "version", 0):
            return False
        if getattr(sink["engine"], "bwd_count", sink["bwd"]) != sink["bwd"] or not getattr(sink["engine"], "wgrad_overwrite_first", True):
            return False      # another backward accumulated into the arena after the armed one (compute_loss / training_step between window and step): full pass
        ops = self.ops
        lo, hi = lz["base"], lz["base"] + lz["stride"] * lz["count"]
        ops.sumsq_partials_accum(sink["partials"], sink["n"], self._sumsq)
        for x, y in ((0, lo), (hi, a.numel)):
            if y > x:
                ops.sumsq_accum(a.grad[x:y], self._sumsq)
        _, small = self._zero_spans(a.numel)
        if small:
            ops.sumsq_ranges_periodic(a.grad, lz["base"], lz["stride"], lz["count"], small, self._sumsq)
        return True

    def _zero_spans(self, n):
        """[(a, b, zero_flag)] covering [0, n) for the fused AdamW launches + the periodic clean-up of the small per-layer tensors."""
        lz = self._lz()
        if not lz:
            return [(0, n, True)], None
        a, b = lz["base"], lz["base"] + lz["stride"] * lz["count"]
        spans = [(x, y, z) for x, y, z in ((0, a, True), (a, b, False), (b, n, True)) if y > x]
        small, pos = [], 0
        for ka, kb in lz["keep"]:
            if ka > pos:
                small.append((pos, ka))
            pos = max(pos, kb)
        if pos < lz["stride"]:
            small.append((pos, lz["stride"]))
        return spans, small

    def step(self, lr=None):
        """Averages grads across ranks, clips by global norm, applies AdamW, refreshes the bf16 working weights, zeroes grads.
        Returns the (pre-clip) gradient norm as a device scalar (no host sync)."""
        a = self.params.train
        if self.sharded:
            return self._step_sharded(lr)
        if self.dp.enabled and not self.sync.active:
            self.sync.begin()           # nobody overlapped the exchange with backward: reduce everything now
        # bf16 wire: the SUM over ranks stays in the wire buffer and norm + AdamW read it there (no copy back into the fp32 accumulator, which
        # the fused kernel only zeroes); fp32 wire / single process: the accumulator itself.  The mean is folded into grad_mult.
        g16 = self.sync.stage if (self.dp.enabled and self.sync.stage is not None) else None
        self.sync.finish(copy_back=g16 is None)
        mult = 1.0 / self.dp.world
        self._sumsq.zero_()
        # (diagnostic: did the weight-gradient epilogues supply the large matrices' norm?  With a group of ONE rank the summed wire gradient is this rank's own,
        # so the epilogue sums are the clipped norm there too - the single-rank-group run of tools/ab_dp_single_rank.sh; with N > 1 the norm is taken from g16)
        self.norm_from_sink = (g16 is None or self.dp.world == 1) and self._norm_from_sink()
        if not self.norm_from_sink:
            self.ops.sumsq_accum(a.grad if g16 is None else g16, self._sumsq)
        self.step_count += 1
        a.version = getattr(a, "version", 0) + 1
        spans, small = self._zero_spans(a.numel)
        for x, y, z in spans:
            kw = {} if g16 is None else {"g16": g16[x:y]}
            self.ops.adamw_step(a.master[x:y], a.m[x:y], a.v[x:y], a.grad[x:y], a.w16[x:y], self.lr if lr is None else lr, self.betas[0], self.betas[1], self.eps,
                                self.weight_decay, self.step_count, sumsq=self._sumsq, max_norm=self.max_grad_norm, grad_mult=mult, zero_grad=z, **kw)
        if small:
            lz = self._lz()
            self.ops.zero_ranges_periodic(a.grad, lz["base"], lz["stride"], lz["count"], small)
        return self._sumsq.sqrt() * mult

    def _step_sharded(self, lr=None):
        """reduce-scatter (overlapped with the backward through ShardSync.ready, finished here) -> global norm from the local shards (one
        scalar all-reduce) -> fused AdamW on the local chunk of every segment, writing the new bf16 weights into a send buffer (1/world of the
        arena) -> per-segment all-gather of those chunks into the full working arena, issued right behind the segment's AdamW launch so
        the exchange of segment s overlaps the update of segment s+1."""
        a, ops, W = self.params.train, self.ops, self.dp.world
        if not self.sync.active:
            self.sync.begin()
        self.sync.finish()
        g = self.sync.gshard
        mult = 1.0 / W
        self._sumsq.zero_()
        ops.sumsq_accum(g, self._sumsq)
        dist.all_reduce(self._sumsq, op=dist.ReduceOp.SUM)
        self.step_count += 1
        a.version = getattr(a, "version", 0) + 1
        works = []
        if self._send is None:          # the updated bf16 chunks are sent from their own buffer (1/world of the arena), not from inside the gather's
            self._send = torch.empty(a.numel // W, dtype=a.w16.dtype, device=a.w16.device)       # output: no aliasing between send and receive views
        for (_, sa, sb), (ca, cb, la) in zip(a.segments, a.chunks()):
            n = cb - ca
            ops.adamw_step(a.master[la:la + n], a.m[la:la + n], a.v[la:la + n], g[la:la + n], self._send[la:la + n], self.lr if lr is None else lr,
                           self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, sumsq=self._sumsq, max_norm=self.max_grad_norm,
                           grad_mult=mult, zero_grad=False)
            works.append(dist.all_gather_into_tensor(a.w16[sa:sb], self._send[la:la + n], async_op=True))
        spans, small = self._zero_spans(a.numel)      # the full fp32 accumulator (the fused kernel only sees the reduced shard)
        for x, y, z in spans:
            if z:
                a.grad[x:y].zero_()
        if small:
            lz = self._lz()
            ops.zero_ranges_periodic(a.grad, lz["base"], lz["stride"], lz["count"], small)
        for w in works:
            w.wait()
        return self._sumsq.sqrt() * mult

    def state_dict(self):
        a = self.params.train
        return dict(step=self.step_count, master=a.master, m=a.m, v=a.v, shard=a.shard)

    def load_state_dict(self, sd):
        a = self.params.train
        if sd.get("shard") != a.shard:
            raise ValueError("optimizer state was saved with shard %s but this run uses %s (resume with the same world size and optimizer layout: GRPOConfig.shard_optimizer=False / bench.py --replicated-optimizer selects the replicated form that checkpoints written before round 3 used by default)"
                             % (sd.get("shard"), a.shard))
        self.step_count = int(sd["step"])
        a.master.copy_(sd["master"]); a.m.copy_(sd["m"]); a.v.copy_(sd["v"])
        if a.shard is None:
            a.w16.copy_(a.master)
        else:       # every rank restores its chunks; the other ranks' chunks come from the checkpoint's full 16-bit weights (already loaded)
            for ca, cb, la in a.chunks():
                a.w16[ca:cb].copy_(a.master[la:la + cb - ca])
