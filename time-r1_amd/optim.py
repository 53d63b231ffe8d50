"""Optimizer step over the flat trainable arena: global grad-norm clip + fused AdamW (one HIP kernel each), preceded by the
data-parallel gradient exchange.  Replaces HF Trainer.training_step's clip + DeepSpeed engine.step()
(reference: TF trainer.py:1785, scripts/zero3.json:13-21, :35 gradient_clipping auto = max_grad_norm 1.0)."""
import torch

from .dist import DataParallel, GradSync


class AdamWFlat:
    def __init__(self, params, ops, lr=1e-6, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, dp: DataParallel = None,
                 grad_wire_dtype=torch.bfloat16):
        self.params, self.ops = params, ops
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        self.dp = dp or DataParallel()
        self.sync = GradSync(params.train.grad, self.dp, wire_dtype=grad_wire_dtype)
        self._sumsq = ops.zeros(1, dtype=torch.float32)

    def step(self, lr=None):
        """Averages grads across ranks, clips by global norm, applies AdamW, refreshes the bf16 working weights, zeroes grads.
        Returns the (pre-clip) gradient norm as a device scalar (no host sync)."""
        a = self.params.train
        if self.dp.enabled and not self.sync.active:
            self.sync.begin()           # nobody overlapped the exchange with backward: reduce everything now
        self.sync.finish()              # grad arena now holds the SUM over ranks; the mean is folded into grad_mult
        mult = 1.0 / self.dp.world
        self._sumsq.zero_()
        self.ops.sumsq_accum(a.grad, self._sumsq)
        self.step_count += 1
        a.version = getattr(a, "version", 0) + 1
        self.ops.adamw_step(a.master, a.m, a.v, a.grad, a.w16, self.lr if lr is None else lr, self.betas[0], self.betas[1], self.eps,
                            self.weight_decay, self.step_count, sumsq=self._sumsq, max_norm=self.max_grad_norm, grad_mult=mult, zero_grad=True)
        return self._sumsq.sqrt() * mult

    def state_dict(self):
        a = self.params.train
        return dict(step=self.step_count, master=a.master, m=a.m, v=a.v)

    def load_state_dict(self, sd):
        a = self.params.train
        self.step_count = int(sd["step"])
        a.master.copy_(sd["master"]); a.m.copy_(sd["m"]); a.v.copy_(sd["v"])
        a.w16.copy_(a.master)
