"""Host-enqueue time vs GPU time of the decode loop (7B width, few layers): is rollout launch-bound?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time_r1_amd  # noqa: E402,F401
from time_r1_amd.ops import HipOps  # noqa: E402
from time_r1_amd.config import qwen2_vl_7b  # noqa: E402
from time_r1_amd.params import ModelParams  # noqa: E402
from time_r1_amd.model import Engine  # noqa: E402
from time_r1_amd.grpo import GRPOCore  # noqa: E402
from time_r1_amd.synthetic import synthetic_prompt  # noqa: E402

L = int(os.environ.get("L", 6))
C = int(os.environ.get("C", 60))
B = int(os.environ.get("B", 2))
ops = HipOps("cuda:0")
cfg = qwen2_vl_7b()
cfg.text.n_layers = L
cfg.vision.depth = 2
params = ModelParams(cfg, ops, init="none", optimizer_state=False)
params.init_random_device(0)
eng = Engine(cfg, ops, params)
core = GRPOCore(eng, None, 8, C, beta=0.0, seed=1, rope_index_mode="hf4")
sts = []
for b in range(B):
    ids, pix, grid = synthetic_prompt(cfg, (16, 22, 38), 20, 30, seed=b)
    sts.append(core.prepare(ids, pix, grid))
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    core.rollout_many(sts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    steps = C - 1
    print("rep %d: host enqueue %.1f ms, total %.1f ms  -> per decode step: host %.3f ms, wall %.3f ms, per layer wall %.1f us"
          % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e3 * (t1 - t0) / steps, 1e3 * (t2 - t0) / steps, 1e6 * (t2 - t0) / steps / L))
