"""Per-step GPU time of the native decode loop WITHOUT a profiler (one HIP event per decode step): are there periodic hiccups?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time_r1_amd  # noqa: E402,F401
from time_r1_amd.ops import HipOps  # noqa: E402
from time_r1_amd.config import qwen2_vl_7b, PRESETS  # noqa: E402
from time_r1_amd.params import ModelParams  # noqa: E402
from time_r1_amd.model import Engine  # noqa: E402
from time_r1_amd.grpo import GRPOCore  # noqa: E402
from time_r1_amd.synthetic import synthetic_prompt  # noqa: E402

C = int(os.environ.get("C", 200))
ops = HipOps("cuda:0")
MODEL = os.environ.get("MODEL", "7b")      # MODEL=2b: BASELINE config 2 (Qwen2-VL-2B, 16 frames: grid 8 x 26 x 46)
cfg = qwen2_vl_7b() if MODEL == "7b" else PRESETS["qwen2-vl-2b"]()
GRID = (16, 22, 38) if MODEL == "7b" else (8, 26, 46)
cfg.vision.depth = 2
params = ModelParams(cfg, ops, init="none", optimizer_state=False)
params.init_random_device(0)
eng = Engine(cfg, ops, params)
core = GRPOCore(eng, None, 8, C, beta=0.0, seed=1, rope_index_mode="hf4")
sts = [core.prepare(*synthetic_prompt(cfg, GRID, 64, 64, seed=b)) for b in range(2)]
evs = []
orig = ops.decode_step


def timed(*a, **k):
    r = orig(*a, **k)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    return r


ops.decode_step = timed
for rep in range(2):
    evs.clear()
    core.rollout_many(sts)
    torch.cuda.synchronize()
    d = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]) * 1e3
    print("rep %d: %d steps  median %.0f us  p90 %.0f  max %.0f   steps > median + 100 us: %d   sum of excess %.1f ms" %
          (rep, len(d), np.median(d), np.percentile(d, 90), d.max(), int((d > np.median(d) + 100).sum()), float((d - np.median(d)).clip(min=0).sum() / 1e3)))
    print("   first 40 step times (us):", " ".join("%.0f" % x for x in d[:40]))
