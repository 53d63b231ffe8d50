#!/bin/bash
# round 6, first call: full GPU suite at the new HEAD, the attention bench with the vendor SDPA yardstick, baseline config-3 / config-2 lines for this box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r06_c1_gputest.txt
timeout 600 python tools/bench_attn.py --yardstick --iters 20 > $O/r06_c1_attn.json 2> $O/r06_c1_attn.err
X="--no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg"
timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 $X > $O/r06_c1_bench7b.json 2> $O/r06_c1_bench7b.err
timeout 600 python bench.py --model qwen2-vl-2b --frames 16 --steps 8 --warmup 2 $X > $O/r06_c1_cfg2.json 2> $O/r06_c1_cfg2.err
cat $O/r06_c1_gputest.txt; cat $O/r06_c1_attn.json; tail -3 $O/r06_c1_attn.err
python - <<PY
import json
for c in ("bench7b", "cfg2"):
    try:
        d = json.loads(open("$O/r06_c1_%s.json" % c).read().strip().splitlines()[-1])
        print(c, round(d["value"], 4), round(d["ms_per_step"], 1), d.get("rollout_tokens_per_sec"), d.get("phases_ms_per_step"), d["config"]["workload"][:120])
    except Exception as e:
        print(c, "failed", e)
PY
