#!/bin/bash
# round 5 mid-round check: full GPU suite, headline bench, configs 4 / 2 / 5
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; 
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/r05_mid_gputest.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_mid_bench7b.json 2> $O/r05_mid_bench7b.err
X="--no-cpu-baseline --no-roofline --no-peak-probe --no-engine-leg"
timeout 900 python bench.py --model qwen2.5-vl-7b --frames 64 --G 16 --C 1024 --beta 0 --clip-loss --steps 2 --warmup 2 $X > $O/r05_mid_cfg4.json 2> $O/r05_mid_cfg4.err
timeout 600 python bench.py --model qwen2-vl-2b --frames 16 --steps 8 --warmup 2 $X > $O/r05_mid_cfg2.json 2> $O/r05_mid_cfg2.err
timeout 600 python bench.py --rollout-fp8 --steps 8 --warmup 2 $X > $O/r05_mid_cfg5.json 2> $O/r05_mid_cfg5.err
cat $O/r05_mid_gputest.txt
python - <<PY
import json
for c in ("bench7b", "cfg4", "cfg2", "cfg5"):
    try:
        d = json.loads(open("$O/r05_mid_%s.json" % c).read().strip().splitlines()[-1])
        print(c, round(d["value"], 4), round(d["ms_per_step"], 1), d.get("rollout_tokens_per_sec"), d.get("phases_ms_per_step"), d.get("hbm_gb"), (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(c, "failed", e)
PY
