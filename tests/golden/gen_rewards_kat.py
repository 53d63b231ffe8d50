"""Known-answer table for the reward / metric callbacks, captured from the reference's own functions (main.py:122-366).
Run in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_rewards_kat.py   -> tests/golden/rewards_kat.json
Floats are stored with repr() so the comparison is bit-exact."""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(__file__))
from _ref_harness import load_ref_main  # noqa: E402

THINKS = [
    "", "<think>short</think>", "<think>I observe the person.\nTherefore the event is <timestep>3 to 9.5</timestep>.</think>",
    "<think>step one: analyze.\n\nstep two: compare, however wait because notice identify deduce.</think>",
    "<think>" + "x" * 700 + "</think>", "<think>first</think><think>second block\nline2\nline3</think>",
    "<THINK>wrong case</THINK>", "<think>unterminated",
]
ANSWERS = [
    "<answer>12.54 to 17.83</answer>", "<answer>The event happens from 3 to 9 seconds</answer>", "<answer>5 and 20.</answer>",
    "<answer>10 TO 30</answer>", "<answer>7.5  to 9</answer>", "<answer>nothing</answer>", "<answer>1 to 2</answer> <answer>20 to 40</answer>",
    "<answer>0.0 to 100.25</answer>", "<answer>25 to 5</answer>", "<answer>3 to 4 and 6 to 7</answer>", "<answer>\n2.0 to 12.0\n</answer>",
    "no tags 2 to 12", "<answer>2. to 12.</answer>", "<answer>90 to 95</answer>", "<answer>-3 to 8</answer>",
]
GTS = [((2.0, 12.0), 30.0), ((10, 20), 30), ((0.0, 5.5), 7.25), ((33.3, 66.6), 99.9)]


def main():
    ref = load_ref_main()
    comps = []
    for t in THINKS:
        for a in ANSWERS:
            comps.append(t + a)
            if t and len(comps) % 3 == 0:
                comps.append("  " + t + "\n\n" + a + "  \n")
            if len(comps) % 7 == 0:
                comps.append(t + a + " trailing text")
    rows = []
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        for (gt, dur) in GTS:
            n = len(comps)
            sol, durs = [list(gt)] * n, [dur] * n
            iou, iou2 = [], []
            for c in comps:  # one call per completion: the reference's stale-`iou` quirk must not leak between items
                try:
                    iou.append(ref.iou_timestamp_reward([c], [list(gt)])[0])
                except UnboundLocalError:
                    iou.append("UnboundLocalError")
                try:
                    iou2.append(ref.iou_timestamp_reward_v2([c], [list(gt)], durations=[dur])[0])
                except UnboundLocalError:
                    iou2.append("UnboundLocalError")
            fmt = ref.format_reward(comps)
            mets = {k: f(comps) for k, f in ref.metric_funcs_registry.items()}
            parsed = [ref.parse_timestamp_output(c) for c in comps]
            for i, c in enumerate(comps):
                rows.append({"completion": c, "solution": list(gt), "duration": dur, "parse": None if parsed[i] is None else [repr(x) for x in parsed[i]],
                             "iou": iou[i] if isinstance(iou[i], str) else repr(float(iou[i])),
                             "iou_v2": iou2[i] if isinstance(iou2[i], str) else repr(float(iou2[i])), "format": repr(float(fmt[i])),
                             **{k: repr(float(v[i])) for k, v in mets.items()}})
    out = os.path.join(os.path.dirname(__file__), "rewards_kat.json")
    json.dump({"source": "reference main.py:122-366 via tests/golden/gen_rewards_kat.py", "rows": rows}, open(out, "w"), indent=0)
    print(len(rows), "rows ->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
