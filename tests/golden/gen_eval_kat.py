"""Known-answer table for the evaluation helpers, captured from the reference's own functions:
  extract_answer(output, "tg")   /root/reference/evaluate.py:125-149
  compute_IoU(pred, gt)          /root/reference/src/vllm_inference/eval_all.py:65-86
  calc_score(scores, name)       /root/reference/src/vllm_inference/eval_all.py:121-137
Those modules import vllm / requests / decord-backed loaders at the top, so the three function definitions are compiled on their own
from the reference files where they lie (ast -> exec, unmodified bodies) instead of importing the modules.
Run in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_eval_kat.py   -> tests/golden/eval_kat.json
"""
import ast
import json
import os
import re

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_functions(path, names):
    tree = ast.parse(open(path).read())
    ns = {"re": re, "np": np, "json": json, "os": os}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return [ns[n] for n in names]


OUTPUTS = [
    "<think>x</think><answer>3.5 to 9</answer>", "about 2 and 4 seconds", "nothing", "<answer>12.54 to 17.83</answer>",
    "<think>from 1 to 2 maybe</think><answer>20 to 40</answer>", "<answer>1 to 2</answer> <answer>20 to 40</answer>", "<answer>10 TO 30</answer>",
    "<answer>\n2.0 to 12.0\n</answer>", "<answer>25 to 5</answer>", "<answer>3 to 4 and 6 to 7</answer>", "<answer>2. to 12.</answer>",
    "<answer>-3 to 8</answer>", "<ANSWER>5 to 6</ANSWER>", "<answer>7.5  to 9</answer>", "the event spans 0.0 to 100.25 overall", "5 and 20.", "",
]
GTS = [[2.0, 12.0], [10, 20], [0.0, 5.5], [33.3, 66.6]]
SCORE_SETS = [[0.2, 0.4, 0.6, 0.8], [0.3, 0.5, 0.7], [0.0], [1.0, 0.30000001, 0.5, 0.69999], [0.31, 0.29, 0.71, 0.51, 0.49]]


def main():
    (extract_answer,) = load_functions(os.path.join(REF, "evaluate.py"), ["extract_answer"])
    compute_IoU, calc_score = load_functions(os.path.join(REF, "src/vllm_inference/eval_all.py"), ["compute_IoU", "calc_score"])
    rows = []
    for o in OUTPUTS:
        pred = extract_answer(o, "tg")
        ious = []
        for gt in GTS:
            if None in pred:
                ious.append(repr(0.0))                  # load_scored_data: score stays 0.0 when the prediction is missing
            else:
                with np.errstate(all="ignore"):
                    ious.append(repr(float(compute_IoU(list(pred), list(gt)))))
        rows.append({"output": o, "pred": pred, "ious": ious})
    scores = []
    for s in SCORE_SETS:
        sc = calc_score({i: v for i, v in enumerate(s)}, "charades")
        scores.append({"ious": s, "scores": {str(k): repr(float(v)) for k, v in sc.items()}})
    json.dump({"gts": GTS, "rows": rows, "scores": scores}, open(os.path.join(HERE, "eval_kat.json"), "w"), indent=1)
    print(len(rows), "outputs,", len(scores), "score sets")


if __name__ == "__main__":
    main()
