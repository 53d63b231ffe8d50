"""Known-answer table for the per-epoch sample filtering, captured from the reference's own code:
  compute_IoU / calc_difficulty / extract_answer_force / load_new_data / calc_score   /root/reference/src/vllm_inference/calc_difficulty.py
  process_ddata (tasks 0070_all, gaussian_03, random_sample)                          /root/reference/src/utils/process_data.py
calc_difficulty.py imports its data loader at the top, so its function definitions are compiled on their own from the file where it lies
(ast -> exec, unmodified bodies); process_data.py is imported as a module.  Inputs are synthetic.
Run in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_filtering_kat.py   -> tests/golden/filtering_kat.json
"""
import ast
import contextlib
import importlib.util
import io
import json
import math
import os
import random
import re
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_functions(path, names):
    tree = ast.parse(open(path).read())
    ns = {"re": re, "np": np, "json": json, "os": os}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return ns


def jsonable(x):
    if isinstance(x, (np.floating, float)):
        x = float(x)
        return x if math.isfinite(x) else repr(x)
    if isinstance(x, list):
        return [jsonable(v) for v in x]
    return x


def synthetic_items(n, seed):
    rng = random.Random(seed)
    items = []
    for i in range(n):
        d = rng.choice([0.0, 0.0, rng.uniform(0, 100), rng.uniform(0, 100), rng.uniform(0, 70), 70.0, 100.0])
        if i % 11 == 3:
            d = None
        if i % 13 == 5:
            d = "nan"
        if i % 17 == 7:
            d = str(round(rng.uniform(1, 60), 3))
        if i % 19 == 9:
            d = float("inf")
        items.append({"video": "v%03d.mp4" % i, "duration": 30.0 + i, "timestamp": [1.0, 2.0 + i % 5], "pred": [None, None], "sentence": "event %d" % i,
                      "qid": "q%03d" % i, "video_start": None, "video_end": None, "difficulty": d})
    return items


def main():
    ns = load_functions(os.path.join(REF, "src/vllm_inference/calc_difficulty.py"),
                        ["compute_IoU", "calc_difficulty", "extract_answer_force", "load_new_data", "calc_score"])
    spec = importlib.util.spec_from_file_location("ref_process_data", os.path.join(REF, "src/utils/process_data.py"))
    pd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pd)

    records = []
    texts = ["<answer>3.5 to 9</answer>", "from 2 and 4 seconds", "nothing", "12.54 17.83", "the 1st event is at 20 to 40", "10", "7.5 to 7.5", "0 to 0",
             "between 100.25 and 5", "<think>3 things</think><answer>4 to 8</answer>"]
    preds = [[3.5, 9.0], [None, None], [None, None], [None, None], [20.0, 40.0], [None, None], [7.5, 7.5], [0.0, 0.0], [None, None], [4.0, 8.0]]
    targets = [[2.0, 12.0], [1.0, 5.0], [0.0, 5.5], [10.0, 20.0], [33.3, 66.6], [1.0, 2.0], [7.5, 7.5], [0.0, 0.0], [0.0, 50.0], [4.0, 8.0]]
    for i, (t, p, g) in enumerate(zip(texts, preds, targets)):
        records.append({"qid": "q%03d" % i, "pred": p, "target": g, "output_text": t})
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "part0.jsonl"), "w") as f:
            for r in records[:6]:
                f.write(json.dumps(r) + "\n")
        with open(os.path.join(td, "part1.jsonl"), "w") as f:
            for r in records[6:]:
                f.write(json.dumps(r) + "\n")
        with np.errstate(all="ignore"):
            table = ns["load_new_data"](td)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ns["calc_score"](table)
    shares = [float(x) for x in buf.getvalue().split()]
    out = {"records": records, "table": {q: {"difficulty": jsonable(v["difficulty"]), "pred": jsonable(v["pred"])} for q, v in table.items()},
           "shares": shares, "selections": []}

    for n, seed in ((40, 1), (200, 2), (7, 3)):
        items = synthetic_items(n, seed)
        for task, suffix in (("0070_all", "_0070_all.json"), ("gaussian_03", "_gaussian_03.json"), ("random_sample", "_random.json")):
            for k in (5, 12, 1000):
                with tempfile.TemporaryDirectory() as td:
                    src = os.path.join(td, "in.json")
                    with open(src, "w") as f:
                        json.dump(items, f)
                    np.random.seed(100 + k); random.seed(200 + k)
                    with contextlib.redirect_stdout(io.StringIO()):
                        pd.process_ddata(src, os.path.join(td, "out"), task, k)
                    path = os.path.join(td, "out" + suffix)
                    qids = [it["qid"] for it in json.load(open(path))] if os.path.exists(path) else None
                out["selections"].append({"n": n, "seed": seed, "task": task, "k": k, "np_seed": 100 + k, "py_seed": 200 + k, "qids": qids})
    with open(os.path.join(HERE, "filtering_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["table"]), "difficulties,", len(out["selections"]), "selections")


if __name__ == "__main__":
    main()
