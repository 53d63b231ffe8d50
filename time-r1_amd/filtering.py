"""Per-epoch sample filtering of the curriculum loop (reference scripts/posttrain/train_rl_SF.sh:86-110): after an epoch the
model answers every training query, each sample gets difficulty = 100 x tIoU of that answer, and the next epoch trains on a
subset chosen by difficulty.  Restates src/vllm_inference/calc_difficulty.py (difficulty, forced answer extraction, the
>30 / >50 / >70 shares) and src/utils/process_data.py (the three selection rules); the predictions come from the in-engine
greedy evaluation (evaluate.py) instead of vLLM.  Host-side data plumbing: python floats, numpy, and torch only where the
reference itself uses torch (the linspace index rule), so the selected sample lists are identical to the reference's.
"""
import json
import math
import os
import random
import re

import numpy as np
import torch


def compute_iou(pred, gt):
    """calc_difficulty.py:10-32 for one [start, end] pair each (numpy float64, hull union; 0 / 0 stays NaN like the reference)."""
    pred, gt = np.array([pred]), np.array([gt])
    inter = np.maximum(0.0, np.minimum(pred[:, 1, None], gt[None, :, 1]) - np.maximum(pred[:, 0, None], gt[None, :, 0]))
    union = np.maximum(0.0, np.maximum(pred[:, 1, None], gt[None, :, 1]) - np.minimum(pred[:, 0, None], gt[None, :, 0]))
    with np.errstate(divide="ignore", invalid="ignore"):
        return (1.0 * inter / union)[0, 0]


def calc_difficulty(pred, gt):
    """calc_difficulty.py:35-38: 0.0 for an unparsed answer, else 100 x IoU."""
    if None in pred:
        return 0.0
    return compute_iou(list(pred), list(gt)) * 100.0


def extract_answer_force(output_string):
    """calc_difficulty.py:41-47: the first two numbers anywhere in the output ("may not follow the rules but still be correct")."""
    matches = re.findall(r"\d+(?:\.\d+)?", output_string)
    out = [float(m) for m in matches[:2]]
    return out if len(out) == 2 else [None, None]


def difficulty_table(records):
    """records: dicts with qid, pred ([s, e] or [None, None]), target, output_text (the jsonl lines of the reference's inference
    stage, calc_difficulty.py:50-67) -> {qid: {"difficulty", "pred"}}."""
    data = {}
    for r in records:
        pred = r["pred"]
        if pred is None or None in pred:
            pred = extract_answer_force(r["output_text"])
        data[r["qid"]] = {"difficulty": calc_difficulty(pred, r["target"]), "pred": pred}
    return data


def difficulty_shares(table, thresholds=(30.0, 50.0, 70.0)):
    """calc_difficulty.py:70-75: percentage of samples with difficulty above each threshold, rounded to one decimal."""
    vals = list(table.values())
    return [round(len([v for v in vals if v["difficulty"] > t]) / len(table) * 100, 1) for t in thresholds]


def attach_difficulty(rows, table):
    """calc_difficulty.py:86-92: rows that were evaluated, in their original order, with `difficulty` and `pred` attached."""
    out = []
    for row in rows:
        if row["qid"] in table:
            row = dict(row)
            row["difficulty"] = table[row["qid"]]["difficulty"]
            row["pred"] = table[row["qid"]]["pred"]
            out.append(row)
    return out


_SPLIT_KEYS = ("video", "duration", "timestamp", "pred", "sentence", "qid", "video_start", "video_end")


def load_filter_split(path):
    """src/vllm_inference/data/data_loader.py:84-112 (load_tvgbench_filter): the annotation items of the split being filtered, reduced to the
    eight keys the reference keeps (a missing key raises KeyError, as there)."""
    with open(path, "r", encoding="utf-8") as f:
        data = json.load(f)
    return [{k: item[k] for k in _SPLIT_KEYS} for item in data]


def records_from_evaluation(rows, eval_records):
    """evaluate.evaluate_grounding's per-row records -> the inference-stage schema (qid, pred, target, output_text).  `rows` are the
    annotation items the evaluated dataset was built from, in the same order (one record per item)."""
    from .evaluate import extract_answer_span
    assert len(rows) == len(eval_records), (len(rows), len(eval_records))
    out = []
    for row, rec in zip(rows, eval_records):
        span = extract_answer_span(rec["completion"])
        out.append({"qid": row["qid"], "pred": list(span) if span is not None else [None, None], "target": list(rec["solution"]),
                    "output_text": rec["completion"]})
    return out


def _difficulty_safe(item):
    """process_data.py:11-24."""
    d = item.get("difficulty") if isinstance(item, dict) else None
    if d is None:
        return None
    try:
        f = float(d)
    except (ValueError, TypeError):
        return None
    return None if (math.isnan(f) or math.isinf(f)) else f


def select_samples(rows, task, k=2500):
    """process_data.py:114-152.  task "0070_all": 0 < p <= 0.7, sorted by difficulty (descending, stable), k indices
    torch.linspace(0, n-1, k).round() with duplicates removed; "gaussian_03": p > 0, k draws without replacement with weights
    exp(-(p - 0.3)^2 / (2 * 0.2^2)) from numpy's GLOBAL generator; "random_sample": random.sample from python's global generator.
    Returns the selected rows (None where the reference writes no file)."""
    valid = []
    for item in rows:
        d = _difficulty_safe(item)
        if isinstance(item, dict) and d is not None:
            valid.append({"difficulty_float": d, "p_value": d / 100.0, "data": item})
    if not valid:
        return None
    if task == "0070_all":
        sub = [v for v in valid if 0 < v["p_value"] <= 0.7]
        if not sub or k <= 0:
            return None
        n = len(sub)
        srt = sorted(sub, key=lambda x: x["difficulty_float"], reverse=True)
        if min(n, k) >= n:
            picked = srt
        else:
            idx = torch.unique(torch.clamp(torch.linspace(0, n - 1, steps=min(n, k)).round().long(), 0, n - 1))
            picked = [srt[i] for i in idx]
    elif task == "gaussian_03":
        sub = [v for v in valid if v["p_value"] > 0]
        if not sub or k <= 0 or min(len(sub), k) == 0:
            return None
        p = np.exp(-((np.array([v["difficulty_float"] / 100.0 for v in sub]) - 0.3) ** 2) / (2 * 0.2 ** 2))
        p /= np.sum(p)
        try:
            picked = [sub[i] for i in np.random.choice(len(sub), k, False, p=p)]
        except ValueError:
            return None
    elif task == "random_sample":
        kk = min(len(valid), k)
        picked = valid if kk >= len(valid) else random.sample(valid, kk)
    else:
        return None
    return [v["data"] for v in picked] or None


_SUFFIX = {"0070_all": "_0070_all.json", "gaussian_03": "_gaussian_03.json", "random_sample": "_random.json"}


def filter_epoch(rows, eval_records, out_dir, task="0070_all", k=2500, name="train_v4_cloud"):
    """One filtering stage of train_rl_SF.sh: difficulty from this epoch's answers -> `<out_dir>/<name>.json` (every evaluated row
    with difficulty / pred, calc_difficulty.py:99-102) and `<out_dir>/<name>_<task>.json` (the next epoch's training set,
    process_data.py:27-43: indent 4, ensure_ascii False).  Returns (shares, path of the selected set or None)."""
    table = difficulty_table(records_from_evaluation(rows, eval_records))
    shares = difficulty_shares(table)
    scored = attach_difficulty(rows, table)
    os.makedirs(out_dir, exist_ok=True)
    base = os.path.join(out_dir, name)
    with open(base + ".json", "w") as f:
        json.dump(scored, f)
    picked = select_samples(scored, task, k)
    if not picked:
        return shares, None
    path = base + _SUFFIX[task]
    with open(path, "w", encoding="utf-8") as f:
        json.dump(picked, f, indent=4, ensure_ascii=False)
    return shares, path
