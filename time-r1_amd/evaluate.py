"""In-engine evaluation for temporal grounding: greedy rollout with the training engine (no vLLM), the reference's answer regex and
its IoU / R1@k / mIoU aggregation (reference evaluate.py:125-149 answer extraction, src/vllm_inference/eval_all.py:65-137 metrics)."""
import re

import numpy as np


_SPAN = re.compile(r"(\d+\.?\d*) (to|and) (\d+\.?\d*)")


def extract_answer_span(text):
    """The reference's `extract_answer(..., "tg")` (evaluate.py:125-149): the LAST 'a to b' / 'a and b' pair anywhere in the output
    (case-sensitive); only when the whole string has none, the first <answer>..</answer> body (single line) is searched."""
    s = _SPAN.findall(text)
    if not s:
        m = re.search(r"<answer>(.*?)</answer>", text)
        s = _SPAN.findall(m.group(1).strip()) if m else []
        if not s:
            return None
    return float(s[-1][0]), float(s[-1][2])


def compute_iou(pred, gt):
    """Temporal IoU of two [start, end] spans (eval_all.py:65-86 semantics: intersection clipped at 0, union of the hull)."""
    if pred is None:
        return 0.0
    (ps, pe), (gs, ge) = pred, gt
    inter = max(0.0, min(pe, ge) - max(ps, gs))
    union = max(pe, ge) - min(ps, gs)
    return inter / union if union > 0 else 0.0


def grounding_metrics(ious, thresholds=(0.3, 0.5, 0.7)):
    ious = np.asarray(ious, dtype=np.float64)
    out = {"mIoU": float(ious.mean() * 100) if ious.size else 0.0}
    for t in thresholds:
        out["R1@%.1f" % t] = float((ious > t).mean() * 100) if ious.size else 0.0     # strict, like eval_all.py:129
    out["avg"] = sum(out.values()) / len(out)                                          # eval_all.py:131
    return out


def evaluate_grounding(trainer, dataset, max_new_tokens=None, limit=None):
    """Greedy-decodes one completion per row with the trainer's engine and scores it. Returns (metrics, per-row records)."""
    from .grpo import GRPOCore
    a = trainer.args
    core = GRPOCore(trainer.engine, None, 1, max_new_tokens or trainer.max_completion_length, beta=0.0, temperature=1.0, top_k=1, seed=0,
                    rope_index_mode=a.rope_index_mode, stop_at_eos=True, reuse_prefill=False)
    records, ious = [], []
    n = len(dataset) if limit is None else min(limit, len(dataset))
    for i in range(n):
        row = dataset[i]
        video_inputs, fps_inputs = trainer._video_inputs(row)
        conv = trainer.make_conversation_video(row)
        text = trainer.processing_class.apply_chat_template(conv, tokenize=False, add_generation_prompt=True)
        pi = trainer.processing_class(text=[text], images=None, videos=[video_inputs[0]], fps=[fps_inputs[0]], padding=True, return_tensors="pt",
                                      padding_side="left", add_special_tokens=False)
        st = core.prepare(np.asarray(pi["input_ids"]).reshape(-1), pi["pixel_values_videos"], np.asarray(pi["video_grid_thw"]))
        toks = core.rollout(st).cpu()
        completion = trainer.processing_class.batch_decode(toks, skip_special_tokens=True)[0]
        iou = compute_iou(extract_answer_span(completion), row["solution"])
        ious.append(iou)
        records.append({"problem": row["problem"], "solution": list(row["solution"]), "completion": completion, "iou": iou})
    return grounding_metrics(ious), records
