"""Verifiable rewards for temporal video grounding - drop-in equivalents of the reference's reward / metric callbacks.

Callback protocol (reference src/time_r1/rl/timer1_trainer.py:685-698): fn(prompts=..., completions=list[str], **columns) -> list[float];
`fn.__name__` becomes the metric key `rewards/<name>`, so the names below match the reference's (main.py:145-366, registries :416-428).
The reference's own callbacks can be passed to the trainer unchanged; these exist so the engine is usable stand-alone and are pinned
bit-exactly (Python float64 arithmetic, same operation order) against outputs captured from the reference in tests/golden/rewards_kat.json.

Conscious deviation: when prediction and ground truth are both the same zero-length instant (union <= 0) the
reference reads an unbound / stale `iou`; here the reward is 0.0.
"""
import os
import re
from datetime import datetime
from typing import List, Optional

_ANSWER_RE = re.compile(r"<answer>(.*?)</answer>", re.DOTALL)
_SPAN_RE = re.compile(r"(\d+\.?\d*) (to|and) (\d+\.?\d*)", re.IGNORECASE)
_FORMAT_RE = re.compile(r"<think>.*?</think>\s*<answer>.*?</answer>", re.DOTALL)
_THINK_RE = re.compile(r"<think>(.*?)</think>", re.DOTALL)
_TIMESTEP_RE = re.compile(r"<timestep>\s*(\d+\.?\d*)\s+to\s+(\d+\.?\d*)\s*</timestep>", re.IGNORECASE | re.DOTALL)

STRUCTURE_KEYWORDS = ["analyze", "compare", "deduce", "however", "therefore", "because", "step", "observe", "notice", "identify", "wait"]


def parse_timestamp_output(text: str):
    """Last "<a> to|and <b>" pair inside the LAST <answer> block, else None (reference main.py:122-142)."""
    blocks = _ANSWER_RE.findall(text)
    if not blocks:
        return None
    spans = _SPAN_RE.findall(blocks[-1])
    if not spans:
        return None
    first, _, second = spans[-1]
    return float(first), float(second)


def _tiou(pred, gt):
    ps, pe = pred
    gs, ge = gt
    inter = max(0, min(pe, ge) - max(ps, gs))
    union = max(pe, ge) - min(ps, gs)
    return inter / union if union > 0 else 0.0


def _debug_log(content, pred, gt, reward, stamp):
    if os.getenv("DEBUG_MODE") == "true" and os.getenv("LOG_PATH"):
        with open(os.getenv("LOG_PATH"), "a", encoding="utf-8") as f:
            f.write("Content: %s\npred second: %s, %s\ngt second: %s, %s\n------------- %s IoU reward: %s -------------\n"
                    % (content, pred[0], pred[1], gt[0], gt[1], stamp, reward))


def iou_timestamp_reward(completions, solution, **kwargs):
    """tIoU between the predicted and ground-truth span (reference main.py:145-181)."""
    stamp = datetime.now().strftime("%d-%H-%M-%S-%f")
    out = []
    for text, gt in zip(completions, solution):
        pred = parse_timestamp_output(text)
        reward = _tiou(pred, gt) if pred else 0.0
        out.append(reward)
        _debug_log(text, pred or (0, 0), gt, reward, stamp)
    return out


def iou_timestamp_reward_v2(completions, solution, **kwargs):
    """tIoU scaled by (1 - |ds|/dur)(1 - |de|/dur) (reference main.py:184-231)."""
    stamp = datetime.now().strftime("%d-%H-%M-%S-%f")
    out = []
    for text, gt, dur in zip(completions, solution, kwargs.get("durations")):
        pred = parse_timestamp_output(text)
        reward = 0.0
        if pred:
            gs, ge = gt
            iou = _tiou(pred, gt)
            gs_n, ge_n = 1.0 * gs / dur, 1.0 * ge / dur
            ps_n, pe_n = 1.0 * pred[0] / dur, 1.0 * pred[1] / dur
            reward = iou * (1 - abs(gs_n - ps_n)) * (1 - abs(ge_n - pe_n))
        out.append(reward)
        _debug_log(text, pred or (0, 0), gt, reward, stamp)
    return out


def format_reward(completions, **kwargs):
    """1.0 iff the stripped completion is exactly <think>...</think> <answer>...</answer> (reference main.py:234-239)."""
    return [1.0 if _FORMAT_RE.fullmatch(c.strip()) else 0.0 for c in completions]


def extract_think_content(completion: str) -> Optional[str]:
    found = _THINK_RE.findall(completion)
    return found[-1].strip() if found else None


def _think_score(completions, fn):
    out = []
    for c in completions:
        think = extract_think_content(c)
        out.append(max(0.0, fn(think)) if think else 0.0)
    return out


def reward_timestep_pair(completions: List[str], weight: float = 0.2, max_count: int = 1, **kwargs) -> List[float]:
    return _think_score(completions, lambda th: weight * min(len(_TIMESTEP_RE.findall(th)), max_count))


def reward_think_length(completions: List[str], weight: float = 0.001, max_length: int = 500, **kwargs) -> List[float]:
    return _think_score(completions, lambda th: weight * min(len(th), max_length))


def reward_keyword_usage(completions: List[str], keywords: Optional[List[str]] = None, weight: float = 0.1, max_count: int = 2, **kwargs) -> List[float]:
    kws = STRUCTURE_KEYWORDS if keywords is None else keywords
    return _think_score(completions, lambda th: weight * min(sum(1 for k in kws if k in th.lower()), max_count))


def reward_paragraph_structure(completions: List[str], weight: float = 0.05, max_paragraphs: int = 2, **kwargs) -> List[float]:
    return _think_score(completions, lambda th: weight * min(len([p for p in th.split("\n") if p.strip()]), max_paragraphs))


reward_funcs_registry = {"iou": iou_timestamp_reward, "iou_v2": iou_timestamp_reward_v2, "format": format_reward}
metric_funcs_registry = {
    "reward_timestep_pair": reward_timestep_pair,
    "reward_think_length": reward_think_length,
    "reward_keyword_usage": reward_keyword_usage,
    "reward_paragraph_structure": reward_paragraph_structure,
}
