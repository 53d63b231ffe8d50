"""Rewards: bit-exact against the known-answer table captured from the reference's own callbacks (tests/golden/rewards_kat.json)."""
import json
import os

import pytest

import time_r1_amd  # noqa: F401
from time_r1_amd import rewards as R

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rewards_kat.json")))["rows"]


def test_kat_size_and_coverage():
    assert len(KAT) >= 50
    assert any(r["parse"] is None for r in KAT) and any(r["parse"] is not None for r in KAT)
    assert any(r["format"] == "1.0" for r in KAT) and any(r["format"] == "0.0" for r in KAT)


def test_rewards_bit_exact():
    n_unbound = 0
    for r in KAT:
        c, sol, dur = r["completion"], tuple(r["solution"]), r["duration"]
        p = R.parse_timestamp_output(c)
        assert (None if p is None else [repr(x) for x in p]) == r["parse"], c
        got = R.iou_timestamp_reward([c], [sol])[0]
        got2 = R.iou_timestamp_reward_v2([c], [sol], durations=[dur])[0]
        if r["iou"] == "UnboundLocalError":     # reference bug: documented deviation -> 0.0
            n_unbound += 1
            assert got == 0.0 and got2 == 0.0
        else:
            assert repr(float(got)) == r["iou"], (c, got, r["iou"])
            assert repr(float(got2)) == r["iou_v2"], (c, got2, r["iou_v2"])
        assert repr(float(R.format_reward([c])[0])) == r["format"], c
        for k, fn in R.metric_funcs_registry.items():
            assert repr(float(fn([c])[0])) == r[k], (k, c)


def test_sanity_anchors():
    # SURVEY appendix B anchors (captured from the reference in the survey session)
    c = "<think>x</think><answer>12.54 to 17.83</answer>"
    assert R.iou_timestamp_reward([c], [(10, 20)])[0] == 0.5289999999999999
    assert R.iou_timestamp_reward_v2([c], [(10, 20)], durations=[30])[0] == 0.4491867135555555
    assert R.format_reward([c]) == [1.0]


def test_callback_protocol_batch():
    comps = ["<think>a</think><answer>1 to 2</answer>", "junk"]
    out = R.iou_timestamp_reward_v2(prompts=None, completions=comps, solution=[(1, 2), (1, 2)], durations=[10, 10], video_path=["a", "b"])
    assert out == [1.0, 0.0]
    assert R.reward_funcs_registry["iou_v2"].__name__ == "iou_timestamp_reward_v2"
    assert R.reward_funcs_registry["format"].__name__ == "format_reward"
