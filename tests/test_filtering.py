"""Per-epoch sample filtering (curriculum loop, reference scripts/posttrain/train_rl_SF.sh:86-110) against a known-answer table captured from
the reference's calc_difficulty.py / process_data.py (tests/golden/gen_filtering_kat.py)."""
import importlib.util
import json
import math
import os
import random

import numpy as np
import pytest

import time_r1_amd  # noqa: F401
from time_r1_amd import filtering as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "filtering_kat.json")))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_filtering_kat", os.path.join(ROOT, "tests", "golden", "gen_filtering_kat.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_difficulty_table_and_shares_match_the_reference():
    table = F.difficulty_table(KAT["records"])
    assert list(table) == list(KAT["table"])
    for q, want in KAT["table"].items():
        got = table[q]
        assert got["pred"] == want["pred"], q
        if want["difficulty"] == "nan":
            assert math.isnan(got["difficulty"]), q          # 0 / 0 union: NaN survives, the selection stage drops it
        else:
            assert float(got["difficulty"]) == want["difficulty"], (q, got["difficulty"], want["difficulty"])       # bit-exact (float64)
    assert F.difficulty_shares(table) == KAT["shares"]
    assert F.extract_answer_force("no numbers") == [None, None] and F.extract_answer_force("1 2 3") == [1.0, 2.0]
    assert F.calc_difficulty([None, None], [1.0, 2.0]) == 0.0


@pytest.mark.parametrize("case", KAT["selections"], ids=lambda c: "%s-n%d-k%d" % (c["task"], c["n"], c["k"]))
def test_selection_rules_pick_the_same_samples_as_the_reference(case):
    items = _gen().synthetic_items(case["n"], case["seed"])
    np.random.seed(case["np_seed"]); random.seed(case["py_seed"])
    picked = F.select_samples(items, case["task"], case["k"])
    assert (None if picked is None else [it["qid"] for it in picked]) == case["qids"]


def test_filter_epoch_writes_the_two_files_of_the_loop(tmp_path):
    items = [{"video": "v%d.mp4" % i, "duration": 30.0, "timestamp": [2.0, 12.0], "pred": [None, None], "sentence": "s%d" % i, "qid": i,
              "video_start": None, "video_end": None, "extra": "dropped by the split loader"} for i in range(6)]
    src = tmp_path / "split.json"
    src.write_text(json.dumps(items))
    rows = F.load_filter_split(str(src))
    assert set(rows[0]) == {"video", "duration", "timestamp", "pred", "sentence", "qid", "video_start", "video_end"}
    completions = ["<answer>2 to 12</answer>", "<answer>2 to 7</answer>", "no idea", "maybe 4 and 9", "<answer>20 to 30</answer>", "<answer>3 to 11</answer>"]
    recs = [{"problem": r["sentence"], "solution": r["timestamp"], "completion": c, "iou": 0.0} for r, c in zip(rows, completions)]
    shares, path = F.filter_epoch(rows, recs, str(tmp_path / "filtering_epoch0"), task="0070_all", k=10)
    scored = json.load(open(tmp_path / "filtering_epoch0" / "train_v4_cloud.json"))
    assert [r["difficulty"] for r in scored] == [100.0, 50.0, 0.0, 50.0, 0.0, 80.0] and shares == [66.7, 33.3, 33.3]
    assert path.endswith("train_v4_cloud_0070_all.json")
    nxt = json.load(open(path))
    assert [r["qid"] for r in nxt] == [1, 3]                       # 0 < p <= 0.7, difficulty-descending, stable
    assert F.load_filter_split(path)[0]["pred"] == [2.0, 7.0]      # the selected set is a valid split for the next stage
    del items[0]["qid"]
    src.write_text(json.dumps(items))
    with pytest.raises(KeyError):
        F.load_filter_split(str(src))
