"""Dataset row schemas, the pre-decoded clip format and the evaluation metrics (CPU)."""
import json
import os

import pytest
import torch

import time_r1_amd  # noqa: F401
from time_r1_amd import data as D
from time_r1_amd import evaluate as E

REF_JSON = "/root/reference/dataset/timer1/annotations/train_2k5.json"


def test_tg_rows_schema(tmp_path):
    items = [{"video": str(tmp_path / "a.mp4"), "timestamp": [1, 5.5], "sentence": " Person Opens a Door. ", "duration": 30.2, "video_start": None, "video_end": None},
             {"video": str(tmp_path / "missing.mp4"), "timestamp": [0, 1], "sentence": "x", "duration": 3}]
    (tmp_path / "a.mp4").write_bytes(b"")
    p = tmp_path / "t.json"
    p.write_text(json.dumps(items))
    ds = D.load_json_dataset_tg(str(p))
    assert len(ds) == 1
    assert ds[0] == {"task_type": "tg", "problem": "person opens a door", "choices": "", "solution": (1.0, 5.5), "video_path": str(tmp_path / "a.mp4"),
                     "durations": 30.2, "video_start": None, "video_end": None, "preprocessed_path": ""}
    assert len(D.load_json_dataset_tg(str(p), require_files=False)) == 2


@pytest.mark.skipif(not os.path.exists(REF_JSON), reason="reference annotations only exist in the build container")
def test_reads_the_reference_annotation_file():
    ds = D.load_json_dataset_tg(REF_JSON, is_curriculum_learning=True, require_files=False)
    assert len(ds) == 2500 and set(ds[0]) == {"task_type", "problem", "choices", "solution", "video_path", "durations", "video_start", "video_end", "preprocessed_path"}


def test_predecoded_clip_roundtrip_and_ft_rows(tmp_path):
    clip = [torch.rand(4, 3, 56, 84) * 255]
    D.save_preprocessed(str(tmp_path / "pre" / "vid1"), clip, {"fps": [2.0]})
    ann = {"vid1": {"duration": 12.0, "timestamps": [[1.0, 4.0], [5, 9]], "sentences": ["A man runs.", "he sits"]}}
    p = tmp_path / "ft.json"
    p.write_text(json.dumps(ann))
    ds = D.load_json_dataset(str(p), str(tmp_path), str(tmp_path / "pre"))
    assert len(ds) == 2
    row = ds[0]
    assert row["use_preprocessed"] is True and row["video_kwargs"] == {"fps": [2.0]} and torch.equal(row["video_inputs"][0], clip[0])
    assert row["problem"] in ("a man runs", "he sits") and row["video_path"] is None


def test_metrics():
    assert E.extract_answer_span("<think>x</think><answer>3.5 to 9</answer>") == (3.5, 9.0)
    assert E.extract_answer_span("about 2 and 4 seconds") == (2.0, 4.0)
    assert E.extract_answer_span("nothing") is None
    assert E.compute_iou((3.0, 9.0), (2.0, 12.0)) == 0.6
    assert E.compute_iou(None, (0, 1)) == 0.0
    m = E.grounding_metrics([0.2, 0.4, 0.6, 0.8])
    assert m == {"mIoU": 50.0, "R1@0.3": 75.0, "R1@0.5": 50.0, "R1@0.7": 25.0, "avg": 50.0}


def test_eval_helpers_match_reference_kat():
    """extract_answer / compute_IoU / calc_score outputs captured from the reference (tests/golden/gen_eval_kat.py): bit-exact."""
    import json
    k = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_kat.json")))
    for r in k["rows"]:
        p = E.extract_answer_span(r["output"])
        assert p == (None if None in r["pred"] else tuple(r["pred"])), r["output"]
        for gt, want in zip(k["gts"], r["ious"]):
            assert repr(float(E.compute_iou(p, gt))) == want, (r["output"], gt)
    for s in k["scores"]:
        m = E.grounding_metrics(s["ious"])
        assert {"mIoU": repr(m["mIoU"]), "0.3": repr(m["R1@0.3"]), "0.5": repr(m["R1@0.5"]), "0.7": repr(m["R1@0.7"]), "avg": repr(m["avg"])} == s["scores"]


def test_cli_parses_reference_style_flags(monkeypatch):
    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("train_grpo", os.path.join(os.path.dirname(os.path.dirname(__file__)), "train_grpo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cb = mod.StopAfterNEpochsCallback(1)
    import types
    control = types.SimpleNamespace(should_training_stop=False)
    cb.on_epoch_end(None, types.SimpleNamespace(epoch=1.0), control)
    assert control.should_training_stop and mod.str2bool("True") and not mod.str2bool("false")


def test_in_engine_greedy_evaluation_runs_on_cpu_oracle_ops():
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from helpers import load_case
    from test_trainer_host_logic import make_trainer, frames_for
    fx = load_case("clip_nobeta")
    cfg, tr = make_trainer(fx)
    rows = []
    for i in range(2):
        r = dict(fx["row"]); r["video_frames"] = frames_for(fx); rows.append(r)
    tr._video_inputs = lambda ex: ([ex["video_frames"]], [2.0])
    m1, rec1 = E.evaluate_grounding(tr, D.RowDataset(rows), max_new_tokens=6)
    m2, rec2 = E.evaluate_grounding(tr, D.RowDataset(rows), max_new_tokens=6)
    assert set(m1) == {"mIoU", "R1@0.3", "R1@0.5", "R1@0.7", "avg"} and len(rec1) == 2
    assert [r["completion"] for r in rec1] == [r["completion"] for r in rec2], "greedy decoding is deterministic"
