"""Dataset rows in the reference's schemas (the trainer's identity collator hands them to compute_loss unchanged).

  post-training : `load_json_dataset_tg`  - reference main.py:431-494 (TimeRFT json: video, timestamp, sentence, duration, video_start/end)
  fine-tuning   : `load_json_dataset`     - reference finetune.py:541-632 (per-video sentences/timestamps + pre-decoded clips on disk:
                                            <preprocessed>/<video_id>/video_inputs.pt = [float T x 3 x H x W], video_kwargs.json = {"fps": [f]};
                                            written by src/utils/preprocess_dataset.py:78-93)
`save_preprocessed` / `load_preprocessed` are the writer / reader of that on-disk clip format.
"""
import json
import os
import random

import torch


class RowDataset:
    """Minimal map-style dataset of dict rows (what the trainer needs from datasets.Dataset: len() and integer indexing)."""

    def __init__(self, rows, getitem=None):
        self.rows = rows
        self._getitem = getitem

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        row = self.rows[i]
        return self._getitem(row) if self._getitem else row


def _clean_sentence(s):
    s = s.strip().lower()
    return s[:-1] if s.endswith(".") else s


def load_json_dataset_tg(train_data_path, is_curriculum_learning=False, preprocessed_data_path=None, require_files=True):
    """TimeRFT / TVGBench style list of items -> rows {task_type, problem, choices, solution, video_path, durations, video_start,
    video_end, preprocessed_path}. Items whose video file is missing are skipped (reference :452-453) unless require_files=False."""
    with open(train_data_path, "r", encoding="utf-8") as f:
        data = json.load(f)
    rows = []
    for item in data:
        video_path = item.get("video")
        if require_files and not (video_path and os.path.isfile(video_path)):
            continue
        ts = item.get("timestamp")
        rows.append({"task_type": "tg", "problem": _clean_sentence(item.get("sentence")), "choices": "",
                     "solution": (float(ts[0]), float(ts[1])), "video_path": video_path, "durations": item.get("duration"),
                     "video_start": item.get("video_start"), "video_end": item.get("video_end"), "preprocessed_path": ""})
    if not rows:
        return None
    if not is_curriculum_learning:
        random.shuffle(rows)
    return RowDataset(rows)


def load_preprocessed(path):
    """-> (video_inputs list[float tensor T x 3 x H x W], video_kwargs dict) from a pre-decoded clip directory."""
    vids = torch.load(os.path.join(path, "video_inputs.pt"), weights_only=False)
    with open(os.path.join(path, "video_kwargs.json"), "r") as f:
        kw = json.load(f)
    return vids, kw


def save_preprocessed(path, video_inputs, video_kwargs):
    os.makedirs(path, exist_ok=True)
    torch.save(list(video_inputs), os.path.join(path, "video_inputs.pt"))
    with open(os.path.join(path, "video_kwargs.json"), "w") as f:
        json.dump(video_kwargs, f)


def load_json_dataset(train_data_path, video_folder, preprocessed_data_path=None):
    """Charades / ActivityNet style {video_id: {duration, timestamps, sentences}} -> rows carrying the pre-decoded clip
    (`video_inputs`, `video_kwargs`, `use_preprocessed`) loaded lazily in __getitem__, as in reference finetune.py:594-623."""
    with open(train_data_path, "r") as f:
        data = json.load(f)
    rows = []
    for video_id, vd in data.items():
        for ts, sentence in zip(vd["timestamps"], vd["sentences"]):
            video_path = None
            for ext in ("mp4", "mkv", "webm"):
                cand = os.path.join(video_folder, "%s.%s" % (video_id, ext))
                if os.path.isfile(cand):
                    video_path = cand
                    break
            rows.append({"problem": _clean_sentence(sentence), "solution": (ts[0], ts[1]), "video_path": video_path, "durations": vd["duration"],
                         "video_start": None, "video_end": None,
                         "preprocessed_path": os.path.join(preprocessed_data_path, video_id) if preprocessed_data_path else ""})
    random.shuffle(rows)

    def getitem(row):
        out = dict(row)
        if row["preprocessed_path"] and os.path.isdir(row["preprocessed_path"]):
            vids, kw = load_preprocessed(row["preprocessed_path"])
            out["video_inputs"], out["video_kwargs"], out["use_preprocessed"] = vids, kw, True
        return out
    return RowDataset(rows, getitem)
