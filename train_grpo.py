#!/usr/bin/env python
"""Command-line entry mirroring the reference's main.py / finetune.py wiring (main.py:550-625, finetune.py:666-728) on the MI355X engine.

    torchrun --nproc_per_node 8 train_grpo.py --model_name_or_path <hf dir> --train_data_path dataset/timer1/annotations/train_2k5.json \\
        --output_dir out --reward_funcs iou_v2 format --num_generations 8 --max_completion_length 200 --beta 0.04 \\
        --gradient_accumulation_steps 2 --num_train_epochs 5 --use_grpo false [--finetune --video_folder ... --preprocessed_data_path ...]

Flags keep the reference's names; DeepSpeed / attention / checkpointing flags are accepted and ignored (nothing to configure).
"""
import argparse
import dataclasses
import json
import math
import os
import random

import numpy as np
import torch

import time_r1_amd  # noqa: F401
from time_r1_amd.data import load_json_dataset, load_json_dataset_tg
from time_r1_amd.dist import init_from_env
from time_r1_amd.rewards import metric_funcs_registry, reward_funcs_registry
from time_r1_amd.trainer import GRPOConfig, TimeR1_Trainer, TimeR1_Trainer_ft


def str2bool(v):
    return str(v).lower() in ("1", "true", "yes", "y")


class StopAfterNEpochsCallback:
    """reference main.py:520-539: stop after `n` epochs (the curriculum script chains one epoch per launch)."""

    def __init__(self, num_epochs_to_train=1):
        self.n = num_epochs_to_train

    def on_epoch_end(self, args, state, control, **kw):
        if state.epoch is not None and round(state.epoch) >= self.n:
            control.should_training_stop = True


def set_global_seed(seed=42):
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)


def main():
    ap = argparse.ArgumentParser()
    cfg_fields = {f.name: f for f in dataclasses.fields(GRPOConfig)}
    for name, f in cfg_fields.items():
        if f.type in (bool, "bool") or isinstance(f.default, bool):
            ap.add_argument("--" + name, type=str2bool, default=f.default)
        elif isinstance(f.default, (int, float, str)) and f.default is not None:
            ap.add_argument("--" + name, type=type(f.default), default=f.default)
        else:
            ap.add_argument("--" + name, default=f.default)
    ap.add_argument("--model_name_or_path", required=True)
    ap.add_argument("--train_data_path", required=True)
    ap.add_argument("--reward_funcs", nargs="+", default=["iou", "format"])
    ap.add_argument("--max_pixels", type=int, default=12845056)
    ap.add_argument("--min_pixels", type=int, default=3136)
    ap.add_argument("--is_curriculum_learning", type=str2bool, default=False)
    ap.add_argument("--is_early_stopping", type=str2bool, default=False)
    ap.add_argument("--finetune", action="store_true", help="fine-tune wiring: pre-decoded clips + TimeR1_Trainer_ft (reference finetune.py)")
    ap.add_argument("--video_folder", default=None)
    ap.add_argument("--preprocessed_data_path", default=None)
    # sample filtering stage of the curriculum loop (reference scripts/posttrain/train_rl_SF.sh:86-110): after training, answer every query of
    # --filter_split greedily with the trained engine, score difficulty = 100 x tIoU and write the next epoch's training set
    ap.add_argument("--filter_split", default=None, help="annotation json to score and filter after training (usually --train_data_path)")
    ap.add_argument("--filter_output_dir", default=None)
    ap.add_argument("--filter_task", default="0070_all", choices=["0070_all", "gaussian_03", "random_sample"])
    ap.add_argument("--filter_k", type=int, default=2500)
    ap.add_argument("--filter_max_new_tokens", type=int, default=1024)
    ns, _unknown = ap.parse_known_args()      # unknown reference flags (--fp16 ...) are ignored on purpose
    init_from_env("cuda")
    set_global_seed(42)
    args = GRPOConfig(**{k: getattr(ns, k) for k in cfg_fields})
    if ns.finetune:
        dataset = load_json_dataset(ns.train_data_path, ns.video_folder, ns.preprocessed_data_path)
        cls = TimeR1_Trainer_ft
    else:
        dataset = load_json_dataset_tg(ns.train_data_path, ns.is_curriculum_learning)
        cls = TimeR1_Trainer
    trainer = cls(model=ns.model_name_or_path, reward_funcs=[reward_funcs_registry[f] for f in ns.reward_funcs],
                  metric_funcs=list(metric_funcs_registry.values()), args=args, train_dataset=dataset,
                  callbacks=[StopAfterNEpochsCallback()] if ns.is_early_stopping else None, max_pixels=ns.max_pixels, min_pixels=ns.min_pixels)
    # resume arithmetic of the reference (main.py:589-618): continue from checkpoint-N and EXTEND max_steps by this launch's epochs
    ckpt = args.resume_from_checkpoint
    if ckpt and os.path.isdir(ckpt):
        st = json.load(open(os.path.join(ckpt, "trainer_state.json")))
        ga = max(1, args.gradient_accumulation_steps)
        per_epoch = max(len(trainer.get_train_dataloader()) // ga, 1)
        trainer.state.max_steps = int(st["global_step"]) + math.ceil(args.num_train_epochs * per_epoch)
        trainer.train(resume_from_checkpoint=ckpt)
    else:
        trainer.train()
    trainer.save_model(args.output_dir)
    if ns.filter_split and trainer.state.is_world_process_zero:
        from time_r1_amd import filtering
        from time_r1_amd.evaluate import evaluate_grounding
        from time_r1_amd.data import _clean_sentence
        items = filtering.load_filter_split(ns.filter_split)
        rows = [{"task_type": "tg", "problem": _clean_sentence(it["sentence"]), "choices": "", "solution": (float(it["timestamp"][0]), float(it["timestamp"][1])),
                 "video_path": it["video"], "durations": it["duration"], "video_start": it["video_start"], "video_end": it["video_end"],
                 "preprocessed_path": ""} for it in items]
        _, records = evaluate_grounding(trainer, rows, max_new_tokens=ns.filter_max_new_tokens)
        shares, path = filtering.filter_epoch(items, records, ns.filter_output_dir or os.path.join(args.output_dir, "filtering"), ns.filter_task, ns.filter_k)
        print("filtering: share of samples with tIoU > 0.3 / 0.5 / 0.7 = %s; next training set: %s" % (shares, path))


if __name__ == "__main__":
    main()
