"""Host logic fixed in round 2: HF resume arithmetic, sequential (non-batched) rollouts inside an accumulation window,
equal-length rank shards, processor saved with the model, logged loss / epoch.  CPU, oracle ops injected through `ops=`."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import load_case
from test_trainer_host_logic import make_trainer, _dataset


def _run(fx, tmp, epochs, max_steps=-1, save_steps=1, ckpt=None, state_max=None, n_rows=4, events=None):
    cfg, tr = make_trainer(fx, ga=2)
    tr.args.output_dir = str(tmp)
    tr.args.num_train_epochs = epochs
    tr.args.max_steps = max_steps
    tr.args.save_strategy = "steps" if save_steps else "no"
    tr.args.save_steps = save_steps or 0
    tr.args.learning_rate = 1e-4
    tr.train_dataset = _dataset(fx, n_rows)
    if events is not None:
        class CB:
            def on_epoch_end(self, args, state, control, **kw):
                events.append(("epoch", state.epoch, state.global_step))
        tr.callbacks = [CB()]
    if state_max:
        tr.state.max_steps = state_max
    tr.train(resume_from_checkpoint=ckpt)
    return tr


def test_resume_from_epoch_boundary_with_one_epoch_trains_it(tmp_path):
    """main.py:600-618: max_steps = global_step + epochs * steps_per_epoch, --num_train_epochs 1.  Resuming from the checkpoint written at the
    end of epoch 1 must train one MORE epoch (it used to replay an empty epoch and stop at once), and equal the uninterrupted 2-epoch run."""
    fx = load_case("grpo_beta")
    straight = _run(fx, tmp_path / "a", epochs=2, save_steps=2)
    assert straight.state.global_step == 4
    ev = []
    resumed = _run(fx, tmp_path / "b", epochs=1, save_steps=0, ckpt=str(tmp_path / "a" / "checkpoint-2"), state_max=2 + 1 * 2, events=ev)
    assert resumed.state.global_step == 4
    assert ev == [("epoch", 2.0, 4)], ev                      # the replayed epoch fires no callback / save
    # same schedule horizon (max_steps 4) -> same weights as the straight run
    assert torch.allclose(resumed.params.train.master, straight.params.train.master, atol=1e-6)


def test_resume_mid_epoch_skips_consumed_batches_and_runs_to_max_steps(tmp_path):
    fx = load_case("grpo_beta")
    straight = _run(fx, tmp_path / "a", epochs=1, save_steps=1, n_rows=8)          # 4 steps per epoch
    assert straight.state.global_step == 4
    ev = []
    resumed = _run(fx, tmp_path / "b", epochs=1, save_steps=0, ckpt=str(tmp_path / "a" / "checkpoint-1"), n_rows=8, events=ev)
    assert resumed.state.global_step == 4 and ev == [("epoch", 1.0, 4)]
    assert torch.allclose(resumed.params.train.master, straight.params.train.master, atol=1e-6)
    assert abs(resumed.state.log_history[-1]["epoch"] - 1.0) < 1e-9
    # logged loss = mean of the micro-step losses of the step (HF), not divided by GA once more
    assert "loss" in resumed.state.log_history[-1]


def test_sequential_rollouts_in_a_window_equal_one_by_one_compute_loss():
    """rollout_batching=False with GA=2: each prompt's rollout is followed by ITS update, so the saved prefill of prompt 1 is not overwritten by
    prompt 2's prefill (slot 0 is shared).  Gradients must equal two independent compute_loss calls."""
    fx = load_case("grpo_beta")
    grads = []
    for mode in ("window", "single"):
        cfg, tr = make_trainer(fx, ga=2, rollout_batching=False)
        tr.train_dataset = _dataset(fx, 2)
        batches = [[tr.train_dataset[0]], [tr.train_dataset[1]]]
        if mode == "window":
            tr.accumulation_window(batches)
        else:
            for b in batches:
                tr.compute_loss(tr.model, b)
        grads.append(tr.params.train.grad.clone())
    assert float(grads[1].abs().max()) > 0
    assert torch.allclose(grads[0], grads[1], atol=1e-7), float((grads[0] - grads[1]).abs().max())


def test_batched_and_sequential_windows_agree():
    fx = load_case("grpo_beta")
    grads = []
    for batching in (True, False):
        cfg, tr = make_trainer(fx, ga=2, rollout_batching=batching)
        tr.train_dataset = _dataset(fx, 2)
        tr.accumulation_window([[tr.train_dataset[0]], [tr.train_dataset[1]]])
        grads.append(tr.params.train.grad.clone())
    assert torch.allclose(grads[0], grads[1], atol=1e-6), float((grads[0] - grads[1]).abs().max())


@pytest.mark.parametrize("n,world", [(5, 2), (2500, 8), (7, 4), (8, 8)])
def test_rank_shards_have_equal_length_and_cover_the_dataset(n, world):
    fx = load_case("grpo_beta")
    cfg, tr = make_trainer(fx)
    tr.train_dataset = list(range(n))
    lens, seen = [], []
    for r in range(world):
        tr.dp.rank, tr.dp.world = r, world
        ld = tr.get_train_dataloader()
        items = [b[0] for b in ld]
        assert len(items) == len(ld)
        lens.append(len(items))
        seen += items
    assert len(set(lens)) == 1 and lens[0] == -(-n // world)
    assert set(seen) == set(range(n))                          # wrap-around padding only repeats rows, never drops one


def test_save_model_writes_processor_and_generation_config(tmp_path):
    fx = load_case("grpo_beta")
    cfg, tr = make_trainer(fx)
    calls = []
    tr.processing_class.save_pretrained = lambda d: calls.append(d)
    tr.save_model(str(tmp_path / "m"))
    assert calls == [str(tmp_path / "m")]
    gc = json.load(open(tmp_path / "m" / "generation_config.json"))
    assert gc["eos_token_id"] == cfg.eos_token_id


def test_deepspeed_zero_json_selects_the_sharded_optimizer():
    from time_r1_amd.trainer import TimeR1_Trainer, GRPOConfig
    assert TimeR1_Trainer._wants_shard(GRPOConfig(deepspeed="scripts/zero3.json"))
    assert TimeR1_Trainer._wants_shard(GRPOConfig(deepspeed="scripts/zero3_offload.json"))
    assert not TimeR1_Trainer._wants_shard(GRPOConfig())
    assert not TimeR1_Trainer._wants_shard(GRPOConfig(deepspeed="scripts/zero3.json", shard_optimizer=False))
    # world sizes whose chunks are not 128-byte aligned (3, 5, 6, 7) fall back to the replicated optimizer instead of asserting
    import types
    for world, want in ((2, True), (3, False), (4, True), (6, False), (8, True)):
        dp = types.SimpleNamespace(enabled=True, world=world, rank=1)
        assert TimeR1_Trainer._wants_shard(GRPOConfig(deepspeed="scripts/zero3.json"), dp) == want


def test_row_chunked_lm_head_equals_whole_head(monkeypatch):
    """Config 4 (G*C = 16384 prediction rows) never holds [G*C, V] logits: the head runs in row chunks and the backward recomputes each
    chunk's logits.  Same loss / metrics / gradients as the single-piece head."""
    from helpers import frames_for
    from time_r1_amd.model import Engine
    fx = load_case("clip_beta")
    out = []
    for ch in (4096, 7):
        monkeypatch.setattr(Engine, "HEAD_CHUNK_ROWS", ch)
        cfg, tr = make_trainer(fx)
        row = dict(fx["row"])
        row["_forced_completion_ids"] = fx["completion_ids"].numpy()
        tr._video_inputs = lambda ex: ([frames_for(fx)], [2.0])
        loss = tr.compute_loss(tr.model, [row])
        out.append((float(loss), dict(tr._metrics), tr.params.train.grad.clone()))
    assert abs(out[0][0] - out[1][0]) < 1e-7 and out[0][1].keys() == out[1][1].keys()
    assert torch.allclose(out[0][2], out[1][2], atol=1e-6), float((out[0][2] - out[1][2]).abs().max())


def test_stashed_prefill_window_equals_full_slot_window(monkeypatch):
    """Config 4 memory scheme: later prompts of a batched window keep only their prompt rows (stash) and move them into the ONE full
    saved-activation set when their update starts.  Same gradients as one full set per prompt."""
    from time_r1_amd.model import Engine
    fx = load_case("grpo_beta")
    grads = []
    for gb in (40.0, 0.0):           # 0 GB threshold: every sequence counts as large -> slot 1 is stashed
        monkeypatch.setattr(Engine, "CTX_STASH_GB", gb)
        cfg, tr = make_trainer(fx, ga=2, rollout_batching=True)
        tr.train_dataset = _dataset(fx, 2)
        # the second prompt is LONGER than the first (more video tokens): its packed sequence outgrows the full set sized for prompt 0
        tr.train_dataset.rows[1]["video_frames"] = torch.randint(0, 256, (4, 3, 84, 112), generator=torch.Generator().manual_seed(9), dtype=torch.uint8).float()
        tr.accumulation_window([[tr.train_dataset[0]], [tr.train_dataset[1]]])
        pool = tr.engine._ctx_pool
        assert (("stash", 1) in pool) == (gb == 0.0) and ((1 in pool) == (gb != 0.0))
        grads.append(tr.params.train.grad.clone())
    # (the large-sequence regime also runs every row through the last layer - Engine.tail_rows_from - so the two windows differ in fp32 summation
    # order there, not in what is summed)
    diff = float((grads[0] - grads[1]).abs().max())
    assert float(grads[0].abs().max()) > 0 and diff < 1e-6 * max(1.0, float(grads[0].abs().max()) * 1e3), diff
    monkeypatch.setattr(Engine, "TAIL_SKIP", False)      # with the row skip off in both, the stash is a pure memory scheme: bit-identical
    g2 = []
    for gb in (40.0, 0.0):
        monkeypatch.setattr(Engine, "CTX_STASH_GB", gb)
        cfg, tr = make_trainer(fx, ga=2, rollout_batching=True)
        tr.train_dataset = _dataset(fx, 2)
        tr.train_dataset.rows[1]["video_frames"] = torch.randint(0, 256, (4, 3, 84, 112), generator=torch.Generator().manual_seed(9), dtype=torch.uint8).float()
        tr.accumulation_window([[tr.train_dataset[0]], [tr.train_dataset[1]]])
        g2.append(tr.params.train.grad.clone())
    assert torch.equal(g2[0], g2[1])
