"""Pins the oracle (oracle/ref_model.py + oracle/ref_grpo.py) against golden vectors captured from the UNMODIFIED reference
compute_loss + transformers (tests/golden/gen_grpo_golden.py).  fp32 on both sides: logps / loss / grads <= 1e-5."""
import pytest
import torch

from helpers import CASES, load_case, golden_params, golden_inputs, golden_rewards, HF_GRAD_KEYS, pick_grad
from oracle.ref_ops import RefOps
from oracle import ref_model as RM
from oracle import ref_grpo as RG
from oracle.text import fake_decode


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_reference_step(case):
    fx = load_case(case)
    cfg, pol, ref = golden_params(RefOps(), fx)
    W = RM.weights_from_params(pol, requires_grad=True)
    Wr = RM.weights_from_params(ref)
    pv, grid = golden_inputs(fx)
    # decoded strings (the fixture's came from the harness' fake processor) and rewards (bit-exact path tested in test_rewards.py)
    assert [fake_decode(r.tolist(), skip=(1, 0)) for r in fx["completion_ids"]] == fx["completions"]
    rew, fns = golden_rewards(fx)
    for j, fn in enumerate(fns):
        assert abs(rew[:, j].mean().item() - fx["metrics"]["rewards/" + fn.__name__][0]) < 1e-7
    out = RG.grpo_step(W, Wr, cfg, fx["prompt_ids"], pv, grid, fx["completion_ids"], rew, fx["beta"], fx["use_grpo"], rope_mode="hf5")
    # positions after the first EOS are masked out of the loss and every metric (reference :590, :725, :737); there the reference's
    # values depend on HF's padding-mask handling, so they are compared on the unmasked positions only
    m = out["mask"].bool()
    assert torch.allclose(out["logp"][m], fx["logp"][m], atol=1e-5, rtol=1e-5)
    assert torch.allclose(out["entropy"][m], fx["entropy"][m], atol=1e-5, rtol=1e-5)
    if fx["beta"] != 0:
        assert torch.allclose(out["ref_logp"][m], fx["ref_logp"][m], atol=1e-5, rtol=1e-5)
    assert abs(out["loss"].item() - fx["loss"].item()) < 1e-6
    for k, v in out["metrics"].items():
        assert abs(v - fx["metrics"][k][0]) < 2e-5, (k, v, fx["metrics"][k])
    out["loss"].backward()
    for hk, g in fx["grads"].items():
        if hk not in HF_GRAD_KEYS:
            continue
        mine = pick_grad(cfg, lambda n: W[n].grad, hk)
        assert torch.allclose(mine, g, atol=1e-6 + 1e-5 * g.abs().max().item(), rtol=1e-4), hk
    for tok, g in fx["embed_grad_rows"].items():
        assert torch.allclose(W["embed"].grad[tok], g, atol=1e-6 + 1e-5 * g.abs().max().item(), rtol=1e-4), tok


def test_oracle_reproduces_reference_ft_step():
    """The step captured from the UNMODIFIED `TimeR1_Trainer_ft.compute_loss` (timer1_trainer_ft.py:536-852; finetune.py:693-713 settings:
    Qwen2.5-VL, PPO-clip, beta 0, ragged EOS, template v2, pre-decoded frames in the row, finetune.py's metric registry): every logged value."""
    from time_r1_amd import rewards as R
    fx = load_case("ft_clip_nobeta_ragged_v2")
    assert fx["trainer"] == "TimeR1_Trainer_ft" and fx["n_logp_calls"] == 1 and fx["ref_logp"] is None      # beta 0: no reference forward
    cfg, pol, _ = golden_params(RefOps(), fx)
    W = RM.weights_from_params(pol, requires_grad=True)
    pv, grid = golden_inputs(fx)
    assert [fake_decode(r.tolist(), skip=(1, 0)) for r in fx["completion_ids"]] == fx["completions"]
    rew, fns = golden_rewards(fx)
    assert [f.__name__ for f in fns] == fx["reward_func_names"]
    out = RG.grpo_step(W, None, cfg, fx["prompt_ids"], pv, grid, fx["completion_ids"], rew, 0.0, False, rope_mode="hf5")
    m = out["mask"].bool()
    assert torch.allclose(out["logp"][m], fx["logp"][m], atol=1e-5, rtol=1e-5)
    assert torch.allclose(out["entropy"][m], fx["entropy"][m], atol=1e-5, rtol=1e-5)
    assert abs(out["loss"].item() - fx["loss"].item()) < 1e-6
    G = fx["G"]
    metric_fns = [R.metric_funcs_registry[n] for n in fx["metric_func_names"]]
    kw = {k: [v] * G for k, v in fx["row"].items()}
    mets = dict(out["metrics"])
    for j, fn in enumerate(fns):
        mets["rewards/" + fn.__name__] = rew[:, j].mean().item()
    mets.update(RG.ft_extra_metrics(out["logp"], out["advantages"], out["mask"], fx["completions"], metric_fns, kw))
    assert set(mets) == set(fx["metrics"])
    for k, v in mets.items():
        assert abs(v - fx["metrics"][k][0]) < 2e-5, (k, v, fx["metrics"][k])
    out["loss"].backward()
    for hk, g in fx["grads"].items():
        if hk in HF_GRAD_KEYS:
            mine = pick_grad(cfg, lambda n: W[n].grad, hk)
            assert torch.allclose(mine, g, atol=1e-6 + 1e-5 * g.abs().max().item(), rtol=1e-4), hk


@pytest.mark.parametrize("case", ["grpo_beta", "q25_grpo_beta", "q25_clip_beta_ragged"])
def test_oracle_model_matches_transformers_logits(case):
    """Independent of the reference: oracle forward vs transformers' Qwen2VL / Qwen2_5_VL ForConditionalGeneration on the same weights."""
    pytest.importorskip("transformers")
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from gen_grpo_golden import hf_tiny
    fx = load_case(case)
    cfg, pol, _ = golden_params(RefOps(), fx)
    hf = hf_tiny(cfg).eval()
    hf.load_state_dict({k: v.float() for k, v in pol.export_hf_state_dict().items()}, strict=True)
    pv, grid = golden_inputs(fx)
    G = fx["G"]
    ids = torch.cat([torch.tensor(fx["prompt_ids"])[None].repeat(G, 1), fx["completion_ids"].long()], 1)
    with torch.no_grad():
        ref_logits = hf(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values_videos=pv.repeat(G, 1),
                        video_grid_thw=torch.tensor(grid).repeat_interleave(G, 0), mm_token_type_ids=(ids == cfg.video_token_id).int() * 2).logits
        W = RM.weights_from_params(pol)
        mine = RM.llm_logits(W, cfg, ids, RM.vision_tower(W, cfg, pv, grid), grid, "hf5")
    assert torch.allclose(mine, ref_logits, atol=2e-5, rtol=1e-4)
