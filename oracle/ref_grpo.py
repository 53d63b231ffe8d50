"""ORACLE - TEST INFRASTRUCTURE ONLY.  The GRPO micro-step of the reference's `TimeR1_Trainer.compute_loss`
(/root/reference/src/time_r1/rl/timer1_trainer.py:512-782) restated as plain tensor algebra on top of oracle/ref_model.py
(line numbers below refer to the reference file).  Autograd provides the backward.

Pinned by tests/test_oracle_vs_golden.py against outputs of the reference itself (tests/golden/grpo_step_*.pt).
"""
import torch

from . import ref_model as RM


def completion_mask(completion_ids, eos_token_id):
    """:580-590  mask[g,t] = t <= first EOS (EOS kept); all ones if none."""
    is_eos = completion_ids == eos_token_id
    eos_idx = torch.full((is_eos.size(0),), is_eos.size(1), dtype=torch.long)
    eos_idx[is_eos.any(1)] = is_eos.int().argmax(1)[is_eos.any(1)]
    return (torch.arange(is_eos.size(1))[None, :] <= eos_idx[:, None]).int()


def grpo_step(W, W_ref, cfg, prompt_ids, pixel_values, grid_thw, completion_ids, rewards_per_func, beta, use_grpo, rope_mode="hf5",
              eps_low=0.2, eps_high=0.2):
    """One prompt, G completions. W / W_ref: weight dicts (W requires grad). rewards_per_func: fp32 [G, n_funcs] (host callbacks' output).
    Returns dict(loss, logp, entropy, ref_logp, advantages, metrics)."""
    G, C = completion_ids.shape
    P = len(prompt_ids)
    ids = torch.cat([torch.tensor(prompt_ids)[None].repeat(G, 1), completion_ids.long()], 1)            # :575-578
    mask = completion_mask(completion_ids, cfg.eos_token_id)                                             # :580-590

    def logps(Wx):
        vid = RM.vision_tower(Wx, cfg, pixel_values, grid_thw)                                           # (reference: G-fold replicated, :594-599)
        logits = RM.llm_logits(Wx, cfg, ids, vid, grid_thw, rope_mode)
        lp, ent = RM.per_token_logps(logits, ids)                                                        # :452-481
        return lp[:, P - 1:], ent[:, P - 1:]                                                             # :609-612
    logp, ent = logps(W)
    ref_logp = None
    kl = None
    if beta != 0.0:
        with torch.no_grad():
            ref_logp, _ = logps(W_ref)                                                                   # :613-632
        kl = torch.exp(ref_logp - logp) - (ref_logp - logp) - 1                                          # :635-639
    rewards = rewards_per_func.sum(1)                                                                    # :701
    mean = rewards.view(-1, G).mean(1).repeat_interleave(G, 0)
    std = rewards.view(-1, G).std(1).repeat_interleave(G, 0)                                             # unbiased, :704
    adv = (rewards - mean) / (std + 1e-4)                                                                # :712
    ratio = torch.exp(logp - logp.detach())
    if use_grpo:                                                                                         # :713-727
        l = ratio * adv[:, None]
        l = -(l - beta * kl) if beta != 0.0 else -l
        loss = ((l * mask).sum(1) / mask.sum(1)).mean()
    else:                                                                                                # :729-737
        l = -torch.min(ratio * adv[:, None], torch.clamp(ratio, 1 - eps_low, 1 + eps_high) * adv[:, None])
        if beta != 0.0:
            l = l + beta * kl
        loss = (l * mask).sum() / mask.sum()
    metrics = {"completion_length": mask.sum(1).float().mean().item(), "reward": rewards.mean().item(), "reward_std": std.mean().item()}   # :739-761
    if beta != 0.0:
        metrics["kl"] = ((kl * mask).sum(1) / mask.sum(1)).mean().item()                                 # :762-768
    metrics["generation_entropy"] = ((ent * mask).sum(1) / mask.sum(1).clamp(min=1)).mean().item()       # :769-777
    return dict(loss=loss, logp=logp, entropy=ent, ref_logp=ref_logp, advantages=adv, mask=mask, metrics=metrics, rewards=rewards)


def ft_extra_metrics(logp, advantages, mask, completions, metric_funcs, metric_kwargs, eps_low=0.2, eps_high=0.2):
    """What `TimeR1_Trainer_ft.compute_loss` logs on top of the base trainer (reference timer1_trainer_ft.py, single process so every
    `gather_for_metrics` is the identity): `metrics/<fn>` = mean over the G completions of each metric callback (:670-691, :789-794) and
    the clip ratios of the PPO-clip branch (:820-842; `coef_1 = exp(logp - logp.detach())` is identically 1, so they are 0 by construction)."""
    out = {}
    for fn in metric_funcs:
        vals = torch.tensor(fn(prompts=None, completions=completions, **metric_kwargs), dtype=torch.float32)       # :685-691
        out["metrics/" + fn.__name__] = vals.mean().item()                                                          # :789-794
    coef_1 = torch.exp(logp - logp.detach())                                                                        # :756
    adv = advantages[:, None]
    low = (coef_1 < 1 - eps_low) & (adv < 0)                                                                        # :821
    high = (coef_1 > 1 + eps_high) & (adv > 0)                                                                      # :822-824
    tot = mask.sum()
    low_r, high_r, reg_r = (low * mask).sum() / tot, (high * mask).sum() / tot, ((low | high) * mask).sum() / tot   # :827-829
    out.update({"clip_ratio/low_mean": low_r.item(), "clip_ratio/low_min": low_r.item(), "clip_ratio/high_mean": high_r.item(),     # :831-842
                "clip_ratio/high_max": high_r.item(), "clip_ratio/region_mean": reg_r.item()})
    return out
