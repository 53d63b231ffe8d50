"""smoke(): one tiny GRPO micro-step on cuda:0 through the HIP path, checked against the CPU oracle (oracle/ is the checker only)."""
import numpy as np
import torch


def run_smoke(verbose=True):
    from .config import tiny_test
    from .params import ModelParams
    from .model import Engine
    from .grpo import GRPOCore, eos_mask, group_advantages
    from .ops import HipOps
    from .synthetic import synthetic_prompt
    from oracle.ref_ops import RefOps  # checker

    assert torch.cuda.is_available(), "smoke() needs a HIP device"
    cfg = tiny_test()
    G, C = 4, 8
    ids, pix, grid = synthetic_prompt(cfg, (2, 4, 6), 5, 5, seed=3, text_vocab=400)
    out = {}
    for name, ops in (("hip", HipOps("cuda:0")), ("ref", RefOps())):
        params = ModelParams(cfg, ops, seed=0)
        eng = Engine(cfg, ops, params)
        core = GRPOCore(eng, params.train.clone_weights_only(), G, C, beta=0.04, use_grpo=True, seed=11, rope_index_mode="hf5")
        st = core.prepare(ids, pix, grid)
        if name == "hip":
            toks = core.rollout(st)
            toks_host = toks.cpu()
        else:
            from .positions import PackedLayout
            st.layout = PackedLayout(st.P, G, C)
            st.completion_ids = toks_host.clone()
        core.forward_logps(st)
        mask = torch.tensor(eos_mask(toks_host.numpy(), cfg.eos_token_id))
        rew = torch.rand(G, 2, generator=torch.Generator().manual_seed(5))
        _, adv, _ = group_advantages(rew, G)
        out3, _ = core.loss_backward(st, mask.to(ops.device), adv.to(ops.device), 1.0)
        out[name] = dict(logp=st.logp.float().cpu(), ent=st.entropy.float().cpu(), grad=params.train.grad.float().cpu(), out3=out3.float().cpu())
    dl = (out["hip"]["logp"] - out["ref"]["logp"]).abs().max().item()
    de = (out["hip"]["ent"] - out["ref"]["ent"]).abs().max().item()
    gn = out["ref"]["grad"].norm().item()
    dg = (out["hip"]["grad"] - out["ref"]["grad"]).norm().item() / max(gn, 1e-12)
    if verbose:
        print("smoke: max|dlogp| %.4f  max|dent| %.4f  rel grad err %.4f  (grad norm %.4f)" % (dl, de, dg, gn))
    # bf16 activations vs fp32 oracle: logp/entropy within 0.06 (SURVEY 7 hard part 3), gradient direction within 5 %
    assert dl < 0.06 and de < 0.06, ("logp/entropy mismatch vs oracle", dl, de)
    assert dg < 0.08, ("gradient mismatch vs oracle", dg)
    return out
