"""Round 3 debugging aid: policy vs reference log-probs of one synthetic prompt through the engine (found the fused lm_head epilogue bug at V = 151936).
   python tools/debug_ref_logp.py [qwen2-vl-2b|qwen2-vl-7b]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from time_r1_amd.ops import HipOps
args = bench.parse_args(["--model", sys.argv[1] if len(sys.argv) > 1 else "qwen2-vl-2b", "--frames", "16", "--no-cpu-baseline"])
ops = HipOps("cuda:0"); ops.use_priority_stream()
wl = bench.Workload(args, ops, "cuda:0", 0)
a, core, v, tr = wl.args, wl.core, wl.cfg.vision, wl.trainer
row = wl.dataset[0]
frames = row["video_frames"].to(ops.device)
pix, g = ops.video_preprocess(frames, wl.target, v.patch_dim_padded, v.patch_size, v.temporal_patch_size, v.spatial_merge_size)
n_tok = g[0] * g[1] * g[2] // v.merge_unit
ids = tr.processing_class.prompt_ids(tr.processing_class.apply_chat_template(tr.make_conversation_video(row)), n_tok)
for reuse in (True, False):
    core.reuse_prefill = reuse
    st = core.prepare(ids, pix, np.asarray([g]))
    core.rollout(st)
    core.forward_logps(st)
    torch.cuda.synchronize()
    lp, rp = st.logp.float(), st.ref_logp.float()
    d = (rp - lp)
    print("reuse_prefill", reuse, "logp finite", bool(torch.isfinite(lp).all()), "ref finite", bool(torch.isfinite(rp).all()),
          "logp range", lp.min().item(), lp.max().item(), "ref range", rp.min().item(), rp.max().item(), "max |ref - logp|", d.abs().max().item(),
          "mean", d.abs().mean().item(), "ent", st.entropy.float().mean().item())
    i = d.abs().argmax().item(); print("   worst at (g, c) =", divmod(i, lp.shape[1]), lp.flatten()[i].item(), rp.flatten()[i].item())
    print("   per-position mean |d| first 8:", d.abs().mean(0)[:8].tolist(), " last 4:", d.abs().mean(0)[-4:].tolist())
