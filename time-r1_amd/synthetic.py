"""Synthetic prompts: no weights, tokenizer files or videos exist on either box, so benchmarks and parity tests
use random-init models of the exact architecture and synthetic inputs of the exact shapes."""
import numpy as np
import torch

from .config import ModelConfig


def synthetic_prompt(cfg: ModelConfig, grid_thw, n_text_before=64, n_text_after=64, seed=0, text_vocab=None):
    """[text] <|vision_start|> <|video_pad|> x T_vid <|vision_end|> [text]; returns (ids list, pixel_values [N_v, patch_dim], grid)."""
    t, h, w = grid_thw
    v = cfg.vision
    n_v = t * h * w
    t_vid = n_v // v.merge_unit
    rng = np.random.RandomState(seed)
    hi = text_vocab or min(cfg.text.vocab_size, 151643)
    special = {cfg.image_token_id, cfg.video_token_id, cfg.vision_start_token_id, cfg.vision_end_token_id, cfg.eos_token_id, cfg.pad_token_id}

    def text(n):
        out = []
        while len(out) < n:
            x = int(rng.randint(2, hi))
            if x not in special:
                out.append(x)
        return out

    ids = text(n_text_before) + [cfg.vision_start_token_id] + [cfg.video_token_id] * t_vid + [cfg.vision_end_token_id] + text(n_text_after)
    g = torch.Generator().manual_seed(seed)
    # normalised patches: uint8 uniform[0,255] -> (x/255 - mean)/std has roughly unit scale
    pix = (torch.rand(n_v, v.patch_dim, generator=g) - 0.45) / 0.27
    return ids, pix, [(t, h, w)]
