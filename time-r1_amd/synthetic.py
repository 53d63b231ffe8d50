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


_PIECES = ["<think>", "</think>", "<answer>", "</answer>", " to ", " and ", "1", "2", "3", "4", "5", "6", "7", "8", "9", "0", ".", " ", "the", "person",
           "because", "\n", "step", "observe", "<timestep>", "</timestep>"]


def piece_decode(ids_row, skip=()):
    """No tokenizer files exist offline: a fixed id -> text piece map, so that the real reward callbacks run on real strings."""
    return "".join(_PIECES[int(i) % len(_PIECES)] for i in ids_row if int(i) not in skip)


class SyntheticProcessor:
    """What `TimeR1_Trainer` needs from a Qwen2-VL processor (reference timer1_trainer.py:536-556, :695) when no tokenizer files exist:
    `apply_chat_template` (the prompt string), `prompt_ids` (token ids with the video placeholder expanded - random text ids seeded by the
    prompt string, the real special tokens around the video pads), `batch_decode` (fixed id -> piece map).  Used by bench.py to drive the
    trainer class itself on synthetic rows; the model-side work is unchanged by what the ids are."""

    def __init__(self, cfg, n_text_before=64, n_text_after=64):
        self.cfg, self.nb, self.na = cfg, int(n_text_before), int(n_text_after)
        self.eos_token_id, self.pad_token_id = cfg.eos_token_id, cfg.pad_token_id

    def apply_chat_template(self, conv, tokenize=False, add_generation_prompt=True):
        text = "".join(c.get("text", "") for m in conv for c in m["content"] if c.get("type") == "text")
        return "<|im_start|>user\n<|vision_start|><|video_pad|><|vision_end|>%s<|im_end|>\n<|im_start|>assistant\n" % text

    def prompt_ids(self, text, n_video_tokens):
        import zlib
        c = self.cfg
        rng = np.random.RandomState(zlib.crc32(text.encode()) & 0x7FFFFFFF)
        special = {c.image_token_id, c.video_token_id, c.vision_start_token_id, c.vision_end_token_id, c.eos_token_id, c.pad_token_id}
        hi = min(c.text.vocab_size, 151643)

        def words(n):
            out = []
            while len(out) < n:
                x = int(rng.randint(2, hi))
                if x not in special:
                    out.append(x)
            return out
        return words(self.nb) + [c.vision_start_token_id] + [c.video_token_id] * n_video_tokens + [c.vision_end_token_id] + words(self.na)

    def batch_decode(self, ids, skip_special_tokens=True):
        skip = (self.eos_token_id, self.pad_token_id) if skip_special_tokens else ()
        return [piece_decode(r.tolist(), skip) for r in ids]


class SyntheticClips:
    """Dataset rows in the reference's schema (main.py:349-373: problem / solution / video_path / durations) whose `video_frames` are decoded
    uint8 frames [T, 3, H, W] (what the reference's video reader returns, src/utils/vision_process.py:467-472), drawn once per row and kept
    on `device` (HBM-resident input) or in pinned host memory (`pin=True`: the PCIe-inclusive variant)."""

    def __init__(self, n, n_frames, src_hw, device="cpu", pin=False, seed=7):
        self.rows = []
        for i in range(n):
            g = torch.Generator().manual_seed(seed + i)
            fr = torch.randint(0, 256, (n_frames, 3) + tuple(src_hw), generator=g, dtype=torch.uint8)
            fr = fr.pin_memory() if pin else fr.to(device)
            self.rows.append(dict(problem="synthetic query %d" % i, solution=(2.0, 12.0), durations=30.0, video_path="synthetic://clip%d" % i, video_frames=fr))

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        return self.rows[i]
