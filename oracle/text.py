"""ORACLE / TEST INFRASTRUCTURE: deterministic id -> text map standing in for a tokenizer's batch_decode (no tokenizer files exist
offline - SURVEY.md 0.9). Used by the golden-capture harness and the tests so that reward callbacks run on real strings."""

PIECES = ["<think>", "</think>", "<answer>", "</answer>", " to ", " and ", "1", "2", "3", "4", "5", "6", "7", "8", "9", "0", ".", " ", "the", "person",
          "because", "\n", "step", "observe", "<timestep>", "</timestep>"]


def fake_decode(ids, skip=()):
    return "".join(PIECES[int(i) % len(PIECES)] for i in ids if int(i) not in skip)


class FakeProcessor:
    """Stand-in for the HF Qwen2-VL processor in tests (no tokenizer files offline): fixed prompt ids around the expanded
    video pads, the HF patchify layout (time-r1_amd/vision_process.py:patchify, itself checked against the HF video processor's layout)
    and the deterministic id -> text map above for batch_decode."""
    eos_token_id, pad_token_id = 1, 0

    def __init__(self, cfg, head=(5, 6, 7), tail=(8, 9, 10)):
        self.cfg, self.head, self.tail = cfg, list(head), list(tail)

    def apply_chat_template(self, conv, tokenize=False, add_generation_prompt=True):
        return "PROMPT"

    def __call__(self, text=None, images=None, videos=None, fps=None, **kw):
        import torch
        from time_r1_amd import vision_process as VP
        pv, grid = VP.patchify(videos[0])
        n_tok = grid[0] * grid[1] * grid[2] // 4
        c = self.cfg
        ids = torch.tensor([self.head + [c.vision_start_token_id] + [c.video_token_id] * n_tok + [c.vision_end_token_id] + self.tail])
        return {"input_ids": ids, "attention_mask": torch.ones_like(ids), "pixel_values_videos": pv, "video_grid_thw": torch.tensor([list(grid)])}

    def prompt_ids(self, text, n_video_tokens):
        c = self.cfg
        return self.head + [c.vision_start_token_id] + [c.video_token_id] * n_video_tokens + [c.vision_end_token_id] + self.tail

    def batch_decode(self, ids, skip_special_tokens=True):
        return [fake_decode(r.tolist(), skip=(self.eos_token_id, self.pad_token_id) if skip_special_tokens else ()) for r in ids]
