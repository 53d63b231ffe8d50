"""Frame sampling / sizing: bit-exact (integers) against the table captured from the reference; patchify against the HF video processor's
documented layout and SURVEY appendix D shapes."""
import json
import os

import pytest
import torch

import time_r1_amd  # noqa: F401
from time_r1_amd import vision_process as VP

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sizing_kat.json")))


def test_smart_resize():
    for r in KAT["smart_resize"]:
        try:
            got = list(VP.smart_resize(r["h"], r["w"], min_pixels=r["min_pixels"], max_pixels=r["max_pixels"]))
        except ValueError:
            got = "ValueError"
        assert got == r["out"], r


def test_smart_nframes():
    for r in KAT["smart_nframes"]:
        try:
            got = VP.smart_nframes(dict(r["ele"]), r["total"], r["fps"])
        except (ValueError, AssertionError) as e:
            got = type(e).__name__
        assert got == r["out"], r


def test_video_size_and_plan():
    for r in KAT["video"]:
        if r["kind"] == "size":
            assert list(VP.video_target_size(dict(r["ele"]), r["nframes"], r["h"], r["w"])) == r["out"], r
        else:
            try:
                idx, sf = VP.frame_plan(dict(r["ele"]), r["total"], r["fps"])
                got = {"idx": idx, "sample_fps": repr(float(sf))}
            except ValueError:
                got = "ValueError"
            assert got == r["out"], r


@pytest.mark.parametrize("nframes,grid,tokens", [(8, (4, 26, 46), 1196), (16, (8, 26, 46), 2392), (32, (16, 22, 38), 3344), (64, (32, 14, 28), 3136)])
def test_baseline_config_shapes(nframes, grid, tokens):
    """360x640 source under the trainer's budget (total_pixels 3584*784, min_pixels 16*784)."""
    ele = {"total_pixels": 3584 * 28 * 28, "min_pixels": 16 * 28 * 28}
    h, w = VP.video_target_size(ele, nframes, 360, 640)
    assert (nframes // 2, h // 14, w // 14) == grid
    assert grid[0] * grid[1] * grid[2] // 4 == tokens


def test_patchify_layout():
    T, H, W = 4, 56, 84
    frames = torch.arange(T * 3 * H * W, dtype=torch.float32).reshape(T, 3, H, W) % 251
    pv, grid = VP.patchify(frames)
    assert grid == (2, 4, 6) and pv.shape == (48, 1176)
    # element check against the definition: patch index enumerates (t, h/2, w/2, 2, 2); features are (C, 2, 14, 14)
    x = (frames / 255.0 - torch.tensor(VP.CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(VP.CLIP_STD).view(1, 3, 1, 1)
    n = 0
    for t in range(2):
        for bh in range(2):
            for bw in range(3):
                for ih in range(2):
                    for iw in range(2):
                        hh, ww = (bh * 2 + ih) * 14, (bw * 2 + iw) * 14
                        ref = x[2 * t:2 * t + 2, :, hh:hh + 14, ww:ww + 14].permute(1, 0, 2, 3).reshape(-1)
                        assert torch.allclose(pv[n], ref, atol=1e-6, rtol=0)
                        n += 1


def _hf_patchify_cases():
    fx = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "patchify_hf.pt"), weights_only=False)
    for c in fx["cases"]:
        frames = torch.randint(0, 256, tuple(c["shape"]), generator=torch.Generator().manual_seed(c["seed"]), dtype=torch.uint8)
        yield frames, c["pixel_values_videos"], tuple(c["video_grid_thw"])


def test_patchify_matches_hf_video_processor_golden():
    """VP.patchify == transformers' Qwen2VLVideoProcessor (fixture captured by tests/golden/gen_patchify_golden.py from the unmodified HF
    class: layout, temporal padding of odd frame counts, fused rescale/normalise constants).  Reference call: timer1_trainer.py:547-556."""
    n = 0
    for frames, want, grid in _hf_patchify_cases():
        pv, g = VP.patchify(frames.float())
        assert g == grid and pv.shape == want.shape
        assert torch.allclose(pv, want, atol=2e-6, rtol=0), float((pv - want).abs().max())
        n += 1
    assert n == 4


def test_oracle_video_preprocess_matches_hf_golden(ref_ops):
    """The oracle of the fused GPU preprocessing kernel (uint8 frames, no resize needed: target == source size) reproduces the HF patches."""
    for frames, want, grid in _hf_patchify_cases():
        T, _, H, W = frames.shape
        out, g = ref_ops.video_preprocess(frames, (H, W), 1216)
        assert tuple(g) == grid
        assert torch.allclose(out[:, :1176].float(), want, atol=2e-6), float((out[:, :1176].float() - want).abs().max())
        assert float(out[:, 1176:].abs().max()) == 0.0


def test_process_vision_info_predecoded():
    frames = torch.randint(0, 256, (8, 3, 360, 640), dtype=torch.uint8)
    conv = [{"role": "user", "content": [{"type": "video", "video": frames, "total_pixels": 3584 * 784, "min_pixels": 16 * 784}]}]
    _, vids, kw = VP.process_vision_info_v3([conv], return_video_kwargs=True)
    assert vids[0].shape == (8, 3, 364, 644) and vids[0].dtype == torch.float32 and kw == {"fps": [2.0]}
    assert float(vids[0].min()) >= 0 and float(vids[0].max()) <= 255
    with pytest.raises(RuntimeError):
        VP.read_video({"video": "/nonexistent.mp4"})
