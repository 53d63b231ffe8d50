"""Capture `pixel_values_videos` / `video_grid_thw` from transformers' Qwen2VLVideoProcessor (the processor call of the reference,
src/time_r1/rl/timer1_trainer.py:547-556 -> transformers/models/qwen2_vl/video_processing_qwen2_vl.py:236-274 in v5.15.0) for a few
frame shapes, so that the repo's own patchify (time-r1_amd/vision_process.py) and the fused HIP preprocessing kernel are pinned to the
HF layout and constants instead of to their own definition.

torchvision is absent offline, and the HF module imports it at the top.  As in _ref_harness.py a stub module stands in; the processor
path used here (do_resize=False, do_convert_rgb=False) touches exactly one torchvision function, `normalize(x, mean, std)`, which is
(x - mean[:, None, None]) / std[:, None, None] on float tensors - stated below; everything pinned by the fixture (fused rescale +
normalise constants, temporal padding, the 9-D view and its permutation, the grid) is transformers' own code running unmodified.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_patchify_golden.py      ->  tests/golden/patchify_hf.pt
"""
import importlib.machinery
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = [(4, 3, 56, 84), (6, 3, 84, 56), (3, 3, 56, 56), (4, 3, 84, 112)]      # (3, ...): odd frame count -> HF repeats the last frame


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


def install_torchvision_stub():
    import transformers  # noqa: F401  (before the stubs, like _ref_harness.install_stubs)

    class InterpolationMode:
        BICUBIC = "bicubic"

    def normalize(x, mean, std):
        mean = torch.as_tensor(mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        std = torch.as_tensor(std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return (x - mean) / std
    _stub("torchvision")
    _stub("torchvision.transforms", InterpolationMode=InterpolationMode)
    _stub("torchvision.transforms.v2")
    f = _stub("torchvision.transforms.v2.functional", InterpolationMode=InterpolationMode, normalize=normalize)
    import transformers.image_processing_backends as ipb
    ipb.tvF = f          # that module imports torchvision only when it is "available"


def main():
    install_torchvision_stub()
    from transformers.models.qwen2_vl.video_processing_qwen2_vl import Qwen2VLVideoProcessor
    import transformers
    proc = Qwen2VLVideoProcessor(do_resize=False, do_sample_frames=False, do_convert_rgb=False)
    cases = []
    for i, shape in enumerate(SHAPES):
        frames = torch.randint(0, 256, shape, generator=torch.Generator().manual_seed(900 + i), dtype=torch.uint8)
        out = proc(videos=[frames], return_tensors="pt")
        cases.append(dict(shape=shape, seed=900 + i, pixel_values_videos=out["pixel_values_videos"].float().clone(),
                          video_grid_thw=out["video_grid_thw"][0].tolist()))
        print(shape, tuple(out["pixel_values_videos"].shape), cases[-1]["video_grid_thw"])
    torch.save(dict(transformers=transformers.__version__, cases=cases), os.path.join(HERE, "patchify_hf.pt"))


if __name__ == "__main__":
    main()
