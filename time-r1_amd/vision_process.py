"""Frame sampling / sizing / patchify for the video branch - the integer logic of the reference's vision preprocessing.

Restates (own code, pinned by tests/golden/sizing_kat.json captured from the reference):
  smart_resize / smart_nframes        reference src/utils/vision_process.py:60-90, :154-199
  frame indices + sample fps           :285-334  (_read_video_decord_w_timestamp)
  per-frame pixel budget               :440-466  (fetch_video_v3)
  patchify                             transformers/models/qwen2_vl/video_processing_qwen2_vl.py:236-274 (rescale, normalise, 2x14x14 patches
                                       in merge-block order)
Video DECODING is host I/O outside the hot path: it needs decord or torchvision, which are absent offline, so
`read_video` raises unless one of them is importable; the trainer also accepts pre-decoded frame tensors (fine-tune format,
reference finetune.py:594-623).
"""
import math
import os

import numpy as np
import torch

IMAGE_FACTOR = 28
MIN_PIXELS = 4 * 28 * 28
MAX_PIXELS = 16384 * 28 * 28
MAX_RATIO = 200
VIDEO_MIN_PIXELS = 128 * 28 * 28
VIDEO_MAX_PIXELS = 768 * 28 * 28
FRAME_FACTOR = 2
FPS = 2.0
FPS_MIN_FRAMES = 4
FPS_MAX_FRAMES = 768
VIDEO_TOTAL_PIXELS = int(float(os.environ.get("VIDEO_MAX_PIXELS", 128000 * 28 * 28 * 0.9)))

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def round_by_factor(x, f):
    return round(x / f) * f


def ceil_by_factor(x, f):
    return math.ceil(x / f) * f


def floor_by_factor(x, f):
    return math.floor(x / f) * f


def smart_resize(height, width, factor=IMAGE_FACTOR, min_pixels=MIN_PIXELS, max_pixels=MAX_PIXELS):
    """(h', w') divisible by `factor`, pixel count within [min_pixels, max_pixels], aspect ratio kept as closely as possible."""
    ratio = max(height, width) / min(height, width)
    if ratio > MAX_RATIO:
        raise ValueError("absolute aspect ratio must be smaller than %d, got %s" % (MAX_RATIO, ratio))
    h = max(factor, round_by_factor(height, factor))
    w = max(factor, round_by_factor(width, factor))
    if h * w > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h, w = floor_by_factor(height / beta, factor), floor_by_factor(width / beta, factor)
    elif h * w < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h, w = ceil_by_factor(height * beta, factor), ceil_by_factor(width * beta, factor)
    return h, w


def smart_nframes(ele, total_frames, video_fps):
    if "fps" in ele and "nframes" in ele:
        raise AssertionError("Only accept either `fps` or `nframes`")
    if "nframes" in ele:
        n = round_by_factor(ele["nframes"], FRAME_FACTOR)
    else:
        fps = ele.get("fps", FPS)
        lo = ceil_by_factor(ele.get("min_frames", FPS_MIN_FRAMES), FRAME_FACTOR)
        hi = floor_by_factor(ele.get("max_frames", min(FPS_MAX_FRAMES, total_frames)), FRAME_FACTOR)
        n = total_frames / video_fps * fps
        n = min(min(max(n, lo), hi), total_frames)
        n = floor_by_factor(n, FRAME_FACTOR)
    if not (FRAME_FACTOR <= n <= total_frames):
        raise ValueError("nframes should in interval [%d, %d], but got %s." % (FRAME_FACTOR, total_frames, n))
    return n


def frame_plan(ele, total_frames, video_fps):
    """-> (frame indices list, sample_fps). Timestamp-aware sampling of reference vision_process.py:285-334."""
    video_start = ele.get("video_start", 0.0)
    video_end = ele.get("video_end", total_frames / video_fps)
    start = max(0, int(video_start * video_fps))
    end = min(total_frames, int(video_end * video_fps))
    if end == start:
        end = start + 1
    if end < start or end > total_frames:
        raise ValueError("Video timestamps are error!")
    effective = end - start
    n = smart_nframes(ele, total_frames=effective, video_fps=video_fps)
    idx = torch.linspace(start, end - 1, n).round().long().tolist()
    return idx, n / max(effective, 1e-6) * video_fps


def video_max_pixels(ele, nframes):
    min_pixels = ele.get("min_pixels", VIDEO_MIN_PIXELS)
    total_pixels = ele.get("total_pixels", VIDEO_TOTAL_PIXELS)
    mp = max(min(VIDEO_MAX_PIXELS, total_pixels / nframes * FRAME_FACTOR), int(min_pixels * 1.05))
    return min_pixels, min(ele.get("max_pixels", mp), mp)


def video_target_size(ele, nframes, height, width, image_factor=IMAGE_FACTOR):
    if "resized_height" in ele and "resized_width" in ele:
        return smart_resize(ele["resized_height"], ele["resized_width"], factor=image_factor)
    min_pixels, max_pixels = video_max_pixels(ele, nframes)
    return smart_resize(height, width, factor=image_factor, min_pixels=min_pixels, max_pixels=max_pixels)


def resize_frames(video_u8, size):
    """uint8 [T,3,H,W] -> float32 [T,3,H',W'] : bicubic, antialias, rounded and clamped to uint8 levels like
    torchvision.transforms.functional.resize on a uint8 tensor followed by .float() (reference :467-472)."""
    x = torch.nn.functional.interpolate(video_u8.float(), size=list(size), mode="bicubic", antialias=True, align_corners=False)
    return x.round().clamp(0, 255)


def read_video(ele):
    """Decode + sample frames. Needs decord (preferred, as in the reference) or torchvision; neither exists offline."""
    try:
        import decord
    except ImportError as e:
        raise RuntimeError("video decoding needs `decord` (host-side I/O, outside the HIP hot path); pass pre-decoded frames "
                           "(`video` = uint8/float tensor [T,3,H,W]) or install decord") from e
    vr = decord.VideoReader(ele["video"])
    total, fps = len(vr), vr.get_avg_fps()
    idx, sample_fps = frame_plan(ele, total, fps)
    video = torch.tensor(vr.get_batch(idx).asnumpy()).permute(0, 3, 1, 2)
    return video, sample_fps


def fetch_video_v3(ele, image_factor=IMAGE_FACTOR, return_video_sample_fps=False):
    v = ele["video"]
    if isinstance(v, str):
        video, sample_fps = read_video(ele)
    elif torch.is_tensor(v):
        video, sample_fps = v, ele.get("fps", FPS)
    else:
        raise TypeError("video must be a path or a [T,3,H,W] tensor")
    n, _, h, w = video.shape
    th, tw = video_target_size(ele, n, h, w, image_factor)
    if video.dtype == torch.uint8:
        video = resize_frames(video, (th, tw))
    elif (h, w) != (th, tw):
        video = torch.nn.functional.interpolate(video.float(), size=[th, tw], mode="bicubic", antialias=True, align_corners=False)
    else:
        video = video.float()
    return (video, sample_fps) if return_video_sample_fps else video


def process_vision_info_v3(conversations, return_video_kwargs=False):
    """Same return shape as the reference (src/utils/vision_process.py:547-578): (image_inputs, video_inputs, {"fps": [...]})."""
    if conversations and isinstance(conversations[0], dict):
        conversations = [conversations]
    videos, fps = [], []
    for conv in conversations:
        for msg in conv:
            if isinstance(msg.get("content"), list):
                for ele in msg["content"]:
                    if ele.get("type") == "video" or "video" in ele:
                        v, f = fetch_video_v3(ele, return_video_sample_fps=True)
                        videos.append(v)
                        fps.append(f)
    videos = videos or None
    return (None, videos, {"fps": fps}) if return_video_kwargs else (None, videos)


def patchify(frames, patch=14, temporal=2, merge=2, mean=CLIP_MEAN, std=CLIP_STD, rescale=1.0 / 255.0):
    """float [T,3,H,W] in 0..255 -> (pixel_values [N_v, 3*temporal*patch*patch] fp32, (t, h, w) grid).
    T is padded to a multiple of `temporal` by repeating the last frame; H, W must be multiples of patch*merge."""
    T, C, H, W = frames.shape
    assert H % (patch * merge) == 0 and W % (patch * merge) == 0, (H, W)
    x = frames.float() * rescale
    x = (x - torch.tensor(mean).view(1, C, 1, 1)) / torch.tensor(std).view(1, C, 1, 1)
    if T % temporal:
        x = torch.cat([x, x[-1:].repeat(temporal - T % temporal, 1, 1, 1)], 0)
        T = x.shape[0]
    gt, gh, gw = T // temporal, H // patch, W // patch
    x = x.view(gt, temporal, C, gh // merge, merge, patch, gw // merge, merge, patch)
    x = x.permute(0, 3, 6, 4, 7, 2, 1, 5, 8).reshape(gt * gh * gw, C * temporal * patch * patch)
    return x.contiguous(), (gt, gh, gw)


def aa_filter(in_size, out_size):
    """Bicubic (a = -0.5) antialiased resampling taps for one axis, evaluated in float32 in the same operation order as ATen's
    _compute_indices_weights_aa (so that the rounded uint8 levels agree with torchvision's resize on all but a handful of pixels):
    -> (first input index per output [out], weights [out, taps] fp32 normalised to 1, zero-padded)."""
    f32 = np.float32
    scale = f32(in_size) / f32(out_size)
    support = f32(2.0) * scale if scale >= 1.0 else f32(2.0)
    invscale = f32(1.0) / scale if scale >= 1.0 else f32(1.0)
    taps = int(math.ceil(float(support))) * 2 + 1
    a = f32(-0.5)

    def cubic(x):
        x = abs(x)
        if x < 1.0:
            return ((a + f32(2.0)) * x - (a + f32(3.0))) * x * x + f32(1.0)
        if x < 2.0:
            return (((x - f32(5.0)) * x + f32(8.0)) * x - f32(4.0)) * a
        return f32(0.0)
    xmin = np.zeros(out_size, dtype=np.int32)
    w = np.zeros((out_size, taps), dtype=np.float32)
    for i in range(out_size):
        center = scale * (f32(i) + f32(0.5))
        lo = max(0, int(center - support + f32(0.5)))
        size = min(int(center + support + f32(0.5)), in_size) - lo
        ws = np.array([cubic((f32(j + lo) - center + f32(0.5)) * invscale) for j in range(size)], dtype=np.float32)
        tot = f32(0.0)
        for v in ws:
            tot = tot + v
        xmin[i] = lo
        w[i, :size] = ws / tot if tot != 0 else ws
    return xmin, w
