"""Known-answer table for frame sampling / sizing captured from the reference's src/utils/vision_process.py.
Run here: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_sizing_kat.py -> tests/golden/sizing_kat.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(__file__))
from _ref_harness import load_ref_vision_process  # noqa: E402


def main():
    vp = load_ref_vision_process()
    out = {"smart_resize": [], "smart_nframes": [], "video": []}
    for h, w in [(360, 640), (270, 480), (240, 320), (720, 1280), (1080, 1920), (100, 100), (28, 28), (31, 1000), (480, 854), (13, 17), (2000, 30)]:
        for mn, mx in [(vp.MIN_PIXELS, vp.MAX_PIXELS), (16 * 784, 768 * 784), (16 * 784, 3584 * 784 / 32 * 2), (128 * 784, 256 * 784), (16 * 784, int(16 * 784 * 1.05))]:
            try:
                r = list(vp.smart_resize(h, w, min_pixels=mn, max_pixels=mx))
            except ValueError as e:
                r = "ValueError"
            out["smart_resize"].append({"h": h, "w": w, "min_pixels": mn, "max_pixels": mx, "out": r})
    for ele in [{}, {"nframes": 32}, {"nframes": 7}, {"fps": 1.0}, {"fps": 4.0, "max_frames": 64}, {"min_frames": 10}, {"nframes": 3}]:
        for total, fps in [(300, 30.0), (480, 29.97), (30, 25.0), (5, 30.0), (2, 10.0), (3000, 24.0), (90000, 30.0), (1, 30.0)]:
            try:
                r = vp.smart_nframes(dict(ele), total, fps)
            except (ValueError, AssertionError) as e:
                r = type(e).__name__
            out["smart_nframes"].append({"ele": ele, "total": total, "fps": fps, "out": r})
    # fetch_video_v3's budget + resize target and the timestamp-aware index plan (pure arithmetic restated from :285-334 / :440-466
    # by calling the reference's own helpers in the reference's order)
    import torch
    for nframes in (4, 8, 16, 32, 64, 128, 768):
        for h, w in [(360, 640), (270, 480), (240, 320), (720, 1280)]:
            for ele in [{"total_pixels": 3584 * 784, "min_pixels": 16 * 784}, {}, {"total_pixels": 3584 * 784, "min_pixels": 16 * 784, "max_pixels": 200 * 784}]:
                min_pixels = ele.get("min_pixels", vp.VIDEO_MIN_PIXELS)
                total_pixels = ele.get("total_pixels", vp.VIDEO_TOTAL_PIXELS)
                max_pixels = max(min(vp.VIDEO_MAX_PIXELS, total_pixels / nframes * vp.FRAME_FACTOR), int(min_pixels * 1.05))
                max_pixels = min(ele.get("max_pixels", max_pixels), max_pixels)
                rh, rw = vp.smart_resize(h, w, factor=vp.IMAGE_FACTOR, min_pixels=min_pixels, max_pixels=max_pixels)
                out["video"].append({"kind": "size", "nframes": nframes, "h": h, "w": w, "ele": ele, "out": [rh, rw]})
    for total, fps in [(300, 30.0), (4795, 29.97), (250, 25.0)]:
        for ele in [{}, {"video_start": 2.0, "video_end": 7.5}, {"video_start": 0.0, "video_end": 1.0, "nframes": 8}, {"nframes": 32}, {"video_start": 3.0, "video_end": 3.0}]:
            e = dict(ele)
            video_start = e.get("video_start", 0.0)
            video_end = e.get("video_end", total / fps)
            s = max(0, int(video_start * fps)); en = min(total, int(video_end * fps))
            if en == s:
                en = s + 1
            eff = en - s
            try:
                n = vp.smart_nframes(e, total_frames=eff, video_fps=fps)
                idx = torch.linspace(s, en - 1, n).round().long().tolist()
                sf = n / max(eff, 1e-6) * fps
                r = {"idx": idx, "sample_fps": repr(float(sf))}
            except ValueError:
                r = "ValueError"
            out["video"].append({"kind": "plan", "total": total, "fps": fps, "ele": ele, "out": r})
    p = os.path.join(os.path.dirname(__file__), "sizing_kat.json")
    json.dump({"source": "reference src/utils/vision_process.py via tests/golden/gen_sizing_kat.py", **out}, open(p, "w"))
    print({k: len(v) for k, v in out.items()}, os.path.getsize(p))


if __name__ == "__main__":
    main()
