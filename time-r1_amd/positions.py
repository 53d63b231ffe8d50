"""Host-side index bookkeeping (integers only): M-RoPE position ids, ViT rotary ids / attention segments, and the packed
"one prompt + G completions" sequence layout the engine runs instead of the reference's G-times replicated batch.

Reference: get_rope_index / get_vision_position_ids in transformers/models/qwen2_vl/modeling_qwen2_vl.py:862-1016 (v5.15.0),
vision_utils.py:42-65 (cu_seqlens), :81-127 (rotary ids); SURVEY.md appendix G.3 for the 4.51.1 ("hf4") resume rule.
"""
import numpy as np


def rope_index(input_ids, grid_thw, video_token_id, image_token_id=None, spatial_merge=2, mode="hf5", time_interval=1):
    """input_ids: 1-D int array (one sequence, no padding); grid_thw: list of (t, h, w) for each vision block in order.
    Returns (pos3 int64 [3, L], delta) with delta = max_pos + 1 - L (the decode-time `rope_deltas`).

    mode "hf5": text resumes at start + max(h, w)//merge            (transformers 5.x, modeling_qwen2_vl.py:1008)
    mode "hf4": text resumes at max(all three axes of the block)+1  (transformers 4.51.1, the version the reference pins)
    """
    ids = np.asarray(input_ids).reshape(-1)
    L = ids.shape[0]
    is_vis = ids == video_token_id
    if image_token_id is not None:
        is_vis = is_vis | (ids == image_token_id)
    pos = np.zeros((3, L), dtype=np.int64)
    cur = 0
    i = 0
    gi = 0
    while i < L:
        j = i
        while j < L and is_vis[j] == is_vis[i]:
            j += 1
        n = j - i
        if not is_vis[i]:
            pos[:, i:j] = np.arange(n)[None, :] + cur
            cur += n
        else:
            t, h, w = [int(x) for x in grid_thw[gi]]
            gi += 1
            gh, gw = h // spatial_merge, w // spatial_merge
            assert t * gh * gw == n, "vision block length %d does not match grid %s" % (n, (t, h, w))
            tt, hh, ww = np.meshgrid(np.arange(t) * time_interval, np.arange(gh) + cur, np.arange(gw) + cur, indexing="ij")
            pos[0, i:j] = tt.reshape(-1) + cur
            pos[1, i:j] = hh.reshape(-1)
            pos[2, i:j] = ww.reshape(-1)
            if mode == "hf5":
                cur += max(h, w) // spatial_merge
            elif mode == "hf4":
                cur = int(pos[:, i:j].max()) + 1
            else:
                raise ValueError("rope_index mode must be hf4 or hf5")
        i = j
    delta = int(pos.max()) + 1 - L if L else 0
    return pos, delta


def vision_hw_ids(grid_thw, spatial_merge=2):
    """[N_v, 2] (h, w) ids in merge-block-major order, repeated per temporal patch (vision_utils.py:81-127)."""
    out = []
    for t, h, w in grid_thw:
        t, h, w = int(t), int(h), int(w)
        hh, ww = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        m = spatial_merge
        hh = hh.reshape(h // m, m, w // m, m).transpose(0, 2, 1, 3).reshape(-1)
        ww = ww.reshape(h // m, m, w // m, m).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hh, ww], -1), (t, 1)))
    return np.concatenate(out, 0).astype(np.int32)


def vision_segments(grid_thw):
    """Per-patch (pre, lo, hi) for the two-interval mask: every temporal patch attends within itself (vision_utils.py:42-65)."""
    lo, hi = [], []
    a = 0
    for t, h, w in grid_thw:
        n = int(h) * int(w)
        for _ in range(int(t)):
            lo += [a] * n
            hi += [a + n - 1] * n
            a += n
    return np.zeros(a, dtype=np.int32), np.asarray(lo, dtype=np.int32), np.asarray(hi, dtype=np.int32)


def segments_from_cu(cu):
    """(pre, lo, hi) int32 [N] from cumulative segment boundaries: row r in [cu[j], cu[j+1]) attends exactly that range."""
    cu = np.asarray(cu, dtype=np.int64)
    n = np.diff(cu)
    lo = np.repeat(cu[:-1], n)
    hi = np.repeat(cu[1:] - 1, n)
    return np.zeros(int(cu[-1]), dtype=np.int32), lo.astype(np.int32), hi.astype(np.int32)


def vision_window_index(grid_thw, spatial_merge=2, window_size=112, patch_size=14):
    """Qwen2.5-VL window partition (transformers vision_utils.py:130-188, get_vision_window_index).  Works in units of merged tokens
    (spatial_merge^2 consecutive patches): every temporal patch's (h/m, w/m) grid is cut into windows of mw x mw merged tokens
    (mw = window_size // merge // patch), ragged at the right/bottom edges.
    Returns (window_index int64 [N/unit]: window-major order -> natural merged-token index,
             cu_window int64 [n_windows+1]: window boundaries in PATCH units)."""
    mw = window_size // spatial_merge // patch_size
    unit = spatial_merge ** 2
    order, cu = [], [0]
    base = 0
    for t, h, w in grid_thw:
        t, gh, gw = int(t), int(h) // spatial_merge, int(w) // spatial_merge
        idx = np.arange(t * gh * gw, dtype=np.int64).reshape(t, gh, gw) + base
        for ti in range(t):
            for a in range(0, gh, mw):
                for b in range(0, gw, mw):
                    win = idx[ti, a:a + mw, b:b + mw].reshape(-1)
                    order.append(win)
                    cu.append(cu[-1] + win.size * unit)
        base += t * gh * gw
    return np.concatenate(order), np.asarray(cu, dtype=np.int64)


class PackedLayout:
    """Packed sequence for one prompt and its G completions of (up to) C tokens:

        row / KV slot index:  [0, P)                      prompt tokens (shared by all groups)
                              P + g*C + s                 completion token s of group g

    Completion (g, s) sees the whole prompt plus its own group's tokens 0..s - exactly the causal context row g of the
    reference's replicated batch has (timer1_trainer.py:592-607), so per-token results are identical while the prompt is
    processed once instead of G times.
    """

    def __init__(self, P, G, C):
        self.P, self.G, self.C = int(P), int(G), int(C)
        self.M = self.P + self.G * self.C
        self.S_cap = (self.M + 63) // 64 * 64

    def masks(self):
        P, G, C = self.P, self.G, self.C
        pre = np.concatenate([np.zeros(P), np.full(G * C, P)]).astype(np.int32)
        lo = np.concatenate([np.zeros(P), P + np.repeat(np.arange(G), C) * C]).astype(np.int32)
        hi = np.arange(self.M).astype(np.int32)
        return pre, lo, hi

    def prompt_masks(self):
        P = self.P
        return np.zeros(P, dtype=np.int32), np.zeros(P, dtype=np.int32), np.arange(P, dtype=np.int32)

    def decode_masks(self, step):
        """masks for the G single-token queries that attend after completion token `step` was appended"""
        P, G, C = self.P, self.G, self.C
        lo = (P + np.arange(G) * C).astype(np.int32)
        return np.full(G, P, dtype=np.int32), lo, (lo + step).astype(np.int32)

    def completion_slots(self, step):
        return (self.P + np.arange(self.G) * self.C + step).astype(np.int32)

    def pred_rows(self):
        """Rows whose hidden state predicts completion token (g, s): the last prompt row for s = 0, else the previous
        completion row. Ordered [the G copies of row P-1] + [(g, s>=1) in row-major order] (see model.lm_head stage)."""
        P, G, C = self.P, self.G, self.C
        first = np.full(G, P - 1)
        rest = (P + np.arange(G)[:, None] * C + np.arange(C - 1)[None, :]).reshape(-1)
        return np.concatenate([first, rest]).astype(np.int32)

    def positions(self, prompt_pos3, delta):
        """pos3 [3, M] for the packed sequence: completion token s of every group sits at text position P + delta + s."""
        P, G, C = self.P, self.G, self.C
        comp = (P + delta + np.arange(C))[None, :].repeat(3, 0)
        return np.concatenate([prompt_pos3] + [comp] * G, axis=1).astype(np.int32)
