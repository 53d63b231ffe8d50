"""Host index bookkeeping (time-r1_amd/positions.py) against transformers' own helpers and hand-derived rules."""
import numpy as np
import pytest
import torch

import time_r1_amd  # noqa: F401
from time_r1_amd import positions as P


GRIDS = [[(2, 4, 6)], [(2, 6, 8)], [(3, 4, 10), (1, 8, 8)], [(16, 12, 22)], [(1, 2, 2)], [(2, 16, 16)]]


@pytest.mark.parametrize("grid", GRIDS)
@pytest.mark.parametrize("window", [56, 112])
def test_window_index_matches_transformers(grid, window):
    vu = pytest.importorskip("transformers.vision_utils")
    want_idx, want_cu = vu.get_vision_window_index(torch.tensor(grid), spatial_merge_size=2, window_size=window, patch_size=14)
    idx, cu = P.vision_window_index(grid, 2, window, 14)
    assert np.array_equal(idx, want_idx.numpy())
    assert np.array_equal(cu, want_cu.numpy().astype(np.int64))
    assert sorted(idx.tolist()) == list(range(len(idx)))               # a permutation of the merged tokens
    # every temporal patch's windows stay contiguous: full-attention segments are valid in window order too
    a = 0
    for t, h, w in grid:
        n = h * w // 4
        for _ in range(t):
            assert sorted(idx[a:a + n].tolist()) == list(range(a, a + n))
            a += n


def test_segments_from_cu():
    pre, lo, hi = P.segments_from_cu([0, 3, 3 + 5, 10])
    assert pre.tolist() == [0] * 10
    assert lo.tolist() == [0] * 3 + [3] * 5 + [8] * 2
    assert hi.tolist() == [2] * 3 + [7] * 5 + [9] * 2


@pytest.mark.parametrize("grid", GRIDS[:4])
def test_vision_segments_and_hw_ids_match_transformers(grid):
    vu = pytest.importorskip("transformers.vision_utils")
    hw = P.vision_hw_ids(grid, 2)
    want = vu.get_vision_position_ids(torch.tensor(grid), 2)
    assert np.array_equal(hw, want.numpy().astype(np.int32))
    pre, lo, hi = P.vision_segments(grid)
    a = 0            # hand rule: one segment per temporal patch
    for t, h, w in grid:
        for _ in range(t):
            assert (lo[a:a + h * w] == a).all() and (hi[a:a + h * w] == a + h * w - 1).all()
            a += h * w
    assert a == len(lo)


@pytest.mark.parametrize("interval", [1, 2, 4])
def test_rope_index_time_interval_and_modes(interval):
    """Qwen2.5-VL temporal spacing (modeling_qwen2_5_vl.py:1043) and the hf4 / hf5 text-resume rules."""
    vid = 501
    grid = [(3, 4, 6)]       # merged 3 x 2 x 3 = 18 tokens
    ids = [5, 6, 502] + [vid] * 18 + [503, 7, 8]
    pos5, d5 = P.rope_index(ids, grid, vid, mode="hf5", time_interval=interval)
    pos4, d4 = P.rope_index(ids, grid, vid, mode="hf4", time_interval=interval)
    assert pos5[:, :3].tolist() == [[0, 1, 2]] * 3
    blk = pos5[:, 3:21]
    assert blk[0].tolist() == sum([[3 + interval * t] * 6 for t in range(3)], [])
    assert blk[1].tolist() == [3, 3, 3, 4, 4, 4] * 3 and blk[2].tolist() == [3, 4, 5] * 6
    assert np.array_equal(pos4[:, :21], pos5[:, :21])
    assert pos5[0, 21] == 3 + 3                                      # hf5: start + max(h, w) // merge
    assert pos4[0, 21] == max(3 + 2 * interval, 5) + 1               # hf4: max over the block's three axes + 1
    assert d5 == int(pos5.max()) + 1 - len(ids) and d4 == int(pos4.max()) + 1 - len(ids)


def test_rope_index_matches_transformers_qwen25():
    """5.15's Qwen2.5-VL get_rope_index with its default second_per_grid_ts (what the reference's logprob forwards get)."""
    pytest.importorskip("transformers")
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from gen_grpo_golden import hf_tiny
    from time_r1_amd.config import tiny_test_25
    cfg = tiny_test_25()
    hf = hf_tiny(cfg)
    grid = [(3, 4, 6)]
    ids = [5, 6, cfg.vision_start_token_id] + [cfg.video_token_id] * 18 + [cfg.vision_end_token_id, 7, 8]
    t = torch.tensor([ids])
    want, delta = hf.model.get_rope_index(t, mm_token_type_ids=(t == cfg.video_token_id).int() * 2, video_grid_thw=torch.tensor(grid),
                                          attention_mask=torch.ones_like(t))
    pos, d = P.rope_index(ids, grid, cfg.video_token_id, cfg.image_token_id, mode="hf5", time_interval=int(cfg.tokens_per_second))
    assert np.array_equal(pos, want[:, 0].numpy())
    assert d == int(delta)
