"""bench.py's launch / step-count logic (the part that crashed the round-1 driver run: `--gpus 1 --steps 20 --warmup 5`)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_driver_command_line_is_accepted():
    a = bench.parse_args(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert (a.gpus, a.steps, a.warmup, a.ga) == (1, 20, 5, 2)
    assert bench.window_plan(a.warmup, a.ga) == [2, 2, 1] and sum(bench.window_plan(a.warmup, a.ga)) == 5
    assert bench.window_plan(a.steps, a.ga) == [2] * 10
    assert bench.resolve_launch(a, {}, 1) == ("run", None)


@pytest.mark.parametrize("k,ga", [(1, 2), (7, 2), (20, 3), (4, 1), (0, 2)])
def test_any_step_count_is_split_into_windows_that_sum_to_it(k, ga):
    plan = bench.window_plan(k, ga)
    assert sum(plan) == k and all(1 <= n <= ga for n in plan) and all(n == ga for n in plan[:-1])


def test_multi_gpu_launch_never_degrades_silently():
    a = bench.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    # under torch.distributed.run: world must equal --gpus
    assert bench.resolve_launch(a, {"WORLD_SIZE": "8"}, 8) == ("run", None)
    with pytest.raises(SystemExit):
        bench.resolve_launch(a, {"WORLD_SIZE": "1"}, 8)
    with pytest.raises(SystemExit):
        bench.resolve_launch(a, {"WORLD_SIZE": "4"}, 8)
    # started plainly: re-launch under torch.distributed.run with 8 ranks on 127.0.0.1 ...
    mode, cmd = bench.resolve_launch(a, {}, 8)
    assert mode == "spawn" and "torch.distributed.run" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-1].endswith("bench.py")
    # ... and fail when the box does not have 8 GPUs
    with pytest.raises(SystemExit):
        bench.resolve_launch(a, {}, 1)
