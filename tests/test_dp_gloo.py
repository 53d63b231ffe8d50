"""Data-parallel path on CPU: 2 processes over gloo. Each rank runs its own prompt (group statistics stay rank-local, reference
timer1_trainer.py:703-712); the optimizer step averages the flat gradient arena. The result must equal a single process that
accumulates both prompts with weight 1/2."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows(fx):
    import numpy as np
    rows = []
    for i in range(2):
        r = dict(fx["row"])
        r["_forced_completion_ids"] = (fx["completion_ids"].numpy() + 3 * i) % 480 + 2
        r["_frames"] = torch.randint(0, 256, (4, 3, 56, 84), generator=torch.Generator().manual_seed(50 + i), dtype=torch.uint8).float()
        rows.append(r)
    return rows


def _worker(rank, world, port, q, wire):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import load_case
    from test_trainer_host_logic import make_trainer
    fx = load_case("grpo_beta")
    cfg, tr = make_trainer(fx, grad_wire_dtype=wire)
    assert tr.dp.enabled and tr.dp.world == 2 and tr.dp.rank == rank
    row = _rows(fx)[rank]
    tr._video_inputs = lambda ex: ([ex["_frames"]], [2.0])
    tr.args.learning_rate = 1e-3
    # one-batch window: the backward of this (last) micro-step hands layer ranges to the all-reduce as they complete (overlap path)
    tr.accumulation_window([[row]])
    assert tr.optimizer.sync.active and len(tr.optimizer.sync.pending) >= cfg.text.n_layers, "layer ranges must be in flight before step()"
    gathered = tr.dp.gather(torch.tensor([float(rank)]))
    tr.optimizer.step()
    assert not tr.optimizer.sync.active
    q.put((rank, tr.params.train.master.clone(), gathered.tolist(), dict(tr._metrics)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_two_rank_step_equals_single_process_average(wire):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_case
    from test_trainer_host_logic import make_trainer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:          # a port the kernel reports free right now (back-to-back runs used to collide on a derived port)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, wire)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1]), "ranks must hold identical weights after the averaged step"
    assert res[0][2] == [0.0, 1.0]
    # metrics are means over the gathered (all-rank) values, like accelerator.gather_for_metrics
    assert res[0][3]["reward"] == res[1][3]["reward"]
    # single process: both prompts, gradient scaled by 1/2 each (gradient_accumulation_steps = 2), one optimizer step
    fx = load_case("grpo_beta")
    cfg, tr = make_trainer(fx, ga=2)
    tr.args.learning_rate = 1e-3
    tr._video_inputs = lambda ex: ([ex["_frames"]], [2.0])
    for row in _rows(fx):
        tr.compute_loss(tr.model, [row])
    tr.optimizer.step()
    if wire == "fp32":
        assert torch.allclose(tr.params.train.master, res[0][1], atol=1e-7, rtol=1e-6)
    else:   # bf16 wire: the summed gradient is rounded to 8 bits of mantissa before AdamW; first-step update = lr * sign-like ratio
        assert torch.allclose(tr.params.train.master, res[0][1], atol=2e-5, rtol=1e-3)
