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
    cfg, tr = make_trainer(fx, grad_wire_dtype=wire, shard_optimizer=False)       # the replicated all-reduce path (N > 1 defaults to sharding)
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
    # numpy (pickled by value): torch tensors travel through shared-memory handles that die with a worker that exits before the parent reads
    q.put((rank, tr.params.train.master.numpy().copy(), gathered.tolist(), dict(tr.flush_metrics())))
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
    res = [(r[0], torch.from_numpy(r[1]), r[2], r[3]) for r in res]
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


# ------------------------------------------------------------------------------------------ ZeRO-style sharded optimizer (config 4)
def _worker_sharded(rank, world, port, q, wire, shard):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import load_case
    from test_trainer_host_logic import make_trainer
    from time_r1_amd.dist import ShardSync, GradSync
    fx = load_case("grpo_beta")
    cfg, tr = make_trainer(fx, grad_wire_dtype=wire, shard_optimizer=shard)
    assert isinstance(tr.optimizer.sync, ShardSync if shard else GradSync)
    a = tr.params.train
    if shard:
        assert a.shard == (rank, world) and a.master.numel() == a.numel // world == a.m.numel() == a.v.numel()
    tr._video_inputs = lambda ex: ([ex["_frames"]], [2.0])
    tr.args.learning_rate = 1e-3
    norms = []
    for step in range(2):                      # two optimizer steps: the second one sees non-zero m / v and freshly zeroed gradients
        row = _rows(fx)[rank]
        row["_forced_completion_ids"] = (row["_forced_completion_ids"] + 5 * step) % 480 + 2
        tr.accumulation_window([[row]])
        if shard:
            assert tr.optimizer.sync.active and len(tr.optimizer.sync.pending) >= cfg.text.n_layers, "layer segments must be in flight before step()"
        norms.append(float(tr.optimizer.step()))
        from helpers import grads_cleared
        assert grads_cleared(tr)
    master_full = torch.zeros(a.numel)
    if shard:
        for (ca, cb, la) in a.chunks():
            master_full[ca:cb] = a.master[la:la + cb - ca]
        dist.all_reduce(master_full)          # chunks of different ranks are disjoint
    else:
        master_full.copy_(a.master)
    q.put((rank, a.w16.float().numpy().copy(), master_full.numpy().copy(), norms))
    dist.barrier()
    dist.destroy_process_group()


def _spawn2(target, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + extra) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return [tuple(torch.from_numpy(x) if hasattr(x, "dtype") and not torch.is_tensor(x) else x for x in r) for r in res]


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_sharded_optimizer_equals_replicated_optimizer(wire):
    """world = 2: reduce-scatter -> AdamW on the local half of every segment -> all-gather must leave every rank with the weights the
    replicated path (all-reduce -> full AdamW everywhere) produces, after TWO optimizer steps, including the global-norm clip."""
    sharded = _spawn2(_worker_sharded, wire, True)
    plain = _spawn2(_worker_sharded, wire, False)
    assert torch.equal(sharded[0][1], sharded[1][1]), "every rank must hold the same full bf16 weights after the all-gather"
    assert torch.equal(sharded[0][2], sharded[1][2])
    assert sharded[0][3] == sharded[1][3] and abs(sharded[0][3][0] - plain[0][3][0]) < 1e-5 * max(1.0, plain[0][3][0])     # same global grad norm
    tol = dict(atol=1e-7, rtol=1e-6) if wire == "fp32" else dict(atol=2e-5, rtol=1e-3)
    assert torch.allclose(sharded[0][2], plain[0][2], **tol), float((sharded[0][2] - plain[0][2]).abs().max())
    assert torch.allclose(sharded[0][1].float(), plain[0][1].float(), atol=1e-2 if wire == "bf16" else 1e-6, rtol=1e-2)


def test_arena_segments_split_evenly_for_every_world_size():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.ref_ops import RefOps
    from time_r1_amd.config import tiny_test, tiny_test_25
    from time_r1_amd.params import ModelParams, SEG_ALIGN
    for cfg in (tiny_test(), tiny_test(tie=True), tiny_test_25()):
        a = ModelParams(cfg, RefOps(), seed=0).train
        assert a.segments[0][1] == 0 and a.segments[-1][2] == a.numel
        assert all(s[2] == n[1] for s, n in zip(a.segments, a.segments[1:]))
        assert all(s[1] % SEG_ALIGN == 0 and s[2] % SEG_ALIGN == 0 for s in a.segments)
        keys = [s[0] for s in a.segments]
        assert keys[0] == "embed" and "l0" in keys and "norm" in keys and "merger" in keys and len(set(keys)) == len(keys)
        for prefix in ["l%d." % i for i in range(cfg.text.n_layers)] + ["norm", "embed", "merger"]:
            r = a.range_of(prefix)
            assert r in [(s[1], s[2]) for s in a.segments], (prefix, r)
        for world in (1, 2, 4, 8):
            cover = torch.zeros(a.numel, dtype=torch.int32)
            for r in range(world):
                loc = 0
                for ca, cb, la in a.chunks(r, world):
                    assert la == loc and (cb - ca) % 64 == 0
                    loc += cb - ca
                    cover[ca:cb] += 1
                assert loc == a.numel // world
            assert bool((cover == 1).all())


# ------------------------------------------------------------------------------------------ 4 ranks, dataset not divisible by the world size
def _worker_train4(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), TR1_DIST_TIMEOUT_S="240")
    torch.set_num_threads(1)
    from time_r1_amd.dist import init_from_env, dist_diagnostics, ShardSync
    r, _, w = init_from_env("cpu")                 # the product's own rendezvous (explicit process-group timeout), gloo on CPU
    assert (r, w) == (rank, world)
    diag = dist_diagnostics("cpu")
    from helpers import load_case
    from test_trainer_host_logic import make_trainer, _dataset
    fx = load_case("grpo_beta")
    cfg, tr = make_trainer(fx, disable_log_print=True)       # shard_optimizer=None: N > 1 defaults to the sharded optimizer
    assert isinstance(tr.optimizer.sync, ShardSync) and tr.params.train.shard == (rank, world)
    tr.args.learning_rate = 1e-3
    tr.args.num_train_epochs = 1
    tr.train_dataset = _dataset(fx, 5)             # 5 rows over 4 ranks: the sampler wraps to 8 = 2 batches per rank (no rank may run short)
    res = tr.train()
    q.put((rank, tr.params.train.w16.float().numpy().copy(), res.global_step, diag, tr.state.log_history[-1]["reward"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_ranks_dataset_not_divisible_by_world(world):
    """4 (and 8: the driver's scaling run) ranks over gloo, len(dataset) % world != 0, default (sharded) optimizer, the
    product's rendezvous with its explicit timeout: every rank takes the same number of optimizer steps (no dead-lock in the exchange), ends
    with identical weights, and the start-up diagnostics see every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker_train4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    steps = 2 if world == 4 else 1               # 5 rows wrap to 8: 2 batches per rank at 4 ranks (GA = 1 each), 1 at 8
    assert [r[2] for r in res] == [steps] * world, [r[2] for r in res]
    for r in res[1:]:
        assert (r[1] == res[0][1]).all(), "ranks must hold identical bf16 weights after the all-gather"
        assert r[4] == res[0][4]                   # gathered metric means agree on every rank
    d = res[0][3]
    assert d["backend"] == "gloo" and d["world"] == world and d["ranks_seen"] == list(range(world)) and d["ranks_seen_ok"] and len(d["devices"]) == world
