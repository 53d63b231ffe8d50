"""bench.py end to end on the GPU with the tiny model: the driver's argument shape (odd warmup), the self-spawned multi-rank launch
(2 ranks sharing the one GPU of the test box over gloo - the rendezvous, the world-size check and the max-over-ranks timing are the
same code the 8-GPU RCCL run uses) and the sharded optimizer flag."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "tiny", "--G", "4", "--C", "8", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_driver_argument_shape_runs_and_reports_contract_fields():
    out = _run(["--gpus", "1", "--steps", "5", "--warmup", "3"])
    assert out["metric"] == "grpo_samples_per_sec" and out["unit"] == "samples/s" and out["n_gpus"] == 1
    assert out["steps"] == 5 and out["warmup"] == 3 and out["optimizer_steps"] == 3 and out["windows"] == "2 x 2 + 1 x 1"
    assert out["value"] > 0 and abs(out["value"] - 1000.0 / out["ms_per_step"]) < 1e-6 * out["value"]
    assert out["scaling"] == "weak" and out["dtype"] == "bf16" and out["vs_baseline"] is None and "workload" in out["config"]
    assert {"preprocess", "vision", "rollout", "logps", "backward", "optimizer"} <= set(out["phases_ms_per_step"])
    for k in ("roofline", "roofline_secondary"):
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "peak_measured"} <= set(out[k])
    assert out["peak_probe"]["hbm_copy_GBs"] > 500 and out["peak_probe"]["hbm_read_stream_GBs"] > 500
    for k in ("roofline", "roofline_secondary"):                # `peak_measured` >= the best own launch, so the fraction cannot exceed 1
        assert out[k]["frac_of_measured"] is None or out[k]["frac_of_measured"] <= 1.0 + 1e-9, out[k]
        assert "traffic_guard" in out[k]
    assert {"samples_per_sec", "rollout_tokens_per_sec", "perf/decode_hbm_frac", "perf/train_mfma_frac"} <= set(out["trainer_log_last"])


def test_ragged_eos_run_counts_only_tokens_up_to_the_injected_eos():
    full = _run(["--steps", "2", "--warmup", "0", "--no-roofline", "--no-peak-probe"])
    rag = _run(["--steps", "2", "--warmup", "0", "--no-roofline", "--no-peak-probe", "--ragged-eos"])
    assert "ragged" in rag["config"]["completion_lengths"] and "all C" in full["config"]["completion_lengths"]
    tok = lambda o: o["generated_tokens_per_sec_end_to_end"] * o["ms_per_step"] * o["steps"] / 1000.0      # tokens counted in the timed region
    assert tok(full) <= 2 * 4 * 8 + 1e-3                       # 2 micro-steps x G = 4 x C = 8 (a sampled EOS may shorten a row)
    assert 0 < tok(rag) < tok(full)                            # same seeds, same samples: the injected EOS (kept, lengths in [C/2 + 1, C]) only shortens rows


@pytest.mark.parametrize("shard", [False, True])
def test_self_spawned_two_rank_run(shard):
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-roofline", "--no-peak-probe"] + ([] if shard else ["--replicated-optimizer"]),
               env={"TR1_FORCE_DEVICE": "0", "TR1_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["steps"] == 3
    assert abs(out["value"] - 2 * 1000.0 / out["ms_per_step"]) < 1e-6 * out["value"]       # whole-job aggregate over both ranks
    assert ("zero-sharded" in out["config"]["optimizer"]) == shard
    d = out["distributed"]                                      # a first multi-GPU run must be diagnosable from its JSON line
    assert d["backend"] == "gloo" and d["world"] == 2 and d["ranks_seen"] == [0, 1] and d["ranks_seen_ok"] and len(d["devices"]) == 2
    assert d["optimizer_sharded"] == shard and d["grad_exchange_exposed_ms_per_optimizer_step"] >= 0.0 and d["process_group_timeout_s"] > 0
    assert "rccl_version" in d
    assert len(d["hbm_gb_allocated_peak_per_rank"]) == 2 and all(m > 0 for m in d["hbm_gb_allocated_peak_per_rank"])       # measured per rank
    test_self_spawned_two_rank_run.mem = getattr(test_self_spawned_two_rank_run, "mem", {})
    test_self_spawned_two_rank_run.mem[shard] = max(d["hbm_gb_allocated_peak_per_rank"])
    if len(test_self_spawned_two_rank_run.mem) == 2:           # master / m / v sharded over 2 ranks: the per-rank peak must not grow
        assert test_self_spawned_two_rank_run.mem[True] <= test_self_spawned_two_rank_run.mem[False] * 1.02, test_self_spawned_two_rank_run.mem


@pytest.mark.parametrize("shard", [False, True])
def test_single_rank_process_group_runs_the_collective_path_on_rccl(shard):
    """TR1_DIST_FORCE=1 under the driver's own launch line (torch.distributed.run, 1 rank): the process group is built on "nccl" (= RCCL) and the
    gradient exchange of the data-parallel path (all-reduce; reduce-scatter + all-gather when sharded), the diagnostics collectives, the barrier and the
    max-over-ranks timing all execute on the GPU with world size 1 - every torch.distributed call the 8-GPU run makes, on the one GPU this box has."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ, TR1_DIST_FORCE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TR1_DIST_BACKEND", "TR1_FORCE_DEVICE"):
        e.pop(k, None)
    base = ["--model", "tiny", "--G", "4", "--C", "8", "--no-cpu-baseline", "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-roofline", "--no-peak-probe"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py")] + base + ([] if shard else ["--replicated-optimizer"])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    d = out["distributed"]
    assert d["backend"] == "nccl" and d["world"] == 1 and d["ranks_seen"] == [0] and d["ranks_seen_ok"] and d["optimizer_sharded"] == shard
    assert ("zero-sharded" in out["config"]["optimizer"]) == shard and out["n_gpus"] == 1
    plain = _run(base[6:])                                       # the same run without a process group: same seeds, same samples, same update
    a, b = out["trainer_log_last"], plain["trainer_log_last"]
    for k in ("loss", "grad_norm", "reward", "completion_length"):
        if k in a and k in b:
            assert abs(a[k] - b[k]) <= 2e-3 * max(1.0, abs(b[k])), (k, a[k], b[k])


def test_world_size_mismatch_is_an_error():
    e = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "tiny", "--gpus", "2"], capture_output=True, text=True, env=e, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
