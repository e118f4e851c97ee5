"""Batch-sharded deployment, rank != 0 path on real hardware: two ranks share the one GPU over gloo."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_one_gpu_identical_logits():
    env = dict(os.environ, RTEN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29557",
           os.path.join(ROOT, "tools", "check_multi_rank.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "identical_logits=True" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["f32", "int8"])
def test_bench_two_ranks_prints_one_line(config):
    """`python bench.py --gpus 2` (the form the driver uses on a multi-GPU node), here with both ranks on the one GPU over gloo: the
    script spawns the ranks itself, rank 1 receives the weight arena only by broadcast, rank 0 prints ONE line that says 2 GPUs,
    carries both ranks' step times and the default schedule (whole-batch chains; consecutive batches on 3 (f32) / 4 (int8) replicas per rank)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTEN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29561" if config == "f32" else "29563")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--config", config], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 4096, r.stdout[-2000:]
    line = json.loads(lines[0])  # the compact line the driver parses ...
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2 and line["ranks"]["distinct_plans"] == 1 and line["config"]["lanes"] == 4
    j = json.load(open(os.path.join(ROOT, line["detail"])))  # ... and the full record it points to
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 64 and j["scaling"] == "weak"
    assert j["ranks"]["world_size"] == 2 and len(j["ranks"]["ms_per_step_per_rank"]) == 2 and j["ranks"]["weight_broadcast_world"] == 2
    assert j["cpu_baseline"] is None and j["roofline"]["frac"] > 0
    assert j["config"]["batch_chains"]["chains"] == 1 and j["config"]["batch_lanes"]["lanes"] == 4
    # every rank ran the same launch plan, and rank r's logits are the oracle's for ITS shard (inputs seeded 1234 + r): the parity
    # definition of a batch-sharded run -- for int8 each shard quantizes with its own statistics, like an independent run
    import hashlib

    import numpy as np
    from tests import baseline_oracle as bo
    assert j["config"]["path"] == "executor"  # the default path: rank 1 loaded with the receive-weights flag and got the arena by broadcast
    assert len(set(j["ranks"]["plan_sha16_per_rank"])) == 1
    for r, (sha, seed) in enumerate(zip(j["ranks"]["logits_sha16_per_rank"], j["ranks"]["input_seed_per_rank"])):
        assert seed == 1234 + r
        want = bo.resnet50_int8_logits(seed) if config == "int8" else bo.resnet50_f32_logits(seed)
        assert hashlib.sha256(np.ascontiguousarray(want.astype(np.float32)).tobytes()).hexdigest()[:16] == sha, f"rank {r}: logits differ from the oracle's for its shard"


@pytest.mark.gpu
def test_bench_two_ranks_tuning_run_broadcasts_one_plan():
    """--autotune with two ranks: rank 0 tunes, the plan is broadcast, both ranks report the same plan hash (no per-rank tuning)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTEN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29565")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "int8", "--autotune"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    j = json.load(open(os.path.join(ROOT, j["detail"])))
    shas = j["ranks"]["plan_sha16_per_rank"]
    assert len(shas) == 2 and shas[0] == shas[1] and shas[0] is not None
    assert "rank 0" in j["config"]["launch_plan"]["source"]
