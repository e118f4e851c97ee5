"""BASELINE configs[4] -- ResNet-50 int8, batch 256 sharded over the 8 GPUs of a node -- as far as it can be checked without 8 GPUs: the driver's
own command line (`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 --config int8`) with bench.py in its recording mode
(RTEN_BENCH_RECORDING=1: a context that logs launches instead of issuing them, CPU tensors, gloo).  Everything that is NOT a kernel runs for real:
rank / world bookkeeping, shard_range (8 x 32 images, per-rank input seeds), the size of the weight arena every rank allocates from host arithmetic,
its broadcast from rank 0, the committed launch plan on every rank and the plan-hash comparison, barrier + max-over-ranks timing, and the ONE
aggregate line rank 0 prints.  No hardware claim: RCCL with more than one rank has never executed for this repository."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, config, port, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTEN_BENCH_RECORDING="1", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="rten_bench_"), "detail.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--config", config, "--recording-test", "--detail-file", detail, *extra]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    r.detail_path = detail
    return r


def _line_and_detail(r):
    """The ONE compact line rank 0 prints (what the driver parses: at most 4 KB) and the full record it points to."""
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) <= 4096
    line = json.loads(lines[0])
    assert line["detail"] == r.detail_path
    full = json.load(open(r.detail_path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "data"):
        assert line[k] == full[k], k
    assert line["config"]["launch_plan"]["sha16"] == full["config"]["launch_plan"]["sha16"]
    return line, full


@pytest.mark.parametrize("config", ["int8", "f32"])
def test_eight_rank_bench_control_flow_of_the_default_path(config):
    """The executor path (the default): rank 0 loads the model for real, ranks 1-7 with the receive-weights flag, every rank sizes the same arena,
    one broadcast, the committed plan on every rank, per-rank seeds, ONE aggregate line that says which path ran."""
    r = _run(8, config, 29615 if config == "int8" else 29617)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line, j = _line_and_detail(r)
    assert line["config"]["path"] == "executor" and line["config"]["lanes"] == 4 and line["config"]["chains"] == 1
    assert line["ranks"]["world_size"] == 8 and line["ranks"]["distinct_plans"] == 1 and len(line["ranks"]["ms_per_step_per_rank"]) == 8
    assert j["config"]["path"] == "executor" and j["n_gpus"] == 8 and j["config"]["global_batch"] == 256 and j["scaling"] == "weak" and j["steps"] == 3
    assert j["data"].startswith("recording") and j["cpu_baseline"] is None and "secondary" not in j
    rk = j["ranks"]
    assert rk["world_size"] == 8 and rk["dist_backend"] == "gloo" and rk["weight_broadcast_world"] == 8
    assert rk["input_seed_per_rank"] == [1234 + r_ for r_ in range(8)] and len(rk["ms_per_step_per_rank"]) == 8
    plan_file = "int8_lanes.json" if config == "int8" else "f32_lanes.json"
    assert j["config"]["launch_plan"]["source"] == os.path.join("profiles", "plans", plan_file)
    assert len(set(rk["plan_sha16_per_rank"])) == 1 and j["config"]["launch_plan"]["identical_on_all_ranks"] is True
    assert len(set(rk["planned_steps_per_rank"])) == 1
    # the default schedule: whole-batch chains, consecutive batches on independent replicas (lanes)
    lanes = 4
    assert j["config"]["batch_chains"]["chains"] == 1 and j["config"]["batch_lanes"]["lanes"] == lanes
    import re
    m = re.search(r"\[recording\] weight arena (\d+) bytes broadcast to 8 ranks", r.stderr)
    assert m and int(m.group(1)) == j["config"]["weight_arena_bytes"] > 0
    assert len(re.findall(r"\[recording\] weight arena \d+ bytes broadcast to 8 ranks", r.stderr)) == 1  # ONE broadcast: the lanes are replicas that share the arena
    rows = {int(m.group(1)): m for m in re.finditer(r"\[recording\] rank (\d+) seed (\d+) shard \[(\d+)\]\.\.\+32 graph_launch (\d+) load (\d+) load_receive (\d+) prepare (\d+) h2d (\d+)", r.stderr)}
    assert sorted(rows) == list(range(8))
    for rank, m in rows.items():
        assert int(m.group(2)) == 1234 + rank and int(m.group(3)) == 32 * rank
        # (the counters are those of the rank's FIRST lane: every lane has a context of its own) only rank 0 loads the weights for real
        assert (int(m.group(5)), int(m.group(6))) == ((1, 0) if rank == 0 else (0, 1))
        assert int(m.group(7)) == 1
    if config == "int8":
        # several ranks per device (the gloo test mode): quantized-output launches stay off, the loader-side quantizers stay in the plan
        plan = json.load(open(os.path.join(ROOT, "profiles", "plans", "int8_lanes.json")))
        assert "qout" not in plan  # replicas side by side: no launch that needs the device to itself
        assert j["config"]["quantized_output_launches"] == [] and sorted(j["config"]["quantize_on_load_layers"]) == sorted(plan["fused_dql"])


@pytest.mark.parametrize("config", ["int8", "f32"])
def test_eight_rank_bench_control_flow(config):
    """... and the same for `--via-runner` (the Python runner: the A/B path)."""
    from rten_amd import lib as L
    from rten_amd.workloads import resnet50, resnet50_int8
    r = _run(8, config, 29611 if config == "int8" else 29613, extra=("--via-runner",))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line, j = _line_and_detail(r)  # rank 0 prints ONE line
    assert line["config"]["path"] == "runner"
    assert j["n_gpus"] == 8 and j["config"]["global_batch"] == 256 and j["scaling"] == "weak" and j["steps"] == 3 and j["warmup"] == 1
    assert j["data"].startswith("recording") and j["cpu_baseline"] is None and "secondary" not in j
    rk = j["ranks"]
    assert rk["world_size"] == 8 and rk["dist_backend"] == "gloo" and rk["weight_broadcast_world"] == 8
    assert len(rk["ms_per_step_per_rank"]) == 8 and rk["input_seed_per_rank"] == [1234 + r_ for r_ in range(8)]
    # one launch plan on every rank: the committed file, hashed per rank AFTER the run (the effective plan), compared by rank 0
    plan_file = "int8.json" if config == "int8" else "f32_4chains.json"
    assert j["config"]["launch_plan"]["source"] == os.path.join("profiles", "plans", plan_file)
    assert len(set(rk["plan_sha16_per_rank"])) == 1 and j["config"]["launch_plan"]["identical_on_all_ranks"] is True
    assert len(set(rk["logits_sha16_per_rank"])) == 1  # (the recording context "downloads" zeros)
    # the arena every rank sized from host arithmetic == what a real build of the network allocates, and it was broadcast to all 8
    so = L.load()
    want_bytes = resnet50_int8.i8_arena_layout(so, 32)[1] if config == "int8" else resnet50.arena_bytes(so, 32)
    assert f"[recording] weight arena {want_bytes} bytes broadcast to 8 ranks" in r.stderr
    # per rank: shard r of the 256-image batch, its own seed, the same launch sequence as every other rank
    import re
    rows = {int(m.group(1)): m for m in re.finditer(r"\[recording\] rank (\d+) seed (\d+) shard \[(\d+)\]\.\.\+32 graph_launch (\d+) qout (\d+) conv (\d+) dql_loader (\d+) h2d (\d+)", r.stderr)}
    assert sorted(rows) == list(range(8))  # (the ranks' stderr lines may interleave: matched by pattern, not by line)
    for rank, m in rows.items():
        assert int(m.group(2)) == 1234 + rank and int(m.group(3)) == 32 * rank
    # ranks 1-7 enqueue exactly the same sequence; rank 0 adds its PCIe-inclusive and instrumented passes on top of it
    seqs = {tuple(int(rows[k].group(i)) for i in (4, 5, 6, 7)) for k in range(1, 8)}
    assert len(seqs) == 1, seqs
    base = next(iter(seqs))
    assert all(int(rows[0].group(i)) >= b for i, b in zip((4, 5, 6, 7), base))
    assert base[0] == 1 + 3 + 3  # graph launches on the rank's main context (f32: chain 0 of 4): warm-up + timed steps + latency pass
    if config == "int8":
        plan = json.load(open(os.path.join(ROOT, "profiles", "plans", "int8.json")))
        # quantized-output launches need a device to themselves: with several ranks per device (the gloo test mode) they stay off; the plan's
        # loader-side quantizers are in the recorded sequence
        assert j["config"]["quantized_output_launches"] == [] and sorted(j["config"]["quantize_on_load_layers"]) == sorted(plan["fused_dql"])


def test_world_size_mismatch_is_refused_in_recording_mode_too():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(RTEN_BENCH_RECORDING="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "int8", "--recording-test"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "refusing to report" in r.stderr
