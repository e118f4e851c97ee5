"""bench.py's launch contract (no GPU needed): `--gpus N` means N ranks.  Under a launcher whose WORLD_SIZE differs the script
must fail loudly instead of printing N x the single-GPU rate; without a launcher it must start the ranks itself."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)


def test_gpus_2_without_matching_ranks_fails_loudly():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert p.returncode != 0
    assert "WORLD_SIZE=1" in p.stderr and "--gpus 2" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]  # and no JSON line


def test_world_size_larger_than_gpus_fails_too():
    p = _run(["--gpus", "1"], {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_int8_cannot_be_split_into_chains():
    """Sub-batch chains are an f32-only schedule: the int8 graph's quantizers reduce over the whole batch."""
    p = _run(["--config", "int8", "--chains", "2"], {})
    assert p.returncode == 2 and "one chain" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    p = _run(["--chains", "9"], {})
    assert p.returncode == 2


def test_gpus_n_without_launcher_spawns_n_ranks(monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, *a, **kw):
        seen["cmd"] = cmd
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "int8"])
    assert bench.main() == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "int8"]


def test_int8_algorithmic_bytes_formula():
    """The HBM floor of the dynamically quantized graph (roofline.step of --config int8) from the layer list alone."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    import bench
    from rten_amd import lib as L
    from rten_amd.workloads import resnet50

    class Net:  # shapes only: no device
        specs = resnet50.conv_specs()
        num_classes = 1000
    shapes = {"x": (32, 3, 224, 224)}
    Net.descs = {}
    for l in Net.specs:
        n, c, h, w = shapes[l["src"]]
        oh = (h + 2 * l["pad"] - l["k"]) // l["stride"] + 1
        Net.descs[l["name"]] = L.Conv2dDesc(n, c, h, w, l["cout"], l["k"], l["k"], (C.c_int32 * 4)(*([l["pad"]] * 4)), l["stride"], l["stride"], 1, 1, 1, oh, oh)
        shapes[l["dst"]] = (n, l["cout"], oh, oh)
        if l["dst"] == "stem":
            shapes["pool"] = (n, l["cout"], (oh + 2 - 3) // 2 + 1, (oh + 2 - 3) // 2 + 1)
    Net.shapes = shapes
    Net.pool_desc = L.Pool2dDesc(32, 64, 112, 112, 3, 3, 2, 2, (C.c_int32 * 4)(1, 1, 1, 1), 56, 56, 0)
    total = bench.int8_algorithmic_bytes(Net)
    assert 3.5e9 < total < 5.0e9  # about 4.3 GB per 32-image batch
    # the executor path of bench.py has no runner object: the same floor from host arithmetic alone, and the as-launched floor of a plan's edges
    assert bench.int8_graph_floor_bytes(32) == total
    import json
    q = json.load(open(os.path.join(ROOT, "profiles", "plans", "int8.json")))["qout"]
    assert 0.8 * total < bench.int8_graph_floor_bytes(32, q) < total


def test_recording_mode_needs_its_flag():
    """RTEN_BENCH_RECORDING=1 alone (an environment leak) never turns a benchmark run into a recording."""
    p = _run(["--steps", "1", "--warmup", "0"], {"RTEN_BENCH_RECORDING": "1"})
    assert p.returncode == 2 and "--recording-test" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_split_batch_covers_the_batch_once():
    sys.path.insert(0, ROOT)
    import pytest
    from rten_amd.workloads.resnet50 import split_batch
    for batch in (1, 5, 32, 33):
        for chains in range(1, min(batch, 8) + 1):
            sizes, starts = split_batch(batch, chains)
            assert sum(sizes) == batch and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
            assert starts == [sum(sizes[:i]) for i in range(chains)] and starts[0] == 0
    assert split_batch(32, 4) == ([8, 8, 8, 8], [0, 8, 16, 24]) and split_batch(5, 2) == ([3, 2], [0, 3])
    with pytest.raises(ValueError):
        split_batch(3, 4)


def test_the_lanes_plan_names_the_same_steps_as_the_one_chain_plan():
    """profiles/plans/f32_lanes.json (the default line's plan: one chain per replica, four replicas side by side) names exactly the steps of f32_1chain.json plus the
    classifier Gemm, pinned to its 64x64 tiles (profiles/r08/classifier_under_lanes.txt).  Since round 6 its convolution entries are chosen PER LAYER UNDER CO-RUN
    (tools/tune_corun.py: three streams running the same layer; profiles/r10/tune_corun3_full.txt) -- larger tiles where the other replicas fill their quantisation
    gaps -- so they differ from the one-replica plan; every entry is a [variant, split mode, K groups, tile order] the backend accepts."""
    import json
    plans = os.path.join(ROOT, "profiles", "plans")
    one, lanes = json.load(open(os.path.join(plans, "f32_1chain.json"))), json.load(open(os.path.join(plans, "f32_lanes.json")))
    assert set(lanes) == set(one) | {"fc", "pairs", "pair_shortcuts"}
    assert lanes["fc"] == [3, 3, 1, 0]
    # "pairs" (round 6, third part): stage 0's three expand layers each run WITH the reduce layer that reads them, in one launch (rten_hip_conv2d_f32_pair: the second
    # convolution takes its operand from LDS; profiles/r11/f32_pairs_ab.txt).  The four-chain plan lists them too; the one-replica one-chain plan gains nothing from them
    pairs = lanes.pop("pairs")
    assert pairs == ["s0b0c3", "s0b1c3", "s0b2c3"] and all(p in one for p in pairs) and "pairs" not in one
    assert json.load(open(os.path.join(plans, "f32_4chains.json")))["pairs"] == pairs
    # "pair_shortcuts": the first pair also computes its residual -- stage 0's shortcut convolution, which nothing else reads -- in the launch
    # (rten_hip_conv2d_f32_pair_shortcut; profiles/r11/f32_pair_shortcut_ab.txt)
    assert lanes.pop("pair_shortcuts") == ["s0b0c3"] and json.load(open(os.path.join(plans, "f32_4chains.json")))["pair_shortcuts"] == ["s0b0c3"]
    for name, e in lanes.items():
        assert len(e) == 4 and 0 <= e[0] <= 32 and 0 <= e[1] <= 6 and 1 <= e[2] <= 32 and 0 <= e[3] <= 3, (name, e)
        assert e[0] != 31 or name == "fc", (name, e)  # (31 = the small-M streaming kernel: a convolution given it runs as variant 3 -- the plan says 3)
    bert = json.load(open(os.path.join(plans, "bert_base_b32_s128.json"))), json.load(open(os.path.join(plans, "bert_base_b32_s128_lanes.json")))
    assert list(bert[0]) == ["32"] and sorted(bert[1]) == ["32", "shapes"] and set(bert[0]["32"]) == set(bert[1]["32"]) and len(bert[1]["32"]) == 48
    # "shapes": the same four choices keyed by product shape (gemm:MxKxN), what another exporter's file of the model takes (include/rten_hip_graph.hpp)
    assert sorted(bert[1]["shapes"]) == ["gemm:4096x3072x768", "gemm:4096x768x2304", "gemm:4096x768x3072", "gemm:4096x768x768"]
    assert all(v in bert[1]["32"].values() for v in bert[1]["shapes"].values())


def _recorded_full_lines():
    """Full (pre-compaction) records of earlier GPU runs kept under profiles/: the inputs the compact line is built from."""
    import glob
    import json
    out = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[6-9]", "bench*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r1*", "bench*detail*.json"))):
        try:
            text = open(path).read().strip()
            j = json.loads(text if text.startswith("{") and "\n{" not in text else [l for l in text.splitlines() if l.startswith("{")][-1])
        except Exception:  # noqa: BLE001
            continue
        if isinstance(j, dict) and "metric" in j and "value" in j:
            out.append((os.path.relpath(path, ROOT), j))
    return out


def test_the_printed_line_is_compact_and_complete():
    """The driver keeps a bounded tail of stdout (round 5: a 23 KB line came back `parsed: null`).  The line rank 0 prints is built from the full record by
    bench.compact_line: at most 4096 bytes, strict JSON (no NaN / Infinity), with every contract field plus `roofline` and `cpu_baseline`."""
    sys.path.insert(0, ROOT)
    import json
    import bench
    records = _recorded_full_lines()
    assert len(records) >= 3, records
    biggest = 0
    for path, full in records:
        biggest = max(biggest, len(json.dumps(full)))
        text = bench.compact_line(full, "gpurun_out/bench_detail.json")
        assert len(text) <= bench.LINE_LIMIT == 4096, (path, len(text))
        assert "NaN" not in text and "Infinity" not in text
        line = json.loads(text)
        assert json.loads(json.dumps(line)) == line
        for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "dtype"):
            assert line[k] == full[k], (path, k)
        assert line["detail"] == "gpurun_out/bench_detail.json"
        if full.get("roofline"):
            r = line["roofline"]
            assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r), (path, r)
            assert r["frac"] == full["roofline"]["frac"]
        if full.get("cpu_baseline"):
            assert {"value", "unit", "cores", "kind"} <= set(line["cpu_baseline"]), path
        for name, sec in (full.get("secondary") or {}).items():
            if "error" in sec:
                continue
            assert line["secondary"][name]["value"] == sec["value"], (path, name)
            if sec.get("roofline"):
                assert line["secondary"][name]["roofline"]["frac"] == sec["roofline"]["frac"]
            if sec.get("cpu_baseline"):
                assert line["secondary"][name]["cpu_baseline"]["value"] == sec["cpu_baseline"]["value"]
    assert biggest > 20000  # round 5's 23 KB record is among the inputs: the case that broke the driver's parse


def test_compact_line_survives_hostile_records():
    """NaN / Infinity become null; a record that is still too long sheds its optional parts (secondary last) instead of overflowing."""
    sys.path.insert(0, ROOT)
    import json
    import bench
    full = {"metric": "m", "value": float("nan"), "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": float("inf"), "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w" * 5000, "launch_plan": {"source": "s", "sha16": "0" * 16}},
            "roofline": {"bound": "mfma", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5, "traffic": None, "kernel": "k" * 300, "shapes": [{"x": 1}] * 500},
            "cpu_baseline": {"value": 1.0, "unit": "u", "cores": 2, "kind": "port", "sample": "s" * 1000},
            "secondary": {f"cfg{i}": {"value": 1.0, "unit": "u", "ms_per_step": 1.0, "roofline": {"bound": "hbm", "frac": 0.1, "kernel": "k" * 400},
                                      "cpu_baseline": {"value": 1.0, "unit": "u", "cores": 1, "kind": "port"}} for i in range(12)}}
    text = bench.compact_line(full)
    assert len(text) <= 4096
    line = json.loads(text)
    assert line["value"] is None and line["ms_per_step"] is None
    assert line["roofline"]["frac"] == 0.5 and line["cpu_baseline"]["value"] == 1.0
