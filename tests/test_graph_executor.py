"""ONNX loader + device-resident graph executor (include/rten_hip_graph.hpp, tools/rten_hip_run.cpp) -- SURVEY 8f ranks 1/3.

Models are manufactured by rten_amd/onnx_writer.py (there is no onnx / onnxruntime package here), read back by the C++
loader, and executed on the GPU with every value resident in HBM between operators.  The logits must carry the same
bits as the CPU oracle's forward pass (oracle/models.py) -- with and without the fusion passes.
"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "rten_amd", "bin", "rten_hip_run")


def build_cli():
    from rten_amd import lib as L
    L.load()  # raises if librten_hip.so is missing
    src = os.path.join(ROOT, "tools", "rten_hip_run.cpp")
    deps = [src] + [os.path.join(ROOT, "include", h) for h in ("rten_hip_graph.hpp", "rten_hip_ops.hpp", "rten_hip_safetensors.hpp", "rten_hip.h")]
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", BIN,
                               "-L" + os.path.join(ROOT, "rten_amd"), "-lrten_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + os.path.join(ROOT, "rten_amd"),
                               "-Wl,-rpath,/opt/rocm/lib"])  # $ORIGIN/.. = rten_amd/: the binary finds the library wherever the tree is mounted
    return BIN


def run_cli(*args, timeout=600):
    return subprocess.run([build_cli(), *args], capture_output=True, text=True, timeout=timeout)


def decode_raw(buf):
    """Independent structural check of the protobuf encoding: walk (key, value) pairs of one message level."""
    out, i = [], 0

    def varint(i):
        v, s = 0, 0
        while True:
            b = buf[i]
            i += 1
            v |= (b & 0x7F) << s
            s += 7
            if not b & 0x80:
                return v, i
    while i < len(buf):
        k, i = varint(i)
        f, w = k >> 3, k & 7
        if w == 0:
            v, i = varint(i)
        elif w == 2:
            n, i = varint(i)
            v = buf[i:i + n]
            assert len(v) == n
            i += n
        elif w == 5:
            v = buf[i:i + 4]
            i += 4
        elif w == 1:
            v = buf[i:i + 8]
            i += 8
        else:
            raise AssertionError(f"wire type {w}")
        out.append((f, w, v))
    return out


def test_writer_emits_wellformed_protobuf():
    from rten_amd import onnx_writer as ow
    m, _ = ow.small_cnn_f32()
    top = decode_raw(m)
    assert [f for f, _, _ in top] == [1, 2, 7, 8]  # ir_version, producer_name, graph, opset_import
    graph = decode_raw([v for f, _, v in top if f == 7][0])
    nodes = [decode_raw(v) for f, _, v in graph if f == 1]
    assert len(nodes) == 9 and sum(1 for f, _, _ in graph if f == 5) == 6
    assert [bytes(v).decode() for f, _, v in nodes[0] if f == 4] == ["Conv"]
    conv_attrs = {bytes(dict((f, v) for f, _, v in decode_raw(a))[1]).decode() for f, _, a in nodes[0] if f == 5}
    assert conv_attrs == {"kernel_shape", "pads", "strides"}


def test_cpp_loader_parses_written_models(tmp_path):
    from rten_amd import onnx_writer as ow
    from rten_amd.workloads import resnet50
    m, _ = ow.small_cnn_f32()
    p = tmp_path / "small.onnx"
    p.write_bytes(m)
    out = run_cli("--parse-only", str(p))
    assert out.returncode == 0, out.stderr
    assert "9 nodes, 6 initializers" in out.stdout and "Conv x2" in out.stdout and "input  x: f32 [batch, 3, 16, 16]" in out.stdout
    w = resnet50.make_weights()
    p2 = tmp_path / "r50i8.onnx"
    p2.write_bytes(ow.resnet50_int8(w))
    out = run_cli("--parse-only", str(p2))
    assert out.returncode == 0, out.stderr
    assert "ConvInteger x53" in out.stdout and "DynamicQuantizeLinear x50" in out.stdout and "MatMulInteger x1" in out.stdout
    # truncated file: a clean error, not a crash
    p3 = tmp_path / "bad.onnx"
    p3.write_bytes(m[: len(m) // 2])
    out = run_cli("--parse-only", str(p3))
    assert out.returncode == 1 and "onnx:" in out.stderr


def test_no_gpu_means_backend_unavailable_not_a_fallback(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rten_amd import onnx_writer as ow
    p = tmp_path / "small.onnx"
    p.write_bytes(ow.small_cnn_f32()[0])
    out = run_cli(str(p))
    assert out.returncode == 2 and "BackendUnavailable" in out.stderr


def _run_model(tmp_path, model_bytes, x, out_name, *extra):
    p = tmp_path / "m.onnx"
    p.write_bytes(model_bytes)
    xin, yout = tmp_path / "x.bin", tmp_path / "y.bin"
    x.astype(np.float32).tofile(xin)
    r = run_cli("-s", f"batch={x.shape[0]}", "--input", f"x={xin}", "--dump", f"{out_name}={yout}", *extra, str(p))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return np.fromfile(yout, np.float32), r.stdout


@pytest.mark.gpu
def test_small_cnn_graph_bit_exact(tmp_path):
    from oracle import ref
    from rten_amd import onnx_writer as ow
    m, w = ow.small_cnn_f32()
    x = np.random.default_rng(3).random((3, 3, 16, 16), dtype=np.float32) - 0.5
    a = ref.conv2d_f32(x, w["c1"][0], w["c1"][1], pads=(1, 1, 1, 1), strides=(2, 2), relu=True)
    p = ref.max_pool(a, (2, 2), (2, 2))
    s = ref.conv2d_f32(p, w["c2"][0], w["c2"][1], residual=p, relu=True)
    g = ref.global_average_pool(s).reshape(3, -1)
    want = ref.gemm_f32(g, w["fc"][0].T, c=np.broadcast_to(w["fc"][1], (3, 5)).astype(np.float32), alpha=1.0, beta=1.0)
    got, log = _run_model(tmp_path, m, x, "y")
    assert "Plan: 6 steps (3 nodes folded into fused steps)" in log
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32))
    got2, log2 = _run_model(tmp_path, m, x, "y", "--no-fuse")
    assert "Plan: 9 steps (0 nodes folded" in log2
    # unfused: Conv, then Add, then Relu as separate kernels -- the same arithmetic in the same order
    assert np.array_equal(got2.view(np.int32), want.ravel().view(np.int32))


@pytest.mark.gpu
def test_resnet50_f32_onnx_graph_bit_exact(tmp_path):
    from oracle import models as om
    from rten_amd import onnx_writer as ow
    from rten_amd.workloads import resnet50
    w = resnet50.make_weights()
    x = np.random.default_rng(1234).random((2, 3, 224, 224), dtype=np.float32)
    want = om.resnet50_forward(resnet50.conv_specs(), w, x)
    got, log = _run_model(tmp_path, ow.resnet50_f32(w), x, "logits", "-n", "2")
    assert "Plan: 57 steps (65 nodes folded into fused steps)" in log  # 53 convs + maxpool + gap + flatten + gemm
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32))
    # per-layer plan selection by measurement + hipGraph replay: any plan the tuner picks must give the same bits
    got2, log2 = _run_model(tmp_path, ow.resnet50_f32(w), x, "logits", "--tune", "--graph", "-n", "3")
    assert "Tuned the launch plan of 54 convolution / MatMul steps" in log2  # 53 convolutions + the classifier Gemm (MatMul-family steps are tuned since round 5)
    assert "Captured the plan into a hipGraph" in log2 and "Captured the plan into a hipGraph" in log2
    assert np.array_equal(got2.view(np.int32), want.ravel().view(np.int32))


@pytest.mark.gpu
def test_resnet50_int8_onnx_graph_bit_exact(tmp_path):
    from oracle import models as om
    from rten_amd import onnx_writer as ow
    from rten_amd.workloads import resnet50
    w = resnet50.make_weights()
    x = np.random.default_rng(1234).random((2, 3, 224, 224), dtype=np.float32)
    want = om.resnet50_int8_forward(resnet50.conv_specs(), om.quantize_weights_int8(w), x)
    got, log = _run_model(tmp_path, ow.resnet50_int8(w), x, "logits")
    assert "ConvInteger x53" in log and "49 DynamicQuantizeLinear write the staged layout directly" in log
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32))
    # node by node (ConvInteger, Cast, Mul, Add([1,O,1,1] broadcast), Add, Relu as separate kernels): the same arithmetic
    got2, log2 = _run_model(tmp_path, ow.resnet50_int8(w), x, "logits", "--no-fuse")
    assert "Plan: 388 steps (0 nodes folded" in log2
    assert np.array_equal(got2.view(np.int32), want.ravel().view(np.int32))


@pytest.mark.gpu
def test_bert_base_layer_onnx_graph_bit_exact(tmp_path):
    """One BERT-base layer (hidden 768, 12 heads x 64, 128 tokens): the attention pattern runs as the fused single-kernel
    sdpa over the column blocks of one QKV GEMM."""
    from oracle import models as om
    from rten_amd import onnx_writer as ow
    from rten_amd.workloads import bert
    cfg = bert.BertConfig(layers=1, vocab=1000, max_pos=128)
    w = bert.make_weights(cfg)
    B, S = 2, 128
    rng = np.random.default_rng(9)
    ids = rng.integers(0, cfg.vocab, (B, S)).astype(np.int32)
    tts = np.zeros((B, S), np.int32)
    mask = np.ones((B, S), np.int32)
    mask[1, 100:] = 0
    want = om.bert_forward(cfg, w, ids, mask, tts)
    p = tmp_path / "bert.onnx"
    p.write_bytes(ow.bert_encoder(cfg, w, S))
    args = ["-s", f"batch={B}", "--dump", f"last_hidden_state={tmp_path / 'y.bin'}", "-t"]
    for name, arr in (("input_ids", ids), ("token_type_ids", tts), ("attention_mask", mask)):
        arr.tofile(tmp_path / (name + ".bin"))
        args += ["--input", f"{name}={tmp_path / (name + '.bin')}"]
    r = run_cli(*args, str(p))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "MultiHeadSdpa(QKV column blocks)" in r.stdout and "FusedMatMul+Gelu" in r.stdout and "Add+LayerNormalization" in r.stdout
    got = np.fromfile(tmp_path / "y.bin", np.float32)
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32))


@pytest.mark.gpu
def test_bert_encoder_onnx_graph_bit_exact(tmp_path):
    """Transformer graph as an exporter writes it (separate Q/K/V MatMul + Add, Reshape / Transpose around the attention
    MatMuls, Div by sqrt(d), Add(mask) -> Softmax, LayerNormalization, Gelu): bit-identical to the oracle's encoder."""
    from oracle import models as om
    from rten_amd import onnx_writer as ow
    from rten_amd.workloads import bert
    cfg = bert.BertConfig(hidden=64, heads=4, layers=2, ffn=128, vocab=100, max_pos=32, type_vocab=2)
    w = bert.make_weights(cfg)
    B, S = 3, 16
    rng = np.random.default_rng(5)
    ids = rng.integers(0, cfg.vocab, (B, S)).astype(np.int32)
    tts = rng.integers(0, 2, (B, S)).astype(np.int32)
    mask = np.ones((B, S), np.int32)
    mask[1, 11:] = 0
    mask[2, 5:] = 0
    want = om.bert_forward(cfg, w, ids, mask, tts)
    p = tmp_path / "bert.onnx"
    p.write_bytes(ow.bert_encoder(cfg, w, S))
    files = {}
    for name, arr in (("input_ids", ids), ("token_type_ids", tts), ("attention_mask", mask)):
        files[name] = tmp_path / (name + ".bin")
        arr.tofile(files[name])
    yout = tmp_path / "y.bin"
    for extra in ((), ("--no-fuse",), ("--graph", "-n", "2")):
        args = ["-s", f"batch={B}", "--dump", f"last_hidden_state={yout}", *extra]
        for name, f in files.items():
            args += ["--input", f"{name}={f}"]
        r = run_cli(*args, str(p))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        got = np.fromfile(yout, np.float32)
        assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32)), extra


def test_safetensors_reader_writer_roundtrip(tmp_path):
    """The golden-exchange format of `rten-cli --inputs / --check-outputs`: files written by the safetensors package are read
    by the C++ reader, re-written by the C++ writer, and load back identically."""
    from safetensors.numpy import load_file, save_file
    rng = np.random.default_rng(2)
    tensors = {"x": rng.random((2, 3, 4), dtype=np.float32), "ids": rng.integers(-5, 5, (2, 7)).astype(np.int64), "q": rng.integers(0, 255, (5,)).astype(np.uint8),
               "scalar": np.array(3.5, np.float32)}
    src, dst = tmp_path / "a.safetensors", tmp_path / "b.safetensors"
    save_file(tensors, str(src), metadata={"note": "golden \"inputs\""})
    out = run_cli("--safetensors-info", str(src), "--save-outputs", str(dst))
    assert out.returncode == 0, out.stderr
    assert "x: F32 [2, 3, 4]" in out.stdout and "ids: I64 [2, 7]" in out.stdout and "scalar: F32 []" in out.stdout
    back = load_file(str(dst))
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v)
    bad = tmp_path / "bad.safetensors"
    bad.write_bytes(src.read_bytes()[:40])
    assert run_cli("--safetensors-info", str(bad)).returncode == 1


@pytest.mark.gpu
def test_cli_safetensors_inputs_and_check_outputs(tmp_path):
    """`rten_hip_run -i inputs.safetensors --check-outputs expected.safetensors` (the rten-cli golden workflow)."""
    from oracle import ref
    from rten_amd import onnx_writer as ow
    from safetensors.numpy import load_file, save_file
    m, w = ow.small_cnn_f32()
    x = np.random.default_rng(3).random((3, 3, 16, 16), dtype=np.float32) - 0.5
    a = ref.conv2d_f32(x, w["c1"][0], w["c1"][1], pads=(1, 1, 1, 1), strides=(2, 2), relu=True)
    p = ref.max_pool(a, (2, 2), (2, 2))
    s = ref.conv2d_f32(p, w["c2"][0], w["c2"][1], residual=p, relu=True)
    g = ref.global_average_pool(s).reshape(3, -1)
    want = ref.gemm_f32(g, w["fc"][0].T, c=np.broadcast_to(w["fc"][1], (3, 5)).astype(np.float32), alpha=1.0, beta=1.0)
    model, fin, fexp, fout = tmp_path / "m.onnx", tmp_path / "in.safetensors", tmp_path / "exp.safetensors", tmp_path / "out.safetensors"
    model.write_bytes(m)
    save_file({"x": x}, str(fin))
    save_file({"y": want}, str(fexp))
    r = run_cli("-i", str(fin), "--check-outputs", str(fexp), "--max-diff", "0", "--save-outputs", str(fout), str(model))
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'Output "y" vs expected: max diff 0.000000 (0 of 15 elements differ in their bits)' in r.stdout
    assert np.array_equal(load_file(str(fout))["y"], want)
    save_file({"y": want + np.float32(0.01)}, str(fexp))
    r = run_cli("-i", str(fin), "--check-outputs", str(fexp), "--max-diff", "1e-6", str(model))
    assert r.returncode == 3 and "max diff 0.01" in r.stdout


@pytest.mark.gpu
def test_graph_load_errors_and_attribute_forms(tmp_path):
    """An operator outside the registry is a load-time error that names the node (never a CPU fallback); auto_pad = SAME_UPPER,
    AveragePool attributes, Squeeze / Unsqueeze / Reshape views and a graph output that is a view."""
    from oracle import ref
    from rten_amd import onnx_writer as ow
    rng = np.random.default_rng(11)
    w = (rng.random((4, 3, 3, 3), dtype=np.float32) - 0.5)
    bad = ow.model([ow.node("Conv", ["x", "w"], ["a"], name="c0", kernel_shape=[3, 3]), ow.node("FooBar", ["a"], ["y"], name="mystery_node")],
                   [ow.value_info("x", ow.FLOAT, [1, 3, 8, 8])], [ow.value_info("y", ow.FLOAT, [1, 4, 6, 6])], [ow.tensor("w", w)])
    p = tmp_path / "bad.onnx"
    p.write_bytes(bad)
    r = run_cli(str(p))
    assert r.returncode == 1 and "mystery_node" in r.stderr and "FooBar" in r.stderr and "no CPU fallback" in r.stderr

    x = rng.random((2, 3, 9, 7), dtype=np.float32) - 0.5
    b = rng.random(4, dtype=np.float32) - 0.5
    nodes = [ow.node("Conv", ["x", "w", "b"], ["a"], name="conv_same", kernel_shape=[3, 3], auto_pad="SAME_UPPER", strides=[2, 2]),
             ow.node("AveragePool", ["a"], ["p"], name="avg", kernel_shape=[2, 2], strides=[1, 1], pads=[0, 0, 1, 1], count_include_pad=1),
             ow.node("Unsqueeze", ["p", "ax0"], ["p5"], name="unsq"),
             ow.node("Squeeze", ["p5", "ax0"], ["p4"], name="sq"),
             ow.node("Reshape", ["p4", "shape"], ["y"], name="reshape")]
    m = ow.model(nodes, [ow.value_info("x", ow.FLOAT, ["batch", 3, 9, 7])], [ow.value_info("y", ow.FLOAT, ["batch", -1])],
                 [ow.tensor("w", w), ow.tensor("b", b), ow.tensor("ax0", np.array([0], np.int64)), ow.tensor("shape", np.array([0, -1], np.int64))])
    oh, ow_, pads = ref.calc_output_size_and_padding((9, 7), (3, 3), (2, 2), "same")
    a = ref.conv2d_f32(x, w, b, pads=tuple(pads), strides=(2, 2))
    want = ref.average_pool(a, (2, 2), (1, 1), (0, 0, 1, 1), True, False).reshape(2, -1)
    got, log = _run_model(tmp_path, m, x, "y")
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32))


@pytest.mark.gpu
def test_conv_transpose_node_in_a_graph(tmp_path):
    from oracle import ref
    from rten_amd import onnx_writer as ow
    rng = np.random.default_rng(13)
    x = rng.random((2, 4, 6, 5), dtype=np.float32) - 0.5
    w = rng.random((4, 3, 3, 3), dtype=np.float32) - 0.5
    b = rng.random(6, dtype=np.float32) - 0.5
    nodes = [ow.node("ConvTranspose", ["x", "w", "b"], ["u"], name="up", kernel_shape=[3, 3], strides=[2, 2], pads=[1, 1, 1, 1], group=2, output_padding=[1, 1]),
             ow.node("Relu", ["u"], ["y"], name="relu")]
    m = ow.model(nodes, [ow.value_info("x", ow.FLOAT, ["batch", 4, 6, 5])], [ow.value_info("y", ow.FLOAT, ["batch", 6, 12, 10])], [ow.tensor("w", w), ow.tensor("b", b)])
    want = ref.relu(ref.conv_transpose2d_f32(x, w, b, (1, 1, 1, 1), (2, 2), (1, 1), 2, (1, 1)))
    got, _ = _run_model(tmp_path, m, x, "y")
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32))


@pytest.mark.gpu
def test_matmul_nbits_node_in_a_graph(tmp_path):
    # com.microsoft MatMulNBits as ONNX Runtime's 4-bit quantiser emits it: B and scales are initialisers, K / N / bits / block_size attributes
    from oracle import ref
    from rten_amd import onnx_writer as ow
    rng = np.random.default_rng(17)
    w = (rng.standard_normal((64, 24)) * 0.1).astype(np.float32)
    quant, scales = ref.quantize_4bit_blocks(w, 16)
    nodes = [ow.node("MatMulNBits", ["x", "wq", "ws"], ["h"], name="proj", domain="com.microsoft", K=64, N=24, bits=4, block_size=16, accuracy_level=0),
             ow.node("Relu", ["h"], ["y"], name="relu")]
    m = ow.model(nodes, [ow.value_info("x", ow.FLOAT, ["batch", 5, 64])], [ow.value_info("y", ow.FLOAT, ["batch", 5, 24])], [ow.tensor("wq", quant), ow.tensor("ws", scales)])
    x = rng.standard_normal((2, 5, 64)).astype(np.float32)
    got, _ = _run_model(tmp_path, m, x, "y")
    want = ref.relu(ref.matmul_nbits_f32(x, quant, scales))
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32))
    x1 = rng.standard_normal((3, 1, 64)).astype(np.float32)  # rows == 1: the vector path
    m1 = ow.model(nodes, [ow.value_info("x", ow.FLOAT, ["batch", 1, 64])], [ow.value_info("y", ow.FLOAT, ["batch", 1, 24])], [ow.tensor("wq", quant), ow.tensor("ws", scales)])
    got, _ = _run_model(tmp_path, m1, x1, "y")
    want = ref.relu(ref.matmul_nbits_f32(x1, quant, scales))
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32))


def _einsum_attention_model(wk, wv, S, H, D):
    """x [batch, S, H*D] -> Reshape [batch, S, H, D] = q; K / V = Einsum projections of x with per-head weights;
    scores = Einsum(bqhd,bkhd->bhqk) -> Softmax -> Einsum(bhqk,bkhd->bqhd) -> ReduceSum over heads."""
    from rten_amd import onnx_writer as ow
    nodes = [ow.node("Reshape", ["x", "shape4"], ["q"]),
             ow.node("Einsum", ["x", "wk"], ["k"], equation="bsc,chd->bshd"),
             ow.node("Einsum", ["x", "wv"], ["v"], equation="bsc,chd->bshd"),
             ow.node("Einsum", ["q", "k"], ["scores"], equation="bqhd,bkhd->bhqk"),
             ow.node("Softmax", ["scores"], ["probs"], axis=-1),
             ow.node("Einsum", ["probs", "v"], ["ctx"], equation="bhqk,bkhd->bqhd"),
             ow.node("ReduceSum", ["ctx", "axes"], ["y"], keepdims=0)]
    inits = [ow.tensor("shape4", np.array([0, S, H, D], np.int64)), ow.tensor("wk", wk), ow.tensor("wv", wv), ow.tensor("axes", np.array([2], np.int64))]
    return ow.model(nodes, [ow.value_info("x", 1, ["batch", S, H * D])], [ow.value_info("y", 1, ["batch", S, D])], inits)


def test_cpp_loader_parses_einsum_graph(tmp_path):
    rng = np.random.default_rng(2)
    p = tmp_path / "einsum.onnx"
    p.write_bytes(_einsum_attention_model(rng.standard_normal((24, 3, 8)).astype(np.float32), rng.standard_normal((24, 3, 8)).astype(np.float32), 10, 3, 8))
    out = run_cli("--parse-only", str(p))
    assert out.returncode == 0 and "Einsum x4" in out.stdout and "ReduceSum x1" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_einsum_attention_graph_bit_exact(tmp_path):
    """Einsum / ReduceSum nodes from an ONNX file through the C++ executor (include/rten_hip_graph.hpp) against the oracle's
    restatement of the reference's Einsum (oracle/einsum.py) and softmax."""
    from oracle import einsum as OE
    from oracle import ref
    rng = np.random.default_rng(2)
    B, S, H, D = 3, 10, 3, 8
    wk = (rng.standard_normal((H * D, H, D)) * 0.2).astype(np.float32)
    wv = (rng.standard_normal((H * D, H, D)) * 0.2).astype(np.float32)
    x = rng.standard_normal((B, S, H * D)).astype(np.float32)
    q = x.reshape(B, S, H, D)
    k, v = OE.einsum("bsc,chd->bshd", x, wk), OE.einsum("bsc,chd->bshd", x, wv)
    probs = ref.softmax(OE.einsum("bqhd,bkhd->bhqk", q, k))
    want = OE.reduce_sum(OE.einsum("bhqk,bkhd->bqhd", probs, v), [2])
    got, log = _run_model(tmp_path, _einsum_attention_model(wk, wv, S, H, D), x, "y")
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32)), np.abs(got - want.ravel()).max()


def test_cpp_loader_parses_pytorch_exported_graph(tmp_path):
    """Real exporter output (PyTorch's TorchScript ONNX exporter driven without the `onnx` package, tools/torch_export.py):
    BatchNorm folded into Conv by the exporter, Gemm with transB, GlobalAveragePool + Flatten, dynamic batch axis."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import torch_export as te

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c, self.b, self.f = torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.Linear(8, 4)

        def forward(self, x):
            y = torch.relu(self.b(self.c(x)))
            return self.f(torch.flatten(torch.nn.functional.adaptive_avg_pool2d(y, 1), 1))
    p = tmp_path / "torch_small.onnx"
    p.write_bytes(te.export_bytes(Net(), (torch.zeros(2, 3, 8, 8),), ["x"], ["y"], {"x": {0: "batch"}}))
    out = run_cli("--parse-only", str(p))
    assert out.returncode == 0, out.stderr
    assert "producer pytorch" in out.stdout and "Conv x1" in out.stdout and "Gemm x1" in out.stdout and "BatchNormalization" not in out.stdout
    assert "input  x: f32 [batch, 3, 8, 8]" in out.stdout


def test_canonicalisation_of_pytorch_exported_encoder(tmp_path):
    """The load-time canonicalisation (no device needed): PyTorch writes nn.LayerNorm as 9 nodes and nn.GELU as 5, scalars as
    Constant nodes; the reference's LayerNormalizationFusion / GeluFusion patterns turn them back into single operators."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch_export as te
    from rten_amd.workloads import bert
    cfg = bert.BertConfig(hidden=64, heads=4, layers=2, ffn=128, vocab=100, max_pos=32, type_vocab=2)
    p = tmp_path / "encoder_torch.onnx"
    p.write_bytes(te.encoder_onnx(cfg, bert.make_weights(cfg), 3, 16))
    out = run_cli("--parse-only", str(p))
    assert out.returncode == 0, out.stderr
    raw, canon = [l for l in out.stdout.splitlines() if "operators:" in l][0], [l for l in out.stdout.splitlines() if "canonical form" in l][0]
    assert "ReduceMean x10" in raw and "Pow x5" in raw and "Sqrt x5" in raw and "Erf x2" in raw and "Constant x" in raw
    assert "LayerNormalization x5" in canon and "Gelu x2" in canon
    for gone in ("ReduceMean", "Pow", "Sqrt", "Erf", "Constant"):
        assert gone not in canon.split("nodes:")[1], canon
    # MatMul x16 (8 per layer), Softmax x2 and the mask arithmetic are untouched
    assert "MatMul x16" in canon and "Softmax x2" in canon


@pytest.mark.gpu
def test_pytorch_exported_resnet50_bit_exact(tmp_path):
    """ResNet-50 v1.5 written by PyTorch's exporter (122 nodes: Conv x53, Relu x49, Add x16, MaxPool, GlobalAveragePool, Flatten,
    Gemm) from a torch.nn module holding the harness's weights: the C++ loader / executor must give the oracle's logits bit for
    bit, fused and unfused, and agree with torch's own CPU forward to f32 accumulation-order tolerance."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import torch_export as te
    from oracle import models as om
    from rten_amd.workloads import resnet50
    w = resnet50.make_weights()
    x = np.random.default_rng(1234).random((2, 3, 224, 224), dtype=np.float32)
    want = om.resnet50_forward(resnet50.conv_specs(), w, x)
    model = te.resnet50_onnx(w)
    got, log = _run_model(tmp_path, model, x, "logits")
    assert "folded into fused steps" in log
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32)), np.abs(got - want.ravel()).max()
    got2, _ = _run_model(tmp_path, model, x, "logits", "--no-fuse")
    assert np.array_equal(got2.view(np.int32), want.ravel().view(np.int32))
    with torch.no_grad():
        ref_t = te.resnet50_module(w)(torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(got.reshape(ref_t.shape), ref_t, rtol=2e-3, atol=2e-3)


@pytest.mark.gpu
def test_pytorch_exported_transformer_encoder_bit_exact(tmp_path):
    """A BERT-style encoder written by PyTorch's exporter: LayerNorm and GELU arrive decomposed (ReduceMean / Sub / Pow / Sqrt /
    Div, Div / Erf / Add / Mul), scalars as Constant nodes, Linear as MatMul + Add.  The executor applies the reference's
    LayerNormalizationFusion / GeluFusion (optimize/fusions.rs:407-430,674-747) on load, so the result is the oracle's encoder
    (the reference's post-fusion operator order) bit for bit, and agrees with torch's own CPU forward to f32 tolerance."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import torch_export as te
    from oracle import models as om
    from rten_amd.workloads import bert
    cfg = bert.BertConfig(hidden=64, heads=4, layers=2, ffn=128, vocab=100, max_pos=32, type_vocab=2)
    w = bert.make_weights(cfg)
    B, S = 3, 16
    rng = np.random.default_rng(5)
    ids = rng.integers(0, cfg.vocab, (B, S)).astype(np.int32)
    tts = rng.integers(0, 2, (B, S)).astype(np.int32)
    mask = np.ones((B, S), np.int32)
    mask[1, 11:] = 0
    mask[2, 5:] = 0
    want = om.bert_forward(cfg, w, ids, mask, tts)
    p = tmp_path / "encoder_torch.onnx"
    p.write_bytes(te.encoder_onnx(cfg, w, B, S))
    parsed = run_cli("--parse-only", str(p))
    assert "ReduceMean x10" in parsed.stdout and "Erf x2" in parsed.stdout and "Constant x" in parsed.stdout, parsed.stdout
    yout = tmp_path / "y.bin"
    for extra in ((), ("--no-fuse",)):
        args = ["--dump", f"last_hidden_state={yout}", *extra]
        for name, arr in (("input_ids", ids), ("token_type_ids", tts), ("attention_mask", mask)):
            arr.tofile(tmp_path / (name + ".bin"))
            args += ["--input", f"{name}={tmp_path / (name + '.bin')}"]
        r = run_cli(*args, "-t", str(p))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        if not extra:  # the backend's own fusions fire on the exporter's spelling too (static-shape `view`s: Reshape [B, S, h, d])
            assert "MultiHeadSdpa(QKV column blocks)" in r.stdout and "FusedMatMul+Gelu" in r.stdout and "Add+LayerNormalization" in r.stdout, r.stdout[-1500:]
        got = np.fromfile(yout, np.float32)
        assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32)), (extra, np.abs(got - want.ravel()).max())
    with torch.no_grad():
        t = te.encoder_module(cfg, w, S)(torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(mask.astype(np.int64)), torch.from_numpy(tts.astype(np.int64))).numpy()
    np.testing.assert_allclose(got.reshape(t.shape), t, rtol=1e-4, atol=1e-4)


# ---- the same executor behind the C ABI (rten_hip_model_*: csrc/graph_abi.cpp): what a Rust `HipSubgraph` operator binds, and what
#      `bench.py --via-executor` times
def test_model_abi_is_declared_bound_and_refuses_bad_arguments_without_a_gpu():
    import ctypes as C
    from rten_amd import lib as L
    so = L.load()
    for name in ("load", "last_error", "info", "input_name", "output_name", "bind_input", "prepare", "run", "sync", "output", "destroy"):
        assert hasattr(so, "rten_hip_model_" + name)
    out = C.c_void_p()
    assert so.rten_hip_model_load(None, b"x", 1, None, 1, 0, C.byref(out)) == L.ERR_INVALID_VALUE and not out.value  # no context
    assert so.rten_hip_model_run(None, 0) == L.ERR_INVALID_VALUE
    assert so.rten_hip_model_destroy(None) == L.OK
    assert so.rten_hip_model_last_error(None) == b"null graph"


@pytest.mark.gpu
def test_model_abi_chains_and_plan_file_give_the_oracle_bits():
    """ResNet-50 f32 (batch 5 -> chains of 2 + 2 + 1 images, and one chain) through rten_hip_model_*: ONNX bytes in, a launch plan file applied by
    step name, hipGraph replay, outputs assembled on the device -- bit-identical to the CPU oracle, run after run."""
    import ctypes as C
    import json
    from oracle import models as om
    from rten_amd import lib as L, onnx_writer as ow
    from rten_amd.tensor import DeviceTensor
    from rten_amd.workloads import resnet50
    w = resnet50.make_weights()
    x = np.random.default_rng(99).random((5, 3, 224, 224), dtype=np.float32)
    want = om.resnet50_forward(resnet50.conv_specs(), w, x)
    ctx = L.Context(0)
    onnx_bytes = ow.resnet50_f32(w)
    flat = {l["name"]: [3, 0, 1, 0] for l in resnet50.conv_specs()}
    flat["s1b1c2"] = [24, 2, 2, 0]   # a wave-tile split-K plan
    flat["s2b1c2"] = [1, 1, 3, 0]
    keyed = {"2": flat, "1": {k: [3, 0, 1, 1] for k in flat}}
    # "pairs" (round 6): an expand layer and the next block's reduce layer as ONE step / one launch (rten_hip_conv2d_f32_pair).  s1b0c3 is listed although the kernel has
    # no form for its 128 input channels: that step runs its two convolutions one after the other
    paired = dict(flat, pairs=["s0b0c3", "s0b1c3", "s0b2c3", "s1b0c3"])
    # "pair_shortcuts": a listed pair also computes its residual -- the stage's shortcut convolution, which nothing else reads -- in the launch
    # (rten_hip_conv2d_f32_pair_shortcut); s1b0c3's shortcut is a stride-2 layer the kernel has no form for: its three convolutions run one after the other
    short = dict(paired, pair_shortcuts=["s0b0c3", "s1b0c3", "s0b1c3"])  # (s0b1c3's residual is a block output other steps read: it keeps its plain pair)
    for chains, plan in ((3, keyed), (1, flat), (2, None), (1, paired), (3, dict(keyed, pairs=paired["pairs"])), (1, short), (3, dict(keyed, pairs=short["pairs"], pair_shortcuts=short["pair_shortcuts"]))):
        m = L.Model(ctx, onnx_bytes, json.dumps(plan) if plan else None, chains)
        try:
            _check_model(m, ctx, x, want, plan, chains)
        finally:
            m.close()  # chain 0 runs on `ctx`: a model must never outlive its context
    # a plan file that is not JSON, and a batch smaller than the chain count
    with pytest.raises(L.HipError):
        L.Model(ctx, onnx_bytes, "{not json", 1)
    m = L.Model(ctx, onnx_bytes, None, 4)
    try:
        with pytest.raises(L.HipError):
            m.bind_input("x", (2, 3, 224, 224))
    finally:
        m.close()


def _check_model(m, ctx, x, want, plan, chains):
    from rten_amd.tensor import DeviceTensor
    n_pairs = len(plan.get("pairs", [])) if plan else 0
    n_short = 2 if plan and plan.get("pair_shortcuts") else 0  # (of the three listed, two pairs have a shortcut convolution of their own behind their residual)
    assert m.inputs == ["x"] and m.outputs == ["logits"] and m.num_steps == 57 - n_pairs - n_short
    xp = m.bind_input("x", x.shape)
    m.prepare()
    assert m.planned_steps == (53 - n_pairs if plan else 0)  # (a pair counts once, its two layers' own entries are not used; a shortcut inside counts once for its unused entry)
    xt = DeviceTensor(ctx, x.shape, np.float32, ptr=xp, keepalive=m)
    for rep in range(2):
        xt.upload(x if rep == 0 else x[::-1].copy())
        m.run(inputs_written_on_caller_stream=True)
        m.sync()
        optr, oshape = m.output(0)
        assert oshape == (5, 1000)
        got = DeviceTensor(ctx, oshape, np.float32, ptr=optr, keepalive=m).numpy()
        ref_out = want if rep == 0 else want[::-1]
        assert np.array_equal(got.view(np.int32), ref_out.view(np.int32)), (chains, rep)  # (chains of 2 + 2 + 1 images: the lone image keeps the blocked order)


@pytest.mark.gpu
def test_model_abi_int8_quantized_output_edges_from_the_plan_file():
    """The dynamically quantized ResNet-50 through rten_hip_model_* with the committed int8 plan: the listed edges run the consumer's
    DynamicQuantizeLinear inside the producing ConvIntegerToFloat launch (opt-in: without a plan file no such launch happens) -- same logits as
    the oracle, eagerly prepared and replayed, and no launch gave up waiting for its grid."""
    import json
    from oracle import models as om
    from rten_amd import lib as L, onnx_writer as ow
    from rten_amd.tensor import DeviceTensor
    from rten_amd.workloads import resnet50
    w = resnet50.make_weights()
    x = np.random.default_rng(1234).random((4, 3, 224, 224), dtype=np.float32)
    want = om.resnet50_int8_forward(resnet50.conv_specs(), om.quantize_weights_int8(w), x)
    plan = json.load(open(os.path.join(ROOT, "profiles", "plans", "int8.json")))
    assert len(plan["qout"]) >= 10
    onnx_bytes = ow.resnet50_int8(w)
    ctx = L.Context(0)
    # s0b0c3: a block output -- its f32 tensor is kept for the residual Add; s0b2c3 / s2b5c3: stage outputs, read by the next stage's shortcut AND
    # first convolution through ONE quantizer with two scale products (an edge since round 5: the launch folds the first product, the second
    # is the graph's own scalar Mul).  The plan also lists quantize-on-load layers ("fused_dql"): every one of them is either a quantized-output
    # producer itself, fed by one, or shares its quantizer with a shortcut convolution (the staging launch would stay: no loader form) -- none
    # converts under this plan, as in the Python runner; tests/test_gpu_model_baseline.py runs the list on its own.
    more = dict(plan, qout=plan["qout"] + ["s0b0c3"])
    for text, edges in ((json.dumps(more), len(more["qout"])), (None, 0), (json.dumps({"qout": ["s0b2c3", "no_such_node"]}), 1)):
        m = L.Model(ctx, onnx_bytes, text, 1)
        try:
            xp = m.bind_input("x", x.shape)
            m.prepare()
            assert m.planned_steps == edges, (m.planned_steps, edges)
            xt = DeviceTensor(ctx, x.shape, np.float32, ptr=xp, keepalive=m)
            for rep in range(3):
                xt.upload(x if rep != 1 else x[::-1].copy())
                m.run(inputs_written_on_caller_stream=True)
                m.sync()  # (fails on the sticky fault of a launch that timed out)
                optr, oshape = m.output(0)
                got = DeviceTensor(ctx, oshape, np.float32, ptr=optr, keepalive=m).numpy()
                ref_out = want if rep != 1 else want[::-1]
                assert np.array_equal(got.view(np.int32), ref_out.view(np.int32)), (edges, rep)
        finally:
            m.close()
    ctx.close()
