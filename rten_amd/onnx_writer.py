"""Minimal ONNX writer (protobuf wire format by hand) -- SURVEY 8(f) rank 3, "model tooling without ORT".

There is no `onnx` / `onnxruntime` package on the GPU boxes, so the named configs are manufactured here: graphs are
serialised directly in the ONNX protobuf encoding (onnx.proto3 field numbers quoted below) and read back by the C++
loader in include/rten_hip_graph.hpp (the backend's counterpart of rten-onnx/src/onnx.rs + src/model/onnx_loader.rs).

Also restated here: the weight side of ort's `quantize_dynamic(..., reduce_range=True)` as RTen's tools/ort-quantize.py
drives it (tools/ort-quantize.py:100-151) -- per-tensor symmetric 7-bit weights, DynamicQuantizeLinear on activations,
ConvInteger / MatMulInteger -> Cast -> Mul(x_scale * w_scale) -> Add(bias).
"""
from __future__ import annotations

import struct

import numpy as np

# TensorProto.DataType
FLOAT, UINT8, INT8, INT32, INT64 = 1, 2, 3, 6, 7
_NP2ONNX = {np.dtype(np.float32): FLOAT, np.dtype(np.uint8): UINT8, np.dtype(np.int8): INT8, np.dtype(np.int32): INT32, np.dtype(np.int64): INT64}


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wire: int) -> bytes:
    return _varint((field << 3) | wire)


def _ld(field: int, payload: bytes) -> bytes:  # length-delimited
    return _key(field, 2) + _varint(len(payload)) + payload


def _vi(field: int, v: int) -> bytes:
    return _key(field, 0) + _varint(int(v))


def _str(field: int, s: str) -> bytes:
    return _ld(field, s.encode())


def tensor(name: str, arr) -> bytes:
    """TensorProto: dims = 1, data_type = 2, name = 8, raw_data = 9."""
    a = np.asarray(arr)
    if a.ndim and not a.flags.c_contiguous:  # (ascontiguousarray would turn a 0-d scalar into shape [1])
        a = np.ascontiguousarray(a)
    out = b"".join(_vi(1, d) for d in a.shape)
    out += _vi(2, _NP2ONNX[a.dtype]) + _str(8, name) + _ld(9, a.tobytes())
    return out


def attr(name: str, value) -> bytes:
    """AttributeProto: name = 1, f = 2, i = 3, s = 4, t = 5, floats = 7, ints = 8, type = 20
    (AttributeType FLOAT = 1, INT = 2, STRING = 3, TENSOR = 4, FLOATS = 6, INTS = 7)."""
    out = _str(1, name)
    if isinstance(value, bool) or isinstance(value, (int, np.integer)):
        return out + _vi(3, int(value)) + _vi(20, 2)
    if isinstance(value, (float, np.floating)):
        return out + _key(2, 5) + struct.pack("<f", float(value)) + _vi(20, 1)
    if isinstance(value, (bytes, str)):
        return out + _ld(4, value.encode() if isinstance(value, str) else value) + _vi(20, 3)
    if isinstance(value, np.ndarray):
        return out + _ld(5, tensor("", value)) + _vi(20, 4)
    value = list(value)
    if all(isinstance(v, (int, np.integer)) for v in value):
        return out + b"".join(_vi(8, int(v)) for v in value) + _vi(20, 7)
    return out + b"".join(_key(7, 5) + struct.pack("<f", float(v)) for v in value) + _vi(20, 6)


def node(op_type: str, inputs, outputs, name: str = "", domain: str = "", **attrs) -> bytes:
    """NodeProto: input = 1, output = 2, name = 3, op_type = 4, attribute = 5, domain = 7."""
    out = b"".join(_str(1, i) for i in inputs) + b"".join(_str(2, o) for o in outputs)
    if name:
        out += _str(3, name)
    out += _str(4, op_type)
    out += b"".join(_ld(5, attr(k, v)) for k, v in attrs.items())
    if domain:
        out += _str(7, domain)
    return out


def value_info(name: str, elem_type: int, shape) -> bytes:
    """ValueInfoProto: name = 1, type = 2 { tensor_type = 1 { elem_type = 1, shape = 2 { dim = 1 { dim_value = 1 | dim_param = 2 } } } }."""
    dims = b"".join(_ld(1, _str(2, d) if isinstance(d, str) else _vi(1, d)) for d in shape)
    ttype = _vi(1, elem_type) + _ld(2, dims)
    return _str(1, name) + _ld(2, _ld(1, ttype))


def model(nodes, inputs, outputs, initializers, opset: int = 17, name: str = "graph", producer: str = "rten_amd.onnx_writer") -> bytes:
    """ModelProto: ir_version = 1, producer_name = 2, graph = 7, opset_import = 8 { domain = 1, version = 2 };
    GraphProto: node = 1, name = 2, initializer = 5, input = 11, output = 12."""
    g = b"".join(_ld(1, n) for n in nodes) + _str(2, name) + b"".join(_ld(5, t) for t in initializers)
    g += b"".join(_ld(11, v) for v in inputs) + b"".join(_ld(12, v) for v in outputs)
    return _vi(1, 8) + _str(2, producer) + _ld(7, g) + _ld(8, _str(1, "") + _vi(2, opset))


# ----------------------------------------------------------------------------------------------------------------
# ResNet-50 v1.5 (BASELINE configs[0..2]) as an exporter would write it: BN folded, Conv / Relu / Add as separate nodes
# ----------------------------------------------------------------------------------------------------------------

def resnet50_f32(weights, batch="batch", image: int = 224) -> bytes:
    from .workloads.resnet50 import conv_specs
    nodes, inits = [], []
    for l in conv_specs():
        w, b = weights[l["name"]]
        inits += [tensor(l["name"] + ".w", w), tensor(l["name"] + ".b", b)]
        conv_out = l["dst"] + ".conv" if (l["relu"] or l["res"]) else l["dst"]
        nodes.append(node("Conv", [l["src"], l["name"] + ".w", l["name"] + ".b"], [conv_out], name=l["name"], dilations=[1, 1], group=1,
                          kernel_shape=[l["k"], l["k"]], pads=[l["pad"]] * 4, strides=[l["stride"]] * 2))
        cur = conv_out
        if l["res"]:
            nxt = l["dst"] + ".sum" if l["relu"] else l["dst"]
            nodes.append(node("Add", [cur, l["res"]], [nxt], name=l["name"] + ".add"))
            cur = nxt
        if l["relu"]:
            nodes.append(node("Relu", [cur], [l["dst"]], name=l["name"] + ".relu"))
        if l["name"] == "stem":
            nodes.append(node("MaxPool", ["stem"], ["pool"], name="maxpool", ceil_mode=0, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2]))
    last = conv_specs()[-1]["dst"]
    fw, fb = weights["fc"]
    inits += [tensor("fc.w", fw), tensor("fc.b", fb)]
    nodes.append(node("GlobalAveragePool", [last], ["gap"], name="gap"))
    nodes.append(node("Flatten", ["gap"], ["flat"], name="flatten", axis=1))
    nodes.append(node("Gemm", ["flat", "fc.w", "fc.b"], ["logits"], name="fc", alpha=1.0, beta=1.0, transB=1))
    return model(nodes, [value_info("x", FLOAT, [batch, 3, image, image])], [value_info("logits", FLOAT, [batch, fw.shape[0]])], inits, name="resnet50")


def quantize_weight_reduce_range(w):
    """Per-tensor symmetric weight quantisation of ort's dynamic quantiser with reduce_range=True (7-bit, [-64, 64]):
    scale = max|w| / 64, q = clip(rint(w / scale)), zero point 0 (tools/ort-quantize.py:124-137)."""
    s = np.float32(np.abs(w).max() / 64.0)
    return np.clip(np.rint(w / s), -64, 64).astype(np.int8), s


def resnet50_int8(weights, batch="batch", image: int = 224) -> bytes:
    """The same network after dynamic quantisation: per Conv
        DynamicQuantizeLinear(x) -> ConvInteger(xq, wq, x_zp, w_zp) -> Cast(FLOAT) -> Mul(Mul(x_scale, w_scale)) -> Add(bias [1,O,1,1])
    one DynamicQuantizeLinear per distinct input tensor (the quantiser caches quantised inputs), bias as a separate Add
    (SURVEY 8d config 3), the classifier as MatMulInteger."""
    from .workloads.resnet50 import conv_specs
    nodes, inits, quantized = [], [], set()

    def dql(src):
        if src not in quantized:
            nodes.append(node("DynamicQuantizeLinear", [src], [src + ".q", src + ".scale", src + ".zp"], name=src + ".dql"))
            quantized.add(src)
        return src + ".q", src + ".scale", src + ".zp"

    inits.append(tensor("zero_i8", np.zeros((), np.int8)))
    for l in conv_specs():
        w, b = weights[l["name"]]
        wq, ws = quantize_weight_reduce_range(w)
        n = l["name"]
        inits += [tensor(n + ".wq", wq), tensor(n + ".ws", np.array(ws, np.float32)), tensor(n + ".b", b.reshape(1, -1, 1, 1))]
        xq, xs, xz = dql(l["src"])
        nodes.append(node("ConvInteger", [xq, n + ".wq", xz, "zero_i8"], [n + ".acc"], name=n, dilations=[1, 1], group=1, kernel_shape=[l["k"], l["k"]],
                          pads=[l["pad"]] * 4, strides=[l["stride"]] * 2))
        nodes.append(node("Cast", [n + ".acc"], [n + ".accf"], name=n + ".cast", to=FLOAT))
        nodes.append(node("Mul", [xs, n + ".ws"], [n + ".scale"], name=n + ".scale_mul"))
        nodes.append(node("Mul", [n + ".accf", n + ".scale"], [n + ".scaled"], name=n + ".mul"))
        cur = l["dst"] + ".biased" if (l["relu"] or l["res"]) else l["dst"]
        nodes.append(node("Add", [n + ".scaled", n + ".b"], [cur], name=n + ".bias"))
        if l["res"]:
            nxt = l["dst"] + ".sum" if l["relu"] else l["dst"]
            nodes.append(node("Add", [cur, l["res"]], [nxt], name=n + ".add"))
            cur = nxt
        if l["relu"]:
            nodes.append(node("Relu", [cur], [l["dst"]], name=n + ".relu"))
        if n == "stem":
            nodes.append(node("MaxPool", ["stem"], ["pool"], name="maxpool", ceil_mode=0, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2]))
    last = conv_specs()[-1]["dst"]
    fw, fb = weights["fc"]
    fq, fs = quantize_weight_reduce_range(fw)
    inits += [tensor("fc.wq", np.ascontiguousarray(fq.T)), tensor("fc.ws", np.array(fs, np.float32)), tensor("fc.b", fb)]
    nodes.append(node("GlobalAveragePool", [last], ["gap"], name="gap"))
    nodes.append(node("Flatten", ["gap"], ["flat"], name="flatten", axis=1))
    xq, xs, xz = dql("flat")
    nodes.append(node("MatMulInteger", [xq, "fc.wq", xz, "zero_i8"], ["fc.acc"], name="fc"))
    nodes.append(node("Cast", ["fc.acc"], ["fc.accf"], name="fc.cast", to=FLOAT))
    nodes.append(node("Mul", [xs, "fc.ws"], ["fc.scale"], name="fc.scale_mul"))
    nodes.append(node("Mul", ["fc.accf", "fc.scale"], ["fc.scaled"], name="fc.mul"))
    nodes.append(node("Add", ["fc.scaled", "fc.b"], ["logits"], name="fc.bias"))
    return model(nodes, [value_info("x", FLOAT, [batch, 3, image, image])], [value_info("logits", FLOAT, [batch, fw.shape[0]])], inits, name="resnet50_int8")


def small_cnn_f32(seed: int = 7):
    """A few-layer CNN with every node kind of the ResNet graph (tests): returns (model bytes, weights dict)."""
    rng = np.random.default_rng(seed)
    w = {"c1": (rng.normal(0, 0.3, (8, 3, 3, 3)).astype(np.float32), rng.normal(0, 0.1, 8).astype(np.float32)),
         "c2": (rng.normal(0, 0.2, (8, 8, 1, 1)).astype(np.float32), rng.normal(0, 0.1, 8).astype(np.float32)),
         "fc": (rng.normal(0, 0.3, (5, 8)).astype(np.float32), rng.normal(0, 0.1, 5).astype(np.float32))}
    inits = [tensor("c1.w", w["c1"][0]), tensor("c1.b", w["c1"][1]), tensor("c2.w", w["c2"][0]), tensor("c2.b", w["c2"][1]),
             tensor("fc.w", w["fc"][0]), tensor("fc.b", w["fc"][1])]
    nodes = [node("Conv", ["x", "c1.w", "c1.b"], ["a"], name="c1", kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2]),
             node("Relu", ["a"], ["a.r"], name="c1.relu"),
             node("MaxPool", ["a.r"], ["p"], name="pool", kernel_shape=[2, 2], strides=[2, 2]),
             node("Conv", ["p", "c2.w", "c2.b"], ["b"], name="c2", kernel_shape=[1, 1]),
             node("Add", ["b", "p"], ["s"], name="add"),
             node("Relu", ["s"], ["s.r"], name="relu"),
             node("GlobalAveragePool", ["s.r"], ["g"], name="gap"),
             node("Flatten", ["g"], ["f"], name="flatten", axis=1),
             node("Gemm", ["f", "fc.w", "fc.b"], ["y"], name="fc", transB=1)]
    return model(nodes, [value_info("x", FLOAT, ["batch", 3, 16, 16])], [value_info("y", FLOAT, ["batch", 5])], inits, name="small_cnn"), w


# ----------------------------------------------------------------------------------------------------------------
# BERT encoder (BASELINE configs[3]) in the shape an exporter gives it: separate Q / K / V projections, Reshape /
# Transpose around the attention MatMuls, Div by sqrt(d), Add(mask) -> Softmax, LayerNormalization (opset 17), Gelu
# ----------------------------------------------------------------------------------------------------------------

def bert_encoder(cfg, weights, seq: int, batch="batch") -> bytes:
    """inputs: input_ids, token_type_ids, attention_mask -- int64 [batch, seq]; output: last_hidden_state [batch, seq, hidden]."""
    H, nh = cfg.hidden, cfg.heads
    dh = H // nh
    w = weights
    nodes, inits = [], []

    def const(name, arr):
        inits.append(tensor(name, arr))
        return name

    const("word", w["word"]); const("type", w["type"]); const("pos", np.ascontiguousarray(w["pos"][:seq]))
    const("emb_ln_g", w["emb_ln_g"]); const("emb_ln_b", w["emb_ln_b"])
    const("one", np.array(1.0, np.float32)); const("f32_min", np.array(np.finfo(np.float32).min, np.float32))
    const("sqrt_dh", np.array(np.sqrt(np.float32(dh)), np.float32))
    const("mask_axes", np.array([1, 2], np.int64))
    const("split_heads", np.array([0, 0, nh, dh], np.int64)); const("merge_heads", np.array([0, 0, H], np.int64))
    # embeddings
    nodes.append(node("Gather", ["word", "input_ids"], ["emb.word"], name="emb.word", axis=0))
    nodes.append(node("Gather", ["type", "token_type_ids"], ["emb.type"], name="emb.type", axis=0))
    nodes.append(node("Add", ["emb.word", "emb.type"], ["emb.wt"], name="emb.add_type"))
    nodes.append(node("Add", ["emb.wt", "pos"], ["emb.sum"], name="emb.add_pos"))
    nodes.append(node("LayerNormalization", ["emb.sum", "emb_ln_g", "emb_ln_b"], ["x0"], name="emb.ln", axis=-1, epsilon=float(cfg.eps)))
    # additive attention mask: (1 - mask) * finfo(f32).min as [B, 1, 1, S]
    nodes.append(node("Unsqueeze", ["attention_mask", "mask_axes"], ["mask.4d"], name="mask.unsqueeze"))
    nodes.append(node("Cast", ["mask.4d"], ["mask.f"], name="mask.cast", to=FLOAT))
    nodes.append(node("Sub", ["one", "mask.f"], ["mask.inv"], name="mask.sub"))
    nodes.append(node("Mul", ["mask.inv", "f32_min"], ["mask.bias"], name="mask.mul"))
    x = "x0"
    for i, lw in enumerate(w["layers"]):
        p = f"l{i}."
        for k in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b"):
            const(p + k, lw[k])
        for t in ("q", "k", "v"):
            nodes.append(node("MatMul", [x, p + "w" + t], [p + t + ".mm"], name=p + t + ".matmul"))
            nodes.append(node("Add", [p + t + ".mm", p + "b" + t], [p + t + ".lin"], name=p + t + ".bias"))
            nodes.append(node("Reshape", [p + t + ".lin", "split_heads"], [p + t + ".4d"], name=p + t + ".reshape"))
            nodes.append(node("Transpose", [p + t + ".4d"], [p + t], name=p + t + ".transpose", perm=[0, 2, 3, 1] if t == "k" else [0, 2, 1, 3]))
        nodes.append(node("MatMul", [p + "q", p + "k"], [p + "scores.raw"], name=p + "qk"))
        nodes.append(node("Div", [p + "scores.raw", "sqrt_dh"], [p + "scores"], name=p + "scale"))
        nodes.append(node("Add", [p + "scores", "mask.bias"], [p + "scores.masked"], name=p + "mask"))
        nodes.append(node("Softmax", [p + "scores.masked"], [p + "probs"], name=p + "softmax", axis=-1))
        nodes.append(node("MatMul", [p + "probs", p + "v"], [p + "ctx.h"], name=p + "pv"))
        nodes.append(node("Transpose", [p + "ctx.h"], [p + "ctx.t"], name=p + "ctx.transpose", perm=[0, 2, 1, 3]))
        nodes.append(node("Reshape", [p + "ctx.t", "merge_heads"], [p + "ctx"], name=p + "ctx.reshape"))
        nodes.append(node("MatMul", [p + "ctx", p + "wo"], [p + "o.mm"], name=p + "o.matmul"))
        nodes.append(node("Add", [p + "o.mm", p + "bo"], [p + "o.lin"], name=p + "o.bias"))
        nodes.append(node("Add", [p + "o.lin", x], [p + "res1"], name=p + "res1"))
        nodes.append(node("LayerNormalization", [p + "res1", p + "ln1_g", p + "ln1_b"], [p + "x1"], name=p + "ln1", axis=-1, epsilon=float(cfg.eps)))
        nodes.append(node("MatMul", [p + "x1", p + "w1"], [p + "h.mm"], name=p + "ffn1.matmul"))
        nodes.append(node("Add", [p + "h.mm", p + "b1"], [p + "h.lin"], name=p + "ffn1.bias"))
        nodes.append(node("Gelu", [p + "h.lin"], [p + "h"], name=p + "gelu"))
        nodes.append(node("MatMul", [p + "h", p + "w2"], [p + "f.mm"], name=p + "ffn2.matmul"))
        nodes.append(node("Add", [p + "f.mm", p + "b2"], [p + "f.lin"], name=p + "ffn2.bias"))
        nodes.append(node("Add", [p + "f.lin", p + "x1"], [p + "res2"], name=p + "res2"))
        out = "last_hidden_state" if i == len(w["layers"]) - 1 else p + "x2"
        nodes.append(node("LayerNormalization", [p + "res2", p + "ln2_g", p + "ln2_b"], [out], name=p + "ln2", axis=-1, epsilon=float(cfg.eps)))
        x = out
    ins = [value_info(n, INT64, [batch, seq]) for n in ("input_ids", "token_type_ids", "attention_mask")]
    return model(nodes, ins, [value_info("last_hidden_state", FLOAT, [batch, seq, H])], inits, opset=20, name="bert_encoder")
