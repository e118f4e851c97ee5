"""CPU-oracle execution of the model graphs (TEST INFRASTRUCTURE ONLY -- see oracle/ref.py).

Runs the ResNet-50 graph layer by layer with the oracle's operators in the order the reference
executor would (Conv+bias, Add, Relu, MaxPool, GlobalAveragePool, Gemm).
"""
from __future__ import annotations

import numpy as np

from . import ref


def resnet50_forward(specs, weights, x, return_activations=False):
    acts = {"x": np.ascontiguousarray(x, np.float32)}
    for i, l in enumerate(specs):
        w, b = weights[l["name"]]
        res = acts[l["res"]] if l["res"] else None
        acts[l["dst"]] = ref.conv2d_f32(acts[l["src"]], w, b, pads=(l["pad"],) * 4, strides=(l["stride"],) * 2,
                                        residual=res, relu=l["relu"])
        if i == 0:
            acts["pool"] = ref.max_pool(acts["stem"], (3, 3), (2, 2), (1, 1, 1, 1))
    last = acts[specs[-1]["dst"]]
    gap = ref.global_average_pool(last).reshape(last.shape[0], -1)
    fc_w, fc_b = weights["fc"]
    # Gemm(transB=1, alpha=1, beta=1, C=bias): output = expand(c); gemm(beta=1)  (matmul.rs:63-82)
    c0 = np.broadcast_to(fc_b, (gap.shape[0], fc_w.shape[0])).astype(np.float32)
    logits = ref.gemm_f32(gap, fc_w.T, c=c0, alpha=1.0, beta=1.0)
    return (logits, acts) if return_activations else logits


def bert_forward(cfg, w, input_ids, attention_mask, token_type_ids, lanes=ref.LANES):
    """BERT encoder with the oracle's operators in the reference's post-fusion op order."""
    ids = np.asarray(input_ids).reshape(-1)
    tts = np.asarray(token_type_ids).reshape(-1)
    B, S = np.asarray(input_ids).shape
    H, nh = cfg.hidden, cfg.heads
    dh = H // nh
    m = np.asarray(attention_mask, np.float32)
    mask = ((np.float32(1.0) - m) * np.finfo(np.float32).min).reshape(B, 1, 1, S).astype(np.float32)
    x = ref.add(w["word"][ids], w["type"][tts])
    x = ref.add(x, w["pos"][:S])                     # broadcast over the batch (period S*H)
    x = ref.layer_norm(x, w["emb_ln_g"], w["emb_ln_b"], eps=cfg.eps, lanes=lanes)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(dh)))
    for lw in w["layers"]:
        q = ref.matmul_f32(x, lw["wq"], bias=lw["bq"]).reshape(B, S, nh, dh).transpose(0, 2, 1, 3)
        k = ref.matmul_f32(x, lw["wk"], bias=lw["bk"]).reshape(B, S, nh, dh).transpose(0, 2, 1, 3)
        v = ref.matmul_f32(x, lw["wv"], bias=lw["bv"]).reshape(B, S, nh, dh).transpose(0, 2, 1, 3)
        # FusedMatMul(alpha) -> AddSoftmax (no NaN flush) -> MatMul
        att = ref.sdpa(q, k, v, mask=mask, scale=scale, lanes=lanes, flush_nan=False)
        att = np.ascontiguousarray(att.transpose(0, 2, 1, 3)).reshape(B * S, H)
        y = ref.add(ref.matmul_f32(att, lw["wo"], bias=lw["bo"]), x)
        x = ref.layer_norm(y, lw["ln1_g"], lw["ln1_b"], eps=cfg.eps, lanes=lanes)
        h = ref.gelu(ref.matmul_f32(x, lw["w1"], bias=lw["b1"]))
        y = ref.add(ref.matmul_f32(h, lw["w2"], bias=lw["b2"]), x)
        x = ref.layer_norm(y, lw["ln2_g"], lw["ln2_b"], eps=cfg.eps, lanes=lanes)
    return x


def quantize_weights_int8(weights):
    """Symmetric per-tensor i8 weights for the dynamically-quantized graphs: {name: (wq i8, w_scale f32, bias f32)}."""
    out = {}
    for name, (w, b) in weights.items():
        s = np.float32(np.abs(w).max() / 64.0)  # reduce_range=True: 7-bit weights (tools/ort-quantize.py:124-137)
        q = np.clip(np.rint(w / s), -64, 64).astype(np.int8)
        out[name] = (q, s, b)
    return out


def resnet50_int8_forward(specs, qweights, x, pad_mode=ref.PAD_RAW0_I8):
    """ResNet-50 as an ort-dynamically-quantized graph (BASELINE configs[2]): every Conv becomes
    DynamicQuantizeLinear -> ConvInteger -> Cast -> Mul(x_scale * w_scale) -> Add(bias) [-> Add(residual)] [-> Relu]
    (the ConvIntegerToFloat fusion of the reference, src/ops/conv.rs:495-587), the classifier
    DynamicQuantizeLinear -> MatMulInteger -> Cast -> Mul -> Add."""
    acts = {"x": np.ascontiguousarray(x, np.float32)}
    for i, l in enumerate(specs):
        wq, ws, b = qweights[l["name"]]
        q, s, z = ref.dynamic_quantize_linear(acts[l["src"]])
        acc = ref.conv2d_int8(q, wq, x_zp=int(z), pads=(l["pad"],) * 4, strides=(l["stride"],) * 2, pad_mode=pad_mode)
        f = ref.cast_scale(acc, np.float32(np.float32(s) * np.float32(ws)))
        f = ref.add(f, np.ascontiguousarray(np.broadcast_to(b[None, :, None, None], f.shape)))
        if l["res"]:
            f = ref.add(f, acts[l["res"]])
        if l["relu"]:
            f = ref.relu(f)
        acts[l["dst"]] = f
        if i == 0:
            acts["pool"] = ref.max_pool(acts["stem"], (3, 3), (2, 2), (1, 1, 1, 1))
    last = acts[specs[-1]["dst"]]
    gap = ref.global_average_pool(last).reshape(last.shape[0], -1)
    wq, ws, b = qweights["fc"]
    q, s, z = ref.dynamic_quantize_linear(gap)
    acc = ref.gemm_int8(q, np.ascontiguousarray(wq.T), np.array(z, np.uint8), None)
    f = ref.cast_scale(acc, np.float32(np.float32(s) * np.float32(ws)))
    return ref.add(f, np.ascontiguousarray(np.broadcast_to(b[None, :], f.shape)))
