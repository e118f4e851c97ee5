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
