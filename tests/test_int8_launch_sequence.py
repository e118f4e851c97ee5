"""The int8 ResNet-50 runner's launch sequence, checked on the CPU with a recording context (host logic only: no kernel runs).

What a forward pass enqueues is decided by host-side bookkeeping -- which quantized tensor is staged where, which convolution finds its
input already quantized by its producer, which Mul(x_scale, w_scale) came out of the quantizer's launch -- and a slip there shows up on
the GPU as a wrong logit, far from its cause.  This pins the sequence itself.
"""
import collections
import ctypes as C

import numpy as np
import pytest

from rten_amd import lib as L
from rten_amd.workloads import resnet50, resnet50_int8


from tests.recording_ctx import RecordingCtx  # noqa: E402  (shared with the world-8 bench control-flow test)


@pytest.fixture(scope="module")
def net():
    ctx = RecordingCtx()
    rng = np.random.default_rng(0)
    weights = {}
    for l in resnet50.conv_specs():
        weights[l["name"]] = ((rng.standard_normal((l["cout"], l["cin"], l["k"], l["k"])) * 0.05).astype(np.float32), np.zeros(l["cout"], np.float32))
    weights["fc"] = ((rng.standard_normal((1000, 2048)) * 0.01).astype(np.float32), np.zeros(1000, np.float32))
    n = resnet50_int8.ResNet50Int8(ctx, 32, weights)
    return n


def _forward(net, **flags):
    for k, v in flags.items():
        setattr(net, k, v)
    del net.ctx.log[:]
    net.forward()
    return collections.Counter(net.ctx.log)


def test_readers_of_one_quantized_tensor_are_found(net):
    # the max-pool's output and the first block input of stages 1-3 are read by the projection shortcut and by c1
    assert net.shared_next == {"s0b0ds": "s0b0c1", "s1b0ds": "s1b0c1", "s2b0ds": "s2b0c1", "s3b0ds": "s3b0c1"}
    assert "pool" in net.stats and len(net.qout_next) == 47 and len(net.qout_keeps_f32) == 12


def test_plain_sequence_one_quantizer_and_one_conv_per_layer(net):
    c = _forward(net, fused_qout=False, fused_dql=False)
    assert c["rten_hip_conv2d_int8_stats"] == 53 and c["rten_hip_conv2d_int8_qout"] == 0
    # 49 quantizers for 53 convolutions (4 tensors have two readers); the stem's input has no producer statistics
    assert c["rten_hip_dynamic_quantize_linear_staged_stats"] == 48 and c["rten_hip_dynamic_quantize_linear_staged"] == 1
    assert c["rten_hip_mul_f32"] == 4 + 1  # the second reader's Mul(x_scale, w_scale) x 4, the classifier's
    assert c["rten_hip_max_pool2d_f32_stats"] == 1 and c["rten_hip_max_pool2d_f32"] == 0
    assert c["rten_hip_minmax_stats_reset"] == 1 and c["rten_hip_gemm_int8"] == 1


def test_quantized_output_sequence_folds_the_scale_products(net):
    net._qout_off = set()
    c = _forward(net, fused_qout=True, fused_dql=False)
    # every single-quantizer edge runs the one-launch form (the recording library accepts all of them); their consumers stage nothing
    assert c["rten_hip_conv2d_int8_qout"] == 47 and c["rten_hip_conv2d_int8_stats"] == 53 - 47
    assert c["rten_hip_dynamic_quantize_linear_staged_products"] == 1  # the max-pool's output: no convolution produces it
    staged = c["rten_hip_dynamic_quantize_linear_staged_stats"] + c["rten_hip_dynamic_quantize_linear_staged"] + c["rten_hip_dynamic_quantize_linear_staged_products"]
    assert staged == 2  # the image and the max-pool's output: everything else arrives quantized from its producer
    assert c["rten_hip_mul_f32"] == 3 + 1  # stages 1-3: the shared tensor is staged by a qout launch (one product); + the classifier's
    assert not net._sc_for


def test_committed_plan_sequence(net):
    import json
    import os
    plan = json.load(open(os.path.join(os.path.dirname(__file__), "..", "profiles", "plans", "int8.json")))
    net._qout_off = {n for n in net.qout_next if n not in set(plan["qout"])}
    net.fused_layers = set(plan["fused_dql"])
    c = _forward(net, fused_qout=True, fused_dql=True)
    q = len(plan["qout"])
    assert c["rten_hip_conv2d_int8_qout"] == q
    loader = c["rten_hip_conv2d_int8_dql"]
    assert c["rten_hip_conv2d_int8_stats"] + loader + q == 53
    # (a second reader finds its input staged by the first: nothing left for its loader to quantize)
    assert loader == len([n for n in plan["fused_dql"] if n not in plan["qout"] and n not in net.shared_next.values()])
    folds = c["rten_hip_dynamic_quantize_linear_staged_products"]
    assert folds >= 1 and c["rten_hip_mul_f32"] == (4 - folds) + 1
    launches = sum(v for k, v in c.items() if not k.startswith("rten_hip_set_"))
    assert launches <= 100, c  # (107 at the end of round 2)
    net._qout_off, net.fused_layers = set(), None


def test_folds_can_be_switched_off_for_an_ab(monkeypatch, net):
    monkeypatch.setenv("RTEN_INT8_NO_FOLD", "1")
    n2 = resnet50_int8.ResNet50Int8(RecordingCtx(), 8, net.weights)
    assert "pool" not in n2.stats and not n2.fold_products
    c = _forward(n2, fused_qout=True, fused_dql=False)
    assert c["rten_hip_max_pool2d_f32"] == 1 and c["rten_hip_dynamic_quantize_linear_staged_products"] == 0
