"""The PRODUCT path at BASELINE size: rten_hip_model_* -- ONNX bytes in, the plan files committed under
profiles/plans/, sub-batch chains, hipGraph replay -- against the CPU oracle, bit for bit.  This is the path `python bench.py` times."""
import json
import os

import numpy as np
import pytest

from tests import baseline_oracle as bo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(name):
    return open(os.path.join(ROOT, "profiles", "plans", name)).read()


def _bits_equal(got, want, what):
    assert got.shape == want.shape, (what, got.shape, want.shape)
    same = got.view(np.int32) == want.view(np.int32)
    assert same.all(), f"{what}: {(~same).sum()} of {same.size} values differ, first at {tuple(np.argwhere(~same)[0])}"


def _run(model, ctx, feeds, reps=2):
    from rten_amd.tensor import DeviceTensor
    outs = []
    for name, arr in feeds.items():
        t = DeviceTensor(ctx, arr.shape, arr.dtype, ptr=model.input_ptrs[name], keepalive=model)
        t.upload(arr)
    for _ in range(reps):
        model.run(inputs_written_on_caller_stream=True)
        model.sync()
        optr, oshape = model.output(0)
        outs.append(DeviceTensor(ctx, oshape, np.float32, ptr=optr, keepalive=model).numpy())
    return outs


def test_resnet50_f32_batch32_four_chains_committed_plan_is_the_oracle():
    """BASELINE configs[1] exactly as bench.py runs it: 32 images as 4 chains of 8 under profiles/plans/f32_4chains.json; also one chain under
    f32_1chain.json.  Every image's logits are the oracle's, replay after replay."""
    from rten_amd import lib as L, onnx_writer as ow
    want = bo.resnet50_f32_logits()
    x = bo.resnet_input()
    onnx_bytes = ow.resnet50_f32(bo.resnet_weights())
    ctx = L.Context(0)
    try:
        # (1, "f32_lanes.json") is the plan of bench.py's DEFAULT line (one chain per replica, two replicas): the one-chain plan + the classifier entry
        for chains, plan in ((4, "f32_4chains.json"), (1, "f32_1chain.json"), (1, "f32_lanes.json")):
            m = L.Model(ctx, onnx_bytes, _plan(plan), chains)
            try:
                m.bind_input("x", x.shape)
                m.prepare()
                # every convolution took its entry of the committed plan (f32_lanes.json: + the classifier Gemm's)
                # (a pair of convolutions that runs as ONE launch -- the plan's "pairs" -- counts once: its two layers' own entries are not used)
                n_pairs = len(json.loads(_plan(plan)).get("pairs", []))
                assert (plan == "f32_1chain.json") == (n_pairs == 0)  # the lanes plan and the four-chain plan run stage 0's three expand -> reduce pairs in one launch each
                assert m.planned_steps == (54 if plan == "f32_lanes.json" else 53) - n_pairs, (plan, m.planned_steps)
                assert sorted(json.loads(m.plan_json()).get("pairs", [])) == sorted(json.loads(_plan(plan)).get("pairs", []))  # the exported plan names the pairs it was built with
                assert m.warning == "", m.warning
                ptr, nbytes = m.weight_arena()
                assert ptr and nbytes > 100 << 20  # 25.5 M f32 parameters + their prepacked images, one allocation
                for got in _run(m, ctx, {"x": x}):
                    _bits_equal(got, want, f"f32 {chains} chain(s)")
                if chains == 1:
                    # a REPLICA on a second context (rten_hip_model_clone: own stream / buffers / hipGraphs, the origin's weights) -- bench.py's "lanes":
                    # consecutive batches on the two overlap on the device; each gives the oracle's bits, run after run, whatever runs beside it
                    ctx2 = L.Context(0)
                    r = m.clone(ctx2)
                    try:
                        assert r.weight_arena() == m.weight_arena()  # ONE weight set
                        r.bind_input("x", x.shape)
                        r.prepare()
                        assert r.planned_steps == m.planned_steps
                        from rten_amd.tensor import DeviceTensor
                        DeviceTensor(ctx2, x.shape, np.float32, ptr=r.input_ptrs["x"], keepalive=r).upload(x[::-1].copy())
                        ctx2.sync()
                        for _ in range(6):
                            m.run(join=False)
                            r.run(join=False)
                        m.sync()
                        r.sync()
                        for mm, cc, ww in ((m, ctx, want), (r, ctx2, want[::-1])):
                            optr, oshape = mm.output(0)
                            _bits_equal(DeviceTensor(cc, oshape, np.float32, ptr=optr, keepalive=mm).numpy(), np.ascontiguousarray(ww), "f32 origin / replica side by side")
                        with pytest.raises(L.HipError):  # the origin cannot go while a replica lives
                            m._check(m.lib.rten_hip_model_destroy(m.h))
                    finally:
                        r.close()
                        ctx2.close()
            finally:
                m.close()
    finally:
        ctx.close()


def test_resnet50_int8_batch32_committed_plan_is_the_oracle():
    """BASELINE configs[2] as bench.py --config int8 runs it: one chain, profiles/plans/int8.json (quantized-output edges incl. a stage output with
    two scale products, one quantize-on-load layer).  A batch-coupled graph refuses sub-batch chains."""
    from rten_amd import lib as L, onnx_writer as ow
    want = bo.resnet50_int8_logits()
    x = bo.resnet_input()
    onnx_bytes = ow.resnet50_int8(bo.resnet_weights())
    plan = json.loads(_plan("int8.json"))
    ctx = L.Context(0)
    try:
        with pytest.raises(L.HipError) as e:  # DynamicQuantizeLinear takes min / max over the whole batch
            L.Model(ctx, onnx_bytes, None, 4)
        assert "couples the rows" in str(e.value), str(e.value)
        with pytest.raises(L.HipError) as e:
            L.Model(ctx, onnx_bytes, _plan("int8.json"), 2)
        assert "qout" in str(e.value), str(e.value)
        for text in (_plan("int8.json"), None, json.dumps({"fused_dql": plan["fused_dql"]})):
            m = L.Model(ctx, onnx_bytes, text, 1)
            try:
                m.bind_input("x", x.shape)
                m.prepare()
                if text is not None and "qout" in text:
                    assert m.planned_steps == len(plan["qout"]), (m.planned_steps, len(plan["qout"]))  # every listed edge (no loader layer converts beside them)
                elif text is None:
                    assert m.planned_steps == 0
                else:
                    assert m.planned_steps == 5, m.planned_steps  # the listed layers whose quantizer has no other reader (not s1b0c1 / s2b0c1: shared with a shortcut)
                for got in _run(m, ctx, {"x": x}, reps=3):
                    _bits_equal(got, want, f"int8, plan {'file' if text else 'none'}")
            finally:
                m.close()
    finally:
        ctx.close()


def test_resnet50_int8_four_replicas_side_by_side_on_different_batches():
    """bench.py --config int8 exactly as the default line runs it: profiles/plans/int8_lanes.json (quantize-on-load layers, NO quantized-output edges), four replicas
    of one model on four contexts, consecutive batches handed round robin -- here two DIFFERENT batches (seeds 1234 / 1235) alternating over the replicas, all in
    flight together.  Every replica's logits are the oracle's for ITS batch (each batch quantizes with its own statistics).  And the rule that makes the plan
    what it is: a model whose plan lists quantized-output edges cannot be cloned."""
    from rten_amd import lib as L, onnx_writer as ow
    from rten_amd.tensor import DeviceTensor
    onnx_bytes = ow.resnet50_int8(bo.resnet_weights())
    seeds = (1234, 1235, 1234, 1235)
    plan = json.loads(_plan("int8_lanes.json"))
    assert "qout" not in plan
    ctxs = [L.Context(0) for _ in seeds]
    models = []
    try:
        q = L.Model(ctxs[0], onnx_bytes, _plan("int8.json"), 1)  # (its plan has "qout" edges)
        try:
            with pytest.raises(L.HipError) as e:
                q.clone(ctxs[1])
            assert "qout" in str(e.value) and "replicas" in str(e.value), str(e.value)
        finally:
            q.close()
        models.append(L.Model(ctxs[0], onnx_bytes, _plan("int8_lanes.json"), 1))
        for c in ctxs[1:]:
            models.append(models[0].clone(c))
        for m, c, seed in zip(models, ctxs, seeds):
            x = bo.resnet_input(seed)
            m.bind_input("x", x.shape)
            m.prepare()
            # the quantize-on-load layers whose quantizer has no other reader (5) + the per-layer workgroup tiles chosen under co-run (round 6: "<layer>": [tile, 0, 1, 0])
            n_tiles = sum(1 for k, v in plan.items() if k not in ("fused_dql", "qout", "qout2") and isinstance(v, list) and len(v) == 4)
            assert n_tiles >= 10 and m.planned_steps == 5 + n_tiles, (m.planned_steps, n_tiles)
            assert m.weight_arena() == models[0].weight_arena()
            DeviceTensor(c, x.shape, np.float32, ptr=m.input_ptrs["x"], keepalive=m).upload(x)
            c.sync()
        for _ in range(5):
            for m in models:
                m.run(join=False)
        for m in models:
            m.sync()
        for i, (m, c, seed) in enumerate(zip(models, ctxs, seeds)):
            optr, oshape = m.output(0)
            _bits_equal(DeviceTensor(c, oshape, np.float32, ptr=optr, keepalive=m).numpy(), bo.resnet50_int8_logits(seed), f"int8 replica {i} (seed {seed}) beside three others")
    finally:
        for m in reversed(models):
            m.close()
        for c in ctxs:
            c.close()


def test_bert_base_batch32_seq128_through_the_model_abi_is_the_oracle():
    """BASELINE configs[3]: the 12-layer encoder from ONNX bytes (separate Q / K / V projections, Reshape / Transpose around the attention MatMuls,
    Add(mask) -> Softmax, LayerNormalization, Gelu as an exporter writes them) through rten_hip_model_*, ragged masks, tuned launch plan."""
    from rten_amd import lib as L, onnx_writer as ow
    B, S = 32, 128
    cfg, w, ids, am, tts, want = bo.bert_base_case(B, S)
    onnx_bytes = ow.bert_encoder(cfg, w, S)
    ctx = L.Context(0)
    try:
        # no plan, the one-replica plan, and the plan of tools/bench_bert.py's default line (four replicas: chosen under co-run, round 6 -- 128x128 tiles)
        texts = [None] + [open(os.path.join(ROOT, "profiles", "plans", f)).read() for f in ("bert_base_b32_s128.json", "bert_base_b32_s128_lanes.json")]
        for text in texts:
            m = L.Model(ctx, onnx_bytes, text, 1)
            try:
                feeds = {"input_ids": ids.astype(np.int32), "token_type_ids": tts.astype(np.int32), "attention_mask": am.astype(np.int32)}
                for i, name in enumerate(m.inputs):
                    m.bind_input(name, feeds[name].shape)
                    # the element types a host must hand over (rten_hip_model_input_dtype): ONNX int64 ids / masks are int32 on the device, like in the reference
                    assert m.input_dtype(i) == "int32", (name, m.input_dtype(i))
                m.prepare()
                assert m.output_dtype(0) == "float32"
                if text:
                    assert m.planned_steps >= 48, m.planned_steps  # 4 products per layer x 12 layers took their plan entry
                for got in _run(m, ctx, feeds):
                    _bits_equal(got.reshape(want.shape), want, "BERT-base b32 x 128")
            finally:
                m.close()
    finally:
        ctx.close()
