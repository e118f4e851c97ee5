#!/usr/bin/env python3
"""Package the BASELINE graphs for a check against the REAL reference (`rten`, Rust) on any box that has cargo.

The reference cannot be built in this image or on the GPU boxes (no Rust toolchain, no network), so the semantics the
oracle restates from reading the code -- above all the int8 padded-tap value RAW0_I8 (rten-gemm/src/im2col.rs:340-358), the
16-lane reduction order and the kc = 256 fold -- are pinned only to the reference's own unit-test literals
(tests/golden/reference_literals.json).  This script writes everything a maintainer needs to close that gap in one command
per model (see tools/make_rten_golden.md):

    <out>/resnet50_f32.onnx   resnet50_int8.onnx   bert_base.onnx        manufactured graphs (rten_amd/onnx_writer.py,
                                                                         seeded weights: byte-identical on every run)
    <out>/<model>.inputs.safetensors                                     seeded inputs (rten --inputs)
    <out>/<model>.expected.safetensors                                   outputs to check (rten --check-outputs)

`--source hip` (default when a GPU is present) takes the expected outputs from this backend (rten_hip_run --save-outputs);
`--source oracle` takes them from the CPU oracle (no GPU needed).  `rten model.onnx -i inputs --check-outputs expected`
prints `Output "<name>" vs expected: max diff <d>` (rten-cli/src/main.rs:366-457): 0.000000 means the real reference agrees
with this repository bit for bit on that graph; anything else localises the unpinned assumption (`--pad-mode` rebuilds the
int8 expectation under the other padded-tap modes).
"""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/rten_golden")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--source", choices=("hip", "oracle"), default=None)
    ap.add_argument("--pad-mode", choices=("raw0_i8", "zero_point", "raw0_u8"), default="raw0_i8")
    ap.add_argument("--models", default="resnet50_f32,resnet50_int8,bert_base")
    ap.add_argument("--reference-threads", type=int, default=0,
                    help="--source oracle, --batch 1: the reference's thread-pool size (RTEN_NUM_THREADS) its one-row gemv column blocks depend on "
                         "(rten-gemm/src/lib.rs:697); 0 = at least N / 128 threads, the backend's default assumption")
    args = ap.parse_args()
    from safetensors.numpy import save_file
    from rten_amd import onnx_writer as ow
    from rten_amd.workloads import bert, resnet50
    os.makedirs(args.out, exist_ok=True)
    source = args.source
    if source is None:
        import torch
        source = "hip" if torch.cuda.is_available() else "oracle"
    w = resnet50.make_weights()
    specs = resnet50.conv_specs()
    rng = np.random.default_rng(1234)
    jobs = {}
    if "resnet50_f32" in args.models:
        jobs["resnet50_f32"] = (ow.resnet50_f32(w), {"x": rng.random((args.batch, 3, 224, 224), dtype=np.float32)}, "logits")
    if "resnet50_int8" in args.models:
        jobs["resnet50_int8"] = (ow.resnet50_int8(w), {"x": rng.random((args.batch, 3, 224, 224), dtype=np.float32)}, "logits")
    cfg = bert.BertConfig(hidden=768, heads=12, layers=12, ffn=3072, vocab=4000, max_pos=128)
    if "bert_base" in args.models:
        wb = bert.make_weights(cfg)
        S = 128
        ids = rng.integers(0, cfg.vocab, (args.batch, S)).astype(np.int64)
        mask = np.ones((args.batch, S), np.int64)
        mask[-1, S - 9:] = 0
        jobs["bert_base"] = (ow.bert_encoder(cfg, wb, S), {"input_ids": ids, "token_type_ids": np.zeros((args.batch, S), np.int64), "attention_mask": mask},
                             "last_hidden_state")
    for name, (model_bytes, inputs, out_name) in jobs.items():
        mp = os.path.join(args.out, name + ".onnx")
        open(mp, "wb").write(model_bytes)
        ip = os.path.join(args.out, name + ".inputs.safetensors")
        save_file({k: np.ascontiguousarray(v) for k, v in inputs.items()}, ip)
        ep = os.path.join(args.out, name + ".expected.safetensors")
        if source == "hip":
            from tests.test_graph_executor import build_cli
            r = subprocess.run([build_cli(), "-s", f"batch={args.batch}", "-i", ip, "--save-outputs", ep, mp], capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit(f"{name}: rten_hip_run failed\n{r.stdout[-800:]}{r.stderr[-800:]}")
        else:
            from oracle import models as om, ref
            ref.set_gemv_threads(args.reference_threads)
            if name == "resnet50_f32":
                y = om.resnet50_forward(specs, w, inputs["x"])
            elif name == "resnet50_int8":
                pm = {"raw0_i8": ref.PAD_RAW0_I8, "zero_point": ref.PAD_ZERO_POINT, "raw0_u8": ref.PAD_RAW0_U8}[args.pad_mode]
                y = om.resnet50_int8_forward(specs, om.quantize_weights_int8(w), inputs["x"], pad_mode=pm)
            else:
                y = om.bert_forward(cfg, wb, inputs["input_ids"], inputs["attention_mask"], inputs["token_type_ids"]).reshape(args.batch, 128, cfg.hidden)
            save_file({out_name: np.ascontiguousarray(y, dtype=np.float32)}, ep)
        print(f"{name}: {mp} ({len(model_bytes) / 1e6:.1f} MB), inputs {ip}, expected ({source}) {ep}")
    print(f"\nnext, on a box with cargo (see tools/make_rten_golden.md):\n  for m in {' '.join(jobs)}; do cargo run -r -p rten-cli -- {args.out}/$m.onnx -s batch={args.batch} "
          f"-i {args.out}/$m.inputs.safetensors --check-outputs {args.out}/$m.expected.safetensors; done")


if __name__ == "__main__":
    main()
