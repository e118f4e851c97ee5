"""ONNX files written by PyTorch's own exporter, without the `onnx` python package (absent in this image).

`torch.onnx.export` refuses to run without `onnx`, but that package is only used for post-processing: the TorchScript
exporter's graph passes and the protobuf serialiser are C++ inside torch.  `export_bytes` drives them directly
(`_model_to_graph` + `Graph._export_onnx`, the calls `torch.onnx.export` makes itself), so the tests can feed
`rten_hip_run` / `include/rten_hip_graph.hpp` REAL exporter output -- node naming, initializer layout, BN folding, Gemm /
Flatten / Shape idioms as PyTorch emits them -- next to the graphs `rten_amd/onnx_writer.py` manufactures.

Test and tooling infrastructure only: nothing under rten_amd/ imports this.
    python tools/torch_export.py resnet50 /tmp/resnet50_torch.onnx     # BASELINE topology, the harness's synthetic weights
    python tools/torch_export.py encoder /tmp/bert_base_torch.onnx     # BERT-base sized encoder in plain torch.nn, batch 32 x 128
    python tools/torch_export.py bert /tmp/bert_torch.onnx             # transformers.BertModel, random init, 2 layers (its mask
                                                                       # subgraph needs NonZero / Where / Expand: not loadable yet)
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def export_bytes(model, args, input_names, output_names, dynamic_axes=None, opset: int = 17) -> bytes:
    import torch
    import torch.onnx._internal.torchscript_exporter.utils as TU
    from torch.onnx._internal.torchscript_exporter._globals import GLOBALS
    warnings.filterwarnings("ignore")
    GLOBALS.export_onnx_opset_version = opset
    model.eval()
    ONNX = torch.onnx.OperatorExportTypes.ONNX
    with torch.no_grad(), TU.exporter_context(model, torch.onnx.TrainingMode.EVAL, False):
        graph, params, _ = TU._model_to_graph(model, args, verbose=False, input_names=input_names, output_names=output_names,
                                              operator_export_type=ONNX, do_constant_folding=True, dynamic_axes=dynamic_axes or {})
        proto, _, _, _ = graph._export_onnx(params, opset, dynamic_axes or {}, False, ONNX, True, False, {}, True, "", {})
    return proto


def resnet50_module(weights):
    """ResNet-50 v1.5 as torchvision / timm lay it out, BatchNorm already folded (Conv2d with bias), parameters taken from
    rten_amd.workloads.resnet50.make_weights() so the oracle can run the same network."""
    import torch
    import torch.nn as nn
    from rten_amd.workloads import resnet50 as R

    specs = {l["name"]: l for l in R.conv_specs()}

    def conv(name):
        l = specs[name]
        c = nn.Conv2d(l["cin"], l["cout"], l["k"], stride=l["stride"], padding=l["pad"], bias=True)
        c.weight.data = torch.from_numpy(weights[name][0].copy())
        c.bias.data = torch.from_numpy(weights[name][1].copy())
        return c

    class Bottleneck(nn.Module):
        def __init__(self, pre, first):
            super().__init__()
            self.conv1, self.conv2, self.conv3 = conv(pre + "c1"), conv(pre + "c2"), conv(pre + "c3")
            self.downsample = conv(pre + "ds") if first else None

        def forward(self, x):
            idn = x if self.downsample is None else self.downsample(x)
            y = torch.relu(self.conv1(x))
            y = torch.relu(self.conv2(y))
            return torch.relu(self.conv3(y) + idn)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = conv("stem")
            self.pool = nn.MaxPool2d(3, stride=2, padding=1)
            self.blocks = nn.Sequential(*[Bottleneck(f"s{si}b{bi}", bi == 0) for si, (_, n, _) in enumerate(R.STAGES) for bi in range(n)])
            self.fc = nn.Linear(2048, weights["fc"][0].shape[0])
            self.fc.weight.data = torch.from_numpy(weights["fc"][0].copy())
            self.fc.bias.data = torch.from_numpy(weights["fc"][1].copy())

        def forward(self, x):
            y = self.blocks(self.pool(torch.relu(self.stem(x))))
            return self.fc(torch.flatten(nn.functional.adaptive_avg_pool2d(y, 1), 1))

    return Net().eval()


def resnet50_onnx(weights, image: int = 224) -> bytes:
    import torch
    return export_bytes(resnet50_module(weights), (torch.zeros(2, 3, image, image),), ["x"], ["logits"], {"x": {0: "batch"}, "logits": {0: "batch"}})


def encoder_module(cfg, w, seq):
    """A BERT encoder in plain torch.nn (Embedding, Linear, LayerNorm, GELU, matmul / softmax attention with an additive
    mask) holding rten_amd.workloads.bert.make_weights(cfg): the operator order of oracle.models.bert_forward.  PyTorch's
    exporter writes nn.LayerNorm as ReduceMean / Sub / Pow / Sqrt / Div / Mul / Add and nn.GELU as Div / Erf / Add / Mul / Mul,
    Linear as MatMul + Add, scalars as Constant nodes."""
    import math
    import torch
    import torch.nn as nn

    def lin(wm, b):
        l = nn.Linear(wm.shape[0], wm.shape[1])
        l.weight.data, l.bias.data = torch.from_numpy(np.ascontiguousarray(wm.T)), torch.from_numpy(b.copy())
        return l

    def ln(g, b):
        l = nn.LayerNorm(g.shape[0], eps=cfg.eps)
        l.weight.data, l.bias.data = torch.from_numpy(g.copy()), torch.from_numpy(b.copy())
        return l

    def emb(t):
        e = nn.Embedding(*t.shape)
        e.weight.data = torch.from_numpy(t.copy())
        return e

    class Layer(nn.Module):
        def __init__(self, lw):
            super().__init__()
            self.q, self.k, self.v, self.o = lin(lw["wq"], lw["bq"]), lin(lw["wk"], lw["bk"]), lin(lw["wv"], lw["bv"]), lin(lw["wo"], lw["bo"])
            self.ln1, self.ln2 = ln(lw["ln1_g"], lw["ln1_b"]), ln(lw["ln2_g"], lw["ln2_b"])
            self.f1, self.f2, self.act = lin(lw["w1"], lw["b1"]), lin(lw["w2"], lw["b2"]), nn.GELU()

        def forward(self, x, mask):
            B, S, H = x.shape
            d = H // cfg.heads

            def heads(t):
                return t.view(B, S, cfg.heads, d).permute(0, 2, 1, 3)
            scores = torch.matmul(heads(self.q(x)), heads(self.k(x)).transpose(-1, -2)) / math.sqrt(d) + mask
            ctx = torch.matmul(torch.softmax(scores, -1), heads(self.v(x))).permute(0, 2, 1, 3).reshape(B, S, H)
            x = self.ln1(self.o(ctx) + x)
            return self.ln2(self.f2(self.act(self.f1(x))) + x)

    class Encoder(nn.Module):
        def __init__(self):
            super().__init__()
            self.word, self.ttype, self.pos = emb(w["word"]), emb(w["type"]), emb(w["pos"])
            self.ln = ln(w["emb_ln_g"], w["emb_ln_b"])
            self.layers = nn.ModuleList([Layer(lw) for lw in w["layers"]])
            self.register_buffer("position_ids", torch.arange(seq).unsqueeze(0))

        def forward(self, input_ids, attention_mask, token_type_ids):
            x = self.ln((self.word(input_ids) + self.ttype(token_type_ids)) + self.pos(self.position_ids))
            mask = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * torch.finfo(torch.float32).min
            for l in self.layers:
                x = l(x, mask)
            return x

    return Encoder().eval()


def encoder_onnx(cfg, w, batch, seq) -> bytes:
    import torch
    ids = torch.zeros(batch, seq, dtype=torch.int64)
    return export_bytes(encoder_module(cfg, w, seq), (ids, torch.ones_like(ids), torch.zeros_like(ids)),
                        ["input_ids", "attention_mask", "token_type_ids"], ["last_hidden_state"])


def bert_module(layers=2, hidden=768, heads=12, ffn=3072, vocab=30522, seed=0):
    import torch
    from transformers import BertConfig, BertModel
    torch.manual_seed(seed)
    cfg = BertConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ffn,
                     max_position_embeddings=512, hidden_act="gelu", attn_implementation="eager")
    return BertModel(cfg, add_pooling_layer=False).eval()


def bert_onnx(model, batch=2, seq=128) -> bytes:
    import torch
    ids = torch.zeros(batch, seq, dtype=torch.int64)
    return export_bytes(model, (ids, torch.ones_like(ids), torch.zeros_like(ids)), ["input_ids", "attention_mask", "token_type_ids"],
                        ["last_hidden_state"])


if __name__ == "__main__":
    kind, path = sys.argv[1], sys.argv[2]
    if kind == "resnet50":
        from rten_amd.workloads import resnet50 as R
        data = resnet50_onnx(R.make_weights())
    elif kind == "encoder":  # BERT-base sized (BASELINE configs[3]): 12 layers, hidden 768, 12 heads, batch 32 x 128 tokens
        from rten_amd.workloads import bert as Bw
        cfg = Bw.BertConfig()
        data = encoder_onnx(cfg, Bw.make_weights(cfg), 32, 128)
    else:
        data = bert_onnx(bert_module())
    open(path, "wb").write(data)
    print(f"wrote {path}: {len(data)} bytes")
