#!/bin/bash
# The ONNX path end to end on a GPU box: manufacture the three model files (rten_amd/onnx_writer.py) and run them through
# the C++ loader / executor CLI.  Usage: gpurun --timeout 900 -- 'bash tools/gpu/cli.sh'
python - <<PY
import sys
sys.path.insert(0, ".")
from rten_amd import onnx_writer as ow
from rten_amd.workloads import bert, resnet50
W = resnet50.make_weights()
open("/tmp/resnet50.onnx", "wb").write(ow.resnet50_f32(W))
open("/tmp/resnet50_int8.onnx", "wb").write(ow.resnet50_int8(W))
cfg = bert.BertConfig()
open("/tmp/bert_base.onnx", "wb").write(ow.bert_encoder(cfg, bert.make_weights(cfg), 128))
PY
python -c "from tests.test_graph_executor import build_cli; build_cli()"
./rten_amd/bin/rten_hip_run -n 10 -s batch=32 --tune --graph /tmp/resnet50.onnx | tail -6
./rten_amd/bin/rten_hip_run -n 10 -s batch=1 --tune --graph /tmp/resnet50.onnx | tail -3
./rten_amd/bin/rten_hip_run -n 10 -s batch=32 --graph /tmp/resnet50_int8.onnx | tail -4
./rten_amd/bin/rten_hip_run -n 10 -s batch=32 --graph /tmp/bert_base.onnx | tail -4
