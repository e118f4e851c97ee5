python - <<PY
import sys
sys.path.insert(0,".")
from rten_amd import onnx_writer as ow
from rten_amd.models import bert
cfg = bert.BertConfig()
open("/tmp/bert.onnx","wb").write(ow.bert_encoder(cfg, bert.make_weights(cfg), 128))
PY
./rten_amd/bin/rten_hip_run -n 5 -s batch=32 --graph /tmp/bert.onnx | tail -8
./rten_amd/bin/rten_hip_run -n 2 -s batch=32 -t /tmp/bert.onnx | tail -16
