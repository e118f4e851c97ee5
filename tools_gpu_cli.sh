python - <<PY
import sys
sys.path.insert(0,".")
from rten_amd import onnx_writer as ow
from rten_amd.models import resnet50
W = resnet50.make_weights()
open("/tmp/r50i8.onnx","wb").write(ow.resnet50_int8(W))
PY
./rten_amd/bin/rten_hip_run -n 10 -s batch=32 --graph /tmp/r50i8.onnx | tail -8
./rten_amd/bin/rten_hip_run -n 3 -s batch=32 -t /tmp/r50i8.onnx | tail -14
