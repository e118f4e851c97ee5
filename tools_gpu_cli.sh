python - <<PY
import sys
sys.path.insert(0,".")
from rten_amd import onnx_writer as ow
from rten_amd.models import resnet50
W = resnet50.make_weights()
open("/tmp/r50.onnx","wb").write(ow.resnet50_f32(W)); open("/tmp/r50i8.onnx","wb").write(ow.resnet50_int8(W))
PY
./rten_amd/bin/rten_hip_run -n 10 -s batch=32 --tune --graph /tmp/r50.onnx | tail -16
./rten_amd/bin/rten_hip_run -n 10 -s batch=32 --graph /tmp/r50i8.onnx | tail -14
./rten_amd/bin/rten_hip_run -n 10 -s batch=1 --tune --graph /tmp/r50.onnx | tail -5
