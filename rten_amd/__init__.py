"""rten_amd -- MI355X (gfx950) operator backend for the RTen hot path.

`rten_amd.lib` binds the C ABI (include/rten_hip.h), `rten_amd.ops` mirrors RTen's operator
interface for the path, `rten_amd.workloads` builds the ResNet-50 / BERT-base graphs of BASELINE.json
from those operators.  Nothing in this package falls back to the CPU.
"""
from .lib import BackendUnavailable, Context, HipError  # noqa: F401
from .tensor import DeviceTensor  # noqa: F401

__version__ = "0.1.0"
