"""Round-5 kernels against the oracle: the streaming 3x3 / stride-2 max-pool (16-byte row loads + the left neighbour's lane, statistics folded per
workgroup), and the executor-level changes that move bits through other launches (dropped all-zero weight zero points)."""
import ctypes as C

import numpy as np
import pytest

from oracle import ref
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def _bits(a, b):
    assert a.shape == b.shape
    same = (a.view(np.int32) == b.view(np.int32))
    assert same.all(), f"{(~same).sum()} of {same.size} differ, first at {tuple(np.argwhere(~same)[0])}"


@pytest.mark.parametrize("shape,pads", [((2, 3, 8, 8), (1, 1, 1, 1)), ((1, 5, 9, 12), (1, 1, 0, 1)), ((3, 2, 30, 20), (1, 1, 1, 0)), ((1, 70, 14, 16), (1, 1, 1, 1)),
                                         ((2, 64, 112, 112), (1, 1, 1, 1)), ((1, 1, 7, 4), (1, 1, 1, 1)), ((5, 7, 23, 36), (1, 1, 0, 0))])
def test_streaming_max_pool_bits_and_statistics(ctx, shape, pads):
    """Geometries that take maxpool3x3s2_stream_kernel (W % 4 == 0, out_w == W / 2, one leading padding row and column): ragged row groups (out_h not
    a multiple of 4), rows cut by the bottom edge with and without padding, one 16-byte group per row, thread counts that do not fill the last
    workgroup, planes that start mid-wave (lane 0 fetches its own left neighbour), NaN / -inf inputs (the sign of a zero maximum of +0 and -0 is the host libm's choice in the oracle: not tested) -- bit-identical to the oracle; and the
    statistics block equals the min / max of the result."""
    n, c, h, w = shape
    rng = ref.XorShiftRng(7 + h * w)
    x = (rng.f32(n * c * h * w).reshape(shape) * 4.0 - 2.0).astype(np.float32)
    x.reshape(-1)[5::193] = np.nan
    x.reshape(-1)[11::211] = -np.inf
    oh = (h + pads[0] + pads[2] - 3) // 2 + 1
    ow = (w + pads[1] + pads[3] - 3) // 2 + 1
    assert ow * 2 == w  # (the streaming form's condition: otherwise this test would exercise the round-4 kernel)
    want = ref.max_pool(x, (3, 3), (2, 2), pads)
    pd = L.Pool2dDesc(n, c, h, w, 3, 3, 2, 2, (C.c_int32 * 4)(*pads), oh, ow, 0)
    xd = DeviceTensor.from_numpy(ctx, x)
    y_a, y_b = DeviceTensor(ctx, want.shape, np.float32), DeviceTensor(ctx, want.shape, np.float32)
    st = DeviceTensor(ctx, (ctx.lib.rten_hip_minmax_stats_bytes(),), np.uint8)
    ctx.call("rten_hip_minmax_stats_reset", st.vp, 1)
    ctx.call("rten_hip_max_pool2d_f32", C.byref(pd), xd.vp, y_a.vp)
    ctx.call("rten_hip_max_pool2d_f32_stats", C.byref(pd), xd.vp, y_b.vp, st.vp)
    ctx.sync()
    _bits(y_a.numpy(), want)
    _bits(y_b.numpy(), want)
    # the statistics as ordered uints: 256 minima then 256 maxima (csrc/quantize.h); NaNs never enter (fminf / fmaxf drop them)
    raw = st.numpy().view(np.uint32)

    def ord2f(u):
        u = np.where(u & 0x80000000, u & 0x7fffffff, ~u).astype(np.uint32)
        return u.view(np.float32)
    mins, maxs = ord2f(raw[:256][raw[:256] != 0xffffffff]), ord2f(raw[256:512][raw[256:512] != 0])
    finite = want[~np.isnan(want)]
    assert mins.min() == finite.min() and maxs.max() == finite.max()
