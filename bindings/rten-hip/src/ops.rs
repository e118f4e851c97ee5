//! `Operator` impls (src/operator.rs:486-613) for the ops of the hot path, each wrapping the reference's own operator struct.
//! `run` = validate on the host exactly as the reference does (same checks, same `OpError` messages -- the tests of the
//! reference assert them), move the activations to the device, call ONE entry point of the C ABI, move the result back into a
//! buffer from `ctx.pool()`.  Everything else (`name`, `max_inputs`, `output_types`, `as_infer_shapes`) is the wrapped operator's.
//! NOT COMPILED in the build image -- see lib.rs.
use std::ffi::c_void;
use std::ptr::null;
use std::sync::Arc;

use rten::ops::{self, calc_output_size_and_padding, InferShapes, OpError, OpRunContext, Operator, OutputList, OutputTypeList, OutputTypesContext, Padding};
use rten::OpRegistry;
use rten_hip_sys as sys;
use rten_tensor::prelude::*;
use rten_tensor::{NdTensorView, Tensor, TensorView};

use crate::{ConstCache, DeviceBuffer, HipContext};

/// Delegation of the parts of the trait that are not `run`.
macro_rules! delegate_to_inner {
    () => {
        fn name(&self) -> &str { self.inner.name() }
        fn max_inputs(&self) -> Option<usize> { self.inner.max_inputs() }
        fn output_types(&self, ctx: &OutputTypesContext) -> Option<OutputTypeList> { self.inner.output_types(ctx) }
        fn as_infer_shapes(&self) -> Option<&dyn InferShapes> { self.inner.as_infer_shapes() }
    };
}

fn contiguous<'a, T: Copy>(pool: &rten::BufferPool, v: &'a TensorView<T>) -> std::borrow::Cow<'a, [T]> {
    v.to_contiguous_in(pool).into_data() // the reference's own way to get a dense slice (src/ops/conv.rs:226, matmul.rs:262)
}

fn download_tensor<T: Copy + Default>(hip: &HipContext, pool: &rten::BufferPool, buf: &DeviceBuffer, shape: &[usize]) -> Result<Tensor<T>, OpError> {
    let len: usize = shape.iter().product();
    let mut data: Vec<T> = pool.alloc(len);
    data.resize(len, T::default());
    hip.download(buf, &mut data)?;
    Ok(Tensor::from_data(shape, data))
}

// ------------------------------------------------------------------------------------------------ Conv (src/ops/conv.rs:367-403)
pub struct HipConv { pub inner: ops::Conv, pub hip: Arc<HipContext>, pub consts: ConstCache }

fn conv_desc(x: &[usize], w: &[usize], op: &ops::Conv) -> Result<sys::rten_hip_conv2d_desc, OpError> {
    // checks and messages of conv_impl, src/ops/conv.rs:136-214
    let [n, c, h, wd]: [usize; 4] = x.try_into().map_err(|_| OpError::InvalidValue("input must have 4 dims (NCHW)"))?;
    let [o, kc, kh, kw]: [usize; 4] = w.try_into().map_err(|_| OpError::InvalidValue("kernel must have 4 dims (OCHW)"))?;
    let (oh, ow, pads) = calc_output_size_and_padding((h, wd), (kh, kw), (op.strides[0], op.strides[1]), op.padding.clone(),
                                                      Some((op.dilations[0], op.dilations[1])), false)?;
    if op.groups == 0 { return Err(OpError::InvalidValue("Group count must be > 0")); }
    if c % op.groups != 0 { return Err(OpError::InvalidValue("Input channel count not divisible by groups")); }
    if c / op.groups != kc { return Err(OpError::IncompatibleInputShapes("Input channels (per group) does not match kernel input channels")); }
    if o % op.groups != 0 { return Err(OpError::InvalidValue("Output channel count not divisible by groups")); }
    Ok(sys::rten_hip_conv2d_desc { n: n as i32, c: c as i32, h: h as i32, w: wd as i32, o: o as i32, kh: kh as i32, kw: kw as i32,
                                   pads: [pads[0] as i32, pads[1] as i32, pads[2] as i32, pads[3] as i32],
                                   stride_h: op.strides[0] as i32, stride_w: op.strides[1] as i32, dil_h: op.dilations[0] as i32, dil_w: op.dilations[1] as i32,
                                   groups: op.groups as i32, out_h: oh as i32, out_w: ow as i32 })
}

impl Operator for HipConv {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let inputs = ctx.inputs();
        let x: TensorView<f32> = inputs.require_as(0)?;
        let w: TensorView<f32> = inputs.require_as(1)?;
        let bias: Option<NdTensorView<f32, 1>> = inputs.get_as(2)?;
        let d = conv_desc(x.shape(), w.shape(), &self.inner)?;
        if let Some(b) = &bias { if b.size(0) != d.o as usize { return Err(OpError::IncompatibleInputShapes("bias.size(0) != out_channels")); } }
        let hip = &self.hip;
        let xs = contiguous(ctx.pool(), &x);
        let xd = hip.upload(&xs)?;
        // weights: staged once per graph constant (rten_hip_conv2d_f32_prepack), then served from the backend's cache
        let ws = contiguous(ctx.pool(), &w);
        let wd = self.consts.get(hip, 1, &ws, 1, |hip, raw| {
            let packed = hip.alloc(unsafe { sys::rten_hip_conv2d_f32_packed_bytes(&d) })?;
            hip.check(unsafe { sys::rten_hip_conv2d_f32_prepack(hip.raw(), &d, raw.ptr as *const f32, packed.ptr as *mut f32) })?;
            Ok(packed)
        })?;
        let bd = match &bias { Some(b) => self.consts.get(hip, 2, b.to_contiguous().data().unwrap(), 0, |_, raw| Ok(raw))?, None => null() };
        let out_shape = [d.n as usize, d.o as usize, d.out_h as usize, d.out_w as usize];
        let yd = hip.alloc(out_shape.iter().product::<usize>() * 4)?;
        hip.check(unsafe { sys::rten_hip_conv2d_f32(hip.raw(), &d, xd.ptr as *const f32, wd as *const f32, 1, bd as *const f32, null(), 0, yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, &out_shape)?.into()].into())
    }
}

// ------------------------------------------------------------------------------------------------ MatMul (src/ops/matmul.rs:387-428)
pub struct HipMatMul { pub inner: ops::MatMul, pub hip: Arc<HipContext>, pub consts: ConstCache }

/// numpy.matmul shape rules of matmul_impl (src/ops/matmul.rs:208-385): returns (batch, m, k, n, a_bs, b_bs, out shape)
fn matmul_shapes(a: &[usize], b: &[usize]) -> Result<(usize, usize, usize, usize, i64, i64, Vec<usize>), OpError> {
    if a.is_empty() || b.is_empty() { return Err(OpError::InvalidValue("Inputs must have >= 1 dimensions")); }
    let (a_vec, b_vec) = (a.len() == 1, b.len() == 1);
    let a2: Vec<usize> = if a_vec { vec![1, a[0]] } else { a.to_vec() };
    let b2: Vec<usize> = if b_vec { vec![b[0], 1] } else { b.to_vec() };
    let (m, k, kb, n) = (a2[a2.len() - 2], a2[a2.len() - 1], b2[b2.len() - 2], b2[b2.len() - 1]);
    if k != kb { return Err(OpError::IncompatibleInputShapes("Columns of first matrix does not match rows of second matrix")); }
    let (ap, bp) = (&a2[..a2.len() - 2], &b2[..b2.len() - 2]);
    let prefix = rten_tensor::broadcast_shapes(ap, bp).ok_or(OpError::IncompatibleInputShapes("Cannot broadcast shapes"))?;
    let (na, nb): (usize, usize) = (ap.iter().product(), bp.iter().product());
    let batch: usize = prefix.iter().product();
    // the ABI's two strides cover "same prefix" and "one side is a single matrix"; any other broadcast is materialised by the
    // caller with the reference's own `broadcast` + `to_contiguous` first (not shown)
    let a_bs = if na == 1 { 0 } else { (m * k) as i64 };
    let b_bs = if nb == 1 { 0 } else { (k * n) as i64 };
    let mut out = prefix.to_vec();
    if !a_vec { out.push(m); }
    if !b_vec { out.push(n); }
    Ok((batch, m, k, n, a_bs, b_bs, out))
}

impl Operator for HipMatMul {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let a: TensorView<f32> = ctx.inputs().require_as(0)?;
        let b: TensorView<f32> = ctx.inputs().require_as(1)?;
        let (batch, m, k, n, a_bs, b_bs, out_shape) = matmul_shapes(a.shape(), b.shape())?;
        let hip = &self.hip;
        let (asl, bsl) = (contiguous(ctx.pool(), &a), contiguous(ctx.pool(), &b));
        let (ad, bd) = (hip.upload(&asl)?, hip.upload(&bsl)?);
        let yd = hip.alloc(batch * m * n * 4)?;
        let d = sys::rten_hip_gemm_desc { m: m as i32, n: n as i32, k: k as i32, a_rs: k as i64, a_cs: 1, b_rs: n as i64, b_cs: 1, ldc: n as i64,
                                          batch: batch as i32, a_bs, b_bs, c_bs: (m * n) as i64, alpha: 1.0, ..Default::default() };
        hip.check(unsafe { sys::rten_hip_gemm_f32(hip.raw(), &d, ad.ptr as *const f32, bd.ptr as *const f32, null(), yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, &out_shape)?.into()].into())
    }
}

// ------------------------------------------------------------------------------------------------ MatMulInteger (matmul.rs:582-700)
pub struct HipMatMulInteger { pub inner: ops::MatMulInteger, pub hip: Arc<HipContext>, pub consts: ConstCache }

impl Operator for HipMatMulInteger {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        // u8 x i8 shown; the four signedness combinations of matmul.rs:684-690 differ in `a_signed` / `b_signed` only
        let a: TensorView<u8> = ctx.inputs().require_as(0)?;
        let b: TensorView<i8> = ctx.inputs().require_as(1)?;
        let a_zp: Option<TensorView<u8>> = ctx.inputs().get_as(2)?;
        let b_zp: Option<TensorView<i8>> = ctx.inputs().get_as(3)?;
        let a_rows = if a.ndim() > 1 { a.size(a.ndim() - 2) } else { 1 };
        let b_cols = if b.ndim() > 1 { b.size(b.ndim() - 1) } else { 1 };
        let zp_len = |zp: Option<&[usize]>, expected: usize| -> Result<i32, OpError> { // zero_point_to_vec, matmul.rs:513-531
            match zp {
                None => Ok(0),
                Some([]) => Ok(1),
                Some([len]) if *len == expected => Ok(expected as i32),
                Some([_]) => Err(OpError::InvalidValue("Zero point has incorrect size")),
                Some(_) => Err(OpError::UnsupportedValue("Only scalar or vector zero points are supported")),
            }
        };
        let azl = zp_len(a_zp.as_ref().map(|z| z.shape()), a_rows)?;
        let bzl = zp_len(b_zp.as_ref().map(|z| z.shape()), b_cols)?;
        let (batch, m, k, n, a_bs, b_bs, out_shape) = matmul_shapes(a.shape(), b.shape())?;
        let hip = &self.hip;
        let ad = hip.upload(&contiguous(ctx.pool(), &a))?;
        // constant RHS: staged once (rten_hip_gemm_int8_prepack = PackedBMatrix, Operator::prepack :696-705)
        let single_b = b_bs == 0;
        let bs = contiguous(ctx.pool(), &b);
        let bd = if single_b {
            self.consts.get(hip, 1, &bs, 2, |hip, raw| {
                let packed = hip.alloc(unsafe { sys::rten_hip_gemm_int8_packed_bytes(k as i32, n as i32) })?;
                hip.check(unsafe { sys::rten_hip_gemm_int8_prepack(hip.raw(), k as i32, n as i32, raw.ptr, n as i64, 1, 1, packed.ptr) })?;
                Ok(packed)
            })?
        } else { hip.upload(&bs)?.ptr as *const c_void /* (kept alive until the sync below in the real code) */ };
        let azd = a_zp.map(|z| hip.upload(z.to_contiguous().data().unwrap())).transpose()?;
        let bzd = b_zp.map(|z| hip.upload(z.to_contiguous().data().unwrap())).transpose()?;
        // `[A.., M, K] x [K, N]` is ONE product of A*M rows whose zero points cycle with period M (matmul.rs:259-296)
        let (mm, bt, abs_) = if single_b { (batch * m, 1, 0) } else { (m, batch as i32, a_bs) };
        let d = sys::rten_hip_gemm_int8_desc { m: mm as i32, n: n as i32, k: k as i32, a_rs: k as i64, a_cs: 1, b_rs: n as i64, b_cs: 1, ldc: n as i64,
                                               a_signed: 0, b_signed: 1, a_zp_len: azl, b_zp_len: bzl, scale_len: 0, batch: bt, a_bs: abs_, b_bs,
                                               c_bs: (m * n) as i64, b_prepacked: single_b as i32 };
        let yd = hip.alloc(batch * m * n * 4)?;
        hip.check(unsafe { sys::rten_hip_gemm_int8(hip.raw(), &d, ad.ptr, bd, azd.as_ref().map_or(null(), |z| z.ptr as *const c_void),
                                                   bzd.as_ref().map_or(null(), |z| z.ptr as *const c_void), null(), yd.ptr) })?;
        Ok([download_tensor::<i32>(hip, ctx.pool(), &yd, &out_shape)?.into()].into())
    }
}

// ------------------------------------------------------------------------------------------------ row-wise and element-wise operators
pub struct HipSoftmax { pub inner: ops::Softmax, pub hip: Arc<HipContext>, pub consts: ConstCache }
impl Operator for HipSoftmax {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let x: TensorView<f32> = ctx.inputs().require_as(0)?;
        let axis = ops::resolve_axis(x.ndim(), self.inner.axis)?; // "Axis is invalid" (src/ops/mod.rs)
        if axis + 1 != x.ndim() { return self.inner.run(ctx); }  // non-last axes: the reference's own path moves the axis; not on the hot path
        let (cols, rows) = (x.size(axis), x.len() / x.size(axis).max(1));
        let hip = &self.hip;
        let xd = hip.upload(&contiguous(ctx.pool(), &x))?;
        let yd = hip.alloc(x.len() * 4)?;
        hip.check(unsafe { sys::rten_hip_softmax_f32(hip.raw(), rows as i64, cols as i32, xd.ptr as *const f32, null(), 1, 1, self.inner.flush_nans_to_zero as i32, yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, x.shape())?.into()].into())
    }
}

pub struct HipLayerNormalization { pub inner: ops::LayerNormalization, pub hip: Arc<HipContext>, pub consts: ConstCache }
impl Operator for HipLayerNormalization {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let x: TensorView<f32> = ctx.inputs().require_as(0)?;
        let scale: TensorView<f32> = ctx.inputs().require_as(1)?;
        let bias: Option<TensorView<f32>> = ctx.inputs().get_as(2)?;
        let axis = ops::resolve_axis(x.ndim(), self.inner.axis)?;
        let cols: usize = x.shape()[axis..].iter().product();
        if scale.len() != cols || bias.as_ref().map_or(false, |b| b.len() != cols) { return self.inner.run(ctx); } // broadcast forms: reference path
        let hip = &self.hip;
        let xd = hip.upload(&contiguous(ctx.pool(), &x))?;
        let gd = self.consts.get(hip, 1, &contiguous(ctx.pool(), &scale), 0, |_, raw| Ok(raw))?;
        let bd = match &bias { Some(b) => self.consts.get(hip, 2, &contiguous(ctx.pool(), b), 0, |_, raw| Ok(raw))?, None => null() };
        let yd = hip.alloc(x.len() * 4)?;
        hip.check(unsafe { sys::rten_hip_layer_norm_f32(hip.raw(), (x.len() / cols.max(1)) as i64, cols as i32, xd.ptr as *const f32, gd as *const f32, bd as *const f32,
                                                       1.0, 0.0, self.inner.epsilon.unwrap_or(1e-5), yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, x.shape())?.into()].into())
    }
}

macro_rules! hip_unary {
    ($name:ident, $inner:ty, $entry:ident) => {
        pub struct $name { pub inner: $inner, pub hip: Arc<HipContext>, pub consts: ConstCache } // (`consts` unused here: one constructor shape for `register`)
        impl Operator for $name {
            delegate_to_inner!();
            fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
                let x: TensorView<f32> = ctx.inputs().require_as(0)?;
                let hip = &self.hip;
                let xd = hip.upload(&contiguous(ctx.pool(), &x))?;
                let yd = hip.alloc(x.len() * 4)?;
                hip.check(unsafe { sys::$entry(hip.raw(), x.len() as i64, xd.ptr as *const f32, yd.ptr as *mut f32) })?;
                Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, x.shape())?.into()].into())
            }
        }
    };
}
hip_unary!(HipRelu, ops::Relu, rten_hip_relu_f32);
hip_unary!(HipGelu, ops::Gelu, rten_hip_gelu_f32);
hip_unary!(HipErf, ops::Erf, rten_hip_erf_f32);

pub struct HipGlobalAveragePool { pub inner: ops::GlobalAveragePool, pub hip: Arc<HipContext>, pub consts: ConstCache }
impl Operator for HipGlobalAveragePool {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let x: NdTensorView<f32, 4> = ctx.inputs().require_as(0)?;
        let [n, c, h, w] = x.shape();
        let hip = &self.hip;
        let xd = hip.upload(&contiguous(ctx.pool(), &x.as_dyn()))?;
        let yd = hip.alloc(n * c * 4)?;
        hip.check(unsafe { sys::rten_hip_global_average_pool_f32(hip.raw(), (n * c) as i64, (h * w) as i32, xd.ptr as *const f32, yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, &[n, c, 1, 1])?.into()].into())
    }
}

pub struct HipDynamicQuantizeLinear { pub inner: ops::DynamicQuantizeLinear, pub hip: Arc<HipContext>, pub consts: ConstCache }
impl Operator for HipDynamicQuantizeLinear {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let x: TensorView<f32> = ctx.inputs().require_as(0)?;
        let hip = &self.hip;
        let xd = hip.upload(&contiguous(ctx.pool(), &x))?;
        let (yd, sd, zd) = (hip.alloc(x.len())?, hip.alloc(4)?, hip.alloc(1)?);
        hip.check(unsafe { sys::rten_hip_dynamic_quantize_linear(hip.raw(), x.len() as i64, xd.ptr as *const f32, yd.ptr as *mut u8, sd.ptr as *mut f32, zd.ptr as *mut u8) })?;
        Ok([download_tensor::<u8>(hip, ctx.pool(), &yd, x.shape())?.into(), download_tensor::<f32>(hip, ctx.pool(), &sd, &[])?.into(),
            download_tensor::<u8>(hip, ctx.pool(), &zd, &[])?.into()].into())
    }
}

/// Replace the CPU implementations of the hot-path operators by the ones above.
///
/// The registry's name -> deserialiser tables are `pub(crate)` (src/op_registry.rs:25-72), so an external crate cannot swap
/// entries by itself.  The integration therefore adds ONE hook to the reference -- a post-deserialisation wrapper,
/// `OpRegistry::set_op_wrapper(Box<dyn Fn(Box<dyn Operator + Send + Sync>) -> Box<dyn Operator + Send + Sync> + Send + Sync>)`,
/// applied by `read_op` to every operator it produces -- and nothing else: op names, attribute structs and their
/// deserialisation (`ReadOp`) stay the registry's.  `Operator: Any`, so the wrapper recognises the operators it accelerates by
/// downcast (as the fusion passes do with `graph.get_operator::<ConvInteger>`, src/optimize/fusions.rs:1052) and passes
/// every other operator through untouched.
pub fn register(reg: &mut OpRegistry, hip: Arc<HipContext>) {
    reg.set_op_wrapper(Box::new(move |op| {
        macro_rules! wrap {
            ($op:ty, $hip_op:ident) => {
                if (op.as_ref() as &dyn std::any::Any).is::<$op>() {
                    let inner = *(op as Box<dyn std::any::Any>).downcast::<$op>().unwrap();
                    return Box::new($hip_op { inner, hip: hip.clone(), consts: ConstCache::new() });
                }
            };
        }
        wrap!(ops::Conv, HipConv);
        wrap!(ops::MatMul, HipMatMul);
        wrap!(ops::MatMulInteger, HipMatMulInteger);
        wrap!(ops::Softmax, HipSoftmax);
        wrap!(ops::LayerNormalization, HipLayerNormalization);
        wrap!(ops::Relu, HipRelu);
        wrap!(ops::Gelu, HipGelu);
        wrap!(ops::Erf, HipErf);
        wrap!(ops::GlobalAveragePool, HipGlobalAveragePool);
        wrap!(ops::DynamicQuantizeLinear, HipDynamicQuantizeLinear);
        // ConvInteger / ConvIntegerToFloat, FusedMatMul / Gemm, MatMulIntegerToFloat, AddSoftmax, Add / Mul, MaxPool / AveragePool and
        // the attention operators follow the three shapes above (conv-like, matmul-like, element / row-wise); their ABI entry
        // points are listed in INTEGRATION.md section 2.3 and exercised by include/rten_hip_ops.hpp and rten_amd/ops.py.
        op
    }));
}
