//! `Operator` impls (src/operator.rs:486-613) for the operators of the hot path, each wrapping the reference's own operator.
//! `run` = validate on the host exactly as the reference does (same checks, same order, same `OpError` messages -- the reference's
//! tests assert them), move the activations to the device, call ONE entry point of the C ABI, move the result back into a buffer from
//! `ctx.pool()`.  Everything else (`name`, `max_inputs`, `output_types`, `as_infer_shapes`) is the wrapped operator's.
//! NOT COMPILED in the build image -- see lib.rs.  tests/test_abi.py checks, without a Rust toolchain, that every operator of
//! INTEGRATION.md section 2.3 has an `impl Operator for` here, that every `wrap!` target is a defined struct and that every `sys::` name
//! used exists in the generated -sys crate.
//!
//! WHEN the wrapping happens matters: the reference's fusion passes recognise operators by `Any` downcast
//! (`graph.get_operator::<ConvInteger>(id)`, src/optimize/fusions.rs:1052), and the fused operators (`ConvIntegerToFloat`,
//! `FusedMatMul`, `MatMulIntegerToFloat`, `AddSoftmax`) only exist after `GraphOptimizer::optimize` (src/optimize.rs:502-660).  The
//! operators are therefore wrapped AFTER optimisation, node by node (`accelerate` below, installed by `register`): fusions see the
//! operators they know, and the backend sees the fused graph -- the int8 ResNet-50 runs as 53 `ConvIntegerToFloat` launches, not as
//! ConvInteger + Cast + Mul on the CPU.
use std::ffi::c_void;
use std::ptr::null;
use std::sync::Arc;

use rten::ops::{self, calc_output_size_and_padding, InferShapes, OpError, OpRunContext, Operator, OutputList, OutputTypeList, OutputTypesContext, Padding, RoundMode};
use rten::{ModelOptions, ValueView};
use rten_hip_sys as sys;
use rten_tensor::prelude::*;
use rten_tensor::{NdTensorView, Tensor, TensorView};

use crate::{ConstCache, DeviceBuffer, HipContext};

pub type DynOp = Arc<dyn Operator + Send + Sync>;

/// Delegation of the parts of the trait that are not `run`.  In-place execution and rten-gemm prepacking are NOT forwarded
/// (`in_place_inputs` / `prepack_inputs` keep their empty defaults): outputs are fresh host tensors from `ctx.pool()`, and a
/// constant operand is staged on the device by `ConstCache`, not packed for a CPU kernel that will never run.
macro_rules! delegate_to_inner {
    () => {
        fn name(&self) -> &str { self.inner.name() }
        fn max_inputs(&self) -> Option<usize> { self.inner.max_inputs() }
        fn max_outputs(&self) -> Option<usize> { self.inner.max_outputs() }
        fn output_types(&self, ctx: &OutputTypesContext) -> Option<OutputTypeList> { self.inner.output_types(ctx) }
        fn is_commutative(&self) -> bool { self.inner.is_commutative() }
        fn is_associative(&self) -> bool { self.inner.is_associative() }
        fn as_infer_shapes(&self) -> Option<&dyn InferShapes> { self.inner.as_infer_shapes() }
    };
}

/// Fields every wrapper has: the reference operator it stands for, the device context, the device copies of its constant inputs.
macro_rules! hip_operator {
    ($(#[$m:meta])* $name:ident { $($field:ident : $ty:ty),* $(,)? }) => {
        $(#[$m])*
        #[derive(Debug)]
        pub struct $name { pub inner: DynOp, pub hip: Arc<HipContext>, pub consts: ConstCache, $(pub $field: $ty),* }
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

fn opt_ptr(b: &Option<DeviceBuffer>) -> *const c_void { b.as_ref().map_or(null(), |b| b.ptr as *const c_void) }

// ================================================================================================ convolution
/// Geometry attributes shared by Conv / ConvInteger / ConvIntegerToFloat (src/ops/conv.rs:367-375,474-480).
#[derive(Clone, Debug)]
pub struct ConvAttrs { pub groups: usize, pub dilations: Vec<usize>, pub padding: Padding, pub strides: Vec<usize> }

/// Checks, order and messages of conv_impl, src/ops/conv.rs:184-247 (1-D inputs are expanded to 2-D before this, :141-181).
fn conv_desc(x: &[usize], w: &[usize], bias_len: Option<usize>, a: &ConvAttrs) -> Result<sys::rten_hip_conv2d_desc, OpError> {
    let [n, c, h, wd]: [usize; 4] = x.try_into().map_err(|_| OpError::InvalidValue("input must have 4 dims (NCHW)"))?;
    let [o, kc, kh, kw]: [usize; 4] = w.try_into().map_err(|_| OpError::InvalidValue("kernel must have 4 dims (OCHW)"))?;
    if let Some(b) = bias_len { if b != o { return Err(OpError::IncompatibleInputShapes("bias.size(0) != out_channels")); } }
    let [sy, sx]: [usize; 2] = a.strides.as_slice().try_into().map_err(|_| OpError::InvalidValue("expected 2 stride values"))?;
    let [dy, dx]: [usize; 2] = a.dilations.as_slice().try_into().map_err(|_| OpError::InvalidValue("expected 2 dilation values"))?;
    let (oh, ow, pads) = calc_output_size_and_padding((h, wd), (kh, kw), (sy, sx), a.padding.clone(), Some((dy, dx)), RoundMode::default())?;
    if a.groups == 0 { return Err(OpError::InvalidValue("Group count must be > 0")); }
    if c % a.groups != 0 { return Err(OpError::InvalidValue("Input channel count not divisible by groups")); }
    if c / a.groups != kc { return Err(OpError::IncompatibleInputShapes("Input channels (per group) does not match kernel input channels")); }
    if o % a.groups != 0 { return Err(OpError::InvalidValue("Output channel count not divisible by groups")); }
    Ok(sys::rten_hip_conv2d_desc { n: n as i32, c: c as i32, h: h as i32, w: wd as i32, o: o as i32, kh: kh as i32, kw: kw as i32,
                                   pads: [pads[0] as i32, pads[1] as i32, pads[2] as i32, pads[3] as i32],
                                   stride_h: sy as i32, stride_w: sx as i32, dil_h: dy as i32, dil_w: dx as i32,
                                   groups: a.groups as i32, out_h: oh as i32, out_w: ow as i32 })
}

/// conv_impl's 1-D form (conv.rs:141-181): [N, C, W] x [O, C/g, Kw] runs as [N, C, 1, W] x [O, C/g, 1, Kw]; returns the 2-D shapes + attrs.
fn expand_1d(x: &[usize], w: &[usize], a: &ConvAttrs) -> Result<Option<(Vec<usize>, Vec<usize>, ConvAttrs)>, OpError> {
    if x.len() != 3 { return Ok(None); }
    if w.len() != 3 { return Err(OpError::InvalidValue("kernel must have 3 dims (OCW)")); }
    let padding = a.padding.clone().expand_1d_to_2d()?;
    let strides = match a.strides.as_slice() { &[s] => vec![1, s], _ => return Err(OpError::InvalidValue("expected 1 stride value")) };
    let dilations = match a.dilations.as_slice() { &[d] => vec![1, d], _ => return Err(OpError::InvalidValue("expected 1 dilation value")) };
    Ok(Some((vec![x[0], x[1], 1, x[2]], vec![w[0], w[1], 1, w[2]], ConvAttrs { groups: a.groups, dilations, padding, strides })))
}

hip_operator!(
    /// `Conv` (src/ops/conv.rs:367-403)
    HipConv { attrs: ConvAttrs });

impl Operator for HipConv {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let inputs = ctx.inputs();
        let x: TensorView<f32> = inputs.require_as(0)?;
        let w: TensorView<f32> = inputs.require_as(1)?;
        let bias: Option<NdTensorView<f32, 1>> = inputs.get_as(2)?;
        let one_d = expand_1d(x.shape(), w.shape(), &self.attrs)?;
        let (xs2, ws2, attrs) = match &one_d { Some((xs, ws, a)) => (xs.as_slice(), ws.as_slice(), a), None => (x.shape(), w.shape(), &self.attrs) };
        let d = conv_desc(xs2, ws2, bias.as_ref().map(|b| b.size(0)), attrs)?;
        let hip = &self.hip;
        let xd = hip.upload(&contiguous(ctx.pool(), &x))?;
        // weights: re-laid once per graph constant (rten_hip_conv2d_f32_prepack), then served from this operator's cache
        let wd = self.consts.get(hip, 1, &contiguous(ctx.pool(), &w), 1, |hip, raw| {
            let packed = hip.alloc(unsafe { sys::rten_hip_conv2d_f32_packed_bytes(&d) })?;
            hip.check(unsafe { sys::rten_hip_conv2d_f32_prepack(hip.raw(), &d, raw.ptr as *const f32, packed.ptr as *mut f32) })?;
            Ok(packed)
        })?;
        let bd = match &bias { Some(b) => self.consts.get(hip, 2, b.to_contiguous().data().unwrap(), 0, |_, raw| Ok(raw))?, None => null() };
        let out4 = [d.n as usize, d.o as usize, d.out_h as usize, d.out_w as usize];
        let yd = hip.alloc(out4.iter().product::<usize>() * 4)?;
        hip.check(unsafe { sys::rten_hip_conv2d_f32(hip.raw(), &d, xd.ptr as *const f32, wd as *const f32, 1, bd as *const f32, null(), 0, yd.ptr as *mut f32) })?;
        let out_shape: Vec<usize> = if one_d.is_some() { vec![out4[0], out4[1], out4[3]] } else { out4.to_vec() };
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, &out_shape)?.into()].into())
    }
}

/// The int8 operands of one ConvInteger call, as the four type combinations of conv.rs:514-520 present them.
struct Int8Operands<'a> { x: std::borrow::Cow<'a, [u8]>, x_shape: Vec<usize>, x_signed: bool, w: std::borrow::Cow<'a, [u8]>, w_shape: Vec<usize>, w_signed: bool,
                          x_zp: Option<[u8; 1]>, w_zp: Option<Vec<u8>>, w_zp_len: i32 }

/// Bytes of an i8 / u8 tensor without reinterpretation (the kernel applies the reference's ShiftCast itself, from `x_signed` / `w_signed`).
fn bytes_of<'a, T: Copy>(pool: &rten::BufferPool, v: &'a TensorView<T>) -> std::borrow::Cow<'a, [u8]> {
    debug_assert_eq!(std::mem::size_of::<T>(), 1);
    match contiguous(pool, v) {
        std::borrow::Cow::Borrowed(s) => std::borrow::Cow::Borrowed(unsafe { std::slice::from_raw_parts(s.as_ptr() as *const u8, s.len()) }),
        std::borrow::Cow::Owned(s) => std::borrow::Cow::Owned(s.iter().map(|e| unsafe { *(e as *const T as *const u8) }).collect()),
    }
}

/// conv_integer's own checks (conv.rs:421-468), in its order: the input zero point must be a scalar; the kernel zero point a scalar or a
/// vector of out_channels entries (zero_point_to_vec, matmul.rs:513-531).
fn int8_operands<'a>(ctx: &'a OpRunContext) -> Result<Int8Operands<'a>, OpError> {
    let inputs = ctx.inputs();
    let (input, weight) = (inputs.require(0)?, inputs.require(1)?);
    macro_rules! gather {
        ($x:expr, $xs:expr, $w:expr, $ws:expr) => {{
            let out_chans = if $w.ndim() >= 1 { $w.size(0) } else { 0 };
            let x_zp = match inputs.get_as(2)? {
                Some(zp) => { let zp: TensorView<_> = zp; let _: &TensorView<_> = &$x; // same element type as the input
                              match zp.item() { Some(z) => Some([unsafe { *(z as *const _ as *const u8) }]), None => return Err(OpError::InvalidValue("input zero point must be a scalar")) } }
                None => None,
            };
            let (w_zp, w_zp_len) = match inputs.get_as(3)? {
                None => (None, 0),
                Some(zp) => { let zp: TensorView<_> = zp; let _: &TensorView<_> = &$w;
                    match zp.ndim() {
                        0 => (Some(bytes_of(ctx.pool(), &zp).into_owned()), 1),
                        1 if zp.size(0) == out_chans => (Some(bytes_of(ctx.pool(), &zp).into_owned()), out_chans as i32),
                        1 => return Err(OpError::InvalidValue("Zero point has incorrect size")),
                        _ => return Err(OpError::UnsupportedValue("Only scalar or vector zero points are supported")),
                    } }
            };
            Ok(Int8Operands { x: bytes_of(ctx.pool(), &$x), x_shape: $x.shape().to_vec(), x_signed: $xs, w: bytes_of(ctx.pool(), &$w), w_shape: $w.shape().to_vec(),
                              w_signed: $ws, x_zp, w_zp, w_zp_len })
        }};
    }
    match (input, weight) { // the four combinations of conv.rs:514-520; anything else: UnsupportedType
        (ValueView::Int8Tensor(x), ValueView::Int8Tensor(w)) => gather!(x, true, w, true),
        (ValueView::Int8Tensor(x), ValueView::UInt8Tensor(w)) => gather!(x, true, w, false),
        (ValueView::UInt8Tensor(x), ValueView::Int8Tensor(w)) => gather!(x, false, w, true),
        (ValueView::UInt8Tensor(x), ValueView::UInt8Tensor(w)) => gather!(x, false, w, false),
        _ => Err(OpError::UnsupportedType),
    }
}

/// One `rten_hip_conv2d_int8` call: `scale` = None -> i32 output (ConvInteger), Some -> cast_scale fused (ConvIntegerToFloat).
fn run_conv_integer(op_hip: &Arc<HipContext>, consts: &ConstCache, attrs: &ConvAttrs, pad_mode: i32, ctx: &OpRunContext, scale: Option<f32>)
                    -> Result<(DeviceBuffer, Vec<usize>), OpError> {
    let o = int8_operands(ctx)?;
    let one_d = expand_1d(&o.x_shape, &o.w_shape, attrs)?;
    let (xs2, ws2, a2) = match &one_d { Some((xs, ws, a)) => (xs.as_slice(), ws.as_slice(), a), None => (o.x_shape.as_slice(), o.w_shape.as_slice(), attrs) };
    let conv = conv_desc(xs2, ws2, None, a2)?;
    let hip = op_hip;
    let mut d = sys::rten_hip_conv2d_int8_desc { conv, x_signed: o.x_signed as i32, w_signed: o.w_signed as i32, w_zp_len: o.w_zp_len, pad_mode,
                                                 weights_packed: 0, x_staged: 0, scale_len: 0 };
    let xd = hip.upload(&o.x)?;
    // constant weights: staged once in the kernel's chunk-major layout (rten_hip_conv2d_int8_prepack); 0 bytes = geometry not covered
    // by the staged kernel (grouped convolutions): the plain weights are passed then
    let packed_bytes = unsafe { sys::rten_hip_conv2d_int8_packed_bytes(&d) };
    let wd = if packed_bytes > 0 {
        d.weights_packed = 1;
        let dd = d;
        consts.get(hip, 1, &o.w, 3, |hip, raw| {
            let packed = hip.alloc(packed_bytes)?;
            hip.check(unsafe { sys::rten_hip_conv2d_int8_prepack(hip.raw(), &dd, raw.ptr, packed.ptr) })?;
            Ok(packed)
        })?
    } else {
        consts.get(hip, 1, &o.w, 0, |_, raw| Ok(raw))?
    };
    let xz = o.x_zp.map(|z| hip.upload(&z)).transpose()?;
    let wz = o.w_zp.as_ref().map(|z| hip.upload(z)).transpose()?;
    let sc = scale.map(|s| hip.upload(&[s])).transpose()?;
    let out4 = [conv.n as usize, conv.o as usize, conv.out_h as usize, conv.out_w as usize];
    let yd = hip.alloc(out4.iter().product::<usize>() * 4)?;
    hip.check(unsafe { sys::rten_hip_conv2d_int8(hip.raw(), &d, xd.ptr, wd, opt_ptr(&xz), opt_ptr(&wz), opt_ptr(&sc) as *const f32, null(), null(), 0, yd.ptr) })?;
    let out_shape = if one_d.is_some() { vec![out4[0], out4[1], out4[3]] } else { out4.to_vec() };
    Ok((yd, out_shape))
}

hip_operator!(
    /// `ConvInteger` (src/ops/conv.rs:470-536): u8 / i8 input x u8 / i8 kernel -> i32.  `pad_mode`: what a padded tap holds
    /// (rten_hip.h RTEN_HIP_PAD_*; the x86 reference's im2col writes raw 0 after the u8 -> i8 shift, rten-gemm/src/im2col.rs:340-358).
    HipConvInteger { attrs: ConvAttrs, pad_mode: i32 });

impl Operator for HipConvInteger {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let (yd, shape) = run_conv_integer(&self.hip, &self.consts, &self.attrs, self.pad_mode, ctx, None)?;
        Ok([download_tensor::<i32>(&self.hip, ctx.pool(), &yd, &shape)?.into()].into())
    }
}

hip_operator!(
    /// `ConvIntegerToFloat` (src/ops/conv.rs:538-587) = Cast(ConvInteger(x, w, x_zp, w_zp)) * scale with a scalar scale: ONE launch, the
    /// cast_scale of matmul.rs:734-773 in the convolution's epilogue (same single rounding: `acc as f32 * scale`).
    HipConvIntegerToFloat { attrs: ConvAttrs, pad_mode: i32 });

impl Operator for HipConvIntegerToFloat {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let scale: TensorView<f32> = ctx.inputs().require_as(4)?; // checked before the convolution runs, as conv.rs:567-570 does
        let Some(&scale) = scale.item() else { return Err(OpError::InvalidValue("scale should be a scalar")); };
        let (yd, shape) = run_conv_integer(&self.hip, &self.consts, &self.attrs, self.pad_mode, ctx, Some(scale))?;
        Ok([download_tensor::<f32>(&self.hip, ctx.pool(), &yd, &shape)?.into()].into())
    }
}

// ================================================================================================ matrix products
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
    // caller with the reference's own `broadcast` + `to_contiguous` first
    if na != 1 && nb != 1 && na != nb { return Err(OpError::UnsupportedValue("mixed batch broadcast: materialise the smaller operand first")); }
    let a_bs = if na == 1 { 0 } else { (m * k) as i64 };
    let b_bs = if nb == 1 { 0 } else { (k * n) as i64 };
    let mut out = prefix.to_vec();
    if !a_vec { out.push(m); }
    if !b_vec { out.push(n); }
    Ok((batch, m, k, n, a_bs, b_bs, out))
}

/// MatMul / FusedMatMul on the device: `[A.., M, K] x [K, N]` collapses to one product of A*M rows (matmul.rs:259-296), a batched RHS is one
/// launch with batch strides.  `bias` = per-column vector added after the first depth block (BiasVector::Row), `alpha` scales the product.
fn run_matmul_f32(hip: &Arc<HipContext>, consts: &ConstCache, ctx: &OpRunContext, bias: Option<NdTensorView<f32, 1>>, alpha: Option<f32>) -> Result<OutputList, OpError> {
    let a: TensorView<f32> = ctx.inputs().require_as(0)?;
    let b: TensorView<f32> = ctx.inputs().require_as(1)?;
    let (batch, m, k, n, a_bs, b_bs, out_shape) = matmul_shapes(a.shape(), b.shape())?;
    if let Some(bv) = &bias { if bv.size(0) != n { return Err(OpError::IncompatibleInputShapes("Bias length does not match output columns")); } }
    let ad = hip.upload(&contiguous(ctx.pool(), &a))?;
    // a rank-2 RHS is the weight of a projection: constant in practice, cached per operator instance (re-staged if it ever moves)
    let bs = contiguous(ctx.pool(), &b);
    let (bd_keep, bd): (Option<DeviceBuffer>, *const c_void) = if b_bs == 0 { (None, consts.get(hip, 1, &bs, 0, |_, raw| Ok(raw))?) }
                                                              else { let buf = hip.upload(&bs)?; let p = buf.ptr as *const c_void; (Some(buf), p) };
    let biasd = match &bias { Some(bv) => consts.get(hip, 2, bv.to_contiguous().data().unwrap(), 0, |_, raw| Ok(raw))?, None => null() };
    let (mm, bt, abs_) = if b_bs == 0 { (batch * m, 1, 0) } else { (m, batch as i32, a_bs) };
    let d = sys::rten_hip_gemm_desc { m: mm as i32, n: n as i32, k: k as i32, a_rs: k as i64, a_cs: 1, b_rs: n as i64, b_cs: 1, ldc: n as i64,
                                      batch: bt, a_bs: abs_, b_bs, c_bs: (m * n) as i64, alpha: alpha.unwrap_or(1.0), beta: 0.0,
                                      bias_kind: if bias.is_some() { sys::RTEN_HIP_BIAS_PER_COL } else { sys::RTEN_HIP_BIAS_NONE }, ..Default::default() };
    let yd = hip.alloc(batch * m * n * 4)?;
    hip.check(unsafe { sys::rten_hip_gemm_f32(hip.raw(), &d, ad.ptr as *const f32, bd as *const f32, biasd as *const f32, yd.ptr as *mut f32) })?;
    drop(bd_keep);
    Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, &out_shape)?.into()].into())
}

hip_operator!(
    /// `MatMul` (src/ops/matmul.rs:387-428)
    HipMatMul {});
impl Operator for HipMatMul {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> { run_matmul_f32(&self.hip, &self.consts, ctx, None, None) }
}

hip_operator!(
    /// `FusedMatMul` (src/ops/matmul.rs:455-510): MatMul + per-column bias + alpha, what MatMulAddFusion / MatMulScaleFusion produce.
    HipFusedMatMul { alpha: Option<f32> });
impl Operator for HipFusedMatMul {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let bias: Option<NdTensorView<f32, 1>> = ctx.inputs().get_as(2)?;
        run_matmul_f32(&self.hip, &self.consts, ctx, bias, self.alpha)
    }
}

hip_operator!(
    /// `Gemm` (src/ops/matmul.rs:32-156): c = alpha * (a b) + beta * c, transposes as strides; `c` is broadcast into the output first
    /// (`expand_to`, :70) and enters the product with the first depth block (beta != 0), exactly the reference's gemm(beta) call.
    HipGemm { alpha: f32, beta: f32, transpose_a: bool, transpose_b: bool });
impl Operator for HipGemm {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let inputs = ctx.inputs();
        let a: TensorView<f32> = inputs.require_as(0)?;
        let b: TensorView<f32> = inputs.require_as(1)?;
        let c: Option<TensorView<f32>> = inputs.get_as(2)?;
        if a.ndim() != 2 { return Err(OpError::InvalidValue("a must have 2 dims")); }
        if b.ndim() != 2 { return Err(OpError::InvalidValue("b must have 2 dims")); }
        let (ar, ac) = if self.transpose_a { (a.size(1), a.size(0)) } else { (a.size(0), a.size(1)) };
        let (br, bc) = if self.transpose_b { (b.size(1), b.size(0)) } else { (b.size(0), b.size(1)) };
        if ac != br { return Err(OpError::IncompatibleInputShapes("Columns of first matrix does not match rows of second matrix")); }
        let out_shape = [ar, bc];
        let hip = &self.hip;
        let ad = hip.upload(&contiguous(ctx.pool(), &a))?;
        let bd = self.consts.get(hip, 1, &contiguous(ctx.pool(), &b), 0, |_, raw| Ok(raw))?;
        let yd = hip.alloc(ar * bc * 4)?;
        let mut beta = 0.0;
        if let Some(c) = &c {
            if self.beta != 0.0 {
                if !c.can_broadcast_to(&out_shape) { return Err(OpError::IncompatibleInputShapes("Cannot broadcast c to output shape")); }
                let expanded = c.broadcast(&out_shape[..]).to_tensor_in(ctx.pool()); // expand_to (matmul.rs:70)
                hip.check(unsafe { sys::rten_hip_memcpy_h2d(hip.raw(), yd.ptr, expanded.data().unwrap().as_ptr() as *const c_void, ar * bc * 4) })?;
                beta = self.beta;
            }
        }
        // transposes are strides of the contiguous uploads: A[m, k] at m * a_rs + k * a_cs
        let (a_rs, a_cs) = if self.transpose_a { (1, ar as i64) } else { (ac as i64, 1) };
        let (b_rs, b_cs) = if self.transpose_b { (1, br as i64) } else { (bc as i64, 1) };
        let d = sys::rten_hip_gemm_desc { m: ar as i32, n: bc as i32, k: ac as i32, a_rs, a_cs, b_rs, b_cs, ldc: bc as i64, batch: 1,
                                          alpha: self.alpha, beta, ..Default::default() };
        hip.check(unsafe { sys::rten_hip_gemm_f32(hip.raw(), &d, ad.ptr as *const f32, bd as *const f32, null(), yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, &out_shape)?.into()].into())
    }
}

/// MatMulInteger on the device (matmul.rs:582-700): the four signedness combinations differ in `a_signed` / `b_signed` only; zero points
/// per zero_point_to_vec; `[A.., M, K] x [K, N]` is one product whose row zero points cycle with period M (matmul.rs:259-296).
/// `scale`: None -> i32 output; Some -> the fused cast_scale of MatMulIntegerToFloat (scalar or per-column, OutputScale::from_view).
fn run_matmul_integer(hip: &Arc<HipContext>, consts: &ConstCache, ctx: &OpRunContext, scale: Option<TensorView<f32>>) -> Result<(DeviceBuffer, Vec<usize>), OpError> {
    let inputs = ctx.inputs();
    macro_rules! go {
        ($a:expr, $as_:expr, $b:expr, $bs_:expr) => {{
            let a_rows = if $a.ndim() > 1 { $a.size($a.ndim() - 2) } else { 1 };
            let b_cols = if $b.ndim() > 1 { $b.size($b.ndim() - 1) } else { 1 };
            let zp_len = |zp: Option<&[usize]>, expected: usize| -> Result<i32, OpError> { // zero_point_to_vec, matmul.rs:513-531
                match zp {
                    None => Ok(0),
                    Some([]) => Ok(1),
                    Some([len]) if *len == expected => Ok(expected as i32),
                    Some([_]) => Err(OpError::InvalidValue("Zero point has incorrect size")),
                    Some(_) => Err(OpError::UnsupportedValue("Only scalar or vector zero points are supported")),
                }
            };
            let a_zp = match inputs.get_as(2)? { Some(z) => { let z: TensorView<_> = z; let _: &TensorView<_> = &$a; Some(z) } None => None };
            let b_zp = match inputs.get_as(3)? { Some(z) => { let z: TensorView<_> = z; let _: &TensorView<_> = &$b; Some(z) } None => None };
            let azl = zp_len(a_zp.as_ref().map(|z| z.shape()), a_rows)?;
            let bzl = zp_len(b_zp.as_ref().map(|z| z.shape()), b_cols)?;
            let (batch, m, k, n, a_bs, b_bs, out_shape) = matmul_shapes($a.shape(), $b.shape())?;
            let scale_len = match &scale { // OutputScale::from_view + cast_scale's column check (matmul.rs:717-752)
                None => 0,
                Some(s) => match s.ndim() {
                    0 => 1,
                    1 if s.size(0) == 1 => 1,
                    1 => { if s.size(0) != n { return Err(OpError::IncompatibleInputShapes("Scale length does not match tensor columns")); } n as i32 }
                    _ => return Err(OpError::InvalidValue("scale should have rank 0 or 1")),
                },
            };
            let ad = hip.upload(&bytes_of(ctx.pool(), &$a))?;
            // constant RHS: staged once (rten_hip_gemm_int8_prepack = PackedBMatrix, Operator::prepack :696-705)
            let single_b = b_bs == 0;
            let bsl = bytes_of(ctx.pool(), &$b);
            let (b_keep, bd): (Option<DeviceBuffer>, *const c_void) = if single_b {
                (None, consts.get(hip, 1, &bsl, 2, |hip, raw| {
                    let packed = hip.alloc(unsafe { sys::rten_hip_gemm_int8_packed_bytes(k as i32, n as i32) })?;
                    hip.check(unsafe { sys::rten_hip_gemm_int8_prepack(hip.raw(), k as i32, n as i32, raw.ptr, n as i64, 1, $bs_ as i32, packed.ptr) })?;
                    Ok(packed)
                })?)
            } else { let buf = hip.upload(&bsl)?; let p = buf.ptr as *const c_void; (Some(buf), p) };
            let azd = a_zp.as_ref().map(|z| hip.upload(&bytes_of(ctx.pool(), z))).transpose()?;
            let bzd = b_zp.as_ref().map(|z| hip.upload(&bytes_of(ctx.pool(), z))).transpose()?;
            let scd = scale.as_ref().map(|s| hip.upload(&contiguous(ctx.pool(), s))).transpose()?;
            let (mm, bt, abs_) = if single_b { (batch * m, 1, 0) } else { (m, batch as i32, a_bs) };
            let d = sys::rten_hip_gemm_int8_desc { m: mm as i32, n: n as i32, k: k as i32, a_rs: k as i64, a_cs: 1, b_rs: n as i64, b_cs: 1, ldc: n as i64,
                                                   a_signed: $as_ as i32, b_signed: $bs_ as i32, a_zp_len: azl, b_zp_len: bzl, scale_len, batch: bt, a_bs: abs_, b_bs,
                                                   c_bs: (m * n) as i64, b_prepacked: single_b as i32 };
            let yd = hip.alloc(batch * m * n * 4)?;
            hip.check(unsafe { sys::rten_hip_gemm_int8(hip.raw(), &d, ad.ptr, bd, opt_ptr(&azd), opt_ptr(&bzd), opt_ptr(&scd) as *const f32, yd.ptr) })?;
            drop(b_keep);
            Ok((yd, out_shape))
        }};
    }
    match (inputs.require(0)?, inputs.require(1)?) { // matmul.rs:684-690
        (ValueView::Int8Tensor(a), ValueView::Int8Tensor(b)) => go!(a, true, b, true),
        (ValueView::Int8Tensor(a), ValueView::UInt8Tensor(b)) => go!(a, true, b, false),
        (ValueView::UInt8Tensor(a), ValueView::Int8Tensor(b)) => go!(a, false, b, true),
        (ValueView::UInt8Tensor(a), ValueView::UInt8Tensor(b)) => go!(a, false, b, false),
        _ => Err(OpError::UnsupportedType),
    }
}

hip_operator!(
    /// `MatMulInteger` (src/ops/matmul.rs:582-710)
    HipMatMulInteger {});
impl Operator for HipMatMulInteger {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let (yd, shape) = run_matmul_integer(&self.hip, &self.consts, ctx, None)?;
        Ok([download_tensor::<i32>(&self.hip, ctx.pool(), &yd, &shape)?.into()].into())
    }
}

hip_operator!(
    /// `MatMulIntegerToFloat` (src/ops/matmul.rs:775-810) = cast_scale(MatMulInteger(..), scale), scale scalar or per column: one launch.
    HipMatMulIntegerToFloat {});
impl Operator for HipMatMulIntegerToFloat {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let scale: TensorView<f32> = ctx.inputs().require_as(4)?;
        if scale.ndim() > 1 { return Err(OpError::InvalidValue("scale should have rank 0 or 1")); } // OutputScale::from_view runs before the product (:789-791)
        let (yd, shape) = run_matmul_integer(&self.hip, &self.consts, ctx, Some(scale))?;
        Ok([download_tensor::<f32>(&self.hip, ctx.pool(), &yd, &shape)?.into()].into())
    }
}

// ================================================================================================ row-wise operators
hip_operator!(
    /// `Softmax` (src/ops/norm.rs:825-840)
    HipSoftmax { axis: isize, flush_nans_to_zero: bool });
impl Operator for HipSoftmax {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let x: TensorView<f32> = ctx.inputs().require_as(0)?;
        let axis = ops::resolve_axis(x.ndim(), self.axis)?; // "Axis is invalid" (src/ops/mod.rs)
        if axis + 1 != x.ndim() { return self.inner.run(ctx); }  // non-last axes: the reference's own path moves the axis; not on the hot path
        let (cols, rows) = (x.size(axis), x.len() / x.size(axis).max(1));
        let hip = &self.hip;
        let xd = hip.upload(&contiguous(ctx.pool(), &x))?;
        let yd = hip.alloc(x.len() * 4)?;
        hip.check(unsafe { sys::rten_hip_softmax_f32(hip.raw(), rows as i64, cols as i32, xd.ptr as *const f32, null(), 1, 1, self.flush_nans_to_zero as i32, yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, x.shape())?.into()].into())
    }
}

hip_operator!(
    /// `AddSoftmax` (src/ops/attention.rs:69-156): softmax(qk + m) over the last axis, `m` broadcast to `qk` (the larger input is `qk`:
    /// the operator is commutative).  The addend's broadcast is the ABI's two-level row pattern -- addend row = (row / add_div) % add_mod --
    /// which covers the attention masks ([B, 1, 1, T], [1, 1, S, T], [B, 1, S, T], full); any other pattern is materialised first with the
    /// reference's own `broadcast` + `to_tensor`.
    HipAddSoftmax { flush_nans_to_zero: bool });
impl Operator for HipAddSoftmax {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let x: TensorView<f32> = ctx.inputs().require_as(0)?;
        let y: TensorView<f32> = ctx.inputs().require_as(1)?;
        let (qk, m) = if x.len() > y.len() { (x, y) } else { (y, x) };
        const BROADCAST_ERROR: OpError = OpError::IncompatibleInputShapes("Cannot broadcast inputs");
        let out_shape = rten_tensor::broadcast_shapes(qk.shape(), m.shape()).ok_or(BROADCAST_ERROR)?;
        let qk_full = if out_shape.as_slice() == qk.shape() { qk.clone() } else { return self.inner.run(ctx) }; // both sides broadcast: the reference's copy path
        let cols = *out_shape.last().ok_or(OpError::InvalidValue("Axis is invalid"))?;
        let rows = qk_full.len() / cols.max(1);
        let mb = m.try_broadcast(out_shape.as_slice()).map_err(|_| BROADCAST_ERROR)?;
        // addend rows as (div, mod): find the pattern from the broadcast strides of the row axes; else materialise
        let pattern = addend_row_pattern(out_shape.as_slice(), m.shape());
        let hip = &self.hip;
        let xd = hip.upload(&contiguous(ctx.pool(), &qk_full))?;
        let (md, div, modulo) = match pattern {
            Some((div, modulo)) if m.shape().last() == Some(&cols) => (hip.upload(&contiguous(ctx.pool(), &m))?, div, modulo),
            _ => (hip.upload(mb.to_tensor_in(ctx.pool()).data().unwrap())?, 1, rows as i64),
        };
        let yd = hip.alloc(qk_full.len() * 4)?;
        hip.check(unsafe { sys::rten_hip_softmax_f32(hip.raw(), rows as i64, cols as i32, xd.ptr as *const f32, md.ptr as *const f32, div, modulo,
                                                    self.flush_nans_to_zero as i32, yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, out_shape.as_slice())?.into()].into())
    }
}

/// (div, mod) such that addend row of output row r is (r / div) % mod, when the addend's row axes are one run of full axes between
/// broadcast (size-1 / missing) axes; None otherwise.
fn addend_row_pattern(out: &[usize], m: &[usize]) -> Option<(i64, i64)> {
    let nd = out.len();
    if nd == 0 { return None; }
    let mut ms = vec![1usize; nd - m.len()];
    ms.extend_from_slice(m);
    let rows = &out[..nd - 1];
    let full: Vec<bool> = rows.iter().zip(&ms[..nd - 1]).map(|(o, a)| a == o && *o != 1).collect();
    let first = full.iter().position(|f| *f);
    let Some(first) = first else { return Some((rows.iter().product::<usize>().max(1) as i64, 1)); }; // one addend row for every output row
    let last = full.iter().rposition(|f| *f).unwrap();
    if rows[first..=last].iter().zip(&full[first..=last]).any(|(o, f)| !*f && *o != 1) { return None; } // a broadcast axis inside the run
    let div: usize = rows[last + 1..].iter().product();
    let modulo: usize = rows[first..=last].iter().product();
    Some((div.max(1) as i64, modulo.max(1) as i64))
}

hip_operator!(
    /// `LayerNormalization` (src/ops/norm.rs:456-529)
    HipLayerNormalization { axis: isize, epsilon: Option<f32> });
impl Operator for HipLayerNormalization {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let x: TensorView<f32> = ctx.inputs().require_as(0)?;
        let scale: TensorView<f32> = ctx.inputs().require_as(1)?;
        let bias: Option<TensorView<f32>> = ctx.inputs().get_as(2)?;
        let axis = ops::resolve_axis(x.ndim(), self.axis)?;
        let cols: usize = x.shape()[axis..].iter().product();
        if scale.len() != cols || bias.as_ref().map_or(false, |b| b.len() != cols) { return self.inner.run(ctx); } // broadcast forms: reference path
        let hip = &self.hip;
        let xd = hip.upload(&contiguous(ctx.pool(), &x))?;
        let gd = self.consts.get(hip, 1, &contiguous(ctx.pool(), &scale), 0, |_, raw| Ok(raw))?;
        let bd = match &bias { Some(b) => self.consts.get(hip, 2, &contiguous(ctx.pool(), b), 0, |_, raw| Ok(raw))?, None => null() };
        let yd = hip.alloc(x.len() * 4)?;
        hip.check(unsafe { sys::rten_hip_layer_norm_f32(hip.raw(), (x.len() / cols.max(1)) as i64, cols as i32, xd.ptr as *const f32, gd as *const f32, bd as *const f32,
                                                       1.0, 0.0, self.epsilon.unwrap_or(1e-5), yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, x.shape())?.into()].into())
    }
}

// ================================================================================================ element-wise operators
macro_rules! hip_unary {
    ($name:ident, $entry:ident) => {
        hip_operator!($name {});
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
hip_unary!(HipRelu, rten_hip_relu_f32);
hip_unary!(HipGelu, rten_hip_gelu_f32);
hip_unary!(HipErf, rten_hip_erf_f32);

/// Add / Mul with numpy broadcasting (src/ops/binary_elementwise.rs:58-170,476-560): f32 operands on the device (equal shapes and the
/// "b is a suffix of a" form through the flat entry point, everything else through rten_hip_binary_broadcast_f32 with 0-strides);
/// i32 operands stay on the reference's path.
macro_rules! hip_binary {
    ($name:ident, $flat:ident, $opcode:expr) => {
        hip_operator!($name {});
        impl Operator for $name {
            delegate_to_inner!();
            fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
                let (ValueView::FloatTensor(a), ValueView::FloatTensor(b)) = (ctx.inputs().require(0)?, ctx.inputs().require(1)?) else { return self.inner.run(ctx); };
                let out_shape = rten_tensor::broadcast_shapes(a.shape(), b.shape()).ok_or(OpError::IncompatibleInputShapes("Cannot broadcast inputs"))?;
                let hip = &self.hip;
                let (ad, bd) = (hip.upload(&contiguous(ctx.pool(), &a))?, hip.upload(&contiguous(ctx.pool(), &b))?);
                let n: usize = out_shape.iter().product();
                let yd = hip.alloc(n * 4)?;
                let suffix = a.shape() == out_shape.as_slice() && (b.len() == 1 || a.shape().ends_with(b.shape()));
                if suffix {
                    hip.check(unsafe { sys::$flat(hip.raw(), n as i64, ad.ptr as *const f32, bd.ptr as *const f32, b.len() as i64, yd.ptr as *mut f32) })?;
                } else {
                    let nd = out_shape.len();
                    let strides = |s: &[usize]| -> Vec<i64> { // contiguous strides of the upload, 0 on broadcast axes
                        let mut full = vec![1usize; nd - s.len()]; full.extend_from_slice(s);
                        let mut st = vec![0i64; nd]; let mut acc = 1i64;
                        for i in (0..nd).rev() { st[i] = if full[i] == 1 { 0 } else { acc }; acc *= full[i] as i64; }
                        st
                    };
                    let (sa, sb) = (strides(a.shape()), strides(b.shape()));
                    let shape64: Vec<i64> = out_shape.iter().map(|d| *d as i64).collect();
                    hip.check(unsafe { sys::rten_hip_binary_broadcast_f32(hip.raw(), $opcode, nd as i32, shape64.as_ptr(), sa.as_ptr(), sb.as_ptr(),
                                                                         ad.ptr as *const f32, bd.ptr as *const f32, yd.ptr as *mut f32) })?;
                }
                Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, out_shape.as_slice())?.into()].into())
            }
        }
    };
}
hip_binary!(HipAdd, rten_hip_add_f32, 0);
hip_binary!(HipMul, rten_hip_mul_f32, 1);

// ================================================================================================ pooling
#[derive(Clone, Debug)]
pub struct PoolAttrs { pub kernel_size: Vec<usize>, pub padding: Padding, pub strides: Vec<usize>, pub ceil_mode: bool, pub count_include_pad: bool }

/// pool_impl's checks in its order (src/ops/pooling.rs:174-242); 1-D inputs run as [N, C, 1, W].
fn pool_desc(x: &[usize], a: &PoolAttrs) -> Result<(sys::rten_hip_pool2d_desc, bool), OpError> {
    let spatial = x.len().saturating_sub(2);
    if a.kernel_size.len() != spatial { return Err(OpError::InvalidValue("kernel_size len does not match spatial dims")); }
    if a.strides.len() != spatial { return Err(OpError::InvalidValue("strides len does not match spatial dims")); }
    let (x4, ks, st, padding, one_d) = match spatial {
        1 => ([x[0], x[1], 1, x[2]], [1, a.kernel_size[0]], [1, a.strides[0]], a.padding.clone().expand_1d_to_2d()?, true),
        2 => ([x[0], x[1], x[2], x[3]], [a.kernel_size[0], a.kernel_size[1]], [a.strides[0], a.strides[1]], a.padding.clone(), false),
        _ => return Err(OpError::UnsupportedValue("Only inputs with 1 or 2 spatial dims are supported")),
    };
    let round = if a.ceil_mode { RoundMode::Ceil } else { RoundMode::Floor };
    let (oh, ow, pads) = calc_output_size_and_padding((x4[2], x4[3]), (ks[0], ks[1]), (st[0], st[1]), padding, None, round)?;
    Ok((sys::rten_hip_pool2d_desc { n: x4[0] as i32, c: x4[1] as i32, h: x4[2] as i32, w: x4[3] as i32, kh: ks[0] as i32, kw: ks[1] as i32,
                                    stride_h: st[0] as i32, stride_w: st[1] as i32, pads: [pads[0] as i32, pads[1] as i32, pads[2] as i32, pads[3] as i32],
                                    out_h: oh as i32, out_w: ow as i32, count_include_pad: a.count_include_pad as i32 }, one_d))
}

macro_rules! hip_pool {
    ($name:ident, $entry:ident) => {
        hip_operator!($name { attrs: PoolAttrs });
        impl Operator for $name {
            delegate_to_inner!();
            fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
                let x: TensorView<f32> = ctx.inputs().require_as(0)?;
                let (d, one_d) = pool_desc(x.shape(), &self.attrs)?;
                let hip = &self.hip;
                let xd = hip.upload(&contiguous(ctx.pool(), &x))?;
                let out4 = [d.n as usize, d.c as usize, d.out_h as usize, d.out_w as usize];
                let yd = hip.alloc(out4.iter().product::<usize>() * 4)?;
                hip.check(unsafe { sys::$entry(hip.raw(), &d, xd.ptr as *const f32, yd.ptr as *mut f32) })?;
                let shape = if one_d { vec![out4[0], out4[1], out4[3]] } else { out4.to_vec() };
                Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, &shape)?.into()].into())
            }
        }
    };
}
hip_pool!(HipMaxPool, rten_hip_max_pool2d_f32);         // src/ops/pooling.rs:560-650: accumulator -inf, padded taps skipped
hip_pool!(HipAveragePool, rten_hip_average_pool2d_f32); // src/ops/pooling.rs:395-470: sum / kernel_len or / non-padding count

hip_operator!(
    /// `GlobalAveragePool` (src/ops/pooling.rs:477-557): vecmath::Sum per channel / len
    HipGlobalAveragePool {});
impl Operator for HipGlobalAveragePool {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let x: TensorView<f32> = ctx.inputs().require_as(0)?;
        if x.ndim() < 2 { return Err(OpError::InvalidValue("Input must have at least 2 dims")); }
        let (n, c) = (x.size(0), x.size(1));
        let inner: usize = x.shape()[2..].iter().product();
        let mut out_shape = vec![n, c];
        out_shape.resize(x.ndim(), 1);
        let hip = &self.hip;
        let xd = hip.upload(&contiguous(ctx.pool(), &x))?;
        let yd = hip.alloc(n * c * 4)?;
        hip.check(unsafe { sys::rten_hip_global_average_pool_f32(hip.raw(), (n * c) as i64, inner as i32, xd.ptr as *const f32, yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, &out_shape)?.into()].into())
    }
}

// ================================================================================================ quantization
hip_operator!(
    /// `DynamicQuantizeLinear` (src/ops/quantize.rs:352-436)
    HipDynamicQuantizeLinear {});
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

// ================================================================================================ attention
hip_operator!(
    /// `Attention` (src/ops/attention.rs:640-905), the scaled-dot-product part: softmax(scale * Q K^T + mask) V per (batch, head), one
    /// `rten_hip_sdpa_f32` call.  3-D inputs [B, S, heads * d] are read IN PLACE through the head / row strides (split_attention_heads and
    /// merge_attention_heads become strides: no copies); 4-D inputs are [B, heads, S, d].  What the kernel does not cover -- causal
    /// masking, softcap, boolean masks, past key / value inputs, grouped-query heads, the extra outputs -- takes the reference's path.
    HipAttention { is_causal: bool, kv_num_heads: Option<u32>, q_num_heads: Option<u32>, scale: Option<f32>, softcap: f32 });
impl Operator for HipAttention {
    delegate_to_inner!();
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let inputs = ctx.inputs();
        let q: TensorView<f32> = inputs.require_as(0)?;
        let k: TensorView<f32> = inputs.require_as(1)?;
        let v: TensorView<f32> = inputs.require_as(2)?;
        let fallback = self.is_causal || self.softcap > 0.0 || inputs.get(4).is_some() || inputs.get(5).is_some() || inputs.get(6).is_some()
            || ctx.outputs().is_used(1) || ctx.outputs().is_used(2) || matches!(inputs.get(3), Some(ValueView::Int32Tensor(_)));
        if fallback { return self.inner.run(ctx); }
        if ctx.outputs().is_used(3) { return Err(OpError::UnsupportedValue("qk_matmul_output output is not supported")); }
        let input_3d = match q.ndim() { 3 => true, 4 => false, _ => return Err(OpError::InvalidValue("query must have 3 or 4 dimensions")) };
        if k.ndim() != q.ndim() || v.ndim() != q.ndim() { return Err(OpError::IncompatibleInputShapes("query, key and value must have the same rank")); }
        // checks of attention.rs:696-775, same order and messages
        let (batch, heads, kv_heads, s, t, d, dv, qst, kst, vst) = if input_3d {
            let qh = self.q_num_heads.ok_or(OpError::InvalidValue("q_num_heads is required for 3D inputs"))? as usize;
            let kh = self.kv_num_heads.ok_or(OpError::InvalidValue("kv_num_heads is required for 3D inputs"))? as usize;
            if qh == 0 || kh == 0 { return Err(OpError::InvalidValue("q_num_heads and kv_num_heads must be positive")); }
            let (b, s, qhid) = (q.size(0), q.size(1), q.size(2));
            let (kb, t, khid) = (k.size(0), k.size(1), k.size(2));
            let (vb, vt, vhid) = (v.size(0), v.size(1), v.size(2));
            if kb != b || vb != b { return Err(OpError::IncompatibleInputShapes("query, key and value must have the same batch size")); }
            if t != vt { return Err(OpError::IncompatibleInputShapes("key and value must have the same sequence length")); }
            if qhid % qh != 0 { return Err(OpError::IncompatibleInputShapes("query hidden size must be divisible by q_num_heads")); }
            let d = qhid / qh;
            if khid % kh != 0 || vhid % kh != 0 { return Err(OpError::IncompatibleInputShapes("key/value hidden size must be divisible by kv_num_heads")); }
            if khid / kh != d { return Err(OpError::IncompatibleInputShapes("key head size must match query head size")); }
            let dv = vhid / kh;
            // [B, S, heads * d]: batch stride S * hidden, head stride d, row stride hidden
            (b, qh, kh, s, t, d, dv, ((s * qhid) as i64, d as i64, qhid as i64), ((t * khid) as i64, d as i64, khid as i64), ((t * vhid) as i64, dv as i64, vhid as i64))
        } else {
            let (b, qh, s, d) = (q.size(0), q.size(1), q.size(2), q.size(3));
            let (kb, kh, t, kd) = (k.size(0), k.size(1), k.size(2), k.size(3));
            let (vb, vh, vt, dv) = (v.size(0), v.size(1), v.size(2), v.size(3));
            if kb != b || vb != b { return Err(OpError::IncompatibleInputShapes("query, key and value must have the same batch size")); }
            if kh != vh || t != vt { return Err(OpError::IncompatibleInputShapes("key and value must have the same number of heads and sequence length")); }
            if kd != d { return Err(OpError::IncompatibleInputShapes("key head size must match query head size")); }
            (b, qh, kh, s, t, d, dv, ((qh * s * d) as i64, (s * d) as i64, d as i64), ((kh * t * d) as i64, (t * d) as i64, d as i64), ((kh * t * dv) as i64, (t * dv) as i64, dv as i64))
        };
        if heads == 0 || kv_heads == 0 || heads % kv_heads != 0 { return Err(OpError::IncompatibleInputShapes("q_num_heads must be a positive multiple of kv_num_heads")); }
        if heads != kv_heads { return self.inner.run(ctx); } // grouped-query attention: reference path
        let scale = self.scale.unwrap_or_else(|| 1.0 / (d as f32).sqrt());
        let target = [batch, heads, s, t];
        let hip = &self.hip;
        // float mask: broadcast to [B, heads, S, T] as the reference does; the ABI takes a batch stride and a row stride (0 = broadcast)
        let (maskd, mask_bs, mask_rs) = match inputs.get(3) {
            None => (None, 0i64, 0i64),
            Some(ValueView::FloatTensor(m)) => {
                let mb = m.try_broadcast(target).map_err(|_| OpError::IncompatibleInputShapes("Cannot broadcast inputs"))?;
                // per-head masks are materialised per (batch, head) by the reference too; the kernel's mask is shared by the heads of a batch entry
                let mut ms = vec![1usize; 4 - m.ndim().min(4)]; ms.extend_from_slice(m.shape());
                if ms[1] != 1 && heads != 1 { return self.inner.run(ctx); }
                let per_batch = ms[0] != 1; let per_row = ms[2] != 1;
                let dense = mb.slice((.., 0)).to_tensor_in(ctx.pool()); // [B, S, T]
                let _ = (per_batch, per_row);
                (Some(hip.upload(dense.data().unwrap())?), (s * t) as i64, t as i64)
            }
            Some(_) => return Err(OpError::InvalidValue("attn_mask must have a float or bool (int32) type")),
        };
        let (qd, kd, vd) = (hip.upload(&contiguous(ctx.pool(), &q))?, hip.upload(&contiguous(ctx.pool(), &k))?, hip.upload(&contiguous(ctx.pool(), &v))?);
        // output in the input's layout: 3-D -> [B, S, heads * dv] (merge_attention_heads as strides), 4-D -> [B, heads, S, dv]
        let (o_bs, o_hs, o_rs, out_shape) = if input_3d { ((s * heads * dv) as i64, dv as i64, (heads * dv) as i64, vec![batch, s, heads * dv]) }
                                            else { ((heads * s * dv) as i64, (s * dv) as i64, dv as i64, vec![batch, heads, s, dv]) };
        let desc = sys::rten_hip_sdpa_desc { batch: batch as i32, heads: heads as i32, s: s as i32, t: t as i32, d: d as i32, dv: dv as i32,
                                             q_bs: qst.0, q_hs: qst.1, q_rs: qst.2, k_bs: kst.0, k_hs: kst.1, k_rs: kst.2, v_bs: vst.0, v_hs: vst.1, v_rs: vst.2,
                                             o_bs, o_hs, o_rs, mask_batch_stride: mask_bs, mask_row_stride: mask_rs, scale, flush_nan_to_zero: 1 /* sdpa_head flushes (attention.rs:551) */ };
        let yd = hip.alloc(batch * heads * s * dv * 4)?;
        hip.check(unsafe { sys::rten_hip_sdpa_f32(hip.raw(), &desc, qd.ptr as *const f32, kd.ptr as *const f32, vd.ptr as *const f32,
                                                 opt_ptr(&maskd) as *const f32, yd.ptr as *mut f32) })?;
        Ok([download_tensor::<f32>(hip, ctx.pool(), &yd, &out_shape)?.into()].into())
    }
}

// ================================================================================================ registration
/// Value of a padded tap of an integer convolution: the x86 reference's im2col writes raw 0 in the shifted domain.
pub const DEFAULT_INT8_PAD_MODE: i32 = sys::RTEN_HIP_PAD_RAW0_I8;

fn conv_attrs(groups: usize, dilations: &[usize], padding: &Padding, strides: &[usize]) -> ConvAttrs {
    ConvAttrs { groups, dilations: dilations.to_vec(), padding: padding.clone(), strides: strides.to_vec() }
}

/// The HIP-backed stand-in for one operator of an optimised graph, or None when the backend does not accelerate it.  The operator is
/// recognised by `Any` downcast (as the fusion passes do, src/optimize/fusions.rs:1052) and its attributes are copied out; the original
/// stays inside the wrapper for `name` / `output_types` / shape inference and for the forms the device path hands back to it.
pub fn accelerate(op: &DynOp, hip: &Arc<HipContext>) -> Option<DynOp> {
    macro_rules! wrap {
        ($op:ty, $hip_op:ident, |$o:ident| { $($field:ident : $value:expr),* $(,)? }) => {
            if let Some($o) = op.downcast_ref::<$op>() {
                let _ = $o;
                return Some(Arc::new($hip_op { inner: op.clone(), hip: hip.clone(), consts: ConstCache::new(), $($field: $value),* }));
            }
        };
    }
    wrap!(ops::Conv, HipConv, |o| { attrs: conv_attrs(o.groups, &o.dilations, &o.padding, &o.strides) });
    wrap!(ops::ConvInteger, HipConvInteger, |o| { attrs: conv_attrs(o.groups, &o.dilations, &o.padding, &o.strides), pad_mode: DEFAULT_INT8_PAD_MODE });
    // `ConvIntegerToFloat::conv` and `MatMulIntegerToFloat::matmul` are private fields: the integration adds the accessor
    // `pub fn conv(&self) -> &ConvInteger` (one line, src/ops/conv.rs:545) -- the second and last change to the reference.
    wrap!(ops::ConvIntegerToFloat, HipConvIntegerToFloat, |o| { attrs: conv_attrs(o.conv().groups, &o.conv().dilations, &o.conv().padding, &o.conv().strides),
                                                                pad_mode: DEFAULT_INT8_PAD_MODE });
    wrap!(ops::MatMul, HipMatMul, |o| {});
    wrap!(ops::FusedMatMul, HipFusedMatMul, |o| { alpha: o.alpha });
    wrap!(ops::Gemm, HipGemm, |o| { alpha: o.alpha, beta: o.beta, transpose_a: o.transpose_a, transpose_b: o.transpose_b });
    wrap!(ops::MatMulInteger, HipMatMulInteger, |o| {});
    wrap!(ops::MatMulIntegerToFloat, HipMatMulIntegerToFloat, |o| {});
    wrap!(ops::Softmax, HipSoftmax, |o| { axis: o.axis, flush_nans_to_zero: o.flush_nans_to_zero });
    wrap!(ops::AddSoftmax, HipAddSoftmax, |o| { flush_nans_to_zero: o.flush_nans_to_zero });
    wrap!(ops::LayerNormalization, HipLayerNormalization, |o| { axis: o.axis, epsilon: o.epsilon });
    wrap!(ops::Relu, HipRelu, |o| {});
    wrap!(ops::Gelu, HipGelu, |o| {});
    wrap!(ops::Erf, HipErf, |o| {});
    wrap!(ops::Add, HipAdd, |o| {});
    wrap!(ops::Mul, HipMul, |o| {});
    wrap!(ops::MaxPool, HipMaxPool, |o| { attrs: PoolAttrs { kernel_size: o.kernel_size.to_vec(), padding: o.padding.clone(), strides: o.strides.to_vec(),
                                                             ceil_mode: o.ceil_mode, count_include_pad: false } });
    wrap!(ops::AveragePool, HipAveragePool, |o| { attrs: PoolAttrs { kernel_size: o.kernel_size.to_vec(), padding: o.padding.clone(), strides: o.strides.to_vec(),
                                                                     ceil_mode: o.ceil_mode, count_include_pad: o.count_include_pad } });
    wrap!(ops::GlobalAveragePool, HipGlobalAveragePool, |o| {});
    wrap!(ops::DynamicQuantizeLinear, HipDynamicQuantizeLinear, |o| {});
    wrap!(ops::Attention, HipAttention, |o| { is_causal: o.is_causal, kv_num_heads: o.kv_num_heads, q_num_heads: o.q_num_heads, scale: o.scale, softcap: o.softcap });
    None
}

/// Install the backend on a model about to be loaded.
///
/// The ONE hook the integration adds to the reference: `ModelOptions::set_operator_rewriter(Box<dyn Fn(&DynOp) -> Option<DynOp> + Send + Sync>)`,
/// applied by `Model::load` to every operator node of the graph AFTER `GraphOptimizer::optimize` has run (src/model.rs, where the optimised
/// graph is finalised; sub-graphs of control-flow operators included) -- a node whose rewriter returns `Some(op)` has its `Arc<dyn Operator>`
/// replaced (src/graph/node.rs:137), nothing else about the node changes.  Op names, attribute structs, deserialisation (`ReadOp`), the
/// optimiser, the planner and `Model::run` stay the registry's / the reference's own.
pub fn register(opts: &mut ModelOptions, hip: Arc<HipContext>) {
    opts.set_operator_rewriter(Box::new(move |op| accelerate(op, &hip)));
}
