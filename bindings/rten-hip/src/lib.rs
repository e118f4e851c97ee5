//! rten-hip: the reference's `Operator` interface (src/operator.rs:486-613) implemented over librten_hip.so.
//!
//! NOT COMPILED in the build image (no Rust toolchain there); the `extern "C"` layer it calls, `rten-hip-sys`, is generated
//! from include/rten_hip.h and checked against it by tests/test_abi.py.  What is tested in this repository is the same layer
//! written in C++ (include/rten_hip_ops.hpp) and Python (rten_amd/ops.py).
//!
//! Design constraint: the reference's surface stays IDENTICAL.  No new `Value` variant, no new accessor on `OpRunContext`, no
//! change to `PrepackedInput`:
//!   * every operator here wraps the reference's own operator struct (`inner`) and delegates `name`, `max_inputs`,
//!     `output_types`, `as_infer_shapes` to it -- attributes, shape inference and the planner see the operators they know;
//!   * `run` takes the reference's host `ValueView`s and returns host `Value`s allocated from `ctx.pool()`;
//!   * constant inputs (weights) are staged on the device ONCE: `PrepackedInput` is a closed enum of rten-gemm types, so the
//!     device copy lives in the OPERATOR INSTANCE that uses it (`ConstCache`, one slot per input): an operator belongs to one
//!     graph, so the copies die with the model that owns them, and a slot is re-staged if the constant behind it is ever a
//!     different allocation (graph constants do not move while the model is alive -- `Graph` owns them, src/graph.rs:488-562);
//!   * activations cross PCIe per operator in this drop-in form.  Residency (values stay in HBM between operators, SURVEY 8(f)
//!     rank 1) would need a `Value::Device` variant -- a change to the reference -- at operator granularity; at SUBGRAPH granularity
//!     it needs nothing: `HipSubgraph` (subgraph.rs) is one `Operator` that owns the C++ plan executor behind the C ABI
//!     (`rten_hip_model_*`), the path `bench.py --via-executor` measures.
//!   * one `HipContext` may be shared by every thread that calls `Model::run(&self)`: the C ABI locks per call.
//!
//! Registration (what a user writes):
//! ```ignore
//! let hip = HipContext::new(0)?;                             // Err(NoDevice) without an MI355X: there is no CPU fallback inside the backend
//! let mut opts = ModelOptions::with_all_ops();               // src/model.rs:679-700
//! rten_hip::register(&mut opts, hip);                        // post-optimisation operator rewrite (ops.rs, `register`)
//! let model = opts.load_file("resnet50.onnx")?;
//! let out = model.run_one(input.view().into(), None)?;      // unchanged call site; rten-cli / rten-examples unchanged
//! ```
//! The operators are wrapped AFTER the reference's optimiser has run, so its fusion passes (which recognise operators by `Any`
//! downcast) see the operators they know and the backend sees `ConvIntegerToFloat` / `FusedMatMul` / `AddSoftmax` / `MatMulIntegerToFloat`.
//! Two additions to the reference, both listed in INTEGRATION.md section 2.1: `ModelOptions::set_operator_rewriter` and the accessors
//! `ConvIntegerToFloat::conv()` / `MatMulIntegerToFloat::matmul()` for fields that are private today.
mod install;
mod ops;
mod subgraph;
pub use install::{install_resident, load_resident, ResidentPlan};
pub use subgraph::{HipSubgraph, HipSubgraphPool};

use std::collections::HashMap;
use std::ffi::{c_void, CStr};
use std::ptr;
use std::sync::{Arc, Mutex};

use rten::ops::OpError;
use rten_hip_sys as sys;

pub use ops::{accelerate, register};

/// RAII over `rten_hip_ctx`.  `Send + Sync`: the C ABI serialises entry per call (rten_hip.h, "Thread safety").
pub struct HipContext {
    raw: *mut sys::rten_hip_ctx,
}
impl std::fmt::Debug for HipContext { // `Operator: Debug` (src/operator.rs:486): every wrapper derives it
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "HipContext({:p})", self.raw) }
}
unsafe impl Send for HipContext {}
unsafe impl Sync for HipContext {}

#[derive(Debug)]
pub enum HipInitError {
    /// no gfx950 device / library unusable: the backend has no CPU fallback
    NoDevice,
    Hip(String),
}

impl HipContext {
    pub fn new(device: i32) -> Result<Arc<HipContext>, HipInitError> {
        let mut raw = ptr::null_mut();
        match unsafe { sys::rten_hip_init(device, ptr::null_mut(), &mut raw) } {
            sys::RTEN_HIP_OK => Ok(Arc::new(HipContext { raw })),
            sys::RTEN_HIP_ERR_NO_DEVICE => Err(HipInitError::NoDevice),
            _ => Err(HipInitError::Hip("rten_hip_init failed".into())),
        }
    }

    pub fn raw(&self) -> *mut sys::rten_hip_ctx {
        self.raw
    }

    /// How one-row matrix products are accumulated (rten_hip.h, `rten_hip_set_gemv_order`).  The reference takes its
    /// vector-matrix kernels when the LHS has one row and the RHS is not prepacked (rten-gemm/src/lib.rs:876-891); a model
    /// loaded with `ModelOptions::prepack_weights(true)` passes `prepacked = true` here once, any other model may pass the
    /// size of its thread pool (`threads`; 0 = at least n / 128), which decides the reference's column blocks.
    pub fn set_gemv_order(&self, prepacked: bool, threads: u32) -> Result<(), OpError> {
        self.check(unsafe { sys::rten_hip_set_gemv_order(self.raw, if prepacked { 0 } else { 1 }, threads as i32) })
    }

    /// Status code -> the reference's `OpError` (include/rten_hip.h, "status codes").  A HIP runtime failure is not an
    /// operator error and never turns into a silent CPU fallback.
    pub fn check(&self, status: i32) -> Result<(), OpError> {
        let msg = || -> &'static str {
            // OpError carries &'static str: the ABI's validation messages are static strings of the library
            let p = unsafe { sys::rten_hip_last_error(self.raw) };
            Box::leak(unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned().into_boxed_str())
        };
        match status {
            sys::RTEN_HIP_OK => Ok(()),
            sys::RTEN_HIP_ERR_INVALID_VALUE => Err(OpError::InvalidValue(msg())),
            sys::RTEN_HIP_ERR_INCOMPATIBLE_SHAPES => Err(OpError::IncompatibleInputShapes(msg())),
            sys::RTEN_HIP_ERR_UNSUPPORTED => Err(OpError::UnsupportedValue(msg())),
            _ => panic!("HIP failure in librten_hip.so: {}", msg()),
        }
    }

    pub fn alloc(self: &Arc<Self>, bytes: usize) -> Result<DeviceBuffer, OpError> {
        let mut p: *mut c_void = ptr::null_mut();
        self.check(unsafe { sys::rten_hip_malloc(self.raw, bytes, &mut p) })?;
        Ok(DeviceBuffer { ptr: p, bytes, ctx: self.clone() })
    }

    /// Host slice -> fresh device buffer (activations of the drop-in form).
    pub fn upload<T: Copy>(self: &Arc<Self>, host: &[T]) -> Result<DeviceBuffer, OpError> {
        let buf = self.alloc(std::mem::size_of_val(host))?;
        self.check(unsafe { sys::rten_hip_memcpy_h2d(self.raw, buf.ptr, host.as_ptr() as *const c_void, buf.bytes) })?;
        Ok(buf)
    }

    pub fn download<T: Copy>(&self, buf: &DeviceBuffer, host: &mut [T]) -> Result<(), OpError> {
        debug_assert!(std::mem::size_of_val(host) <= buf.bytes);
        self.check(unsafe { sys::rten_hip_memcpy_d2h(self.raw, host.as_mut_ptr() as *mut c_void, buf.ptr, std::mem::size_of_val(host)) })
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        // every DeviceBuffer holds an Arc of its context, so the last buffer is gone by now; the context owns none itself (no cycle)
        unsafe { sys::rten_hip_destroy(self.raw) };
    }
}

/// Device copies of ONE operator's constant inputs (weights, biases, scales), created on first use: uploaded, then handed to `stage`,
/// which returns either the upload itself (`|_, raw| Ok(raw)`) or a re-laid copy (rten_hip_conv2d_f32_prepack,
/// rten_hip_gemm_int8_prepack).  The backend's analogue of `Graph::prepack_weights` + `WeightCache`, owned by the operator instance:
/// dropped with the model, never shared between models, and a slot whose constant is no longer the allocation it was staged from
/// (address or length changed) is staged again instead of serving stale weights.
pub struct ConstCache {
    slots: Mutex<HashMap<(usize, u32), (usize, usize, DeviceBuffer)>>, // (input index, layout tag) -> (host address, bytes, device copy)
}

impl std::fmt::Debug for ConstCache {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "ConstCache({} staged)", self.slots.lock().map(|s| s.len()).unwrap_or(0)) }
}

impl ConstCache {
    pub fn new() -> Self {
        ConstCache { slots: Mutex::new(HashMap::new()) }
    }

    pub fn get<T: Copy>(
        &self,
        hip: &Arc<HipContext>,
        input: usize,
        host: &[T],
        layout: u32,
        stage: impl FnOnce(&Arc<HipContext>, DeviceBuffer) -> Result<DeviceBuffer, OpError>,
    ) -> Result<*const c_void, OpError> {
        let id = (host.as_ptr() as usize, std::mem::size_of_val(host));
        let mut slots = self.slots.lock().unwrap();
        if let Some((addr, len, buf)) = slots.get(&(input, layout)) {
            if (*addr, *len) == id {
                return Ok(buf.ptr as *const c_void);
            }
        }
        let raw = hip.upload(host)?;
        let staged = stage(hip, raw)?; // by value: `stage` either returns it or frees it after re-laying
        let p = staged.ptr as *const c_void;
        slots.insert((input, layout), (id.0, id.1, staged)); // a stale entry is dropped (and freed) here
        Ok(p)
    }
}

/// Device allocation freed through its context.
pub struct DeviceBuffer {
    pub ptr: *mut c_void,
    pub bytes: usize,
    ctx: Arc<HipContext>,
}
unsafe impl Send for DeviceBuffer {}

impl Drop for DeviceBuffer {
    fn drop(&mut self) {
        unsafe { sys::rten_hip_free(self.ctx.raw, self.ptr) };
    }
}

/// Weight broadcast of a batch-sharded deployment: one process per GPU, rank 0 stages the weight arena, the
/// other ranks receive it over xGMI.  `id` travels from rank 0 to the others by any host channel.
pub struct Communicator {
    raw: *mut sys::rten_hip_comm,
    ctx: Arc<HipContext>,
}

impl Communicator {
    pub fn unique_id(ctx: &HipContext) -> Result<[u8; 128], OpError> {
        let mut id = [0u8; 128];
        ctx.check(unsafe { sys::rten_hip_comm_get_unique_id(ctx.raw, id.as_mut_ptr()) })?;
        Ok(id)
    }
    pub fn new(ctx: Arc<HipContext>, id: &[u8; 128], world_size: i32, rank: i32) -> Result<Self, OpError> {
        let mut raw = ptr::null_mut();
        ctx.check(unsafe { sys::rten_hip_comm_init_rank(ctx.raw, id.as_ptr(), world_size, rank, &mut raw) })?;
        Ok(Communicator { raw, ctx })
    }
    pub fn broadcast(&self, buf: &DeviceBuffer, root: i32) -> Result<(), OpError> {
        self.ctx.check(unsafe { sys::rten_hip_broadcast(self.ctx.raw, self.raw, buf.ptr, buf.bytes, root) })
    }
}

impl Drop for Communicator {
    fn drop(&mut self) {
        unsafe { sys::rten_hip_comm_destroy(self.ctx.raw, self.raw) };
    }
}
