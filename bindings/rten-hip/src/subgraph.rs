//! `HipSubgraph`: ONE `Operator` that owns a whole device-resident subgraph.
//!
//! The per-operator drop-in of ops.rs moves every activation across PCIe twice.  Keeping values in HBM between operators would need a
//! `Value::Device` variant -- a change to the reference -- unless a maximal run of accelerated nodes is handed to the backend as a single
//! operator: the reference's own precedent for an operator that owns a graph is `SubgraphOperator` (src/operator.rs:630-646; `If` / `Loop`).
//! The graph behind this operator is the C++ plan executor exported through the C ABI as `rten_hip_model_*` (include/rten_hip.h;
//! rten_amd/csrc/graph_abi.cpp): ONNX bytes in, constants uploaded and prepacked once, the reference's fusions applied, a committed launch plan
//! (profiles/plans/*.json) by step name, the batch run as `chains` independent dim-0 slices on their own streams, each a hipGraph.  `bench.py`
//! (its default path since round 5) measures exactly this path (ResNet-50 f32 batch 32 as 4 chains on one replica: 2.68 ms, the hand-planned runner the
//! same within 0.01 ms on the same box, same logits; the default line -- one chain per replica, four replicas, plans chosen under co-run -- 2.39-2.44 ms: DESIGN.md section 2).
//! `install.rs` puts one of these in place of a loaded model's whole graph (`rten_hip::load_resident`).
//!
//! NOT COMPILED in the build image -- see lib.rs.  tests/test_abi.py checks that every `sys::` name used here exists in the generated -sys crate
//! with the arity used.
use std::ffi::{c_void, CStr, CString};
use std::ptr;
use std::sync::atomic::{AtomicUsize, Ordering};
use std::sync::{Arc, Mutex};

use rten::ops::{OpError, OpRunContext, Operator, OutputList, OutputType, OutputTypeList, OutputTypesContext};
use rten::{DataType, ValueType};
use rten_hip_sys as sys;
use rten_tensor::prelude::*;
use rten_tensor::{Tensor, TensorView};

use crate::HipContext;

/// One bound input: its name in the subgraph, the FULL-batch shape it was bound with and where the backend wants its bytes.
#[derive(Debug)]
struct BoundInput {
    name: String,
    shape: Vec<usize>,
    dev: *mut c_void,
    dtype: i32, // RTEN_HIP_DTYPE_* (rten_hip_model_input_dtype): the element type the device graph reads
}

/// A resident subgraph as an operator.  Inputs are tensors of the shapes given to `load` (a static plan: the launch plan, the buffer plan and
/// the captured hipGraphs are per shape) and of the element types the graph declares -- f32, i32 (ONNX int32 / int64 inputs: token ids, masks,
/// positions of BERT-class graphs), u8, i8 (`rten_hip_model_input_dtype`); outputs come back with their own types (`rten_hip_model_output_dtype`).
#[derive(Debug)]
pub struct HipSubgraph {
    model: *mut sys::rten_hip_model,
    hip: Arc<HipContext>, // chain 0 runs on this context's stream: the context must outlive the model (rten_hip.h), which this Arc guarantees
    inputs: Vec<BoundInput>,
    n_outputs: usize,
    output_dtypes: Vec<i32>, // RTEN_HIP_DTYPE_* per output (known after prepare)
    run_lock: Mutex<()>, // the model object is not thread-safe (one caller at a time); `Model::run(&self)` may be called from several threads
    origin: Option<Arc<HipSubgraph>>, // a replica (rten_hip_model_clone) shares its origin's weights: the origin outlives it
}
unsafe impl Send for HipSubgraph {}
unsafe impl Sync for HipSubgraph {}

impl HipSubgraph {
    /// `onnx`: the serialized `ModelProto` of the subgraph (the loader already holds the bytes).  `plan_json`: a launch-plan file
    /// (`{step: [variant, split mode, K groups, order]}`, optionally keyed by sub-batch size; `{"qout": [...]}` opts int8 edges into the
    /// quantized-output launch) or `None` for the backend's defaults.  `chains`: independent dim-0 slices run side by side (4 for the f32
    /// ResNet-50 at batch 32; 1 for dynamically quantized graphs, whose DynamicQuantizeLinear statistics span the batch).
    pub fn load(hip: Arc<HipContext>, onnx: &[u8], plan_json: Option<&str>, chains: i32, input_shapes: &[(&str, Vec<usize>)]) -> Result<Self, OpError> {
        let plan = plan_json.map(|p| CString::new(p).map_err(|_| OpError::InvalidValue("launch plan contains a NUL byte"))).transpose()?;
        let mut model: *mut sys::rten_hip_model = ptr::null_mut();
        // `_load_ex`: every chain is created on the device of `hip` (the round-4 entry point took a separate device id, and this binding passed 0
        // whatever device the context lived on); flags 0 = this process uploads the weights itself
        let status = unsafe {
            sys::rten_hip_model_load_ex(hip.raw(), onnx.as_ptr() as *const c_void, onnx.len(), plan.as_ref().map_or(ptr::null(), |p| p.as_ptr()), chains, 0, &mut model)
        };
        if status != 0 {
            // a failed load has no model object: the reason (parse error, operator outside the backend's registry, bad plan file, a batch-coupled
            // graph with chains > 1) is per calling thread
            let why = unsafe { CStr::from_ptr(sys::rten_hip_model_load_error()) }.to_string_lossy().into_owned();
            if !why.is_empty() { eprintln!("rten-hip: subgraph: {why}"); }
            hip.check(status)?;
        }
        let mut this = HipSubgraph { model, hip, inputs: Vec::new(), n_outputs: 0, output_dtypes: Vec::new(), run_lock: Mutex::new(()), origin: None };
        this.bind_and_prepare(input_shapes)?;
        Ok(this)
    }

    /// Inputs bound to their full-batch shapes, launch plan applied, one hipGraph per chain captured.
    fn bind_and_prepare(&mut self, input_shapes: &[(&str, Vec<usize>)]) -> Result<(), OpError> {
        let this = self;
        let (mut n_in, mut n_out, mut n_steps, mut n_planned) = (0i32, 0i32, 0i32, 0i32);
        this.check(unsafe { sys::rten_hip_model_info(this.model, &mut n_in, &mut n_out, &mut n_steps, &mut n_planned) })?;
        this.n_outputs = n_out as usize;
        for i in 0..n_in {
            let name = unsafe { CStr::from_ptr(sys::rten_hip_model_input_name(this.model, i)) }.to_string_lossy().into_owned();
            let shape = input_shapes.iter().find(|(n, _)| *n == name).map(|(_, s)| s.clone()).ok_or(OpError::MissingInputs)?;
            let dims: Vec<i64> = shape.iter().map(|&d| d as i64).collect();
            let mut dev: *mut c_void = ptr::null_mut();
            this.check(unsafe { sys::rten_hip_model_bind_input(this.model, i, dims.as_ptr(), dims.len() as i32, &mut dev) })?;
            let mut dtype = sys::RTEN_HIP_DTYPE_F32;
            this.check(unsafe { sys::rten_hip_model_input_dtype(this.model, i, &mut dtype) })?;
            this.inputs.push(BoundInput { name, shape, dev, dtype });
        }
        this.check(unsafe { sys::rten_hip_model_prepare(this.model, 0) })?; // buffers planned, launch plan applied, one hipGraph per chain captured
        for i in 0..n_out {
            let mut dtype = sys::RTEN_HIP_DTYPE_F32;
            this.check(unsafe { sys::rten_hip_model_output_dtype(this.model, i, &mut dtype) })?;
            this.output_dtypes.push(dtype);
        }
        Ok(())
    }

    /// Another REPLICA of this subgraph on `hip` (a context -- a stream -- of its own on the same device): the same graph and plan, its own buffers and
    /// hipGraphs, THIS subgraph's constants and prepacked weights (`rten_hip_model_clone`).  Independent batches handed to different replicas overlap
    /// on the device: the batch-level analogue of sub-batch chains, and the only one a batch-coupled graph (the dynamically quantized ResNet-50) can
    /// use.  Measured (`bench.py --lanes`): int8 ResNet-50 1.50 -> 0.90 ms per batch of 32 at 4 replicas, f32 2.70 -> 2.40 ms at 4 (with the lanes plan), BERT-base 6.67 -> 5.85.
    pub fn replica(self: &Arc<Self>, hip: Arc<HipContext>) -> Result<HipSubgraph, OpError> {
        let origin = self.origin.clone().unwrap_or_else(|| self.clone());
        let mut model: *mut sys::rten_hip_model = ptr::null_mut();
        let status = unsafe { sys::rten_hip_model_clone(origin.model, hip.raw(), &mut model) };
        if status != 0 {
            let why = unsafe { CStr::from_ptr(sys::rten_hip_model_load_error()) }.to_string_lossy().into_owned();
            if !why.is_empty() { eprintln!("rten-hip: subgraph replica: {why}"); }
            hip.check(status)?;
        }
        let shapes: Vec<(&str, Vec<usize>)> = origin.inputs.iter().map(|b| (b.name.as_str(), b.shape.clone())).collect();
        let mut this = HipSubgraph { model, hip, inputs: Vec::new(), n_outputs: 0, output_dtypes: Vec::new(), run_lock: Mutex::new(()), origin: Some(origin.clone()) };
        this.bind_and_prepare(&shapes)?;
        Ok(this)
    }

    /// Names of the subgraph's inputs in the order `run` expects them (the model file's declaration order), and its output count.
    pub fn input_names(&self) -> impl Iterator<Item = &str> { self.inputs.iter().map(|b| b.name.as_str()) }
    pub fn num_outputs(&self) -> usize { self.n_outputs }

    /// The backend's status codes as `OpError`s, with the model's own message where it has one (the text is logged: `OpError` carries
    /// `&'static str`s).
    fn check(&self, status: i32) -> Result<(), OpError> {
        if status != 0 {
            let msg = unsafe { CStr::from_ptr(sys::rten_hip_model_last_error(self.model)) }.to_string_lossy().into_owned();
            if !msg.is_empty() { eprintln!("rten-hip: subgraph: {msg}"); }
        }
        self.hip.check(status)
    }
}

impl Drop for HipSubgraph {
    fn drop(&mut self) {
        unsafe { sys::rten_hip_model_destroy(self.model) }; // before `hip` and `origin` (fields) are released: a replica goes before its origin
    }
}

/// N replicas of one resident subgraph behind ONE operator: every `run` takes the next lane round robin; each lane has its own lock, so `Model::run`
/// callers on different threads run side by side on the device (lanes) instead of queueing behind one model object.
#[derive(Debug)]
pub struct HipSubgraphPool {
    lanes: Vec<Arc<HipSubgraph>>,
    next: AtomicUsize,
}

impl HipSubgraphPool {
    /// `first`: a loaded subgraph; `contexts`: one further `HipContext` (same device) per additional lane.
    pub fn new(first: HipSubgraph, contexts: Vec<Arc<HipContext>>) -> Result<Self, OpError> {
        let first = Arc::new(first);
        let mut lanes = vec![first.clone()];
        for hip in contexts { lanes.push(Arc::new(first.replica(hip)?)); }
        Ok(HipSubgraphPool { lanes, next: AtomicUsize::new(0) })
    }
}

impl Operator for HipSubgraphPool {
    fn name(&self) -> &str { "HipSubgraphPool" }
    fn max_inputs(&self) -> Option<usize> { self.lanes[0].max_inputs() }
    fn max_outputs(&self) -> Option<usize> { self.lanes[0].max_outputs() }
    fn output_types(&self, ctx: &OutputTypesContext) -> Option<OutputTypeList> { self.lanes[0].output_types(ctx) }
    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let lane = self.next.fetch_add(1, Ordering::Relaxed) % self.lanes.len();
        self.lanes[lane].run(ctx)
    }
}

impl Operator for HipSubgraph {
    fn name(&self) -> &str { "HipSubgraph" }
    fn max_inputs(&self) -> Option<usize> { Some(self.inputs.len()) }
    fn max_outputs(&self) -> Option<usize> { Some(self.n_outputs) }
    fn output_types(&self, _ctx: &OutputTypesContext) -> Option<OutputTypeList> {
        Some(self.output_dtypes.iter().map(|&d| OutputType::Fixed(ValueType::Tensor(match d {
            sys::RTEN_HIP_DTYPE_I32 => DataType::Int32,
            sys::RTEN_HIP_DTYPE_U8 => DataType::UInt8,
            sys::RTEN_HIP_DTYPE_I8 => DataType::Int8,
            _ => DataType::Float,
        }))).collect())
    }

    fn run(&self, ctx: &OpRunContext) -> Result<OutputList, OpError> {
        let _one_at_a_time = self.run_lock.lock().unwrap();
        // inputs -> device, once (host-synchronous copies: no ordering flag needed for the run)
        for (i, b) in self.inputs.iter().enumerate() {
            // one upload per input, in the element type the device graph reads (`require_as` fails with the reference's own cast error for another type)
            macro_rules! upload {
                ($t:ty) => {{
                    let x: TensorView<$t> = ctx.inputs().require_as(i)?;
                    if x.shape() != b.shape.as_slice() {
                        eprintln!("rten-hip: subgraph input \"{}\" was bound with another shape", b.name);
                        return Err(OpError::IncompatibleInputShapes("Input shape does not match the shape the subgraph was planned for"));
                    }
                    let host = x.to_contiguous_in(ctx.pool());
                    let data = host.data().ok_or(OpError::InvalidValue("input is not contiguous"))?;
                    self.hip.check(unsafe { sys::rten_hip_memcpy_h2d(self.hip.raw(), b.dev, data.as_ptr() as *const c_void, std::mem::size_of_val(data)) })?;
                }};
            }
            match b.dtype {
                sys::RTEN_HIP_DTYPE_I32 => upload!(i32),
                sys::RTEN_HIP_DTYPE_U8 => upload!(u8),
                sys::RTEN_HIP_DTYPE_I8 => upload!(i8),
                _ => upload!(f32),
            }
        }
        // every chain replays its hipGraph; the caller's stream is ordered behind them (flags 0), then waited for
        self.check(unsafe { sys::rten_hip_model_run(self.model, 0) })?;
        self.check(unsafe { sys::rten_hip_model_sync(self.model) })?; // also reports a sticky device fault (rten_hip.h)
        // outputs -> host, once
        let mut out = OutputList::new();
        for i in 0..self.n_outputs {
            let (mut dev, mut shape, mut ndim): (*const c_void, [i64; 8], i32) = (ptr::null(), [0; 8], 0);
            self.check(unsafe { sys::rten_hip_model_output(self.model, i as i32, &mut dev, shape.as_mut_ptr(), &mut ndim) })?;
            let dims: Vec<usize> = shape[..ndim as usize].iter().map(|&d| d as usize).collect();
            let len: usize = dims.iter().product();
            macro_rules! download {
                ($t:ty, $zero:expr) => {{
                    let mut data: Vec<$t> = ctx.pool().alloc(len);
                    data.resize(len, $zero);
                    self.hip.check(unsafe { sys::rten_hip_memcpy_d2h(self.hip.raw(), data.as_mut_ptr() as *mut c_void, dev, len * std::mem::size_of::<$t>()) })?;
                    out.push(Tensor::from_data(&dims, data).into());
                }};
            }
            match self.output_dtypes[i] {
                sys::RTEN_HIP_DTYPE_I32 => download!(i32, 0),
                sys::RTEN_HIP_DTYPE_U8 => download!(u8, 0),
                sys::RTEN_HIP_DTYPE_I8 => download!(i8, 0),
                _ => download!(f32, 0.0),
            }
        }
        Ok(out)
    }
}
