//! The installer: a loaded model whose WHOLE graph runs as one resident `HipSubgraph`.
//!
//! `rten_hip::register` (ops.rs) replaces operators one by one: correct, drop-in, and PCIe-bound -- every node uploads its inputs and downloads
//! its outputs (hard part 1).  `load_resident` is the other end of the trade: the model file's bytes go to the C++ plan executor
//! behind the C ABI (`rten_hip_model_load_ex`), which keeps every value in HBM, and the reference's `Graph` gets ONE operator node whose inputs are
//! the graph's inputs and whose outputs are the graph's outputs.  `Graph::add_op` makes a new operator the SOURCE of the value nodes it is given as
//! outputs ("enables replacing an operator while preserving metadata of the output value", src/graph.rs:405-452), so the planner
//! (`Graph::execution_plan`, src/graph.rs:880-1286), which walks from the requested outputs through `source_ids`, schedules that one node and
//! nothing else; names, shapes and dtypes of the model's inputs and outputs are the loader's own.  `Model::run`, `rten-cli`, `rten-examples`: unchanged.
//!
//! What the reference has to offer for this (INTEGRATION.md section 2.1 lists the edits):
//!   * `ModelOptions::set_graph_rewriter(Box<dyn Fn(&mut Graph) + Send + Sync>)` -- applied by `load_graph` to the top-level graph where the operator
//!     rewriter of `register` is applied: after `GraphOptimizer::optimize` (src/model/onnx_loader.rs:293-299, src/model/rten_loader.rs:185-190);
//!   * `Graph` added to the crate's re-exports (src/lib.rs:200: `pub use graph::{Dimension, NodeId, ...}`); the methods used here are already `pub`
//!     (`input_ids`, `output_ids`, `node_name`, `add_op`).
//! If the backend refuses the graph (an operator outside its registry, a shape-driven graph), nothing is replaced and the per-operator wrappers
//! are installed instead: a model always loads, the log says which form it got.
//!
//! NOT COMPILED in the build image -- see lib.rs.  tests/test_abi.py checks the `sys::` names and arities used here.
use std::path::Path;
use std::sync::Arc;

use rten::{Graph, LoadError, Model, ModelOptions, NodeId};

use crate::{accelerate, HipContext, HipSubgraph, HipSubgraphPool};

/// How the resident form was asked for: the launch-plan file of the model (`profiles/plans/*.json`), the number of sub-batch chains and the
/// shapes the static plan is built for (one per graph input, by name).
pub struct ResidentPlan<'a> {
    pub plan_json: Option<&'a str>,
    pub chains: i32,
    /// replicas of the subgraph behind the one operator (`HipSubgraphPool`): `Model::run` callers on different threads run side by side on the device.
    /// 1 = a single resident subgraph.  (bench.py's defaults: f32 ResNet-50 2 lanes of one chain, the dynamically quantized graph 4 lanes.)
    pub lanes: usize,
    pub input_shapes: Vec<(String, Vec<usize>)>,
}

/// Replace everything between `graph`'s inputs and outputs by one `HipSubgraph` built from the model's ONNX bytes.  `Err` leaves the graph as it was.
pub fn install_resident(graph: &mut Graph, hip: &Arc<HipContext>, onnx: &[u8], plan: &ResidentPlan) -> Result<NodeId, rten::ops::OpError> {
    let shapes: Vec<(&str, Vec<usize>)> = plan.input_shapes.iter().map(|(n, s)| (n.as_str(), s.clone())).collect();
    let sub = HipSubgraph::load(hip.clone(), onnx, plan.plan_json, plan.chains, &shapes)?;
    // the operator takes its inputs in the SUBGRAPH's declaration order, which is the model file's: match the reference graph's input nodes by name
    let mut inputs: Vec<Option<NodeId>> = Vec::new();
    for name in sub.input_names() {
        let id = graph.input_ids().iter().copied().find(|id| graph.node_name(*id) == name).ok_or(rten::ops::OpError::MissingInputs)?;
        inputs.push(Some(id));
    }
    let outputs: Vec<Option<NodeId>> = graph.output_ids().iter().copied().map(Some).collect();
    if outputs.len() != sub.num_outputs() {
        return Err(rten::ops::OpError::InvalidValue("the resident subgraph and the loaded graph disagree on the number of outputs"));
    }
    // the new node becomes the source of every graph output: the nodes that produced them before are no longer reachable from the outputs
    if plan.lanes > 1 {
        // one context (stream) per further lane, on the device of `hip`; the replicas share the first subgraph's weights (rten_hip_model_clone)
        let device = unsafe { rten_hip_sys::rten_hip_device_id(hip.raw()) };
        let mut contexts = Vec::new();
        for _ in 1..plan.lanes {
            contexts.push(HipContext::new(device).map_err(|_| rten::ops::OpError::InvalidValue("could not create a context for a further lane"))?);
        }
        let pool = HipSubgraphPool::new(sub, contexts)?;
        return Ok(graph.add_op(Some("hip_resident_subgraph"), Arc::new(pool), &inputs, &outputs));
    }
    Ok(graph.add_op(Some("hip_resident_subgraph"), Arc::new(sub), &inputs, &outputs))
}

/// `ModelOptions::load_file` with the whole graph resident on the device when the backend covers it, and the per-operator wrappers otherwise.
///
/// ```ignore
/// let hip = HipContext::new(0)?;
/// let plan = ResidentPlan { plan_json: Some(include_str!("f32_lanes.json")), chains: 1, lanes: 2, input_shapes: vec![("x".into(), vec![32, 3, 224, 224])] };
/// let model = rten_hip::load_resident(ModelOptions::with_all_ops(), hip, "resnet50.onnx", plan)?;
/// let logits = model.run_one(batch.view().into(), None)?;      // the reference's call, one H2D + one D2H per run
/// ```
pub fn load_resident<P: AsRef<Path>>(mut opts: ModelOptions, hip: Arc<HipContext>, path: P, plan: ResidentPlan<'static>) -> Result<Model, LoadError> {
    let bytes = Arc::new(std::fs::read(path.as_ref()).map_err(LoadError::from)?);
    let (onnx, hip_for_graph, hip_for_ops) = (bytes.clone(), hip.clone(), hip);
    opts.set_graph_rewriter(Box::new(move |graph: &mut Graph| {
        match install_resident(graph, &hip_for_graph, &onnx, &plan) {
            Ok(_) => eprintln!("rten-hip: the whole graph runs as one resident subgraph"),
            Err(e) => eprintln!("rten-hip: resident form refused ({e:?}); operators are accelerated one by one"),
        }
    }));
    // nodes the resident operator did not replace (all of them, if it was refused) still get their device wrappers
    opts.set_operator_rewriter(Box::new(move |op| accelerate(op, &hip_for_ops)));
    opts.load((*bytes).clone())
}
