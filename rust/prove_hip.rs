//! prove_hip.rs — the Rust side of the boundary: `CSReferenceAssembly::prove_hip`, a sibling of `prove_cpu_basic`
//! (src/cs/implementations/prover.rs:153-168) that hands the whole proof to libboojum_hip.so and rebuilds the reference's
//! `Proof<F, H, EXT>` (src/cs/implementations/proof.rs:121-136) from the library's flat serialisation.
//!
//! Drop this file next to `prover.rs` (`mod prove_hip;` in src/cs/implementations/mod.rs) together with the generated
//! `boojum_hip_sys.rs` (tools/gen_rust_bindings.py).  Nothing here is specific to a circuit: the gate list comes from the
//! assembly's own evaluator table and, for evaluators without a hand-written kernel, from the op lists the reference's
//! `gpu_synthesizer` captures (`GatesSetForGPU`, src/gpu_synthesizer/mod.rs:354-506).
//!
//! There is no Rust toolchain in the image this repository is built in, so this file is shipped as source; its
//! marshalling is the 1:1 counterpart of `era_boojum_amd/binding.py::ProverSetup` and `era_boojum_amd/proof_format.py::parse`,
//! which the GPU tests exercise against the same C ABI.
#![allow(clippy::too_many_arguments)]

use std::ffi::CStr;
use std::os::raw::c_uint;

use crate::cs::implementations::polynomial_storage::SetupBaseStorage;
use crate::cs::implementations::proof::{OracleQuery, Proof, SingleRoundQueries};
use crate::cs::implementations::prover::ProofConfig;
use crate::cs::implementations::reference_cs::CSReferenceAssembly;
use crate::cs::implementations::transcript::Transcript;
use crate::cs::implementations::utils::make_non_residues;
use crate::cs::implementations::verifier::VerificationKey;
use crate::cs::implementations::witness::WitnessSet;
use crate::cs::oracle::TreeHasher;
use crate::cs::traits::evaluator::{GatePlacementType, GatePurpose};
use crate::cs::traits::GoodAllocator;
use crate::cs::{CSGeometry, LookupParameters};
use crate::config::CSConfig;
use crate::field::goldilocks::GoldilocksField;
use crate::field::{ExtensionField, FieldExtension, PrimeField, U64Representable};
use crate::gpu_synthesizer::{GPUDataCapture, GatesSetForGPU, Index, Relation};

#[path = "boojum_hip_sys.rs"]
pub mod sys;
use sys::*;

type F = GoldilocksField;

// ------------------------------------------------------------------------------------------------------------------
// context: one per GPU, one host thread (the `Worker` of the CPU path has no counterpart: the parallelism is on the device)
// ------------------------------------------------------------------------------------------------------------------
pub struct HipCtx {
    raw: *mut bj_ctx,
}
unsafe impl Send for HipCtx {}

impl HipCtx {
    pub fn new(device: i32) -> Result<Self, String> {
        // struct layouts (bj_gate_desc, bj_comm, ...) are those of the header these bindings were generated from
        let abi = unsafe { bj_abi_version() };
        if abi != BJ_ABI_VERSION as i32 {
            return Err(format!("libboojum_hip reports ABI version {abi}, boojum_hip_sys.rs was generated for {BJ_ABI_VERSION}"));
        }
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { bj_ctx_create(device, &mut raw) };
        if rc != BJ_OK {
            let msg = unsafe { CStr::from_ptr(bj_status_string(rc)) }.to_string_lossy().into_owned();
            return Err(format!("bj_ctx_create({device}): {msg} (libboojum_hip has no CPU fallback)"));
        }
        Ok(Self { raw })
    }
    /// Non-zero status -> panic with the library's message: the reference panics / asserts at the same places
    /// (e.g. "unsatisfied", prover.rs:1425-1438).
    fn check(&self, rc: i32) {
        if rc != BJ_OK {
            let msg = unsafe { CStr::from_ptr(bj_last_error(self.raw)) }.to_string_lossy().into_owned();
            panic!("libboojum_hip: {msg}");
        }
    }
}
impl Drop for HipCtx {
    fn drop(&mut self) {
        unsafe { bj_ctx_destroy(self.raw) }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// marshalling helpers
// ------------------------------------------------------------------------------------------------------------------
/// `[column][row]` u64, natural row order.  `GoldilocksField` is `repr(transparent)` over u64 and may be non-canonical in
/// memory (goldilocks/mod.rs:98-107); the library accepts any representative, so this is a plain copy.
fn flatten<'a>(cols: impl Iterator<Item = &'a [F]>, n: usize) -> Vec<u64> {
    let mut out = Vec::new();
    for c in cols {
        assert_eq!(c.len(), n);
        out.extend(c.iter().map(|el| el.as_u64()));
    }
    out
}

fn index_of(ix: &Index<F>, values: &mut Vec<u64>) -> bj_gate_index {
    match ix {
        Index::VariablePoly(i) => bj_gate_index { kind: BJ_IDX_VARIABLE_POLY as u32, index: *i as u32 },
        Index::WitnessPoly(i) => bj_gate_index { kind: BJ_IDX_WITNESS_POLY as u32, index: *i as u32 },
        Index::ConstantPoly(i) => bj_gate_index { kind: BJ_IDX_CONSTANT_POLY as u32, index: *i as u32 },
        Index::TemporaryValue(i) => bj_gate_index { kind: BJ_IDX_TEMPORARY as u32, index: *i as u32 },
        Index::ConstantValue(v) => {
            let pos = values.iter().position(|x| *x == v.as_u64_reduced()).unwrap_or_else(|| {
                values.push(v.as_u64_reduced());
                values.len() - 1
            });
            bj_gate_index { kind: BJ_IDX_CONSTANT_VALUE as u32, index: pos as u32 }
        }
    }
}

/// One `GPUDataCapture` (gpu_synthesizer/mod.rs:354-444) as the arrays of a `bj_gate_program`.  The capture numbers its
/// temporaries globally (one fresh number per operation from a counter per process); they are renumbered densely here only
/// to keep `num_temporaries` small.  Nothing else is needed on this side: the library brings every list into a canonical
/// form (csrc/gate_canon.h — slots by live range, common subexpressions merged, a structural fingerprint independent of
/// numbering and relation order), picks its build-time kernel for the evaluators of src/cs/gates/ by that fingerprint and
/// compiles one at `bj_setup_create` for any other capture.
pub struct OwnedProgram {
    relations: Vec<bj_gate_relation>,
    values: Vec<u64>,
    writes: Vec<bj_gate_index>,
    num_temporaries: u32,
    raw: bj_gate_program,
}
impl OwnedProgram {
    pub fn from_capture(c: &GPUDataCapture) -> Box<Self> {
        let mut values = Vec::new();
        let mut rename = std::collections::HashMap::new();
        let mut tmp = |ix: &Index<F>, values: &mut Vec<u64>, define: bool| -> bj_gate_index {
            if let Index::TemporaryValue(t) = ix {
                let next = rename.len();
                let slot = if define { *rename.entry(*t).or_insert(next) } else { *rename.get(t).expect("temporary read before it is written") };
                return bj_gate_index { kind: BJ_IDX_TEMPORARY as u32, index: slot as u32 };
            }
            index_of(ix, values)
        };
        let mut relations = Vec::with_capacity(c.relations.len());
        for (dst, rel) in c.relations.iter() {
            let (op, a, b) = match rel {
                Relation::Add(a, b) => (BJ_OP_ADD, a, Some(b)),
                Relation::Sub(a, b) => (BJ_OP_SUB, a, Some(b)),
                Relation::Mul(a, b) => (BJ_OP_MUL, a, Some(b)),
                Relation::Double(a) => (BJ_OP_DOUBLE, a, None),
                Relation::Negate(a) => (BJ_OP_NEGATE, a, None),
                Relation::Square(a) => (BJ_OP_SQUARE, a, None),
                Relation::Inverse(a) => (BJ_OP_INVERSE, a, None),
            };
            let a = tmp(a, &mut values, false);
            let b = b.map(|b| tmp(b, &mut values, false)).unwrap_or(bj_gate_index { kind: 0, index: 0 });
            let d = tmp(&dst.idx, &mut values, true);
            relations.push(bj_gate_relation { op: op as u32, dst: d.index, a, b });
        }
        let writes: Vec<bj_gate_index> = c.writes_per_repetition.iter().map(|w| tmp(w, &mut values, false)).collect();
        let num_temporaries = rename.len() as u32;
        let mut me = Box::new(Self { relations, values, writes, num_temporaries, raw: unsafe { std::mem::zeroed() } });
        me.raw = bj_gate_program {
            relations: me.relations.as_ptr(),
            num_relations: me.relations.len() as u32,
            values: me.values.as_ptr(),
            num_values: me.values.len() as u32,
            writes: me.writes.as_ptr(),
            num_writes: me.writes.len() as u32,
            num_temporaries: me.num_temporaries,
        };
        me
    }
}

/// Evaluators with a hand-written kernel, by the name `TypeErasedGateEvaluationFunction::debug_name` carries
/// (src/cs/traits/evaluator.rs:528-541).  Everything else goes through its op list.
fn builtin_kind(debug_name: &str) -> Option<i32> {
    if debug_name.contains("ConstantAllocatorConstraintEvaluator") {
        Some(BJ_GATE_CONSTANT_ALLOCATOR)
    } else if debug_name.contains("FmaGateInBaseWithoutConstantConstraintEvaluator") {
        Some(BJ_GATE_FMA_NO_CONSTANT)
    } else if debug_name.contains("ReductionGateConstraintEvaluator<4") {
        Some(BJ_GATE_REDUCTION4)
    } else if debug_name.contains("Poseidon2FlattenedGate") || debug_name.contains("Poseidon2RoundFunctionFlattenedEvaluator") {
        Some(BJ_GATE_POSEIDON2_FLATTENED)
    } else {
        None
    }
}

/// Device-resident setup: `SetupBaseStorage` + the `VerificationKeyCircuitGeometry` fields the prover reads, uploaded once
/// per circuit and reused for every proof (the prover-side half of `get_full_setup`, setup.rs:1273-1300).
pub struct HipSetup {
    raw: *mut bj_setup,
    num_vars: usize,
    has_lookup: bool,
    n: usize,
    // descriptors handed to bj_setup_create only borrow these for the duration of the call; kept for clarity of ownership
    _programs: Vec<Box<OwnedProgram>>,
}
impl Drop for HipSetup {
    fn drop(&mut self) {
        unsafe { bj_setup_destroy(self.raw) }
    }
}

impl<P: crate::field::traits::field_like::PrimeFieldLikeVectorized<Base = F>, CFG: CSConfig, A: GoodAllocator>
    CSReferenceAssembly<F, P, CFG, A>
{
    /// `bj_setup_create`: everything that does not depend on the witness.  `gates_for_gpu` holds one `GPUDataCapture` per
    /// evaluator that has no hand-written kernel (`GatesSetForGPU::add_gate::<G>(params)` for each such gate, in any
    /// order; matched by `evaluator_name`).  `comm` = `Some(bj_comm)` from `bj_comm_rccl_create` shards the proof by LDE
    /// cosets over the processes of the communicator (one GPU each).
    /// `POW` is the proof-of-work type parameter of `prove_cpu_basic` (prover.rs:153-168; `NoPow`, `Blake2s256` or `Keccak256`,
    /// pow.rs): it travels to the library as `bj_proof_config.pow_runner` and is read only when `proof_config.pow_bits != 0`.
    pub fn hip_setup<H: TreeHasher<F>, POW: crate::cs::implementations::pow::PoWRunner>(
        &self,
        ctx: &HipCtx,
        setup_base: &SetupBaseStorage<F, P>,
        vk: &VerificationKey<F, H>,
        proof_config: &ProofConfig,
        transcript_kind: u32,
        tree_hasher_kind: u32,
        gates_for_gpu: &GatesSetForGPU,
        comm: Option<&bj_comm>,
    ) -> HipSetup {
        let n = self.max_trace_len;
        let geometry: &CSGeometry = &vk.fixed_parameters.parameters;
        let fixed = &vk.fixed_parameters;
        // (width, repetitions, table id as a variable column?).  The two modes over general-purpose columns are left unimplemented by the
        // reference's own prover (prover.rs:412-437) and stay refused here.
        let (lookup_width, lookup_reps, table_id_as_variable) = match fixed.lookup_parameters {
            LookupParameters::NoLookup => (0u32, 0usize, false),
            LookupParameters::UseSpecializedColumnsWithTableIdAsConstant { width, num_repetitions, share_table_id } => {
                assert!(share_table_id, "a table id in constant columns must be shared (lookup_argument_in_ext.rs:371)");
                assert_eq!(fixed.table_ids_column_idxes.len(), 1);
                (width, num_repetitions, false)
            }
            // width + 1 variable columns per sub-argument, the last one the table id; no table-id constant (setup.rs:970-971)
            LookupParameters::UseSpecializedColumnsWithTableIdAsVariable { width, num_repetitions, .. } => {
                assert!(fixed.table_ids_column_idxes.is_empty());
                (width, num_repetitions, true)
            }
            other => panic!("lookup mode {other:?}: the reference's prover does not implement lookups over general-purpose columns either"),
        };
        // ---- evaluators over general purpose columns, in evaluator order, with their selector paths (prover.rs:995-1013) ----
        let mut programs: Vec<Box<OwnedProgram>> = Vec::new();
        let mut gates: Vec<bj_gate_desc> = Vec::new();
        let evaluators = &self.evaluation_data_over_general_purpose_columns.evaluators_over_general_purpose_columns;
        for (idx, ev) in evaluators.iter().enumerate() {
            let mut d: bj_gate_desc = unsafe { std::mem::zeroed() };
            let path = match ev.gate_purpose {
                GatePurpose::MarkerWithoutSelector => continue, // public input marker: no row of its own
                _ => fixed.selectors_placement.output_placement(idx).expect("selector must exist"),
            };
            assert!(path.len() <= 8);
            d.path_len = path.len() as c_uint;
            for (b, bit) in path.iter().enumerate() {
                d.path[b] = *bit as u8; // true = the constant column itself, false = 1 - it (prover.rs:2775-2916)
            }
            d.num_repetitions = ev.num_repetitions_on_row as c_uint;
            d.num_terms = ev.num_quotient_terms as c_uint;
            if let GatePlacementType::MultipleOnRow { per_chunk_offset } = ev.placement_type {
                d.var_stride = per_chunk_offset.variables_offset as c_uint;
                d.wit_stride = per_chunk_offset.witnesses_offset as c_uint; // non-copiable witness columns per repetition
                d.const_stride = per_chunk_offset.constants_offset as c_uint;
            }
            match ev.gate_purpose {
                GatePurpose::MarkerNeedsSelector => {
                    d.kind = BJ_GATE_NOP;
                    d.num_terms = 0;
                }
                _ => match builtin_kind(&ev.debug_name) {
                    Some(k) => d.kind = k,
                    None => {
                        let capture = gates_for_gpu
                            .descriptions
                            .iter()
                            .find(|c| c.evaluator_name == ev.unique_name || ev.debug_name.to_lowercase().contains(&c.evaluator_name.replace('_', "")))
                            .unwrap_or_else(|| panic!("no GPUDataCapture for evaluator {} — add it to the GatesSetForGPU", ev.debug_name));
                        assert_eq!(capture.num_quotient_terms, ev.num_quotient_terms);
                        let p = OwnedProgram::from_capture(capture);
                        d.kind = BJ_GATE_PROGRAM;
                        d.program = &p.raw;
                        programs.push(p);
                    }
                },
            }
            gates.push(d);
        }
        // ---- gates over specialized columns (evaluator.rs:190-236): op lists without a selector ----
        let mut spec: Vec<bj_gate_desc> = Vec::new();
        for (idx, ev) in self.evaluation_data_over_specialized_columns.evaluators_over_specialized_columns.iter().enumerate() {
            if !matches!(ev.gate_purpose, GatePurpose::Evaluatable { .. }) {
                continue; // the lookup marker
            }
            let capture = gates_for_gpu
                .descriptions
                .iter()
                .find(|c| c.evaluator_name == ev.unique_name)
                .unwrap_or_else(|| panic!("no GPUDataCapture for the specialized-columns evaluator {}", ev.debug_name));
            let p = OwnedProgram::from_capture(capture);
            let mut d: bj_gate_desc = unsafe { std::mem::zeroed() };
            d.kind = BJ_GATE_PROGRAM;
            d.num_repetitions = ev.num_repetitions_on_row as c_uint;
            d.num_terms = ev.num_quotient_terms as c_uint;
            d.var_stride = match ev.placement_type {
                GatePlacementType::MultipleOnRow { per_chunk_offset } => per_chunk_offset.variables_offset as c_uint,
                GatePlacementType::UniqueOnRow => capture_principal_width(capture) as c_uint,
            };
            // its own constant columns per repetition (share_constants = false): per_repetition_offset.constants_offset of
            // offsets_for_specialized_evaluators (evaluator_data.rs:196-238); the library finds the columns behind the table-id one
            let (_initial, per_repetition, _available) = self.evaluation_data_over_specialized_columns.offsets_for_specialized_evaluators[idx];
            d.const_stride = per_repetition.constants_offset as c_uint;
            d.program = &p.raw;
            programs.push(p);
            spec.push(d);
        }
        // ---- the circuit descriptor ----
        let num_vars = setup_base.copy_permutation_polys.len();
        let non_residues: Vec<u64> = make_non_residues::<F>(num_vars - 1, n).into_iter().map(|el| el.as_u64_reduced()).collect();
        let non_residues: Vec<u64> = std::iter::once(1u64).chain(non_residues).collect(); // k_0 = 1 (copy_permutation.rs:512-523)
        let pub_cols: Vec<c_uint> = fixed.public_inputs_locations.iter().map(|(c, _)| *c as c_uint).collect();
        let pub_rows: Vec<c_uint> = fixed.public_inputs_locations.iter().map(|(_, r)| *r as c_uint).collect();
        let circuit = bj_circuit {
            log_n: n.trailing_zeros(),
            num_vars: num_vars as c_uint,
            num_gp_vars: geometry.num_columns_under_copy_permutation as c_uint,
            num_witness_cols: geometry.num_witness_columns as c_uint,
            num_constant_cols: setup_base.constant_columns.len() as c_uint,
            lookup_width,
            lookup_reps: lookup_reps as c_uint,
            table_id_col: if table_id_as_variable { BJ_TABLE_ID_AS_VARIABLE } else { fixed.table_ids_column_idxes.first().copied().unwrap_or(0) as c_uint },
            quotient_degree: fixed.quotient_degree as c_uint,
            num_gates: gates.len() as c_uint,
            gates: gates.as_ptr(),
            non_residues: non_residues.as_ptr(),
            num_public_inputs: pub_cols.len() as c_uint,
            public_input_cols: pub_cols.as_ptr(),
            public_input_rows: pub_rows.as_ptr(),
            num_specialized_gates: spec.len() as c_uint,
            specialized_gates: if spec.is_empty() { std::ptr::null() } else { spec.as_ptr() },
        };
        let cfg = bj_proof_config {
            fri_lde_factor: proof_config.fri_lde_factor as c_uint,
            cap_size: proof_config.merkle_tree_cap_size as c_uint,
            security_level: proof_config.security_level as c_uint,
            pow_bits: proof_config.pow_bits,
            transcript: transcript_kind,
            tree_hasher: tree_hasher_kind,
            pow_runner: {
                use std::any::TypeId;
                let id = TypeId::of::<POW>();
                if id == TypeId::of::<sha3::Keccak256>() {
                    BJ_POW_KECCAK256
                } else if id == TypeId::of::<blake2::Blake2s256>() {
                    BJ_POW_BLAKE2S256
                } else {
                    assert_eq!(proof_config.pow_bits, 0, "NoPow (or an unknown PoWRunner) with pow_bits != 0");
                    0
                }
            },
        };
        assert!(proof_config.fri_folding_schedule.is_none(), "libboojum_hip computes the schedule (compute_fri_schedule, prover.rs:2281-2372)");
        let sigmas = flatten(setup_base.copy_permutation_polys.iter().map(|p| &p.storage[..]), n);
        let constants = flatten(setup_base.constant_columns.iter().map(|p| &p.storage[..]), n);
        let tables = flatten(setup_base.lookup_tables_columns.iter().map(|p| &p.storage[..]), n);
        let mut raw = std::ptr::null_mut();
        let rc = unsafe {
            bj_setup_create_sharded(
                ctx.raw,
                &circuit,
                sigmas.as_ptr(),
                constants.as_ptr(),
                if lookup_reps > 0 { tables.as_ptr() } else { std::ptr::null() },
                &cfg,
                comm.map(|c| c as *const bj_comm).unwrap_or(std::ptr::null()),
                &mut raw,
            )
        };
        ctx.check(rc);
        // the setup's own cap must be the verification key's (same tree hasher, same leaf layout sigma || constants || tables)
        let mut cap = vec![0u64; proof_config.merkle_tree_cap_size * 4];
        ctx.check(unsafe { bj_setup_cap(raw, cap.as_mut_ptr()) });
        debug_assert_eq!(caps_from_words::<H>(&cap), vk.setup_merkle_tree_cap, "setup cap differs from the verification key");
        HipSetup { raw, num_vars, has_lookup: lookup_reps > 0, n, _programs: programs }
    }

    /// Same inputs as `prove_cpu_basic` minus the CPU-side `SetupStorage` / setup tree (they live in `HipSetup`), same
    /// `Proof` out.  `TR` only fixes the proof's formal transcript / hasher types: the Fiat–Shamir transcript runs inside the
    /// library, in the order of prover.rs (cap of the VK, public inputs, witness cap, beta / gamma, ...).
    pub fn prove_hip<EXT, TR, H>(
        &self,
        ctx: &HipCtx,
        setup: &HipSetup,
        witness_set: WitnessSet<F>,
        proof_config: ProofConfig,
    ) -> Proof<F, H, EXT>
    where
        EXT: FieldExtension<2, BaseField = F>,
        TR: Transcript<F>,
        H: TreeHasher<F, Output = TR::CompatibleCap>,
    {
        let WitnessSet { public_inputs_values, public_inputs_with_locations, variables, witness, multiplicities } = witness_set;
        assert_eq!(public_inputs_values.len(), public_inputs_with_locations.len());
        assert_eq!(variables.len(), setup.num_vars);
        // the non-copiable witness columns travel right behind the variable columns (leaf = variables || witness || multiplicities)
        let vars = flatten(variables.iter().chain(witness.iter()).map(|p| &p.storage[..]), setup.n);
        let mult = flatten(multiplicities.iter().map(|p| &p.storage[..]), setup.n);
        let publics: Vec<u64> = public_inputs_values.iter().map(|el| el.as_u64_reduced()).collect();
        let mut proof = std::ptr::null_mut();
        ctx.check(unsafe {
            bj_prove(
                ctx.raw,
                setup.raw,
                vars.as_ptr(),
                if setup.has_lookup { mult.as_ptr() } else { std::ptr::null() },
                if publics.is_empty() { std::ptr::null() } else { publics.as_ptr() },
                &mut proof,
            )
        });
        let mut words = vec![0u64; unsafe { bj_proof_size_u64(proof) }];
        ctx.check(unsafe { bj_proof_serialize(proof, words.as_mut_ptr()) });
        unsafe { bj_proof_destroy(proof) };
        proof_from_bjpf::<H, EXT>(&words, proof_config)
    }

    /// The pipelined form of `prove_hip` for a host that loops over witnesses (the shape of convenience.rs:119-196 called in a
    /// loop): queues the proof on one of the context's two lanes and returns at once; `HipTicket::wait` hands the `Proof` over.
    ///     let mut prev = cs.prove_hip_async(&ctx, &setup, ws[0].clone());
    ///     for w in &ws[1..] { let cur = cs.prove_hip_async(&ctx, &setup, w.clone()); proofs.push(prev.wait(cfg)); prev = cur; }
    /// The next proof's PCIe transfer and first transforms run under the previous proof's latency-bound tail; every proof is the
    /// bytes `prove_hip` returns.  The ticket owns the flattened witness (the library reads it until `wait` returns).
    pub fn prove_hip_async<'a>(&self, ctx: &'a HipCtx, setup: &HipSetup, witness_set: WitnessSet<F>) -> HipTicket<'a> {
        let WitnessSet { public_inputs_values, public_inputs_with_locations, variables, witness, multiplicities } = witness_set;
        assert_eq!(public_inputs_values.len(), public_inputs_with_locations.len());
        assert_eq!(variables.len(), setup.num_vars);
        let vars = flatten(variables.iter().chain(witness.iter()).map(|p| &p.storage[..]), setup.n);
        let mult = flatten(multiplicities.iter().map(|p| &p.storage[..]), setup.n);
        let publics: Vec<u64> = public_inputs_values.iter().map(|el| el.as_u64_reduced()).collect();
        let mut raw = std::ptr::null_mut();
        ctx.check(unsafe {
            bj_prove_async(
                ctx.raw,
                setup.raw,
                vars.as_ptr(),
                if setup.has_lookup { mult.as_ptr() } else { std::ptr::null() },
                if publics.is_empty() { std::ptr::null() } else { publics.as_ptr() },
                &mut raw,
            )
        });
        HipTicket { ctx, raw, _vars: vars, _mult: mult }
    }
}

/// A proof in flight (`bj_ticket`).  Dropping it without `wait` waits for the proof and discards it.
pub struct HipTicket<'a> {
    ctx: &'a HipCtx,
    raw: *mut bj_ticket,
    _vars: Vec<u64>,
    _mult: Vec<u64>,
}

impl<'a> HipTicket<'a> {
    pub fn is_done(&self) -> bool {
        unsafe { bj_proof_poll(self.raw) == 1 }
    }
    pub fn wait<H: TreeHasher<F>, EXT: FieldExtension<2, BaseField = F>>(mut self, proof_config: ProofConfig) -> Proof<F, H, EXT> {
        let mut proof = std::ptr::null_mut();
        let raw = std::mem::replace(&mut self.raw, std::ptr::null_mut());
        self.ctx.check(unsafe { bj_proof_wait(raw, &mut proof) });
        let mut words = vec![0u64; unsafe { bj_proof_size_u64(proof) }];
        self.ctx.check(unsafe { bj_proof_serialize(proof, words.as_mut_ptr()) });
        unsafe { bj_proof_destroy(proof) };
        proof_from_bjpf::<H, EXT>(&words, proof_config)
    }
}

impl<'a> Drop for HipTicket<'a> {
    fn drop(&mut self) {
        if !self.raw.is_null() {
            unsafe { bj_proof_wait(self.raw, std::ptr::null_mut()) };
        }
    }
}

/// The same proof from the reference's own dumps — for a host that keeps the witness as `WitnessVec` + `DenseVariablesCopyHint`
/// (what `CSReferenceAssembly` has before `witness_set_from_witness_vec`, witness.rs:386-443) or hands work to another process:
/// `MemcopySerializable::write_into_buffer` bytes go straight into `bj_prove_from_dumps`, the cells are materialised on the GPU.
/// `bj_setup_create_from_dump` is the setup-side counterpart (`SetupBaseStorage::write_into_buffer`): with it the `bj_circuit` only
/// needs the geometry and the gate list, the library reads columns, selector tree, table-id column and quotient degree itself.
pub fn prove_hip_from_dumps<H: TreeHasher<F>, EXT: FieldExtension<2, BaseField = F>>(
    ctx: &HipCtx,
    setup: &HipSetup,
    witness_vec: &crate::cs::implementations::witness::WitnessVec<F>,
    variables_hint: &crate::cs::implementations::hints::DenseVariablesCopyHint,
    witness_hint: Option<&crate::cs::implementations::hints::DenseWitnessCopyHint>,     // circuits with non-copiable witness columns
    proof_config: ProofConfig,
) -> Proof<F, H, EXT> {
    use crate::cs::implementations::fast_serialization::MemcopySerializable;
    let (mut w, mut h, mut x) = (Vec::new(), Vec::new(), Vec::new());
    witness_vec.write_into_buffer(&mut w).expect("WitnessVec serialises");
    variables_hint.write_into_buffer(&mut h).expect("DenseVariablesCopyHint serialises");
    if let Some(wh) = witness_hint {
        wh.write_into_buffer(&mut x).expect("DenseWitnessCopyHint serialises");
    }
    let mut proof = std::ptr::null_mut();
    ctx.check(unsafe {
        bj_prove_from_dumps(
            ctx.raw,
            setup.raw,
            w.as_ptr() as *const _,
            w.len(),
            h.as_ptr() as *const _,
            h.len(),
            if witness_hint.is_some() { x.as_ptr() as *const _ } else { std::ptr::null() },
            x.len(),
            &mut proof,
        )
    });
    let mut words = vec![0u64; unsafe { bj_proof_size_u64(proof) }];
    ctx.check(unsafe { bj_proof_serialize(proof, words.as_mut_ptr()) });
    unsafe { bj_proof_destroy(proof) };
    proof_from_bjpf::<H, EXT>(&words, proof_config)
}

fn capture_principal_width(c: &GPUDataCapture) -> usize {
    let mut w = 0;
    let mut see = |ix: &Index<F>| {
        if let Index::VariablePoly(i) = ix {
            w = w.max(*i + 1)
        }
    };
    for (_, r) in c.relations.iter() {
        match r {
            Relation::Add(a, b) | Relation::Sub(a, b) | Relation::Mul(a, b) => {
                see(a);
                see(b)
            }
            Relation::Double(a) | Relation::Negate(a) | Relation::Square(a) | Relation::Inverse(a) => see(a),
        }
    }
    c.writes_per_repetition.iter().for_each(&mut see);
    w
}

// ------------------------------------------------------------------------------------------------------------------
// BJPF (bj_proof_serialize) -> Proof<F, H, EXT>.  Layout: era_boojum_amd/proof_format.py (all u64, little endian):
//   header[19] = 'BJPF', version 2, n_public, cap_size, n_values_at_z, n_values_at_z_omega, n_values_at_0, n_fri_oracles,
//                final_degree, n_queries, witness / stage-2 / quotient / setup leaf widths, base-oracle path depth, log_n,
//                fri_lde_factor, pow_bits, pow_challenge
//   schedule | public inputs | witness, stage-2, quotient caps | values at z, z*omega, 0 | FRI caps | final monomials c0, c1
//   per query: index | 4 x (leaf elements, path) | per FRI oracle (2 * 2^k leaf elements, path)
// ------------------------------------------------------------------------------------------------------------------
struct Reader<'a> {
    w: &'a [u64],
    pos: usize,
}
impl<'a> Reader<'a> {
    fn take(&mut self, k: usize) -> &'a [u64] {
        let s = &self.w[self.pos..self.pos + k];
        self.pos += k;
        s
    }
    fn one(&mut self) -> u64 {
        self.take(1)[0]
    }
}

/// Four canonical field elements per digest for the algebraic hashers (`H::Output = [F; 4]`), the 32 digest bytes packed
/// little-endian into four words for Blake2s / Keccak256 (`H::Output = [u8; 32]`): both are 32 plain bytes in memory.
fn caps_from_words<H: TreeHasher<F>>(words: &[u64]) -> Vec<H::Output> {
    assert_eq!(std::mem::size_of::<H::Output>(), 32);
    words
        .chunks_exact(4)
        .map(|d| {
            let mut out = std::mem::MaybeUninit::<H::Output>::uninit();
            unsafe {
                std::ptr::copy_nonoverlapping(d.as_ptr() as *const u8, out.as_mut_ptr() as *mut u8, 32);
                out.assume_init()
            }
        })
        .collect()
}
fn fields(words: &[u64]) -> Vec<F> {
    words.iter().map(|w| F::from_u64_unchecked(*w)).collect() // the library emits canonical residues
}
fn ext_values<EXT: FieldExtension<2, BaseField = F>>(words: &[u64]) -> Vec<ExtensionField<F, 2, EXT>> {
    words
        .chunks_exact(2)
        .map(|c| ExtensionField::<F, 2, EXT>::from_coeff_in_base([F::from_u64_unchecked(c[0]), F::from_u64_unchecked(c[1])]))
        .collect()
}

pub fn proof_from_bjpf<H: TreeHasher<F>, EXT: FieldExtension<2, BaseField = F>>(words: &[u64], proof_config: ProofConfig) -> Proof<F, H, EXT> {
    let mut r = Reader { w: words, pos: 0 };
    let h = r.take(19);
    assert!(h[0] == 0x424A_5046 && h[1] == 2, "not a BJPF v2 proof");
    let (n_pub, cap, nz, nzo, n0, n_fri, final_degree, n_queries) =
        (h[2] as usize, h[3] as usize, h[4] as usize, h[5] as usize, h[6] as usize, h[7] as usize, h[8] as usize, h[9] as usize);
    let widths = [h[10] as usize, h[11] as usize, h[12] as usize, h[13] as usize];
    let (depth, log_n, fri_lde, pow_challenge) = (h[14] as usize, h[15] as usize, h[16] as usize, h[18]);
    assert_eq!(cap, proof_config.merkle_tree_cap_size);
    assert_eq!(fri_lde, proof_config.fri_lde_factor);
    let schedule: Vec<usize> = r.take(n_fri).iter().map(|k| *k as usize).collect();
    let public_inputs = fields(r.take(n_pub));
    let witness_oracle_cap = caps_from_words::<H>(r.take(4 * cap));
    let stage_2_oracle_cap = caps_from_words::<H>(r.take(4 * cap));
    let quotient_oracle_cap = caps_from_words::<H>(r.take(4 * cap));
    let values_at_z = ext_values::<EXT>(r.take(2 * nz));
    let values_at_z_omega = ext_values::<EXT>(r.take(2 * nzo));
    let values_at_0 = ext_values::<EXT>(r.take(2 * n0));
    let mut fri_caps: Vec<Vec<H::Output>> = (0..n_fri).map(|_| caps_from_words::<H>(r.take(4 * cap))).collect();
    let fri_base_oracle_cap = fri_caps.remove(0);
    let final_fri_monomials = [fields(r.take(final_degree)), fields(r.take(final_degree))];
    let mut queries_per_fri_repetition = Vec::with_capacity(n_queries);
    for _ in 0..n_queries {
        let _index = r.one(); // the verifier re-derives it from the transcript (prover.rs:2161-2182)
        let mut base = widths.iter().map(|w| OracleQuery::<F, H> {
            leaf_elements: fields(r.take(*w)),
            proof: caps_from_words::<H>(r.take(4 * depth)),
        });
        let witness_query = base.next().unwrap();
        let stage_2_query = base.next().unwrap();
        let quotient_query = base.next().unwrap();
        let setup_query = base.next().unwrap();
        let mut fri_queries = Vec::with_capacity(n_fri);
        let mut leaves = (1usize << log_n) * fri_lde;
        for k in schedule.iter() {
            let d = ((leaves >> k) / cap).trailing_zeros() as usize;
            fri_queries.push(OracleQuery::<F, H> { leaf_elements: fields(r.take(2usize << k)), proof: caps_from_words::<H>(r.take(4 * d)) });
            leaves >>= k;
        }
        queries_per_fri_repetition.push(SingleRoundQueries { witness_query, stage_2_query, quotient_query, setup_query, fri_queries });
    }
    assert_eq!(r.pos, words.len(), "trailing data in the proof buffer");
    Proof {
        proof_config,
        public_inputs,
        witness_oracle_cap,
        stage_2_oracle_cap,
        quotient_oracle_cap,
        final_fri_monomials,
        values_at_z,
        values_at_z_omega,
        values_at_0,
        fri_base_oracle_cap,
        fri_intermediate_oracles_caps: fri_caps,
        queries_per_fri_repetition,
        pow_challenge,
        _marker: std::marker::PhantomData,
    }
}
