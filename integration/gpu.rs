//! zkir-runtime/src/gpu.rs — `VM::new(program, inputs, config).run()` on an MI355X through the C ABI of include/zkir_amd.h.
//!
//! NOT COMPILED IN THIS REPOSITORY: the build image has no Rust toolchain (no rustc / cargo; the workspace's crates.io dependencies cannot be fetched).
//! It is the module a maintainer of seceq/zkir adds (`#[cfg(feature = "gpu")] pub mod gpu;` in zkir-runtime/src/lib.rs, build.rs next to it); every call
//! below is exercised from C++ by tests/cpp/reference_tests.cpp and from Python by zkir_amd/runtime.py.  Reference seam: zkir-runtime/src/vm.rs:138 (VM::new),
//! :208 (VM::run), :54-103 (ExecutionResult); error kinds zkir-runtime/src/error.rs:7-37.
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

use zkir_spec::Program;

use crate::error::RuntimeError;
use crate::vm::{HaltReason, VMConfig};

#[repr(C)]
pub struct ZkirVmConfig {
    pub max_cycles: u64,
    pub trace: u8,
    pub enable_range_checking: u8,
    pub enable_execution_trace: u8,
    pub enable_deferred_model: u8,
}

/// Device pointers (HBM), struct of arrays, 372 B per row (include/zkir_amd.h: zkir_trace_columns).
#[repr(C)]
pub struct ZkirTraceColumns {
    pub cycle: *mut u64,
    pub pc: *mut u64,
    pub instruction: *mut u32,
    pub registers: *mut u64,
    pub bound_bits: *mut u32,
    pub bound_tag: *mut u8,
    pub bound_payload: *mut u64,
    pub reg_state: *mut u8,
    pub reg_stride: u64,
}

#[repr(C)]
pub struct ZkirPublicInputs {
    pub n_real: u64,
    pub entry_point: u64,
    pub deferred: u32,
    pub fri_params: u32,   // num_queries | pow_bits << 16, 0 = the defaults (zkir_public_inputs_set_params)
    pub program_digest: [u32; 4],
    pub io_digest: [u32; 4],
    pub program_blob: *const u8, // borrowed: must outlive zkir_prove
    pub program_blob_len: u64,
    // mode 2 (deferred == 2: + the I/O argument): the tapes and the halt reason in the clear (borrowed; zkir_public_inputs_of fills them)
    pub inputs: *const u64,
    pub n_inputs: u64,
    pub outputs: *const u64,
    pub n_outputs: u64,
    pub halt_kind: u32,
    pub reserved2: u32,
    pub halt_code: u64,
    pub writes_before: u64,
    pub reads_before: u64,
    // mode 3 (deferred == 3: + the memory argument): the run's memory witness (borrowed; zkir_public_inputs_set_memory points them at a ZkirMemcheckWitness)
    pub mem_old: *const u64,
    pub mem_told: *const u32,
    pub cell_addr: *const u64,
    pub cell_bytes: *const u64,
    pub cell_time: *const u32,
    pub n_cells: u64,
    // mode 4 (round 6): the hash calls of the run as the proof's hash section — borrowed from a zkir_memcheck_witness made with zkir_memcheck_witness_of_mode(.., 4, ..)
    pub hash_section: *const u32,
    pub hash_section_words: u64,
}
pub enum ZkirMemcheckWitness {}

pub enum ZkirResult {}
pub enum ZkirDeltaLog {}
pub enum ZkirStarkCtx {}

extern "C" {
    fn zkir_exec(blob: *const u8, len: usize, inputs: *const u64, n_inputs: usize, cfg: *const ZkirVmConfig, out: *mut *mut ZkirResult) -> c_int;
    fn zkir_result_free(r: *mut ZkirResult);
    fn zkir_result_delta_log(r: *const ZkirResult) -> *const ZkirDeltaLog;
    fn zkir_result_trace(r: *const ZkirResult) -> *const ZkirTraceColumns;
    fn zkir_result_copy_column(r: *const ZkirResult, field: c_int, reg: c_int, dst: *mut c_void) -> c_int;
    fn zkir_delta_log_cycles(l: *const ZkirDeltaLog) -> u64;
    fn zkir_delta_log_halt_kind(l: *const ZkirDeltaLog) -> c_int;
    fn zkir_delta_log_halt_code(l: *const ZkirDeltaLog) -> u64;
    fn zkir_delta_log_n_outputs(l: *const ZkirDeltaLog) -> usize;
    fn zkir_delta_log_outputs(l: *const ZkirDeltaLog) -> *const u64;
    fn zkir_last_error() -> *const c_char;
    // proving (self-defined stages: the reference has none — DESIGN.md §8)
    fn zkir_stark_ctx_create(log_n: u32, log_blowup: u32, out: *mut *mut ZkirStarkCtx) -> c_int;
    fn zkir_stark_ctx_free(ctx: *mut ZkirStarkCtx);
    fn zkir_padded_log_n(n_real: u64) -> u32;
    fn zkir_public_inputs_of(log: *const ZkirDeltaLog, blob: *const u8, len: usize, inputs: *const u64, n_inputs: usize, deferred: u32, out: *mut ZkirPublicInputs) -> c_int;
    fn zkir_memcheck_witness_of(log: *const ZkirDeltaLog, blob: *const u8, len: usize, out: *mut *mut ZkirMemcheckWitness) -> c_int;   // mode 3: the host's sequential memory replay
    fn zkir_memcheck_witness_free(w: *mut ZkirMemcheckWitness);
    fn zkir_public_inputs_set_memory(public: *mut ZkirPublicInputs, w: *const ZkirMemcheckWitness);                                      // sets deferred = 3
    fn zkir_prove(ctx: *const ZkirStarkCtx, trace: *const ZkirTraceColumns, public: *const ZkirPublicInputs, proof: *mut *mut u32, words: *mut u64, stage_ms: *mut f32,
                  stream: *mut c_void) -> c_int;
    fn zkir_verify(proof: *const u32, words: u64, expect: *const ZkirPublicInputs) -> c_int;
    fn zkir_proof_free(proof: *mut u32);
    // prove(result, params) -> proof bytes (the handle of zkir_exec keeps the program and the input tape)
    fn zkir_prove_result(result: *const ZkirResult, params: *const ZkirProverParams, proof: *mut *mut u8, len: *mut usize) -> c_int;
    fn zkir_proof_bytes_free(proof: *mut u8);
    fn zkir_abi_version() -> u32;
}

/// zkir_prover_params: the proof's mode (0 default VM mode, 1 deferred model, 2 + the I/O argument, 3 + the memory argument) and its FRI parameters
/// (0 = the defaults: 50 queries, 12 grinding bits; accepted 50..128 / 12..24).  They are header words of the proof: a verifier's `expect` must name the same.
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct ZkirProverParams { pub mode: u32, pub num_queries: u32, pub pow_bits: u32 }
pub const ZKIR_AMD_ABI_VERSION: u32 = 6;

/// include/zkir_amd.h return codes <-> RuntimeError (error.rs:7-37); the message text is the reference's own.
fn map_error(code: c_int) -> RuntimeError {
    let msg = unsafe { CStr::from_ptr(zkir_last_error()) }.to_string_lossy().into_owned();
    match code {
        // 1 MisalignedAccess / 2 InvalidMemoryAccess / 3 DivisionByZero / 4 InvalidSyscall carry their fields in the message; a maintainer who wants the
        // structured variants parses them back or extends the ABI with zkir_last_error_fields() — the kinds are distinct codes already
        5 | 6 | 7 | 8 => RuntimeError::Other(msg),
        _ => RuntimeError::Other(format!("zkir_amd error {code}: {msg}")),
    }
}

/// `ExecutionResult` with the execution trace left in HBM (vm.rs:54-103).
pub struct GpuExecutionResult {
    handle: *mut ZkirResult,
    blob: Vec<u8>,
    inputs: Vec<u64>,
    deferred: bool,
}

impl GpuExecutionResult {
    fn log(&self) -> *const ZkirDeltaLog { unsafe { zkir_result_delta_log(self.handle) } }
    pub fn cycles(&self) -> u64 { unsafe { zkir_delta_log_cycles(self.log()) } }
    pub fn outputs(&self) -> Vec<u64> {
        unsafe { std::slice::from_raw_parts(zkir_delta_log_outputs(self.log()), zkir_delta_log_n_outputs(self.log())) }.to_vec()
    }
    pub fn halt_reason(&self) -> HaltReason {
        match unsafe { zkir_delta_log_halt_kind(self.log()) } {
            0 => HaltReason::Ebreak,
            1 => HaltReason::Exit(unsafe { zkir_delta_log_halt_code(self.log()) }),
            _ => HaltReason::CycleLimit,
        }
    }
    /// Device-resident columns for a prover that links HIP itself.
    pub fn trace_columns(&self) -> &ZkirTraceColumns { unsafe { &*zkir_result_trace(self.handle) } }
    /// One column copied to the host (field ids: include/zkir_amd.h ZKIR_FIELD_*): rebuilding `Vec<TraceRow>` for tests.
    pub fn copy_column_u64(&self, field: c_int, reg: c_int) -> Result<Vec<u64>, RuntimeError> {
        let mut v = vec![0u64; self.cycles() as usize];
        let rc = unsafe { zkir_result_copy_column(self.handle, field, reg, v.as_mut_ptr() as *mut c_void) };
        if rc != 0 { Err(map_error(rc)) } else { Ok(v) }
    }
    /// `zkir_runtime::prove(&result, &params)` as SURVEY 8(b) writes it: the proof bytes of the whole run behind this handle.
    pub fn prove_with(&self, params: &ZkirProverParams) -> Result<Vec<u8>, RuntimeError> {
        assert_eq!(unsafe { zkir_abi_version() }, ZKIR_AMD_ABI_VERSION, "libzkir_amd.so was built from another revision of include/zkir_amd.h");
        let (mut proof, mut len) = (std::ptr::null_mut::<u8>(), 0usize);
        let rc = unsafe { zkir_prove_result(self.handle, params, &mut proof, &mut len) };
        if rc != 0 { return Err(map_error(rc)); }
        let out = unsafe { std::slice::from_raw_parts(proof, len) }.to_vec();
        unsafe { zkir_proof_bytes_free(proof) };
        Ok(out)
    }
    /// `prove()` of north_star: the proof words (u32 little-endian, format zkir_proof_version()) and the public inputs it is bound to.
    /// Mode 0 (default VM mode) or 1 (the deferred model), as the run was configured.
    pub fn prove(&self) -> Result<(Vec<u32>, ZkirPublicInputs), RuntimeError> { self.prove_mode(self.deferred as u32) }
    /// mode 2 = the default VM mode + the I/O argument (the proof says what was read and written), 3 = mode 2 + the memory argument (loads / stores constrained,
    /// memory consistent; whole runs only; refused for runs that execute a hash syscall or touch an address of 2^40 or more)
    pub fn prove_mode(&self, mode: u32) -> Result<(Vec<u32>, ZkirPublicInputs), RuntimeError> {
        let mut public = std::mem::MaybeUninit::<ZkirPublicInputs>::uninit();
        let rc = unsafe { zkir_public_inputs_of(self.log(), self.blob.as_ptr(), self.blob.len(), self.inputs.as_ptr(), self.inputs.len(), mode, public.as_mut_ptr()) };
        if rc != 0 { return Err(map_error(rc)); }
        let mut public = unsafe { public.assume_init() };
        let mut witness: *mut ZkirMemcheckWitness = std::ptr::null_mut();
        if mode == 3 {
            let rc = unsafe { zkir_memcheck_witness_of(self.log(), self.blob.as_ptr(), self.blob.len(), &mut witness) };
            if rc != 0 { return Err(map_error(rc)); }
            unsafe { zkir_public_inputs_set_memory(&mut public, witness) };
        }
        struct FreeWitness(*mut ZkirMemcheckWitness);
        impl Drop for FreeWitness { fn drop(&mut self) { if !self.0.is_null() { unsafe { zkir_memcheck_witness_free(self.0) } } } }
        let _free = FreeWitness(witness);                     // the borrowed mode-3 pointers are only read by zkir_prove below
        let mut ctx: *mut ZkirStarkCtx = std::ptr::null_mut();
        let rc = unsafe { zkir_stark_ctx_create(zkir_padded_log_n(public.n_real), 1, &mut ctx) };
        if rc != 0 { return Err(map_error(rc)); }
        let (mut proof, mut words) = (std::ptr::null_mut::<u32>(), 0u64);
        let rc = unsafe { zkir_prove(ctx, zkir_result_trace(self.handle), &public, &mut proof, &mut words, std::ptr::null_mut(), std::ptr::null_mut()) };
        unsafe { zkir_stark_ctx_free(ctx) };
        if rc != 0 { return Err(map_error(rc)); }
        let out = unsafe { std::slice::from_raw_parts(proof, words as usize) }.to_vec();
        unsafe { zkir_proof_free(proof) };
        // the returned struct is what a verifier's `expect` needs (row count, mode, entry point, digests); the prover-side borrowed pointers end with this call
        public.mem_old = std::ptr::null(); public.mem_told = std::ptr::null(); public.cell_addr = std::ptr::null(); public.cell_bytes = std::ptr::null();
        public.cell_time = std::ptr::null(); public.n_cells = 0; public.hash_section = std::ptr::null(); public.hash_section_words = 0;
        Ok((out, public))
    }
}

impl Drop for GpuExecutionResult {
    fn drop(&mut self) { unsafe { zkir_result_free(self.handle) } }
}

/// Host-only check of a proof against the public inputs the caller expects (0 = accepted).
pub fn verify(proof: &[u32], expect: &ZkirPublicInputs) -> bool { unsafe { zkir_verify(proof.as_ptr(), proof.len() as u64, expect) == 0 } }

/// Drop-in for `VM::new(program, inputs, config).run()` (vm.rs:138, :208).  A debug-format program is reported as an error (code 7), not a panic (vm.rs:141-147).
pub fn run_on_gpu(program: &Program, inputs: &[u64], config: &VMConfig) -> Result<GpuExecutionResult, RuntimeError> {
    let blob = program.to_bytes(); // zkir-spec/src/program.rs:300-315
    let cfg = ZkirVmConfig {
        max_cycles: config.max_cycles,
        trace: 0,
        enable_range_checking: config.enable_range_checking as u8,
        enable_execution_trace: config.enable_execution_trace as u8,
        enable_deferred_model: config.enable_deferred_model as u8,
    };
    let mut res: *mut ZkirResult = std::ptr::null_mut();
    let rc = unsafe { zkir_exec(blob.as_ptr(), blob.len(), inputs.as_ptr(), inputs.len(), &cfg, &mut res) };
    if rc != 0 { return Err(map_error(rc)); }
    Ok(GpuExecutionResult { handle: res, blob, inputs: inputs.to_vec(), deferred: config.enable_deferred_model })
}
