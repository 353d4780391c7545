/* zkir_amd.h — C ABI of the MI355X-native ZKIR v3.4 execution-trace path.
 *
 * The reference (seceq/zkir) has NO FFI/plugin boundary: its only seam is the Rust API
 *     VM::new(program, inputs, config) -> VM          zkir-runtime/src/vm.rs:138
 *     VM::run(self) -> Result<ExecutionResult>        zkir-runtime/src/vm.rs:208
 *     ExecutionResult::get_memory_trace()             zkir-runtime/src/vm.rs:85-94
 * so this header defines the boundary a thin Rust shim would bind to replace that
 * path.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Two layers:
 *   1. zkir_exec / zkir_result_*  — drop-in for VM::new + VM::run: host interpreter -> delta log ->
 *      HIP kernels that materialise the wide SoA trace in HBM.
 *   2. zkir_interpret / zkir_*_launch — the same stages individually (host delta log; device kernels
 *      on caller-owned device buffers and a caller-chosen HIP stream), used by bench.py and by
 *      multi-GPU row sharding.
 *
 * Error codes mirror RuntimeError (zkir-runtime/src/error.rs:7-37) + program validation
 * (zkir-spec/src/program.rs:147-167).  The message of the last failure on the calling thread is
 * returned by zkir_last_error().
 */
#ifndef ZKIR_AMD_H
#define ZKIR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes --------------------------------------------------------------------------- */
enum {
  ZKIR_OK = 0,
  ZKIR_ERR_MISALIGNED = 1,      /* RuntimeError::MisalignedAccess      error.rs:15 */
  ZKIR_ERR_INVALID_MEMORY = 2,  /* RuntimeError::InvalidMemoryAccess   error.rs:18 */
  ZKIR_ERR_DIV_ZERO = 3,        /* RuntimeError::DivisionByZero        error.rs:21 */
  ZKIR_ERR_INVALID_SYSCALL = 4, /* RuntimeError::InvalidSyscall        error.rs:24 */
  ZKIR_ERR_DECODE = 5,          /* RuntimeError::Other("Decode error") vm.rs:376   */
  ZKIR_ERR_OTHER = 6,           /* RuntimeError::Other                 error.rs:36 */
  ZKIR_ERR_BAD_PROGRAM = 7,     /* ZkIrError header/size validation program.rs:147-167,318-346; debug-format
                                   programs (vm.rs:141-147 panics) are reported here instead of aborting */
  ZKIR_ERR_DEVICE = 8,          /* HIP failure / no device: the product path never falls back to the CPU */
  ZKIR_ERR_ARGUMENT = 9
};

/* HaltReason, zkir-runtime/src/state.rs:8-15 */
enum { ZKIR_HALT_EBREAK = 0, ZKIR_HALT_EXIT = 1, ZKIR_HALT_CYCLE_LIMIT = 2 };

/* BoundSource tag, zkir-spec/src/bound.rs:82-93 (declaration order); payload meaning per tag */
enum {
  ZKIR_BOUND_PROGRAM_WIDTH = 0, /* payload 0 */
  ZKIR_BOUND_TYPE_WIDTH = 1,    /* payload = bits */
  ZKIR_BOUND_CRYPTO_OUTPUT = 2, /* payload = CryptoType: 0 Sha256, 1 Keccak256, 2 Poseidon2, 3 Blake3 (bound.rs:10-19) */
  ZKIR_BOUND_COMPUTED = 3,      /* payload 0 */
  ZKIR_BOUND_CONSTANT = 4       /* payload = the constant */
};

/* VMConfig, zkir-runtime/src/vm.rs:15-50 (defaults: 1_000_000, false, false, false, false) */
typedef struct zkir_vm_config {
  uint64_t max_cycles;
  uint8_t trace;                  /* per-cycle eprintln; ignored */
  uint8_t enable_range_checking;
  uint8_t enable_execution_trace;
  uint8_t enable_deferred_model;
} zkir_vm_config;

/* ---- delta log (host memory): what the sequential interpreter hands to the GPU ---------------- */

/* One register write: the full (value, bound, storage-state) triple of register `reg`, visible in the
 * pre-state of every row >= vis (TraceRow holds the state BEFORE its instruction, vm.rs:245-253).
 * Events are ordered by vis; there is at most one event per (reg, vis).  The first 16 events are the
 * initial snapshot (event r describes register r, vis = 0), so every register always has a writer. */
typedef struct zkir_reg_event {
  uint64_t value;     /* VMState.regs[reg] raw u64 (un-masked, state.rs:87-91) */
  uint64_t payload;   /* BoundSource payload */
  uint32_t max_bits;  /* ValueBound.max_bits */
  uint32_t vis;       /* shard-relative row index from which the write is visible (= cycle + 1) */
  uint8_t reg;        /* 0..15 */
  uint8_t state;      /* RegisterState: 0 Normalized, 1 Accumulated (trace.rs:11-17) */
  uint8_t tag;        /* ZKIR_BOUND_* */
  uint8_t pad[5];
} zkir_reg_event;     /* 32 bytes */

/* One architectural data-memory access (MemoryOp, trace.rs:149-167); bound is always
 * TypeWidth(8*width) (memory.rs:245) and is re-derived on the device. */
typedef struct zkir_mem_event {
  uint64_t address;
  uint64_t value;
  uint32_t row;       /* = timestamp = cycle */
  uint8_t is_write;
  uint8_t width;      /* 1, 2, 4, 8 */
  uint16_t pad;
} zkir_mem_event;     /* 24 bytes */

/* A deferred range check flushed at a checkpoint (range_check.rs:59-66, 140-168) */
typedef struct zkir_rc_event {
  uint64_t value;     /* Value40::to_u64() */
  uint64_t pc;
} zkir_rc_event;      /* 16 bytes */

/* An observation-point normalization (normalization_witness.rs:19-43), before expansion */
typedef struct zkir_norm_event {
  uint64_t cycle;
  uint64_t pc;
  uint64_t raw_value; /* register value before normalization */
  uint8_t reg;
  uint8_t state;      /* storage state before normalization (selects 20- or 30-bit unpacking, state.rs:202-220) */
  uint8_t opcode;     /* triggering opcode byte */
  uint8_t pad[5];
} zkir_norm_event;    /* 32 bytes */

/* A single-block SHA-256 syscall (input_len < 56): the padded 64-byte message block, big-endian words
 * as parse_message_block produces them (crypto.rs:127-139), and the row it belongs to. */
typedef struct zkir_sha_block {
  uint32_t message_block[16];
  uint64_t timestamp;
} zkir_sha_block;     /* 72 bytes */

typedef struct zkir_delta_log zkir_delta_log;   /* opaque, host memory */

/* Run the program on the host interpreter (bit-exact to VM::run, vm.rs:208-358) and record the delta
 * log.  tile_rows (power of two, 256..2048; 0 = default: 256 when max_cycles <= 2^21, else 512) fixes
 * the granularity of the tile index (one K1 workgroup per tile).
 * Returns ZKIR_OK or an error code (then *out is NULL).  No device WORK is queued; when the process has a device, the log's large
 * arrays (pc, instruction words, register events: >= 8 MiB) live in PINNED host blocks recycled by a process-wide pool, so that
 * zkir_host_to_device / hipMemcpyAsync out of them is plain DMA at link rate (ZKIR_PIN_LOG=0: pageable memory).  A caller that
 * queues asynchronous copies out of a log must let them complete before zkir_delta_log_free hands the blocks back to the pool. */
int zkir_interpret(const uint8_t* program_blob, size_t blob_len, const uint64_t* inputs, size_t n_inputs,
                   const zkir_vm_config* cfg, uint32_t tile_rows, zkir_delta_log** out);
void zkir_delta_log_free(zkir_delta_log* log);

/* Trace WINDOW: the delta log of rows [row_begin, min(row_end, rows of the run)) only (cycle_base = row_begin).  VM::run is a
 * sequential chain (vm.rs:208-358: cycle c + 1 needs the registers, memory and pc of cycle c), so the state at row_begin exists only
 * after rows [0, row_begin) have been executed: this call executes them UNTRACED (no log stores; measured 1.5-2x the traced rate),
 * takes the register snapshot from the live machine, records the window and stops.  Multi-GPU: rank g of G calls it
 * with its own row range on its own host core — every GPU gets its shard after g * n * t_untraced + n * t_traced, sooner than a
 * single traced interpreter reaches those rows, with no inter-process transport.  cfg->enable_execution_trace must be set.
 * zkir_delta_log_cycles / halt / outputs describe the machine where the interpretation stopped; zkir_delta_log_window_open() = 1 if
 * that was the window's end rather than a halt (the run continues past row_end). */
int zkir_interpret_window(const uint8_t* program_blob, size_t blob_len, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config* cfg,
                          uint32_t tile_rows, uint64_t row_begin, uint64_t row_end, zkir_delta_log** out);
int zkir_delta_log_window_open(const zkir_delta_log*);

/* Row sharding (multi-GPU): a self-contained delta log for rows [row_begin, row_end) of `log` (ABSOLUTE row numbers: a source that
 * is itself a window or a shard holds rows [cycle_base, cycle_base + n_rows))
 * (any row range; a row_begin that is not a multiple of tile_rows — segment proofs overlap by one row — gets the shard its own
 * tiling, rebuilt from the events).  Its first 16 events are the register snapshot at row_begin, event
 * `vis`, mem-event rows and the tile index are rebased to the shard, and zkir_delta_log_cycle_base()
 * returns row_begin.  Side logs (range checks, normalizations, SHA blocks) are cut by the same cycle range.
 * Outputs / halt reason / cycles stay those of the whole run. */
int zkir_delta_log_shard(const zkir_delta_log* log, uint64_t row_begin, uint64_t row_end, zkir_delta_log** out);
uint64_t zkir_delta_log_cycle_base(const zkir_delta_log*);

uint64_t zkir_delta_log_cycles(const zkir_delta_log*);        /* ExecutionResult.cycles */
int zkir_delta_log_halt_kind(const zkir_delta_log*);          /* ZKIR_HALT_* */
uint64_t zkir_delta_log_halt_code(const zkir_delta_log*);     /* Exit(code) */
size_t zkir_delta_log_n_outputs(const zkir_delta_log*);
const uint64_t* zkir_delta_log_outputs(const zkir_delta_log*);
uint64_t zkir_delta_log_n_rows(const zkir_delta_log*);        /* 0 unless enable_execution_trace */
uint32_t zkir_delta_log_tile_rows(const zkir_delta_log*);
const uint64_t* zkir_delta_log_pc(const zkir_delta_log*);     /* [n_rows] TraceRow.pc          */
const uint32_t* zkir_delta_log_inst(const zkir_delta_log*);   /* [n_rows] TraceRow.instruction */
size_t zkir_delta_log_n_reg_events(const zkir_delta_log*);    /* includes the 16 snapshot events */
const zkir_reg_event* zkir_delta_log_reg_events(const zkir_delta_log*);
size_t zkir_delta_log_n_tiles(const zkir_delta_log*);
const uint32_t* zkir_delta_log_tile_ev_off(const zkir_delta_log*); /* [n_tiles+1] first event with vis > tile start */
const uint32_t* zkir_delta_log_tile_snap(const zkir_delta_log*);   /* [n_tiles][16] last event per register with vis <= tile start */
size_t zkir_delta_log_n_mem_events(const zkir_delta_log*);
const zkir_mem_event* zkir_delta_log_mem_events(const zkir_delta_log*);
size_t zkir_delta_log_n_rc_events(const zkir_delta_log*);
const zkir_rc_event* zkir_delta_log_rc_events(const zkir_delta_log*);
size_t zkir_delta_log_n_rc_witnesses(const zkir_delta_log*);
const uint64_t* zkir_delta_log_rc_offsets(const zkir_delta_log*);  /* [n_rc_witnesses+1] CSR over rc_events */
const uint64_t* zkir_delta_log_rc_cycles(const zkir_delta_log*);   /* [n_rc_witnesses] cycle of the checkpoint that produced witness k (vm.rs:316-344) */
uint32_t zkir_delta_log_rc_chunk_bits(const zkir_delta_log*);      /* header limb_bits / 2 (range_check.rs:29) */
size_t zkir_delta_log_n_norm_events(const zkir_delta_log*);
const zkir_norm_event* zkir_delta_log_norm_events(const zkir_delta_log*);
size_t zkir_delta_log_n_sha_blocks(const zkir_delta_log*);
const zkir_sha_block* zkir_delta_log_sha_blocks(const zkir_delta_log*);

/* ---- device stage: kernels on caller-owned device memory ------------------------------------- */

/* The wide execution trace in HBM, struct-of-arrays (one contiguous column per field per register;
 * TraceRow schema zkir-spec/src/trace.rs:24-50).  Register-indexed arrays are [16][reg_stride]
 * elements, column r starting at base + r*reg_stride. 372 bytes per row in total. */
typedef struct zkir_trace_columns {
  uint64_t* cycle;          /* [n_rows]                                        8 B/row  */
  uint64_t* pc;             /* [n_rows]  (may alias the uploaded delta-log pc)  8 B/row  */
  uint32_t* instruction;    /* [n_rows]  (may alias the uploaded delta-log inst)4 B/row  */
  uint64_t* registers;      /* [16][reg_stride]                               128 B/row */
  uint32_t* bound_bits;     /* [16][reg_stride]                                64 B/row */
  uint8_t* bound_tag;       /* [16][reg_stride]                                16 B/row */
  uint64_t* bound_payload;  /* [16][reg_stride]                               128 B/row */
  uint8_t* reg_state;       /* [16][reg_stride]                                16 B/row */
  uint64_t reg_stride;      /* elements between consecutive register columns (>= n_rows, multiple of 16) */
} zkir_trace_columns;

typedef struct zkir_trace_fill_args {
  const zkir_reg_event* events;   /* device, [n_events] */
  const uint32_t* tile_ev_off;    /* device, [n_tiles+1] */
  const uint32_t* tile_snap;      /* device, [n_tiles][16] */
  uint64_t n_rows;
  uint64_t cycle_base;            /* TraceRow.cycle of row 0 (row-sharding across GPUs) */
  uint32_t tile_rows;
  uint32_t n_events;
  zkir_trace_columns out;         /* device */
} zkir_trace_fill_args;

/* K1: expand the register-write log into the SoA trace columns (cycle, registers, bounds, states).
 * Asynchronous on `hip_stream` (a hipStream_t; NULL = default stream). */
int zkir_trace_fill_launch(const zkir_trace_fill_args* args, void* hip_stream);
/* The same for tiles [tile_begin, tile_end) only (args describe the whole trace): used to fill the tiles whose log has already been
 * uploaded while the host interpreter is still producing the rest (zkir_exec does this internally). */
int zkir_trace_fill_range_launch(const zkir_trace_fill_args* args, uint64_t tile_begin, uint64_t tile_end, void* hip_stream);

/* Algorithmic HBM bytes one zkir_trace_fill_launch moves: 360*n_rows written +
 * 32*n_events + 68*n_tiles read. */
uint64_t zkir_trace_fill_bytes(uint64_t n_rows, uint64_t n_events, uint64_t n_tiles);

/* ---- witness expansion kernels (witness.hip); every pointer is device memory, all launches async ----- */

/* MemoryOp columns (trace.rs:149-167), 39 bytes per op; each array has n_ops elements */
typedef struct zkir_memop_columns {
  uint64_t* address;
  uint64_t* value;
  uint64_t* timestamp;
  uint8_t* is_write;       /* MemOpType: 0 Read, 1 Write */
  uint8_t* width;
  uint32_t* bound_bits;    /* ValueBound::from_type_width(8*width), memory.rs:245 */
  uint8_t* bound_tag;
  uint64_t* bound_payload;
} zkir_memop_columns;

/* TraceRow.memory_ops flattened in row order */
int zkir_memops_expand_launch(const zkir_mem_event* events, uint64_t n_ops, uint64_t cycle_base, const zkir_memop_columns* out, void* hip_stream);
/* TraceRow.memory_ops in row order, the CSR row offsets (n_rows + 1 entries) and the per-row shape flags the sort needs (n_rows bytes;
 * 1 = the row's ops are not [ascending reads][ascending writes]) — all three from ONE pass over the events (24 B read per op) */
int zkir_memops_expand_csr_launch(const zkir_mem_event* events, uint64_t n_ops, uint64_t n_rows, uint64_t cycle_base, const zkir_memop_columns* out,
                                  uint64_t* row_offsets, uint8_t* seg_flags, void* hip_stream);
/* ExecutionResult::get_memory_trace() given the offsets and flags of zkir_memops_expand_csr_launch: one more pass over the events */
int zkir_memops_sort_prepared_launch(const zkir_mem_event* events, uint64_t n_ops, uint64_t cycle_base, const uint64_t* row_offsets, const uint8_t* seg_flags,
                                     const zkir_memop_columns* out, void* hip_stream);
/* CSR alone: offsets[r] = number of ops whose row < r, r = 0..n_rows (n_rows+1 entries); one binary search per row */
int zkir_memops_row_offsets_launch(const zkir_mem_event* events, uint64_t n_ops, uint64_t n_rows, uint64_t* offsets, void* hip_stream);
/* ExecutionResult::get_memory_trace() (vm.rs:85-94): stable order by (timestamp, address, Read<Write).
 * row_offsets from zkir_memops_row_offsets_launch; seg_scratch = n_rows bytes of device scratch. */
int zkir_memops_sort_launch(const zkir_mem_event* events, uint64_t n_ops, uint64_t n_rows, uint64_t cycle_base, const uint64_t* row_offsets,
                            uint8_t* seg_scratch, const zkir_memop_columns* out, void* hip_stream);

/* RangeCheckWitness entries (range_check.rs:175-192,212): value[n], pc[n], chunks[4][chunk_stride] (u16) and, if
 * multiplicity != NULL, the lookup multiplicities of all chunks: multiplicity[2^chunk_bits] (zeroed by the call). */
int zkir_range_check_expand_launch(const zkir_rc_event* events, uint64_t n, uint32_t chunk_bits, uint64_t* value, uint64_t* pc,
                                   uint16_t* chunks, uint64_t chunk_stride, uint32_t* multiplicity, void* hip_stream);

/* NormalizationEvent columns (normalization_witness.rs:19-43; normalized_bits = 20, limb_bits = 30, cause = ObservationPoint) */
typedef struct zkir_norm_columns {
  uint64_t* cycle; uint64_t* pc; uint8_t* reg; uint8_t* opcode;
  uint64_t* accumulated0; uint64_t* accumulated1;
  uint32_t* normalized0; uint32_t* normalized1;
  uint32_t* carry0; uint32_t* carry1;
} zkir_norm_columns;
int zkir_norm_expand_launch(const zkir_norm_event* events, uint64_t n, const zkir_norm_columns* out, void* hip_stream);

/* SHA-256 chip (Sha256Witness, trace.rs:236-256; crypto.rs:142-207): out = 608 word-columns [608][stride] (u32):
 * [0,16) message_block, [16,24) initial_state, [24,88) message_schedule, [88,600) round_states[64][8], [600,608) final_state;
 * timestamps[n] optional. */
int zkir_sha256_chip_launch(const zkir_sha_block* blocks, uint64_t n, uint32_t* out, uint64_t stride, uint64_t* timestamps, void* hip_stream);

/* ---- prover stages over Baby Bear (stark.hip, ntt.hip, verify.cpp) ---------------------------------
 * NOT in the reference (no prove(), no Plonky3: Cargo.toml:67-69) => self-defined
 * ("ZKIR-STARK v1"), parity unpinned; spec = oracle/stark_oracle.cpp, frozen by tests/golden/stark_goldens.json.
 * Field elements are canonical u32 (< p = 2^31 - 2^27 + 1) at rest.
 * Matrix layout "B8": the columns of a matrix with `width` columns and n rows are grouped in ceil(width/8) blocks of 8; block b is
 * the array [n][8] of u32 (32 contiguous bytes per row position: columns 8b..8b+7), blocks follow each other; element (column k,
 * row j) is word ((k/8)*n + j)*8 + k%8.  Columns past `width` in the last block are zero.  Every kernel moves 16-byte vectors and
 * the NTT shares its twiddles across the columns a lane carries; one block is one absorption of the rate-8 Poseidon2 sponge.
 * Demonstrator parameters: blow-up 2, 50 FRI queries + 12 bits of grinding (~62 bits, conjectured), Poseidon2 width 12 with
 * capacity 4 (~62-bit collisions).  What the AIR does and does not constrain is stated in zkir_amd/csrc/air.h. */
typedef struct zkir_stark_ctx zkir_stark_ctx;     /* device tables (twiddles, coset powers, Poseidon2 constants) + workspace for 2^log_n rows.
                                                     One proof at a time per context; different contexts are independent (no process-wide state). */
int zkir_stark_ctx_create(uint32_t log_n, uint32_t log_blowup /* must be 1 */, zkir_stark_ctx** out);
void zkir_stark_ctx_free(zkir_stark_ctx* ctx);
uint32_t zkir_main_trace_width(void);             /* 152: COMMITTED main-trace columns of a default-mode run (the AIR's 172 logical columns minus the ones that
                                                     are identically zero there: R0's limbs and the 16 storage states; zkir_amd/csrc/air.h) */
uint32_t zkir_main_trace_width_for(uint32_t deferred);   /* 152 (deferred = 0) / 168 (VMConfig.enable_deferred_model: the storage states are committed) */
uint32_t zkir_padded_log_n(uint64_t n_real);      /* log2 of the padded trace length: max(3, ceil(log2(n_real))) */
/* trace columns (K1 output, n_real executed rows) -> main trace matrix (B8: zkir_main_trace_width_for(deferred) / 8 blocks [N][8]), N = 2^zkir_padded_log_n(n_real): rows past n_real are
 * padding (class "pad": state of the last executed row, cycle keeps counting).  deferred = VMConfig.enable_deferred_model of the run. */
int zkir_main_trace_launch(const zkir_trace_columns* trace, uint64_t n_real, uint32_t deferred, uint32_t* out, void* hip_stream);
/* The main trace of MODE 2 (default VM mode + the I/O argument: 160 committed columns): ECALL rows are dispatched on R10, every row shows the counters of the WRITE ecalls /
 * consumed inputs before it — a prefix count over the rows (scratch: 2 N + 2 (N / 1024 + 2) words of device memory, N = the padded row count).  inputs = the input tape ON
 * THE DEVICE (a live READ row's written value is looked up against it). */
typedef struct zkir_io_args { const uint64_t* inputs; uint64_t n_inputs; uint64_t writes_before, reads_before; } zkir_io_args;
int zkir_main_trace_io_launch(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, uint32_t* scratch, uint32_t* out, void* hip_stream);
/* the same on the HOST (host pointers everywhere; no scratch): a test entry point like zkir_main_trace_host */
int zkir_main_trace_io_host(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, uint32_t* out);
/* The main trace of MODE 3 (mode 2 + the memory argument, the bitwise opcodes, the shifts and MUL: 264 committed columns): load / store rows also show the accessed window, the cell's bytes before the access, the
 * time of its previous access and the pieces of the value moved.  mem_old / mem_told: [n_real] DEVICE arrays (zkir_memcheck_witness_of computes them on the host: memory is a
 * sequential chain); scratch as zkir_main_trace_io_launch.  _host: host pointers everywhere, a test entry point. */
int zkir_main_trace_mem_launch(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, const uint64_t* mem_old, const uint32_t* mem_told, uint32_t* scratch, uint32_t* out,
                               void* hip_stream);
int zkir_main_trace_mem_host(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, const uint64_t* mem_old, const uint32_t* mem_told, uint32_t* out);
/* The main trace of MODE 4 (round 6: mode 3 + the wide-arithmetic class — MULH / DIVU / REMU / DIV / REM, execute.rs:101-183: by a chunk relation on operands below 2^40, through the verifier-recomputed WIDE TAPE on raw 64-bit ones — + hash syscalls as a
 * tape + the code segment's boundary cell: 288 committed columns, six more 10-bit range values per row); arguments as zkir_main_trace_mem_launch / _host, plus the program's
 * code_size (header bytes 16..20): when it is 4 modulo 8 the last code word shares an 8-byte cell with the first data bytes, and stores into that cell's low half have no proof. */
int zkir_main_trace_wide_launch(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, const uint64_t* mem_old, const uint32_t* mem_told, uint64_t code_size,
                                uint32_t* scratch, uint32_t* out, void* hip_stream);
int zkir_main_trace_wide_host(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, const uint64_t* mem_old, const uint32_t* mem_told, uint64_t code_size, uint32_t* out);
/* the same rows computed on the HOST (trace = host pointers, out = host buffer, same B8 layout): the kernel's per-row code is one host + device
 * function, so the CPU test suite checks it against the oracle without a GPU.  A test / diagnostic entry point — the product never calls it
 * (there is no CPU fallback). */
int zkir_main_trace_host(const zkir_trace_columns* trace, uint64_t n_real, uint32_t deferred, uint32_t* out);
/* (the measurement probes and kernel experiments the library also exports — ALU / HBM-copy peaks, the fused first blocks, the strided-pass tilings — are declared in
 * zkir_amd_experimental.h: not part of the drop-in boundary) */
/* Test entry points of the AIR evaluation as the quotient kernel runs it (stark_prove.inl: QuotientOps — lazy 32-bit arithmetic, 96-bit sums, one
 * accumulator per row selector), host builds of the same code; nothing in the product calls them.
 * zkir_air_eval_host: sum_c alpha^c C_c (canonical E4 -> out4) of one (row, next row) pair given as LOGICAL columns (172 main, 40 aux; canonical
 * words), lookup parameters lk[56] (alpha, lambda^0..11, T / N), selector values, the two boundary states and alpha.
 * zkir_air_check_bounds: the static soundness check of that arithmetic on the constraint list (air.h: BoundOps): 0 = sound, else the broken rule. */
void zkir_air_eval_host(const uint32_t* loc, const uint32_t* nxt, const uint32_t* aloc, const uint32_t* anxt, const uint32_t* lk, uint32_t is_first, uint32_t is_last, uint32_t is_trans,
                        const uint32_t* first68, const uint32_t* last68, const uint32_t* alpha4, uint32_t mode /* 0 default, 1 deferred, 2 default + I/O: 180 / 48 columns, lk[57] */,
                        const uint32_t* cnt4 /* mode 2: (oc, ic) of the first and of the last row; else NULL */, uint32_t* out4);
int zkir_air_check_bounds(uint32_t deferred, char* why, size_t why_len);
/* per-column low-degree extension: in = B8 matrix with N rows (evaluations over <w_N>, natural order; CLOBBERED as scratch when N >= 1024)
 * -> out = B8 matrix with 2N rows = evaluations over the coset 31*<w_2N>, natural order.  All 8 columns of every block are transformed. */
int zkir_lde_launch(const zkir_stark_ctx* ctx, uint32_t* in, uint32_t width, uint32_t* out, void* hip_stream);
/* Poseidon2-12 Merkle tree over the n_leaves rows of the B8 matrix `mat` (leaf j = sponge over the `width` real columns of row j);
 * tree = 4*(2*n_leaves-1) words, leaf digests first, root = last 4 words */
int zkir_merkle_commit_launch(const zkir_stark_ctx* ctx, const uint32_t* mat, uint32_t width, uint64_t n_leaves, uint32_t* tree, void* hip_stream);

/* The two halves of zkir_merkle_commit_launch, for callers that time or schedule them separately: the leaf digests (tree[0..4n), one
 * sponge per row: leaf_hash_kernel, the dominant kernel of the commit step), then zkir_merkle_cap_launch(ctx, tree, n_leaves) for the
 * levels above them. */
int zkir_merkle_leaves_launch(const zkir_stark_ctx* ctx, const uint32_t* mat, uint32_t width, uint64_t n_leaves, uint32_t* digests, void* hip_stream);

/* Top of a row-sharded commitment: tree[0..4n) holds n (power of two) digests — the all-gathered subtree roots of the row
 * shards, in rank order — and the call appends the log2(n) upper levels; root = last 4 words of the 4*(2n-1)-word buffer. */
int zkir_merkle_cap_launch(const zkir_stark_ctx* ctx, uint32_t* tree, uint64_t n_digests, void* hip_stream);

/* Public inputs of a proof: absorbed by the Fiat-Shamir transcript before the first commitment and carried in the proof header. */
typedef struct zkir_public_inputs {
  uint64_t n_real;             /* executed rows = ExecutionResult.cycles */
  uint64_t entry_point;        /* ProgramHeader.entry_point: pc of row 0 (constrained) */
  uint32_t deferred;           /* the proof's MODE: 0 = default VM mode, 1 = VMConfig.enable_deferred_model (relaxed AIR), 2 = default mode + the I/O argument (below), 3 = 2 + the memory argument */
  uint32_t fri_params;         /* the prover's parameters: num_queries | pow_bits << 16; 0 = the defaults (50 queries, 12 grinding bits).  Set with zkir_public_inputs_set_params.
                                * They are header words 4 and 6 of the proof (observed by the transcript); a verifier with an `expect` requires exactly these. */
  uint32_t program_digest[4];  /* zkir_digest_bytes(program blob) */
  uint32_t io_digest[4];       /* zkir_digest_bytes(LE u64 words [n_inputs, inputs.., n_outputs, outputs.., halt kind, halt code, cycles]) */
  /* PROVER side only (ignored when the struct is the `expect` of a verifier): the program itself, BORROWED — set by
   * zkir_public_inputs_of to the caller's blob, which must stay valid while the struct is passed to zkir_prove.  Its code words are
   * the instruction ROM of the lookup argument; the proof carries the program (format v5 on; zkir_proof_version() = the current format), the verifier checks it against program_digest. */
  const uint8_t* program_blob;
  uint64_t program_blob_len;
  /* MODE 2 (`deferred` == 2, round 4): the default VM mode WITH the I/O argument — WRITE / READ ecalls are tied by a lookup to the tapes below, which the proof then
   * carries (their digest, with the halt reason and the cycle count, is io_digest); the verifier also checks that the run ends on the instruction the halt reason names.
   * PROVER side (BORROWED pointers, filled by zkir_public_inputs_of; ignored in a verifier's `expect`).  For a SEGMENT of a run the caller sets writes_before /
   * reads_before: the WRITE / READ ecalls the run executed before the segment's first row. */
  const uint64_t* inputs;
  uint64_t n_inputs;
  const uint64_t* outputs;
  uint64_t n_outputs;
  uint32_t halt_kind;          /* ZKIR_HALT_* */
  uint32_t reserved2;
  uint64_t halt_code;
  uint64_t writes_before, reads_before;
  /* MODE 3 (`deferred` == 3, round 4): mode 2 WITH the memory argument — loads and stores are constrained and every access is tied to a consistent memory (air.h "MODE 3").
   * PROVER side (ignored in a verifier's `expect`).  mem_old == NULL (what zkir_public_inputs_of leaves): zkir_prove computes the run's memory witness ON THE DEVICE
   * (memcheck.hip: the accesses sorted address-major, a segmented scan per cell).  Otherwise BORROWED pointers set by zkir_public_inputs_set_memory from a zkir_memcheck_witness
   * (the host's independent replay): per ROW the bytes of the accessed 8-byte cell before the access and the time of the cell's previous access (0 on rows that are no load /
   * store), and the touched cells by increasing address.  The proof carries the touched cells either way. */
  const uint64_t* mem_old;     /* [n_real] */
  const uint32_t* mem_told;    /* [n_real] */
  const uint64_t* cell_addr;   /* [n_cells] multiples of 8 below 2^40, strictly increasing */
  const uint64_t* cell_bytes;  /* [n_cells] the cell's final bytes, little-endian */
  const uint32_t* cell_time;   /* [n_cells] the time of its last access = that row's cycle + 1 */
  uint64_t n_cells;
  /* MODE 4 (`deferred` == 4, round 6): mode 3 WITH the wide-arithmetic class (MULH / DIVU / REMU / DIV / REM: constrained by a chunk relation on operands below 2^40; on raw 64-bit operands — `as i64`,
   * 128-bit products — through the WIDE TAPE: one record (cycle, rs1, rs2, opcode) per such row, gathered by zkir_prove itself, the verifier computes the result), the code segment's boundary cell, and
   * HASH SYSCALLS as a tape: the proof carries one record per SHA-256 / Keccak-256 / BLAKE3 call (cycle, pointers, length, kind, and per touched 8-byte cell its bytes before
   * the call and the time of its previous access) and the verifier computes every digest itself.  PROVER side: hash_section = those records in the proof's own word layout
   * (csrc/hashcall.h), BORROWED from a zkir_memcheck_witness made with zkir_memcheck_witness_of_mode(.., 4, ..) (zkir_public_inputs_set_memory sets it).  A run that makes
   * hash calls is proven from that host witness: the device witness (mem_old == NULL) covers loads and stores only, and zkir_prove refuses such a run without the section. */
  const uint32_t* hash_section;
  uint64_t hash_section_words;
} zkir_public_inputs;
/* (mode 3) The memory witness of a WHOLE run (host, sequential like the interpreter: memory is a chain — what a load returns depends on every earlier store): the
 * log's rows are replayed with their register state (rebuilt from the register events), every load / store looks up its aligned 8-byte cell — the program image at first
 * (code at 0x1000, data behind it: vm.rs:153-170), zero elsewhere — and records the cell's bytes and the time of its previous access.  Refused (ZKIR_ERR_ARGUMENT): a shard /
 * window, an address of 2^40 or more (addr_limbs = 2, config.rs:30), an executed hash syscall (its memory effect is not stated by the AIR).  Free with zkir_memcheck_witness_free. */
typedef struct zkir_memcheck_witness zkir_memcheck_witness;
int zkir_memcheck_witness_of(const zkir_delta_log* log, const uint8_t* program_blob, size_t blob_len, zkir_memcheck_witness** out);
/* the same for a given AIR mode: 3 = zkir_memcheck_witness_of; 4 (round 6) also replays the run's hash syscalls (SHA-256 / Keccak-256 / BLAKE3: the message read out of the
 * replayed memory, the digest computed and laid over the output range) and records them as the proof's hash section (zkir_public_inputs::hash_section) */
int zkir_memcheck_witness_of_mode(const zkir_delta_log* log, const uint8_t* program_blob, size_t blob_len, uint32_t mode, zkir_memcheck_witness** out);
uint64_t zkir_memcheck_witness_n_hash_calls(const zkir_memcheck_witness* w);
void zkir_memcheck_witness_free(zkir_memcheck_witness* w);
uint64_t zkir_memcheck_witness_n_cells(const zkir_memcheck_witness* w);
uint64_t zkir_memcheck_witness_n_accesses(const zkir_memcheck_witness* w);
/* The same witness made ON THE DEVICE (memcheck.hip: what zkir_prove runs in mode 3 when the public inputs bring no witness), as a call of its own: trace = the DEVICE columns of
 * a whole run, the outputs are HOST arrays (mem_old / mem_told: n_real entries, zero where the row is no load / store; the cell arrays: capacity `cap`, *n_cells = the count). */
int zkir_memcheck_witness_device(const zkir_trace_columns* trace, uint64_t n_real, const uint8_t* program_blob, size_t blob_len, uint64_t* mem_old, uint32_t* mem_told,
                                 uint64_t* cell_addr, uint64_t* cell_bytes, uint32_t* cell_time, uint64_t cap, uint64_t* n_cells, void* hip_stream);
/* points pub's mode-3 fields at the witness (which must outlive the proving call) and sets pub->deferred = 3 */
void zkir_public_inputs_set_memory(zkir_public_inputs* pub, const zkir_memcheck_witness* w);
/* Poseidon2 sponge digest of a byte string (host): [len as four 16-bit pieces] ++ [LE 16-bit halfwords] */
void zkir_digest_bytes(const uint8_t* bytes, size_t len, uint32_t out[4]);
/* the public inputs of a finished run (host; `log`: the run's delta log, a shard of it, or the trace window that reached the run's
 * halt — anything whose cycles / outputs / halt reason are the finished run's; a window with zkir_delta_log_window_open() is refused) */
int zkir_public_inputs_of(const zkir_delta_log* log, const uint8_t* program_blob, size_t blob_len, const uint64_t* inputs, size_t n_inputs,
                          uint32_t deferred, zkir_public_inputs* out);

/* The prover's parameters (zkir_prover_params).  mode: the proof's mode, the same value as zkir_public_inputs.deferred (0 default VM mode, 1 deferred carry model, 2 default +
 * the I/O argument, 3 = 2 + the memory argument).  num_queries / pow_bits: FRI queries and grinding bits, 0 = the defaults (50, 12); accepted: 50..128 queries, 12..24 bits (a
 * proof may say more than the defaults, never less).  Conjectured FRI soundness at blow-up 2 is one bit per query + the grinding bits: 50 + 12 = 62, 84 + 16 = 100 — the
 * capacity-4 Poseidon2 sponge (rate 8, digests of 4 x 31 bits) caps collision resistance at ~62 bits whatever these say (README: a demonstrator instance). */
typedef struct zkir_result zkir_result;   /* the drop-in layer's handle (zkir_exec, below) */
typedef struct zkir_prover_params {
  uint32_t mode;
  uint32_t num_queries;
  uint32_t pow_bits;
} zkir_prover_params;
/* validates `params` (ZKIR_ERR_ARGUMENT otherwise: the ranges above; params->mode must be pub->deferred) and records the FRI parameters in pub->fri_params */
int zkir_public_inputs_set_params(zkir_public_inputs* pub, const zkir_prover_params* params);

/* Full proof of the execution whose K1 output is `trace` (pub->n_real rows; ctx built for zkir_padded_log_n(pub->n_real)).  *proof_out is a
 * malloc'ed array of u32 words (little-endian canonical field elements; format v10 in modes 0 / 1, v11 in modes 2 / 3 — zkir_proof_version_of_mode; layout in oracle/stark_oracle.cpp so::prove),
 * pub->program_blob must be the program that ran: every row's (pc, instruction word) is looked up in its code table, and a run that executes
 * anything else (self-modified code, a pc outside the code segment) is refused with ZKIR_ERR_ARGUMENT — it has no proof in this AIR.
 * released with zkir_proof_free.  stage_ms (NINE floats, nullable): main trace, LDE, trace Merkle, lookup argument (aux trace + its LDE and tree), quotient, openings, DEEP, FRI, queries.
 * The trace may be that of a whole run or of a SEGMENT of one (the K1 output of a row shard, zkir_delta_log_shard(log, a, b) with
 * cycle_base = a): the proof header records the 68-word state (cycle, pc limbs, register limbs, storage states) of the first and of
 * the last row and the AIR pins those rows to it; pub->n_real is then the segment's row count, pub->io_digest the RUN's. */
int zkir_prove(const zkir_stark_ctx* ctx, const zkir_trace_columns* trace, const zkir_public_inputs* pub, uint32_t** proof_out, uint64_t* proof_words,
               float* stage_ms, void* hip_stream);
void zkir_proof_free(uint32_t* proof);
/* SURVEY 8(b)'s shape of the call: the WHOLE-RUN handle of zkir_exec in (enable_execution_trace), the proof out as bytes (the little-endian u32 words of zkir_prove; free with
 * zkir_proof_bytes_free).  params may be NULL (mode 0, 50 queries, 12 bits).  Makes and frees a context for the run's size: a caller proving many runs keeps one and uses zkir_prove. */
int zkir_prove_result(const zkir_result* result, const zkir_prover_params* params, uint8_t** proof, size_t* proof_len);
void zkir_proof_bytes_free(uint8_t* proof);
uint32_t zkir_proof_num_queries(void);            /* the default (50) */
uint32_t zkir_proof_version(void);                /* of modes 0 / 1 (10) */
uint32_t zkir_proof_version_of_mode(uint32_t mode);   /* modes 2 / 3: 11 (round 5: EBREAK is a class of its own, the I/O section is in the transcript, mode 3 refuses accesses to the code segment); mode 4: 12 (round 6) */
/* Verifier of the proof of a WHOLE run (host only, no device): 0 = accepted, otherwise the number of the failed check (1-5
 * malformed, 6 public inputs differ from `expect`, 7 the run does not start in the VM's initial state (cycle 0, entry point, zero
 * registers), 8 the program carried in the proof is malformed or is not the one program_digest / entry_point name, 10 constraints at
 * zeta (incl. the lookup argument: the verifier computes the table side from that program and the multiplicities in the proof), 11 final
 * codeword degree, 12 grinding, 20-27 query / Merkle / FRI checks, 30 length; modes 2 / 3 also 50-53 = zkir_verify_io's checks on the tapes the proof carries, 51 the
 * counters' ends; mode 3 also 54 = the touched cells are not canonical 8-byte cell addresses in strictly increasing order; a mode-3 proof is never a segment: 2;
 * modes 3 / 4 also 55 = a touched cell / a hash call's output overlaps the code segment (mode 4 admits the boundary cell); mode 4 also 56 = a malformed hash-tape record
 * (ranges, order, cell count, a previous access that is not before the call), 57 = a malformed wide-tape record (a limb out of range, not an opcode 3..7, a zero divisor, cycles
 * not increasing); a tape whose records are well-formed but are not the run's fails the lookup argument (10)).
 * expect may be NULL: the header's own public inputs are then only checked for internal consistency.
 * The queries (independent, ~10,000 Poseidon2 permutations of Merkle paths at 2^20 rows) are checked on up to eight host threads (half the logical cores; ZKIR_VERIFY_THREADS=n
 * overrides); the verdict is the first failing query's in query order, as a sequential check would give. */
int zkir_verify(const uint32_t* proof, uint64_t proof_words, const zkir_public_inputs* expect);
/* A run proven in SEGMENTS (multi-GPU: one row shard per device; consecutive segments overlap by one row — the last row of segment i,
 * labelled "halt" there, is row 0 of segment i + 1).  zkir_verify_segment: the same checks without check 7; first_state / last_state
 * (68 words each, nullable) receive the header's boundary states.  zkir_verify_chain: every segment verifies, the first starts in the
 * initial state, each later one starts in exactly the state its predecessor ended in (the cycle counter is part of the state), mode /
 * entry point / program digest / io digest agree; expect (nullable) = the RUN's public inputs, n_real = its total rows
 * = sum(n_i - 1) + 1.  0 = accepted; 40 empty, 41 first state, 42 link, 43 public inputs, 44 row count; 1000 (i + 1) + c = check c
 * of segment i. */
int zkir_verify_segment(const uint32_t* proof, uint64_t proof_words, const zkir_public_inputs* expect, uint32_t first_state[68], uint32_t last_state[68]);
int zkir_verify_chain(const uint32_t* const* proofs, const uint64_t* proof_words, uint32_t n_segments, const zkir_public_inputs* expect);
/* The run's CLAIM in the clear on top of zkir_verify / zkir_verify_chain (round 4): the I/O tapes and the halt reason (ZKIR_HALT_*, exit code) must hash to the proof's io
 * digest together with the proof's row count (else 50), and the HALT ROW — the last executed row, pinned to the public last state by the AIR — must be the instruction the
 * halt reason names, looked up in the program the proof carries: EBREAK for an Ebreak halt, ECALL reading R10 = 0 and R11 = the exit code for Exit(code) (52: another
 * instruction; 53: another syscall / exit code) — vm.rs:302-347, syscall.rs:101-107.  CycleLimit names no instruction.  The outputs themselves stay unproven (air.h). */
int zkir_verify_io(const uint32_t* proof, uint64_t words, const zkir_public_inputs* expect, const uint64_t* inputs, size_t n_inputs, const uint64_t* outputs, size_t n_outputs,
                   int halt_kind, uint64_t halt_code);
int zkir_verify_chain_io(const uint32_t* const* proofs, const uint64_t* words, uint32_t n_segments, const zkir_public_inputs* expect, const uint64_t* inputs, size_t n_inputs,
                         const uint64_t* outputs, size_t n_outputs, int halt_kind, uint64_t halt_code);
uint32_t zkir_proof_state_words(void);
/* Host-side Poseidon2-12 permutation of the transcript (canonical words in and out; no device needed): what a verifier or an
 * integrator re-deriving the Fiat-Shamir challenges calls.  Same code as the device kernels (poseidon2.h), compiled for the host. */
void zkir_poseidon2_permute(uint32_t state[12]);
/* The same permutation in the formulation the hash kernels run (scaled state words, poseidon2.h: permute_scaled), host build:
 * `rounds` chained applications, canonical words in and out.  Exists so that the formulation can be checked without a device. */
void zkir_poseidon2_permute_scaled(uint32_t state[12], uint32_t rounds);

/* ---- drop-in layer: VM::new + VM::run --------------------------------------------------------- */
/* (zkir_result: the opaque handle declared above with zkir_prove_result; owns host metadata + device columns) */

/* program_blob is Program::to_bytes() (program.rs:300-315).  On success the trace columns are resident
 * on the current HIP device.  Fails with ZKIR_ERR_DEVICE if no GPU is usable (no CPU fallback). */
int zkir_exec(const uint8_t* program_blob, size_t blob_len, const uint64_t* inputs, size_t n_inputs,
              const zkir_vm_config* cfg, zkir_result** out);
/* The same call for ONE GPU'S SHARE of a run: rows [0, row_begin) are executed untraced on the calling thread, rows [row_begin,
 * row_end) are traced, uploaded and filled on the current device while the interpreter runs (zkir_interpret_window + the streaming
 * of zkir_exec).  The handle's trace columns hold absolute cycles; zkir_result_delta_log(r): cycle_base = row_begin. */
int zkir_exec_window(const uint8_t* program_blob, size_t blob_len, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config* cfg,
                     uint64_t row_begin, uint64_t row_end, zkir_result** out);
/* The same handle for a ROW SHARD of a finished interpretation (zkir_interpret): rows [row_begin, row_end) of `log` are cut out
 * (zkir_delta_log_shard), uploaded to the CURRENT HIP device and filled there.  Multi-GPU: one call per device, each with its row range
 * (hipSetDevice / one process per GPU); segment proofs: ranges that share one row.  The result owns the shard (zkir_result_delta_log:
 * cycle_base = row_begin); trace columns hold absolute cycles; the witness accessors describe the shard's rows. */
int zkir_exec_shard(const zkir_delta_log* log, uint64_t row_begin, uint64_t row_end, zkir_result** out);
void zkir_result_free(zkir_result* r);
const zkir_delta_log* zkir_result_delta_log(const zkir_result* r);       /* host-side metadata */
const zkir_trace_columns* zkir_result_trace(const zkir_result* r);       /* device pointers */
/* wall-clock breakdown of the zkir_exec call that produced r (ms): host interpretation | device allocation | H2D of the delta log |
 * K1 launch + synchronisation */
void zkir_result_stage_ms(const zkir_result* r, float out[4]);
/* copy one device column to host: field = 0 cycle,1 pc,2 instruction,3 registers,4 bound_bits,5 bound_tag,
 * 6 bound_payload,7 reg_state; reg ignored for fields 0-2.  dst must hold n_rows elements. */
int zkir_result_copy_column(const zkir_result* r, int field, int reg, void* dst);


/* ---- the rest of ExecutionResult (vm.rs:54-103) behind the handle: witness streams as DEVICE columns ---------------------------
 * Each call expands the corresponding side log of the run on the device the first time it is made (witness.hip kernels), caches
 * the columns in the handle and returns the same pointers afterwards; they stay valid until zkir_result_free.  Thread-safe. */
typedef struct zkir_memory_witness {     /* TraceRow.memory_ops of every row + ExecutionResult::get_memory_trace() (vm.rs:85-94) */
  uint64_t n_ops;                        /* = ExecutionResult::memory_op_count() (vm.rs:97-102) */
  uint64_t n_rows;
  zkir_memop_columns row_order;          /* ops in execution order; row r owns ops [row_offsets[r], row_offsets[r+1]) */
  const uint64_t* row_offsets;           /* device, [n_rows+1] */
  zkir_memop_columns sorted;             /* stable order by (timestamp, address, Read<Write): trace.rs:210-223 */
} zkir_memory_witness;
int zkir_result_memory_trace(zkir_result* r, zkir_memory_witness* out);

typedef struct zkir_range_check_witness { /* ExecutionResult.range_check_witnesses: Vec<RangeCheckWitness> (range_check.rs:209-238) */
  uint64_t n_checks;                     /* total checks over all witnesses */
  uint64_t n_witnesses;                  /* non-empty checkpoints (vm.rs:340-342) */
  const uint64_t* witness_offsets;       /* HOST, [n_witnesses+1]: witness k owns checks [off[k], off[k+1]) */
  const uint64_t* witness_cycles;        /* HOST, [n_witnesses]: cycle of the checkpoint */
  const uint64_t* value;                 /* device [n_checks] */
  const uint64_t* pc;                    /* device [n_checks] */
  const uint16_t* chunks;                /* device [4][chunk_stride]: chunk c of check i at chunks[c*chunk_stride + i] (range_check.rs:175-192) */
  uint64_t chunk_stride;
  uint32_t chunk_bits;
  const uint32_t* multiplicity;          /* device [2^chunk_bits]: lookup multiplicities of all chunks */
} zkir_range_check_witness;
int zkir_result_range_check_witnesses(zkir_result* r, zkir_range_check_witness* out);

typedef struct zkir_normalization_witness { /* ExecutionResult.normalization_witnesses (normalization_witness.rs:129-138) */
  uint64_t n_events;
  zkir_norm_columns columns;             /* device */
} zkir_normalization_witness;
int zkir_result_normalization_witnesses(zkir_result* r, zkir_normalization_witness* out);

typedef struct zkir_sha256_witness {     /* Sha256Witness per single-block SHA-256 syscall (trace.rs:236-285); never reached from VM::run in the
                                            reference (syscall.rs:127 passes None) — BASELINE configs[4]'s "syscall-chip trace columns" */
  uint64_t n_blocks;
  const uint32_t* columns;               /* device [608][stride], layout of zkir_sha256_chip_launch */
  uint64_t stride;
  const uint64_t* timestamps;            /* device [n_blocks] */
} zkir_sha256_witness;
int zkir_result_sha256_witnesses(zkir_result* r, zkir_sha256_witness* out);

/* D2H copy for hosts that do not link a HIP runtime themselves (the pointers above are device memory) */
int zkir_device_to_host(void* host_dst, const void* device_src, size_t bytes);
/* H2D copy of a host-side log array (zkir_delta_log_* pointers) into caller-owned device memory, ordered on `hip_stream`: the copy
 * path of zkir_exec for callers that drive the stages themselves */
int zkir_host_to_device(void* device_dst, const void* host_src, size_t bytes, void* hip_stream);

const char* zkir_last_error(void);
/* The ABI revision of this header: bumped whenever a struct layout, a buffer size or a signature of an EXISTING entry point changes (new entry points alone do not bump it).
 *   4: zkir_prove's stage_ms is NINE floats (round 4 added the lookup stage: a caller built against eight overflows by 4 bytes)
 *   5: zkir_public_inputs.reserved became fri_params (same offset; zero = the old behaviour); zkir_verify* compare the proof's FRI parameters with `expect`'s
 * A binding checks zkir_abi_version() == ZKIR_AMD_ABI_VERSION when it loads the library. */
#define ZKIR_AMD_ABI_VERSION 6u   /* 6 (round 6): zkir_public_inputs grew (hash_section), mode 4 entry points, pinned log blocks */
uint32_t zkir_abi_version(void);
const char* zkir_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ZKIR_AMD_H */
