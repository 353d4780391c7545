// zkir_amd.hpp — C++ host mirror of the reference's interface for the hot path, above the C ABI of zkir_amd.h.
//
// The reference is Rust and its toolchain is absent from the image, so the host side a maintainer would write in
// `zkir-runtime` is given here in C++ with the reference's names, argument meaning and error behaviour:
//
//   zkir_spec::Opcode / encode / Program          zkir-spec/src/opcode.rs:24-144, zkir-assembler/src/encoder.rs:18-151,
//                                                 zkir-spec/src/program.rs:62-346
//   zkir_runtime::VMConfig                        zkir-runtime/src/vm.rs:15-50 (same defaults)
//   zkir_runtime::HaltReason                      zkir-runtime/src/state.rs:8-15
//   zkir_runtime::RuntimeError                    zkir-runtime/src/error.rs:7-37 (kind + the reference's message text)
//   zkir_runtime::VM::new(..).run()               zkir-runtime/src/vm.rs:138-358 (`run` consumes the VM)
//   zkir_runtime::ExecutionResult                 zkir-runtime/src/vm.rs:54-103
//   zkir_runtime::run(program, inputs)            zkir-runtime/src/lib.rs:59-62
//
// Header-only, C++17, no HIP types: link with -lzkir_amd.  With enable_execution_trace the run goes through zkir_exec and the
// trace stays in HBM (ExecutionTrace copies columns/rows to the host on demand); without it only the host interpreter runs.
// tests/cpp/reference_tests.cpp re-states a handful of the reference's own tests against this header.
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "zkir_amd.h"

namespace zkir_spec {

enum class Opcode : uint8_t {                                   // opcode.rs:24-144
  ADD = 0x00, SUB = 0x01, MUL = 0x02, MULH = 0x03, DIVU = 0x04, REMU = 0x05, DIV = 0x06, REM = 0x07, ADDI = 0x08,
  AND = 0x10, OR = 0x11, XOR = 0x12, ANDI = 0x13, ORI = 0x14, XORI = 0x15,
  SLL = 0x18, SRL = 0x19, SRA = 0x1A, SLLI = 0x1B, SRLI = 0x1C, SRAI = 0x1D,
  SLTU = 0x20, SGEU = 0x21, SLT = 0x22, SGE = 0x23, SEQ = 0x24, SNE = 0x25, CMOV = 0x26, CMOVZ = 0x27, CMOVNZ = 0x28,
  LB = 0x30, LBU = 0x31, LH = 0x32, LHU = 0x33, LW = 0x34, LD = 0x35, SB = 0x38, SH = 0x39, SW = 0x3A, SD = 0x3B,
  BEQ = 0x40, BNE = 0x41, BLT = 0x42, BGE = 0x43, BLTU = 0x44, BGEU = 0x45, JAL = 0x48, JALR = 0x49, ECALL = 0x50, EBREAK = 0x51
};

// encoder.rs:100-151; immediates are silently masked to 17 bits (quirk Q11), S/B-type keep rs1 in bits 10:7 and rs2 in 14:11
inline uint32_t enc_r(Opcode op, int rd, int rs1, int rs2) { return (uint32_t)op | (rd & 0xF) << 7 | (rs1 & 0xF) << 11 | (rs2 & 0xF) << 15; }
inline uint32_t enc_i(Opcode op, int rd, int rs1, int32_t imm) { return (uint32_t)op | (rd & 0xF) << 7 | (rs1 & 0xF) << 11 | ((uint32_t)imm & 0x1FFFF) << 15; }
inline uint32_t enc_j(Opcode op, int rd, int32_t off) { return (uint32_t)op | (rd & 0xF) << 7 | ((uint32_t)off & 0x1FFFFF) << 11; }

// the reference's `Instruction::X { .. }` vocabulary
inline uint32_t add(int rd, int rs1, int rs2) { return enc_r(Opcode::ADD, rd, rs1, rs2); }
inline uint32_t sub(int rd, int rs1, int rs2) { return enc_r(Opcode::SUB, rd, rs1, rs2); }
inline uint32_t mul(int rd, int rs1, int rs2) { return enc_r(Opcode::MUL, rd, rs1, rs2); }
inline uint32_t div_(int rd, int rs1, int rs2) { return enc_r(Opcode::DIV, rd, rs1, rs2); }
inline uint32_t divu(int rd, int rs1, int rs2) { return enc_r(Opcode::DIVU, rd, rs1, rs2); }
inline uint32_t addi(int rd, int rs1, int32_t imm) { return enc_i(Opcode::ADDI, rd, rs1, imm); }
inline uint32_t slli(int rd, int rs1, int shamt) { return enc_i(Opcode::SLLI, rd, rs1, shamt); }
inline uint32_t lw(int rd, int rs1, int32_t imm) { return enc_i(Opcode::LW, rd, rs1, imm); }
inline uint32_t sw(int rs1, int rs2, int32_t imm) { return enc_i(Opcode::SW, rs1, rs2, imm); }     // mem[rs1 + imm] = rs2
inline uint32_t beq(int rs1, int rs2, int32_t off) { return enc_i(Opcode::BEQ, rs1, rs2, off); }
inline uint32_t bne(int rs1, int rs2, int32_t off) { return enc_i(Opcode::BNE, rs1, rs2, off); }
inline uint32_t jal(int rd, int32_t off) { return enc_j(Opcode::JAL, rd, off); }
inline uint32_t ecall() { return (uint32_t)Opcode::ECALL; }
inline uint32_t ebreak() { return (uint32_t)Opcode::EBREAK; }

// program.rs:62-346: 32-byte little-endian header + code words + data bytes
struct Program {
  uint8_t limb_bits = 20, data_limbs = 2, addr_limbs = 2, flags = 0;
  uint32_t entry_point = 0x1000, bss_size = 0, stack_size = 1u << 20;
  std::vector<uint32_t> code;
  std::vector<uint8_t> data;

  static Program from_code(std::vector<uint32_t> words) { Program p; p.code = std::move(words); return p; }

  std::vector<uint8_t> to_bytes() const {                       // program.rs:300-315
    std::vector<uint8_t> b(32 + 4 * code.size() + data.size());
    auto put32 = [&](size_t at, uint32_t v) { for (int i = 0; i < 4; i++) b[at + i] = (uint8_t)(v >> (8 * i)); };
    put32(0, 0x52494B5Au); put32(4, 0x00030004u);
    b[8] = limb_bits; b[9] = data_limbs; b[10] = addr_limbs; b[11] = flags;
    put32(12, entry_point); put32(16, (uint32_t)(4 * code.size())); put32(20, (uint32_t)data.size()); put32(24, bss_size); put32(28, stack_size);
    for (size_t i = 0; i < code.size(); i++) put32(32 + 4 * i, code[i]);
    if (!data.empty()) std::memcpy(b.data() + 32 + 4 * code.size(), data.data(), data.size());
    return b;
  }
};


// Mersenne31 (zkir-spec/src/field.rs:16-189): the field type the spec crate defines (p = 2^31 - 1), canonical representation.  The
// runtime never uses it (no caller outside the reference's own tests) — mirrored for completeness of the spec surface,
// pinned by the reference's unit tests (field.rs:231-321, re-stated in tests/cpp/reference_tests.cpp).
class Mersenne31 {
 public:
  static constexpr uint32_t PRIME = (1u << 31) - 1;
  constexpr Mersenne31() : v_(0) {}
  constexpr explicit Mersenne31(uint32_t value) : v_(reduce(value)) {}               // Mersenne31::new (field.rs:33-35)
  static constexpr Mersenne31 zero() { return Mersenne31(); }
  static constexpr Mersenne31 one() { return Mersenne31(1); }
  constexpr uint32_t value() const { return v_; }
  constexpr bool is_zero() const { return v_ == 0; }
  constexpr bool is_one() const { return v_ == 1; }
  constexpr Mersenne31 operator+(Mersenne31 r) const { return raw(reduce(v_ + r.v_)); }                       // field.rs:129-136
  constexpr Mersenne31 operator-(Mersenne31 r) const { return raw(reduce(v_ + PRIME - r.v_)); }               // field.rs:145-153
  constexpr Mersenne31 operator*(Mersenne31 r) const { return raw(reduce64((uint64_t)v_ * r.v_)); }           // field.rs:162-169
  constexpr Mersenne31 operator-() const { return v_ == 0 ? Mersenne31() : raw(PRIME - v_); }                 // field.rs:80-86
  Mersenne31& operator+=(Mersenne31 r) { return *this = *this + r; }
  Mersenne31& operator-=(Mersenne31 r) { return *this = *this - r; }
  Mersenne31& operator*=(Mersenne31 r) { return *this = *this * r; }
  constexpr bool operator==(Mersenne31 r) const { return v_ == r.v_; }
  constexpr bool operator!=(Mersenne31 r) const { return v_ != r.v_; }
  constexpr Mersenne31 pow(uint32_t exp) const {                                                              // field.rs:102-115
    Mersenne31 base = *this, result = one();
    while (exp > 0) { if (exp & 1) result = result * base; base = base * base; exp >>= 1; }
    return result;
  }
  Mersenne31 inv() const {                                                                                    // field.rs:92-99 (panics on zero)
    if (v_ == 0) throw std::domain_error("Division by zero in Mersenne31");
    return pow(PRIME - 2);
  }

 private:
  static constexpr Mersenne31 raw(uint32_t canonical) { Mersenne31 m; m.v_ = canonical; return m; }
  static constexpr uint32_t reduce(uint32_t x) {                                                              // field.rs:53-67: (x & p) + (x >> 31), one conditional subtraction
    const uint32_t sum = (x & PRIME) + (x >> 31);
    return sum >= PRIME ? sum - PRIME : sum;
  }
  static constexpr uint32_t reduce64(uint64_t x) { return reduce(((uint32_t)x & PRIME) + (uint32_t)(x >> 31)); }   // field.rs:70-77
  uint32_t v_;
};

}  // namespace zkir_spec

namespace zkir_runtime {

struct VMConfig {                                               // vm.rs:15-50
  uint64_t max_cycles = 1000000;
  bool trace = false, enable_range_checking = false, enable_execution_trace = false, enable_deferred_model = false;
};

struct HaltReason {                                             // state.rs:8-15
  enum Kind { Ebreak = ZKIR_HALT_EBREAK, Exit = ZKIR_HALT_EXIT, CycleLimit = ZKIR_HALT_CYCLE_LIMIT } kind;
  uint64_t code = 0;                                            // Exit(code)
  bool operator==(const HaltReason& o) const { return kind == o.kind && (kind != Exit || code == o.code); }
  bool operator!=(const HaltReason& o) const { return !(*this == o); }
  static HaltReason exit(uint64_t c) { return {Exit, c}; }
  static HaltReason ebreak() { return {Ebreak, 0}; }
  static HaltReason cycle_limit() { return {CycleLimit, 0}; }
};

class RuntimeError : public std::runtime_error {                // error.rs:7-37; what() is the reference's Display text
 public:
  enum Kind { MisalignedAccess = ZKIR_ERR_MISALIGNED, InvalidMemoryAccess = ZKIR_ERR_INVALID_MEMORY, DivisionByZero = ZKIR_ERR_DIV_ZERO,
              InvalidSyscall = ZKIR_ERR_INVALID_SYSCALL, Decode = ZKIR_ERR_DECODE, Other = ZKIR_ERR_OTHER, BadProgram = ZKIR_ERR_BAD_PROGRAM,
              Device = ZKIR_ERR_DEVICE, Argument = ZKIR_ERR_ARGUMENT };
  RuntimeError(int code, const std::string& msg) : std::runtime_error(msg), kind((Kind)code) {}
  Kind kind;
};

namespace detail {
[[noreturn]] inline void raise(int rc) { const char* m = zkir_last_error(); throw RuntimeError(rc, m ? m : "zkir_amd error"); }
}  // namespace detail

struct ValueBound { uint32_t max_bits; uint8_t source_tag; uint64_t source_payload; };   // bound.rs:116-121 (tag = ZKIR_BOUND_*)

struct TraceRow {                                               // trace.rs:24-50 (memory_ops are served by ExecutionResult)
  uint64_t cycle, pc;
  uint32_t instruction;
  std::array<uint64_t, 16> registers;
  std::array<ValueBound, 16> bounds;
  std::array<uint8_t, 16> register_states;                      // 0 Normalized, 1 Accumulated
};

struct MemoryOp {                                               // trace.rs:149-167
  uint64_t address, value, timestamp;
  uint8_t op;                                                   // MemOpType: 0 Read, 1 Write
  uint8_t width;
  ValueBound bound;
  bool is_read() const { return op == 0; }
  bool is_write() const { return op == 1; }
};
struct RangeCheck { uint64_t value; std::array<uint16_t, 4> chunks; uint64_t pc; };   // range_check.rs:209-238 (one entry of a RangeCheckWitness)
using RangeCheckWitness = std::vector<RangeCheck>;
struct NormalizationEvent {                                     // normalization_witness.rs:19-43, :129-138
  uint64_t cycle, pc;
  uint8_t reg, triggering_opcode;
  std::array<uint64_t, 2> accumulated;
  std::array<uint32_t, 2> normalized, carries;
};

class ExecutionResult;

// Vec<TraceRow> of the reference, resident in HBM as struct-of-arrays (zkir_trace_columns)
class ExecutionTrace {
 public:
  size_t len() const { return n_rows_; }
  bool is_empty() const { return n_rows_ == 0; }
  const zkir_trace_columns* device_columns() const { return r_ ? zkir_result_trace(r_) : nullptr; }
  // field ids of zkir_result_copy_column: 0 cycle, 1 pc, 2 instruction, 3 registers, 4 bound_bits, 5 bound_tag, 6 bound_payload, 7 reg_state
  template <typename T>
  std::vector<T> column(int field, int reg = 0) const {
    std::vector<T> out(n_rows_);
    if (n_rows_) { const int rc = zkir_result_copy_column(r_, field, reg, out.data()); if (rc != ZKIR_OK) detail::raise(rc); }
    return out;
  }
  std::vector<TraceRow> rows() const {                          // host copy of every row (tests; O(372 B/row) over PCIe)
    std::vector<TraceRow> out(n_rows_);
    const auto cyc = column<uint64_t>(0), pc = column<uint64_t>(1);
    const auto ins = column<uint32_t>(2);
    for (size_t i = 0; i < n_rows_; i++) { out[i].cycle = cyc[i]; out[i].pc = pc[i]; out[i].instruction = ins[i]; }
    for (int g = 0; g < 16; g++) {
      const auto v = column<uint64_t>(3, g), pay = column<uint64_t>(6, g);
      const auto bits = column<uint32_t>(4, g);
      const auto tag = column<uint8_t>(5, g), st = column<uint8_t>(7, g);
      for (size_t i = 0; i < n_rows_; i++) { out[i].registers[g] = v[i]; out[i].bounds[g] = {bits[i], tag[i], pay[i]}; out[i].register_states[g] = st[i]; }
    }
    return out;
  }

 private:
  friend class ExecutionResult;
  const zkir_result* r_ = nullptr;
  size_t n_rows_ = 0;
};

class ExecutionResult {                                         // vm.rs:54-78
 public:
  uint64_t cycles = 0;
  std::vector<uint64_t> outputs;
  HaltReason halt_reason{HaltReason::Ebreak, 0};
  ExecutionTrace execution_trace;

  size_t memory_op_count() const { return zkir_delta_log_n_mem_events(log()); }          // vm.rs:97-102

  // ExecutionResult::get_memory_trace (vm.rs:85-94): every data-memory op sorted by (timestamp, address, Read<Write); the sort and
  // the column expansion run on the device behind zkir_result_memory_trace, this copies the result to the host
  std::vector<MemoryOp> get_memory_trace() const { return memops(true, 0, (uint64_t)-1); }
  // TraceRow.memory_ops of one row (trace.rs:49).  `row` indexes the rows of THIS handle: for a row shard (zkir_exec_shard) it is
  // relative to the shard's first row (absolute cycle = row + zkir_delta_log_cycle_base); out of range throws std::out_of_range.
  std::vector<MemoryOp> row_memory_ops(uint64_t row) const { return memops(false, row, row + 1); }
  std::vector<RangeCheckWitness> range_check_witnesses() const {                        // vm.rs:66-69
    std::vector<RangeCheckWitness> out;
    if (!res_) return out;
    zkir_range_check_witness w;
    int rc = zkir_result_range_check_witnesses(res_, &w);
    if (rc != ZKIR_OK) detail::raise(rc);
    std::vector<uint64_t> v(w.n_checks), pc(w.n_checks);
    std::vector<uint16_t> ch(4 * w.chunk_stride);
    d2h(v.data(), w.value, 8 * w.n_checks); d2h(pc.data(), w.pc, 8 * w.n_checks); d2h(ch.data(), w.chunks, 2 * ch.size());
    for (uint64_t k = 0; k < w.n_witnesses; k++) {
      RangeCheckWitness wit;
      for (uint64_t i = w.witness_offsets[k]; i < w.witness_offsets[k + 1]; i++)
        wit.push_back({v[i], {ch[i], ch[w.chunk_stride + i], ch[2 * w.chunk_stride + i], ch[3 * w.chunk_stride + i]}, pc[i]});
      out.push_back(std::move(wit));
    }
    return out;
  }
  std::vector<NormalizationEvent> normalization_witnesses() const {                     // vm.rs:75-77
    std::vector<NormalizationEvent> out;
    if (!res_) return out;
    zkir_normalization_witness w;
    int rc = zkir_result_normalization_witnesses(res_, &w);
    if (rc != ZKIR_OK) detail::raise(rc);
    const size_t n = w.n_events;
    std::vector<uint64_t> cyc(n), pc(n), a0(n), a1(n);
    std::vector<uint32_t> n0(n), n1(n), c0(n), c1(n);
    std::vector<uint8_t> reg(n), op(n);
    d2h(cyc.data(), w.columns.cycle, 8 * n); d2h(pc.data(), w.columns.pc, 8 * n); d2h(a0.data(), w.columns.accumulated0, 8 * n); d2h(a1.data(), w.columns.accumulated1, 8 * n);
    d2h(n0.data(), w.columns.normalized0, 4 * n); d2h(n1.data(), w.columns.normalized1, 4 * n); d2h(c0.data(), w.columns.carry0, 4 * n); d2h(c1.data(), w.columns.carry1, 4 * n);
    d2h(reg.data(), w.columns.reg, n); d2h(op.data(), w.columns.opcode, n);
    for (size_t i = 0; i < n; i++) out.push_back({cyc[i], pc[i], reg[i], op[i], {a0[i], a1[i]}, {n0[i], n1[i]}, {c0[i], c1[i]}});
    return out;
  }
  size_t range_check_witness_count() const { return zkir_delta_log_n_rc_witnesses(log()); }
  size_t normalization_event_count() const { return zkir_delta_log_n_norm_events(log()); }
  const zkir_delta_log* delta_log() const { return log(); }     // host-side logs for the witness kernels (zkir_memops_*_launch, ...)

  ExecutionResult(ExecutionResult&& o) noexcept { *this = std::move(o); }
  ExecutionResult& operator=(ExecutionResult&& o) noexcept {
    if (this != &o) {
      release();
      cycles = o.cycles; outputs = std::move(o.outputs); halt_reason = o.halt_reason; execution_trace = o.execution_trace;
      res_ = o.res_; own_log_ = o.own_log_; o.res_ = nullptr; o.own_log_ = nullptr; o.execution_trace = ExecutionTrace();
    }
    return *this;
  }
  ExecutionResult(const ExecutionResult&) = delete;
  ExecutionResult& operator=(const ExecutionResult&) = delete;
  ~ExecutionResult() { release(); }

 private:
  friend class VM;
  ExecutionResult() = default;
  const zkir_delta_log* log() const { return res_ ? zkir_result_delta_log(res_) : own_log_; }
  static void d2h(void* dst, const void* src, size_t bytes) { const int rc = zkir_device_to_host(dst, src, bytes); if (rc != ZKIR_OK) detail::raise(rc); }
  std::vector<MemoryOp> memops(bool sorted, uint64_t row_lo, uint64_t row_hi) const {
    std::vector<MemoryOp> out;
    if (!res_) return out;
    zkir_memory_witness w;
    const int rc = zkir_result_memory_trace(res_, &w);
    if (rc != ZKIR_OK) detail::raise(rc);
    uint64_t lo = 0, hi = w.n_ops;
    if (!sorted && (row_lo >= w.n_rows || row_hi > w.n_rows)) throw std::out_of_range("row_memory_ops: row " + std::to_string(row_lo) + " is past the handle's " + std::to_string(w.n_rows) + " rows");
    if (!sorted) { uint64_t o[2]; d2h(&o[0], w.row_offsets + row_lo, 8); d2h(&o[1], w.row_offsets + row_hi, 8); lo = o[0]; hi = o[1]; }
    const zkir_memop_columns& c = sorted ? w.sorted : w.row_order;
    const size_t n = hi - lo;
    std::vector<uint64_t> ad(n), va(n), ts(n), pay(n);
    std::vector<uint32_t> bits(n);
    std::vector<uint8_t> wr(n), wd(n), tag(n);
    d2h(ad.data(), c.address + lo, 8 * n); d2h(va.data(), c.value + lo, 8 * n); d2h(ts.data(), c.timestamp + lo, 8 * n); d2h(pay.data(), c.bound_payload + lo, 8 * n);
    d2h(bits.data(), c.bound_bits + lo, 4 * n); d2h(wr.data(), c.is_write + lo, n); d2h(wd.data(), c.width + lo, n); d2h(tag.data(), c.bound_tag + lo, n);
    for (size_t i = 0; i < n; i++) out.push_back({ad[i], va[i], ts[i], wr[i], wd[i], {bits[i], tag[i], pay[i]}});
    return out;
  }
  void fill() {
    const zkir_delta_log* l = log();
    cycles = zkir_delta_log_cycles(l);
    const int kind = zkir_delta_log_halt_kind(l);
    halt_reason = {(HaltReason::Kind)kind, kind == ZKIR_HALT_EXIT ? zkir_delta_log_halt_code(l) : 0};
    const uint64_t* o = zkir_delta_log_outputs(l);
    outputs.assign(o, o + zkir_delta_log_n_outputs(l));
    execution_trace.r_ = res_;
    execution_trace.n_rows_ = res_ ? (size_t)zkir_delta_log_n_rows(l) : 0;
  }
  void release() {
    if (res_) zkir_result_free(res_);                           // also frees the delta log it owns
    else if (own_log_) zkir_delta_log_free(own_log_);
    res_ = nullptr; own_log_ = nullptr;
  }
  zkir_result* res_ = nullptr;
  zkir_delta_log* own_log_ = nullptr;
};

class VM {                                                      // vm.rs:138-358
 public:
  static VM new_(const zkir_spec::Program& program, std::vector<uint64_t> inputs, VMConfig config = {}) { return VM(program.to_bytes(), std::move(inputs), config); }
  VM(std::vector<uint8_t> program_blob, std::vector<uint64_t> inputs, VMConfig config = {})
      : blob_(std::move(program_blob)), inputs_(std::move(inputs)), config_(config) {}

  ExecutionResult run() && {                                    // consumes the VM, as in the reference
    const zkir_vm_config cfg{config_.max_cycles, (uint8_t)config_.trace, (uint8_t)config_.enable_range_checking,
                             (uint8_t)config_.enable_execution_trace, (uint8_t)config_.enable_deferred_model};
    ExecutionResult r;
    int rc;
    if (config_.enable_execution_trace) rc = zkir_exec(blob_.data(), blob_.size(), inputs_.data(), inputs_.size(), &cfg, &r.res_);
    else rc = zkir_interpret(blob_.data(), blob_.size(), inputs_.data(), inputs_.size(), &cfg, 0, &r.own_log_);   // nothing to materialise on the device
    if (rc != ZKIR_OK) detail::raise(rc);
    r.fill();
    return r;
  }

 private:
  std::vector<uint8_t> blob_;
  std::vector<uint64_t> inputs_;
  VMConfig config_;
};

inline std::vector<uint64_t> run(const zkir_spec::Program& program, std::vector<uint64_t> inputs) {      // lib.rs:59-62
  return VM::new_(program, std::move(inputs), VMConfig{}).run().outputs;
}

}  // namespace zkir_runtime

// ---- the self-defined prover stages (ABSENT from the reference, parity unpinned) ---------------------------------
namespace zkir_prover {

class StarkContext {                                            // device tables + workspace for traces of 2^log_n rows
 public:
  explicit StarkContext(uint32_t log_n) : log_n_(log_n) { const int rc = zkir_stark_ctx_create(log_n, 1, &h_); if (rc != ZKIR_OK) zkir_runtime::detail::raise(rc); }
  ~StarkContext() { zkir_stark_ctx_free(h_); }
  StarkContext(const StarkContext&) = delete;
  StarkContext& operator=(const StarkContext&) = delete;
  const zkir_stark_ctx* handle() const { return h_; }
  uint32_t log_n() const { return log_n_; }

 private:
  zkir_stark_ctx* h_ = nullptr;
  uint32_t log_n_;
};

// Public inputs of a proof of `result` (row count, mode, entry pc, program digest, io digest).  The C struct borrows the program blob (the
// prover reads the instruction ROM of the lookup argument from it; the proof carries it): this wrapper OWNS the bytes and keeps the
// struct's pointer on them through copies and moves.
struct PublicInputs : zkir_public_inputs {
  PublicInputs() : zkir_public_inputs{} {}
  PublicInputs(const zkir_public_inputs& c, std::vector<uint8_t> blob) : zkir_public_inputs(c), blob_(std::move(blob)) {
    if (c.inputs && c.n_inputs) inputs_.assign(c.inputs, c.inputs + c.n_inputs);           // (modes 2 / 3: the tapes travel in the clear — owned here too)
    if (c.outputs && c.n_outputs) outputs_.assign(c.outputs, c.outputs + c.n_outputs);
    repoint();
  }
  PublicInputs(const PublicInputs& o) : zkir_public_inputs(o), blob_(o.blob_), inputs_(o.inputs_), outputs_(o.outputs_) { repoint(); }
  PublicInputs(PublicInputs&& o) noexcept : zkir_public_inputs(o), blob_(std::move(o.blob_)), inputs_(std::move(o.inputs_)), outputs_(std::move(o.outputs_)) { repoint(); }
  PublicInputs& operator=(PublicInputs o) {
    static_cast<zkir_public_inputs&>(*this) = o; blob_ = std::move(o.blob_); inputs_ = std::move(o.inputs_); outputs_ = std::move(o.outputs_); repoint(); return *this;
  }
  const std::vector<uint8_t>& program_bytes() const { return blob_; }

 private:
  void repoint() {
    program_blob = blob_.data(); program_blob_len = blob_.size();
    inputs = inputs_.empty() ? nullptr : inputs_.data(); n_inputs = inputs_.size();
    outputs = outputs_.empty() ? nullptr : outputs_.data(); n_outputs = outputs_.size();
    mem_old = nullptr; mem_told = nullptr; cell_addr = nullptr; cell_bytes = nullptr; cell_time = nullptr; n_cells = 0;   // mode 3: zkir_prove makes the memory witness on the device
    hash_section = nullptr; hash_section_words = 0;                                                                     // mode 4: a run with hash syscalls brings its tape (zkir_memcheck_witness_of_mode)
  }
  std::vector<uint8_t> blob_;
  std::vector<uint64_t> inputs_, outputs_;
};
// The proof's MODE: what an accepted proof says beyond the trace's control flow and the 20 opcodes of the default AIR
enum class ProofMode : uint32_t {
  Default = 0,    // the default VM mode
  Deferred = 1,   // VMConfig::enable_deferred_model (relaxed AIR)
  Io = 2,         // default + the I/O argument: what the run read and wrote, and that it ended on the instruction its halt reason names
  Memory = 3,     // Io + the memory argument, the bitwise opcodes, the shifts and MUL: 43 of 50 opcodes constrained, memory consistent (whole runs; no hash syscalls)
  Wide = 4,       // Memory + MULH / DIVU / REMU / DIV / REM (a chunk relation below 2^40, the verifier-recomputed wide tape on raw 64-bit operands), the code segment's boundary cell, and — given the host witness with its hash tape
                  // (zkir_memcheck_witness_of_mode(.., 4, ..) + zkir_public_inputs_set_memory) — hash syscalls, whose digests the verifier computes (round 6)
};
inline PublicInputs public_inputs(const zkir_runtime::ExecutionResult& result, const zkir_spec::Program& program, const std::vector<uint64_t>& inputs,
                                  const zkir_runtime::VMConfig& config = {}) {
  zkir_public_inputs pub;
  std::vector<uint8_t> blob = program.to_bytes();
  const int rc = zkir_public_inputs_of(result.delta_log(), blob.data(), blob.size(), inputs.data(), inputs.size(), config.enable_deferred_model ? 1u : 0u, &pub);
  if (rc != ZKIR_OK) zkir_runtime::detail::raise(rc);
  return PublicInputs(pub, std::move(blob));
}
inline PublicInputs public_inputs(const zkir_runtime::ExecutionResult& result, const zkir_spec::Program& program, const std::vector<uint64_t>& inputs, ProofMode mode) {
  zkir_public_inputs pub;
  std::vector<uint8_t> blob = program.to_bytes();
  const int rc = zkir_public_inputs_of(result.delta_log(), blob.data(), blob.size(), inputs.data(), inputs.size(), (uint32_t)mode, &pub);
  if (rc != ZKIR_OK) zkir_runtime::detail::raise(rc);
  return PublicInputs(pub, std::move(blob));
}

// Full proof (u32 little-endian words, format zkir_proof_version()) of a run whose execution trace is resident in HBM; the context must be built for
// zkir_padded_log_n(rows).  zkir_prover::verify is the host-side check (0 = accepted).
inline std::vector<uint32_t> prove(const StarkContext& ctx, const zkir_runtime::ExecutionResult& result, const zkir_public_inputs& pub, void* hip_stream = nullptr) {
  uint32_t* words = nullptr;
  uint64_t n = 0;
  const int rc = zkir_prove(ctx.handle(), result.execution_trace.device_columns(), &pub, &words, &n, nullptr, hip_stream);
  if (rc != ZKIR_OK) zkir_runtime::detail::raise(rc);
  std::vector<uint32_t> out(words, words + n);
  zkir_proof_free(words);
  return out;
}
inline int verify(const std::vector<uint32_t>& proof, const zkir_public_inputs* expect = nullptr) { return zkir_verify(proof.data(), proof.size(), expect); }
// verify + the run's CLAIM in the clear (zkir_verify_io): the I/O tapes and the halt reason must hash to the proof's io digest and the halt row must be the
// instruction the halt reason names (50 / 52 / 53, include/zkir_amd.h); what a caller who holds an ExecutionResult-shaped claim calls
inline int verify_io(const std::vector<uint32_t>& proof, const std::vector<uint64_t>& inputs, const std::vector<uint64_t>& outputs, const zkir_runtime::HaltReason& halt,
                     const zkir_public_inputs* expect = nullptr) {
  const int kind = (int)halt.kind;                                  // HaltReason::Kind carries the ZKIR_HALT_* values
  return zkir_verify_io(proof.data(), proof.size(), expect, inputs.data(), inputs.size(), outputs.data(), outputs.size(), kind, halt.code);
}
inline int verify_chain_io(const std::vector<std::vector<uint32_t>>& proofs, const std::vector<uint64_t>& inputs, const std::vector<uint64_t>& outputs,
                           const zkir_runtime::HaltReason& halt, const zkir_public_inputs* expect = nullptr) {
  std::vector<const uint32_t*> ptrs; std::vector<uint64_t> lens;
  for (const auto& p : proofs) { ptrs.push_back(p.data()); lens.push_back(p.size()); }
  const int kind = (int)halt.kind;
  return zkir_verify_chain_io(ptrs.data(), lens.data(), (uint32_t)proofs.size(), expect, inputs.data(), inputs.size(), outputs.data(), outputs.size(), kind, halt.code);
}
// A run proven in segments (one row shard per GPU, consecutive segments sharing one row): every segment on its own, and the chain as one run.
struct BoundaryStates { uint32_t first[68], last[68]; };
inline int verify_segment(const std::vector<uint32_t>& proof, BoundaryStates* states = nullptr, const zkir_public_inputs* expect = nullptr) {
  return zkir_verify_segment(proof.data(), proof.size(), expect, states ? states->first : nullptr, states ? states->last : nullptr);
}
inline int verify_chain(const std::vector<std::vector<uint32_t>>& proofs, const zkir_public_inputs* expect = nullptr) {
  std::vector<const uint32_t*> ptrs; std::vector<uint64_t> lens;
  for (const auto& p : proofs) { ptrs.push_back(p.data()); lens.push_back(p.size()); }
  return zkir_verify_chain(ptrs.data(), lens.data(), (uint32_t)proofs.size(), expect);
}

}  // namespace zkir_prover

