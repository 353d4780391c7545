// reference_tests.cpp — a handful of the reference's own tests re-stated against the C++ host mirror (include/zkir_amd.hpp):
// same programs, same assertions, the reference file:line next to each.  Expected values are the literals those tests assert
// (also held as data in tests/golden/reference_kats.json).
//
//   ./reference_tests host   — tests that run with VMConfig defaults (no execution trace: host interpreter only, no GPU needed)
//   ./reference_tests gpu    — the tests that enable the execution trace (zkir_exec: trace materialised in HBM) as well
//
// Built and run by tests/test_cpp_mirror.py (g++ -std=c++17 -Iinclude tests/cpp/reference_tests.cpp -Lzkir_amd -lzkir_amd).
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "zkir_amd.hpp"

using namespace zkir_spec;
using zkir_runtime::ExecutionResult;
using zkir_runtime::HaltReason;
using zkir_runtime::RuntimeError;
using zkir_runtime::VM;
using zkir_runtime::VMConfig;

static int failures = 0;
#define CHECK(cond)                                                                           \
  do { if (!(cond)) { std::printf("    FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

static std::vector<uint32_t> cat(std::initializer_list<std::vector<uint32_t>> parts) {
  std::vector<uint32_t> out;
  for (auto& p : parts) out.insert(out.end(), p.begin(), p.end());
  return out;
}
static std::vector<uint32_t> write_reg(int r) { return {addi(11, r, 0), addi(10, 0, 2), ecall()}; }       // SYS_WRITE(R11 = r)
static const std::vector<uint32_t> EXIT0 = {addi(10, 0, 0), addi(11, 0, 0), ecall()};                      // SYS_EXIT(0)

// ---- tests/cross_module.rs ----------------------------------------------------------------------------------------------
static void test_fibonacci() {                                   // cross_module.rs:140-175: fib(5) = 5, exit code 0
  auto code = cat({{addi(1, 0, 0), addi(2, 0, 1), addi(3, 0, 4), add(4, 1, 2), addi(1, 2, 0), addi(2, 4, 0), addi(3, 3, -1), bne(3, 0, -16)}, write_reg(2), EXIT0});
  CHECK(code[3] == 0x00010A00u && code[7] == 0xFFF801C1u);       // the encodings the reference's assembler produces
  ExecutionResult r = VM::new_(Program::from_code(code), {}).run();
  CHECK(r.outputs == std::vector<uint64_t>{5});
  CHECK(r.halt_reason == HaltReason::exit(0));
}
static void test_sum_loop() {                                    // cross_module.rs:406-437: 1 + 2 + .. + 5 = 15
  auto code = cat({{addi(1, 0, 0), addi(2, 0, 1), addi(3, 0, 6), add(1, 1, 2), addi(2, 2, 1), bne(2, 3, -8)}, write_reg(1), EXIT0});
  CHECK(zkir_runtime::run(Program::from_code(code), {}) == std::vector<uint64_t>{15});
}
static void test_arithmetic_chain() {                            // cross_module.rs:60-87: 10 + 20 + 30 = 60
  auto code = cat({{addi(1, 0, 10), addi(2, 0, 20), addi(3, 0, 30), add(4, 1, 2), add(4, 4, 3)}, write_reg(4), EXIT0});
  CHECK(zkir_runtime::run(Program::from_code(code), {}) == std::vector<uint64_t>{60});
}
static void test_branch_taken_skips() {                          // cross_module.rs:370-403: r3 stays 0
  auto code = cat({{addi(1, 0, 10), addi(2, 0, 10), beq(1, 2, 8), addi(3, 0, 1), addi(4, 0, 2)}, write_reg(3), EXIT0});
  CHECK(zkir_runtime::run(Program::from_code(code), {}) == std::vector<uint64_t>{0});
}
static void test_store_load_roundtrip() {                        // cross_module.rs:335-363
  auto code = cat({{addi(1, 0, 42), addi(2, 0, 0x1000), sw(2, 1, 0), lw(3, 2, 0)}, write_reg(3), EXIT0});
  CHECK(zkir_runtime::run(Program::from_code(code), {}) == std::vector<uint64_t>{42});
}
static void test_echo_input() {                                  // cross_module.rs:32-57; vm.rs:489-533
  auto code = cat({{addi(10, 0, 1), ecall(), addi(11, 10, 0), addi(10, 0, 2), ecall()}, EXIT0});
  ExecutionResult r = VM::new_(Program::from_code(code), {123}).run();
  CHECK(r.outputs == std::vector<uint64_t>{123} && r.halt_reason == HaltReason::exit(0));
}
// ---- zkir-runtime/src/vm.rs unit tests ----------------------------------------------------------------------------------
static void test_exit_code() {                                   // vm.rs:464-486: Exit(42) after 3 cycles
  ExecutionResult r = VM::new_(Program::from_code({addi(10, 0, 0), addi(11, 0, 42), ecall()}), {}).run();
  CHECK(r.halt_reason == HaltReason::exit(42) && r.cycles == 3);
}
static void test_basic_ebreak() {                                // vm.rs:434-461
  ExecutionResult r = VM::new_(Program::from_code({addi(1, 0, 10), addi(2, 0, 20), add(3, 1, 2), ebreak()}), {}).run();
  CHECK(r.halt_reason == HaltReason::ebreak() && r.cycles == 4);
}
static void test_cycle_limit() {                                 // vm.rs:536-552: jal r0, 0 forever, max_cycles = 100
  VMConfig cfg; cfg.max_cycles = 100;
  ExecutionResult r = VM::new_(Program::from_code({jal(0, 0)}), {}, cfg).run();
  CHECK(r.halt_reason == HaltReason::cycle_limit() && r.cycles == 100);
}
static void test_trace_disabled_by_default() {                   // vm.rs:866-904
  ExecutionResult r = VM::new_(Program::from_code({addi(1, 0, 0x42), addi(3, 0, 0x1000), sw(3, 1, 0), ebreak()}), {}).run();
  CHECK(r.execution_trace.is_empty() && r.memory_op_count() == 0);
}
// ---- tests/stress_tests.rs ----------------------------------------------------------------------------------------------
static void test_thousand_instructions() {                       // stress_tests.rs:25-56: 1003 cycles
  std::vector<uint32_t> code(1000, add(1, 1, 0));
  code.insert(code.end(), EXIT0.begin(), EXIT0.end());
  ExecutionResult r = VM::new_(Program::from_code(code), {}).run();
  CHECK(r.halt_reason == HaltReason::exit(0) && r.cycles == 1003);
}
static void test_divu_by_one() {                                 // stress_tests.rs:437-460
  auto code = cat({{addi(1, 0, 12345), addi(2, 0, 1), divu(3, 1, 2)}, write_reg(3), EXIT0});
  CHECK(zkir_runtime::run(Program::from_code(code), {}) == std::vector<uint64_t>{12345});
}
// ---- zkir-spec/src/field.rs: Mersenne31 (field.rs:231-321) -------------------------------------------------------------------
static void test_mersenne31_field() {
  using M = zkir_spec::Mersenne31;
  const uint32_t p = 2147483647u;
  CHECK(M::PRIME == p && M::zero().value() == 0 && M::one().value() == 1);                                   // test_constants
  CHECK(M(0).value() == 0 && M(1).value() == 1 && M(p).value() == 0 && M(p + 1).value() == 1 && M(2 * p).value() == 0);   // test_reduce
  CHECK((M(100) + M(200)).value() == 300 && (M(p - 1) + M(5)).value() == 4);                                 // test_addition
  CHECK((M(200) - M(100)).value() == 100 && (M(5) - M(10)).value() == p - 5);                                // test_subtraction
  CHECK((M(100) * M(200)).value() == 20000 && (M(1u << 20) * M(1u << 20)).value() < p);                     // test_multiplication
  CHECK((M(1u << 20) * M(1u << 20)).value() == 512);                                                         // 2^40 mod (2^31 - 1) = 2^9
  CHECK((M(100) + -M(100)).value() == 0 && (-M::zero()).value() == 0 && (-M::one()).value() == p - 1);       // test_negation
  CHECK((M(7) * M(7).inv()).value() == 1 && (M(12345) * M(12345).inv()).value() == 1);                       // test_inversion
  bool threw = false;
  try { M::zero().inv(); } catch (const std::domain_error& e) { threw = std::strstr(e.what(), "Division by zero") != nullptr; }   // test_inverse_zero
  CHECK(threw);
  CHECK(M(2).pow(0).value() == 1 && M(2).pow(1).value() == 2 && M(2).pow(10).value() == 1024 && M(123).pow(p - 1).value() == 1);   // test_pow
  M acc(5); acc += M(7); acc *= M(3); acc -= M(40);                                                          // assign forms (field.rs:138-176)
  CHECK(acc.value() == p - 4);
}
// ---- error behaviour ----------------------------------------------------------------------------------------------------
template <typename F>
static bool throws(RuntimeError::Kind kind, const char* needle, F f) {
  try { f(); } catch (const RuntimeError& e) { return e.kind == kind && std::strstr(e.what(), needle) != nullptr; }
  return false;
}
static void test_division_by_zero() {                            // execute.rs:849-867: RuntimeError::DivisionByZero { pc }
  CHECK(throws(RuntimeError::DivisionByZero, "Division by zero", [] { VM::new_(Program::from_code({addi(1, 0, 100), div_(3, 1, 2)}), {}).run(); }));
}
static void test_invalid_syscall() {                             // syscall.rs:262-277: InvalidSyscall { syscall: 999 }
  CHECK(throws(RuntimeError::InvalidSyscall, "999", [] { VM::new_(Program::from_code({addi(10, 0, 999), ecall()}), {}).run(); }));
}
static void test_poseidon2_syscall_is_an_error() {               // syscall_integration.rs:401-422; crypto.rs:462-466
  CHECK(throws(RuntimeError::Other, "", [] { VM::new_(Program::from_code({addi(10, 0, 4), ecall()}), {}).run(); }));
}
static void test_run_consumes_the_vm() {                         // vm.rs:208 takes self: in C++ run() is rvalue-qualified
  VM vm = VM::new_(Program::from_code({ebreak()}), {});
  ExecutionResult r = std::move(vm).run();
  CHECK(r.cycles == 1);
}
// ---- execution trace (GPU) ----------------------------------------------------------------------------------------------
static void test_execution_trace_rows() {                        // vm.rs:906-964; cross_module.rs:444-468: 4 rows, cycle = index, pre-state
  VMConfig cfg; cfg.enable_execution_trace = true;
  ExecutionResult r = VM::new_(Program::from_code({addi(1, 0, 100), addi(2, 0, 200), add(3, 1, 2), ebreak()}), {}, cfg).run();
  CHECK(r.execution_trace.len() == 4);
  auto rows = r.execution_trace.rows();
  for (size_t i = 0; i < rows.size(); i++) CHECK(rows[i].cycle == i && rows[i].pc == 0x1000 + 4 * i);
  CHECK(rows[0].registers[1] == 0 && rows[1].registers[1] == 100 && rows[2].registers[2] == 200 && rows[3].registers[3] == 300);   // state BEFORE each instruction
  CHECK(rows[3].instruction == ebreak());
  CHECK(r.halt_reason == HaltReason::ebreak());
}
static void test_trace_with_memory_ops() {                       // vm.rs:996-1070: 5 rows, one store and one load recorded
  VMConfig cfg; cfg.enable_execution_trace = true;
  ExecutionResult r = VM::new_(Program::from_code({addi(1, 0, 0x42), addi(3, 0, 0x1000), sw(3, 1, 0), lw(4, 3, 0), ebreak()}), {}, cfg).run();
  CHECK(r.execution_trace.len() == 5 && r.memory_op_count() == 2);
  auto rows = r.execution_trace.rows();
  CHECK(rows[4].registers[4] == 0x42);
}
static void test_cycle_limit_trace() {                           // vm.rs:211-214: CycleLimit is a normal halt with exactly max_cycles rows
  VMConfig cfg; cfg.enable_execution_trace = true; cfg.max_cycles = 1000;
  ExecutionResult r = VM::new_(Program::from_code({addi(1, 1, 1), jal(0, -4)}), {}, cfg).run();
  CHECK(r.execution_trace.len() == 1000 && r.halt_reason == HaltReason::cycle_limit());
  auto r1 = r.execution_trace.column<uint64_t>(3, 1);             // registers[1] column straight from HBM
  CHECK(r1[0] == 0 && r1[2] == 1 && r1[999] == 500);
}
static void test_trace_timestamp_synchronization() {             // vm.rs:1073-1200: 2 SW + 2 LW; row memory_ops, timestamps = cycles, sorted trace
  VMConfig cfg; cfg.enable_execution_trace = true;
  ExecutionResult r = VM::new_(Program::from_code({addi(1, 0, 0x100), addi(2, 0, 0x1000), sw(2, 1, 0), addi(3, 0, 0x200), sw(2, 3, 4), lw(4, 2, 0), lw(5, 2, 4), ebreak()}), {}, cfg).run();
  CHECK(r.halt_reason == HaltReason::ebreak());
  const auto all = r.get_memory_trace();
  CHECK(all.size() == 4);
  size_t writes = 0, reads = 0;
  for (const auto& op : all) { writes += op.is_write(); reads += op.is_read(); }
  CHECK(writes == 2 && reads == 2);
  const struct { uint64_t row; bool write; } expect[] = {{2, true}, {4, true}, {5, false}, {6, false}};
  for (const auto& e : expect) {
    const auto ops = r.row_memory_ops(e.row);
    CHECK(ops.size() == 1 && ops[0].is_write() == e.write && ops[0].timestamp == e.row);
  }
  for (uint64_t row : {0, 1, 3, 7}) CHECK(r.row_memory_ops(row).empty());
  for (size_t i = 1; i < all.size(); i++) CHECK(all[i - 1].timestamp <= all[i].timestamp);
  CHECK(all[2].value == 0x100 && all[3].value == 0x200 && all[0].address == 0x1000 && all[1].address == 0x1004);
  CHECK(all[0].bound.max_bits == 32 && all[0].bound.source_tag == ZKIR_BOUND_TYPE_WIDTH);          // memory.rs:245
}
static void test_bound_propagation_and_deferred_checks() {       // vm.rs:698-752: 30 doublings then SW -> range-check witnesses
  std::vector<uint32_t> code = {addi(1, 0, (1 << 15) - 1)};
  for (int i = 0; i < 30; i++) code.push_back(add(1, 1, 1));
  code.push_back(addi(2, 0, 0x1000)); code.push_back(sw(2, 1, 0)); code.push_back(ebreak());
  VMConfig cfg; cfg.enable_range_checking = true; cfg.enable_execution_trace = true;
  ExecutionResult r = VM::new_(Program::from_code(code), {}, cfg).run();
  CHECK(r.halt_reason == HaltReason::ebreak());
  const auto w = r.range_check_witnesses();
  CHECK(w.size() > 0 && w.size() == r.range_check_witness_count());
  for (const auto& wit : w)
    for (const auto& c : wit) {                                  // range_check.rs:175-192: 10-bit chunks of the two 20-bit limbs
      const uint32_t l0 = c.value & 0xFFFFF, l1 = (c.value >> 20) & 0xFFFFF;
      CHECK(c.chunks[0] == (l0 & 1023) && c.chunks[1] == (l0 >> 10) && c.chunks[2] == (l1 & 1023) && c.chunks[3] == (l1 >> 10));
    }
}
static void test_deferred_carry_normalization_event() {          // deferred_integration_test.rs:270-316 through the VM: (2^20 - 10) + 100 -> [90, 1], carry 1
  // r1 = 2^20 - 10 (slli + addi keep it Normalized), r2 = 100, r3 = r1 + r2 (deferred: Accumulated), then an observation point on r3
  VMConfig cfg; cfg.enable_deferred_model = true; cfg.enable_execution_trace = true;
  ExecutionResult r = VM::new_(Program::from_code({addi(1, 0, 1), slli(1, 1, 20), addi(1, 1, -10), addi(2, 0, 100), add(3, 1, 2), beq(3, 0, 8), ebreak(), ebreak()}), {}, cfg).run();
  const auto ev = r.normalization_witnesses();
  CHECK(ev.size() == r.normalization_event_count() && !ev.empty());
  const zkir_runtime::NormalizationEvent& e = ev.back();                       // the BEQ's rs1 = r3 (execute.rs:903-916: rs1 only)
  CHECK(e.reg == 3 && e.triggering_opcode == 0x40 && e.normalized[0] == 90 && e.normalized[1] == 1 && e.carries[0] == 1);
}

// ---- prover stages (no counterpart in the reference; the words are compared with the oracle's prover in tests/test_gpu_stark.py) --
static void test_prove_and_verify() {
  // a run that halts on its own (Exit after 3 + 5*11 + 6 rows): padded to 2^7, bound to its program / outputs / halt reason
  auto code = cat({{addi(1, 0, 0), addi(2, 0, 1), addi(3, 0, 11), add(4, 1, 2), addi(1, 2, 0), addi(2, 4, 0), addi(3, 3, -1), bne(3, 0, -16)}, write_reg(2), EXIT0});
  const Program prog = Program::from_code(code);
  VMConfig cfg; cfg.enable_execution_trace = true;
  ExecutionResult r = VM::new_(prog, {}, cfg).run();
  CHECK(r.outputs == std::vector<uint64_t>{144} && r.cycles == 64);
  const zkir_prover::PublicInputs pub = zkir_prover::public_inputs(r, prog, {}, cfg);     // owns the program bytes the prover reads the ROM from
  CHECK(pub.n_real == 64 && pub.entry_point == 0x1000 && pub.deferred == 0);
  zkir_prover::StarkContext ctx(zkir_padded_log_n(pub.n_real));
  const std::vector<uint32_t> proof = zkir_prover::prove(ctx, r, pub);
  CHECK(proof.size() > 1000 && proof[0] == 0x46504B5Au && proof[1] == zkir_proof_version() && proof[2] == 6 && proof[3] == zkir_main_trace_width() && proof[4] == zkir_proof_num_queries());
  CHECK(zkir_prover::prove(ctx, r, pub) == proof);               // deterministic transcript
  CHECK(zkir_prover::verify(proof) == 0 && zkir_prover::verify(proof, &pub) == 0);
  // the claim in the clear: outputs and halt reason hash to the io digest, the halt row is the exit ECALL with R11 = 0 (zkir_verify_io)
  CHECK(zkir_prover::verify_io(proof, {}, r.outputs, r.halt_reason, &pub) == 0);
  CHECK(zkir_prover::verify_io(proof, {}, {145}, r.halt_reason) == 50 && zkir_prover::verify_io(proof, {}, r.outputs, HaltReason::exit(1)) == 50);
  zkir_public_inputs other = pub; other.io_digest[0] ^= 1;       // someone claims other outputs
  CHECK(zkir_prover::verify(proof, &other) == 6);
  std::vector<uint32_t> bad = proof; bad[bad.size() / 2] = (bad[bad.size() / 2] + 1) % 2013265921u;
  CHECK(zkir_prover::verify(bad) != 0);
}

// Modes 2 and 3 through the C ABI: the I/O argument, + the memory argument and the bitwise opcodes; the memory witness is made on the device by zkir_prove
static void test_prove_modes() {
  // store i * 3 into an array, read it back through LW / LBU, AND / XOR it, WRITE the sum, exit 0
  auto code = cat({{addi(6, 0, 0x4000), addi(1, 0, 0), addi(3, 0, 9), addi(4, 0, 0),
                    add(2, 1, 1), add(2, 2, 1), sw(6, 2, 0), lw(7, 6, 0), enc_i(Opcode::LBU, 8, 6, 0),
                    enc_i(Opcode::ANDI, 7, 7, 0xFF), enc_r(Opcode::XOR, 9, 7, 8), add(4, 4, 7), add(4, 4, 9),
                    addi(6, 6, 4), addi(1, 1, 1), addi(3, 3, -1), bne(3, 0, -48)}, write_reg(4), EXIT0});
  const Program prog = Program::from_code(code);
  VMConfig cfg; cfg.enable_execution_trace = true;
  ExecutionResult r = VM::new_(prog, {}, cfg).run();
  CHECK(r.outputs.size() == 1 && r.outputs[0] == 3 * 36);                                 // sum of 3 i, i = 0..8 (the XOR of equal bytes adds 0)
  zkir_prover::StarkContext ctx(zkir_padded_log_n(r.cycles));
  for (zkir_prover::ProofMode mode : {zkir_prover::ProofMode::Io, zkir_prover::ProofMode::Memory}) {
    const zkir_prover::PublicInputs pub = zkir_prover::public_inputs(r, prog, {}, mode);
    CHECK(pub.deferred == (uint32_t)mode && pub.n_outputs == 1);
    const std::vector<uint32_t> proof = zkir_prover::prove(ctx, r, pub);
    CHECK(proof[9] == (uint32_t)mode && proof[3] == zkir_main_trace_width_for((uint32_t)mode));   // 160 / 264 committed columns
    CHECK(zkir_prover::verify(proof) == 0 && zkir_prover::verify(proof, &pub) == 0);
    std::vector<uint32_t> bad = proof; bad[bad.size() / 3] = (bad[bad.size() / 3] + 1) % 2013265921u;
    CHECK(zkir_prover::verify(bad) != 0);
    // a claim of another output: the prover makes a proof of it, the verifiers reject it (the table side of the tape lookup comes from the claim)
    zkir_prover::PublicInputs forged = pub;
    zkir_public_inputs raw = forged;
    std::vector<uint64_t> outs{r.outputs[0] + 1};
    raw.outputs = outs.data();
    const zkir_prover::PublicInputs claim(raw, prog.to_bytes());
    const std::vector<uint32_t> fproof = zkir_prover::prove(ctx, r, claim);
    CHECK(zkir_prover::verify(fproof) != 0);
  }
}

// SURVEY 8(b)'s shape of the seam: zkir_exec -> zkir_result, zkir_prove_result(result, params) -> proof bytes; the prover parameters are part of the statement.
static void test_prove_result_and_parameters() {
  auto code = cat({{addi(1, 0, 0), addi(2, 0, 1), addi(3, 0, 11), add(4, 1, 2), addi(1, 2, 0), addi(2, 4, 0), addi(3, 3, -1), bne(3, 0, -16)}, write_reg(2), EXIT0});
  const std::vector<uint8_t> blob = Program::from_code(code).to_bytes();
  zkir_vm_config cfg{}; cfg.max_cycles = 1000000; cfg.enable_execution_trace = 1;
  zkir_result* res = nullptr;
  CHECK(zkir_exec(blob.data(), blob.size(), nullptr, 0, &cfg, &res) == ZKIR_OK);
  uint8_t* bytes = nullptr; size_t len = 0;
  CHECK(zkir_prove_result(res, nullptr, &bytes, &len) == ZKIR_OK && len % 4 == 0 && len > 4000);          // defaults: mode 0, 50 queries, 12 bits
  const uint32_t* w = reinterpret_cast<const uint32_t*>(bytes);
  CHECK(w[0] == 0x46504B5Au && w[1] == zkir_proof_version() && w[4] == 50 && w[6] == 12 && w[9] == 0);
  CHECK(zkir_verify(w, len / 4, nullptr) == 0);
  const size_t len_default = len;
  zkir_proof_bytes_free(bytes);
  const zkir_prover_params big{2, 84, 16};                                                                   // mode 2 (the I/O argument), 84 queries + 16 grinding bits
  CHECK(zkir_prove_result(res, &big, &bytes, &len) == ZKIR_OK && len > len_default);
  w = reinterpret_cast<const uint32_t*>(bytes);
  CHECK(w[1] == zkir_proof_version_of_mode(2) && w[4] == 84 && w[6] == 16 && w[9] == 2);
  CHECK(zkir_verify(w, len / 4, nullptr) == 0);
  zkir_public_inputs expect;
  CHECK(zkir_public_inputs_of(zkir_result_delta_log(res), blob.data(), blob.size(), nullptr, 0, 2, &expect) == ZKIR_OK);
  CHECK(zkir_verify(w, len / 4, &expect) == 2);                                                              // a verifier expecting the defaults refuses other parameters
  CHECK(zkir_public_inputs_set_params(&expect, &big) == ZKIR_OK && zkir_verify(w, len / 4, &expect) == 0);
  const uint64_t out144 = 144;
  CHECK(zkir_verify_io(w, len / 4, &expect, nullptr, 0, &out144, 1, ZKIR_HALT_EXIT, 0) == 0);                 // the mode-2 proof says what the run wrote
  zkir_proof_bytes_free(bytes);
  const zkir_prover_params weak{0, 49, 12}, wrong_mode{1, 0, 0};
  zkir_public_inputs p0;
  CHECK(zkir_public_inputs_of(zkir_result_delta_log(res), blob.data(), blob.size(), nullptr, 0, 0, &p0) == ZKIR_OK);
  CHECK(zkir_public_inputs_set_params(&p0, &weak) == ZKIR_ERR_ARGUMENT && zkir_public_inputs_set_params(&p0, &wrong_mode) == ZKIR_ERR_ARGUMENT);
  CHECK(zkir_prove_result(res, &weak, &bytes, &len) == ZKIR_ERR_ARGUMENT && bytes == nullptr);
  CHECK(zkir_abi_version() == ZKIR_AMD_ABI_VERSION);
  zkir_result_free(res);
}

// A run proven in segments through the plain C ABI, as a multi-GPU caller without any HIP code of its own would: one interpretation,
// zkir_exec_shard per row range (ranges share one row), zkir_prove per shard, zkir_verify_chain over the proofs.
static void test_segment_proofs_through_the_c_abi() {
  auto code = cat({{addi(1, 0, 0), addi(2, 0, 1), addi(3, 0, 150), add(4, 1, 2), addi(1, 2, 0), addi(2, 4, 0), addi(3, 3, -1), bne(3, 0, -16)}, write_reg(2), EXIT0});
  const Program prog = Program::from_code(code);
  const std::vector<uint8_t> blob = prog.to_bytes();
  zkir_vm_config cfg{}; cfg.max_cycles = 1000000; cfg.enable_execution_trace = 1;
  zkir_delta_log* log = nullptr;
  CHECK(zkir_interpret(blob.data(), blob.size(), nullptr, 0, &cfg, 0, &log) == ZKIR_OK);
  const uint64_t n = zkir_delta_log_n_rows(log);                 // 3 + 5 * 150 + 6 = 759 rows
  CHECK(n == 759);
  zkir_public_inputs run_pub;
  CHECK(zkir_public_inputs_of(log, blob.data(), blob.size(), nullptr, 0, 0, &run_pub) == ZKIR_OK);
  const uint64_t S = 256;
  std::vector<std::vector<uint32_t>> proofs;
  for (uint64_t a = 0; a + 1 < n; a += S - 1) {
    const uint64_t b = a + S < n ? a + S : n;
    zkir_result* r = nullptr;
    CHECK(zkir_exec_shard(log, a, b, &r) == ZKIR_OK);
    CHECK(zkir_delta_log_cycle_base(zkir_result_delta_log(r)) == a && zkir_delta_log_n_rows(zkir_result_delta_log(r)) == b - a);
    zkir_public_inputs pub = run_pub; pub.n_real = b - a;
    zkir_stark_ctx* ctx = nullptr;
    CHECK(zkir_stark_ctx_create(zkir_padded_log_n(b - a), 1, &ctx) == ZKIR_OK);
    uint32_t* words = nullptr; uint64_t nw = 0;
    CHECK(zkir_prove(ctx, zkir_result_trace(r), &pub, &words, &nw, nullptr, nullptr) == ZKIR_OK);
    proofs.emplace_back(words, words + nw);
    zkir_proof_free(words); zkir_stark_ctx_free(ctx); zkir_result_free(r);
  }
  CHECK(proofs.size() == 3);
  CHECK(zkir_prover::verify_chain(proofs, &run_pub) == 0 && zkir_prover::verify_chain(proofs) == 0);
  zkir_prover::BoundaryStates st0, st1;
  CHECK(zkir_prover::verify_segment(proofs[0], &st0) == 0 && zkir_prover::verify_segment(proofs[1], &st1) == 0);
  CHECK(st0.first[0] == 0 && st0.last[0] == S - 1 && st1.first[0] == S - 1 && std::memcmp(st0.last, st1.first, sizeof st0.last) == 0);
  CHECK(zkir_prover::verify(proofs[0]) == 0 && zkir_prover::verify(proofs[1]) == 7);          // only the first starts in the initial state
  CHECK(zkir_prover::verify_chain_io(proofs, {}, std::vector<uint64_t>(zkir_delta_log_outputs(log), zkir_delta_log_outputs(log) + zkir_delta_log_n_outputs(log)), HaltReason::exit(0), &run_pub) == 0);
  CHECK(zkir_prover::verify_chain_io(proofs, {}, {1}, HaltReason::exit(0)) == 50);
  CHECK(zkir_prover::verify_chain({proofs[0], proofs[2]}) == 42 && zkir_prover::verify_chain({proofs[1], proofs[2]}) == 41);
  zkir_public_inputs wrong = run_pub; wrong.n_real += 1;
  CHECK(zkir_prover::verify_chain(proofs, &wrong) == 44);
  zkir_delta_log_free(log);
}

int main(int argc, char** argv) {
  const bool gpu = argc > 1 && std::string(argv[1]) == "gpu";
  struct T { const char* name; std::function<void()> f; bool needs_gpu; };
  const std::vector<T> tests = {
      {"test_fibonacci", test_fibonacci, false}, {"test_sum_loop", test_sum_loop, false}, {"test_arithmetic_chain", test_arithmetic_chain, false},
      {"test_branch_taken_skips", test_branch_taken_skips, false}, {"test_store_load_roundtrip", test_store_load_roundtrip, false},
      {"test_echo_input", test_echo_input, false}, {"test_exit_code", test_exit_code, false}, {"test_basic_ebreak", test_basic_ebreak, false},
      {"test_cycle_limit", test_cycle_limit, false}, {"test_trace_disabled_by_default", test_trace_disabled_by_default, false},
      {"test_thousand_instructions", test_thousand_instructions, false}, {"test_divu_by_one", test_divu_by_one, false},
      {"test_mersenne31_field", test_mersenne31_field, false}, {"test_division_by_zero", test_division_by_zero, false}, {"test_invalid_syscall", test_invalid_syscall, false},
      {"test_poseidon2_syscall_is_an_error", test_poseidon2_syscall_is_an_error, false}, {"test_run_consumes_the_vm", test_run_consumes_the_vm, false},
      {"test_execution_trace_rows", test_execution_trace_rows, true}, {"test_trace_with_memory_ops", test_trace_with_memory_ops, true},
      {"test_cycle_limit_trace", test_cycle_limit_trace, true},
      {"test_trace_timestamp_synchronization", test_trace_timestamp_synchronization, true},
      {"test_bound_propagation_and_deferred_checks", test_bound_propagation_and_deferred_checks, true},
      {"test_deferred_carry_normalization_event", test_deferred_carry_normalization_event, true},
      {"test_prove_and_verify", test_prove_and_verify, true}, {"test_prove_modes", test_prove_modes, true}, {"test_prove_result_and_parameters", test_prove_result_and_parameters, true}, {"test_segment_proofs_through_the_c_abi", test_segment_proofs_through_the_c_abi, true}};
  int ran = 0;
  for (const auto& t : tests) {
    if (t.needs_gpu && !gpu) continue;
    const int before = failures;
    try { t.f(); } catch (const std::exception& e) { std::printf("    EXCEPTION in %s: %s\n", t.name, e.what()); failures++; }
    std::printf("%s ... %s\n", t.name, failures == before ? "ok" : "FAILED");
    ran++;
  }
  std::printf("%d tests, %d failures\n", ran, failures);
  return failures ? 1 : 0;
}
