// interp.cpp — product host stage: linear-time ZKIR v3.4 interpreter that records a compact delta log.
//
// Execution is a strict sequential dependency chain (cycle c+1 needs the registers, memory and PC of
// cycle c), so it runs on one host core.  What it emits is NOT the wide trace: per cycle it appends
// (pc, instruction word) and, for every register whose (value, bound, storage-state) triple was written,
// one 32-byte zkir_reg_event; plus flat side logs (data-memory accesses, deferred range checks,
// observation-point normalizations, single-block SHA-256 messages).  The HIP kernels in this directory
// expand those logs into the wide SoA witness columns in HBM.
//
// Semantics follow the reference bit-for-bit (all paths relative to /root/reference):
//   VM::new / VM::run            zkir-runtime/src/vm.rs:138-358
//   execute                      zkir-runtime/src/execute.rs:35-673
//   execute_with_deferred        zkir-runtime/src/execute.rs:888-1003, deferred.rs:81-274, normalize.rs:52-154
//   handle_syscall               zkir-runtime/src/syscall.rs:94-177, crypto.rs:98-105,306-395
//   Memory                       zkir-runtime/src/memory.rs:243-489
//   RangeCheckTracker            zkir-runtime/src/range_check.rs:100-168
//   decode                       zkir-disassembler/src/decoder.rs:20-192
//   ValueBound algebra           zkir-spec/src/bound.rs:126-281
//   Program::from_bytes          zkir-spec/src/program.rs:189-213,318-346
// including the quirks listed in SURVEY.md §8a (Q1-Q10).  Unlike the reference it is O(N): the row's
// memory_ops are the accesses recorded during the cycle (the reference re-filters the whole cumulative
// memory trace every cycle, vm.rs:287-298), with the same `address != fetch_pc` exclusion (Q9).
//
// This file shares no code with oracle/ (the test oracle is a separate, literal restatement).

#include "host.h"

#include <emmintrin.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>

namespace zkir {

// ---- block pool (host.h) --------------------------------------------------------------------------------------------------
namespace {
std::mutex g_pool_mu;
std::vector<Block> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t POOL_MAX_BLOCKS = 8, POOL_MAX_BYTES = 4ull << 30;
void block_free(const Block& b) { if (b.pinned) pinned_free(b.p); else free(b.p); }
}  // namespace

Block block_acquire(size_t min_bytes) {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (int i = 0; i < (int)g_pool.size(); i++)
      if (g_pool[i].bytes >= min_bytes && g_pool[i].bytes <= 4 * min_bytes && (best < 0 || g_pool[i].bytes < g_pool[best].bytes)) best = i;
    if (best >= 0) {
      const Block b = g_pool[best];
      g_pool.erase(g_pool.begin() + best);
      g_pool_bytes -= b.bytes;
      return b;
    }
  }
  static const bool pin = !(getenv("ZKIR_PIN_LOG") && atoi(getenv("ZKIR_PIN_LOG")) == 0);
  if (pin) if (void* q = pinned_alloc(min_bytes)) return Block{q, min_bytes, true};
  void* q = malloc(min_bytes);
  // pageable: the trace logs are tens of MB written once, front to back: with 4 KiB pages two thirds of the interpreter's time went to
  // first-touch page faults (21 ms vs 7 ms without the trace at 2^20 rows); ask for transparent huge pages (THP = madvise here)
  if (q) (void)madvise(q, min_bytes, MADV_HUGEPAGE);
  return Block{q, q ? min_bytes : 0, false};
}

void block_release(const Block& b) {
  if (!b.p) return;
  std::vector<Block> drop;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (b.bytes > POOL_MAX_BYTES) drop.push_back(b);
    else {
      g_pool.push_back(b);
      g_pool_bytes += b.bytes;
      while (g_pool.size() > POOL_MAX_BLOCKS || g_pool_bytes > POOL_MAX_BYTES) {    // drop the oldest
        g_pool_bytes -= g_pool.front().bytes;
        drop.push_back(g_pool.front());
        g_pool.erase(g_pool.begin());
      }
    }
  }
  for (const Block& d : drop) block_free(d);
}


// ------------------------------------------------------------------------------------------------
// bounds (bound.rs)
// ------------------------------------------------------------------------------------------------
namespace {

inline uint32_t add_sat(uint32_t a, uint32_t b) { uint32_t s = a + b; return s < a ? 0xFFFFFFFFu : s; }
inline uint32_t sub_sat(uint32_t a, uint32_t b) { return a < b ? 0u : a - b; }
inline uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
inline uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

inline BoundT computed(uint32_t bits) { return BoundT{bits, ZKIR_BOUND_COMPUTED, 0}; }                       // bound.rs:168
inline BoundT type_width(uint32_t bits) { return BoundT{bits, ZKIR_BOUND_TYPE_WIDTH, bits}; }               // bound.rs:135
inline BoundT constant(uint64_t v) { return BoundT{v ? 64u - (uint32_t)__builtin_clzll(v) : 0u, ZKIR_BOUND_CONSTANT, v}; }  // bound.rs:154

constexpr uint64_t M40 = (1ull << 40) - 1;
constexpr uint64_t SIGN40 = 1ull << 39;

// Value40 shifts (value.rs:658-697)
inline uint64_t shl40(uint64_t v, uint32_t s) { return s >= 40 ? 0 : (v << s) & M40; }
inline uint64_t srl40(uint64_t v, uint32_t s) { return s >= 40 ? 0 : v >> s; }
inline uint64_t sra40(uint64_t v, uint32_t s) {
  const bool neg = (v & SIGN40) != 0;
  if (s >= 40) return neg ? M40 : 0;
  uint64_t r = v >> s;
  if (neg) r |= (((1ull << s) - 1) << (40 - s));
  return r & M40;
}
inline bool slt40(uint64_t a, uint64_t b) { return (a ^ SIGN40) < (b ^ SIGN40); }   // value.rs:710-716

// ------------------------------------------------------------------------------------------------
// decode (decoder.rs) — one table-driven pass, cached per pc
// ------------------------------------------------------------------------------------------------
enum Fmt : uint8_t { F_BAD = 0, F_R, F_I, F_SH, F_SB, F_J, F_SYS };
struct FmtTable {
  uint8_t f[128];
  constexpr FmtTable() : f{} {
    for (int i = 0; i < 128; i++) f[i] = F_BAD;
    for (int i = 0x00; i <= 0x07; i++) f[i] = F_R;
    f[0x08] = F_I;
    for (int i = 0x10; i <= 0x12; i++) f[i] = F_R;
    for (int i = 0x13; i <= 0x15; i++) f[i] = F_I;
    for (int i = 0x18; i <= 0x1A; i++) f[i] = F_R;
    for (int i = 0x1B; i <= 0x1D; i++) f[i] = F_SH;
    for (int i = 0x20; i <= 0x28; i++) f[i] = F_R;
    for (int i = 0x30; i <= 0x35; i++) f[i] = F_I;
    for (int i = 0x38; i <= 0x3B; i++) f[i] = F_SB;
    for (int i = 0x40; i <= 0x45; i++) f[i] = F_SB;
    f[0x48] = F_J; f[0x49] = F_I;
    f[0x50] = F_SYS; f[0x51] = F_SYS;
  }
};
constexpr FmtTable FMT{};

struct Decoded {
  uint8_t op, a, b, c;   // a = bits 10:7, b = bits 14:11, c = bits 18:15 (register fields by position)
  int32_t imm;           // sign-extended imm17 / off21, or shamt (8 bits)
};
inline bool decode_word(uint32_t w, Decoded& d) {
  const uint8_t op = w & 0x7F;
  const uint8_t fmt = FMT.f[op];
  if (fmt == F_BAD) return false;
  d.op = op; d.a = (w >> 7) & 0xF; d.b = (w >> 11) & 0xF; d.c = (w >> 15) & 0xF;
  switch (fmt) {
    case F_I: case F_SB: d.imm = (int32_t)(w & 0xFFFF8000u) >> 15; break;     // 17-bit field at bit 15, arithmetic shift sign-extends
    case F_SH: d.imm = (w >> 15) & 0xFF; break;                                // decoder.rs:140
    case F_J: d.imm = (int32_t)(w & 0xFFFFF800u) >> 11; break;                 // 21-bit field at bit 11
    default: d.imm = 0;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// memory (memory.rs): sparse 4 KiB pages, little-endian, zero-fill reads; small direct-mapped TLB
// ------------------------------------------------------------------------------------------------
class PagedMemory {
 public:
  static constexpr uint64_t PAGE_BITS = 12, PAGE = 1ull << PAGE_BITS, OFF_MASK = PAGE - 1;
  ~PagedMemory() { for (auto& kv : table_) free(kv.second); }
  inline const uint8_t* rpage(uint64_t pn) {
    Tlb& t = tlb_[pn & (NTLB - 1)];
    if (t.pn == pn) return t.p;
    auto it = table_.find(pn);
    if (it == table_.end()) return nullptr;        // absent pages are not cached: a later write may create them
    t.pn = pn; t.p = it->second;
    return t.p;
  }
  inline uint8_t* wpage(uint64_t pn) {
    Tlb& t = tlb_[pn & (NTLB - 1)];
    if (t.pn == pn) return t.p;
    uint8_t*& slot = table_[pn];
    if (!slot) slot = (uint8_t*)calloc(PAGE, 1);
    t.pn = pn; t.p = slot;
    return slot;
  }
  // aligned power-of-two accesses never straddle a page
  template <typename T> inline T load(uint64_t addr) {
    const uint8_t* p = rpage(addr >> PAGE_BITS);
    if (!p) return 0;
    T v; memcpy(&v, p + (addr & OFF_MASK), sizeof(T));
    return v;
  }
  template <typename T> inline void store(uint64_t addr, T v) { memcpy(wpage(addr >> PAGE_BITS) + (addr & OFF_MASK), &v, sizeof(T)); }

 private:
  static constexpr size_t NTLB = 64;
  struct Tlb { uint64_t pn = ~0ull; uint8_t* p = nullptr; };
  Tlb tlb_[NTLB];
  std::unordered_map<uint64_t, uint8_t*> table_;
};

struct PendingCheck { uint64_t value40; uint32_t max_bits; uint64_t pc; };

std::string hexs(uint64_t v) { char b[32]; snprintf(b, sizeof b, "0x%llx", (unsigned long long)v); return b; }

}  // namespace

// ------------------------------------------------------------------------------------------------
// program blob (program.rs:189-213, 318-346; config.rs:154-174)
// ------------------------------------------------------------------------------------------------
static inline uint32_t le32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

Status parse_program(const uint8_t* b, size_t len, ProgramView& pv) {
  char m[160];
  if (len < 32) { snprintf(m, sizeof m, "Invalid header size: expected 32 bytes, found %zu bytes", len); return {ZKIR_ERR_BAD_PROGRAM, m}; }
  const uint32_t magic = le32(b), version = le32(b + 4);
  if (magic != 0x52494B5Au) { snprintf(m, sizeof m, "Invalid program magic: expected 0x5A4B4952, got 0x%08x", magic); return {ZKIR_ERR_BAD_PROGRAM, m}; }
  if (version != 0x00030004u) { snprintf(m, sizeof m, "Invalid program version: expected 0x00030004, found 0x%08x", version); return {ZKIR_ERR_BAD_PROGRAM, m}; }
  pv.limb_bits = b[8]; pv.data_limbs = b[9]; pv.addr_limbs = b[10];
  if (pv.limb_bits < 16 || pv.limb_bits > 30) return {ZKIR_ERR_BAD_PROGRAM, "Invalid configuration: limb_bits must be in range [16, 30]"};
  if (pv.limb_bits & 1) return {ZKIR_ERR_BAD_PROGRAM, "Invalid configuration: limb_bits must be even"};
  if (pv.data_limbs < 1 || pv.data_limbs > 4) return {ZKIR_ERR_BAD_PROGRAM, "Invalid configuration: data_limbs must be in range [1, 4]"};
  if (pv.addr_limbs < 1 || pv.addr_limbs > 2) return {ZKIR_ERR_BAD_PROGRAM, "Invalid configuration: addr_limbs must be in range [1, 2]"};
  pv.entry_point = le32(b + 12);
  const uint64_t code_size = le32(b + 16), data_size = le32(b + 20);
  const uint64_t need = 32 + code_size + data_size;
  if (len < need) { snprintf(m, sizeof m, "Invalid program size: expected %llu bytes, found %zu bytes", (unsigned long long)need, len); return {ZKIR_ERR_BAD_PROGRAM, m}; }
  if (code_size % 4) { snprintf(m, sizeof m, "Invalid code size: expected %llu bytes, found %llu bytes", (unsigned long long)code_size, (unsigned long long)(code_size / 4 * 4)); return {ZKIR_ERR_BAD_PROGRAM, m}; }
  pv.code = b + 32; pv.n_code_words = code_size / 4;
  pv.data = b + 32 + code_size; pv.n_data = data_size;
  return {};
}

// ------------------------------------------------------------------------------------------------
// the machine
// ------------------------------------------------------------------------------------------------
namespace {

class Machine {
 public:
  Machine(const ProgramView& pv, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config& cfg, DeltaLog& log, Progress* progress)
      : cfg_(cfg), log_(log), inputs_(inputs), n_inputs_(n_inputs), progress_(progress) {
    tracing_ = cfg.enable_execution_trace != 0;
    deferred_ = cfg.enable_deferred_model != 0;
    range_ = cfg.enable_range_checking != 0;
    data_bits_ = (uint32_t)pv.limb_bits * pv.data_limbs;            // Config::data_bits, config.rs:60
    log.rc_chunk_bits = pv.limb_bits / 2;                           // range_check.rs:29
    pc_ = pv.entry_point;
    for (int r = 0; r < 16; r++) { reg_[r] = 0; bound_[r] = BoundT{40, ZKIR_BOUND_PROGRAM_WIDTH, 0}; state_[r] = 0; }   // state.rs:55-71
    bound_[0] = constant(0);
    // load_code at CODE_BASE, data right behind it (vm.rs:153-170); not part of the memory trace
    for (uint64_t i = 0; i < pv.n_code_words; i++) mem_.store<uint32_t>(0x1000 + 4 * i, le32(pv.code + 4 * i));
    const uint64_t dbase = 0x1000 + 4 * pv.n_code_words;
    for (uint64_t i = 0; i < pv.n_data; i++) mem_.store<uint8_t>(dbase + i, pv.data[i]);
    for (auto& e : icache_) e.pc = ~0ull;
    code_bytes_ = 4 * pv.n_code_words;
    words_.resize(pv.n_code_words); dec_.resize(pv.n_code_words);
    for (uint64_t i = 0; i < pv.n_code_words; i++) decode_slot(i);
  }

  // Trace WINDOW (multi-GPU row shards, include/zkir_amd.h: zkir_interpret_window): execution is sequential, so the state at row
  // `begin` is only known by executing rows [0, begin) — which this machine does in the UNTRACED specialisation of the loop (no log
  // stores at all), then it switches to the traced one with the 16 snapshot events taken from the live registers, records rows
  // [begin, end) relative to `begin` (cycle_base = begin) and stops.
  void set_window(uint64_t begin, uint64_t end) { win_begin_ = begin; win_end_ = end; }

  Status run() {                                                       // the loop is specialised on the three mode flags
    if (tracing_ && win_begin_ > 0) {                                  // fast-forward: same semantics, nothing recorded
      tracing_ = false; skipping_ = true;                              // (skipping_: the side logs — range-check witnesses, normalization events — of the skipped rows are not recorded either)
      stop_at_ = win_begin_;
      Status st = dispatch();
      tracing_ = true; skipping_ = false;
      if (!st.ok()) return st;
      // side logs of the skipped rows do not belong to the window
      log_.rc_events.clear(); log_.rc_offsets.clear(); log_.rc_cycles.clear(); log_.norm_events.clear(); log_.sha_blocks.clear();
      base_ = cycle_;                                                  // = win_begin_, or the halting cycle if the run ended before the window
      log_.cycle_base = base_;                                         // known to a streaming consumer before the first tile is published
      if (halted_) { finish_log(true); log_.rc_offsets.push_back(0); return {}; }
    }
    stop_at_ = win_end_;
    return dispatch();
  }

 private:
  Status dispatch() {
    const int m = (tracing_ ? 4 : 0) | (deferred_ ? 2 : 0) | (range_ ? 1 : 0);
    switch (m) {
      case 0: return run_loop<false, false, false>(); case 1: return run_loop<false, false, true>();
      case 2: return run_loop<false, true, false>(); case 3: return run_loop<false, true, true>();
      case 4: return run_loop<true, false, false>(); case 5: return run_loop<true, false, true>();
      case 6: return run_loop<true, true, false>(); default: return run_loop<true, true, true>();
    }
  }
  void finish_log(bool traced) {
    log_.cycles = cycle_; log_.halt_kind = halt_kind_; log_.halt_code = halt_code_;
    log_.n_rows = traced ? cycle_ - base_ : 0;
    log_.cycle_base = base_;
    log_.window_open = !halted_;                                       // stopped at the window's end, not at a halt
  }

  template <bool TRACING, bool DEFERRED, bool RANGE> Status run_loop();
  // ---- register file with write tracking ----
  inline uint64_t rd(uint8_t r) const { return reg_[r]; }            // reg_[0] is never written, so it reads 0 (state.rs:76-82)
  inline void wr(uint8_t r, uint64_t v, const BoundT& b) { if (r) { reg_[r] = v; bound_[r] = b; dirty_ |= 1u << r; } }   // state.rs:110-113
  inline void wr_value(uint8_t r, uint64_t v) { if (r) { reg_[r] = v; dirty_ |= 1u << r; } }
  inline void wr_bound(uint8_t r, const BoundT& b) { if (r) { bound_[r] = b; dirty_ |= 1u << r; } }
  inline void wr_state(uint8_t r, uint8_t s) { if (r) { state_[r] = s; dirty_ |= 1u << r; } }

  // ---- data memory with the architectural-access log (memory.rs:243-253) ----
  inline void note(uint64_t addr, uint64_t value, bool is_write, uint8_t width) {
    if (tracing_ && addr != fetch_pc_) log_.mem_events.push(zkir_mem_event{addr, value, (uint32_t)(cycle_ - base_), (uint8_t)is_write, width, 0});  // Q9
  }
  inline Status misaligned(uint64_t addr, unsigned al) { return {ZKIR_ERR_MISALIGNED, "Misaligned access: address " + hexs(addr) + ", alignment " + std::to_string(al)}; }
  // Errors are rare: the hot path returns plain bools and the Status (with its message string) is only built on failure.
  Status err_;
  inline bool fail(Status s) { err_ = std::move(s); return false; }
  template <typename T> inline bool load(uint64_t addr, T& out) {
    if (sizeof(T) > 1 && (addr & (sizeof(T) - 1))) return fail(misaligned(addr, sizeof(T)));
    out = mem_.load<T>(addr);
    note(addr, (uint64_t)out, false, sizeof(T));
    return true;
  }
  template <typename T> inline bool store(uint64_t addr, T v) {
    if (sizeof(T) > 1 && (addr & (sizeof(T) - 1))) return fail(misaligned(addr, sizeof(T)));
    mem_.store<T>(addr, v);
    if (__builtin_expect(addr - 0x1000 < code_bytes_, 0)) refresh_code(addr);    // self-modifying code: the pre-decoded image follows memory
    note(addr, (uint64_t)v, true, sizeof(T));
    return true;
  }
  // pre-decoded code image: word i of the loaded program (address 0x1000 + 4 i) as fetched from memory and its decoded fields
  // (op = 0xFF: not a valid opcode).  Kept in step with stores into the code region (aligned accesses never straddle a word
  // boundary, except SD which covers two).
  void decode_slot(uint64_t i) {
    const uint32_t w = mem_.load<uint32_t>(0x1000 + 4 * i);
    words_[i] = w;
    if (!decode_word(w, dec_[i])) dec_[i].op = 0xFF;
  }
  void refresh_code(uint64_t addr) {
    const uint64_t i = (addr - 0x1000) >> 2;
    decode_slot(i);
    if (i + 1 < words_.size()) decode_slot(i + 1);
  }

  // ---- deferred carry model helpers (state.rs:184-220, normalize.rs) ----
  inline void limbs_of(uint8_t r, uint64_t out[2]) const {
    const uint64_t v = reg_[r];
    const unsigned bits = (r == 0 || state_[r] == 0) ? 20 : 30;
    const uint64_t mask = (1ull << bits) - 1;
    out[0] = v & mask; out[1] = (v >> bits) & mask;
  }
  inline void renormalize(uint8_t r) {                               // carry extraction + repack at 20 bits, mark Normalized
    uint64_t acc[2]; limbs_of(r, acc);
    const uint64_t c0 = acc[0] >> 20, n0 = acc[0] & 0xFFFFF;
    const uint64_t t = acc[1] + (uint32_t)c0;
    const uint64_t n1 = t & 0xFFFFF;
    wr_value(r, n0 | (n1 << 20)); wr_state(r, 0);
  }
  inline void normalize_silent(uint8_t r) { if (r && state_[r]) renormalize(r); }                 // normalize.rs:65-106
  inline void normalize_observed(uint8_t r, uint8_t opcode) {                                     // normalize.rs:121-154 + execute.rs:903-929
    if (!r) return;
    if (!skipping_) log_.norm_events.push(zkir_norm_event{cycle_, fetch_pc_, reg_[r], r, state_[r], opcode, {0, 0, 0, 0, 0}});
    renormalize(r);
  }
  inline void write_accumulated(uint8_t r, const uint64_t l[2]) { if (r) { wr_value(r, l[0] | (l[1] << 30)); wr_state(r, 1); } }  // state.rs:184-192

  inline __attribute__((always_inline)) bool step(const Decoded& d);            // false => err_ holds the RuntimeError
  bool step_deferred(const Decoded& d);
  bool syscall();
  bool hash_syscall(int which);
  void flush_range_checks();

  const zkir_vm_config cfg_;
  DeltaLog& log_;
  const uint64_t* inputs_; size_t n_inputs_; size_t input_pos_ = 0;
  bool tracing_, deferred_, range_, skipping_ = false;
  uint32_t data_bits_;

  uint64_t pc_ = 0, fetch_pc_ = 0, cycle_ = 0;
  uint64_t win_begin_ = 0, win_end_ = ~0ull, stop_at_ = ~0ull, base_ = 0;   // trace window; rows of the log are cycle_ - base_
  uint64_t reg_[16]; BoundT bound_[16]; uint8_t state_[16];
  uint32_t dirty_ = 0;
  bool halted_ = false; int halt_kind_ = ZKIR_HALT_EBREAK; uint64_t halt_code_ = 0;

  PagedMemory mem_;
  struct ICacheEntry { uint64_t pc; uint32_t word; Decoded d; };
  static constexpr size_t NICACHE = 1 << 12;
  ICacheEntry icache_[NICACHE];

  std::vector<PendingCheck> pending_;
  Progress* progress_ = nullptr;
  uint64_t code_bytes_ = 0;
  std::vector<uint32_t> words_;
  std::vector<Decoded> dec_;
};

// execute.rs:35-673.  Register fields by position: R/I-type rd=a rs1=b rs2=c; S/B-type rs1=a rs2=b.
inline __attribute__((always_inline)) bool Machine::step(const Decoded& d) {
  const uint64_t immu = (uint64_t)(int64_t)d.imm;            // `imm as u64` sign-extends (Q4)
  switch (d.op) {
    case 0x00: {  // ADD :43-63
      const uint64_t v = ((rd(d.b) & M40) + (rd(d.c) & M40)) & M40;
      const BoundT nb = computed(add_sat(umax(bound_[d.b].max_bits, bound_[d.c].max_bits), 1));
      wr(d.a, v, nb);
      if (range_ && nb.max_bits > data_bits_) pending_.push_back({v, nb.max_bits, pc_});
      pc_ += 4; break;
    }
    case 0x01: {  // SUB :65-77
      wr(d.a, ((rd(d.b) & M40) - (rd(d.c) & M40)) & M40, computed(umax(bound_[d.b].max_bits, bound_[d.c].max_bits)));
      pc_ += 4; break;
    }
    case 0x02: {  // MUL :79-99
      const uint64_t v = ((rd(d.b) & M40) * (rd(d.c) & M40)) & M40;
      const BoundT nb = computed(add_sat(bound_[d.b].max_bits, bound_[d.c].max_bits));
      wr(d.a, v, nb);
      if (range_ && nb.max_bits > data_bits_) pending_.push_back({v, nb.max_bits, pc_});
      pc_ += 4; break;
    }
    case 0x03: {  // MULH :101-115 — raw operands (Q3)
      const unsigned __int128 p = (unsigned __int128)rd(d.b) * rd(d.c);
      wr(d.a, (uint64_t)(p >> 40) & M40, computed(add_sat(bound_[d.b].max_bits, bound_[d.c].max_bits)));
      pc_ += 4; break;
    }
    case 0x04: case 0x05: {  // DIVU / REMU :134-149, :168-183
      const uint64_t a = rd(d.b), b = rd(d.c);
      if (b == 0) return fail({ZKIR_ERR_DIV_ZERO, "Division by zero at PC " + hexs(pc_)});
      wr(d.a, d.op == 0x04 ? a / b : a % b, computed(bound_[d.b].max_bits));     // after_div for both (Q2)
      pc_ += 4; break;
    }
    case 0x06: case 0x07: {  // DIV / REM :117-132, :151-166 — raw u64 reinterpreted as i64 (Q2)
      const int64_t a = (int64_t)rd(d.b), b = (int64_t)rd(d.c);
      if (b == 0) return fail({ZKIR_ERR_DIV_ZERO, "Division by zero at PC " + hexs(pc_)});
      uint64_t r;
      if (a == INT64_MIN && b == -1) r = d.op == 0x06 ? (uint64_t)INT64_MIN : 0;   // wrapping_div / wrapping_rem
      else r = d.op == 0x06 ? (uint64_t)(a / b) : (uint64_t)(a % b);
      wr(d.a, r, computed(bound_[d.b].max_bits));
      pc_ += 4; break;
    }
    case 0x08: {  // ADDI :185-197
      wr(d.a, ((rd(d.b) & M40) + (immu & M40)) & M40, computed(add_sat(umax(bound_[d.b].max_bits, constant(immu).max_bits), 1)));
      pc_ += 4; break;
    }
    case 0x10: wr(d.a, (rd(d.b) & rd(d.c)) & M40, computed(umin(bound_[d.b].max_bits, bound_[d.c].max_bits))); pc_ += 4; break;  // AND :200-212
    case 0x11: wr(d.a, (rd(d.b) | rd(d.c)) & M40, computed(umax(bound_[d.b].max_bits, bound_[d.c].max_bits))); pc_ += 4; break;  // OR  :214-226
    case 0x12: wr(d.a, (rd(d.b) ^ rd(d.c)) & M40, computed(umax(bound_[d.b].max_bits, bound_[d.c].max_bits))); pc_ += 4; break;  // XOR :228-240
    case 0x13: wr(d.a, (rd(d.b) & immu) & M40, computed(umin(bound_[d.b].max_bits, constant(immu).max_bits))); pc_ += 4; break;  // ANDI :242-254
    case 0x14: wr(d.a, (rd(d.b) | immu) & M40, computed(umax(bound_[d.b].max_bits, constant(immu).max_bits))); pc_ += 4; break;  // ORI  :256-268
    case 0x15: wr(d.a, (rd(d.b) ^ immu) & M40, computed(umax(bound_[d.b].max_bits, constant(immu).max_bits))); pc_ += 4; break;  // XORI :270-282
    case 0x18: case 0x1B: {  // SLL / SLLI :285-296, :324-334
      const uint32_t s = d.op == 0x18 ? (uint32_t)(rd(d.c) & 0x3F) : (uint32_t)d.imm;
      wr(d.a, shl40(rd(d.b) & M40, s), computed(umin(add_sat(bound_[d.b].max_bits, s), 40)));
      pc_ += 4; break;
    }
    case 0x19: case 0x1C: {  // SRL / SRLI :298-309, :336-346
      const uint32_t s = d.op == 0x19 ? (uint32_t)(rd(d.c) & 0x3F) : (uint32_t)d.imm;
      wr(d.a, srl40(rd(d.b) & M40, s), computed(sub_sat(bound_[d.b].max_bits, s)));
      pc_ += 4; break;
    }
    case 0x1A: case 0x1D: {  // SRA / SRAI :311-322, :348-358
      const uint32_t s = d.op == 0x1A ? (uint32_t)(rd(d.c) & 0x3F) : (uint32_t)d.imm;
      const uint32_t bb = bound_[d.b].max_bits;
      wr(d.a, sra40(rd(d.b) & M40, s), computed(bb >= 40 ? 40 : sub_sat(bb, s)));
      pc_ += 4; break;
    }
    case 0x20: wr(d.a, (rd(d.b) & M40) < (rd(d.c) & M40), computed(1)); pc_ += 4; break;          // SLTU :373-383
    case 0x21: wr(d.a, !((rd(d.b) & M40) < (rd(d.c) & M40)), computed(1)); pc_ += 4; break;       // SGEU :397-407
    case 0x22: wr(d.a, slt40(rd(d.b) & M40, rd(d.c) & M40), computed(1)); pc_ += 4; break;        // SLT  :361-371
    case 0x23: wr(d.a, !slt40(rd(d.b) & M40, rd(d.c) & M40), computed(1)); pc_ += 4; break;       // SGE  :385-395
    case 0x24: wr(d.a, rd(d.b) == rd(d.c), computed(1)); pc_ += 4; break;                         // SEQ  :409-419 raw compare (Q6)
    case 0x25: wr(d.a, rd(d.b) != rd(d.c), computed(1)); pc_ += 4; break;                         // SNE  :421-431
    case 0x26: case 0x28: case 0x27: {  // CMOV ≡ CMOVNZ, CMOVZ :434-474
      const bool cond = d.op == 0x27 ? rd(d.c) == 0 : rd(d.c) != 0;
      if (cond) wr(d.a, rd(d.b), computed(umax(bound_[d.b].max_bits, bound_[d.a].max_bits)));
      pc_ += 4; break;
    }
    case 0x30: case 0x31: {  // LB / LBU :477-499 (Q1: LB keeps the 64-bit sign extension)
      uint8_t v; if (!load<uint8_t>(rd(d.b) + immu, v)) return false;
      wr(d.a, d.op == 0x30 ? (uint64_t)(int64_t)(int8_t)v : (uint64_t)v, type_width(8));
      pc_ += 4; break;
    }
    case 0x32: case 0x33: {  // LH / LHU :501-523
      uint16_t v; if (!load<uint16_t>(rd(d.b) + immu, v)) return false;
      wr(d.a, d.op == 0x32 ? (uint64_t)(int64_t)(int16_t)v : (uint64_t)v, type_width(16));
      pc_ += 4; break;
    }
    case 0x34: { uint32_t v; if (!load<uint32_t>(rd(d.b) + immu, v)) return false; wr(d.a, v, type_width(32)); pc_ += 4; break; }   // LW :525-535 zero-extends
    case 0x35: { uint64_t v; if (!load<uint64_t>(rd(d.b) + immu, v)) return false; wr(d.a, v, type_width(40)); pc_ += 4; break; }   // LD :537-546
    case 0x38: if (!store<uint8_t>(rd(d.a) + immu, (uint8_t)rd(d.b))) return false; pc_ += 4; break;      // SB :549-554
    case 0x39: if (!store<uint16_t>(rd(d.a) + immu, (uint16_t)rd(d.b))) return false; pc_ += 4; break;    // SH :556-561
    case 0x3A: if (!store<uint32_t>(rd(d.a) + immu, (uint32_t)rd(d.b))) return false; pc_ += 4; break;    // SW :563-568
    case 0x3B: if (!store<uint64_t>(rd(d.a) + immu, rd(d.b))) return false; pc_ += 4; break;              // SD :570-575
    case 0x40: pc_ += (rd(d.a) == rd(d.b)) ? (uint64_t)(int64_t)d.imm : 4; break;                          // BEQ :578-586 raw compare; target = own pc + off (Q7)
    case 0x41: pc_ += (rd(d.a) != rd(d.b)) ? (uint64_t)(int64_t)d.imm : 4; break;                          // BNE :588-596
    case 0x42: pc_ += slt40(rd(d.a) & M40, rd(d.b) & M40) ? (uint64_t)(int64_t)d.imm : 4; break;           // BLT :598-606
    case 0x43: pc_ += !slt40(rd(d.a) & M40, rd(d.b) & M40) ? (uint64_t)(int64_t)d.imm : 4; break;          // BGE :608-616
    case 0x44: pc_ += ((rd(d.a) & M40) < (rd(d.b) & M40)) ? (uint64_t)(int64_t)d.imm : 4; break;           // BLTU :618-626
    case 0x45: pc_ += !((rd(d.a) & M40) < (rd(d.b) & M40)) ? (uint64_t)(int64_t)d.imm : 4; break;          // BGEU :628-636
    case 0x48: { const uint64_t ra = pc_ + 4; wr(d.a, ra, constant(ra)); pc_ += (uint64_t)(int64_t)d.imm; break; }   // JAL :639-647
    case 0x49: {  // JALR :649-658
      const uint64_t ra = pc_ + 4, target = rd(d.b) + immu;
      wr(d.a, ra, constant(ra));
      pc_ = target & ~1ull; break;
    }
    case 0x50: pc_ += 4; break;                                                                     // ECALL :661-665 (syscall runs afterwards, Q8)
    case 0x51: halted_ = true; halt_kind_ = ZKIR_HALT_EBREAK; break;                                // EBREAK :667-669
  }
  return true;
}

// execute.rs:888-1003
bool Machine::step_deferred(const Decoded& d) {
  enum { NONE, ONE, TWO_RI, TWO_SB };
  int kind = NONE;
  switch (d.op) {
    case 0x40: case 0x41: case 0x42: case 0x43: case 0x44: case 0x45:      // branches
    case 0x3A: case 0x39: case 0x38: kind = TWO_SB; break;                  // SW SH SB (not SD)
    case 0x10: case 0x11: case 0x12: case 0x18: case 0x19: case 0x1A:       // and or xor sll srl sra
    case 0x02: case 0x03: case 0x04: case 0x05: case 0x06: case 0x07:       // mul mulh divu remu div rem
    case 0x20: case 0x21: case 0x22: case 0x23: case 0x24: case 0x25: kind = TWO_RI; break;   // compares
    case 0x13: case 0x14: case 0x15: case 0x1B: case 0x1C: case 0x1D: kind = ONE; break;       // immediates
    default: break;
  }
  if (kind == TWO_SB) { normalize_observed(d.a, d.op); normalize_silent(d.b); }
  else if (kind == TWO_RI) { normalize_observed(d.b, d.op); normalize_silent(d.c); }
  else if (kind == ONE) { normalize_observed(d.b, d.op); }

  if (d.op == 0x00) {                         // execute_add_deferred, deferred.rs:81-138
    uint64_t a[2], b[2]; limbs_of(d.b, a); limbs_of(d.c, b);
    uint64_t r[2] = {a[0] + b[0], a[1] + b[1]};
    if ((r[0] | r[1]) >> 30) {                // would_overflow (normalize.rs:230)
      normalize_silent(d.b); normalize_silent(d.c);
      limbs_of(d.b, a); limbs_of(d.c, b);
      r[0] = a[0] + b[0]; r[1] = a[1] + b[1];
    }
    write_accumulated(d.a, r);
    wr_bound(d.a, computed(add_sat(umax(bound_[d.b].max_bits, bound_[d.c].max_bits), 1)));
    pc_ += 4;
  } else if (d.op == 0x01) {                  // execute_sub_deferred, deferred.rs:163-206
    uint64_t a[2], b[2]; limbs_of(d.b, a); limbs_of(d.c, b);
    const uint64_t r[2] = {a[0] - b[0], a[1] - b[1]};
    write_accumulated(d.a, r);
    wr_bound(d.a, computed(umax(bound_[d.b].max_bits, bound_[d.c].max_bits)));
    pc_ += 4;
  } else if (d.op == 0x08) {                  // execute_addi_deferred, deferred.rs:220-274
    const uint64_t imm = (uint64_t)(int64_t)d.imm;
    const uint64_t il[2] = {imm & 0xFFFFF, (imm >> 20) & 0xFFFFF};
    uint64_t a[2]; limbs_of(d.b, a);
    uint64_t r[2] = {a[0] + il[0], a[1] + il[1]};
    if ((r[0] | r[1]) >> 30) {
      normalize_silent(d.b);
      limbs_of(d.b, a);
      r[0] = a[0] + il[0]; r[1] = a[1] + il[1];
    }
    write_accumulated(d.a, r);
    wr_bound(d.a, computed(add_sat(umax(bound_[d.b].max_bits, constant(imm).max_bits), 1)));
    pc_ += 4;
  } else {
    return step(d);
  }
  return true;
}

// syscall.rs:94-177
bool Machine::syscall() {
  const uint64_t num = reg_[10];
  switch (num) {
    case 0: halted_ = true; halt_kind_ = ZKIR_HALT_EXIT; halt_code_ = reg_[11]; return true;
    case 1: wr_value(10, input_pos_ < n_inputs_ ? inputs_[input_pos_++] : 0); return true;   // READ: bound of R10 untouched (a10)
    case 2: log_.outputs.push_back(reg_[11]); return true;
    case 3: return hash_syscall(0);
    case 4: return fail({ZKIR_ERR_OTHER, "Poseidon2 not yet implemented"});                    // crypto.rs:306-315
    case 5: return hash_syscall(1);
    case 6: return hash_syscall(2);
    default: return fail({ZKIR_ERR_INVALID_SYSCALL, "Invalid syscall: " + std::to_string(num)});
  }
}

bool Machine::hash_syscall(int which) {
  const uint64_t ip = reg_[11], il = reg_[12], op = reg_[13];
  if (il > (1ull << 32)) return fail({ZKIR_ERR_OTHER, "hash input length too large"});   // the reference would abort in Vec::with_capacity
  std::vector<uint8_t> in((size_t)il);
  for (uint64_t i = 0; i < il; i++) { uint8_t b; load<uint8_t>(ip + i, b); in[i] = b; }           // one width-1 Read per byte (crypto.rs:232-235)
  if (which == 0) {
    uint32_t h[8];
    sha256(in.data(), in.size(), h);
    for (int i = 0; i < 8; i++) if (!store<uint32_t>(op + 4 * (uint64_t)i, h[i])) return false;  // crypto.rs:252-255
    wr_value(10, 0);
    wr_bound(14, BoundT{32, ZKIR_BOUND_CRYPTO_OUTPUT, 0});                                        // syscall.rs:135
    if (tracing_ && il < 56) {                                                                    // single-block message -> SHA chip input (crypto.rs:108-139)
      zkir_sha_block blk; uint8_t raw[64] = {0};
      memcpy(raw, in.data(), in.size()); raw[in.size()] = 0x80;
      const uint64_t bits = il * 8;
      for (int i = 0; i < 8; i++) raw[56 + i] = (uint8_t)(bits >> (8 * (7 - i)));
      for (int i = 0; i < 16; i++) blk.message_block[i] = ((uint32_t)raw[4 * i] << 24) | ((uint32_t)raw[4 * i + 1] << 16) | ((uint32_t)raw[4 * i + 2] << 8) | raw[4 * i + 3];
      blk.timestamp = cycle_;
      log_.sha_blocks.push(blk);
    }
  } else {
    uint8_t dgst[32];
    if (which == 1) keccak256(in.data(), in.size(), dgst); else blake3(in.data(), in.size(), dgst);
    for (int i = 0; i < 32; i++) store<uint8_t>(op + (uint64_t)i, dgst[i]);                       // crypto.rs:351-353, :390-392
    wr_value(10, 0);
  }
  return true;
}

void Machine::flush_range_checks() {                      // RangeCheckTracker::checkpoint, range_check.rs:140-168
  if (!skipping_) {                                       // (a window's fast-forward keeps the tracker's state in step and records nothing)
    for (const PendingCheck& p : pending_) log_.rc_events.push(zkir_rc_event{p.value40, p.pc});
    if (!pending_.empty()) { log_.rc_offsets.push_back(log_.rc_events.size()); log_.rc_cycles.push_back(cycle_); }   // vm.rs:340-342: empty witnesses are dropped
  }
  pending_.clear();
}

template <bool TRACING, bool DEFERRED, bool RANGE>
Status Machine::run_loop() {
  constexpr bool tracing_ = TRACING, deferred_ = DEFERRED, range_ = RANGE;   // shadow the members: compile-time in this instantiation
  const uint32_t T = log_.tile_rows;
  // initial snapshot = events 0..15
  uint32_t last_ev[16];
  if (tracing_) {
    for (int r = 0; r < 16; r++) {
      log_.reg_events.push(zkir_reg_event{reg_[r], bound_[r].payload, bound_[r].max_bits, 0, (uint8_t)r, state_[r], bound_[r].tag, {0, 0, 0, 0, 0}});
      last_ev[r] = r;
    }
  }
  if (log_.rc_offsets.empty()) log_.rc_offsets.push_back(0);
  const uint64_t max_cycles = cfg_.max_cycles, stop_at = stop_at_ < max_cycles ? stop_at_ : max_cycles, base = base_;
  // The loop runs tile by tile: the per-tile bookkeeping (tile index, output capacity) is done once per T rows and the rows
  // themselves are written through raw pointers (pc, instruction word, register events), so the per-instruction work is
  // fetch (pre-decoded image) -> execute -> at most a few 32-byte event stores.
  while (!halted_) {
    if (cycle_ >= max_cycles) { halted_ = true; halt_kind_ = ZKIR_HALT_CYCLE_LIMIT; break; }            // vm.rs:211-214
    if (cycle_ >= stop_at) break;                                       // end of the trace window (or of the fast-forward): not a halt
    uint64_t chunk = T - ((cycle_ - base) & (T - 1));
    if (chunk > stop_at - cycle_) chunk = stop_at - cycle_;
    uint64_t* pc_out = nullptr; uint32_t* inst_out = nullptr; zkir_reg_event* ev_out = nullptr;
    uint32_t ev_base = 0;
    if (tracing_) {
      if (cycle_ - base + chunk >= 0xFFFFFFF0ull || log_.reg_events.size() + 4 * chunk >= 0xFFFFFFC0ull)       // event / row indices are 32-bit (tile index, vis)
        return {ZKIR_ERR_OTHER, "trace longer than 2^32-16 rows or 2^32-64 register events is not supported"};
      if (((cycle_ - base) & (T - 1)) == 0) {                           // tile index: events visible at the tile's first row
        if (progress_ && progress_->stable.load(std::memory_order_relaxed) &&
            (log_.reg_events.size() + 4 * chunk > log_.reg_events.capacity() || log_.tile_ev_off.size() + 2 > log_.tile_ev_off.capacity() ||
             log_.pc.size() + chunk > log_.pc.capacity())) {            // a buffer is about to move: the consumer must let go of the log first
          progress_->stable.store(false);                               // (sequentially consistent: paired with the consumer's idle / stable handshake)
          while (!progress_->consumer_idle.load()) _mm_pause();
        }
        log_.tile_ev_off.push_back((uint32_t)log_.reg_events.size());
        for (int r = 0; r < 16; r++) log_.tile_snap.push_back(last_ev[r]);
        if (progress_ && (log_.tile_ev_off.size() & 511) == 1 && progress_->stable.load(std::memory_order_relaxed)) {   // every 512 tiles
          _mm_sfence();                                                 // the streaming stores of the finished tiles are visible
          const uint64_t t = log_.tile_ev_off.size() - 1;
          progress_->events.store(log_.reg_events.size(), std::memory_order_relaxed);
          progress_->rows.store(cycle_ - base, std::memory_order_relaxed);
          progress_->tiles.store(t, std::memory_order_release);
        }
      }
      pc_out = log_.pc.grow(chunk); inst_out = log_.inst.grow(chunk);
      ev_out = log_.reg_events.grow(4 * chunk);                         // at most 3 registers change per instruction (deferred-mode normalisations + rd)
      ev_base = (uint32_t)log_.reg_events.size();
    }
    uint64_t rows = 0; uint32_t n_ev = 0;
    for (; rows < chunk; rows++) {
      fetch_pc_ = pc_;
      uint32_t word; Decoded d;
      const uint64_t off = pc_ - 0x1000;
      if (__builtin_expect(off < code_bytes_ && !(off & 3), 1)) {       // inside the loaded image: pre-decoded
        word = words_[off >> 2]; d = dec_[off >> 2];
        if (__builtin_expect(d.op == 0xFF, 0)) { char m[64]; snprintf(m, sizeof m, "Decode error: Unknown opcode: 0x%02X", word & 0x7F); return {ZKIR_ERR_DECODE, m}; }
      } else {
        if (pc_ & 3) return {ZKIR_ERR_OTHER, "Misaligned PC: " + hexs(pc_)};                          // vm.rs:364-369
        word = mem_.load<uint32_t>(pc_);                                                               // the fetch is never part of a row (vm.rs:295)
        ICacheEntry& ic = icache_[(pc_ >> 2) & (NICACHE - 1)];
        if (ic.pc != pc_ || ic.word != word) {
          Decoded nd;
          if (!decode_word(word, nd)) { char m[64]; snprintf(m, sizeof m, "Decode error: Unknown opcode: 0x%02X", word & 0x7F); return {ZKIR_ERR_DECODE, m}; }
          ic.pc = pc_; ic.word = word; ic.d = nd;
        }
        d = ic.d;
      }
      if (tracing_) {                                                   // row pre-state is implied by the events so far (vm.rs:245-253)
        // streaming stores: the logs are written once, front to back, and read next by the DMA engine — keeping them out of the
        // cache hierarchy avoids the read-for-ownership of every line (the traced loop is store-bandwidth-bound otherwise)
        _mm_stream_si64((long long*)(pc_out + rows), (long long)fetch_pc_);
        _mm_stream_si32((int*)(inst_out + rows), (int)word);
      }
      dirty_ = 0;
      if (!(deferred_ ? step_deferred(d) : step(d))) return err_;
      if (d.op == 0x50 && !syscall()) return err_;                      // vm.rs:277-279
      if (tracing_ && dirty_) {
        uint32_t m = dirty_;
        do {
          const int r = __builtin_ctz(m); m &= m - 1;
          last_ev[r] = ev_base + n_ev;
          const zkir_reg_event e{reg_[r], bound_[r].payload, bound_[r].max_bits, (uint32_t)(cycle_ - base + 1), (uint8_t)r, state_[r], bound_[r].tag, {0, 0, 0, 0, 0}};
          __m128i lo, hi;
          memcpy(&lo, &e, 16); memcpy(&hi, (const char*)&e + 16, 16);
          _mm_stream_si128((__m128i*)(ev_out + n_ev), lo); _mm_stream_si128((__m128i*)(ev_out + n_ev) + 1, hi);   // 32-byte events on 16-byte aligned storage
          n_ev++;
        } while (m);
      }
      if (range_) {                                                     // vm.rs:316-344
        bool cp = (d.op >= 0x38 && d.op <= 0x3B) || (d.op >= 0x40 && d.op <= 0x45) || d.op == 0x48 || d.op == 0x49 || (d.op >= 0x04 && d.op <= 0x07);
        if (!cp && !pending_.empty()) {                                 // should_checkpoint, range_check.rs:122-135
          if (pending_.size() >= 16) cp = true;
          else for (const PendingCheck& p : pending_) if (p.max_bits >= data_bits_ + 4) { cp = true; break; }
        }
        if (cp) flush_range_checks();
      }
      cycle_++;                                                         // vm.rs:347
      if (halted_) { rows++; break; }
    }
    if (tracing_) { log_.pc.commit(rows); log_.inst.commit(rows); log_.reg_events.commit(n_ev); }
  }
  if (tracing_) _mm_sfence();                                           // streaming stores globally visible before the log is handed on
  if (tracing_) log_.tile_ev_off.push_back((uint32_t)log_.reg_events.size());
  finish_log(tracing_);
  return {};
}

}  // namespace

Status interpret(const uint8_t* blob, size_t len, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config& cfg, uint32_t tile_rows, DeltaLog& log,
                 Progress* progress, uint64_t win_begin, uint64_t win_end) {
  if (win_begin > win_end) return {ZKIR_ERR_ARGUMENT, "trace window: row_begin > row_end"};
  ProgramView pv;
  Status st = parse_program(blob, len, pv);
  if (!st.ok()) return st;
  if (pv.entry_point < 0x1000) {                                      // vm.rs:141-147 panics; reported as an error here
    return {ZKIR_ERR_BAD_PROGRAM, "Program appears to be in debug format (entry_point=" + hexs(pv.entry_point) + "). Use release format (zkir-llvm without --debug) for execution."};
  }
  // rows the log can hold at most: the window, cut by max_cycles
  const uint64_t hi = win_end < cfg.max_cycles ? win_end : cfg.max_cycles, cap_rows = hi > win_begin ? hi - win_begin : 0;
  if (tile_rows == 0) tile_rows = cap_rows <= (1ull << 21) ? 256u : 512u;         // measured best on MI355X (profiles/r01_sweep_trace_fill.txt)
  if (tile_rows < 256 || tile_rows > 2048 || (tile_rows & (tile_rows - 1))) return {ZKIR_ERR_ARGUMENT, "tile_rows must be a power of two in 256..2048"};   // K1 instantiations (trace_fill.hip); 4096 would need 262 KB of LDS
  log.tile_rows = tile_rows;
  if (cfg.enable_execution_trace && cap_rows <= (1ull << 28)) {
    log.pc.reserve(cap_rows); log.inst.reserve(cap_rows);
    // with a streaming consumer the event log and the tile index must not move while it reads them: room for 1.25 events per row
    // (the fib loop writes 0.8) and for every tile; a run that needs more makes the consumer give up streaming (Progress::stable)
    log.reg_events.reserve(progress ? cap_rows + cap_rows / 4 + 4096 : cap_rows + 16);
    if (progress) { log.tile_ev_off.reserve(cap_rows / tile_rows + 4); log.tile_snap.reserve((cap_rows / tile_rows + 4) * 16); }
  }
  std::unique_ptr<Machine> m(new Machine(pv, inputs, n_inputs, cfg, log, progress));   // ~170 KB (icache): heap, released on every path
  m->set_window(win_begin, win_end);
  return m->run();
}

}  // namespace zkir
