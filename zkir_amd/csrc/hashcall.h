// hashcall.h — (AIR mode 4) the hash syscalls of a run as a TAPE: what the prover's host side (stark_prove.inl), the host witness (zkir_memcheck_witness_of_mode) and the
// verifier (verify.cpp) share.  Host only.  Spec: oracle/stark_oracle.cpp "MODE 4 (b)"; reference semantics: zkir-runtime/src/syscall.rs:121-171, crypto.rs:223-395.
//
// A record = (cycle, input pointer, input length, output pointer, kind 3 / 5 / 6) + per aligned 8-byte cell the call touches — the cells under [in, in + len) and
// [out, out + 32), ascending, each once — the cell's bytes BEFORE the call and the time of its previous access.  In a proof: [n] then per call [cycle] [in: two 20-bit limbs]
// [len] [out: two limbs] [kind] [touched cells] and per cell [time] [bytes: four 16-bit pieces].
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

#include "host.h"

namespace hashcall {

constexpr uint64_t MAX_LEN = 1u << 20;                           // a proof states hash calls of up to 1 MiB of input
struct Cell { uint64_t addr, bytes; uint32_t t; };
struct Call { uint64_t cycle, in_ptr, len, out_ptr; uint32_t kind; std::vector<Cell> cells; };

inline bool in_range(uint64_t in_ptr, uint64_t len, uint64_t out_ptr, uint32_t kind) {
  return (kind == 3 || kind == 5 || kind == 6) && len <= MAX_LEN && in_ptr < (1ull << 40) && in_ptr + len <= (1ull << 40) && out_ptr < (1ull << 40) && out_ptr + 32 <= (1ull << 40);
}
inline void cells_of(uint64_t in_ptr, uint64_t len, uint64_t out_ptr, std::vector<uint64_t>& addrs) {
  addrs.clear();
  if (len) for (uint64_t a = in_ptr & ~7ull; a < in_ptr + len; a += 8) addrs.push_back(a);
  for (uint64_t a = out_ptr & ~7ull; a < out_ptr + 32; a += 8) addrs.push_back(a);
  std::sort(addrs.begin(), addrs.end());
  addrs.erase(std::unique(addrs.begin(), addrs.end()), addrs.end());
}
// the 32 bytes the syscall leaves at out .. out + 32: SHA-256 writes its eight big-endian-parsed words with write_u32 (little-endian), crypto.rs:251-254; the other two the digest's bytes in order
inline void output_bytes(uint32_t kind, const uint8_t* msg, size_t len, uint8_t out[32]) {
  if (kind == 3) { uint32_t h[8]; zkir::sha256(msg, len, h); for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(h[i] >> (8 * k)); }
  else if (kind == 5) zkir::keccak256(msg, len, out);
  else zkir::blake3(msg, len, out);
}
// the bytes of every touched cell AFTER the call (the message is read out of the OLD bytes: reads come first, crypto.rs:232-235)
inline void new_bytes(const Call& c, std::vector<uint64_t>& nb) {
  auto index_of = [&](uint64_t cell) { size_t lo = 0, hi = c.cells.size(); while (lo + 1 < hi) { const size_t m = (lo + hi) / 2; if (c.cells[m].addr <= cell) lo = m; else hi = m; } return lo; };
  std::vector<uint8_t> msg((size_t)c.len);
  for (uint64_t k = 0; k < c.len; k++) { const uint64_t a = c.in_ptr + k; msg[(size_t)k] = (uint8_t)(c.cells[index_of(a & ~7ull)].bytes >> (8 * (a & 7))); }
  uint8_t d[32];
  output_bytes(c.kind, msg.data(), msg.size(), d);
  nb.resize(c.cells.size());
  for (size_t i = 0; i < c.cells.size(); i++) nb[i] = c.cells[i].bytes;
  for (int k = 0; k < 32; k++) { const uint64_t a = c.out_ptr + k; const size_t i = index_of(a & ~7ull); const int sh = 8 * (int)(a & 7); nb[i] = (nb[i] & ~(0xFFull << sh)) | ((uint64_t)d[k] << sh); }
}
inline void put_section(const std::vector<Call>& calls, std::vector<uint32_t>& w) {
  w.push_back((uint32_t)calls.size());
  for (const Call& c : calls) {
    w.push_back((uint32_t)c.cycle); w.push_back((uint32_t)(c.in_ptr & 0xFFFFF)); w.push_back((uint32_t)(c.in_ptr >> 20)); w.push_back((uint32_t)c.len);
    w.push_back((uint32_t)(c.out_ptr & 0xFFFFF)); w.push_back((uint32_t)(c.out_ptr >> 20)); w.push_back(c.kind); w.push_back((uint32_t)c.cells.size());
    for (const Cell& x : c.cells) { w.push_back(x.t); for (int i = 0; i < 4; i++) w.push_back((uint32_t)((x.bytes >> (16 * i)) & 0xFFFF)); }
  }
}
// Parses AND checks a hash section (the verifier's checks; the prover runs them on what it is given): 0 = well-formed, 4 = truncated, 55 = a call's output lands on code
// bytes, 56 = a malformed record (ranges, order, cell count, a piece above 16 bits, a previous access that is not before the call).  *words_used = the section's length.
inline int parse_section(const uint32_t* w, size_t avail, uint64_t n_real, uint64_t code_end, std::vector<Call>& calls, size_t* words_used) {
  calls.clear();
  if (avail < 1) return 4;
  const size_t nh = w[0];
  if (nh > n_real) return 56;
  size_t q = 1;
  std::vector<uint64_t> addrs;
  calls.reserve(std::min(nh, avail / 28));                      // (a record is at least 8 + 4 x 5 words: a forged count cannot make the parser allocate beyond what the proof holds)
  for (size_t k = 0; k < nh; k++) {
    if (q + 8 > avail) return 4;
    const uint32_t* c = w + q;
    calls.emplace_back();
    Call& hc = calls[k];
    if (c[1] >= (1u << 20) || c[2] >= (1u << 20) || c[4] >= (1u << 20) || c[5] >= (1u << 20)) return 56;
    hc.cycle = c[0]; hc.in_ptr = (uint64_t)c[1] | ((uint64_t)c[2] << 20); hc.len = c[3]; hc.out_ptr = (uint64_t)c[4] | ((uint64_t)c[5] << 20); hc.kind = c[6];
    if (hc.cycle >= n_real || (k && hc.cycle <= calls[k - 1].cycle) || !in_range(hc.in_ptr, hc.len, hc.out_ptr, hc.kind)) return 56;
    if (hc.out_ptr < code_end && hc.out_ptr + 32 > 0x1000) return 55;
    cells_of(hc.in_ptr, hc.len, hc.out_ptr, addrs);
    if (c[7] != addrs.size()) return 56;
    if (q + 8 + 5 * addrs.size() > avail) return 4;
    q += 8;
    hc.cells.resize(addrs.size());
    for (size_t j = 0; j < addrs.size(); j++, q += 5) {
      uint64_t bytes = 0;
      for (int i = 0; i < 4; i++) { if (w[q + 1 + i] > 0xFFFF) return 56; bytes |= (uint64_t)w[q + 1 + i] << (16 * i); }
      if (w[q] > hc.cycle) return 56;                              // the time read (the previous access's cycle + 1) is smaller than the time written (cycle + 1)
      hc.cells[j] = Cell{addrs[j], bytes, w[q]};
    }
  }
  *words_used = q;
  return 0;
}

// The per-call work of the prover's and the verifier's table side (a digest, ~17 fingerprints and their share of a batch inversion per call) split over host threads:
// fn(part, first_call, last_call) for `parts` contiguous ranges; the caller sums the parts.  A thread that cannot be started runs on the calling thread instead.
template <class F>
inline void for_calls(size_t n_calls, unsigned parts, F&& fn) {
  if (parts <= 1 || n_calls < 2) { fn(0u, (size_t)0, n_calls); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < parts; t++) {
    const size_t lo = n_calls * t / parts, hi = n_calls * (t + 1) / parts;
    if (t + 1 == parts) { fn(t, lo, hi); break; }
    try { th.emplace_back([&fn, t, lo, hi] { fn(t, lo, hi); }); } catch (...) { fn(t, lo, hi); }
  }
  for (auto& x : th) x.join();
}
inline unsigned parts_for(size_t n_calls) {
  unsigned hw = std::thread::hardware_concurrency() / 2;
  if (hw < 1) hw = 1;
  if (hw > 32) hw = 32;                                         // (half the logical cores, at most 32: a 2^22-cycle hash chain is 2.4 s of table-side work on one)
  const size_t by_work = n_calls / 512 + 1;                      // (a part should be worth a thread's start)
  return (unsigned)std::min<size_t>(hw, by_work);
}

}  // namespace hashcall
