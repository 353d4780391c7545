// poseidon2.h — Poseidon2 permutation over Baby Bear, width 12, S-box x^7, R_F = 8, R_P = 22 (host + device).
//
// Self-defined instance ("Poseidon2-12" of BASELINE.json:north_star; the reference has no Poseidon2 — its syscall is a
// stub that always errors, zkir-runtime/src/crypto.rs:306-315).  Structure follows the Poseidon2 paper: initial external
// linear layer, 4 full rounds, 22 partial rounds, 4 full rounds; external matrix circ(2*M4, M4, M4) with
// M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]; internal matrix = all-ones + diag(-2, 1, 2, 4, ..., 1024).
// Round constants: SplitMix64 seeded with the ASCII bytes "ZKIR-P2-", top 31 bits of each output, rejection-sampled below p,
// 96 external then 22 internal.  A demonstrator instance (rate 8 / capacity 4 => ~62-bit collision resistance), not a vetted one.
//
// All state words handled here are in MONTGOMERY form.
#pragma once
#include "babybear.h"

namespace p2 {

constexpr int T = 12, RF = 8, RP = 22, RATE = 8, DIGEST = 4;

struct Consts {              // Montgomery form
  uint32_t ext[RF][T];
  uint32_t in[RP];
  uint32_t diag[T];
};

inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline void generate(Consts& c) {
  uint64_t s = 0x5A4B49522D50322Dull;
  auto next = [&]() -> uint32_t { for (;;) { const uint32_t v = (uint32_t)(splitmix64(s) >> 33); if (v < bb::P) return bb::to_mont(v); } };
  for (int r = 0; r < RF; r++) for (int i = 0; i < T; i++) c.ext[r][i] = next();
  for (int r = 0; r < RP; r++) c.in[r] = next();
  c.diag[0] = bb::to_mont(bb::P - 2);
  for (int i = 1; i < T; i++) c.diag[i] = bb::to_mont(1u << (i - 1));
}

BB_HD uint32_t sbox(uint32_t x) { const uint32_t x2 = bb::mont_mul(x, x), x3 = bb::mont_mul(x2, x), x6 = bb::mont_mul(x3, x3); return bb::mont_mul(x6, x); }

// M4 * (a,b,c,d) with additions only: 5a+7b+c+3d, 4a+6b+c+d, a+3b+5c+7d, a+b+4c+6d
BB_HD void m4(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
  using namespace bb;
  const uint32_t ab = add(a, b), cd = add(c, d);
  const uint32_t a2 = dbl(a), b2 = dbl(b), c2 = dbl(c), d2 = dbl(d);
  const uint32_t a4 = dbl(a2), b4 = dbl(b2), c4 = dbl(c2), d4 = dbl(d2);
  const uint32_t y0 = add(add(add(a4, ab), add(b4, b2)), add(cd, d2));           // 5a + 7b + c + 3d
  const uint32_t y1 = add(add(a4, add(b4, b2)), cd);                              // 4a + 6b + c + d
  const uint32_t y2 = add(add(ab, b2), add(add(c4, cd), add(d4, d2)));            // a + 3b + 5c + 7d
  const uint32_t y3 = add(ab, add(c4, add(d4, d2)));                              // a + b + 4c + 6d
  a = y0; b = y1; c = y2; d = y3;
}
BB_HD void ext_linear(uint32_t* s) {
  m4(s[0], s[1], s[2], s[3]); m4(s[4], s[5], s[6], s[7]); m4(s[8], s[9], s[10], s[11]);
  uint32_t sum[4];
#pragma unroll
  for (int j = 0; j < 4; j++) sum[j] = bb::add(bb::add(s[j], s[4 + j]), s[8 + j]);
#pragma unroll
  for (int k = 0; k < T; k++) s[k] = bb::add(s[k], sum[k & 3]);
}
BB_HD void int_linear(uint32_t* s, const Consts& c) {
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < T; i++) sum = bb::add(sum, s[i]);
#pragma unroll
  for (int i = 0; i < T; i++) s[i] = bb::add(sum, bb::mont_mul(s[i], c.diag[i]));
}
BB_HD void permute(uint32_t* s, const Consts& c) {
  ext_linear(s);
#pragma unroll 1
  for (int r = 0; r < RF / 2; r++) {
#pragma unroll
    for (int i = 0; i < T; i++) s[i] = sbox(bb::add(s[i], c.ext[r][i]));
    ext_linear(s);
  }
#pragma unroll 1
  for (int r = 0; r < RP; r++) { s[0] = sbox(bb::add(s[0], c.in[r])); int_linear(s, c); }
#pragma unroll 1
  for (int r = RF / 2; r < RF; r++) {
#pragma unroll
    for (int i = 0; i < T; i++) s[i] = sbox(bb::add(s[i], c.ext[r][i]));
    ext_linear(s);
  }
}

}  // namespace p2
