// poseidon2.h — Poseidon2 permutation over Baby Bear, width 12, S-box x^7, R_F = 8, R_P = 22 (host + device).
//
// Self-defined instance ("Poseidon2-12" of BASELINE.json:north_star; the reference has no Poseidon2 — its syscall is a
// stub that always errors, zkir-runtime/src/crypto.rs:306-315).  Structure follows the Poseidon2 paper: initial external
// linear layer, 4 full rounds, 22 partial rounds, 4 full rounds; external matrix circ(2*M4, M4, M4) with
// M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]; internal matrix = all-ones + diag(-2, 1, 2, 4, ..., 1024).
// Round constants: SplitMix64 seeded with the ASCII bytes "ZKIR-P2-", top 31 bits of each output, rejection-sampled below p,
// 96 external then 22 internal.  A demonstrator instance (rate 8 / capacity 4 => ~62-bit collision resistance), not a vetted one.
//
// All state words handled here are in MONTGOMERY form.  The function computed is exactly the textbook permutation (the oracle,
// oracle/stark_oracle.cpp, implements it naively and every kernel is compared with it bit for bit); what is specific to gfx950
// is HOW: the hash kernels sit at ~100 % VALU issue utilisation (profiles/*_valu_busy.txt), so the formulation minimises the
// instruction count —
//   * linear layers in 64 bits without intermediate reductions (v_lshl_add_u64 / v_mad_u64_u32 are single full-rate
//     instructions), one Barrett reduction per output;
//   * Montgomery products without their final conditional subtraction wherever the bounds allow ("lazy", see babybear.h);
//   * additions folded into the 64-bit addend of a Montgomery product: (a*b + c) / R = mont(a, b) + c / R costs nothing extra.
//     The partial rounds add the column sum that way, the full rounds add the NEXT round's constants pulled back through the
//     linear layer (pre[r] = M_ext^-1 * ext[r + 1], computed once by generate()).
#pragma once
#include "babybear.h"

#ifndef P2_RF_UNROLL
#define P2_RF_UNROLL 1
#endif
namespace p2 {

constexpr int T = 12, RF = 8, RP = 22, RATE = 8, DIGEST = 4;

struct Consts {              // Montgomery form unless noted
  uint32_t ext[RF][T];
  uint32_t in[RP];
  uint32_t diag[T];
  uint32_t pre[RF][T];       // (M_ext^-1 * ext[r + 1]) * R^2 mod p: addend of the last S-box product of full round r; zero for r = 3, 7
  // permute_scaled(): the same constants carried by the factor the state has at that point (canonical values)
  uint32_t ext0_s[T];        // ext[0] * F_IN
  uint64_t pre_b[RF][T];     // p * 2^32 + (M_ext^-1 * ext[r + 1]) * h_r * R, h_r = factor of the state after the S-boxes of full round r: the 64-bit
                             // addend of the last S-box product (constant + the bias that makes the signed result an unsigned word)
  uint32_t in_scale;         // mont_mul(x, in_scale) = x * F_IN: canonical value -> input word of permute_scaled
  uint32_t out_scale;        // mont_mul(s, out_scale) = s / F_OUT: output word -> canonical value
  uint32_t carry;            // mont_mul(s, carry) = s * F_IN / F_OUT: output word -> input word of the next permutation (sponge capacity)
  uint32_t r3;               // R^3 mod p
  uint32_t diag0;            // diag[0] = -2 in Montgomery form
  uint32_t in_r[RP + 1];     // in[r] * R (Montgomery form of the Montgomery form): the NEXT partial round's constant rides the addend of
                             // word 0's update product; in_r[RP] = ext[RF/2][0] * R (the constant of the full round that follows)
  uint32_t ext4_r[T];        // ext[RF/2][i] * R: rides the addends of the LAST partial round's update products
  int32_t in0_neg;           // in[0] - p (in (-p, 0]): word 0 enters the partial rounds as a signed residue
};

inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// inverse of the external matrix circ(2*M4, M4, M4) over F_p (canonical entries), Gauss-Jordan; cold path
inline void ext_matrix_inverse(uint32_t inv[T][T]) {
  static const uint32_t M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
  uint32_t a[T][2 * T];
  for (int i = 0; i < T; i++)
    for (int j = 0; j < T; j++) {
      a[i][j] = M4[i & 3][j & 3] * ((i >> 2) == (j >> 2) ? 2u : 1u);
      a[i][T + j] = i == j;
    }
  for (int col = 0; col < T; col++) {
    int piv = col;
    while (a[piv][col] == 0) piv++;                             // the matrix is invertible (MDS-derived), a pivot exists
    for (int j = 0; j < 2 * T; j++) { const uint32_t t = a[col][j]; a[col][j] = a[piv][j]; a[piv][j] = t; }
    const uint32_t pinv = bb::inv(a[col][col]);
    for (int j = 0; j < 2 * T; j++) a[col][j] = bb::mul(a[col][j], pinv);
    for (int i = 0; i < T; i++) {
      if (i == col || a[i][col] == 0) continue;
      const uint32_t f = a[i][col];
      for (int j = 0; j < 2 * T; j++) a[i][j] = bb::sub(a[i][j], bb::mul(f, a[col][j]));
    }
  }
  for (int i = 0; i < T; i++) for (int j = 0; j < T; j++) inv[i][j] = a[i][T + j];
}

inline void generate(Consts& c) {
  uint64_t s = 0x5A4B49522D50322Dull;
  auto next = [&]() -> uint32_t { for (;;) { const uint32_t v = (uint32_t)(splitmix64(s) >> 33); if (v < bb::P) return bb::to_mont(v); } };
  for (int r = 0; r < RF; r++) for (int i = 0; i < T; i++) c.ext[r][i] = next();
  for (int r = 0; r < RP; r++) c.in[r] = next();
  c.diag[0] = bb::to_mont(bb::P - 2);
  for (int i = 1; i < T; i++) c.diag[i] = bb::to_mont(1u << (i - 1));
  uint32_t minv[T][T];
  ext_matrix_inverse(minv);
  for (int r = 0; r < RF; r++)
    for (int i = 0; i < T; i++) {
      uint32_t v = 0;
      if (r != RF / 2 - 1 && r != RF - 1)
        for (int j = 0; j < T; j++) v = bb::add(v, bb::mul(minv[i][j], bb::from_mont(c.ext[r + 1][j])));
      c.pre[r][i] = bb::to_mont(bb::to_mont(v));
    }
  // ---- permute_scaled(): factors.  A state word holds f * v for the true value v.  S-boxes in Montgomery arithmetic take the
  // factor f to f^7 / R^6, a linear layer keeps it, its wide Montgomery reduction divides it by R.  The 22 partial rounds S-box
  // one word only, so they want the factor that S-boxes preserve, f = R; working backwards from there through full rounds 3..0
  // and the initial layer fixes F_IN, forwards through rounds 4..7 gives F_OUT.  (x -> x^7 is a bijection: gcd(7, p - 1) = 1.)
  {
    constexpr uint64_t INV7 = 1725656503ull;                   // 7^-1 mod (p - 1)
    const uint32_t R = bb::R1, Rinv = bb::inv(R);
    const uint32_t R6 = bb::pow(R, 6);
    uint32_t g[RF + 1], h[RF];                                 // g[r]: factor entering the S-boxes of full round r; h[r]: after them
    g[RF / 2] = R;
    for (int r = RF / 2 - 1; r >= 0; r--) { h[r] = bb::mul(g[r + 1], R); g[r] = bb::pow(bb::mul(h[r], R6), INV7); }
    for (int r = RF / 2; r < RF; r++) { h[r] = bb::mul(bb::pow(g[r], 7), bb::inv(R6)); g[r + 1] = bb::mul(h[r], Rinv); }
    const uint32_t f_in = bb::mul(g[0], R), f_out = g[RF];
    for (int i = 0; i < T; i++) c.ext0_s[i] = bb::mul(bb::from_mont(c.ext[0][i]), f_in);
    for (int r = 0; r < RF; r++)
      for (int i = 0; i < T; i++) {
        uint32_t v = 0;
        if (r != RF / 2 - 1 && r != RF - 1)
          for (int j = 0; j < T; j++) v = bb::add(v, bb::mul(minv[i][j], bb::from_mont(c.ext[r + 1][j])));
        c.pre_b[r][i] = ((uint64_t)bb::P << 32) | bb::mul(bb::mul(v, h[r]), R);
      }
    c.in_scale = bb::mul(f_in, R);
    c.out_scale = bb::mul(R, bb::inv(f_out));
    c.carry = bb::mul(bb::mul(f_in, R), bb::inv(f_out));
    c.r3 = bb::pow(R, 3);
    c.diag0 = c.diag[0];
    for (int r = 0; r < RP; r++) c.in_r[r] = bb::to_mont(c.in[r]);
    c.in_r[RP] = bb::to_mont(c.ext[RF / 2][0]);
    for (int i = 0; i < T; i++) c.ext4_r[i] = bb::to_mont(c.ext[RF / 2][i]);
    c.in0_neg = (int32_t)c.in[0] - (int32_t)bb::P;
  }
}

// x^7 for canonical x; only x^3 needs its reduction (it is squared), the other products stay within the lazy bounds of
// bb::mont_mul_lazy: x2 < 1.469p, x3 < p, x6 < 1.469p, result < 1.689p.  `addend` (canonical, = v * R^2) adds v * R for free.
BB_HD uint32_t sbox_lazy(uint32_t x, uint32_t addend = 0) {
  const uint32_t x2 = bb::mont_mul_lazy(x, x), x3 = bb::mont_mul(x2, x), x6 = bb::mont_mul_lazy(x3, x3);
  return bb::mont_mul_add_lazy(x6, x, addend);
}
BB_HD uint32_t sbox(uint32_t x) { return bb::reduce_2p(sbox_lazy(x)); }

// M4 * (a,b,c,d) = (5a+7b+c+3d, 4a+6b+c+d, a+3b+5c+7d, a+b+4c+6d) with 8 additions and 4 shifts (the evaluation order of the
// Poseidon2 paper, appendix B), WITHOUT reductions, in 64 bits (outputs below 16 max(a,b,c,d)): on gfx950 every line is one
// full-rate v_lshl_add_u64 / v_mad_u64_u32, against three instructions for a modular addition.  Inputs may be lazy.
BB_HD void m4_wide(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint64_t* y) {
  const uint64_t t0 = bb::acc_add(a, b), t1 = bb::acc_add(c, d);
  const uint64_t t2 = bb::mad_wide<2>(t1, b), t3 = bb::mad_wide<2>(t0, d);
  const uint64_t t4 = (t1 << 2) + t3, t5 = (t0 << 2) + t2;                        // a + b + 4c + 6d ; 4a + 6b + c + d
  y[0] = t3 + t5; y[1] = t5; y[2] = t2 + t4; y[3] = t4;
}
// External linear layer circ(2*M4, M4, M4), optionally followed by the addition of `rc`; inputs below 1.689p + 1 (sbox_lazy),
// canonical out.  Every output is below 64 * 1.69p + p < 2^38 before its single reduction.
template <bool ADD_RC>
BB_HD void ext_linear(uint32_t* s, const uint32_t* rc) {
  uint64_t y[T];
  m4_wide(s[0], s[1], s[2], s[3], y); m4_wide(s[4], s[5], s[6], s[7], y + 4); m4_wide(s[8], s[9], s[10], s[11], y + 8);
  uint64_t sum[4];
#pragma unroll
  for (int j = 0; j < 4; j++) sum[j] = y[j] + y[4 + j] + y[8 + j];
#pragma unroll
  for (int k = 0; k < T; k++) {
    uint64_t v = y[k] + sum[k & 3];
    if (ADD_RC) v += rc[k];
    s[k] = bb::reduce_wide<6>(v);
  }
}
// The 22 partial rounds: s[0] <- sbox(s[0] + rc); s[i] <- sum(s) + diag[i] * s[i] with diag = (-2, 1, 2, 4, ..., 1024).
// s[1..11] stay LAZY (below 2p, unreduced) between rounds: they only feed the 64-bit sum and one Montgomery multiplication
// by a canonical constant, and that multiplication adds the sum on its way: (s[i] * diag[i] + sum * R) / R.  s[0] is canonical.
BB_HD void int_rounds(uint32_t* s, const Consts& c) {
#pragma unroll 1
  for (int r = 0; r < RP; r++) {
    const uint32_t s0 = sbox(bb::add(s[0], c.in[r]));
    uint64_t acc = s0;                                         // < p + 11 * 2p < 2^36
#pragma unroll
    for (int i = 1; i < T; i++) acc = bb::acc_add(acc, s[i]);
    const uint32_t sum = bb::reduce_wide<4>(acc);
    const uint32_t sum_r = bb::mont_mul_lazy(sum, bb::R2);     // sum * R mod p, below 1.469p
    s[0] = bb::sub(sum, bb::dbl(s0));
#pragma unroll
    for (int i = 1; i < T; i++) s[i] = bb::mont_mul_add_lazy(s[i], c.diag[i], sum_r);      // < 1.9375p + 1
  }
#pragma unroll
  for (int i = 1; i < T; i++) s[i] = bb::reduce_2p(s[i]);
}
BB_HD void permute(uint32_t* s, const Consts& c) {
  ext_linear<true>(s, c.ext[0]);
#pragma unroll 1
  for (int r = 0; r < RF; r++) {
    if (r == RF / 2) {
      int_rounds(s, c);
#pragma unroll
      for (int i = 0; i < T; i++) s[i] = bb::add(s[i], c.ext[RF / 2][i]);
    }
#pragma unroll
    for (int i = 0; i < T; i++) s[i] = sbox_lazy(s[i], c.pre[r][i]);
    ext_linear<false>(s, nullptr);
  }
}

// ---- the same permutation with SCALED state words (throughput variant) ------------------------------------------------------------
// The 108 outputs of the nine external linear layers and the 22 partial-round sums are reduced with bb::mont_reduce_wide (2
// instructions) instead of the Barrett reduce_wide (6): that divides the state by R each time, so a state word holds f * v with a
// factor f that changes from layer to layer in a fixed, data-independent way; generate() pre-multiplies every round constant by the
// factor of the place it is added at and chooses the input factor so that the partial rounds see f = R (plain Montgomery form).
//   in : s[i] = mont_mul_lazy(x_i, c.in_scale) for canonical x_i, or mont_mul_lazy(previous output word, c.carry); below 1.96p
//   out: s[i] = F_OUT * v_i below p + 64; canonical v_i = mont_mul(s[i], c.out_scale)
// The function computed on the x_i / v_i is exactly permute()'s (tests/test_stark_oracle.py, test_gpu_stark.py: bit for bit).
template <bool ADD_RC>
BB_HD void ext_linear_scaled(uint32_t* s, const uint32_t* rc) {
  uint64_t y[T];
  m4_wide(s[0], s[1], s[2], s[3], y); m4_wide(s[4], s[5], s[6], s[7], y + 4); m4_wide(s[8], s[9], s[10], s[11], y + 8);
  uint64_t sum[4];
#pragma unroll
  for (int j = 0; j < 4; j++) sum[j] = y[j] + y[4 + j] + y[8 + j];
#pragma unroll
  for (int k = 0; k < T; k++) {
    uint64_t v = y[k] + sum[k & 3];
    if (ADD_RC) v += rc[k];
    s[k] = bb::mont_reduce_wide(v);                            // < 2^38 / 2^32 + p
  }
}
// S-boxes in SIGNED Montgomery arithmetic (bb::smont_mul): inputs are int32 residues with |x| <= p + 128 — the "canonical + a little"
// outputs of ext_linear_scaled read as signed words, or the (-p, p) words the partial rounds leave — and no product of the chain needs
// a correction (every intermediate stays within 0.97 p).  The last product takes a 64-bit `addend`: v * R^2 (adds v * R) and, in the
// full rounds, the bias p * 2^32 (one constant, Consts::pre_b): its result r + p lies in (0.04 p, 1.96 p) and is handed to the 64-bit linear layer as an UNSIGNED word
// (zero-extension is a register of zeros; a signed word would need a shift per operand).
BB_HD uint32_t sbox_biased(uint32_t xu, uint64_t addend) {
  const int32_t x = (int32_t)xu;
  const int32_t x2 = bb::smont_mul(x, x), x3 = bb::smont_mul(x2, x), x6 = bb::smont_mul(x3, x3);
  return (uint32_t)bb::smont_mul_add(x6, x, addend);
}
BB_HD int32_t sbox_signed(int32_t x) {
  const int32_t x2 = bb::smont_mul(x, x), x3 = bb::smont_mul(x2, x), x6 = bb::smont_mul(x3, x3);
  return bb::smont_mul(x6, x);
}
// The 22 partial rounds on signed residues in (-p, p): state factor R (plain Montgomery form) on entry and exit, and NOTHING is
// reduced anywhere — a signed product of a word below p with a constant below p is within 0.97 p again, the 64-bit column sum takes
// its operands sign-extended by the multiply-add that accumulates them.  The constant of the full round that follows rides the
// addends of the last round's update products (LAST), so the words leave ready for its S-boxes.
template <bool LAST>
BB_HD void int_round_signed(int32_t& x, int32_t* t, const Consts& c, int r) {
  const int32_t s0 = sbox_signed(x);
  int64_t acc = s0;
#pragma unroll
  for (int i = 1; i < T; i++) acc = bb::sacc_add(acc, t[i]);
  const int64_t sum_r = bb::smont_mul(bb::smont_reduce_wide(acc), (int32_t)c.r3);                // (sum / R) * R^3 / R = sum * R
  x = bb::smont_mul_add(s0, (int32_t)c.diag0, (uint64_t)(sum_r + c.in_r[r + 1]));               // sum - 2 s0 + the next constant: the next S-box input
#pragma unroll
  for (int i = 1; i < T; i++) t[i] = bb::smont_mul_add(t[i], (int32_t)c.diag[i], (uint64_t)(LAST ? sum_r + c.ext4_r[i] : sum_r));
}
BB_HD void int_rounds_scaled(uint32_t* s, const Consts& c) {   // in: unsigned words below p + 128; out: signed words in (-p, p), next round's constants added
  int32_t x = (int32_t)s[0] + c.in0_neg;
  int32_t t[T];
#pragma unroll
  for (int i = 1; i < T; i++) t[i] = (int32_t)s[i];
#pragma unroll 1
  for (int r = 0; r < RP - 1; r++) int_round_signed<false>(x, t, c, r);
  int_round_signed<true>(x, t, c, RP - 1);
  s[0] = (uint32_t)x;
#pragma unroll
  for (int i = 1; i < T; i++) s[i] = (uint32_t)t[i];
}
BB_HD void permute_scaled(uint32_t* s, const Consts& c) {
  ext_linear_scaled<true>(s, c.ext0_s);
#pragma unroll P2_RF_UNROLL
  for (int r = 0; r < RF; r++) {
    if (r == RF / 2) int_rounds_scaled(s, c);
#pragma unroll
    for (int i = 0; i < T; i++) s[i] = sbox_biased(s[i], c.pre_b[r][i]);
    ext_linear_scaled<false>(s, nullptr);
  }
}

#if defined(__HIPCC__)
// ---- the same permutation spread over a QUAD of lanes (latency variant) ---------------------------------------------------------
// A lone wave issues one VALU instruction every ~4 cycles, so one permutation per lane (~4250 instructions) takes ~8 us no matter
// how few lanes are busy — and the upper Merkle levels / small FRI layers keep only a few lanes busy.  Here lane l of an aligned
// quad holds state words l, 4 + l, 8 + l (column l of the three M4 blocks): the S-box layer is 3 chains per lane instead of 12, the
// M4 products take the block's four words through DPP quad broadcasts (row l of M4 as per-lane multipliers), the column sums stay
// inside a lane, and the partial-round sum is a 2-step quad butterfly.  ~1800 instructions per lane.  All four lanes of a quad must
// be active.  Same function, same bounds as permute(); which words a lane keeps lazy differs, the canonical results do not.
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ uint64_t quad_perm64(uint64_t v) { return ((uint64_t)quad_perm<CTRL>((uint32_t)(v >> 32)) << 32) | quad_perm<CTRL>((uint32_t)v); }

template <bool ADD_RC>
__device__ __forceinline__ void ext_linear_quad(uint32_t* s, const uint32_t* m4row, const uint32_t* rc) {
  uint64_t y[3];
#pragma unroll
  for (int b = 0; b < 3; b++) {
    const uint32_t v0 = quad_perm<0x00>(s[b]), v1 = quad_perm<0x55>(s[b]), v2 = quad_perm<0xAA>(s[b]), v3 = quad_perm<0xFF>(s[b]);
    y[b] = (uint64_t)v0 * m4row[0] + (uint64_t)v1 * m4row[1] + (uint64_t)v2 * m4row[2] + (uint64_t)v3 * m4row[3];
  }
  const uint64_t sum = y[0] + y[1] + y[2];
#pragma unroll
  for (int b = 0; b < 3; b++) {
    uint64_t v = y[b] + sum;
    if (ADD_RC) v += rc[b];
    s[b] = bb::reduce_wide<6>(v);
  }
}
// s[b] = state word 4 b + l of the quad's permutation, l = lane & 3; Montgomery form, canonical in and out
__device__ __forceinline__ void permute_quad(uint32_t* s, int l, const Consts& c) {
  const uint32_t packed = l == 0 ? 0x03010705u : l == 1 ? 0x01010604u : l == 2 ? 0x07050301u : 0x06040101u;   // row l of M4, one byte per entry
  const uint32_t m4row[4] = {packed & 0xFF, (packed >> 8) & 0xFF, (packed >> 16) & 0xFF, packed >> 24};
  const uint32_t diag[3] = {c.diag[l], c.diag[4 + l], c.diag[8 + l]};
  { const uint32_t rc[3] = {c.ext[0][l], c.ext[0][4 + l], c.ext[0][8 + l]}; ext_linear_quad<true>(s, m4row, rc); }
#pragma unroll 1
  for (int r = 0; r < RF; r++) {
    if (r == RF / 2) {
#pragma unroll 1
      for (int q = 0; q < RP; q++) {
        const uint32_t sb = sbox(bb::add(s[0], c.in[q]));                    // meaningful in lane 0 only (word 0)
        const uint32_t s0 = l == 0 ? sb : s[0];
        uint64_t acc = bb::acc_add(bb::acc_add((uint64_t)s0, s[1]), s[2]);
        acc += quad_perm64<0xB1>(acc);                                       // lanes (1,0,3,2)
        acc += quad_perm64<0x4E>(acc);                                       // lanes (2,3,0,1): every lane now holds the sum of all 12 words
        const uint32_t sum_r = bb::mont_mul_lazy(bb::reduce_wide<4>(acc), bb::R2);
        s[0] = bb::reduce_2p(bb::mont_mul_add_lazy(s0, diag[0], sum_r));     // word 0 must be canonical for the next S-box (diag[0] = -2)
        s[1] = bb::mont_mul_add_lazy(s[1], diag[1], sum_r);
        s[2] = bb::mont_mul_add_lazy(s[2], diag[2], sum_r);
      }
      s[1] = bb::reduce_2p(s[1]); s[2] = bb::reduce_2p(s[2]);
#pragma unroll
      for (int b = 0; b < 3; b++) s[b] = bb::add(s[b], c.ext[RF / 2][4 * b + l]);
    }
#pragma unroll
    for (int b = 0; b < 3; b++) s[b] = sbox_lazy(s[b], c.pre[r][4 * b + l]);
    ext_linear_quad<false>(s, m4row, nullptr);
  }
}
#endif

}  // namespace p2
