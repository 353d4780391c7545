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
//     The full rounds add the NEXT round's constants pulled back through the linear layer that way (pre[r] = M_ext^-1 * ext[r + 1],
//     computed once by generate());
//   * (round 4) the partial rounds' eleven passive words as 64-bit lazy integers, updated by a shift and an add — the diagonal is
//     (1, 2, 4, .., 1024) — and Montgomery-reduced every third round only (int_rounds_scaled: 159 vector instructions per three rounds
//     against 207; the leaf hash 3.77 -> 3.36 ms at 152 x 2^21).
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
  int32_t in0_neg;           // in[0] - p (in (-p, 0]): word 0 enters the partial rounds as a signed residue
  // int_rounds_scaled(): the eleven passive words are 64-bit LAZY integers there, reduced (Montgomery: / R) after every third round, so their factor
  // f_r = R^(1 - floor(r / 3)) drifts; word 0 carries h_r = (f_r R^6)^(1/7), the factor whose S-box output has the factor f_r of the words it is added to
  int32_t pr_c[RP];          // h_(r+1) R^2 / f_r: what the reduced  sum - 2 s0  of round r is multiplied by on its way to the next S-box (r = RP - 1: towards G = f_(RP-1) / R)
  uint64_t pr_a[RP];         // h_(r+1) in[r + 1] R: the next round's constant, the addend of that product (r = RP - 1: G ext[RF/2][0] R)
  uint64_t pr_k[T];          // f_(RP-1) ext[RF/2][i]: the constants of the full round that follows, added to the passive words before their last reduction
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
  // factor f to f^7 / R^6, a linear layer keeps it, its wide Montgomery reduction divides it by R.  The 22 partial rounds are
  // ENTERED with the factor that S-boxes preserve, f = R; working backwards from there through full rounds 3..0 and the initial
  // layer fixes F_IN.  Inside them the passive words' factor drifts by 1 / R per reduction (eight of them) and word 0 follows with
  // the factor whose S-box output matches (below); they are left with the common factor G = R^-7, and forwards from G through
  // rounds 4..7 gives F_OUT.  (x -> x^7 is a bijection: gcd(7, p - 1) = 1.)
  {
    constexpr uint64_t INV7 = 1725656503ull;                   // 7^-1 mod (p - 1)
    const uint32_t R = bb::R1, Rinv = bb::inv(R);
    const uint32_t R6 = bb::pow(R, 6);
    uint32_t g[RF + 1], h[RF];                                 // g[r]: factor entering the S-boxes of full round r; h[r]: after them
    g[RF / 2] = R;
    for (int r = RF / 2 - 1; r >= 0; r--) { h[r] = bb::mul(g[r + 1], R); g[r] = bb::pow(bb::mul(h[r], R6), INV7); }
    {
      // the partial rounds (int_rounds_scaled): f[r] the passive words' factor during round r, hx[r] word 0's at its S-box, G the common factor they leave with
      uint32_t f[RP], hx[RP];
      for (int r = 0; r < RP; r++) { f[r] = r < 3 ? R : bb::mul(f[r - 3], Rinv); hx[r] = bb::pow(bb::mul(f[r], R6), INV7); }
      const uint32_t G = bb::mul(f[RP - 1], Rinv), R2c = bb::mul(R, R);
      for (int r = 0; r < RP; r++) {
        const uint32_t hn = r + 1 < RP ? hx[r + 1] : G, rcn = bb::from_mont(r + 1 < RP ? c.in[r + 1] : c.ext[RF / 2][0]);
        c.pr_c[r] = (int32_t)bb::mul(bb::mul(hn, R2c), bb::inv(f[r]));
        c.pr_a[r] = bb::mul(bb::mul(hn, rcn), R);
      }
      c.pr_k[0] = 0;
      for (int i = 1; i < T; i++) c.pr_k[i] = bb::mul(f[RP - 1], bb::from_mont(c.ext[RF / 2][i]));
      g[RF / 2] = G;                                           // .. and the second half of the full rounds starts from it
    }
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
#if defined(P2_ZEXT_MAD) && defined(__HIP_DEVICE_COMPILE__)
  // EXPERIMENT (profiles/r05_p2_mov_variants.txt): the zero-extension of a / c — a Montgomery result sits in the ODD register of its pair, and a 64-bit operand must be an
  // even-aligned pair on gfx950, so the compiler copies it next to a register of zeros (v_mov_b32, 2 SIMD-cycles) — made by a multiplier-class instruction instead (x * 1 + 0, 4 cycles)
  uint64_t za, zc;
  asm("v_mad_u64_u32 %0, vcc, %1, 1, 0" : "=v"(za) : "v"(a) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, 1, 0" : "=v"(zc) : "v"(c) : "vcc");
  const uint64_t t0 = bb::mad_wide<1>(za, b), t1 = bb::mad_wide<1>(zc, d);
#else
  const uint64_t t0 = bb::acc_add(a, b), t1 = bb::acc_add(c, d);
#endif
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
// The 22 partial rounds:  s0 = sbox(x);  sum = s0 + Σ t_i;  x <- sum - 2 s0 + next constant;  t_i <- sum + 2^(i-1) t_i   (diag = -2, 1, 2, 4, .., 1024).
// The update of a passive word is a SHIFT AND AN ADD, so the eleven t_i are kept as 64-bit lazy integers and updated without any multiplication or reduction
// (round 3 of this repo multiplied each by its diagonal constant — a Montgomery product, three instructions — every round): a word grows by ten bits a round, so
// after every THIRD round all eleven are Montgomery-reduced (two instructions a word), which divides their common factor f by R.  Word 0 is the only word an S-box
// sees; it carries the factor h = (f R^6)^(1/7), for which the S-box output x^7 / R^6 has exactly the factor f of the words it is summed with, and the one product
// on its way to the next S-box (pr_c) moves it from f to the next h while adding the next round constant (pr_a).  Per three rounds: 3 x (12 S-box + 11 sum + 1 + 5)
// + 11 + 17 + 17 updates (one instruction where the word is still 32 bits or the shift is at most 4, two otherwise) + 22 for the reduction = 154 vector
// instructions against 3 x 69.  Bounds: |t| < 2^31 after a reduction, < 2^41.1, 2^51.1, 2^61.1 after one, two, three rounds; |sum| < 2^51.5; the reductions take
// |acc| < 2^62.  The words leave with the common factor G = R^-7 (generate(): the second half of the full rounds starts from it), the constants of full round 4 added.
template <int K>
BB_HD int64_t smad(int64_t acc, int32_t x) {                  // acc + K x, x sign-extended by the instruction (v_mad_i64_i32; K an inline constant or a scalar register)
#if defined(__HIP_DEVICE_COMPILE__)
  int64_t r;
  if constexpr (K >= -16 && K <= 64) asm("v_mad_i64_i32 %0, vcc, %1, %3, %2" : "=v"(r) : "v"(x), "v"(acc), "n"(K) : "vcc");
  else asm("v_mad_i64_i32 %0, vcc, %1, %3, %2" : "=v"(r) : "v"(x), "v"(acc), "s"(K) : "vcc");
  return r;
#else
  return acc + (int64_t)K * x;
#endif
}
// word 0 of the next round out of  v = f (sum - 2 s0)
BB_HD int32_t int_next_x(int64_t v, const Consts& c, int r) { return bb::smont_mul_add(bb::smont_reduce_wide(v), c.pr_c[r], c.pr_a[r]); }
// a round whose passive words are 32-bit (the first after a reduction): they leave as 64-bit integers
BB_HD void int_round_a(int32_t& x, const int32_t* t, int64_t* w, const Consts& c, int r) {
  const int32_t s0 = sbox_signed(x);
  int64_t sum = s0;
#pragma unroll
  for (int i = 1; i < T; i++) sum = bb::sacc_add(sum, t[i]);
  x = int_next_x(smad<-2>(sum, s0), c, r);
  w[1] = smad<1>(sum, t[1]); w[2] = smad<2>(sum, t[2]); w[3] = smad<4>(sum, t[3]); w[4] = smad<8>(sum, t[4]); w[5] = smad<16>(sum, t[5]); w[6] = smad<32>(sum, t[6]);
  w[7] = smad<64>(sum, t[7]); w[8] = smad<128>(sum, t[8]); w[9] = smad<256>(sum, t[9]); w[10] = smad<512>(sum, t[10]); w[11] = smad<1024>(sum, t[11]);
}
// a round on 64-bit passive words
BB_HD void int_round_w(int32_t& x, int64_t* w, const Consts& c, int r) {
  const int32_t s0 = sbox_signed(x);
  uint64_t acc = (uint64_t)w[1];
#pragma unroll
  for (int i = 2; i < T; i++) acc += (uint64_t)w[i];
  const int64_t sum = bb::sacc_add((int64_t)acc, s0);
  x = int_next_x(smad<-2>(sum, s0), c, r);
#pragma unroll
  for (int i = 1; i < T; i++) w[i] = (int64_t)(((uint64_t)w[i] << (i - 1)) + (uint64_t)sum);
}
// The bounds above, computed instead of asserted in prose: interval arithmetic over the eleven words through `rounds` rounds without a reduction, from words of at most
// `start`; returns the largest magnitude any 64-bit value reaches (a word after its update, the sum, sum - 2 s0).  A Montgomery reduction adds m p with |m| <= 2^31 before
// it shifts, so it needs |acc| < 2^63 - 2^31 p, for which 2^62 is enough; it returns at most |acc| / 2^32 + p / 2 + 1.
constexpr uint64_t lazy_rounds_bound(uint64_t start, int rounds) {
  uint64_t b[T] = {}, top = 0;
  for (int i = 1; i < T; i++) b[i] = start;
  for (int r = 0; r < rounds; r++) {
    uint64_t sum = bb::P;                                      // |s0| < p
    for (int i = 1; i < T; i++) sum += b[i];
    if (sum + 2 * (uint64_t)bb::P > top) top = sum + 2 * (uint64_t)bb::P;
    for (int i = 1; i < T; i++) { b[i] = sum + (b[i] << (i - 1)); if (b[i] > top) top = b[i]; }
  }
  return top;
}
constexpr uint64_t lazy_reduced_bound(uint64_t acc) { return (acc >> 32) + bb::P / 2 + 2; }
static_assert(lazy_rounds_bound(1ull << 31, 3) < (1ull << 62), "three partial rounds on words below 2^31 stay inside what a Montgomery reduction takes");
static_assert(lazy_reduced_bound(lazy_rounds_bound(1ull << 31, 3)) <= (1ull << 31), ".. and their reduction is a word below 2^31 again");
static_assert(lazy_reduced_bound(lazy_rounds_bound(1ull << 31, 1) + bb::P) < bb::P, "the words leave the last round (one round + the next full round's constant) inside (-p, p)");
BB_HD void int_rounds_scaled(uint32_t* s, const Consts& c) {   // in: unsigned words below p + 128, factor R; out: signed words in (-p, p), factor G, next round's constants added
  int32_t x = (int32_t)s[0] + c.in0_neg;
  int32_t t[T];
  int64_t w[T];
#pragma unroll
  for (int i = 1; i < T; i++) t[i] = (int32_t)s[i];
#pragma unroll 1
  for (int r = 0; r < RP - 1; r += 3) {
    int_round_a(x, t, w, c, r);
    int_round_w(x, w, c, r + 1);
    int_round_w(x, w, c, r + 2);
#pragma unroll
    for (int i = 1; i < T; i++) t[i] = bb::smont_reduce_wide(w[i]);
  }
  int_round_a(x, t, w, c, RP - 1);
  s[0] = (uint32_t)x;
#pragma unroll
  for (int i = 1; i < T; i++) s[i] = (uint32_t)bb::smont_reduce_wide((int64_t)((uint64_t)w[i] + c.pr_k[i]));
}
BB_HD void permute_scaled(uint32_t* s, const Consts& c) {
  ext_linear_scaled<true>(s, c.ext0_s);
#pragma unroll P2_RF_UNROLL
  for (int r = 0; r < RF / 2; r++) {
#pragma unroll
    for (int i = 0; i < T; i++) s[i] = sbox_biased(s[i], c.pre_b[r][i]);
    ext_linear_scaled<false>(s, nullptr);
  }
  int_rounds_scaled(s, c);
#pragma unroll P2_RF_UNROLL
  for (int r = RF / 2; r < RF; r++) {
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
// inside a lane, and the partial-round sum is a 2-step quad butterfly.  1,585 instructions per lane (round 3, in permute()'s arithmetic: 1,912).  All four lanes of a
// quad must be active.  Same function as permute() / permute_scaled(); the canonical results do not depend on the formulation.
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ uint64_t quad_perm64(uint64_t v) { return ((uint64_t)quad_perm<CTRL>((uint32_t)(v >> 32)) << 32) | quad_perm<CTRL>((uint32_t)v); }

// ---- the quad formulation runs the arithmetic of permute_scaled() (round 4; rounds 2-3 ran permute()'s: 1,912 instructions per lane) — wide Montgomery reductions (2
// instructions) instead of Barrett ones (6), signed S-boxes without corrections, and the partial rounds' passive words as 64-bit lazy integers reduced every
// third round (int_rounds_scaled above: here every lane keeps three of them and its own copy of word 0, whose S-box all four lanes run; the sum crosses the quad in two DPP
// steps).  Same Consts, same factors: words enter with F_IN (mont_mul_lazy(x, in_scale) / mont_mul_lazy(previous output, carry)) and leave with F_OUT.
template <bool ADD_RC>
__device__ __forceinline__ void ext_linear_quad_scaled(uint32_t* s, const uint32_t* m4row, const uint32_t* rc) {
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
    s[b] = bb::mont_reduce_wide(v);
  }
}
// mask0: all ones, but zero in lane 0 — its first slot is word 0 itself, not a passive word, and stays 0 ((0 << k) + (sum & 0); a select would be a v_cndmask: 20 SIMD-cycles)
__device__ __forceinline__ void int_round_quad(int32_t& x, int64_t* w, uint64_t mask0, const int* k, const Consts& c, int r) {
  const int32_t s0 = sbox_signed(x);
  uint64_t acc = (uint64_t)w[0] + (uint64_t)w[1] + (uint64_t)w[2];
  acc += quad_perm64<0xB1>(acc);
  acc += quad_perm64<0x4E>(acc);                                // every lane: the sum of the eleven passive words
  const int64_t sum = bb::sacc_add((int64_t)acc, s0);
  x = int_next_x(smad<-2>(sum, s0), c, r);
  w[0] = (int64_t)(((uint64_t)w[0] << k[0]) + ((uint64_t)sum & mask0));
  w[1] = (int64_t)(((uint64_t)w[1] << k[1]) + (uint64_t)sum);
  w[2] = (int64_t)(((uint64_t)w[2] << k[2]) + (uint64_t)sum);
}
// s[b] = state word 4 b + l of the quad's permutation, l = lane & 3
__device__ __forceinline__ void permute_quad_scaled(uint32_t* s, int l, const Consts& c) {
  const uint32_t packed = l == 0 ? 0x03010705u : l == 1 ? 0x01010604u : l == 2 ? 0x07050301u : 0x06040101u;   // row l of M4, one byte per entry
  const uint32_t m4row[4] = {packed & 0xFF, (packed >> 8) & 0xFF, (packed >> 16) & 0xFF, packed >> 24};
  { const uint32_t rc[3] = {c.ext0_s[l], c.ext0_s[4 + l], c.ext0_s[8 + l]}; ext_linear_quad_scaled<true>(s, m4row, rc); }
#pragma unroll 1
  for (int r = 0; r < RF / 2; r++) {
#pragma unroll
    for (int b = 0; b < 3; b++) { uint32_t v = s[b]; asm("" : "+v"(v)); s[b] = sbox_biased(v, c.pre_b[r][4 * b + l]); }   // (opaque: LLVM otherwise proves a sign here and trades v_mad_i64_i32 for an unsigned product + corrections: 108 instructions a round instead of 75)
    ext_linear_quad_scaled<false>(s, m4row, nullptr);
  }
  {
    int32_t x = (int32_t)quad_perm<0x00>(s[0]) + c.in0_neg;
    int64_t w[3] = {l == 0 ? 0 : (int64_t)(int32_t)s[0], (int64_t)(int32_t)s[1], (int64_t)(int32_t)s[2]};
    const int k[3] = {l == 0 ? 0 : l - 1, 3 + l, 7 + l};       // word i is multiplied by 2^(i - 1)
    uint64_t mask0 = l == 0 ? 0ull : ~0ull;
    asm("" : "+v"(mask0));                                // (opaque: an AND with a visible 0 / ~0 select comes back as the v_cndmask it is meant to avoid)
#pragma unroll 1
    for (int r = 0; r < RP - 1; r += 3) {
      int_round_quad(x, w, mask0, k, c, r);
      int_round_quad(x, w, mask0, k, c, r + 1);
      int_round_quad(x, w, mask0, k, c, r + 2);
#pragma unroll
      for (int b = 0; b < 3; b++) w[b] = (int64_t)bb::smont_reduce_wide(w[b]);
    }
    int_round_quad(x, w, mask0, k, c, RP - 1);
#pragma unroll
    for (int b = 0; b < 3; b++) s[b] = (uint32_t)bb::smont_reduce_wide((int64_t)((uint64_t)w[b] + c.pr_k[4 * b + l]));
    if (l == 0) s[0] = (uint32_t)x;
  }
#pragma unroll 1
  for (int r = RF / 2; r < RF; r++) {
#pragma unroll
    for (int b = 0; b < 3; b++) s[b] = sbox_biased(s[b], c.pre_b[r][4 * b + l]);
    ext_linear_quad_scaled<false>(s, m4row, nullptr);
  }
}
// ---- the same permutation spread over a ROW of 16 lanes (round 5: the shortest latency) --------------------------------------------------------
// Lane l < 12 of an aligned group of 16 holds state word l; lanes 12-15 ride along (their M4 multipliers are zero, their passive word stays zero: nothing they hold reaches
// a real word).  Same arithmetic, same Consts and factors as permute_quad_scaled; per lane ONE S-box per full round instead of three and ONE passive word instead of
// three, the sums over the quads / over the row by DPP rotations inside the row (direction-free: every lane adds up all of them).  A lone wave is bound by the chain of dependent
// multiplier instructions, not by issue: ~33 instructions per full round and ~34 per partial round per lane against 75 and ~50 — about 2 us a permutation instead of
// 4.5 (the tree tails, the FRI layers and the transcript steps are chains of such permutations: DESIGN.md 8.9b).  All 16 lanes must be active.
template <int N>
__device__ __forceinline__ uint64_t row_ror64(uint64_t v) {   // the value of the lane N places away in its row of 16 (rotation)
  const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, 0x120 + N, 0xF, 0xF, true), hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), 0x120 + N, 0xF, 0xF, true);
  return ((uint64_t)hi << 32) | lo;
}
template <bool ADD_RC>
__device__ __forceinline__ uint32_t ext_linear_row16_scaled(uint32_t x, const uint32_t* m4row, uint32_t rc) {
  const uint32_t v0 = quad_perm<0x00>(x), v1 = quad_perm<0x55>(x), v2 = quad_perm<0xAA>(x), v3 = quad_perm<0xFF>(x);
  const uint64_t y = (uint64_t)v0 * m4row[0] + (uint64_t)v1 * m4row[1] + (uint64_t)v2 * m4row[2] + (uint64_t)v3 * m4row[3];       // M4 on the lane's quad (lanes 12-15: zero multipliers)
  uint64_t t = y + row_ror64<4>(y);
  t += row_ror64<8>(t);                                        // the sum over the four quads of the row: the column sums of circ(2 M4, M4, M4)
  uint64_t v = y + t;
  if (ADD_RC) v += rc;
  return bb::mont_reduce_wide(v);
}
__device__ __forceinline__ void int_round_row16(int32_t& x, int64_t& w, uint64_t mask, int k, const Consts& c, int r) {
  const int32_t s0 = sbox_signed(x);
  uint64_t acc = (uint64_t)w;
  acc += row_ror64<1>(acc); acc += row_ror64<2>(acc); acc += row_ror64<4>(acc); acc += row_ror64<8>(acc);       // every lane: the sum of the eleven passive words
  const int64_t sum = bb::sacc_add((int64_t)acc, s0);
  x = int_next_x(smad<-2>(sum, s0), c, r);
  w = (int64_t)(((uint64_t)w << k) + ((uint64_t)sum & mask));
}
// s = state word l of the row's permutation (l = lane & 15; anything for l >= 12)
__device__ __forceinline__ uint32_t permute_row16_scaled(uint32_t s, int l, const Consts& c) {
  // (every loop fully unrolled: a lone wave waits for each round constant it loads inside the chain — ~200 cycles a scalar load, 22 + 8 of them; unrolled, the loads
  // have static addresses and are issued ahead of the chain)
  const int q = l & 3, li = l < T ? l : 0;                      // (li: a valid index for the lanes that ride along)
  const uint32_t packed = l >= T ? 0u : q == 0 ? 0x03010705u : q == 1 ? 0x01010604u : q == 2 ? 0x07050301u : 0x06040101u;   // row q of M4, one byte per entry
  const uint32_t m4row[4] = {packed & 0xFF, (packed >> 8) & 0xFF, (packed >> 16) & 0xFF, packed >> 24};
  s = ext_linear_row16_scaled<true>(s, m4row, c.ext0_s[li]);
#pragma unroll
  for (int r = 0; r < RF / 2; r++) {
    uint32_t v = s; asm("" : "+v"(v));                          // (opaque: see permute_quad_scaled)
    s = ext_linear_row16_scaled<false>(sbox_biased(v, c.pre_b[r][li]), m4row, 0);
  }
  {
    const uint32_t word0 = (uint32_t)__shfl((int)s, 0, 16);     // every lane its own copy of word 0, whose S-box all of them run
    int32_t x = (int32_t)word0 + c.in0_neg;
    const bool passive = l >= 1 && l < T;
    int64_t w = passive ? (int64_t)(int32_t)s : 0;
    const int k = passive ? l - 1 : 0;                          // word i is multiplied by 2^(i - 1)
    uint64_t mask = passive ? ~0ull : 0ull;
    asm("" : "+v"(mask));                                      // (opaque: see permute_quad_scaled)
#pragma unroll
    for (int r = 0; r < RP - 1; r += 3) {
      int_round_row16(x, w, mask, k, c, r);
      int_round_row16(x, w, mask, k, c, r + 1);
      int_round_row16(x, w, mask, k, c, r + 2);
      w = (int64_t)bb::smont_reduce_wide(w);
    }
    int_round_row16(x, w, mask, k, c, RP - 1);
    s = (uint32_t)bb::smont_reduce_wide((int64_t)((uint64_t)w + c.pr_k[li]));
    if (l == 0) s = (uint32_t)x;
  }
#pragma unroll
  for (int r = RF / 2; r < RF; r++) s = ext_linear_row16_scaled<false>(sbox_biased(s, c.pre_b[r][li]), m4row, 0);
  return s;
}
#endif

}  // namespace p2
