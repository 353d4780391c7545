// babybear.h — Baby Bear field (p = 2^31 - 2^27 + 1) and its quartic extension F[X]/(X^4 - 11), host + device.
//
// Self-defined prover stages (absent from the reference, parity unpinned).  Values at rest in HBM and
// in proofs are CANONICAL (0 <= x < p).  Multiplication uses Montgomery reduction with R = 2^32:
//     mont_mul(a, b) = a * b * R^-1 mod p
// so a canonical value times a constant stored in Montgomery form (c*R) gives the canonical product directly, and
// data*data products are done on values converted with to_mont()/from_mont() inside a kernel.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define BB_HD __host__ __device__ __forceinline__
#else
#define BB_HD inline
#endif

namespace bb {

constexpr uint32_t P = 0x78000001u;          // 2013265921
constexpr uint32_t NEG_PINV = 0x77FFFFFFu;   // -p^-1 mod 2^32
constexpr uint32_t R1 = 0x0FFFFFFEu;         // 2^32 mod p   (Montgomery form of 1)
constexpr uint32_t R2 = 0x45DDDDE3u;         // 2^64 mod p
constexpr uint32_t GEN = 31u;                // multiplicative generator, LDE coset shift
constexpr uint32_t ROOT27 = 0x1A427A41u;     // 31^15: primitive 2^27-th root of unity
constexpr uint32_t W_EXT = 11u;              // X^4 = 11

// add/sub as (op, op, unsigned min): the wrong candidate always wraps around 2^32 and loses the min
BB_HD uint32_t add(uint32_t a, uint32_t b) { const uint32_t s = a + b, d = s - P; return d < s ? d : s; }
BB_HD uint32_t sub(uint32_t a, uint32_t b) { const uint32_t d = a - b, e = d + P; return e < d ? e : d; }
BB_HD uint32_t neg(uint32_t a) { return a ? P - a : 0u; }
BB_HD uint32_t dbl(uint32_t a) { return add(a, a); }
BB_HD uint32_t mont_mul(uint32_t a, uint32_t b) {
  const uint64_t t = (uint64_t)a * b;
  const uint32_t m = (uint32_t)t * NEG_PINV;
  const uint32_t u = (uint32_t)((t + (uint64_t)m * P) >> 32);     // < 2p
  const uint32_t d = u - P;                                       // wraps above u when u < p
  return d < u ? d : u;                                           // min(u, u - p): 32-bit ops only
}
// ---- lazy helpers (gfx950: v_add/v_sub issue in ~2.3 cycles per wave, v_min/v_mul*/v_mad_u64 in ~4.2-4.5, measured with
// scripts/ubench_alu.hip; a modular add is add+add+min = 8.7 cycles, so hot loops skip the reduction where a value only
// feeds a Montgomery multiplication or a 64-bit accumulation) -------------------------------------------------------
// mont_mul() accepts ONE operand below 2p when the other is canonical: t + m*p < 2p^2 + 2^32 p < 2^64 and u < 2p still holds.
// add_lazy / sub_lazy: unreduced sum / difference of canonical values, below 2p — only ever that lazy operand of a product.
BB_HD uint32_t add_lazy(uint32_t a, uint32_t b) { return a + b; }
BB_HD uint32_t sub_lazy(uint32_t a, uint32_t b) { return a - b + P; }
BB_HD uint32_t reduce_2p(uint32_t x) { const uint32_t d = x - P; return d < x ? d : x; }       // [0, 2p) -> [0, p)
// Montgomery product WITHOUT the final conditional subtraction: result < a*b/2^32 + p.  With p/2^32 = 0.46875:
//   a, b < p            -> result < 1.469 p
//   a < 1.469 p, b < p  -> result < 1.689 p          (needs a*b + 2^32 p < 2^64 and the result below 2^32 = 2.133 p: both hold)
BB_HD uint32_t mont_mul_lazy(uint32_t a, uint32_t b) {
  const uint64_t t = (uint64_t)a * b;
  const uint32_t m = (uint32_t)t * NEG_PINV;
  return (uint32_t)((t + (uint64_t)m * P) >> 32);
}
// (a*b + c) / R mod p, lazy: the 64-bit multiply-add takes the addend for free, so mont(a, b) + c/R costs the three instructions
// of the product alone.  a < 2p, b < p, c < 2^34: a*b + c + 2^32 p < 2^64 and the result is below a*b/2^32 + p + 4 (< 2p).
BB_HD uint32_t mont_mul_add_lazy(uint32_t a, uint32_t b, uint64_t c) {
  const uint64_t t = (uint64_t)a * b + c;
  const uint32_t m = (uint32_t)t * NEG_PINV;
  return (uint32_t)((t + (uint64_t)m * P) >> 32);
}
// acc + K*x in 64 bits as ONE full-rate instruction on the device (v_mad_u64_u32 with an inline-constant multiplier);
// takes the 32-bit x as it is, no zero-extended register pair needed
template <int K>
BB_HD uint64_t mad_wide(uint64_t acc, uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t r;
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %2" : "=v"(r) : "v"(x), "v"(acc), "n"(K) : "vcc");
  return r;
#else
  return acc + (uint64_t)K * x;
#endif
}
BB_HD uint64_t acc_add(uint64_t acc, uint32_t x) { return mad_wide<1>(acc, x); }
// canonical residue of a small multiple of p held in 64 bits: acc < 2^(32+S) and acc < 200 p.  With M = floor(2^(32+S) / p)
// (34 for S = 4, 136 for S = 6) the estimate q = floor((acc >> S) * M / 2^32) never exceeds Q = floor(acc / p) and falls
// short of acc / p by less than Q * 0.0039 + 2^-20 < 1, so q is Q or Q - 1: the remainder candidate is below 2p and fits 32 bits.
// (The device multiply is spelled as an instruction: written in C++, LLVM proves acc >> S fits 32 bits, drops the truncation
// and then emits a 64 x 32-bit product with a dead high half: two extra instructions per reduction.)
BB_HD uint32_t mulhi_u32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  asm("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
  return r;
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
template <int S>
BB_HD uint32_t reduce_wide(uint64_t acc) {
  constexpr uint32_t M = (uint32_t)((1ull << (32 + S)) / P);
  const uint32_t q = mulhi_u32((uint32_t)(acc >> S), M);
  return reduce_2p((uint32_t)acc - q * P);
}
// ---- 96-bit accumulation of plain integer products (round 4: the constraint combination, the DEEP sums, the barycentric dot products) ----
// A sum of products  x_i * y_i  (x_i, y_i ANY 32-bit words: canonical, lazy, Montgomery ...) is formed as an exact 96-bit integer — one 64-bit
// multiply-add with carry-out plus one add-with-carry per term (a multiplier-class + a full-rate instruction, ~6.8 SIMD-cycles per wave64 against the
// ~17 of a lazy Montgomery product added into a 64-bit sum) — and reduced ONCE at the end.  acc96_div_R returns  (sum / 2^32) mod p, canonical: for
// Montgomery-form operands (x R, y R) that is the Montgomery form R * sum(x y) of the sum.  Up to 2^9 terms (hi < 2^9: the reduction's bound).
struct Acc96 { uint64_t lo; uint32_t hi; };
BB_HD Acc96 acc96_zero() { return Acc96{0, 0}; }
#if defined(__HIP_DEVICE_COMPILE__)
// x in a scalar register (a per-proof constant: alpha^c, gamma^k), y in a vector register
__device__ __forceinline__ void mad96_s(Acc96& a, uint32_t x, uint32_t y) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a.lo), "+v"(a.hi) : "s"(x), "v"(y) : "vcc");
}
// four sums at once (the coordinates of an extension-field coefficient times one value): ONE asm statement, so the compiler pads it with hazard
// nops once, not four times
__device__ __forceinline__ void mad96x4_s(Acc96* a, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t y) {
  asm("v_mad_u64_u32 %0, vcc, %8, %12, %0\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "v_mad_u64_u32 %1, vcc, %9, %12, %1\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %10, %12, %2\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
      "v_mad_u64_u32 %3, vcc, %11, %12, %3\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc"
      : "+v"(a[0].lo), "+v"(a[1].lo), "+v"(a[2].lo), "+v"(a[3].lo), "+v"(a[0].hi), "+v"(a[1].hi), "+v"(a[2].hi), "+v"(a[3].hi)
      : "s"(x0), "s"(x1), "s"(x2), "s"(x3), "v"(y) : "vcc");
}
__device__ __forceinline__ void mad96(Acc96& a, uint32_t x, uint32_t y) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a.lo), "+v"(a.hi) : "v"(x), "v"(y) : "vcc");
}
#else
inline void mad96(Acc96& a, uint32_t x, uint32_t y) { const uint64_t t = (uint64_t)x * y, s = a.lo + t; a.hi += s < t; a.lo = s; }
inline void mad96_s(Acc96& a, uint32_t x, uint32_t y) { mad96(a, x, y); }
inline void mad96x4_s(Acc96* a, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t y) { mad96(a[0], x0, y); mad96(a[1], x1, y); mad96(a[2], x2, y); mad96(a[3], x3, y); }
#endif
BB_HD uint32_t acc96_div_R(const Acc96& a) {          // (hi 2^64 + lo) / 2^32 mod p = hi R + lo_hi + lo_lo / R; hi < 2^9
  const uint32_t l0 = (uint32_t)a.lo, l1 = (uint32_t)(a.lo >> 32);
  const uint32_t m = l0 * NEG_PINV;
  const uint32_t r0 = (uint32_t)(((uint64_t)l0 + (uint64_t)m * P) >> 32);      // l0 / R mod p, at most p
  return reduce_wide<6>((uint64_t)a.hi * R1 + l1 + r0);                         // < 2^37 + 2^33: inside reduce_wide<6>'s bounds (2^38, 200 p)
}

// acc / R mod p for a 64-bit acc (Montgomery reduction of a wide sum): two instructions against the six of reduce_wide; the result
// is below acc / 2^32 + p, i.e. "canonical + a little" for the acc < 2^38 sums of the Poseidon2 linear layers, and carries the factor
// 1/R, which the caller has to account for.  acc + 2^32 p < 2^64 is all it needs.
BB_HD uint32_t mont_reduce_wide(uint64_t acc) {
  const uint32_t m = (uint32_t)acc * NEG_PINV;
  return (uint32_t)((acc + (uint64_t)m * P) >> 32);
}
// ---- SIGNED Montgomery arithmetic (the Poseidon2 throughput kernels) ---------------------------------------------------------
// Operands are int32 residues in (-p, p).  With m = t * (-p^-1) mod 2^32 read as a SIGNED word, (t + m p) / 2^32 lies within p/2 of
// t / 2^32, so for |a b| < 2^31 p the product is again in (-p, p) (|r| < |a b| / 2^32 + p / 2 <= 0.469 p + 0.5 p) WITHOUT any
// conditional subtraction: chains of products (the x^7 S-box) cost their three multiplier instructions each and nothing else.
// The 64-bit sums are formed modulo 2^64 (no signed overflow in C++); only the high word of the exact multiple of 2^32 is kept.
BB_HD int32_t smont_mul(int32_t a, int32_t b) {
  const int64_t t = (int64_t)a * b;
  const int32_t m = (int32_t)((uint32_t)t * NEG_PINV);
  return (int32_t)(((uint64_t)t + (uint64_t)((int64_t)m * (int64_t)P)) >> 32);
}
// (a b + c) / R: c a 64-bit addend (|a b + c| / 2^32 + p / 2 bounds the result; c = p 2^32 + .. biases it into (0, 2p) for unsigned use)
BB_HD int32_t smont_mul_add(int32_t a, int32_t b, uint64_t c) {
  const uint64_t t = (uint64_t)((int64_t)a * b) + c;
  const int32_t m = (int32_t)((uint32_t)t * NEG_PINV);
  return (int32_t)((t + (uint64_t)((int64_t)m * (int64_t)P)) >> 32);
}
// acc / R for a signed 64-bit sum: within p / 2 of acc / 2^32
BB_HD int32_t smont_reduce_wide(int64_t acc) {
  const int32_t m = (int32_t)((uint32_t)acc * NEG_PINV);
  return (int32_t)(((uint64_t)acc + (uint64_t)((int64_t)m * (int64_t)P)) >> 32);
}
// acc + x with x sign-extended by the instruction (v_mad_i64_i32 with the inline multiplier 1)
BB_HD int64_t sacc_add(int64_t acc, int32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  int64_t r;
  asm("v_mad_i64_i32 %0, vcc, %1, 1, %2" : "=v"(r) : "v"(x), "v"(acc) : "vcc");
  return r;
#else
  return acc + x;
#endif
}
BB_HD uint32_t to_mont(uint32_t a) { return mont_mul(a, R2); }
BB_HD uint32_t from_mont(uint32_t a) { return mont_mul(a, 1u); }
// canonical * canonical -> canonical (two reductions; for cold paths)
BB_HD uint32_t mul(uint32_t a, uint32_t b) { return mont_mul(mont_mul(a, b), R2); }
BB_HD uint32_t pow(uint32_t a, uint64_t e) {          // canonical in/out
  uint32_t r = R1, b = to_mont(a);
  while (e) { if (e & 1) r = mont_mul(r, b); b = mont_mul(b, b); e >>= 1; }
  return from_mont(r);
}
BB_HD uint32_t inv(uint32_t a) { return pow(a, P - 2); }
BB_HD uint32_t root_of_unity(int log_n) { uint32_t w = ROOT27; for (int i = log_n; i < 27; i++) w = mul(w, w); return w; }

// ---- quartic extension; every coefficient in the SAME form (all canonical or all Montgomery) ----------------
struct E4 { uint32_t c[4]; };
BB_HD E4 e_add(const E4& a, const E4& b) { return E4{{add(a.c[0], b.c[0]), add(a.c[1], b.c[1]), add(a.c[2], b.c[2]), add(a.c[3], b.c[3])}}; }
BB_HD E4 e_sub(const E4& a, const E4& b) { return E4{{sub(a.c[0], b.c[0]), sub(a.c[1], b.c[1]), sub(a.c[2], b.c[2]), sub(a.c[3], b.c[3])}}; }
// Montgomery-form product (inputs and output in Montgomery form).  w11m = 11*R mod p.
BB_HD E4 e_mul_m(const E4& a, const E4& b) {
  constexpr uint32_t W11M = (uint32_t)((11ull * R1) % P);
  const uint32_t a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3], b0 = b.c[0], b1 = b.c[1], b2 = b.c[2], b3 = b.c[3];
  const uint32_t t4 = add(add(mont_mul(a1, b3), mont_mul(a2, b2)), mont_mul(a3, b1));
  const uint32_t t5 = add(mont_mul(a2, b3), mont_mul(a3, b2));
  const uint32_t t6 = mont_mul(a3, b3);
  E4 r;
  r.c[0] = add(mont_mul(a0, b0), mont_mul(W11M, t4));
  r.c[1] = add(add(mont_mul(a0, b1), mont_mul(a1, b0)), mont_mul(W11M, t5));
  r.c[2] = add(add(add(mont_mul(a0, b2), mont_mul(a1, b1)), mont_mul(a2, b0)), mont_mul(W11M, t6));
  r.c[3] = add(add(mont_mul(a0, b3), mont_mul(a1, b2)), add(mont_mul(a2, b1), mont_mul(a3, b0)));
  return r;
}
// E4 (any form) times a base-field scalar given in Montgomery form -> same form as `a`
BB_HD E4 e_mul_fm(const E4& a, uint32_t bm) { return E4{{mont_mul(a.c[0], bm), mont_mul(a.c[1], bm), mont_mul(a.c[2], bm), mont_mul(a.c[3], bm)}}; }
BB_HD E4 e_to_mont(const E4& a) { return E4{{to_mont(a.c[0]), to_mont(a.c[1]), to_mont(a.c[2]), to_mont(a.c[3])}}; }
BB_HD E4 e_from_mont(const E4& a) { return E4{{from_mont(a.c[0]), from_mont(a.c[1]), from_mont(a.c[2]), from_mont(a.c[3])}}; }
BB_HD E4 e_one_m() { return E4{{R1, 0, 0, 0}}; }
BB_HD E4 e_zero() { return E4{{0, 0, 0, 0}}; }
// inverse in Montgomery form via the norm to the quadratic subfield F[Y]/(Y^2 - 11), Y = X^2:
//   a = A(Y) + X*B(Y), A = a0 + a2 Y, B = a1 + a3 Y;  a * (A - X B) = A^2 - Y B^2 =: N in F[Y]/(Y^2-11);  N^-1 = conj(N)/(n0^2 - 11 n1^2)
BB_HD E4 e_inv_m(const E4& a) {
  constexpr uint32_t W11M = (uint32_t)((11ull * R1) % P);
  const uint32_t a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
  // A^2 = (a0^2 + 11 a2^2) + (2 a0 a2) Y ;  B^2 = (a1^2 + 11 a3^2) + (2 a1 a3) Y ;  Y*B^2 = 11*(2 a1 a3) + (a1^2 + 11 a3^2) Y
  const uint32_t A0 = add(mont_mul(a0, a0), mont_mul(W11M, mont_mul(a2, a2))), A1 = dbl(mont_mul(a0, a2));
  const uint32_t B0 = add(mont_mul(a1, a1), mont_mul(W11M, mont_mul(a3, a3))), B1 = dbl(mont_mul(a1, a3));
  const uint32_t n0 = sub(A0, mont_mul(W11M, B1)), n1 = sub(A1, B0);
  const uint32_t d = sub(mont_mul(n0, n0), mont_mul(W11M, mont_mul(n1, n1)));          // norm to F, Montgomery form
  // d^-1 in Montgomery form: pow over Montgomery values
  uint32_t r = R1, b = d; uint32_t e = P - 2;
  while (e) { if (e & 1) r = mont_mul(r, b); b = mont_mul(b, b); e >>= 1; }
  const uint32_t i0 = mont_mul(n0, r), i1 = neg(mont_mul(n1, r));                       // N^-1 = (n0 - n1 Y) / d
  // a^-1 = (A - X B) * N^-1, with (A - X B) = a0 - a1 X + a2 X^2 - a3 X^3 and N^-1 = i0 + i1 X^2
  const E4 conj{{a0, neg(a1), a2, neg(a3)}};
  const E4 ninv{{i0, 0, i1, 0}};
  return e_mul_m(conj, ninv);
}

}  // namespace bb
