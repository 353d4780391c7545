// stark_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the self-defined prover stages.
//
// PARITY UNPINNED.  The reference (seceq/zkir) contains no prover: no AIR, NTT/LDE, Merkle tree, FRI or
// proof format, and its Plonky3 dependency is commented out with no rev (Cargo.toml:67-69).
// Everything in this file is therefore defined by THIS repository ("ZKIR-STARK v1") following
// BASELINE.json:north_star literally (Baby Bear, radix-2 NTT/LDE, Poseidon2 width 12, FRI), and is pinned only by
// algebraic self-checks (tests/test_stark_oracle.py: inverse-NTT round trips, naive DFT, Merkle path checks, a full
// verifier).  The GPU implementation (zkir_amd/csrc/stark.hip) is checked bit-for-bit against this file.
//
// Arithmetic here is deliberately the naive canonical form (u64 products reduced with %), independent of the
// Montgomery arithmetic used on the device.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <map>

namespace so {

// ---------------------------------------------------------------------------------------------
// Baby Bear: p = 2^31 - 2^27 + 1
// ---------------------------------------------------------------------------------------------
static const uint32_t P = 2013265921u;
typedef uint32_t F;
static inline F fadd(F a, F b) { uint32_t s = a + b; return s >= P ? s - P : s; }
static inline F fsub(F a, F b) { return a >= b ? a - b : a + P - b; }
static inline F fmul(F a, F b) { return (F)(((uint64_t)a * b) % P); }
static inline F fneg(F a) { return a ? P - a : 0; }
static F fpow(F a, uint64_t e) { F r = 1; while (e) { if (e & 1) r = fmul(r, a); a = fmul(a, a); e >>= 1; } return r; }
static inline F finv(F a) { return fpow(a, P - 2); }
static const F GEN = 31;                       // generator of F*, also the LDE coset shift
static inline F root_of_unity(int log_n) {     // omega_{2^log_n} = (31^15)^(2^(27-log_n)); 31^15 = 0x1a427a41
  F w = fpow(GEN, 15);
  for (int i = log_n; i < 27; i++) w = fmul(w, w);
  return w;
}

// quartic extension F[X]/(X^4 - 11)
struct E { F c[4]; };
static const F WEXT = 11;
static inline E e_from(F a) { return E{{a, 0, 0, 0}}; }
static inline E eadd(const E& a, const E& b) { return E{{fadd(a.c[0], b.c[0]), fadd(a.c[1], b.c[1]), fadd(a.c[2], b.c[2]), fadd(a.c[3], b.c[3])}}; }
static inline E esub(const E& a, const E& b) { return E{{fsub(a.c[0], b.c[0]), fsub(a.c[1], b.c[1]), fsub(a.c[2], b.c[2]), fsub(a.c[3], b.c[3])}}; }
static inline E emul(const E& a, const E& b) {
  F t[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) t[i + j] = fadd(t[i + j], fmul(a.c[i], b.c[j]));
  return E{{fadd(t[0], fmul(WEXT, t[4])), fadd(t[1], fmul(WEXT, t[5])), fadd(t[2], fmul(WEXT, t[6])), t[3]}};
}
static inline E emul_f(const E& a, F b) { return E{{fmul(a.c[0], b), fmul(a.c[1], b), fmul(a.c[2], b), fmul(a.c[3], b)}}; }
static E epow(E a, uint64_t e) { E r = e_from(1); while (e) { if (e & 1) r = emul(r, a); a = emul(a, a); e >>= 1; } return r; }
static E einv(const E& a) {                     // a^(p^4-2) via Frobenius-free generic power: p^4 - 2 as 128-bit exponent
  // a^-1 = a^(p^4 - 2).  Compute with square-and-multiply over the 124-bit exponent.
  unsigned __int128 e = (unsigned __int128)P * P; e = e * P * P - 2;
  E r = e_from(1), b = a;
  while (e) { if (e & 1) r = emul(r, b); b = emul(b, b); e >>= 1; }
  return r;
}
static inline bool eeq(const E& a, const E& b) { return !memcmp(a.c, b.c, 16); }

// ---------------------------------------------------------------------------------------------
// Poseidon2, width 12, x^7, R_F = 8, R_P = 22 (self-generated constants; a demonstrator instance, not a vetted one)
// ---------------------------------------------------------------------------------------------
static const int T = 12, RF = 8, RP = 22, RATE = 8, DIGEST = 4;
struct P2Consts { F ext[RF][T]; F in[RP]; F diag[T]; };
static uint64_t splitmix64(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static const P2Consts& p2consts() {
  static P2Consts c; static bool init = false;
  if (!init) {
    uint64_t s = 0x5A4B49522D50322Dull;        // "ZKIR-P2-" as big-endian ASCII
    auto next = [&]() -> F { for (;;) { uint32_t v = (uint32_t)(splitmix64(s) >> 33); if (v < P) return v; } };   // 31-bit candidates, rejection-sampled
    for (int r = 0; r < RF; r++) for (int i = 0; i < T; i++) c.ext[r][i] = next();
    for (int r = 0; r < RP; r++) c.in[r] = next();
    c.diag[0] = P - 2;                         // internal matrix = all-ones + diag(-2, 1, 2, 4, ..., 1024)
    for (int i = 1; i < T; i++) c.diag[i] = 1u << (i - 1);
    init = true;
  }
  return c;
}
static inline F sbox(F x) { F x2 = fmul(x, x), x3 = fmul(x2, x), x6 = fmul(x3, x3); return fmul(x6, x); }
static void m4(F* x) {                         // Poseidon2 paper M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
  F a = x[0], b = x[1], c = x[2], d = x[3];
  auto mulc = [](F v, uint32_t k) { return fmul(v, k); };
  x[0] = fadd(fadd(mulc(a, 5), mulc(b, 7)), fadd(c, mulc(d, 3)));
  x[1] = fadd(fadd(mulc(a, 4), mulc(b, 6)), fadd(c, d));
  x[2] = fadd(fadd(a, mulc(b, 3)), fadd(mulc(c, 5), mulc(d, 7)));
  x[3] = fadd(fadd(a, b), fadd(mulc(c, 4), mulc(d, 6)));
}
static void ext_linear(F* s) {                 // circ(2*M4, M4, M4): apply M4 per 4-chunk, then add the per-position sums
  for (int k = 0; k < T; k += 4) m4(s + k);
  F sum[4];
  for (int j = 0; j < 4; j++) sum[j] = fadd(fadd(s[j], s[4 + j]), s[8 + j]);
  for (int k = 0; k < T; k++) s[k] = fadd(s[k], sum[k & 3]);
}
static void int_linear(F* s, const P2Consts& c) {
  F sum = 0;
  for (int i = 0; i < T; i++) sum = fadd(sum, s[i]);
  for (int i = 0; i < T; i++) s[i] = fadd(sum, fmul(s[i], c.diag[i]));
}
static void permute(F* s) {
  const P2Consts& c = p2consts();
  ext_linear(s);
  for (int r = 0; r < RF / 2; r++) { for (int i = 0; i < T; i++) s[i] = sbox(fadd(s[i], c.ext[r][i])); ext_linear(s); }
  for (int r = 0; r < RP; r++) { s[0] = sbox(fadd(s[0], c.in[r])); int_linear(s, c); }
  for (int r = RF / 2; r < RF; r++) { for (int i = 0; i < T; i++) s[i] = sbox(fadd(s[i], c.ext[r][i])); ext_linear(s); }
}
// padding-free overwrite sponge: chunks of RATE overwrite state[0..len), permute after each chunk; digest = state[0..4)
static void hash_elems(const F* in, size_t n, F out[DIGEST]) {
  F s[T]; memset(s, 0, sizeof s);
  for (size_t off = 0; off < n; off += RATE) {
    size_t len = n - off < (size_t)RATE ? n - off : RATE;
    for (size_t i = 0; i < len; i++) s[i] = in[off + i];
    permute(s);
  }
  if (n == 0) permute(s);
  memcpy(out, s, DIGEST * sizeof(F));
}
static void compress(const F l[DIGEST], const F r[DIGEST], F out[DIGEST]) {   // truncated permutation of (l || r || 0000)
  F s[T];
  for (int i = 0; i < 4; i++) { s[i] = l[i]; s[4 + i] = r[i]; s[8 + i] = 0; }
  permute(s);
  memcpy(out, s, DIGEST * sizeof(F));
}

// ---------------------------------------------------------------------------------------------
// NTT (naive radix-2, canonical arithmetic).  Natural order in, natural order out.
// ---------------------------------------------------------------------------------------------
static void bit_reverse(std::vector<F>& a) {
  size_t n = a.size(); int lg = 0; while ((1u << lg) < n) lg++;
  for (size_t i = 0; i < n; i++) { size_t j = 0; for (int b = 0; b < lg; b++) if (i >> b & 1) j |= (size_t)1 << (lg - 1 - b); if (i < j) std::swap(a[i], a[j]); }
}
static void ntt(std::vector<F>& a, bool inverse) {
  size_t n = a.size(); int lg = 0; while ((1u << lg) < n) lg++;
  bit_reverse(a);
  for (int s = 1; s <= lg; s++) {
    size_t m = (size_t)1 << s, h = m >> 1;
    F wm = root_of_unity(s); if (inverse) wm = finv(wm);
    for (size_t k = 0; k < n; k += m) { F w = 1; for (size_t j = 0; j < h; j++) { F t = fmul(w, a[k + j + h]), u = a[k + j]; a[k + j] = fadd(u, t); a[k + j + h] = fsub(u, t); w = fmul(w, wm); } }
  }
  if (inverse) { F ninv = finv((F)(n % P)); for (auto& x : a) x = fmul(x, ninv); }
}
// low-degree extension of evaluations over H = <omega_N> (natural order) to the coset GEN * <omega_{N << log_blowup}>
static void lde(const std::vector<F>& evals, int log_blowup, std::vector<F>& coeffs, std::vector<F>& out) {
  coeffs = evals;
  ntt(coeffs, true);
  size_t n = evals.size(), m = n << log_blowup;
  out.assign(m, 0);
  F sh = 1;
  for (size_t k = 0; k < n; k++) { out[k] = fmul(coeffs[k], sh); sh = fmul(sh, GEN); }
  ntt(out, false);
}


// ---------------------------------------------------------------------------------------------
// main trace matrix of ZKIR-STARK: packed 372-byte reference rows -> W = 172 LOGICAL Baby Bear columns (AIR v6), padded to a
// power of two.  Column map:
//   0 cycle | 1-3 pc limbs (20/20/24 bits) | 4 op7 | 5 fa (bits 10:7) | 6 fb (14:11) | 7 fc (18:15) | 8 fhi (31:19)
//   9+3r+l register limbs (20-bit limbs when Normalized, 30-bit when Accumulated; l = 2: the bits above) | 57+r storage state
//   73+(r-1)  wr[r], r = 1..15: the row's instruction writes register r
//   88+(r-1)  selb[r]: one-hot of field b          103+(r-1) selc[r]: one-hot of field c (of field a on BNE rows: rs1 sits there)
//   118-120 xb = limbs of reg[fb]   121-123 xc = limbs of reg[fc | fa]   124-126 y = limbs of the value written to rd
//   127-133 class one-hot: add, addi, bne, jal, oth (any other executed instruction), halt (last executed row), pad
//   134 opclass of the instruction word (0 add, 1 addi, 2 bne, 3 jal, 4 anything else): part of the instruction-ROM tuple, so the class
//       one-hot of an executed row is tied to the PROGRAM's word at pc, not to a free witness
//   135-138 range chunks: y0 = c135 + 1024 c136, y1 = c137 + 1024 c138, each looked up in the 10-bit table (range_check.rs:175-192, config.rs:78-80)
//   139 s = bit 31 of the word (sign of imm17 / off21) | 140 se = (tk + k_jal) s
//   141-142 c0 c1 carries of the value addition | 143-145 d0 d1 d2 carries of pc + delta | 146 dl0 = low limb of the pc increment
//   147 ne = [xb != xc] | 148-150 iv: inverse witness of the first differing limb | 151 tk = branch taken
// AIR v3 (opcode FAMILIES: a family is a pair of opcodes that differ in their low bit, the polarity of one comparison):
//   152-155 class one-hot, continued: sub, bru (BLTU / BGEU), se (SEQ / SNE), su (SLTU / SGEU); column 129 (was "bne") is the family
//           bre (BEQ / BNE).  Class ids = opclass values: add 0, addi 1, bre 2, jal 3, oth 4, (halt 5, pad 6,) sub 7, bru 8, se 9, su 10
//   z = (c135 + 1024 c136, c137 + 1024 c138): the two RANGE-CHECKED 20-bit limbs of the row (columns 156-157 until v5; since v6 z has no columns
//           of its own): the written value's low limbs on add / addi / jal / sub / oth rows (y = z there), the 40-bit difference xb - xc on su
//           rows and xc - xb on bru rows (whose borrow c1 is the comparison, execute.rs:361-407, :598-636), zero elsewhere
//   158 flag = the family's comparison: [xb == xc] (raw 64-bit, all three limbs) on bre / se rows, the borrow c1 on bru / su rows, else 0
//   159 fx = flag XOR (op - the family's even opcode): the branch decision (tk) of bre / bru rows, the value written by se / su rows
// AIR v4 (control flow of every opcode but the two signed branches):
//   160-161 class one-hot, continued: jalr (id 11), oj (id 12) = "other, jumps": BLT / BGE, whose comparison the AIR cannot state yet — their
//           next pc is free, they write nothing.  Class "other" (id 4) is now SEQUENTIAL: pc' = pc + 4 like every instruction that is
//           not a branch or a jump (loads, stores, the remaining ALU opcodes, ECALL: execute.rs advances pc by 4 in each of them)
//   162 b0 = the bit JALR clears: next pc + b0 = rs1 + sext(imm17) over (20, 20, 24)-bit limbs mod 2^64 (execute.rs:649-658), carries d0 d1 d2
// AIR v5 (signed comparisons; the control flow of EVERY opcode): the ordered families have four members, op = base + 2 g + pol — SLTU SGEU SLT
//   SGE (class su, base 0x20, g = signed) and BLT BGE BLTU BGEU (class bru, base 0x42, g = unsigned); class oj is left to the deferred mode
//   163-166 range chunks of u, the row's SECOND range-checked pair: the BIASED high limbs ta, tb of the operands of an ordered comparison
//           (t = limb + 2^19 sgn - 2^20 (sign bit): the limb of value XOR 2^39 when the comparison is signed, value.rs:710-716); since v6 also
//           the bits above 40 of what an "other" row writes (y2 = c163 + 2^10 c164 + 2^20 c165, c166 = 64 c165)
//   167 g = the word's variant bit (eleventh element of the ROM tuple) | 168 sb = sign bit of the second operand (the first one's is column 162)
// AIR v6 (conditional moves; every register limb range-checked):
//   156 nz = [xc != 0] over the raw 64 bits, on every row | 157 ivz = inverse of the sum of xc's limbs
//   169-170 class one-hot, continued: cmn (id 13: CMOV / CMOVNZ), cmz (id 14: CMOVZ) | 171 q = the row is a conditional move whose condition holds
// ---------------------------------------------------------------------------------------------
static const int W_MAIN = 172;                                     // logical columns of modes 0 (default) and 1 (deferred)
// ---- MODE 2 (round 4): the default VM mode WITH the I/O argument.  Eight more logical columns, committed only in this mode (152 + 8 = 160):
//   172 f2 = the row is a WRITE ecall (R10 = 2) | 173 rl = a READ ecall (R10 = 1) that finds the input tape non-empty | 174 re = a READ ecall on an exhausted tape
//   175 fh = a hash ecall (R10 = 3..6) | 176-177 h0, h1: R10 - 3 in binary on hash rows | 178 oc = outputs written before this row | 179 ic = inputs consumed before it
// ECALL is a class of its own there (id 15, no column: Kec = f2 + rl + re + fh); its rows send (oc, R11's limbs) / (ic, the limbs written to R10) into a LogUp relation
// whose table side the VERIFIER forms from the I/O tapes the proof carries (their digest is the public io digest): syscall.rs:94-177.
static const int W_MAIN_IO = 180;
enum { C_F2 = 172, C_RL = 173, C_RE = 174, C_FH = 175, C_H0 = 176, C_H1 = 177, C_OC = 178, C_IC = 179 };
static const int K_ECALL = 15;                                     // class id of the ECALL word in modes 2 / 3 (the ROM tuple's opclass); modes 0 / 1: "other"
static const uint32_t OP_ECALL = 0x50;
// (round 5, format v11) EBREAK in modes 2 / 3: a class id NO selector carries, so constraint 4 (sum_k k K_k = opclass on an executed row) cannot hold on an EBREAK row — the
// VM halts there (execute.rs:667): only the halt row, whose class is not the word's, can sit on it.  Modes 0 / 1: "other" (sequential, one free value), as in v10.
static const int K_EBREAK = 21;
static const uint32_t OP_EBREAK = 0x51;
// ---- MODE 3 (round 4): mode 2 WITH the memory argument — the ten loads and stores (execute.rs:477-575, memory.rs:86-505) are classes of their own (ld = 16, st = 17)
// and every access is tied to a consistent memory by an offline memory check (Blum et al.) over aligned 8-byte CELLS: a load / store row READS the tuple
// (cell address, last-access time, the cell's 8 bytes) and WRITES (cell address, its own cycle + 1, the new 8 bytes) — new = old on loads, old with the accessed
// window replaced by the stored bytes on stores — the time read must be smaller than the time written, and over the whole run {initial cells} + {written} =
// {read} + {final cells} as multisets (LogUp, one relation).  The VERIFIER supplies the two ends: the proof carries the list of touched cells (strictly increasing
// canonical addresses, final bytes, final time), the initial bytes are the program image's (code at 0x1000, data behind it, vm.rs:153-170) or zero.
// 40 more logical columns (180 + 40 = 220, 200 committed):
//   180 kld, 181 kst: the row is a load / a store | 182-196 e_v: one-hot of the accessed WINDOW v = (width, offset in the cell): v 0-7 one byte at offset v, 8-11 two
//   bytes at 2 (v - 8), 12-13 four bytes at 4 (v - 12), 14 the whole cell | 197-204 ob_0..7: the cell's bytes BEFORE the access | 205 told: the time of the previous access
//   206-214 the nine PIECES d0 d1 n0 n1 d3 d4 d5 d6 d7 of the 64-bit window value (bytes, byte 2 as two nibbles n0 + 16 n1: the (20, 20, 24)-bit register limbs are
//   d0 + 2^8 d1 + 2^16 n0 | n1 + 2^4 d3 + 2^12 d4 | d5 + 2^8 d6 + 2^16 d7): the stored register on stores, the loaded window on loads (zero-extended; on byte / halfword
//   loads d6 = 2 x the low seven bits of the top byte) | 215 sgb, 216 sgh: the row is LB / LH (sign-extending) | 217 tb: the top bit of the loaded byte / halfword
//   218 sx = (sgb + sgh) tb | 219 cm2: the carry out of the address's third limb (the address itself must stay below 2^40: addr_limbs = 2, config.rs:30)
// .. and the six bitwise opcodes AND OR XOR ANDI ORI XORI (execute.rs:199-282: on the 40-bit values, the immediate sign-extended and masked) as class lg = 18, NIBBLE by nibble: the
// operands and the result are ten nibbles each, nibble k a tuple (a_k, b_k, r_k) looked up in the 256-entry table of the row's operation — in the nine piece slots (a_k = piece k) and,
// the tenth, in the row's last range slot (a_9 = chunk R7).  24 more logical columns (244, 224 committed):
//   220 klg | 221 oa, 222 oo: the operation is AND / OR (XOR = klg - oa - oo) | 223 li: the second operand is the immediate | 224-233 b_0..9 | 234-243 r_0..9
// .. and the six SHIFTS SLL SRL SRA SLLI SRLI SRAI (execute.rs:284-358, value.rs:658-697: on the 40-bit value; the amount is rs2 & 63 or the word's 8-bit shamt; 40 and more
// shift everything out), class sh = 19, as ONE relation a 2^t = H 2^40 + L: a left shift by sh is L at t = sh, a right shift by sh is H at t = 40 - sh.  a comes in its four 10-bit
// chunks c_i (the row's second range group), t = 10 u + v by two one-hots: c_i 2^v = lo_i + 2^10 hi_i (both in the 10-bit table: unique), the chunks of a 2^v are m_i = lo_i +
// hi_(i-1) — no carry: lo_i ends in v zeros, hi_(i-1) < 2^v — and the one-hot of u picks four of them.  SRA adds sign (2^40 - 2^t).  32 more logical columns (276, 256 committed):
//   244 ksh | 245-249 ul_0..4, 250-254 ur_0..4: the row shifts LEFT / RIGHT with chunk shift u | 255-264 v_0..9: the bit shift | 265 sa: SRA / SRAI | 266 si: the amount is the
//   word's shamt | 267 sb9: bit 39 of a | 268 sgn = sa sb9 | 269-272 pr_i = c_i 2^v | 273-274 the limbs of 2^40 - 2^t on right shifts | 275 sh: the shift amount
// shared columns on a shift row: R0..R3 = lo_i, R4..R7 = c_i, pieces 0-3 = hi_i, piece 4 = 2 (c_3 mod 2^9), piece 5 = the bits of rs2's low limb above its first chunk (or of the
// word's field above the shamt), piece 6 = d (sh beyond what t can say), piece 7 = the shamt's high nibble, piece 8 = rs2's first chunk, looked up with sh in LOW6 = {(v, v & 63)}
// .. and MUL (execute.rs:79-99: the 40-bit values, the product mod 2^40), class mu = 20, as a schoolbook product in 10-bit chunks: a = rs1's low limbs in the row's second range
// group (R4..R7), b = rs2's in pieces 0-3, the result's chunks r_k = R0..R3 (they ARE z), and for k = 0..3   sum_{i+j=k} a_i b_j + carry_(k-1) = r_k + 2^10 carry_k   with
// carry_0 = piece 4 (< 2^10), carry_1 = piece 5 + 2^10 e_1 (< 2^11), carry_2 = piece 6 + 2^10 (e_2 + 2 e_3) (< 3 * 2^10 + 4), carry_3 = piece 7 + 2^10 piece 8 (dropped: mod 2^40)
// — every slot a 10-bit range lookup on such a row, e_1..3 booleans: both sides stay below 2^31 - 2^27, so each equation holds over the integers and the decomposition is unique.
// The products are of degree 2 already, so the class selector cannot multiply them: ma_i = kmu a_i are columns of their own (zero off MUL rows).  8 more logical columns (284, 264):
//   276 kmu | 277-280 ma_0..3 | 281-283 e_1..3
// MULH, DIVU, REMU, DIV, REM stay class "other": they work on the RAW 64-bit registers (Q2, Q3) — 128-bit products, more chunk lookups than a row has slots.
static const int W_MAIN_MEM = 284;
enum { C_KMU = 276, C_MA = 277, C_ME = 281 };
// ---- MODE 4 (round 6, proof format v12) = mode 3 WITH three more things: the WIDE-ARITHMETIC class, HASH SYSCALLS, and the code segment's BOUNDARY CELL. ----
// (a) class wa = 22: MULH, DIVU, REMU, DIV, REM (opcodes 3..7; execute.rs:101-183) on operands BELOW 2^40.
// The reference computes these five on the RAW 64-bit registers (quirks Q2, Q3): MULH = ((a b as u128) >> 40) & (2^40 - 1), DIVU / REMU = a / b, a % b, DIV / REM the same on
// `as i64` (wrapping).  For registers below 2^40 — everything ADD .. MUL, the logic opcodes, the shifts, the comparisons and LW / LHU / LBU ever write — an i64 is non-negative, so
// DIV = DIVU, REM = REMU, the 80-bit product's bits above 80 are empty, and all five are ONE relation over 40-bit integers:
//       F1 F2 + ADD = LO + 2^40 HI        MULH: a b = L + 2^40 y          DIVU / DIV: y b + r = a, r < b          REMU / REM: q b + y = a, y < b
// A wa row states kwa xb2 = kwa xc2 = 0: a run that feeds one of the five a register with bits above 40 (a sign-extended LB / LH result, an LD, a READ input or a link above
// 2^40) has NO mode-4 proof — sound, incomplete, and stated; division by zero never is a row (the VM stops: RuntimeError::DivisionByZero).
// Schoolbook in 10-bit chunks like MUL, but the 80-bit product, the addend and the remainder's range check need 23 lookups where a mode-3 row has 17: the mode adds SIX more
// 10-bit range slots X0..X5 per row (24 aux columns) and 22 main columns (kwa = om + od + orr is an expression; the signed variants DIV / REM are the word's VARIANT BIT g, the
// ROM tuple's eleventh element: op = 3 om + 4 od + 5 orr + 2 g on a wa row):
//   284 om (MULH), 285 od (DIVU / DIV: the quotient is written), 286 orr (REMU / REM: the remainder is)
//   287-290 gf_k = kwa F1_k (the products have degree 2 already: gated copies, like MUL's ma_k) | 291-299 e_1..e_9: the bits of the carries above their 10-bit slot | 300-305 X0..X5
// Slots of a wa row (every one reads the 10-bit table): R0..R3 = LO's chunks, R4..R7 = F1's, pieces 0-3 = F2's (= rs2's), pieces 4-6 = the low parts of carries c0 c1 c2
// (c1 = p5 + 2^10 e1, c2 = p6 + 2^10 (e2 + 2 e3)), pieces 7, 8 and X0, X1 = G4: HI's chunks (MULH) / ADD's = the remainder's (divisions), X2..X5 = G5: on MULH the carries
// c3 c4 c5 (c3 = X2 + 2^10 (e4 + 2 e5), c4 = X3 + 2^10 (e6 + 2 e7), c5 = X4 + 2^10 (e8 + 2 e9)), on divisions the chunks of d = b - r - 1 (>= 0: r < b; borrow e4).
// Position k of the product: sum_{i+j=k} F1_i F2_j + ADD_k + c_(k-1) = LO_k + 2^10 c_k (k <= 3), = HI_(k-4) + 2^10 c_k (k = 4, 5; k = 6: HI_2 + 2^10 HI_3); a division has HI = 0,
// c3 = 0 and no product above position 3 (q b <= a < 2^40).  Every sum stays below 2^23: the equations hold over the integers and every decomposition is unique.
// (b) HASH SYSCALLS (syscall.rs:121-171, crypto.rs:223-395: SHA-256 = 3, Keccak-256 = 5, BLAKE3 = 6; Poseidon2 = 4 is an error in the reference: never a row) are a TAPE, like
// the I/O tapes of mode 2: the proof carries one record per call — (cycle, input pointer, input length, output pointer, kind) and, per aligned 8-byte cell the call touches
// (the cells under [in, in + len) and [out, out + 32), ascending), the cell's bytes BEFORE the call and the time of its previous access — and the VERIFIER does the rest: it reads
// the message out of the old bytes, computes the digest ITSELF, lays it over the output range (SHA-256: eight u32 writes of the big-endian-parsed words, crypto.rs:251-254; the
// other two byte by byte), and adds the call's memory accesses to the table side of the memory check (read (cell, told, old), written (cell, cycle + 1, new); told <= cycle checked
// in the clear).  The AIR ties the ECALL row to its record with ONE lookup — HH (alpha - fp(cycle, R11's limbs, R12's, R13's, kind) - 12 lambda^11) = fh, four aux columns — and
// keeps what mode 2 already states about the row (R10 = 3 + h0 + 2 h1, R10 <- 0 and nothing else written).  Mode 3's "no hash syscall" constraint becomes "no syscall 4".  A
// hash call costs the proof 8 + 5 words per touched cell: the calls are PUBLIC in this design, like the final memory — what it costs NOT to arithmetise SHA-256 (64 rounds x ~20
// lookups per block: a 2^22-cycle hash chain would need a 2^27-row trace).
// (c) the BOUNDARY CELL: mode 3 refuses every touched cell that overlaps the code segment (check 55) — sound, but when code_size % 8 == 4 the last code word shares
// its cell with the first four data bytes (the reference puts the data right behind the code, vm.rs:163-168), and a program that touches them had no proof.  Mode 4 admits THAT
// cell and states that no store ever writes its low half: 306 iws, 307 nb with nb = delta iws, delta = (cell address - B) as a field element (B = the boundary cell, a public
// constant of the program; CODE_BASE when there is none), and tl (kst - nb) = 0 with tl = the sum of the windows that touch bytes 0..3 — a store with such a window needs
// delta != 0.  (Cells with address = B modulo p — 2^9 of the 2^37 cells — are refused with it: stated.)  Loads of the code word stay possible; cells INSIDE the code stay refused.
// (d) the WIDE TAPE (round 6, second half): the five wide opcodes on operands with bits ABOVE 40 — the reference computes them on the raw 64-bit registers (`as i64`, 128-bit
// products: quirks Q2 / Q3; a sign-extended LB result, an LD, a 64-bit input) — are proven like the hash calls: column 131 ot = "this wide row is proven through the tape"
// (the class is om + od + orr + ot; the chunk relation is gated by om + od + orr alone), the proof carries one record (cycle, rs1's three limbs, rs2's, the opcode) per such row,
// the VERIFIER computes the result with the reference's own semantics (wide_result below) and the row looks its tuple (cycle, rs1, rs2, rd's new value, opcode) up:
// WW (alpha - fp - 13 lambda^11) = ot, WW in the four padding columns beside HH.  Any wide row MAY take the tape (sound either way: the verifier recomputes it); the honest
// prover takes it exactly when an operand has bits above 40, so proofs stay canonical.  With it EVERY run of the VM has a mode-4 statement for every opcode it executes.  The
// class "other" does not exist in mode 4 — no opcode is left in it — and ITS COLUMN (131) is ot there: no column is added, no committed position moves (308 logical, 288 committed).
static const int W_MAIN_WIDE = 308, W_MAX = 308;
enum { C_OM = 284, C_OD = 285, C_ORR = 286, C_GF = 287, C_WE = 291, C_X = 300, C_IWS = 306, C_NB = 307, C_OT = 131 /* = C_K + K_OTH: the class column "other", re-used (mode 4 has no such class) */ };
static const int K_WA = 22, N_X = 6, N_WE = 9, TAG_HASH = 12, TAG_WIDE = 13;
static inline bool is_wide(uint32_t op) { return op >= 0x03 && op <= 0x07; }
// what MULH 3 / DIVU 4 / REMU 5 / DIV 6 / REM 7 write, on the raw 64-bit registers (execute.rs:101-183): MULH = bits 40..79 of the 128-bit product; DIVU / REMU unsigned;
// DIV / REM on `as i64` with wrapping_div / wrapping_rem (i64::MIN / -1 = i64::MIN, remainder 0).  b = 0 never is a row of a division (the VM stops with DivisionByZero).
static inline uint64_t wide_result(uint32_t op, uint64_t a, uint64_t b) {
  if (op == 0x03) return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 40) & ((1ull << 40) - 1);
  if (b == 0) return 0;
  if (op == 0x04) return a / b;
  if (op == 0x05) return a % b;
  const int64_t sa = (int64_t)a, sb = (int64_t)b;
  const bool ovf = sa == INT64_MIN && sb == -1;
  return op == 0x06 ? (ovf ? (uint64_t)INT64_MIN : (uint64_t)(sa / sb)) : (ovf ? 0 : (uint64_t)(sa % sb));
}
static const int K_MU = 20;
static const uint32_t OP_MUL = 0x02;
enum { C_KSH = 244, C_UL = 245, C_UR = 250, C_V = 255, C_SA = 265, C_SI = 266, C_SB9 = 267, C_SGN = 268, C_PR = 269, C_ON = 273, C_SH = 275 };
static const int K_SH = 19, TAG_LOW6 = 11;
static inline bool is_shift(uint32_t op) { return op >= 0x18 && op <= 0x1D; }
enum { C_KLD = 180, C_KST = 181, C_E = 182, C_OB = 197, C_TOLD = 205, C_PIECE = 206, C_SGB = 215, C_SGH = 216, C_TB = 217, C_SX = 218, C_CM2 = 219,
       C_KLG = 220, C_OA = 221, C_OO = 222, C_LI = 223, C_LB = 224, C_LR = 234 };
static const int K_LD = 16, K_ST = 17, K_LG = 18, N_WIN = 15, N_PIECE = 9, N_NIB = 10;
static inline bool is_logic(uint32_t op) { return op >= 0x10 && op <= 0x15; }
static inline int win_width(int v) { return v < 8 ? 1 : v < 12 ? 2 : v < 14 ? 4 : 8; }
static inline int win_start(int v) { return v < 8 ? v : v < 12 ? 2 * (v - 8) : v < 14 ? 4 * (v - 12) : 0; }
static inline int win_of(int width, int off) { return width == 1 ? off : width == 2 ? 8 + off / 2 : width == 4 ? 12 + off / 4 : 14; }
static const uint32_t OP_LB = 0x30, OP_LD = 0x35, OP_SB = 0x38, OP_SD = 0x3B;
static inline bool is_load(uint32_t op) { return op >= OP_LB && op <= OP_LD; }
static inline bool is_store(uint32_t op) { return op >= OP_SB && op <= OP_SD; }
static inline int mem_width(uint32_t op) { return is_store(op) ? 1 << (op - OP_SB) : op <= 0x31 ? 1 : op <= 0x33 ? 2 : op == 0x34 ? 4 : 8; }
static inline int logical_width(int mode) { return mode == 4 ? W_MAIN_WIDE : mode == 3 ? W_MAIN_MEM : mode == 2 ? W_MAIN_IO : W_MAIN; }
enum { C_CYCLE = 0, C_PC = 1, C_OP = 4, C_FA = 5, C_FB = 6, C_FC = 7, C_FHI = 8, C_LIMB = 9, C_STATE = 57, C_WR = 73, C_SELB = 88, C_SELC = 103,
       C_XB = 118, C_XC = 121, C_Y = 124, C_K = 127, C_OPC = 134, C_RC = 135, C_S = 139, C_SE = 140, C_C0 = 141, C_C1 = 142, C_D0 = 143, C_D1 = 144, C_D2 = 145,
       C_DL0 = 146, C_NE = 147, C_IV = 148, C_TK = 151, C_K2 = 152, C_NZ = 156, C_IVZ = 157, C_FLAG = 158, C_FX = 159, C_K3 = 160, C_B0 = 162, C_RC2 = 163, C_G = 167, C_SB = 168,
       C_K4 = 169, C_Q = 171 };
static const int N_RC = 8;                                         // range lookups of a row: the chunks of z (C_RC ..) and of u (C_RC2 ..)
static inline int rc_col(int k) { return k < 4 ? C_RC + k : C_RC2 + (k - 4); }
// AUX trace (committed AFTER the lookup challenges are drawn; AIR v2): 40 base columns = ten extension-field columns (24 = six until v4),
// coordinate by coordinate: H0..H7 = 1 / (alpha - range chunk i), HR = 1 / (alpha - fingerprint of the row's instruction tuple),
// S = running sum of (H0 + .. + H7 + HR - T / N) — a LogUp argument whose table side the VERIFIER computes (T) from the program
// carried in the proof and the multiplicities the prover sends before alpha is drawn.
// LOGICAL vs COMMITTED columns (proof format v7).  The map above is the LOGICAL main trace: what the AIR talks about.  Some logical columns
// are identically zero by the AIR's own constraints and are not committed: the three limbs of R0 (hard-wired zero, state.rs:77-85) and its
// storage state, and — in the default VM mode, where no register is ever Accumulated (the deferred model is off, vm.rs:47) — all 16 storage
// states and (v6) the class column "other, jumps".  The committed ("physical") matrix is the logical one with those columns removed:
// 152 columns in default mode (172 - 20), 168 in deferred mode (172 - 4), whole blocks of 8.  A removed column reads as the constant 0 wherever the
// constraints, the boundary states or the lookups mention it.
static const int W_AUX = 40;
static const int W_AUX_IO = 48, W_AUX_MEM = 96, W_AUX_WIDE = 128, W_AUX_MAX = 128;   // mode 4: + XH0..XH5 (the helpers of the six extra range slots), HH (the hash-call helper), four columns of zero padding (whole blocks of 8)    // mode 2: + HO (output helper), HI (input helper); mode 3: + P0..P8 (piece helpers), HMR, HMW (memory read / write helpers), FPN (fingerprint of the new cell bytes)
static inline int aux_width(int mode) { return mode == 4 ? W_AUX_WIDE : mode == 3 ? W_AUX_MEM : mode == 2 ? W_AUX_IO : W_AUX; }
// (AIR v6) the class column "other, jumps" (C_K3 + 1) is identically zero in the default mode too — no opcode's class is oj there (constraint 4) — and is not committed
static const int C_KOJ = C_K3 + 1;
// mode: 0 default, 1 deferred, 2 default + I/O argument (a bool passed for `mode` reads as 0 / 1)
static inline bool is_virtual(int c, int mode) { return (c >= C_LIMB && c < C_LIMB + 3) || (c >= C_F2 && mode < 2) || (c >= C_KLD && mode < 3) || (c >= C_OM && mode != 4) || (mode == 1 ? c == C_STATE : ((c >= C_STATE && c < C_STATE + 16) || c == C_KOJ)); }
static inline int phys_col(int c, int mode) { return c - (c >= C_LIMB + 3 ? 3 : 0) - (mode == 1 ? (c > C_STATE ? 1 : 0) : (c >= C_STATE + 16 ? 16 : 0) + (c > C_KOJ ? 1 : 0)); }   // of a non-virtual column
static inline int phys_width(int mode) { return mode == 1 ? 168 : mode == 2 ? 160 : mode == 3 ? 264 : mode == 4 ? 288 : 152; }   // 172 - 20 = 152, 172 - 4 = 168, 180 - 20 = 160: whole blocks of 8, no padding
// logical [logical_width][N] -> committed [phys_width][N]
static void to_physical(const std::vector<F>& M, size_t N, int mode, std::vector<F>& out) {
  out.assign((size_t)phys_width(mode) * N, 0);
  for (int c = 0; c < logical_width(mode); c++) if (!is_virtual(c, mode)) memcpy(&out[(size_t)phys_col(c, mode) * N], &M[(size_t)c * N], N * sizeof(F));
}
// a row of committed values (base field or extension) -> the logical row the constraints read (W_MAX entries)
template <class V>
static void to_logical_row(const V* phys, int mode, const V& zero, V* logical) {
  for (int c = 0; c < W_MAX; c++) logical[c] = is_virtual(c, mode) ? zero : phys[phys_col(c, mode)];
}
enum { A_H = 0, A_HR = 32, A_S = 36, A_HO = 40, A_HI = 44, A_P = 48, A_HMR = 84, A_HMW = 88, A_FPN = 92, A_X = 96, A_HH = 120, A_WW = 124 };
// (mode 3) the lookup tables beside the 10-bit range table (no tag) and the ROM (tag 1) / tapes (2, 3): LOW3 = {(v, v & 7)}, v < 2^10 (tag 4: the first range chunk of a
// memory row is looked up HERE, with the window's offset — the address's low three bits), BYTE = {v < 2^8} (tag 5), NIBBLE = {v < 2^4} (tag 6); memory tuples carry tag 7
static const int TAG_LOW3 = 4, TAG_BYTE = 5, TAG_NIB = 6, TAG_MEM = 7, TAG_AND = 8, TAG_OR = 9, TAG_XOR = 10;   // (8-10: the nibble tables {(a, b, a op b)} of the bitwise opcodes)
static const int PIECE_TAG[9] = {TAG_BYTE, TAG_BYTE, TAG_NIB, TAG_NIB, TAG_BYTE, 0, TAG_BYTE, TAG_BYTE, 0};   // d0 d1 n0 n1 d3 d4 d5 d6 d7 (0: the 10-bit range table)
static const int RC_BITS = 10, RC_TABLE = 1 << RC_BITS;            // the reference's range-check table: 2^(limb_bits/2) entries (range_check.rs:29, config.rs:78-80)
static const int N_TUPLE = 11;                                     // instruction-ROM tuple: pc limbs (3), op, fa, fb, fc, fhi, s, opclass, g (variant bit)
static inline int state_col(int i) { return i < 4 ? i : C_LIMB + (i - 4); }    // cycle, pc[3], limbs[48], states[16]: columns 0..3 and 9..72

enum { K_ADD = 0, K_ADDI = 1, K_BRE = 2, K_JAL = 3, K_OTH = 4, K_HALT = 5, K_PAD = 6, K_SUB = 7, K_BRU = 8, K_SE = 9, K_SU = 10, K_JALR = 11, K_OJ = 12, K_CMN = 13, K_CMZ = 14, N_CLASS = 15 };
static inline int kcol(int k) { return k < 7 ? C_K + k : k < 11 ? C_K2 + (k - 7) : k < 13 ? C_K3 + (k - 11) : C_K4 + (k - 13); }
static const uint32_t OP_ADD = 0x00, OP_SUB = 0x01, OP_ADDI = 0x08, OP_SLTU = 0x20, OP_SGEU = 0x21, OP_SEQ = 0x24, OP_SNE = 0x25, OP_BEQ = 0x40, OP_BNE = 0x41,
                      OP_BLT = 0x42, OP_BGE = 0x43, OP_BLTU = 0x44, OP_BGEU = 0x45, OP_JAL = 0x48, OP_JALR = 0x49;
// the even opcode of a family (its polarity-0 member); 0 for the classes that are one opcode
static inline uint32_t family_base(int k) { return k == K_BRE ? OP_BEQ : k == K_BRU ? OP_BLT : k == K_SE ? OP_SEQ : k == K_SU ? OP_SLTU : 0; }
// AIR v5: the families of ordered comparisons have FOUR members, op = base + 2 g + pol: SLTU SGEU SLT SGE (base 0x20, g = signed) and
// BLT BGE BLTU BGEU (base 0x42, g = unsigned).  g is the word's VARIANT BIT, part of the ROM tuple (0 for every other opcode).
static const uint32_t OP_SLT = 0x22, OP_SGE = 0x23, OP_CMOV = 0x26, OP_CMOVZ = 0x27, OP_CMOVNZ = 0x28;
static inline uint32_t variant_bit(uint32_t op, int mode = 0) { return (op == OP_SLT || op == OP_SGE || op == OP_BLTU || op == OP_BGEU || (mode == 4 && (op == 0x06 || op == 0x07))) ? 1u : 0u; }   // (mode 4: DIV / REM are the signed variants of DIVU / REMU)

#pragma pack(push, 1)
struct PackedRow { uint64_t cycle, pc; uint32_t instruction; uint64_t registers[16]; uint32_t bound_bits[16]; uint8_t bound_tag[16]; uint64_t bound_payload[16]; uint8_t reg_state[16]; };
#pragma pack(pop)

// Public inputs of a proof: observed by the transcript first and carried in the proof header.
struct Public {
  uint64_t n_real = 0;       // executed rows (ExecutionResult.cycles); the trace is padded to N = max(8, next power of two)
  uint32_t deferred = 0;     // VMConfig.enable_deferred_model: the opcode semantics of the AIR are those of the default mode only
  uint64_t entry = 0x1000;   // header.entry_point: pc of row 0
  F prog[4] = {0, 0, 0, 0};  // digest of the program blob
  F io[4] = {0, 0, 0, 0};    // digest of (inputs, outputs, halt reason, cycles)
  const uint8_t* blob = nullptr; size_t blob_len = 0;   // the program itself (prover side): its code words are the instruction ROM; carried in the proof
  // Boundary states (format v4): the 68 state columns (cycle, pc limbs, 48 register limbs, 16 storage states) of row 0 and of row
  // n_real - 1.  The prover reads them off its main trace and puts them in the header; the AIR pins both rows to them.  A proof of a
  // WHOLE run must start in the VM's initial state (verify(): check 7); a SEGMENT of a run starts where its predecessor ended
  // (verify_chain()).
  F first[68] = {0}, last[68] = {0};
  // ---- mode 2 (deferred == 2): the I/O argument.  The tapes and the halt reason in the clear (the proof carries them; their digest with the run's cycle
  // count is `io`); for a SEGMENT, the ecalls the run made before its first row (raw counts: the prover's input); cnt_first / cnt_last = (oc, ic) at the
  // first / last row, read off the main trace by the prover and carried in the header like the boundary states
  const uint64_t* inputs = nullptr; size_t n_in = 0; const uint64_t* outputs = nullptr; size_t n_out = 0;
  uint32_t halt_kind = 2; uint64_t halt_code = 0;
  uint64_t writes_before = 0, reads_before = 0;
  F cnt_first[2] = {0, 0}, cnt_last[2] = {0, 0};
  // ---- mode 3: the touched memory cells (aligned 8-byte cells, strictly increasing addresses below 2^40): final bytes (little-endian in `bytes`) and the time of the
  // last access (cycle + 1).  The prover reads them off its memory replay (main_trace) and the proof carries them; the verifier forms both ends of the memory check.
  struct Cell { uint64_t addr, bytes; uint32_t t; };
  std::vector<Cell> cells;
  // ---- mode 4: the hash syscalls of the run, in order (the proof carries them; the verifier recomputes every digest): per call the cells it touches in increasing address
  // order, each with its bytes BEFORE the call and the time of its previous access
  struct HashCall { uint64_t cycle, in_ptr, len, out_ptr; uint32_t kind; std::vector<Cell> cells; };
  std::vector<HashCall> hcalls;
  // ---- mode 4: the wide tape — the wide-arithmetic rows whose operands have bits above 40, in order: the row's cycle, rs1, rs2 (raw 64 bits), the opcode 3..7
  struct WideRec { uint64_t cycle, a, b; uint32_t op; };
  std::vector<WideRec> wrecs;
  // ---- prover parameters (round 5): FRI queries and grinding bits, 0 = the defaults.  Carried in the header (words 4 and 6) and observed by the transcript with it.
  uint32_t fri = 0;          // num_queries | pow_bits << 16
  int num_queries() const { return (fri & 0xFFFF) ? (int)(fri & 0xFFFF) : 50; }
  int pow_bits() const { return (fri >> 16) ? (int)(fri >> 16) : 12; }
  int mode() const { return (int)deferred; }
  bool has_io() const { return deferred >= 2; }
  bool has_mem() const { return deferred >= 3; }
  bool has_wide() const { return deferred == 4; }
};
// (mode 3) the bytes of cell `addr` (a multiple of 8) in the VM's INITIAL memory: the code words at 0x1000, the data section right behind them (vm.rs:153-170), zero elsewhere
static uint64_t image_cell(const uint8_t* blob, size_t n, uint64_t addr) {
  if (!blob || n < 32) return 0;
  auto le32 = [&](size_t at) { return (uint64_t)blob[at] | ((uint64_t)blob[at + 1] << 8) | ((uint64_t)blob[at + 2] << 16) | ((uint64_t)blob[at + 3] << 24); };
  const uint64_t code_size = le32(16), data_size = le32(20);
  if (32 + code_size + data_size > n) return 0;
  uint64_t v = 0;
  for (int k = 0; k < 8; k++) { const uint64_t a = addr + k; if (a >= 0x1000 && a - 0x1000 < code_size + data_size) v |= (uint64_t)blob[32 + (a - 0x1000)] << (8 * k); }
  return v;
}
// ---- (mode 4) hash calls: what the verifier recomputes.  The digests are the oracle's own (zkir_oracle.cpp: the functions behind syscalls 3 / 5 / 6, pinned by the reference's KATs).
extern "C" { void zo_sha256(const uint8_t* d, size_t n, uint32_t out_words[8]); void zo_keccak256(const uint8_t* d, size_t n, uint8_t out[32]); void zo_blake3(const uint8_t* d, size_t n, uint8_t out[32]); }
static const uint64_t HASH_MAX_LEN = 1u << 20;                  // a proof states hash calls of up to 1 MiB of input (the record's length word is one field element)
static inline bool hash_call_in_range(uint64_t in_ptr, uint64_t len, uint64_t out_ptr, uint32_t kind) {
  return (kind == 3 || kind == 5 || kind == 6) && len <= HASH_MAX_LEN && in_ptr < (1ull << 40) && in_ptr + len <= (1ull << 40) && out_ptr < (1ull << 40) && out_ptr + 32 <= (1ull << 40);
}
// the aligned 8-byte cells under [in, in + len) and [out, out + 32), ascending, each once
static void hash_call_cells(uint64_t in_ptr, uint64_t len, uint64_t out_ptr, std::vector<uint64_t>& addrs) {
  addrs.clear();
  if (len) for (uint64_t a = in_ptr & ~7ull; a < in_ptr + len; a += 8) addrs.push_back(a);
  for (uint64_t a = out_ptr & ~7ull; a < out_ptr + 32; a += 8) addrs.push_back(a);
  std::sort(addrs.begin(), addrs.end());
  addrs.erase(std::unique(addrs.begin(), addrs.end()), addrs.end());
}
// the 32 bytes a hash syscall leaves at out .. out + 32: SHA-256 writes its eight big-endian-parsed words with write_u32 (little-endian), crypto.rs:251-254; Keccak-256 / BLAKE3 the digest's bytes in order
static void hash_output_bytes(uint32_t kind, const uint8_t* msg, size_t len, uint8_t out[32]) {
  if (kind == 3) { uint32_t h[8]; zo_sha256(msg, len, h); for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(h[i] >> (8 * k)); }
  else if (kind == 5) zo_keccak256(msg, len, out);
  else zo_blake3(msg, len, out);
}
// the bytes of every touched cell AFTER the call (reads come first, crypto.rs:232-235: the message is read out of the OLD bytes); cells must be hash_call_cells' list
template <class Call>
static void hash_call_new_bytes(const Call& c, std::vector<uint64_t>& nb) {
  auto index_of = [&](uint64_t cell) { size_t lo = 0, hi = c.cells.size(); while (lo + 1 < hi) { const size_t m = (lo + hi) / 2; if (c.cells[m].addr <= cell) lo = m; else hi = m; } return lo; };
  std::vector<uint8_t> msg((size_t)c.len);
  for (uint64_t k = 0; k < c.len; k++) { const uint64_t a = c.in_ptr + k; msg[(size_t)k] = (uint8_t)(c.cells[index_of(a & ~7ull)].bytes >> (8 * (a & 7))); }
  uint8_t d[32];
  hash_output_bytes(c.kind, msg.data(), msg.size(), d);
  nb.resize(c.cells.size());
  for (size_t i = 0; i < c.cells.size(); i++) nb[i] = c.cells[i].bytes;
  for (int k = 0; k < 32; k++) { const uint64_t a = c.out_ptr + k; const size_t i = index_of(a & ~7ull); const int sh = 8 * (int)(a & 7); nb[i] = (nb[i] & ~(0xFFull << sh)) | ((uint64_t)d[k] << sh); }
}
// (mode 4) the code segment's BOUNDARY CELL: when code_size % 8 == 4 the last code word shares its cell with the first four data bytes; CODE_BASE (a cell inside the code:
// refused anyway) when there is none
static uint64_t boundary_cell(const uint8_t* blob, size_t n) {
  if (!blob || n < 32) return 0x1000;
  const uint64_t code_size = (uint64_t)blob[16] | ((uint64_t)blob[17] << 8) | ((uint64_t)blob[18] << 16) | ((uint64_t)blob[19] << 24);
  return code_size % 8 == 4 ? 0x1000 + code_size - 4 : 0x1000;
}
static inline bool is_low_window(int v) { return v <= 3 || v == 8 || v == 9 || v == 12 || v == 14; }   // the windows that touch bytes 0..3 of their cell
static const int N_STATE = 68;
static int padded_log_n(uint64_t n_real) { int k = 3; while (((uint64_t)1 << k) < n_real) k++; return k; }

// digest of a byte string: Poseidon2 sponge over [len as four 16-bit pieces] ++ [little-endian 16-bit halfwords, zero-padded]
static void digest_bytes(const uint8_t* b, size_t n, F out[DIGEST]) {
  std::vector<F> e;
  for (int i = 0; i < 4; i++) e.push_back((F)(((uint64_t)n >> (16 * i)) & 0xFFFF));
  for (size_t i = 0; i < n; i += 2) e.push_back((F)b[i] | (i + 1 < n ? (F)b[i + 1] << 8 : 0));
  hash_elems(e.data(), e.size(), out);
}

static inline F opclass_of(uint32_t op, int mode = 0) {
  if (op == OP_ECALL && mode >= 2) return K_ECALL;
  if (op == OP_EBREAK && mode >= 2) return K_EBREAK;
  if (mode >= 3 && is_load(op)) return K_LD;
  if (mode >= 3 && is_store(op)) return K_ST;
  if (mode >= 3 && is_logic(op)) return K_LG;
  if (mode >= 3 && is_shift(op)) return K_SH;
  if (mode >= 3 && op == OP_MUL) return K_MU;
  if (mode == 4 && is_wide(op)) return K_WA;
  switch (op) {
    case OP_ADD: return K_ADD; case OP_ADDI: return K_ADDI; case OP_BEQ: case OP_BNE: return K_BRE; case OP_JAL: return K_JAL; case OP_SUB: return K_SUB;
    case OP_BLTU: case OP_BGEU: case OP_BLT: case OP_BGE: return K_BRU; case OP_SEQ: case OP_SNE: return K_SE;
    case OP_SLTU: case OP_SGEU: case OP_SLT: case OP_SGE: return K_SU;
    case OP_JALR: return K_JALR; case OP_CMOV: case OP_CMOVNZ: return K_CMN; case OP_CMOVZ: return K_CMZ; default: return K_OTH;
  }
}
// The instruction ROM of a program blob (Program::to_bytes layout, program.rs:170-214,300-346): code word t sits at pc = 0x1000 + 4 t
// (VM::new loads the code at CODE_BASE whatever the entry point, vm.rs:153-160); tuple = (pc limbs 20/20/24, op, fa, fb, fc, fhi, s, opclass).
struct Rom { std::vector<F> rows; size_t n = 0; uint64_t entry = 0; bool ok = false; const F* row(size_t t) const { return &rows[t * N_TUPLE]; } };
static inline void rom_tuple(uint64_t pc, uint32_t w, F out[N_TUPLE], int mode = 0) {
  out[0] = (F)(pc & 0xFFFFF); out[1] = (F)((pc >> 20) & 0xFFFFF); out[2] = (F)(pc >> 40);
  out[3] = w & 0x7F; out[4] = (w >> 7) & 0xF; out[5] = (w >> 11) & 0xF; out[6] = (w >> 15) & 0xF; out[7] = w >> 19; out[8] = w >> 31; out[9] = opclass_of(w & 0x7F, mode); out[10] = variant_bit(w & 0x7F, mode);
}
static Rom rom_from_blob(const uint8_t* b, size_t n, int mode = 0) {
  Rom r;
  auto le32 = [&](size_t at) { return (uint32_t)b[at] | ((uint32_t)b[at + 1] << 8) | ((uint32_t)b[at + 2] << 16) | ((uint32_t)b[at + 3] << 24); };
  if (!b || n < 32) return r;
  r.entry = le32(12);
  const uint64_t code_size = le32(16);
  if (code_size % 4 || 32 + code_size > n) return r;
  r.n = code_size / 4; r.rows.resize(r.n * N_TUPLE);
  for (size_t t = 0; t < r.n; t++) rom_tuple(0x1000 + 4 * (uint64_t)t, le32(32 + 4 * t), &r.rows[t * N_TUPLE], mode);
  r.ok = true;
  return r;
}

static inline void reg_limbs(uint64_t v, uint8_t st, F out[3]) {
  const int bits = st ? 30 : 20;
  const uint64_t mask = (1ull << bits) - 1;
  out[0] = (F)(v & mask); out[1] = (F)((v >> bits) & mask); out[2] = (F)(v >> (2 * bits));
}

// col-major out[W_MAIN][N]
static void main_trace(const PackedRow* rows, size_t n_real, const Public& pub, std::vector<F>& out, std::vector<Public::Cell>* cells_out = nullptr, std::vector<Public::HashCall>* hcalls_out = nullptr,
                       std::vector<Public::WideRec>* wrecs_out = nullptr) {
  const int log_n = padded_log_n(n_real);
  const size_t N = (size_t)1 << log_n;
  const int mode = pub.mode();
  out.assign((size_t)logical_width(mode) * N, 0);
  auto col = [&](int k) { return out.data() + (size_t)k * N; };
  const bool D = mode == 1, IO = mode >= 2;
  uint64_t oc = pub.writes_before, reads = pub.reads_before;          // mode 2: WRITE / READ ecalls executed so far (syscall.rs:110-121)
  std::map<uint64_t, std::pair<uint64_t, uint32_t>> memory;           // mode 3: the replayed memory, cell address -> (bytes, time of the last access); untouched cells hold the program image
  const uint64_t Bcell = boundary_cell(pub.blob, pub.blob_len);       // (mode 4)
  if (hcalls_out) hcalls_out->clear();
  if (wrecs_out) wrecs_out->clear();
  for (size_t i = 0; i < N; i++) {
    const bool pad = i >= n_real;
    const PackedRow& r = rows[pad ? n_real - 1 : i];
    const bool last = i + 1 >= n_real;                      // the last executed row and every padding row: no successor to describe
    col(C_CYCLE)[i] = (F)((pad ? r.cycle + (i - (n_real - 1)) : r.cycle) % P);      // padding keeps counting
    const F pc[3] = {(F)(r.pc & 0xFFFFF), (F)((r.pc >> 20) & 0xFFFFF), (F)(r.pc >> 40)};
    for (int l = 0; l < 3; l++) col(C_PC + l)[i] = pc[l];
    const uint32_t w = r.instruction;
    const F op = w & 0x7F, fa = (w >> 7) & 0xF, fb = (w >> 11) & 0xF, fc = (w >> 15) & 0xF, fhi = w >> 19, s = w >> 31;
    col(C_OP)[i] = op; col(C_FA)[i] = fa; col(C_FB)[i] = fb; col(C_FC)[i] = fc; col(C_FHI)[i] = fhi; col(C_S)[i] = s;
    F limb[16][3];
    for (int g = 0; g < 16; g++) {
      reg_limbs(r.registers[g], r.reg_state[g], limb[g]);
      for (int l = 0; l < 3; l++) col(C_LIMB + 3 * g + l)[i] = limb[g][l];
      col(C_STATE + g)[i] = r.reg_state[g];
    }
    int cls = pad ? K_PAD : last ? K_HALT : K_OTH;
    if (cls == K_OTH && !D) cls = (int)opclass_of(op, mode);
    if (cls == K_EBREAK) cls = K_OTH;                         // (an EBREAK that is not the halt row — no honest run has one: the row runs as "other" and constraint 4 fails on it)
    if (cls == K_OTH && D && (opclass_of(op) == K_BRE || opclass_of(op) == K_BRU || opclass_of(op) == K_JAL || opclass_of(op) == K_JALR))
      cls = K_OJ;                                             // deferred mode: no opcode semantics, but "other" is sequential — branches and jumps run as the free-pc class
    if (cls == K_LD) col(C_KLD)[i] = 1;                       // (mode 3: loads and stores, the bitwise opcodes)
    else if (cls == K_ST) col(C_KST)[i] = 1;
    else if (cls == K_LG) col(C_KLG)[i] = 1;
    else if (cls == K_SH) col(C_KSH)[i] = 1;
    else if (cls == K_MU) col(C_KMU)[i] = 1;
    else if (cls != K_ECALL) col(kcol(cls))[i] = 1;           // (mode 2: the ecall class has no column — it is the sum of the four syscall flags)
    col(C_OPC)[i] = opclass_of(op, mode);                           // of the WORD, whatever class the row runs as (halt / pad rows, deferred mode)
    const bool branch = cls == K_BRE || cls == K_BRU;
    const uint32_t tc = (branch || cls == K_ST) ? fa : fc;  // second operand: rs2 = field c, but B-type and S-type words have rs1 in field a (rs2 in field b)
    if (fb) col(C_SELB + fb - 1)[i] = 1;
    if (tc) col(C_SELC + tc - 1)[i] = 1;
    const F* xb = limb[fb]; const F* xc = limb[tc];         // register 0 reads as zero limbs: its columns are constrained to zero
    for (int l = 0; l < 3; l++) { col(C_XB + l)[i] = xb[l]; col(C_XC + l)[i] = xc[l]; }
    F ne = 0;
    for (int l = 0; l < 3; l++) if (!ne && xb[l] != xc[l]) { ne = 1; col(C_IV + l)[i] = finv(fsub(xb[l], xc[l])); }
    col(C_NE)[i] = ne;
    // the 40-bit difference of the masked operands and its borrows: xb - xc (SUB, SLTU / SGEU: rs1 = field b), xc - xb (BLTU / BGEU: rs1 = field a)
    //   (v5) ordered comparisons, signed or not: the high limbs enter BIASED, t = limb + 2^19 sgn - 2^20 (sign bit) — the limb of value XOR 2^39
    //   when the comparison is signed (Value40::signed_lt, value.rs:710-716), the limb itself when it is not; u = (ta, tb) is the row's SECOND
    //   range-checked pair (chunks C_RC2 ..), which forces the sign bits; the borrow out of the biased difference is the comparison
    F z[2] = {0, 0}, c0 = 0, c1 = 0, u[2] = {0, 0}, sa = 0, sb = 0;
    const F g = variant_bit(op, mode);
    col(C_G)[i] = g;
    if (cls == K_SUB || cls == K_SU || cls == K_BRU) {
      const F* a = cls == K_BRU ? xc : xb; const F* b = cls == K_BRU ? xb : xc;
      const F sgn = cls == K_SU ? g : cls == K_BRU ? 1 - g : 0;
      if (sgn) { sa = a[1] >> 19; sb = b[1] >> 19; }
      const int64_t ta = (int64_t)a[1] + ((int64_t)sgn << 19) - ((int64_t)sa << 20), tb = (int64_t)b[1] + ((int64_t)sgn << 19) - ((int64_t)sb << 20);
      const int64_t v0 = (int64_t)a[0] - b[0]; c0 = v0 < 0; z[0] = (F)(v0 + ((int64_t)c0 << 20));
      const int64_t v1 = ta - tb - c0; c1 = v1 < 0; z[1] = (F)(v1 + ((int64_t)c1 << 20));
      if (cls != K_SUB) { u[0] = (F)ta; u[1] = (F)tb; }
    }
    col(C_SB)[i] = sb;
    F rc2[4] = {u[0] & (RC_TABLE - 1), u[0] >> RC_BITS, u[1] & (RC_TABLE - 1), u[1] >> RC_BITS};
    const F flag = (cls == K_BRE || cls == K_SE) ? 1 - ne : (cls == K_BRU || cls == K_SU) ? c1 : 0;
    const F pol = fsub(fsub(op, family_base(cls)), 2 * g);  // op = base + 2 g + pol: 0 / 1 inside a family; the opcode itself on every other row (fx is unused there)
    const F fx = fsub(fadd(flag, pol), fmul(2, fmul(pol, flag)));
    col(C_FLAG)[i] = flag; col(C_FX)[i] = fx;
    const F tk = branch ? fx : 0;
    col(C_TK)[i] = tk;
    const F imm17 = fc + 16 * fhi, im0 = imm17 - (s << 17) + (s << 20), im1 = s * 0xFFFFF;        // sign-extended imm17 mod 2^40, as two 20-bit limbs
    const F lo20 = fb + 16 * fc + 256 * fhi - (s << 20);                                           // low 20 bits of off21
    const F dl0 = cls == K_JAL ? lo20 : tk ? im0 : 4;
    col(C_DL0)[i] = dl0;
    const F se = (tk || cls == K_JAL) ? s : 0;
    col(C_SE)[i] = se;
    F y[3] = {0, 0, 0};
    int rd = -1;
    // (v6) nz = [rs2 != 0] over the raw 64 bits, stated for EVERY row on the sum of the three limbs of xc (each limb is in range — the third
    // one by the range check of every written y2 below — so the sum is zero only if all of them are); q = "a conditional move whose condition holds"
    const F sx = (F)(((uint64_t)xc[0] + xc[1] + xc[2]) % P), nz = sx != 0;
    col(C_NZ)[i] = nz; col(C_IVZ)[i] = nz ? finv(sx) : 0;
    const F q = cls == K_CMN ? nz : cls == K_CMZ ? 1 - nz : 0;
    col(C_Q)[i] = q;
    if (cls == K_ADD || cls == K_ADDI) {
      const F b0 = cls == K_ADD ? xc[0] : im0, b1 = cls == K_ADD ? xc[1] : im1;
      const uint64_t v0 = (uint64_t)xb[0] + b0; c0 = (F)(v0 >> 20); y[0] = (F)(v0 & 0xFFFFF);
      const uint64_t v1 = (uint64_t)xb[1] + b1 + c0; c1 = (F)(v1 >> 20); y[1] = (F)(v1 & 0xFFFFF);
      rd = fa;
    } else if (cls == K_JAL || cls == K_JALR) {                 // the link pc + 4 (execute.rs:639-658)
      const uint64_t v0 = (uint64_t)pc[0] + 4; c0 = (F)(v0 >> 20); y[0] = (F)(v0 & 0xFFFFF);
      const uint64_t v1 = (uint64_t)pc[1] + c0; c1 = (F)(v1 >> 20); y[1] = (F)(v1 & 0xFFFFF);
      y[2] = pc[2] + c1;
      rd = fa;
    } else if (cls == K_SUB) { y[0] = z[0]; y[1] = z[1]; rd = fa; }                  // execute.rs:65-77: Value40 wrapping_sub
    else if (cls == K_SE || cls == K_SU) { y[0] = fx; rd = fa; }                      // execute.rs:373-431: the comparison as 0 / 1
    else if (cls == K_CMN || cls == K_CMZ) { y[0] = xb[0]; y[1] = xb[1]; y[2] = xb[2]; if (q) rd = fa; }   // execute.rs:434-472: rd = rs1 (raw) if the condition holds, nothing changes otherwise
    bool sh_row = false; F sh_lo[4] = {0, 0, 0, 0}, sh_c[4] = {0, 0, 0, 0};
    if (cls == K_MU) {                                        // (mode 3) MUL: Value40::wrapping_mul of the masked operands (execute.rs:79-99), in 10-bit chunks
      sh_row = true;                                          // (the row's range groups are filled like a shift row's: R0..R3 = the result's chunks, R4..R7 = a's)
      const uint64_t a = (uint64_t)xb[0] | ((uint64_t)xb[1] << 20), b = (uint64_t)xc[0] | ((uint64_t)xc[1] << 20);
      F bc[4], carry = 0;
      for (int k = 0; k < 4; k++) { sh_c[k] = (F)((a >> (10 * k)) & 1023); bc[k] = (F)((b >> (10 * k)) & 1023); col(C_MA + k)[i] = sh_c[k]; col(C_PIECE + k)[i] = bc[k]; }
      uint64_t res = 0;
      for (int k = 0; k < 4; k++) {
        uint64_t t = carry;
        for (int j = 0; j <= k; j++) t += (uint64_t)sh_c[j] * bc[k - j];
        sh_lo[k] = (F)(t & 1023); carry = (F)(t >> 10);
        res |= (uint64_t)sh_lo[k] << (10 * k);
        col(C_PIECE + 4 + k)[i] = carry & 1023;
        if (k == 1) col(C_ME)[i] = carry >> 10;
        if (k == 2) { col(C_ME + 1)[i] = (carry >> 10) & 1; col(C_ME + 2)[i] = carry >> 11; }
        if (k == 3) col(C_PIECE + 8)[i] = carry >> 10;
      }
      y[0] = (F)(res & 0xFFFFF); y[1] = (F)(res >> 20); y[2] = 0;
      rd = fa;
    }
    if (cls == K_WA && (xb[2] || xc[2])) {                    // (mode 4 d) an operand with bits above 40: the row is proven through the WIDE TAPE — ot = 1, every chunk column zero,
      sh_row = true;                                          // the verifier computes what is written from the record (cycle, rs1, rs2, opcode) the proof carries
      const uint64_t a = (uint64_t)xb[0] | ((uint64_t)xb[1] << 20) | ((uint64_t)xb[2] << 40), b = (uint64_t)xc[0] | ((uint64_t)xc[1] << 20) | ((uint64_t)xc[2] << 40);
      const uint64_t res = wide_result(op, a, b);
      col(C_OT)[i] = 1;
      y[0] = (F)(res & 0xFFFFF); y[1] = (F)((res >> 20) & 0xFFFFF); y[2] = (F)(res >> 40);
      rd = fa;
      if (wrecs_out) wrecs_out->push_back(Public::WideRec{r.cycle, a, b, op});
    } else
    if (cls == K_WA) {                                        // (mode 4) MULH DIVU REMU DIV REM on operands below 2^40 (execute.rs:101-183): F1 F2 + ADD = LO + 2^40 HI in 10-bit chunks
      sh_row = true;                                          // (R0..R3 = LO's chunks, R4..R7 = F1's)
      const uint64_t M40 = (1ull << 40) - 1;
      const uint64_t a = (uint64_t)xb[0] | ((uint64_t)xb[1] << 20), b = (uint64_t)xc[0] | ((uint64_t)xc[1] << 20);   // (the top limbs must be zero: the constraints say so, a run that breaks it has no proof)
      const bool mulh = op == 0x03, quot = op == 0x04 || op == 0x06;
      col(C_OM)[i] = mulh; col(C_OD)[i] = !mulh && quot; col(C_ORR)[i] = !mulh && !quot;
      uint64_t f1, add, res;
      if (mulh) { f1 = a; add = 0; res = (uint64_t)(((unsigned __int128)a * b) >> 40) & M40; }
      else { const uint64_t qv = b ? a / b : 0, rv = b ? a % b : 0; f1 = qv; add = rv; res = quot ? qv : rv; }       // (b = 0 never is a row: the VM stops with DivisionByZero)
      F f1c[4], f2c[4], addc[4];
      for (int k = 0; k < 4; k++) { f1c[k] = (F)((f1 >> (10 * k)) & 1023); f2c[k] = (F)((b >> (10 * k)) & 1023); addc[k] = (F)((add >> (10 * k)) & 1023); sh_c[k] = f1c[k]; col(C_GF + k)[i] = f1c[k]; col(C_PIECE + k)[i] = f2c[k]; }
      uint64_t carry = 0, cs[7], out[7];
      for (int k = 0; k < 7; k++) {                           // position k of F1 F2 + ADD
        uint64_t t = carry + (k < 4 ? addc[k] : 0);
        for (int j = 0; j < 4; j++) if (k - j >= 0 && k - j < 4) t += (uint64_t)f1c[j] * f2c[k - j];
        out[k] = t & 1023; carry = t >> 10; cs[k] = carry;
      }
      for (int k = 0; k < 4; k++) sh_lo[k] = (F)out[k];      // LO = the product's low 40 bits (MULH) / the dividend (divisions)
      const F g4[4] = {mulh ? (F)out[4] : addc[0], mulh ? (F)out[5] : addc[1], mulh ? (F)out[6] : addc[2], mulh ? (F)cs[6] : addc[3]};   // HI's chunks (HI_3 = the last carry) / the remainder's
      col(C_PIECE + 4)[i] = (F)cs[0];
      col(C_PIECE + 5)[i] = (F)(cs[1] & 1023); col(C_WE)[i] = (F)(cs[1] >> 10);
      col(C_PIECE + 6)[i] = (F)(cs[2] & 1023); col(C_WE + 1)[i] = (F)((cs[2] >> 10) & 1); col(C_WE + 2)[i] = (F)(cs[2] >> 11);
      col(C_PIECE + 7)[i] = g4[0]; col(C_PIECE + 8)[i] = g4[1]; col(C_X)[i] = g4[2]; col(C_X + 1)[i] = g4[3];
      if (mulh) {
        for (int k = 0; k < 3; k++) { col(C_X + 2 + k)[i] = (F)(cs[3 + k] & 1023); col(C_WE + 3 + 2 * k)[i] = (F)((cs[3 + k] >> 10) & 1); col(C_WE + 4 + 2 * k)[i] = (F)(cs[3 + k] >> 11); }
      } else {                                                // d = b - r - 1 in chunks, its borrow between the limbs in e4
        const uint64_t d = (b - add - 1) & M40;
        for (int k = 0; k < 4; k++) col(C_X + 2 + k)[i] = (F)((d >> (10 * k)) & 1023);
        col(C_WE + 3)[i] = (F)(((b & 0xFFFFF) < (add & 0xFFFFF) + 1) ? 1 : 0);
      }
      y[0] = (F)(res & 0xFFFFF); y[1] = (F)(res >> 20); y[2] = 0;
      rd = fa;
    }
    if (cls == K_SH) {                                        // (mode 3) SLL SRL SRA SLLI SRLI SRAI = 0x18 + (0 / 1 / 2) + 3 si (execute.rs:284-358)
      sh_row = true;
      const uint32_t which = (op - 0x18) % 3, si = (op - 0x18) / 3;
      const uint64_t a = (uint64_t)xb[0] | ((uint64_t)xb[1] << 20);
      const uint32_t amount = si ? ((w >> 15) & 0xFF) : (uint32_t)(xc[0] & 63);
      col(C_SA)[i] = which == 2; col(C_SI)[i] = si; col(C_SH)[i] = amount;
      // t = the exponent of a 2^t = H 2^40 + L: sh for a left shift (anything from 40 on shifts everything out: t = 40 says that too), 40 - sh for a right shift (0 from sh = 40 on)
      const uint32_t t = which == 0 ? (amount < 40 ? amount : 40) : (amount < 40 ? 40 - amount : 0);
      const uint32_t d = which == 0 ? amount - t : amount - (40 - t);
      const uint32_t u = t / 10, v = t % 10;
      col((which == 0 ? C_UL : C_UR) + u)[i] = 1; col(C_V + v)[i] = 1;
      F hi[4], m[5];
      for (int k = 0; k < 4; k++) {
        sh_c[k] = (F)((a >> (10 * k)) & 1023);
        const uint32_t pr = sh_c[k] << v;
        col(C_PR + k)[i] = pr; sh_lo[k] = pr & 1023; hi[k] = pr >> 10;
        col(C_PIECE + k)[i] = hi[k];
      }
      const F sb9 = sh_c[3] >> 9, sgn = (which == 2) ? sb9 : 0;
      col(C_SB9)[i] = sb9; col(C_SGN)[i] = sgn;
      col(C_PIECE + 4)[i] = 2 * (sh_c[3] & 511);
      col(C_PIECE + 6)[i] = d;
      if (si) { const F fhi_v = w >> 19; col(C_PIECE + 7)[i] = fhi_v & 15; col(C_PIECE + 5)[i] = fhi_v >> 4; }
      else { col(C_PIECE + 8)[i] = xc[0] & 1023; col(C_PIECE + 5)[i] = xc[0] >> 10; col(C_LB + 8)[i] = amount; }
      m[0] = sh_lo[0]; for (int k = 1; k < 4; k++) m[k] = sh_lo[k] + hi[k - 1]; m[4] = hi[3];
      uint64_t res = 0;
      for (int j = 0; j < 4; j++) { const int idx = which == 0 ? j - (int)u : j + 4 - (int)u; if (idx >= 0 && idx <= 4) res |= (uint64_t)m[idx] << (10 * j); }
      uint64_t ones = 0;
      if (which != 0) { ones = ((1ull << 40) - 1) & ~((1ull << t) - 1); col(C_ON)[i] = (F)(ones & 0xFFFFF); col(C_ON + 1)[i] = (F)(ones >> 20); }
      if (sgn) res |= ones;
      y[0] = (F)(res & 0xFFFFF); y[1] = (F)(res >> 20); y[2] = 0;
      rd = fa;
    }
    F lg_a9 = 0; bool lg_row = false;
    if (cls == K_LG) {                                        // (mode 3) AND OR XOR ANDI ORI XORI on the 40-bit values (execute.rs:199-282), nibble by nibble
      lg_row = true;
      const uint32_t which = (op - 0x10) % 3, li = (op - 0x10) / 3;
      col(C_OA)[i] = which == 0; col(C_OO)[i] = which == 1; col(C_LI)[i] = li;
      const uint64_t a = (uint64_t)xb[0] | ((uint64_t)xb[1] << 20), b = li ? ((uint64_t)im0 | ((uint64_t)im1 << 20)) : ((uint64_t)xc[0] | ((uint64_t)xc[1] << 20));
      const uint64_t rr = which == 0 ? (a & b) : which == 1 ? (a | b) : (a ^ b);
      for (int k = 0; k < N_NIB; k++) {
        const F ak = (F)((a >> (4 * k)) & 15);
        if (k < N_PIECE) col(C_PIECE + k)[i] = ak; else lg_a9 = ak;
        col(C_LB + k)[i] = (F)((b >> (4 * k)) & 15); col(C_LR + k)[i] = (F)((rr >> (4 * k)) & 15);
      }
      y[0] = (F)(rr & 0xFFFFF); y[1] = (F)(rr >> 20); y[2] = 0;
      rd = fa;
    }
    bool mem_row = false;
    F mem_z[2] = {0, 0}, mem_dt = 0;
    if (cls == K_LD || cls == K_ST) {                         // (mode 3) loads and stores (execute.rs:477-575): address = rs1 + sext(imm17) mod 2^64, which must stay below 2^40
      mem_row = true;
      const F* base = cls == K_LD ? xb : xc;                  // loads: rs1 = field b; stores: rs1 = field a (operand c), the stored register rs2 = field b (operand b)
      const uint64_t v0 = (uint64_t)base[0] + im0; c0 = (F)(v0 >> 20); mem_z[0] = (F)(v0 & 0xFFFFF);
      const uint64_t v1 = (uint64_t)base[1] + im1 + c0; c1 = (F)(v1 >> 20); mem_z[1] = (F)(v1 & 0xFFFFF);
      const uint64_t v2 = (uint64_t)base[2] + (uint64_t)s * 0xFFFFFF + c1;
      col(C_CM2)[i] = (F)(v2 >> 24);                          // (v2 & 0xFFFFFF must be 0: an address of 2^40 or more has no proof in this AIR)
      const uint64_t ea = (uint64_t)mem_z[0] | ((uint64_t)mem_z[1] << 20);
      const int width = mem_width(op), off = (int)(ea & 7), v = win_of(width, off - off % width);
      const uint64_t cell = ea - off, mask = width == 8 ? ~0ull : ((1ull << (8 * width)) - 1);
      col(C_E + v)[i] = 1;
      auto it = memory.find(cell);
      const uint64_t ob = it == memory.end() ? image_cell(pub.blob, pub.blob_len, cell) : it->second.first;
      const uint32_t told = it == memory.end() ? 0 : it->second.second;
      for (int k = 0; k < 8; k++) col(C_OB + k)[i] = (F)((ob >> (8 * k)) & 0xFF);
      col(C_TOLD)[i] = told;
      mem_dt = (F)(r.cycle - told);                           // time written = cycle + 1 > told
      uint64_t window, nb = ob;
      if (cls == K_LD) {
        window = (ob >> (8 * off)) & mask;
        uint64_t val = window;
        const bool sgb = op == 0x30, sgh = op == 0x32;
        const F tb = width <= 2 ? (F)((window >> (8 * width - 1)) & 1) : 0;
        if ((sgb || sgh) && tb) val |= ~mask;                 // LB / LH sign-extend to 64 bits (execute.rs:477-511)
        col(C_SGB)[i] = sgb; col(C_SGH)[i] = sgh; col(C_TB)[i] = tb; col(C_SX)[i] = (sgb || sgh) ? tb : 0;
        reg_limbs(val, 0, y);
        rd = fa;
      } else {
        window = r.registers[fb];                             // the raw 64-bit register (execute.rs:548-575: value & mask of the width)
        nb = (ob & ~(mask << (8 * off))) | ((window & mask) << (8 * off));
        reg_limbs(window & mask, 0, y);                       // (nothing is written: y only satisfies the window equations below)
      }
      memory[cell] = std::make_pair(nb, (uint32_t)(r.cycle + 1));
      if (mode == 4 && cls == K_ST && is_low_window(v)) {     // (mode 4) a store into the low half of a cell: not the boundary cell's — nb = delta iws = 1 with delta = cell - B (as a field element) != 0
        const F delta = fadd(fsub(fsub(mem_z[0], (F)off), (F)(Bcell & 0xFFFFF)), fmul((F)(1u << 20), fsub(mem_z[1], (F)((Bcell >> 20) & 0xFFFFF))));
        if (delta) { col(C_IWS)[i] = finv(delta); col(C_NB)[i] = 1; }
      }
      // the nine pieces of the window value: the whole stored register on stores, the zero-extended loaded window on loads
      F pc9[9] = {(F)(window & 0xFF), (F)((window >> 8) & 0xFF), (F)((window >> 16) & 0xF), (F)((window >> 20) & 0xF), (F)((window >> 24) & 0xFF), (F)((window >> 32) & 0xFF),
                  (F)((window >> 40) & 0xFF), (F)((window >> 48) & 0xFF), (F)((window >> 56) & 0xFF)};
      if (cls == K_LD && width <= 2) pc9[7] = (F)(2 * ((window >> (8 * (width - 1))) & 0x7F));   // d6 = twice the low seven bits of the top byte: the byte table then says they ARE seven bits
      for (int k = 0; k < N_PIECE; k++) col(C_PIECE + k)[i] = pc9[k];
    }
    if (IO) {                                                 // the counters every row shows: what happened BEFORE it
      col(C_OC)[i] = (F)(oc % P); col(C_IC)[i] = (F)((reads < pub.n_in ? reads : pub.n_in) % P);
    }
    if (cls == K_ECALL) {                                     // mode 2, an executed ecall: dispatched on R10 (syscall.rs:94-177; 0 = EXIT halts: that row is the halt row)
      const uint64_t num = r.registers[10];
      if (num == 2) { col(C_F2)[i] = 1; oc++; }                                                      // WRITE: R11 goes to the output tape, no register changes
      else if (num == 1) {                                                                           // READ: R10 <- the next input, 0 once the tape is exhausted
        if (reads < pub.n_in) { col(C_RL)[i] = 1; reg_limbs(pub.inputs[reads], 0, y); } else col(C_RE)[i] = 1;
        reads++; rd = 10;
      } else {                                                 // hash syscalls (3, 5, 6): R10 <- 0; their memory effect is stated in mode 4 only (the hash tape), a run that makes one has no mode-3 proof
        col(C_FH)[i] = 1; col(C_H0)[i] = (F)((num - 3) & 1); col(C_H1)[i] = (F)(((num - 3) >> 1) & 1); rd = 10;
        if (mode == 4 && hash_call_in_range(r.registers[11], r.registers[12], r.registers[13], (uint32_t)num)) {
          Public::HashCall hc{r.cycle, r.registers[11], r.registers[12], r.registers[13], (uint32_t)num, {}};
          std::vector<uint64_t> addrs, nbv;
          hash_call_cells(hc.in_ptr, hc.len, hc.out_ptr, addrs);
          for (const uint64_t a : addrs) {
            auto it = memory.find(a);
            hc.cells.push_back(Public::Cell{a, it == memory.end() ? image_cell(pub.blob, pub.blob_len, a) : it->second.first, it == memory.end() ? 0u : it->second.second});
          }
          hash_call_new_bytes(hc, nbv);
          for (size_t k = 0; k < addrs.size(); k++) memory[addrs[k]] = std::make_pair(nbv[k], (uint32_t)(r.cycle + 1));
          if (hcalls_out) hcalls_out->push_back(std::move(hc));
        }
      }
    }
    if (rd > 0) col(C_WR + rd - 1)[i] = 1;
    if (cls == K_OTH || (cls == K_OJ && D)) {               // any other instruction: what it wrote is read off the next row
      const PackedRow& q = rows[i + 1];
      bool first = true;
      for (int g = 1; g < 16; g++) {
        F nl[3]; reg_limbs(q.registers[g], q.reg_state[g], nl);
        if (nl[0] != limb[g][0] || nl[1] != limb[g][1] || nl[2] != limb[g][2] || q.reg_state[g] != r.reg_state[g]) {
          col(C_WR + g - 1)[i] = 1;
          if (first && !D) { y[0] = nl[0]; y[1] = nl[1]; y[2] = nl[2]; first = false; }
        }
      }
    }
    for (int l = 0; l < 3; l++) col(C_Y + l)[i] = y[l];
    if (cls == K_OTH) {                                       // (v6) the bits above 40 of what an "other" row writes are range-checked: y2 = R4 + 2^10 R5 + 2^20 R6, R7 = 64 R6 (so R6 < 16: 24 bits)
      rc2[0] = y[2] & (RC_TABLE - 1); rc2[1] = (y[2] >> RC_BITS) & (RC_TABLE - 1); rc2[2] = y[2] >> (2 * RC_BITS); rc2[3] = 64 * rc2[2];
    }
    if (mem_row) { rc2[0] = mem_dt & (RC_TABLE - 1); rc2[1] = (mem_dt >> RC_BITS) & (RC_TABLE - 1); rc2[2] = mem_dt >> (2 * RC_BITS); rc2[3] = 0; }   // (mode 3) cycle - told in three chunks: the time read is smaller than the time written
    if (lg_row) { rc2[0] = rc2[1] = rc2[2] = 0; rc2[3] = lg_a9; }                              // (mode 3) a bitwise row's tenth nibble tuple sits in the last range slot
    if (sh_row) for (int k = 0; k < 4; k++) rc2[k] = sh_c[k];                                   // (mode 3) a shift row: the chunks of the shifted value
    for (int k = 0; k < 4; k++) col(C_RC2 + k)[i] = rc2[k];
    if (cls == K_ADD || cls == K_ADDI || cls == K_JAL || cls == K_JALR || cls == K_OTH || cls == K_ECALL || (cls == K_OJ && D)) { z[0] = y[0]; z[1] = y[1]; }   // the written value's low limbs are the range-checked pair
    if (mem_row) { z[0] = mem_z[0]; z[1] = mem_z[1]; }         // (mode 3) the address's two low limbs are the range-checked pair
    col(C_RC)[i] = z[0] & (RC_TABLE - 1); col(C_RC + 1)[i] = z[0] >> RC_BITS; col(C_RC + 2)[i] = z[1] & (RC_TABLE - 1); col(C_RC + 3)[i] = z[1] >> RC_BITS;
    if (sh_row) for (int k = 0; k < 4; k++) col(C_RC + k)[i] = sh_lo[k];                        // (mode 3) a shift row: the low halves of c_i 2^v
    col(C_C0)[i] = c0; col(C_C1)[i] = c1;
    if (cls == K_JALR) {                                                                            // pc' + b0 = rs1 + sext(imm17) over (20, 20, 24)-bit limbs, mod 2^64
      const uint64_t v0 = (uint64_t)xb[0] + im0; const F d0 = (F)(v0 >> 20);
      const uint64_t v1 = (uint64_t)xb[1] + im1 + d0; const F d1 = (F)(v1 >> 20);
      const uint64_t v2 = (uint64_t)xb[2] + (uint64_t)s * 0xFFFFFF + d1; const F d2 = (F)(v2 >> 24);
      col(C_D0)[i] = d0; col(C_D1)[i] = d1; col(C_D2)[i] = d2; col(C_B0)[i] = (F)(v0 & 1);
    } else if (cls != K_OJ && cls != K_HALT && cls != K_PAD) {
      col(C_B0)[i] = sa;                                                                              // (v5) column b0 doubles as the sign bit of the first operand of an ordered comparison                                      // pc' = pc + delta over (20, 20, 24)-bit limbs, mod 2^64
      const uint64_t v0 = (uint64_t)pc[0] + dl0; const F d0 = (F)(v0 >> 20);
      const uint64_t v1 = (uint64_t)pc[1] + (uint64_t)se * 0xFFFFF + d0; const F d1 = (F)(v1 >> 20);
      const uint64_t v2 = (uint64_t)pc[2] + (uint64_t)se * 0xFFFFFF + d1; const F d2 = (F)(v2 >> 24);
      col(C_D0)[i] = d0; col(C_D1)[i] = d1; col(C_D2)[i] = d2;
    }
  }
  if (cells_out) { cells_out->clear(); for (auto& kv : memory) cells_out->push_back(Public::Cell{kv.first, kv.second.first, kv.second.second}); }   // (std::map: increasing addresses)
}
// ---------------------------------------------------------------------------------------------
// Merkle tree over the rows of a column-major matrix (leaf j = hash of column values at position j)
// ---------------------------------------------------------------------------------------------
struct Merkle {
  size_t n_leaves = 0;
  std::vector<std::vector<F>> layers;   // layers[0] = leaf digests (4 * n_leaves), ..., back() = root (4)
};
static void merkle_build(const std::vector<F>& mat, int width, size_t n, Merkle& t) {
  t.n_leaves = n;
  t.layers.clear();
  t.layers.emplace_back(4 * n);
  std::vector<F> row(width);
  for (size_t j = 0; j < n; j++) { for (int k = 0; k < width; k++) row[k] = mat[(size_t)k * n + j]; hash_elems(row.data(), width, &t.layers[0][4 * j]); }
  while (t.layers.back().size() > 4) {
    const std::vector<F>& prev = t.layers.back();
    size_t m = prev.size() / 8;
    std::vector<F> cur(4 * m);
    for (size_t i = 0; i < m; i++) compress(&prev[8 * i], &prev[8 * i + 4], &cur[4 * i]);
    t.layers.push_back(std::move(cur));
  }
}


// ---------------------------------------------------------------------------------------------
// Lookup argument (LogUp) of AIR v2: what is looked up, the multiplicities, the aux trace
// ---------------------------------------------------------------------------------------------
struct LookupParams { E alpha; E lam[N_TUPLE + 1]; E t_over_n; F n_in = 0; };   // n_in: mode 2, the length of the input tape (public: the proof carries the tape)
// mode 2: fingerprint of an I/O tuple (index on its tape, the three limbs of the value): sum_j lambda^j g_j + tag lambda^N_TUPLE, tag 2 = output tape, 3 = input tape
// (the instruction ROM's tuples carry tag 1, range values none: the three tables cannot be confused)
static inline void io_limbs(uint64_t v, F out[3]) { out[0] = (F)(v & 0xFFFFF); out[1] = (F)((v >> 20) & 0xFFFFF); out[2] = (F)(v >> 40); }
static inline E io_fingerprint(F idx, const F* limbs, F tag, const LookupParams& lp) {
  E fp = emul_f(lp.lam[N_TUPLE], tag);
  fp = eadd(fp, emul_f(lp.lam[0], idx));
  for (int j = 0; j < 3; j++) fp = eadd(fp, emul_f(lp.lam[1 + j], limbs[j]));
  return fp;
}
// (mode 3) fingerprint of a memory tuple (cell address as two 20-bit limbs, time, the cell's eight bytes): a0 + lambda a1 + lambda^2 t + sum_k lambda^(3+k) byte_k + 7 lambda^11
static inline E mem_bytes_fp(uint64_t bytes, const LookupParams& lp) { E fp = e_from(0); for (int k = 0; k < 8; k++) fp = eadd(fp, emul_f(lp.lam[3 + k], (F)((bytes >> (8 * k)) & 0xFF))); return fp; }
static inline E mem_fingerprint(F a0, F a1, F t, const E& bytes_fp, const LookupParams& lp) {
  return eadd(eadd(eadd(emul_f(lp.lam[N_TUPLE], TAG_MEM), e_from(a0)), eadd(emul_f(lp.lam[1], a1), emul_f(lp.lam[2], t))), bytes_fp);
}
static inline E tagged(F v, int tag, const LookupParams& lp) { return eadd(e_from(v), emul_f(lp.lam[N_TUPLE], (F)tag)); }   // a one-element tuple of table `tag` (0: the plain range value)
static inline E fingerprint(const F* tuple, const LookupParams& lp) {        // sum_j lambda^j f_j + lambda^10 (the tag keeps ROM entries apart from range values)
  E fp = lp.lam[N_TUPLE];
  for (int j = 0; j < N_TUPLE; j++) fp = eadd(fp, emul_f(lp.lam[j], tuple[j]));
  return fp;
}
static inline void row_tuple(const std::vector<F>& M, size_t N, size_t i, F out[N_TUPLE]) {
  static const int cols[N_TUPLE] = {C_PC, C_PC + 1, C_PC + 2, C_OP, C_FA, C_FB, C_FC, C_FHI, C_S, C_OPC, C_G};
  for (int j = 0; j < N_TUPLE; j++) out[j] = M[(size_t)cols[j] * N + i];
}
// Multiplicities of the two tables over ALL N rows of the matrix (padding rows repeat the last executed row's instruction and have
// y = 0).  A value that is not in its table is simply not counted — the sums then cannot match and the proof is rejected.
static inline F row_offset(const std::vector<F>& M, size_t N, size_t i) { F off = 0; for (int v = 0; v < N_WIN; v++) off += M[(size_t)(C_E + v) * N + i] * (F)win_start(v); return off; }   // (mode 3) the window's offset in its cell
static inline F row_kmem(const std::vector<F>& M, size_t N, size_t i) { return M[(size_t)C_KLD * N + i] + M[(size_t)C_KST * N + i]; }
// mem_mult (mode 3): LOW3 (1024) ++ BYTE (256) ++ NIBBLE (16)
static const int MEM_MULT = RC_TABLE + 256 + 16 + 3 * 256 + RC_TABLE, LG_BASE = RC_TABLE + 256 + 16, L6_BASE = LG_BASE + 3 * 256;   // .. ++ AND (256: entry 16 a + b) ++ OR ++ XOR ++ LOW6 (1024: entry v = the tuple (v, v & 63))
static inline bool row_shift(const std::vector<F>& M, size_t N, size_t i) { return M[(size_t)C_KSH * N + i] != 0; }
static inline bool row_mul(const std::vector<F>& M, size_t N, size_t i) { return M[(size_t)C_KMU * N + i] != 0; }
static inline bool row_wide(const std::vector<F>& M, size_t N, size_t i, int mode) { return mode == 4 && (M[(size_t)C_OM * N + i] | M[(size_t)C_OD * N + i] | M[(size_t)C_ORR * N + i] | M[(size_t)C_OT * N + i]) != 0; }
static inline bool row_shift_reg(const std::vector<F>& M, size_t N, size_t i) { return M[(size_t)C_KSH * N + i] != 0 && M[(size_t)C_SI * N + i] == 0; }
// the table a piece slot looks its value up in on a SHIFT row: 0-6 the 10-bit range table, 7 the nibble table, 8 LOW6 (with the amount) when the amount comes from a register
static inline int shift_piece_tag(int k, bool reg) { return k == 7 ? TAG_NIB : (k == 8 && reg) ? TAG_LOW6 : 0; }
static inline int row_logic_op(const std::vector<F>& M, size_t N, size_t i) { return M[(size_t)C_KLG * N + i] ? (M[(size_t)C_OA * N + i] ? 0 : M[(size_t)C_OO * N + i] ? 1 : 2) : -1; }   // 0 AND, 1 OR, 2 XOR; -1: not a bitwise row
static inline F logic_of(int which, F a, F b) { return which == 0 ? (a & b) : which == 1 ? (a | b) : (a ^ b); }
static void lookup_multiplicities(const std::vector<F>& M, size_t N, const Rom& rom, std::vector<F>& rom_mult, std::vector<F>& rc_mult, size_t* first_bad_row = nullptr, int mode = 0,
                                  std::vector<F>* mem_mult = nullptr) {
  rom_mult.assign(rom.n, 0); rc_mult.assign(RC_TABLE, 0);
  const bool MEM = mode >= 3 && mem_mult, WIDE = mode == 4;
  if (MEM) mem_mult->assign(MEM_MULT, 0);
  if (first_bad_row) *first_bad_row = (size_t)-1;
  auto bad = [&](size_t i) { if (first_bad_row && *first_bad_row == (size_t)-1) *first_bad_row = i; };
  for (size_t i = 0; i < N; i++) {
    for (int k = 0; k < N_RC; k++) {
      const F v = M[(size_t)rc_col(k) * N + i];
      if (MEM && k == 0 && row_kmem(M, N, i)) {               // a memory row's first chunk is looked up WITH the window's offset: (v, v & 7)
        if (v < (F)RC_TABLE && (v & 7) == row_offset(M, N, i)) (*mem_mult)[v]++; else bad(i);
      } else if (MEM && k == N_RC - 1 && row_logic_op(M, N, i) >= 0) {   // a bitwise row's tenth nibble tuple (a_9 = this chunk, b_9, r_9)
        const int which = row_logic_op(M, N, i); const F b = M[(size_t)(C_LB + 9) * N + i], r = M[(size_t)(C_LR + 9) * N + i];
        if (v < 16 && b < 16 && r == logic_of(which, v, b)) (*mem_mult)[LG_BASE + 256 * which + 16 * v + b]++; else bad(i);
      } else if (v < (F)RC_TABLE) rc_mult[v]++; else bad(i);
    }
    if (MEM) for (int k = 0; k < N_PIECE; k++) {
      const F v = M[(size_t)(C_PIECE + k) * N + i];
      if (row_logic_op(M, N, i) >= 0) {                       // a bitwise row: every piece slot looks up the nibble tuple (a_k, b_k, r_k) in the operation's table
        const int which = row_logic_op(M, N, i); const F b = M[(size_t)(C_LB + k) * N + i], r = M[(size_t)(C_LR + k) * N + i];
        if (v < 16 && b < 16 && r == logic_of(which, v, b)) (*mem_mult)[LG_BASE + 256 * which + 16 * v + b]++; else bad(i);
        continue;
      }
      if (row_mul(M, N, i) || row_wide(M, N, i, mode)) { if (v < (F)RC_TABLE) rc_mult[v]++; else bad(i); continue; }   // a MUL row (mode 4: a wide-arithmetic row): every piece slot reads the 10-bit range table
      if (row_shift(M, N, i)) {                               // a shift row: the slots are re-typed (shift_piece_tag)
        const int tag = shift_piece_tag(k, row_shift_reg(M, N, i));
        if (tag == TAG_LOW6) { if (v < (F)RC_TABLE && M[(size_t)(C_LB + 8) * N + i] == (v & 63)) (*mem_mult)[L6_BASE + v]++; else bad(i); }
        else if (tag == TAG_NIB) { if (v < 16) (*mem_mult)[RC_TABLE + 256 + v]++; else bad(i); }
        else { if (v < (F)RC_TABLE) rc_mult[v]++; else bad(i); }
        continue;
      }
      if (PIECE_TAG[k] == TAG_BYTE) { if (v < 256) (*mem_mult)[RC_TABLE + v]++; else bad(i); }
      else if (PIECE_TAG[k] == TAG_NIB) { if (v < 16) (*mem_mult)[RC_TABLE + 256 + v]++; else bad(i); }
      else { if (v < (F)RC_TABLE) rc_mult[v]++; else bad(i); }
    }
    if (WIDE) for (int k = 0; k < N_X; k++) { const F v = M[(size_t)(C_X + k) * N + i]; if (v < (F)RC_TABLE) rc_mult[v]++; else bad(i); }   // (mode 4) the six extra range slots, on every row
    F t[N_TUPLE]; row_tuple(M, N, i, t);
    const uint64_t pc = (uint64_t)t[0] | ((uint64_t)t[1] << 20) | ((uint64_t)t[2] << 40);
    const uint64_t u = (pc - 0x1000) / 4;
    if (t[0] < (1u << 20) && t[1] < (1u << 20) && t[2] < (1u << 24) && pc >= 0x1000 && (pc & 3) == 0 && u < rom.n && !memcmp(rom.row(u), t, sizeof t)) rom_mult[u]++;
    else bad(i);
  }
}
// batch inversion in E (Montgomery's trick): one einv + 3 emul per element
static void batch_einv(std::vector<E>& v) {
  const size_t n = v.size();
  if (!n) return;
  std::vector<E> pre(n);
  E acc = e_from(1);
  for (size_t i = 0; i < n; i++) { pre[i] = acc; acc = emul(acc, v[i]); }
  E inv = einv(acc);
  for (size_t i = n; i-- > 0;) { const E t = emul(inv, pre[i]); inv = emul(inv, v[i]); v[i] = t; }
}
// T = sum_t m_t / (alpha - t) + sum_u r_u / (alpha - fingerprint(ROM row u)) (+ mode 3: the LOW3, BYTE and NIBBLE tables): the table side of the LogUp identity
static E lookup_table_sum(const Rom& rom, const F* rom_mult, const F* rc_mult, const LookupParams& lp, const F* mem_mult = nullptr) {
  std::vector<E> d(RC_TABLE + rom.n + (mem_mult ? MEM_MULT : 0));
  for (int t = 0; t < RC_TABLE; t++) d[t] = esub(lp.alpha, e_from((F)t));
  for (size_t u = 0; u < rom.n; u++) d[RC_TABLE + u] = esub(lp.alpha, fingerprint(rom.row(u), lp));
  if (mem_mult) {
    E* m = d.data() + RC_TABLE + rom.n;
    for (int t = 0; t < RC_TABLE; t++) m[t] = esub(lp.alpha, eadd(tagged((F)t, TAG_LOW3, lp), emul_f(lp.lam[1], (F)(t & 7))));
    for (int t = 0; t < 256; t++) m[RC_TABLE + t] = esub(lp.alpha, tagged((F)t, TAG_BYTE, lp));
    for (int t = 0; t < 16; t++) m[RC_TABLE + 256 + t] = esub(lp.alpha, tagged((F)t, TAG_NIB, lp));
    for (int which = 0; which < 3; which++) for (int t = 0; t < 256; t++)      // (a, b, a op b): a + lambda b + lambda^2 r + (8 + which) lambda^11
      m[LG_BASE + 256 * which + t] = esub(lp.alpha, eadd(eadd(tagged((F)(t >> 4), TAG_AND + which, lp), emul_f(lp.lam[1], (F)(t & 15))), emul_f(lp.lam[2], logic_of(which, (F)(t >> 4), (F)(t & 15)))));
    for (int t = 0; t < RC_TABLE; t++) m[L6_BASE + t] = esub(lp.alpha, eadd(tagged((F)t, TAG_LOW6, lp), emul_f(lp.lam[1], (F)(t & 63))));
  }
  batch_einv(d);
  E T = e_from(0);
  for (int t = 0; t < RC_TABLE; t++) T = eadd(T, emul_f(d[t], rc_mult[t]));
  for (size_t u = 0; u < rom.n; u++) T = eadd(T, emul_f(d[RC_TABLE + u], rom_mult[u]));
  if (mem_mult) for (int t = 0; t < MEM_MULT; t++) T = eadd(T, emul_f(d[RC_TABLE + rom.n + t], mem_mult[t]));
  return T;
}
// (mode 3) the two ends of the memory check: + 1 / (alpha - fp(cell, time 0, the program image's bytes)) - 1 / (alpha - fp(cell, final time, final bytes)) per touched cell
static E mem_table_sum(const Public& pub, const LookupParams& lp) {
  std::vector<E> d(2 * pub.cells.size());
  for (size_t k = 0; k < pub.cells.size(); k++) {
    const Public::Cell& c = pub.cells[k];
    const F a0 = (F)(c.addr & 0xFFFFF), a1 = (F)((c.addr >> 20) & 0xFFFFF);
    d[2 * k] = esub(lp.alpha, mem_fingerprint(a0, a1, 0, mem_bytes_fp(image_cell(pub.blob, pub.blob_len, c.addr), lp), lp));
    d[2 * k + 1] = esub(lp.alpha, mem_fingerprint(a0, a1, c.t, mem_bytes_fp(c.bytes, lp), lp));
  }
  batch_einv(d);
  E T = e_from(0);
  for (size_t k = 0; k < pub.cells.size(); k++) T = eadd(T, esub(d[2 * k], d[2 * k + 1]));
  return T;
}
// (mode 4) a hash call's tuple: (cycle, R11's limbs, R12's limbs, R13's limbs, kind), tag 12
static inline E hash_fingerprint(const F e[11], const LookupParams& lp) {
  E fp = emul_f(lp.lam[N_TUPLE], (F)TAG_HASH);
  for (int j = 0; j < 11; j++) fp = eadd(fp, emul_f(lp.lam[j], e[j]));
  return fp;
}
// (mode 4 d) a wide-tape tuple: (cycle, rs1's limbs, rs2's limbs, the written value's limbs, opcode), tag 13
static inline E wide_fingerprint(const F e[11], const LookupParams& lp) {
  E fp = emul_f(lp.lam[N_TUPLE], (F)TAG_WIDE);
  for (int j = 0; j < 11; j++) fp = eadd(fp, emul_f(lp.lam[j], e[j]));
  return fp;
}
static inline void wide_tuple(const Public::WideRec& c, F e[11]) {
  F l[3];
  e[0] = (F)(c.cycle % P);
  io_limbs(c.a, l); e[1] = l[0]; e[2] = l[1]; e[3] = l[2];
  io_limbs(c.b, l); e[4] = l[0]; e[5] = l[1]; e[6] = l[2];
  io_limbs(wide_result(c.op, c.a, c.b), l); e[7] = l[0]; e[8] = l[1]; e[9] = l[2];     // what the reference writes: computed HERE, by whoever forms the table side
  e[10] = (F)c.op;
}
// the wide tape's share of the table side: + 1 / (alpha - fp(record)) per record (its row looks it up)
static E wide_table_sum(const Public& pub, const LookupParams& lp) {
  std::vector<E> d;
  for (const Public::WideRec& c : pub.wrecs) { F e[11]; wide_tuple(c, e); d.push_back(esub(lp.alpha, wide_fingerprint(e, lp))); }
  batch_einv(d);
  E T = e_from(0);
  for (const E& x : d) T = eadd(T, x);
  return T;
}
// (mode 4) the hash calls' share of the table side: + 1 / (alpha - fp(call)) per call (its ECALL row looks it up), and the call's memory accesses, which no row states:
// - [1 / (alpha - fp(cell, told, old bytes)) - 1 / (alpha - fp(cell, cycle + 1, new bytes))] per touched cell (what a row would have added on the row side as HMR - HMW)
static E hash_table_sum(const Public& pub, const LookupParams& lp) {
  std::vector<E> d;
  std::vector<uint64_t> nb;
  for (const Public::HashCall& c : pub.hcalls) {
    F e[11]; F l[3];
    e[0] = (F)(c.cycle % P);
    io_limbs(c.in_ptr, l); e[1] = l[0]; e[2] = l[1]; e[3] = l[2];
    io_limbs(c.len, l); e[4] = l[0]; e[5] = l[1]; e[6] = l[2];
    io_limbs(c.out_ptr, l); e[7] = l[0]; e[8] = l[1]; e[9] = l[2];
    e[10] = (F)c.kind;
    d.push_back(esub(lp.alpha, hash_fingerprint(e, lp)));
    hash_call_new_bytes(c, nb);
    for (size_t k = 0; k < c.cells.size(); k++) {
      const F a0 = (F)(c.cells[k].addr & 0xFFFFF), a1 = (F)((c.cells[k].addr >> 20) & 0xFFFFF);
      d.push_back(esub(lp.alpha, mem_fingerprint(a0, a1, c.cells[k].t, mem_bytes_fp(c.cells[k].bytes, lp), lp)));
      d.push_back(esub(lp.alpha, mem_fingerprint(a0, a1, (F)((c.cycle + 1) % P), mem_bytes_fp(nb[k], lp), lp)));
    }
  }
  batch_einv(d);
  E T = e_from(0);
  size_t at = 0;
  for (const Public::HashCall& c : pub.hcalls) {
    T = eadd(T, d[at++]);
    for (size_t k = 0; k < c.cells.size(); k++) { T = esub(T, esub(d[at], d[at + 1])); at += 2; }
  }
  return T;
}
// (mode 2) the tapes' share of the table side: every output index in [oc_first, oc_last) and every input index in [ic_first, ic_last) exactly once —
// the indices the segment's counters ran through (a whole run: 0 .. n_out and 0 .. the inputs it consumed)
static E io_table_sum(const Public& pub, const LookupParams& lp) {
  E T = e_from(0);
  for (uint64_t k = pub.cnt_first[0]; k < pub.cnt_last[0] && k < pub.n_out; k++) { F v[3]; io_limbs(pub.outputs[k], v); T = eadd(T, einv(esub(lp.alpha, io_fingerprint((F)k, v, 2, lp)))); }
  for (uint64_t k = pub.cnt_first[1]; k < pub.cnt_last[1] && k < pub.n_in; k++) { F v[3]; io_limbs(pub.inputs[k], v); T = eadd(T, einv(esub(lp.alpha, io_fingerprint((F)k, v, 3, lp)))); }
  return T;
}
// aux trace [W_AUX][N] of the main-trace matrix M: helper columns of the five lookups of every row and the running sum
static void aux_trace(const std::vector<F>& M, size_t N, const LookupParams& lp, std::vector<F>& A, int mode = 0) {
  A.assign((size_t)aux_width(mode) * N, 0);
  const int NH = N_RC + 1;
  const bool MEM = mode >= 3, WIDE = mode == 4;
  std::vector<E> d((size_t)NH * N);
  for (size_t i = 0; i < N; i++) {
    for (int k = 0; k < N_RC; k++) d[NH * i + k] = esub(lp.alpha, e_from(M[(size_t)rc_col(k) * N + i]));
    if (MEM && row_kmem(M, N, i)) d[NH * i] = esub(lp.alpha, eadd(tagged(M[(size_t)rc_col(0) * N + i], TAG_LOW3, lp), emul_f(lp.lam[1], row_offset(M, N, i))));   // the LOW3 lookup of a memory row
    if (MEM && row_logic_op(M, N, i) >= 0)                                                                                                                       // the tenth nibble tuple of a bitwise row
      d[NH * i + N_RC - 1] = esub(lp.alpha, eadd(eadd(tagged(M[(size_t)rc_col(N_RC - 1) * N + i], TAG_AND + row_logic_op(M, N, i), lp), emul_f(lp.lam[1], M[(size_t)(C_LB + 9) * N + i])),
                                                emul_f(lp.lam[2], M[(size_t)(C_LR + 9) * N + i])));
    F t[N_TUPLE]; row_tuple(M, N, i, t);
    d[NH * i + N_RC] = esub(lp.alpha, fingerprint(t, lp));
  }
  batch_einv(d);
  E S = e_from(0);
  for (size_t i = 0; i < N; i++) {
    E hs = e_from(0);
    for (int k = 0; k < NH; k++) {
      const E& h = d[NH * i + k];
      for (int c = 0; c < 4; c++) A[(size_t)((k < N_RC ? A_H + 4 * k : A_HR) + c) * N + i] = h.c[c];
      hs = eadd(hs, h);
    }
    auto at = [&](int c) { return M[(size_t)c * N + i]; };
    if (mode >= 2) {                                          // the tape helpers: HO = f2 / (alpha - fp(oc, R11)), HI = rl / (alpha - fp(ic, y)); zero on every other row
      if (at(C_F2)) {
        const F v[3] = {at(C_LIMB + 33), at(C_LIMB + 34), at(C_LIMB + 35)};
        const E h = einv(esub(lp.alpha, io_fingerprint(at(C_OC), v, 2, lp)));
        for (int c = 0; c < 4; c++) A[(size_t)(A_HO + c) * N + i] = h.c[c];
        hs = eadd(hs, h);
      }
      if (at(C_RL)) {
        const F v[3] = {at(C_Y), at(C_Y + 1), at(C_Y + 2)};
        const E h = einv(esub(lp.alpha, io_fingerprint(at(C_IC), v, 3, lp)));
        for (int c = 0; c < 4; c++) A[(size_t)(A_HI + c) * N + i] = h.c[c];
        hs = eadd(hs, h);
      }
    }
    if (MEM) {
      // the nine piece helpers P_k = 1 / (alpha - piece_k - tag_k lambda^11), on EVERY row (the pieces of a row that is no memory row are zero)
      const int lgop = row_logic_op(M, N, i);
      for (int k = 0; k < N_PIECE; k++) {
        const E h = lgop >= 0 ? einv(esub(lp.alpha, eadd(eadd(tagged(at(C_PIECE + k), TAG_AND + lgop, lp), emul_f(lp.lam[1], at(C_LB + k))), emul_f(lp.lam[2], at(C_LR + k)))))
                  : row_shift(M, N, i) ? einv(esub(lp.alpha, eadd(tagged(at(C_PIECE + k), shift_piece_tag(k, row_shift_reg(M, N, i)), lp), emul_f(lp.lam[1], at(C_LB + k)))))
                  : (row_mul(M, N, i) || row_wide(M, N, i, mode)) ? einv(esub(lp.alpha, tagged(at(C_PIECE + k), 0, lp)))
                              : einv(esub(lp.alpha, tagged(at(C_PIECE + k), PIECE_TAG[k], lp)));
        for (int c = 0; c < 4; c++) A[(size_t)(A_P + 4 * k + c) * N + i] = h.c[c];
        hs = eadd(hs, h);
      }
      // FPN = sum_k lambda^(3+k) (new byte k): the old bytes with the window replaced by the window value's bytes D_j (D_2 = n0 + 16 n1); on EVERY row
      F ob[8], D[8] = {at(C_PIECE), at(C_PIECE + 1), fadd(at(C_PIECE + 2), fmul(16, at(C_PIECE + 3))), at(C_PIECE + 4), at(C_PIECE + 5), at(C_PIECE + 6), at(C_PIECE + 7), at(C_PIECE + 8)};
      E obfp = e_from(0);
      for (int k = 0; k < 8; k++) { ob[k] = at(C_OB + k); obfp = eadd(obfp, emul_f(lp.lam[3 + k], ob[k])); }
      E fpn = obfp;
      for (int v = 0; v < N_WIN; v++) if (at(C_E + v)) for (int j = 0; j < win_width(v); j++) fpn = eadd(fpn, emul_f(emul_f(lp.lam[3 + win_start(v) + j], fsub(D[j], ob[win_start(v) + j])), at(C_E + v)));
      for (int c = 0; c < 4; c++) A[(size_t)(A_FPN + c) * N + i] = fpn.c[c];
      if (row_kmem(M, N, i)) {                                // HMR = 1 / (alpha - fp(cell, told, old bytes)) is looked up, HMW = 1 / (alpha - fp(cell, cycle + 1, new bytes)) is provided
        const F z0 = fadd(at(C_RC), fmul(RC_TABLE, at(C_RC + 1))), z1 = fadd(at(C_RC + 2), fmul(RC_TABLE, at(C_RC + 3)));
        const F a0 = fsub(z0, row_offset(M, N, i));
        const E hr = einv(esub(lp.alpha, mem_fingerprint(a0, z1, at(C_TOLD), obfp, lp)));
        const E hw = einv(esub(lp.alpha, mem_fingerprint(a0, z1, fadd(at(C_CYCLE), 1), fpn, lp)));
        for (int c = 0; c < 4; c++) { A[(size_t)(A_HMR + c) * N + i] = hr.c[c]; A[(size_t)(A_HMW + c) * N + i] = hw.c[c]; }
        hs = eadd(hs, esub(hr, hw));
      }
    }
    if (WIDE && at(C_FH)) {                                   // (mode 4) HH = fh / (alpha - fp(cycle, R11, R12, R13, kind)): the hash call's record is looked up in the tape the proof carries
      F e[11] = {at(C_CYCLE)};
      for (int j = 0; j < 9; j++) e[1 + j] = at(C_LIMB + 33 + j);
      e[10] = (F)(3 + at(C_H0) + 2 * at(C_H1));
      const E h = einv(esub(lp.alpha, hash_fingerprint(e, lp)));
      for (int c = 0; c < 4; c++) A[(size_t)(A_HH + c) * N + i] = h.c[c];
      hs = eadd(hs, h);
    }
    if (WIDE && at(C_OT)) {                                   // (mode 4 d) WW = ot / (alpha - fp(cycle, rs1, rs2, rd's new value, opcode)): the row's record is in the wide tape
      F e[11] = {at(C_CYCLE), at(C_XB), at(C_XB + 1), at(C_XB + 2), at(C_XC), at(C_XC + 1), at(C_XC + 2), at(C_Y), at(C_Y + 1), at(C_Y + 2), at(C_OP)};
      const E h = einv(esub(lp.alpha, wide_fingerprint(e, lp)));
      for (int c = 0; c < 4; c++) A[(size_t)(A_WW + c) * N + i] = h.c[c];
      hs = eadd(hs, h);
    }
    if (WIDE) for (int k = 0; k < N_X; k++) {                 // (mode 4) XH_k = 1 / (alpha - X_k), on every row
      const E h = einv(esub(lp.alpha, e_from(at(C_X + k))));
      for (int c = 0; c < 4; c++) A[(size_t)(A_X + 4 * k + c) * N + i] = h.c[c];
      hs = eadd(hs, h);
    }
    for (int c = 0; c < 4; c++) A[(size_t)(A_S + c) * N + i] = S.c[c];       // S_i = sum over rows j < i of (hsum_j - T / N); S_0 = 0
    S = eadd(S, esub(hs, lp.t_over_n));
  }
}

// =================================================================================================
// Stage B: AIR quotient, DEEP openings, FRI, proof bytes, verifier  (ZKIR-STARK v1)
// =================================================================================================
// NUM_QUERIES / POW_BITS: the DEFAULT prover parameters (Public::num_queries / pow_bits; the accepted ranges are MIN_.. / MAX_..: a proof may say more, never less)
static const int NUM_QUERIES = 50, LOG_FINAL = 3, LOG_ARITY = 3, POW_BITS = 12, MAX_CONSTRAINTS = 768, MAX_QUERIES = 128, MAX_POW_BITS = 24;
// modes 0 / 1 keep format v10 word for word; modes 2 / 3 are v11 (round 5): EBREAK is a class of its own there (no row but the halt row can sit on it), mode 3 refuses stores
// into the code segment, and a mode-2 SEGMENT's tapes enter the transcript
static inline uint32_t proof_version(int mode) { return mode == 4 ? 12u : mode >= 2 ? 11u : 10u; }
static const uint32_t PROOF_MAGIC = 0x46504B5Au, PROOF_VERSION = 10;   // "ZKPF"; v5: v4 (boundary states) + the program, lookup multiplicities and the aux commitment (AIR v2); v6: AIR v3 (160 columns); v7: only the columns that are not identically zero are committed (144 in default mode); v8: AIR v4 (163 logical columns); v9: AIR v5 (eight range lookups, 40 aux columns, the variant bit in the ROM tuple); v10: AIR v6 (172 logical columns, 152 / 168 committed)
static const int HEADER_WORDS = 21 + 2 * N_STATE;                     // words before the trace root (layout in header_words())

// FRI schedule: committed layer j has 2^log_m values and is folded ks[j] times (binary folds with beta, beta^2, beta^4, ...)
// before the next commitment.  Layer 0 (the DEEP codeword) is folded once, so that its leaves are the pairs (q, q + N) the trace
// and quotient openings determine; every later layer is folded LOG_ARITY times (less at the end, to land on 2^LOG_FINAL values).
static std::vector<int> fri_schedule(int log_n) {
  std::vector<int> ks;
  for (int log_m = log_n + 1; log_m > LOG_FINAL;) {
    const int k = ks.empty() ? 1 : std::min(LOG_ARITY, log_m - LOG_FINAL);
    ks.push_back(k); log_m -= k;
  }
  return ks;
}

// ---- duplex challenger over Poseidon2-12 (rate 8): overwrite-absorb, squeeze from the end of the rate ----
struct Challenger {
  F st[T]; std::vector<F> in, out;
  Challenger() { memset(st, 0, sizeof st); }
  void duplex() { for (size_t i = 0; i < in.size(); i++) st[i] = in[i]; in.clear(); permute(st); out.assign(st, st + RATE); }
  void observe(F x) { out.clear(); in.push_back(x); if ((int)in.size() == RATE) duplex(); }
  void observe_n(const F* x, size_t n) { for (size_t i = 0; i < n; i++) observe(x[i]); }
  void observe_ext(const E& e) { observe_n(e.c, 4); }
  F sample() { if (!in.empty() || out.empty()) duplex(); F v = out.back(); out.pop_back(); return v; }
  E sample_ext() { E e; for (int i = 0; i < 4; i++) e.c[i] = sample(); return e; }
  uint32_t sample_bits(int b) { return sample() & ((1u << b) - 1); }
  // proof of work: the transcript is flushed (pending input absorbed), then the nonce is the smallest field element whose
  // absorption makes the next squeezed element end in POW_BITS zero bits; grind() returns it, check_pow() re-derives the element
  void flush() { if (!in.empty()) duplex(); out.clear(); }
  bool check_pow(F nonce, int bits = POW_BITS) { flush(); observe(nonce); return (sample() & ((1u << bits) - 1)) == 0; }
  F grind(int bits = POW_BITS) {
    flush();
    for (F nonce = 0;; nonce++) { Challenger c = *this; if (c.check_pow(nonce, bits)) return nonce; }
  }
};

// (mode 3) a long section of the proof (the touched cells: seven words per cell) enters the transcript through a two-level sponge: the words are cut into chunks of
// SECTION_CHUNK, every chunk is hashed on its own (hash_elems: the chunks are independent — the prover hashes them in parallel on the device) and the chunk digests are
// observed in order.  Binding like observing the words themselves, at 1 / 128 of the sequential permutations.
static const size_t SECTION_CHUNK = 512;
static void observe_section(Challenger& ch, const uint32_t* w, size_t n) {
  for (size_t at = 0; at < n; at += SECTION_CHUNK) { F dg[DIGEST]; hash_elems(w + at, std::min(SECTION_CHUNK, n - at), dg); ch.observe_n(dg, DIGEST); }
}
// ---- the AIR: Σ_c alpha^c C_c over one (local, next) row pair; values as E (base-field rows are lifted) ----
// is_first = Z_H(x)/(x - 1), is_last = Z_H(x)/(x - w^(n_real-1)) (the last EXECUTED row), is_trans = x - w^-1.
// Every constraint has degree <= 2 in the columns (x is_trans) or degree 1 (x is_first / is_last): the quotient has degree < N.
static std::vector<E>* g_air_record = nullptr;   // tests: every constraint value of the NEXT constraints_sum call, in list order
struct AirAcc {
  const E* ap; E acc; int c;
  void push(const E& v) { acc = eadd(acc, emul(ap[c], v)); c++; if (g_air_record) g_air_record->push_back(v); }
};
// aloc / anxt: the aux columns (local / next row), lp: the lookup challenges and T / N.
static int constraints_sum(const E* loc, const E* nxt, const E* aloc, const E* anxt, const E& is_first, const E& is_last, const E& is_trans, const Public& pub,
                           const LookupParams& lp, const E* ap, E& result) {
  AirAcc A{ap, e_from(0), 0};
  auto push = [&](const E& v) { A.push(v); };
  auto cst = [](uint64_t v) { return e_from((F)(v % P)); };
  const E one = e_from(1);
  const E Dm = cst(pub.mode() == 1 ? 1 : 0), nD = cst(pub.mode() == 1 ? 0 : 1);       // (mode 2 is the DEFAULT VM mode with the I/O argument)
  const E op = loc[C_OP], fa = loc[C_FA], fb = loc[C_FB], fc = loc[C_FC], fhi = loc[C_FHI], s = loc[C_S], se = loc[C_SE];
  E K[N_CLASS];
  for (int k = 0; k < N_CLASS; k++) K[k] = loc[kcol(k)];
  if (pub.has_wide()) K[K_OTH] = e_from(0);                                // (mode 4) no class "other": its column is ot, the wide-tape selector (block 25)
  // 1. cycle counter, first row, last executed row
  push(emul(esub(esub(nxt[C_CYCLE], loc[C_CYCLE]), one), is_trans));
  for (int i = 0; i < N_STATE; i++) push(emul(esub(loc[state_col(i)], cst(pub.first[i])), is_first));   // row 0 is in the public first state
  push(emul(esub(K[K_HALT], one), is_last));
  // 2. R0 is hard-wired zero
  for (int l = 0; l < 3; l++) push(loc[C_LIMB + l]);
  push(loc[C_STATE]);
  // 3. booleans
  auto boolean = [&](const E& b) { push(emul(b, esub(b, one))); };
  for (int r = 0; r < 16; r++) boolean(loc[C_STATE + r]);
  for (int r = 0; r < 15; r++) { boolean(loc[C_WR + r]); boolean(loc[C_SELB + r]); boolean(loc[C_SELC + r]); }
  for (int k = 0; k < N_CLASS; k++) boolean(K[k]);
  boolean(s); boolean(loc[C_C0]); boolean(loc[C_C1]); boolean(loc[C_D0]); boolean(loc[C_D1]); boolean(loc[C_D2]); boolean(loc[C_NE]); boolean(loc[C_TK]); boolean(loc[C_B0]); boolean(loc[C_SB]); boolean(loc[C_NZ]);
  // 4. exactly one class; an executed row (not halt, not pad) runs as the class of its instruction word: sum_k k K_k = opclass, where
  //    opclass is part of the ROM tuple (constraint 15), i.e. the PROGRAM's word at pc decides it (default mode)
  const bool IO = pub.mode() >= 2, MEM = pub.mode() >= 3, WIDE = pub.mode() == 4;
  const E Kec = IO ? eadd(eadd(loc[C_F2], loc[C_RL]), eadd(loc[C_RE], loc[C_FH])) : e_from(0);   // (mode 2) the ecall class: the sum of its four syscall flags
  const E Kld = MEM ? loc[C_KLD] : e_from(0), Kst = MEM ? loc[C_KST] : e_from(0), Kmem = eadd(Kld, Kst);   // (mode 3) loads, stores
  const E Klg = MEM ? loc[C_KLG] : e_from(0);                                                             // (mode 3) the bitwise opcodes
  const E Kmu = MEM ? loc[C_KMU] : e_from(0);                                                             // (mode 3) MUL
  const E Ksh = MEM ? loc[C_KSH] : e_from(0);                                                             // (mode 3) the shifts
  const E Kin = WIDE ? eadd(eadd(loc[C_OM], loc[C_OD]), loc[C_ORR]) : e_from(0);                          // (mode 4) MULH DIVU REMU DIV REM by the chunk relation: om + od + orr
  const E Kwa = WIDE ? eadd(Kin, loc[C_OT]) : e_from(0);                                                  // .. the class: kwa = om + od + orr + ot (ot: through the wide tape); no column of its own
  { E sum = eadd(eadd(eadd(eadd(eadd(Kec, Kmem), Klg), Ksh), Kmu), Kwa); for (int k = 0; k < N_CLASS; k++) sum = eadd(sum, K[k]); push(esub(sum, one)); }
  {
    E ks = eadd(eadd(eadd(eadd(eadd(emul_f(Kec, (F)K_ECALL), emul_f(Klg, (F)K_LG)), emul_f(Ksh, (F)K_SH)), eadd(emul_f(Kld, (F)K_LD), emul_f(Kst, (F)K_ST))), emul_f(Kmu, (F)K_MU)), emul_f(Kwa, (F)K_WA));
    for (int k = 1; k < N_CLASS; k++) if (k != K_HALT && k != K_PAD) ks = eadd(ks, emul_f(K[k], (F)k));
    push(emul(nD, esub(emul(esub(one, eadd(K[K_HALT], K[K_PAD])), loc[C_OPC]), ks)));
  }
  const E Kbr = eadd(K[K_BRE], K[K_BRU]);                                    // B-type rows: rs1 in field a, nothing written, may be taken
  const E Kcmp = eadd(K[K_SE], K[K_SU]);                                     // comparison rows: the flag is the value written
  // 5. register selectors: wr (written register = field a for add/addi/jal, none for bne/halt/pad, at most one in default mode),
  //    selb = one-hot(fb), selc = one-hot(fc), or one-hot(fa) on BNE rows
  auto moments = [&](int base, E& s0, E& s1, E& s2) {
    s0 = s1 = s2 = e_from(0);
    for (int r = 1; r < 16; r++) { const E& v = loc[base + r - 1]; s0 = eadd(s0, v); s1 = eadd(s1, emul_f(v, r)); s2 = eadd(s2, emul_f(v, r * r)); }
  };
  E w0, w1, w2, b0, b1, b2, c0s, c1s, c2s;
  moments(C_WR, w0, w1, w2); moments(C_SELB, b0, b1, b2); moments(C_SELC, c0s, c1s, c2s);
  push(emul(nD, esub(emul(w1, w1), w2)));
  push(emul(eadd(eadd(eadd(eadd(K[K_ADD], K[K_ADDI]), eadd(K[K_JAL], K[K_SUB])), Kcmp), K[K_JALR]), esub(w1, fa)));
  push(emul(eadd(eadd(eadd(Kbr, emul(nD, K[K_OJ])), K[K_HALT]), K[K_PAD]), w0));   // branches write nothing (BLT / BGE too: class oj in default mode)
  push(esub(b1, fb)); push(esub(emul(b1, b1), b2));
  push(esub(c1s, eadd(fc, emul(eadd(Kbr, Kst), esub(fa, fc))))); push(esub(emul(c1s, c1s), c2s));   // (S-type words: rs1 in field a like B-type ones)
  //    (v6) conditional moves CMOV / CMOVNZ (class cmn: the condition is rs2 != 0) and CMOVZ (class cmz: rs2 == 0), execute.rs:434-472:
  //    q = "this row is a conditional move whose condition holds"; it writes rd = field a exactly then (nothing at all otherwise)
  const E Kcm = eadd(K[K_CMN], K[K_CMZ]), nz = loc[C_NZ], q = loc[C_Q];
  push(esub(q, eadd(emul(K[K_CMN], nz), emul(K[K_CMZ], esub(one, nz)))));
  push(emul(q, esub(w1, fa)));
  push(emul(esub(Kcm, q), w0));
  // 6. operand fetch
  for (int l = 0; l < 3; l++) {
    E xb = e_from(0), xc = e_from(0);
    for (int r = 1; r < 16; r++) { xb = eadd(xb, emul(loc[C_SELB + r - 1], loc[C_LIMB + 3 * r + l])); xc = eadd(xc, emul(loc[C_SELC + r - 1], loc[C_LIMB + 3 * r + l])); }
    push(esub(loc[C_XB + l], xb)); push(esub(loc[C_XC + l], xc));
  }
  // 7. opcode semantics of the value written (execute.rs:43-63 ADD, :185-197 ADDI, :639-647 JAL link), 40-bit wrap as two 20-bit limbs
  const E two20 = cst(1u << 20), two24 = cst(1u << 24);
  const E imm17 = eadd(fc, emul_f(fhi, 16));
  const E im0 = eadd(esub(imm17, emul_f(s, 1u << 17)), emul_f(s, 1u << 20)), im1 = emul_f(s, 0xFFFFF);
  const E lo20 = esub(eadd(eadd(fb, emul_f(fc, 16)), emul_f(fhi, 256)), emul_f(s, 1u << 20));
  //    — stated on z, the range-checked pair of limbs (13.); y = z on the rows that write it (and on "other" rows, whose y stays in range)
  const E *xb = loc + C_XB, *xc = loc + C_XC, *y = loc + C_Y, *pc = loc + C_PC;
  const E* R = loc + C_RC;
  const E z[2] = {eadd(R[0], emul_f(R[1], RC_TABLE)), eadd(R[2], emul_f(R[3], RC_TABLE))};   // (v6) z IS its chunks: no columns of its own
  const E c0 = loc[C_C0], c1 = loc[C_C1];
  push(emul(K[K_ADD], eadd(esub(esub(z[0], xb[0]), xc[0]), emul(two20, c0))));
  push(emul(K[K_ADD], eadd(esub(esub(esub(z[1], xb[1]), xc[1]), c0), emul(two20, c1))));
  push(emul(eadd(K[K_ADD], K[K_SUB]), y[2]));
  push(emul(K[K_ADDI], eadd(esub(esub(z[0], xb[0]), im0), emul(two20, c0))));
  push(emul(K[K_ADDI], eadd(esub(esub(esub(z[1], xb[1]), im1), c0), emul(two20, c1))));
  push(emul(K[K_ADDI], y[2]));
  const E Kj = eadd(K[K_JAL], K[K_JALR]);                                    // both link pc + 4 (execute.rs:639-658)
  push(emul(Kj, eadd(esub(esub(z[0], pc[0]), cst(4)), emul(two20, c0))));
  push(emul(Kj, eadd(esub(esub(z[1], pc[1]), c0), emul(two20, c1))));
  push(emul(Kj, esub(esub(y[2], pc[2]), c1)));
  // 7b. (v3) differences with borrows: z = xb - xc mod 2^40 on SUB and SLTU / SGEU rows (execute.rs:65-77, :373-407), z = xc - xb on BLTU /
  //     BGEU rows (:618-636); c1 = 1 exactly when the minuend is the smaller 40-bit value (z's limbs are in range, so the borrows are forced)
  {
    const E Ks = eadd(K[K_SUB], K[K_SU]);
    // (v5) ordered comparisons, signed or not (op = base + 2 g + pol; sgn = g on SLTU.. rows, 1 - g on BLT.. rows): the high limbs enter biased,
    // ta = a1 + 2^19 sgn - 2^20 sa, tb likewise with sb — the limbs of value XOR 2^39 when sgn = 1 (value.rs:710-716) — so ta - tb = a1 - b1 -
    // 2^20 (sa - sb); u = (ta, tb) is the second range-checked pair of the row, which forces sa / sb to be the sign bits (to be 0 when sgn = 0)
    const E sab = emul(two20, esub(loc[C_B0], loc[C_SB]));
    push(emul(Ks, esub(eadd(esub(z[0], xb[0]), xc[0]), emul(two20, c0))));
    push(emul(K[K_SUB], esub(eadd(eadd(esub(z[1], xb[1]), xc[1]), c0), emul(two20, c1))));
    push(emul(K[K_SU], eadd(esub(eadd(eadd(esub(z[1], xb[1]), xc[1]), c0), emul(two20, c1)), sab)));
    push(emul(K[K_BRU], esub(eadd(esub(z[0], xc[0]), xb[0]), emul(two20, c0))));
    push(emul(K[K_BRU], eadd(esub(eadd(eadd(esub(z[1], xc[1]), xb[1]), c0), emul(two20, c1)), sab)));
    const E* R2 = loc + C_RC2;
    const E u0 = eadd(R2[0], emul_f(R2[1], RC_TABLE)), u1 = eadd(R2[2], emul_f(R2[3], RC_TABLE));
    const E two19 = cst(1u << 19), gsu = emul(two19, loc[C_G]), gbr = emul(two19, esub(one, loc[C_G]));
    push(emul(K[K_SU], eadd(esub(esub(u0, xb[1]), gsu), emul(two20, loc[C_B0]))));
    push(emul(K[K_SU], eadd(esub(esub(u1, xc[1]), gsu), emul(two20, loc[C_SB]))));
    push(emul(K[K_BRU], eadd(esub(esub(u0, xc[1]), gbr), emul(two20, loc[C_B0]))));
    push(emul(K[K_BRU], eadd(esub(esub(u1, xb[1]), gbr), emul(two20, loc[C_SB]))));
  }
  //     the written value: y = z on arithmetic and "other" rows, (fx, 0, 0) on comparison rows
  {
    const E Ky = eadd(eadd(eadd(eadd(K[K_ADD], K[K_ADDI]), eadd(K[K_JAL], K[K_SUB])), eadd(K[K_OTH], K[K_JALR])), emul(Dm, K[K_OJ]));   // (oj writes only in deferred mode)
    push(emul(Ky, esub(y[0], z[0]))); push(emul(Ky, esub(y[1], z[1])));
    push(emul(Kcmp, esub(y[0], loc[C_FX]))); push(emul(Kcmp, y[1])); push(emul(Kcmp, y[2]));
    for (int l = 0; l < 3; l++) push(emul(Kcm, esub(y[l], xb[l])));          // (v6) a conditional move writes rs1's raw value (all three limbs)
    // (v6) the bits above 40 of what an "other" row writes: y2 = R4 + 2^10 R5 + 2^20 R6 with R7 = 64 R6 — all four in the 10-bit table, so y2 < 2^24.
    // With it EVERY limb of every register is in range by induction (constrained classes write 0, pc2 + c1 or an operand's limb there)
    const E* R2 = loc + C_RC2;
    push(emul(K[K_OTH], esub(esub(esub(y[2], R2[0]), emul_f(R2[1], RC_TABLE)), emul_f(R2[2], RC_TABLE * RC_TABLE))));
    push(emul(K[K_OTH], esub(R2[3], emul_f(R2[2], 64))));
  }
  // 7c. (v6) nz = [xc != 0] on every row, on the sum of xc's limbs (in range, so the sum vanishes only if they all do)
  {
    const E sx = eadd(eadd(xc[0], xc[1]), xc[2]);
    push(emul(esub(one, nz), sx));
    push(esub(nz, emul(sx, loc[C_IVZ])));
  }
  // 8. BNE compares the raw 64-bit values (execute.rs:588-596): ne = [xb != xc] over all three limbs
  {
    E dot = e_from(0);
    for (int l = 0; l < 3; l++) { const E d = esub(xb[l], xc[l]); dot = eadd(dot, emul(d, loc[C_IV + l])); push(emul(esub(one, loc[C_NE]), d)); }
    push(esub(loc[C_NE], dot));
  }
  // 8b. (v3) the family's comparison and its polarity: flag = [xb == xc] on BEQ / BNE / SEQ / SNE rows, the borrow c1 on the unsigned
  //     comparisons, 0 elsewhere; fx = flag XOR pol, pol = op - (the family's even opcode) — the ROM ties op to the class, so pol is 0 / 1;
  //     a branch is taken iff fx (execute.rs:578-596, :618-636)
  push(esub(esub(loc[C_FLAG], emul(eadd(K[K_BRE], K[K_SE]), esub(one, loc[C_NE]))), emul(eadd(K[K_BRU], K[K_SU]), c1)));
  {
    E pol = op;
    for (int k = 0; k < N_CLASS; k++) if (family_base(k)) pol = esub(pol, emul_f(K[k], family_base(k)));
    pol = esub(pol, emul_f(loc[C_G], 2));                                   // (v5) op = base + 2 g + pol in the four-member families; g = 0 elsewhere
    push(esub(esub(esub(loc[C_FX], loc[C_FLAG]), pol), emul_f(emul(pol, loc[C_FLAG]), P - 2)));      // fx - flag - pol + 2 pol flag
  }
  push(esub(loc[C_TK], emul(Kbr, loc[C_FX])));
  // 9. next pc: pc + 4, pc + imm17 (branch taken) or pc + off21 (JAL), wrapping at 2^64 over (20, 20, 24)-bit limbs (state.rs:131-133)
  push(esub(loc[C_DL0], eadd(eadd(cst(4), emul(loc[C_TK], esub(im0, cst(4)))), emul(K[K_JAL], esub(lo20, cst(4))))));
  push(esub(se, emul(eadd(loc[C_TK], K[K_JAL]), s)));
  const E kc = esub(esub(esub(esub(one, K[K_JALR]), K[K_OJ]), K[K_HALT]), K[K_PAD]);   // pc' = pc + delta: every class but jalr (below), oj (free), halt / pad (keep);
                                                                              // class "other" is in it with delta 4 (tk = 0 there): it is sequential
  const E hp = eadd(K[K_HALT], K[K_PAD]);
  push(emul(emul(kc, eadd(esub(esub(nxt[C_PC], pc[0]), loc[C_DL0]), emul(two20, loc[C_D0]))), is_trans));
  push(emul(emul(kc, eadd(esub(esub(esub(nxt[C_PC + 1], pc[1]), emul_f(se, 0xFFFFF)), loc[C_D0]), emul(two20, loc[C_D1]))), is_trans));
  push(emul(emul(kc, eadd(esub(esub(esub(nxt[C_PC + 2], pc[2]), emul_f(se, 0xFFFFFF)), loc[C_D1]), emul(two24, loc[C_D2]))), is_trans));
  for (int l = 0; l < 3; l++) push(emul(emul(hp, esub(nxt[C_PC + l], pc[l])), is_trans));
  // 9b. (v4) JALR: pc' + b0 = rs1 + sext(imm17) mod 2^64 over (20, 20, 24)-bit limbs, b0 = the bit that is cleared (execute.rs:649-658); the
  //     limbs of pc' are a code address (every row's pc is looked up in the ROM), so the carries and b0 are forced
  push(emul(emul(K[K_JALR], esub(eadd(eadd(nxt[C_PC], loc[C_B0]), emul(two20, loc[C_D0])), eadd(xb[0], im0))), is_trans));
  push(emul(emul(K[K_JALR], esub(eadd(nxt[C_PC + 1], emul(two20, loc[C_D1])), eadd(eadd(xb[1], im1), loc[C_D0]))), is_trans));
  push(emul(emul(K[K_JALR], esub(eadd(nxt[C_PC + 2], emul(two24, loc[C_D2])), eadd(eadd(xb[2], emul_f(s, 0xFFFFFF)), loc[C_D1]))), is_trans));
  // 10. register file update: the written register takes y (default mode) / may change freely (deferred mode), the others keep
  //     their limbs and storage state; a written register becomes Normalized in default mode
  for (int r = 1; r < 16; r++) {
    const E& wr = loc[C_WR + r - 1];
    for (int l = 0; l < 3; l++) {
      const E &cur = loc[C_LIMB + 3 * r + l], &nx = nxt[C_LIMB + 3 * r + l];
      push(emul(esub(esub(nx, cur), emul(wr, esub(eadd(emul(nD, y[l]), emul(Dm, nx)), cur))), is_trans));
    }
    const E &cur = loc[C_STATE + r], &nx = nxt[C_STATE + r];
    push(emul(esub(esub(nx, cur), emul(wr, esub(emul(Dm, nx), cur))), is_trans));
  }
  // 11. executed rows, then the halt row, then padding only
  push(emul(emul(K[K_HALT], esub(one, nxt[C_K + K_PAD])), is_trans));
  push(emul(emul(K[K_PAD], esub(one, nxt[C_K + K_PAD])), is_trans));
  push(emul(emul(esub(esub(one, K[K_PAD]), K[K_HALT]), nxt[C_K + K_PAD]), is_trans));
  // 12. (v4) the last executed row is in the public last state: what a following segment starts from
  for (int i = 0; i < N_STATE; i++) push(emul(esub(loc[state_col(i)], cst(pub.last[i])), is_last));
  // ---- AIR v2: the lookup argument.  Extension-field columns are four base columns (coordinates in F[X]/(X^4 - 11)); a constraint
  //      between extension values is stated coordinate by coordinate, so every polynomial involved stays a base-field polynomial.
  // product of two extension values given by coordinates: out_k = sum_{i+j=k} h_i d_j + 11 sum_{i+j=k+4} h_i d_j
  auto ext_mul = [&](const E* h, const E* d, E* out) {
    for (int k = 0; k < 4; k++) {
      E lo = e_from(0), hi = e_from(0);
      for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { if (i + j == k) lo = eadd(lo, emul(h[i], d[j])); else if (i + j == k + 4) hi = eadd(hi, emul(h[i], d[j])); }
      out[k] = eadd(lo, emul_f(hi, WEXT));
    }
  };
  // 13. (the limbs of z are two 10-bit chunks each: since v6 z is DEFINED as R0 + 1024 R1, R2 + 1024 R3 — the two constraints that tied the z columns
  //     to the chunks are gone with the columns)
  // 14. range helpers: H_i (alpha - R_i) = 1
  for (int i = 0; i < N_RC; i++) {
    E d[4], pr[4];
    for (int k = 0; k < 4; k++) d[k] = cst(lp.alpha.c[k]);
    d[0] = esub(d[0], loc[rc_col(i)]);
    if (MEM && i == 0) {                                                   // (mode 3) a memory row's first chunk goes to the LOW3 table with the window's offset: alpha - R0 - lambda off - 4 lambda^11 Kmem
      E off = e_from(0);
      for (int v = 0; v < N_WIN; v++) off = eadd(off, emul_f(loc[C_E + v], (F)win_start(v)));
      for (int k = 0; k < 4; k++) d[k] = esub(esub(d[k], emul_f(off, lp.lam[1].c[k])), emul_f(Kmem, fmul(TAG_LOW3, lp.lam[N_TUPLE].c[k])));
    }
    if (MEM && i == N_RC - 1) {                                            // (mode 3) a bitwise row's tenth nibble tuple: alpha - R7 - lambda b_9 - lambda^2 r_9 - (8 oa + 9 oo + 10 ox) lambda^11
      const E ox = esub(esub(Klg, loc[C_OA]), loc[C_OO]);
      const E tg = eadd(eadd(emul_f(loc[C_OA], TAG_AND), emul_f(loc[C_OO], TAG_OR)), emul_f(ox, TAG_XOR));
      for (int k = 0; k < 4; k++) d[k] = esub(esub(esub(d[k], emul_f(loc[C_LB + 9], lp.lam[1].c[k])), emul_f(loc[C_LR + 9], lp.lam[2].c[k])), emul_f(tg, lp.lam[N_TUPLE].c[k]));
    }
    ext_mul(aloc + A_H + 4 * i, d, pr);
    push(esub(pr[0], one)); push(pr[1]); push(pr[2]); push(pr[3]);
  }
  // 15. instruction ROM: HR (alpha - fingerprint(pc limbs, op, fa, fb, fc, fhi, s, opclass)) = 1
  {
    static const int cols[N_TUPLE] = {C_PC, C_PC + 1, C_PC + 2, C_OP, C_FA, C_FB, C_FC, C_FHI, C_S, C_OPC, C_G};
    E d[4], pr[4];
    for (int k = 0; k < 4; k++) {
      E fp = cst(lp.lam[N_TUPLE].c[k]);
      for (int j = 0; j < N_TUPLE; j++) fp = eadd(fp, emul_f(loc[cols[j]], lp.lam[j].c[k]));
      d[k] = esub(cst(lp.alpha.c[k]), fp);
    }
    ext_mul(aloc + A_HR, d, pr);
    push(esub(pr[0], one)); push(pr[1]); push(pr[2]); push(pr[3]);
  }
  // 16. running sum, cyclic over ALL N rows (no selector): S(w x) - S(x) = H0 + H1 + H2 + H3 + HR - T / N.  Summed over the cycle the left
  //     side telescopes to zero, so the row side of the LogUp identity equals T, the table side the verifier computed.
  for (int k = 0; k < 4; k++) {
    E hs = aloc[A_HR + k];
    for (int i = 0; i < N_RC; i++) hs = eadd(hs, aloc[A_H + 4 * i + k]);
    if (IO) hs = eadd(hs, eadd(aloc[A_HO + k], aloc[A_HI + k]));
    if (MEM) { for (int i = 0; i < N_PIECE; i++) hs = eadd(hs, aloc[A_P + 4 * i + k]); hs = eadd(hs, esub(aloc[A_HMR + k], aloc[A_HMW + k])); }   // pieces and the read are looked up, the write is PROVIDED (a table entry)
    if (WIDE) { for (int i = 0; i < N_X; i++) hs = eadd(hs, aloc[A_X + 4 * i + k]); hs = eadd(hs, eadd(aloc[A_HH + k], aloc[A_WW + k])); }   // (mode 4) the six extra range slots, the hash-call lookup
    push(eadd(esub(esub(anxt[A_S + k], aloc[A_S + k]), hs), cst(lp.t_over_n.c[k])));
  }
  // ---- 17. (mode 2, round 4) ECALL rows and the I/O tapes (syscall.rs:94-177).  Appended to the list: modes 0 / 1 stop here. ----
  if (IO) {
    const E f2 = loc[C_F2], rl = loc[C_RL], re = loc[C_RE], fh = loc[C_FH], h0 = loc[C_H0], h1 = loc[C_H1], oc = loc[C_OC], ic = loc[C_IC];
    boolean(f2); boolean(rl); boolean(re); boolean(fh); boolean(h0); boolean(h1);
    push(emul(h0, esub(one, fh))); push(emul(h1, esub(one, fh)));                                   // the two bits of R10 - 3 live on hash rows only
    // the syscall number: R10 = 1 (READ), 2 (WRITE), 3 + h0 + 2 h1 (hash) — R10 = 0 (EXIT) halts and is no executed row, anything else is a run-time error
    const E* r10 = loc + C_LIMB + 30; const E* r11 = loc + C_LIMB + 33;
    push(emul(Kec, r10[1])); push(emul(Kec, r10[2]));
    push(esub(emul(Kec, r10[0]), eadd(eadd(eadd(rl, re), emul_f(f2, 2)), eadd(emul_f(fh, 3), eadd(h0, emul_f(h1, 2))))));
    // what an ecall writes: READ and the hashes write R10 (exactly one register: R10), WRITE writes nothing
    const E wgrp = eadd(eadd(rl, re), fh);
    push(emul(wgrp, esub(w1, cst(10)))); push(emul(wgrp, esub(w0, one))); push(emul(f2, w0));
    for (int l = 0; l < 3; l++) push(emul(eadd(fh, re), y[l]));                                     // hashes return 0, so does a READ on an exhausted tape
    // the counters: oc counts the WRITEs, ic the inputs consumed
    push(emul(esub(esub(nxt[C_OC], oc), f2), is_trans));
    push(emul(esub(esub(nxt[C_IC], ic), rl), is_trans));
    push(emul(re, esub(ic, cst(lp.n_in))));                                                         // "exhausted" means exactly that: every input has been consumed
    // the two tape lookups: HO (alpha - fp(oc, R11)) = f2, HI (alpha - fp(ic, y)) = rl
    {
      E d[4], pr[4];
      for (int k = 0; k < 4; k++) {
        E fp = eadd(emul_f(cst(lp.lam[N_TUPLE].c[k]), 2), emul_f(oc, lp.lam[0].c[k]));
        for (int j = 0; j < 3; j++) fp = eadd(fp, emul_f(r11[j], lp.lam[1 + j].c[k]));
        d[k] = esub(cst(lp.alpha.c[k]), fp);
      }
      ext_mul(aloc + A_HO, d, pr);
      push(esub(pr[0], f2)); push(pr[1]); push(pr[2]); push(pr[3]);
      for (int k = 0; k < 4; k++) {
        E fp = eadd(emul_f(cst(lp.lam[N_TUPLE].c[k]), 3), emul_f(ic, lp.lam[0].c[k]));
        for (int j = 0; j < 3; j++) fp = eadd(fp, emul_f(y[j], lp.lam[1 + j].c[k]));
        d[k] = esub(cst(lp.alpha.c[k]), fp);
      }
      ext_mul(aloc + A_HI, d, pr);
      push(esub(pr[0], rl)); push(pr[1]); push(pr[2]); push(pr[3]);
    }
    // the counters of the first and of the last executed row are public (header): a whole run starts at (0, 0) and ends with oc = the number of outputs
    push(emul(esub(oc, cst(pub.cnt_first[0])), is_first)); push(emul(esub(ic, cst(pub.cnt_first[1])), is_first));
    push(emul(esub(oc, cst(pub.cnt_last[0])), is_last)); push(emul(esub(ic, cst(pub.cnt_last[1])), is_last));
  }
  // ---- 18. (mode 3, round 4) loads, stores and the memory check (execute.rs:477-575, memory.rs:86-505).  Appended to the list: mode 2 stops above. ----
  if (MEM) {
    const E* Ev = loc + C_E; const E* ob = loc + C_OB; const E* pcs = loc + C_PIECE;
    const E sgb = loc[C_SGB], sgh = loc[C_SGH], tb = loc[C_TB], sx = loc[C_SX], cm2 = loc[C_CM2];
    boolean(Kld); boolean(Kst);
    for (int v = 0; v < N_WIN; v++) boolean(Ev[v]);
    boolean(sgb); boolean(sgh); boolean(tb); boolean(cm2);
    E Esum = e_from(0), WB = e_from(0), WH = e_from(0), WW = e_from(0), off = e_from(0);
    for (int v = 0; v < N_WIN; v++) {
      Esum = eadd(Esum, Ev[v]); off = eadd(off, emul_f(Ev[v], (F)win_start(v)));
      if (win_width(v) == 1) WB = eadd(WB, Ev[v]); else if (win_width(v) == 2) WH = eadd(WH, Ev[v]); else if (win_width(v) == 4) WW = eadd(WW, Ev[v]);
    }
    const E WD = Ev[14];
    push(esub(Esum, Kmem));                                                                         // exactly one window on a memory row, none elsewhere
    if (WIDE) push(emul(loc[C_H0], esub(one, loc[C_H1])));                                          // (mode 4) hash syscalls are a tape (below); syscall 4 — Poseidon2, an error in the reference — never is a row
    else push(loc[C_FH]);                                                                           // no hash syscall in this mode: its memory effect is not stated
    // the opcode names the width (and, for byte / halfword loads, whether the value is sign-extended): LB LBU LH LHU LW LD = 0x30.., SB SH SW SD = 0x38..
    push(emul(Kld, eadd(esub(esub(esub(esub(esub(op, cst(0x30)), emul_f(WH, 2)), emul_f(WW, 4)), emul_f(WD, 5)), eadd(WB, WH)), eadd(sgb, sgh))));
    push(emul(Kst, esub(esub(esub(esub(op, cst(0x38)), WH), emul_f(WW, 2)), emul_f(WD, 3))));
    push(emul(sgb, esub(esub(cst(2), Kld), WB))); push(emul(sgh, esub(esub(cst(2), Kld), WH)));       // sgb only on byte loads, sgh only on halfword loads
    // what they write: a load rd = field a, a store nothing
    push(emul(Kld, esub(w1, fa))); push(emul(Kst, w0));
    // the address rs1 + sext(imm17) mod 2^64 (rs1 = operand b on loads, operand c on stores) = z (two range-checked 20-bit limbs) — its third limb must be zero
    for (int l = 0; l < 2; l++) {
      const E cin = l ? c0 : e_from(0), cout = l ? c1 : c0, im = l ? im1 : im0;
      push(eadd(emul(Kld, eadd(esub(esub(esub(z[l], xb[l]), im), cin), emul(two20, cout))), emul(Kst, eadd(esub(esub(esub(z[l], xc[l]), im), cin), emul(two20, cout)))));
    }
    push(eadd(emul(Kld, esub(eadd(eadd(xb[2], emul_f(s, 0xFFFFFF)), c1), emul(two24, cm2))), emul(Kst, esub(eadd(eadd(xc[2], emul_f(s, 0xFFFFFF)), c1), emul(two24, cm2)))));
    // the time read is smaller than the time written (cycle + 1): cycle - told = R4 + 2^10 R5 + 2^20 R6, three chunks of the second range-checked group
    const E* R2 = loc + C_RC2;
    push(emul(Kmem, esub(esub(esub(esub(loc[C_CYCLE], loc[C_TOLD]), R2[0]), emul_f(R2[1], RC_TABLE)), emul_f(R2[2], RC_TABLE * RC_TABLE))));
    // stores: the pieces are the stored register's (rs2 = operand b), limb by limb — byte and nibble lookups make the decomposition unique
    const E lim0 = eadd(eadd(pcs[0], emul_f(pcs[1], 1u << 8)), emul_f(pcs[2], 1u << 16)), lim1 = eadd(eadd(pcs[3], emul_f(pcs[4], 1u << 4)), emul_f(pcs[5], 1u << 12)),
            lim2 = eadd(eadd(pcs[6], emul_f(pcs[7], 1u << 8)), emul_f(pcs[8], 1u << 16));
    push(emul(Kst, esub(xb[0], lim0))); push(emul(Kst, esub(xb[1], lim1))); push(emul(Kst, esub(xb[2], lim2)));
    // y = the window value's limbs, zero-extended from the width, sign-extended when sx (on stores nothing is written: y merely satisfies this)
    const E W48 = eadd(WW, WD);
    push(esub(emul(Kmem, y[0]), eadd(eadd(eadd(emul(WB, pcs[0]), emul(WH, eadd(pcs[0], emul_f(pcs[1], 1u << 8)))), emul(W48, lim0)),
                                     emul(sx, esub(esub(two20, emul_f(WB, 1u << 8)), emul_f(WH, 1u << 16))))));
    push(esub(emul(Kmem, y[1]), eadd(eadd(emul(WW, eadd(pcs[3], emul_f(pcs[4], 1u << 4))), emul(WD, lim1)), emul_f(sx, 0xFFFFF))));
    push(esub(emul(Kmem, y[2]), eadd(emul(WD, lim2), emul_f(sx, 0xFFFFFF))));
    // sign extension: sx = (sgb + sgh) tb; on LB rows d0 = 128 tb + d6 / 2, on LH rows d1 = 128 tb + d6 / 2 — d6 is in the byte table, so d6 / 2 has seven bits
    push(esub(sx, emul(eadd(sgb, sgh), tb)));
    const E half = cst(finv(2));
    push(emul(sgb, esub(esub(pcs[0], emul_f(tb, 128)), emul(pcs[7], half))));
    push(emul(sgh, esub(esub(pcs[1], emul_f(tb, 128)), emul(pcs[7], half))));
    // the cell's new bytes, as their fingerprint FPN (an aux column: it depends on lambda): the old bytes with the window replaced by the window value's bytes D_j; a load keeps the cell
    const E D[8] = {pcs[0], pcs[1], eadd(pcs[2], emul_f(pcs[3], 16)), pcs[4], pcs[5], pcs[6], pcs[7], pcs[8]};
    for (int k = 0; k < 4; k++) {
      E obfp = e_from(0), fpn;
      for (int j = 0; j < 8; j++) obfp = eadd(obfp, emul_f(ob[j], lp.lam[3 + j].c[k]));
      fpn = obfp;
      for (int v = 0; v < N_WIN; v++) {
        E dl = e_from(0);
        for (int j = 0; j < win_width(v); j++) dl = eadd(dl, emul_f(esub(D[j], ob[win_start(v) + j]), lp.lam[3 + win_start(v) + j].c[k]));
        fpn = eadd(fpn, emul(Ev[v], dl));
      }
      push(esub(aloc[A_FPN + k], fpn));
    }
    for (int k = 0; k < 4; k++) { E obfp = e_from(0); for (int j = 0; j < 8; j++) obfp = eadd(obfp, emul_f(ob[j], lp.lam[3 + j].c[k])); push(emul(Kld, esub(aloc[A_FPN + k], obfp))); }
    // the memory check: HMR (alpha - fp(cell, told, old bytes)) = Kmem, HMW (alpha - fp(cell, cycle + 1, new bytes)) = Kmem; cell = (z0 - off, z1)
    for (int rw = 0; rw < 2; rw++) {
      E d[4], pr[4];
      for (int k = 0; k < 4; k++) {
        E fp = eadd(cst(fmul(TAG_MEM, lp.lam[N_TUPLE].c[k])), emul_f(esub(z[0], off), lp.lam[0].c[k]));
        fp = eadd(fp, emul_f(z[1], lp.lam[1].c[k]));
        if (rw == 0) { fp = eadd(fp, emul_f(loc[C_TOLD], lp.lam[2].c[k])); for (int j = 0; j < 8; j++) fp = eadd(fp, emul_f(ob[j], lp.lam[3 + j].c[k])); }
        else { fp = eadd(fp, emul_f(eadd(loc[C_CYCLE], one), lp.lam[2].c[k])); fp = eadd(fp, aloc[A_FPN + k]); }
        d[k] = esub(cst(lp.alpha.c[k]), fp);
      }
      ext_mul(aloc + (rw ? A_HMW : A_HMR), d, pr);
      push(esub(pr[0], Kmem)); push(pr[1]); push(pr[2]); push(pr[3]);
    }
    // the nine piece lookups: P_k (alpha - piece_k - tag_k lambda^11) = 1
    // (on a bitwise row the slot looks up the nibble tuple (piece_k, b_k, r_k) in the operation's table instead: tag_k (1 - klg) + 8 oa + 9 oo + 10 ox)
    const E ox = esub(esub(Klg, loc[C_OA]), loc[C_OO]);
    const E lgtag = eadd(eadd(emul_f(loc[C_OA], TAG_AND), emul_f(loc[C_OO], TAG_OR)), emul_f(ox, TAG_XOR));
    for (int i = 0; i < N_PIECE; i++) {
      E d[4], pr[4];
      // (.. and on a shift row in the table shift_piece_tag names; piece 8's second element is the amount when it comes from a register)
      E tg = eadd(emul_f(esub(esub(esub(esub(one, Klg), Ksh), Kmu), Kwa), PIECE_TAG[i]), lgtag);  // (a MUL row, a wide-arithmetic row: the 10-bit range table in every slot)
      if (i == 7) tg = eadd(tg, emul_f(Ksh, TAG_NIB));
      if (i == 8) tg = eadd(tg, emul_f(esub(Ksh, loc[C_SI]), TAG_LOW6));
      for (int k = 0; k < 4; k++) d[k] = esub(esub(esub(cst(lp.alpha.c[k]), emul_f(tg, lp.lam[N_TUPLE].c[k])), emul_f(loc[C_LB + i], lp.lam[1].c[k])), emul_f(loc[C_LR + i], lp.lam[2].c[k]));
      d[0] = esub(d[0], pcs[i]);
      ext_mul(aloc + A_P + 4 * i, d, pr);
      push(esub(pr[0], one)); push(pr[1]); push(pr[2]); push(pr[3]);
    }
    // ---- 19. (mode 3) the bitwise opcodes AND OR XOR ANDI ORI XORI = 0x10 + (0 / 1 / 2) + 3 li (execute.rs:199-282), nibble by nibble ----
    const E oa = loc[C_OA], oo = loc[C_OO], li = loc[C_LI];
    boolean(Klg); boolean(oa); boolean(oo); boolean(ox); boolean(li);                            // (ox boolean: exactly one operation on a bitwise row, none elsewhere)
    push(eadd(eadd(esub(emul_f(li, 3), emul(Klg, esub(op, cst(0x10)))), oo), emul_f(ox, 2)));      // 3 li = klg (op - 0x10) - oo - 2 ox
    push(emul(li, esub(one, Klg)));
    push(emul(Klg, esub(w1, fa)));                                                               // rd = field a
    E a_lo = e_from(0), a_hi = e_from(0), b_lo = e_from(0), b_hi = e_from(0), r_lo = e_from(0), r_hi = e_from(0);
    for (int k = 0; k < 5; k++) {
      const F sh = (F)1 << (4 * k);
      a_lo = eadd(a_lo, emul_f(pcs[k], sh)); a_hi = eadd(a_hi, emul_f(k + 5 < N_PIECE ? pcs[k + 5] : R2[3], sh));        // a_9 = the last range chunk
      b_lo = eadd(b_lo, emul_f(loc[C_LB + k], sh)); b_hi = eadd(b_hi, emul_f(loc[C_LB + 5 + k], sh));
      r_lo = eadd(r_lo, emul_f(loc[C_LR + k], sh)); r_hi = eadd(r_hi, emul_f(loc[C_LR + 5 + k], sh));
    }
    push(emul(Klg, esub(xb[0], a_lo))); push(emul(Klg, esub(xb[1], a_hi)));                      // rs1's 40 bits
    push(eadd(emul(esub(Klg, li), esub(xc[0], b_lo)), emul(li, esub(im0, b_lo))));               // rs2's, or the sign-extended immediate's
    push(eadd(emul(esub(Klg, li), esub(xc[1], b_hi)), emul(li, esub(im1, b_hi))));
    push(emul(Klg, esub(y[0], r_lo))); push(emul(Klg, esub(y[1], r_hi))); push(emul(Klg, y[2]));   // the result: 40 bits
    for (int k = 0; k < N_NIB; k++) {                                                             // no second / third tuple element off the bitwise rows (b_8: nor off the register shifts)
      push(emul(k == 8 ? eadd(esub(esub(one, Klg), Ksh), loc[C_SI]) : esub(one, Klg), loc[C_LB + k])); push(emul(esub(one, Klg), loc[C_LR + k]));
    }
    // ---- 20. (mode 3) the shifts SLL SRL SRA SLLI SRLI SRAI = 0x18 + (0 / 1 / 2) + 3 si (execute.rs:284-358): a 2^t = H 2^40 + L, t = 10 u + v ----
    {
      const E* UL = loc + C_UL; const E* UR = loc + C_UR; const E* V = loc + C_V; const E* PR = loc + C_PR; const E* Rlo = loc + C_RC; const E* Rc = loc + C_RC2;
      const E sa = loc[C_SA], si = loc[C_SI], sb9 = loc[C_SB9], sgn = loc[C_SGN], shv = loc[C_SH], dd = pcs[6];
      boolean(Ksh);
      for (int u = 0; u < 5; u++) boolean(UL[u]);
      for (int u = 0; u < 5; u++) boolean(UR[u]);
      for (int v = 0; v < 10; v++) boolean(V[v]);
      boolean(sa); boolean(si); boolean(sb9);
      E sUL = e_from(0), sUR = e_from(0), sV = e_from(0), tt = e_from(0), PV = e_from(0), vsum = e_from(0);
      for (int u = 0; u < 5; u++) { sUL = eadd(sUL, UL[u]); sUR = eadd(sUR, UR[u]); tt = eadd(tt, emul_f(eadd(UL[u], UR[u]), 10 * u)); }
      for (int v = 0; v < 10; v++) { sV = eadd(sV, V[v]); tt = eadd(tt, emul_f(V[v], v)); PV = eadd(PV, emul_f(V[v], 1u << v)); if (v) vsum = eadd(vsum, emul_f(V[v], v)); }
      push(esub(eadd(sUL, sUR), Ksh)); push(esub(sV, Ksh));                                       // one chunk shift (left or right) and one bit shift on a shift row, none elsewhere
      push(esub(esub(esub(emul(Ksh, esub(op, cst(0x18))), sUR), sa), emul_f(si, 3)));             // the opcode: 0x18 + [right] + [arithmetic] + 3 [immediate]
      push(emul(sa, esub(one, sUR))); push(emul(si, esub(one, Ksh)));
      push(emul(Ksh, esub(w1, fa)));                                                              // rd = field a
      push(emul(Ksh, esub(esub(xb[0], Rc[0]), emul_f(Rc[1], RC_TABLE)))); push(emul(Ksh, esub(esub(xb[1], Rc[2]), emul_f(Rc[3], RC_TABLE))));   // a's four chunks
      for (int k = 0; k < 4; k++) push(esub(PR[k], emul(Rc[k], PV)));                             // c_k 2^v ..
      for (int k = 0; k < 4; k++) push(emul(Ksh, esub(esub(PR[k], Rlo[k]), emul_f(pcs[k], RC_TABLE))));   // .. = lo_k + 2^10 hi_k
      push(emul(Ksh, esub(esub(Rc[3], emul_f(sb9, 512)), emul(pcs[4], cst(finv(2))))));           // bit 39 of a: c_3 = 512 sb9 + piece_4 / 2
      push(esub(sgn, emul(sa, sb9)));
      // the amount: rs2's low six bits (its first chunk piece_8 with sh in LOW6, the rest of the limb in piece_5), or the word's shamt fc + 16 (fhi mod 16)
      push(emul(esub(Ksh, si), esub(esub(xc[0], pcs[8]), emul_f(pcs[5], RC_TABLE))));
      push(emul(esub(Ksh, si), esub(loc[C_LB + 8], shv)));
      push(emul(si, esub(esub(fhi, pcs[7]), emul_f(pcs[5], 16))));
      push(emul(si, esub(esub(shv, fc), emul_f(pcs[7], 16))));
      push(emul(esub(one, Ksh), shv));
      // sh = t + d on a left shift, 40 - t + d on a right shift; d (>= 0: a range lookup) only where t cannot say more: t = 40 resp. t = 0
      push(eadd(emul(sUL, esub(esub(shv, tt), dd)), emul(sUR, esub(eadd(esub(shv, cst(40)), tt), dd))));
      push(emul(dd, vsum)); push(emul(esub(sUL, UL[4]), dd)); push(emul(esub(sUR, UR[0]), dd));
      push(emul(UR[4], esub(one, V[0])));                                                        // a right shift keeps at most 40 bits: t <= 40
      // 2^40 - 2^t in limbs (right shifts): t < 20: (2^20 - 2^t, 2^20 - 1); 20 <= t < 40: (0, 2^20 - 2^(t-20)); t = 40: (0, 0)
      {
        const E low = eadd(UR[0], UR[1]);
        push(esub(loc[C_ON], esub(emul_f(low, 1u << 20), emul(eadd(UR[0], emul_f(UR[1], 1u << 10)), PV))));
        push(esub(loc[C_ON + 1], esub(esub(esub(emul_f(sUR, 1u << 20), low), emul(eadd(UR[2], emul_f(UR[3], 1u << 10)), PV)), emul_f(UR[4], 1u << 20))));
      }
      // the chunks of a 2^v: m_0 = lo_0, m_i = lo_i + hi_(i-1), m_4 = hi_3; result chunk j = m_(j-u) on a left shift, m_(j+4-u) on a right shift
      E m[5] = {Rlo[0], eadd(Rlo[1], pcs[0]), eadd(Rlo[2], pcs[1]), eadd(Rlo[3], pcs[2]), pcs[3]}, res[4];
      for (int j = 0; j < 4; j++) {
        res[j] = e_from(0);
        for (int u = 0; u < 5; u++) { if (j - u >= 0) res[j] = eadd(res[j], emul(UL[u], m[j - u])); if (j + 4 - u >= 0 && j + 4 - u <= 4) res[j] = eadd(res[j], emul(UR[u], m[j + 4 - u])); }
      }
      push(esub(esub(emul(Ksh, y[0]), eadd(res[0], emul_f(res[1], RC_TABLE))), emul(sgn, loc[C_ON])));
      push(esub(esub(emul(Ksh, y[1]), eadd(res[2], emul_f(res[3], RC_TABLE))), emul(sgn, loc[C_ON + 1])));
      push(emul(Ksh, y[2]));
    }
    // ---- 21. (mode 3) MUL (execute.rs:79-99): the product of the 40-bit operands mod 2^40, schoolbook in 10-bit chunks ----
    {
      const E* Rlo = loc + C_RC; const E* Rc = loc + C_RC2; const E* ma = loc + C_MA; const E* me = loc + C_ME;
      boolean(Kmu); boolean(me[0]); boolean(me[1]); boolean(me[2]);
      push(emul(Kmu, esub(w1, fa)));                                                              // rd = field a
      push(emul(Kmu, esub(esub(xb[0], Rc[0]), emul_f(Rc[1], RC_TABLE)))); push(emul(Kmu, esub(esub(xb[1], Rc[2]), emul_f(Rc[3], RC_TABLE))));       // a's four chunks
      push(emul(Kmu, esub(esub(xc[0], pcs[0]), emul_f(pcs[1], RC_TABLE)))); push(emul(Kmu, esub(esub(xc[1], pcs[2]), emul_f(pcs[3], RC_TABLE))));   // b's
      for (int k = 0; k < 4; k++) push(esub(ma[k], emul(Kmu, Rc[k])));                            // ma_k = kmu a_k
      const E carry[4] = {pcs[4], eadd(pcs[5], emul_f(me[0], RC_TABLE)), eadd(pcs[6], emul_f(eadd(me[1], emul_f(me[2], 2)), RC_TABLE)), eadd(pcs[7], emul_f(pcs[8], RC_TABLE))};
      for (int k = 0; k < 4; k++) {                                                               // sum_{i+j=k} a_i b_j + carry_(k-1) = r_k + 2^10 carry_k
        E t = e_from(0);
        for (int j = 0; j <= k; j++) t = eadd(t, emul(ma[j], pcs[k - j]));
        E lin = eadd(Rlo[k], emul_f(carry[k], RC_TABLE));
        if (k) lin = esub(lin, carry[k - 1]);
        push(esub(t, emul(Kmu, lin)));
      }
      push(emul(Kmu, esub(y[0], eadd(Rlo[0], emul_f(Rlo[1], RC_TABLE))))); push(emul(Kmu, esub(y[1], eadd(Rlo[2], emul_f(Rlo[3], RC_TABLE))))); push(emul(Kmu, y[2]));
    }
    // ---- 22. (mode 4, round 6) MULH DIVU REMU DIV REM on operands below 2^40 (execute.rs:101-183): F1 F2 + ADD = LO + 2^40 HI, schoolbook in 10-bit chunks.  Appended: mode 3 stops above. ----
    if (WIDE) {
      const E* Rlo = loc + C_RC; const E* Rf = loc + C_RC2; const E* gf = loc + C_GF; const E* we = loc + C_WE; const E* X = loc + C_X;
      const E om = loc[C_OM], od = loc[C_OD], orr = loc[C_ORR], kd = eadd(od, orr);
      boolean(Kwa); boolean(om); boolean(od); boolean(orr);                                       // (kwa boolean: at most one of the four kinds — ot's own boolean is constraint 25)
      for (int k = 0; k < N_WE; k++) boolean(we[k]);
      push(esub(emul(Kin, esub(op, emul_f(loc[C_G], 2))), eadd(eadd(emul_f(om, 3), emul_f(od, 4)), emul_f(orr, 5))));   // the opcode: MULH 3, DIVU 4, REMU 5, DIV 6 = 4 + 2 g, REM 7 = 5 + 2 g (g: the word's variant bit, from the ROM)
      push(emul(Kwa, esub(w1, fa)));                                                              // rd = field a
      push(emul(Kin, xb[2])); push(emul(Kin, xc[2]));                                             // the chunk relation's operands are below 2^40 (what makes DIV = DIVU, REM = REMU, MULH the product's bits 40..79); wider ones go through the tape (25)
      const E two10 = cst(RC_TABLE);
      push(emul(Kin, esub(esub(xc[0], pcs[0]), emul(two10, pcs[1])))); push(emul(Kin, esub(esub(xc[1], pcs[2]), emul(two10, pcs[3]))));      // F2 = rs2, always
      push(emul(om, esub(esub(xb[0], Rf[0]), emul(two10, Rf[1])))); push(emul(om, esub(esub(xb[1], Rf[2]), emul(two10, Rf[3]))));            // MULH: F1 = rs1
      push(emul(kd, esub(esub(xb[0], Rlo[0]), emul(two10, Rlo[1])))); push(emul(kd, esub(esub(xb[1], Rlo[2]), emul(two10, Rlo[3]))));        // divisions: LO = rs1, the dividend
      for (int k = 0; k < 4; k++) push(esub(gf[k], emul(Kin, Rf[k])));                            // gf_k = (om + od + orr) F1_k
      const E G4[4] = {pcs[7], pcs[8], X[0], X[1]};
      const E c[6] = {pcs[4], eadd(pcs[5], emul(two10, we[0])), eadd(pcs[6], emul(two10, eadd(we[1], emul_f(we[2], 2)))),
                      eadd(X[2], emul(two10, eadd(we[3], emul_f(we[4], 2)))), eadd(X[3], emul(two10, eadd(we[5], emul_f(we[6], 2)))), eadd(X[4], emul(two10, eadd(we[7], emul_f(we[8], 2))))};
      for (int k = 0; k < 7; k++) {
        E t = e_from(0);
        for (int j = 0; j < 4; j++) if (k - j >= 0 && k - j < 4) t = eadd(t, emul(gf[j], pcs[k - j]));
        if (k < 4) {                                                                              // low half: + ADD_k + c_(k-1) = LO_k + 2^10 c_k; the carry OUT of position 3 exists on MULH rows only
          t = eadd(t, emul(kd, G4[k]));
          E lin = Rlo[k];
          if (k < 3) lin = eadd(lin, emul(two10, c[k]));
          if (k) lin = esub(lin, c[k - 1]);
          t = esub(t, emul(Kin, lin));
          if (k == 3) t = esub(t, emul(om, emul(two10, c[3])));
        } else {                                                                                  // high half (MULH): + c_(k-1) = HI_(k-4) + 2^10 c_k (k = 6: 2^10 HI_3); a division has nothing there
          const E hi = k < 6 ? eadd(G4[k - 4], emul(two10, c[k])) : eadd(G4[2], emul(two10, G4[3]));
          t = eadd(t, emul(om, esub(c[k - 1], hi)));
        }
        push(t);
      }
      // divisions: the remainder is smaller than the divisor: d = rs2 - r - 1 >= 0 in chunks X2..X5, borrow e4 between the limbs (which also says rs2 != 0)
      const E r0 = eadd(G4[0], emul(two10, G4[1])), r1 = eadd(G4[2], emul(two10, G4[3])), d0 = eadd(X[2], emul(two10, X[3])), d1 = eadd(X[4], emul(two10, X[5]));
      push(emul(kd, eadd(esub(esub(esub(xc[0], r0), one), d0), emul(two20, we[3]))));
      push(emul(kd, esub(esub(esub(xc[1], r1), we[3]), d1)));
      // what is written: HI (MULH) and the remainder (REMU / REM) sit in G4, the quotient (DIVU / DIV) in F1
      const E g = eadd(om, orr);
      push(emul(g, esub(y[0], r0))); push(emul(g, esub(y[1], r1)));
      push(emul(od, esub(esub(y[0], Rf[0]), emul(two10, Rf[1])))); push(emul(od, esub(esub(y[1], Rf[2]), emul(two10, Rf[3]))));
      push(emul(Kin, y[2]));
      // the six extra range slots: XH_i (alpha - X_i) = 1, on every row
      for (int i = 0; i < N_X; i++) {
        E d[4], pr[4];
        for (int k = 0; k < 4; k++) d[k] = cst(lp.alpha.c[k]);
        d[0] = esub(d[0], X[i]);
        ext_mul(aloc + A_X + 4 * i, d, pr);
        push(esub(pr[0], one)); push(pr[1]); push(pr[2]); push(pr[3]);
      }
      // ---- 23. (mode 4) the boundary cell: no store writes the low half of cell B — nb = delta iws, delta = the row's cell address minus B as ONE field element; tl (kst - nb) = 0
      {
        const uint64_t Bc = boundary_cell(pub.blob, pub.blob_len);
        E tl = e_from(0);
        for (int v = 0; v < N_WIN; v++) if (is_low_window(v)) tl = eadd(tl, Ev[v]);
        const E delta = eadd(esub(esub(z[0], off), cst(Bc & 0xFFFFF)), emul(two20, esub(z[1], cst((Bc >> 20) & 0xFFFFF))));
        push(esub(loc[C_NB], emul(delta, loc[C_IWS])));
        push(emul(tl, esub(Kst, loc[C_NB])));
      }
      // ---- 24. (mode 4) hash syscalls: HH (alpha - fp(cycle, R11's limbs, R12's, R13's, 3 + h0 + 2 h1) - 12 lambda^11) = fh: the call's record is in the tape the proof carries
      {
        E d[4], pr[4];
        const E kind = eadd(eadd(emul_f(loc[C_FH], 3), loc[C_H0]), emul_f(loc[C_H1], 2));           // (h0 = h1 = 0 off the hash rows, where the helper is zero anyway)
        for (int k = 0; k < 4; k++) {
          E fp = eadd(emul_f(cst(lp.lam[N_TUPLE].c[k]), TAG_HASH), emul_f(loc[C_CYCLE], lp.lam[0].c[k]));
          for (int j = 0; j < 9; j++) fp = eadd(fp, emul_f(loc[C_LIMB + 33 + j], lp.lam[1 + j].c[k]));
          fp = eadd(fp, emul_f(kind, lp.lam[10].c[k]));
          d[k] = esub(cst(lp.alpha.c[k]), fp);
        }
        ext_mul(aloc + A_HH, d, pr);
        push(esub(pr[0], loc[C_FH])); push(pr[1]); push(pr[2]); push(pr[3]);
      }
      // ---- 25. (mode 4 d) the wide tape: ot boolean; WW (alpha - fp(cycle, rs1's limbs, rs2's, y's, opcode) - 13 lambda^11) = ot: a wide row outside the chunk relation's domain
      // writes what the verifier computed from its record
      {
        boolean(loc[C_OT]);
        E d[4], pr[4];
        const E* tup[10] = {&loc[C_CYCLE], &xb[0], &xb[1], &xb[2], &xc[0], &xc[1], &xc[2], &y[0], &y[1], &y[2]};
        for (int k = 0; k < 4; k++) {
          E fp = emul_f(cst(lp.lam[N_TUPLE].c[k]), TAG_WIDE);
          for (int j = 0; j < 10; j++) fp = eadd(fp, emul_f(*tup[j], lp.lam[j].c[k]));
          fp = eadd(fp, emul_f(op, lp.lam[10].c[k]));
          d[k] = esub(cst(lp.alpha.c[k]), fp);
        }
        ext_mul(aloc + A_WW, d, pr);
        push(esub(pr[0], loc[C_OT])); push(pr[1]); push(pr[2]); push(pr[3]);
      }
    }
  }
  result = A.acc;
  return A.c;
}
static int num_constraints(int mode = 0) {                                // by a dry run (the list above is the definition)
  std::vector<E> z(W_MAX, e_from(0)), ap(MAX_CONSTRAINTS, e_from(0));
  E r; Public pub; LookupParams lp{};
  pub.deferred = (uint32_t)mode;
  return constraints_sum(z.data(), z.data(), z.data(), z.data(), e_from(0), e_from(0), e_from(0), pub, lp, ap.data(), r);
}

struct Proof { std::vector<uint32_t> w; };
static void put_e(std::vector<uint32_t>& w, const E& e) { for (int i = 0; i < 4; i++) w.push_back(e.c[i]); }

static void merkle_path(const Merkle& t, size_t leaf, std::vector<uint32_t>& w) {
  size_t j = leaf;
  for (size_t lv = 0; lv + 1 < t.layers.size(); lv++) { const F* sib = &t.layers[lv][4 * (j ^ 1)]; for (int i = 0; i < 4; i++) w.push_back(sib[i]); j >>= 1; }
}
static E fold_pair(const E& a, const E& b, F x, const E& beta) {          // (a+b)/2 + beta (a-b)/(2x)
  const F half = finv(2), inv2x = finv(fmul(2, x));
  return eadd(emul_f(eadd(a, b), half), emul(beta, emul_f(esub(a, b), inv2x)));
}
static E horner_base(const std::vector<F>& coeffs, const E& z) { E acc = e_from(0); for (size_t k = coeffs.size(); k-- > 0;) acc = eadd(emul(acc, z), e_from(coeffs[k])); return acc; }

// header words 2..20 = everything both sides know before the first commitment; observed by the transcript in this order
static void header_words(int log_n, const Public& pub, std::vector<uint32_t>& w) {
  w.clear();
  w.push_back(PROOF_MAGIC); w.push_back(proof_version(pub.mode())); w.push_back(log_n); w.push_back(phys_width(pub.mode())); w.push_back((uint32_t)pub.num_queries()); w.push_back(LOG_FINAL); w.push_back((uint32_t)pub.pow_bits());
  w.push_back((uint32_t)(pub.n_real & 0x3FFFFFFF)); w.push_back((uint32_t)(pub.n_real >> 30)); w.push_back((uint32_t)pub.mode());
  w.push_back((uint32_t)(pub.entry & 0xFFFFF)); w.push_back((uint32_t)((pub.entry >> 20) & 0xFFFFF)); w.push_back((uint32_t)(pub.entry >> 40));
  for (int i = 0; i < 4; i++) w.push_back(pub.prog[i]);
  for (int i = 0; i < 4; i++) w.push_back(pub.io[i]);
  for (int i = 0; i < N_STATE; i++) w.push_back(pub.first[i]);
  for (int i = 0; i < N_STATE; i++) w.push_back(pub.last[i]);
  if (pub.mode() >= 2) { w.push_back(pub.cnt_first[0]); w.push_back(pub.cnt_first[1]); w.push_back(pub.cnt_last[0]); w.push_back(pub.cnt_last[1]); }   // (oc, ic) of the first / last row
}
static inline int header_words_of(int mode) { return HEADER_WORDS + (mode >= 2 ? 4 : 0); }
// (mode 3) the memory section of a proof, after the I/O section: [n_cells] then per touched cell [address limb 0 (20 bits, a multiple of 8)] [address limb 1 (20 bits)]
// [time of the last access] [the final bytes: four 16-bit pieces], by strictly increasing address
static void put_u64(std::vector<uint32_t>& w, uint64_t v);
static void mem_section(const Public& pub, std::vector<uint32_t>& w) {
  w.push_back((uint32_t)pub.cells.size());
  for (const Public::Cell& c : pub.cells) { w.push_back((uint32_t)(c.addr & 0xFFFFF)); w.push_back((uint32_t)((c.addr >> 20) & 0xFFFFF)); w.push_back(c.t); put_u64(w, c.bytes); }
}
// (mode 2) the I/O section of a proof, after the program: [n_in] [inputs: four 16-bit pieces each] [n_out] [outputs] [halt kind] [halt code: four pieces]
// (mode 4) the hash calls: [n] then per call [cycle] [input pointer: two 20-bit limbs] [input length] [output pointer: two limbs] [kind] [touched cells] and per touched cell
// (ascending; the addresses follow from the pointers) [time of its previous access] [its bytes before the call: four 16-bit pieces]
static void hash_section(const Public& pub, std::vector<uint32_t>& w) {
  w.push_back((uint32_t)pub.hcalls.size());
  for (const Public::HashCall& c : pub.hcalls) {
    w.push_back((uint32_t)c.cycle); w.push_back((uint32_t)(c.in_ptr & 0xFFFFF)); w.push_back((uint32_t)(c.in_ptr >> 20)); w.push_back((uint32_t)c.len);
    w.push_back((uint32_t)(c.out_ptr & 0xFFFFF)); w.push_back((uint32_t)(c.out_ptr >> 20)); w.push_back(c.kind); w.push_back((uint32_t)c.cells.size());
    for (const Public::Cell& x : c.cells) { w.push_back(x.t); for (int i = 0; i < 4; i++) w.push_back((uint32_t)((x.bytes >> (16 * i)) & 0xFFFF)); }
  }
}
// (mode 4 d) the wide tape as a proof section: [n] then per record [cycle] [rs1: limbs of 20, 20, 24 bits] [rs2: likewise] [opcode]
static void wide_section(const Public& pub, std::vector<uint32_t>& w) {
  w.push_back((uint32_t)pub.wrecs.size());
  for (const Public::WideRec& c : pub.wrecs) {
    w.push_back((uint32_t)c.cycle);
    w.push_back((uint32_t)(c.a & 0xFFFFF)); w.push_back((uint32_t)((c.a >> 20) & 0xFFFFF)); w.push_back((uint32_t)(c.a >> 40));
    w.push_back((uint32_t)(c.b & 0xFFFFF)); w.push_back((uint32_t)((c.b >> 20) & 0xFFFFF)); w.push_back((uint32_t)(c.b >> 40));
    w.push_back(c.op);
  }
}
static void put_u64(std::vector<uint32_t>& w, uint64_t v) { for (int i = 0; i < 4; i++) w.push_back((uint32_t)((v >> (16 * i)) & 0xFFFF)); }
static void io_section(const Public& pub, std::vector<uint32_t>& w) {
  w.push_back((uint32_t)pub.n_in); for (size_t i = 0; i < pub.n_in; i++) put_u64(w, pub.inputs[i]);
  w.push_back((uint32_t)pub.n_out); for (size_t i = 0; i < pub.n_out; i++) put_u64(w, pub.outputs[i]);
  w.push_back(pub.halt_kind); put_u64(w, pub.halt_kind == 1 ? pub.halt_code : 0);
}
// the VM's initial state at `entry` (VMState::new, state.rs:55-71): cycle 0, pc = entry, all registers zero and Normalized
static void initial_state(uint64_t entry, F st[N_STATE]) {
  memset(st, 0, N_STATE * sizeof(F));
  st[1] = (F)(entry & 0xFFFFF); st[2] = (F)((entry >> 20) & 0xFFFFF); st[3] = (F)(entry >> 40);
}

struct ProverTrace {     // everything the oracle keeps for inspection by tests
  std::vector<F> M, Mp, L, Qc;        // logical [W_MAIN][N], committed [Wm][N], its LDE [Wm][2N], [4][2N]
  std::vector<F> A, AL;               // aux trace [W_AUX][N] and its LDE [W_AUX][2N]
  std::vector<F> rom_mult, rc_mult, mem_mult;
  LookupParams lp;
  Merkle trace_tree, aux_tree, quot_tree;
  std::vector<std::vector<E>> fri;    // codewords per layer (layer 0 = DEEP codeword, size 2N)
  std::vector<Merkle> fri_trees;
  E alpha, zeta, gamma; std::vector<E> betas; std::vector<uint32_t> queries; F pow_nonce = 0;
};

// matrix_override (tests): prove this main-trace matrix [W][N] instead of the one derived from the rows (a cheating prover)
static void prove(const PackedRow* rows, const Public& pub_in, Proof& proof, ProverTrace& pt, const F* matrix_override = nullptr) {
  Public pub = pub_in;
  const int log_n = padded_log_n(pub.n_real);
  const size_t N = (size_t)1 << log_n, N2 = 2 * N;
  const int Dm = pub.mode();                                              // 0 default, 1 deferred, 2 default + the I/O argument
  const int Wm = phys_width(Dm), Wl = logical_width(Dm), Wa = aux_width(Dm);
  if (matrix_override) pt.M.assign(matrix_override, matrix_override + (size_t)Wl * N);         // LOGICAL; whatever it holds in uncommitted columns is dropped
  else main_trace(rows, pub.n_real, pub, pt.M, &pub.cells, &pub.hcalls, &pub.wrecs);                                   // (mode 3: with the touched cells; an override brings its own in pub_in.cells)
  if (matrix_override && Dm == 4) {                                       // (the wide tape of an overridden matrix: read off its ot rows — the prover's tape always matches its rows)
    pub.wrecs.clear();
    auto m = [&](int c, size_t i) { return (uint64_t)pt.M[(size_t)c * N + i]; };
    for (size_t i = 0; i < pub.n_real; i++) if (m(C_OT, i))
      pub.wrecs.push_back(Public::WideRec{m(C_CYCLE, i), m(C_XB, i) | (m(C_XB + 1, i) << 20) | (m(C_XB + 2, i) << 40), m(C_XC, i) | (m(C_XC + 1, i) << 20) | (m(C_XC + 2, i) << 40), (uint32_t)m(C_OP, i)});
  }
  for (int c = 0; c < Wl; c++) if (is_virtual(c, Dm)) std::fill(pt.M.begin() + (size_t)c * N, pt.M.begin() + (size_t)(c + 1) * N, 0);
  to_physical(pt.M, N, Dm, pt.Mp);
  for (int i = 0; i < N_STATE; i++) {                                     // boundary states: rows 0 and n_real - 1 of the matrix being proven
    pub.first[i] = pt.M[(size_t)state_col(i) * N];
    pub.last[i] = pt.M[(size_t)state_col(i) * N + (pub.n_real - 1)];
  }
  if (Dm >= 2) for (int k = 0; k < 2; k++) { pub.cnt_first[k] = pt.M[(size_t)(C_OC + k) * N]; pub.cnt_last[k] = pt.M[(size_t)(C_OC + k) * N + (pub.n_real - 1)]; }
  pt.L.assign((size_t)Wm * N2, 0);
  std::vector<std::vector<F>> coeffs(Wm);
  for (int k = 0; k < Wm; k++) {
    std::vector<F> e(pt.Mp.begin() + (size_t)k * N, pt.Mp.begin() + (size_t)(k + 1) * N), o;
    lde(e, 1, coeffs[k], o);
    memcpy(&pt.L[(size_t)k * N2], o.data(), N2 * 4);
  }
  merkle_build(pt.L, Wm, N2, pt.trace_tree);
  std::vector<uint32_t>& w = proof.w;
  header_words(log_n, pub, w);
  Challenger ch;
  ch.observe_n(w.data() + 2, w.size() - 2);
  ch.observe_n(pt.trace_tree.layers.back().data(), 4);
  // ---- the program (its code words are the instruction ROM) and the lookup multiplicities, fixed BEFORE the lookup challenges ----
  const Rom rom = rom_from_blob(pub.blob, pub.blob_len, Dm);
  w.push_back((uint32_t)pub.blob_len);
  for (size_t i = 0; i < pub.blob_len; i += 2) w.push_back((uint32_t)pub.blob[i] | (i + 1 < pub.blob_len ? (uint32_t)pub.blob[i + 1] << 8 : 0u));
  if (Dm >= 2) { const size_t at = w.size(); io_section(pub, w); observe_section(ch, w.data() + at, w.size() - at); }   // the tapes and the halt reason: what the io digest is a digest of; (v11) fixed BEFORE the lookup challenges — a SEGMENT's tapes too, whose digest only the chain checks
  if (Dm >= 3) { const size_t at = w.size(); mem_section(pub, w); observe_section(ch, w.data() + at, w.size() - at); }   // the touched cells, fixed BEFORE the lookup challenges like the multiplicities
  if (Dm == 4) { const size_t at = w.size(); hash_section(pub, w); observe_section(ch, w.data() + at, w.size() - at); }   // (mode 4) the hash calls, likewise
  if (Dm == 4) { const size_t at = w.size(); wide_section(pub, w); observe_section(ch, w.data() + at, w.size() - at); }   // .. and the wide tape
  lookup_multiplicities(pt.M, N, rom, pt.rom_mult, pt.rc_mult, nullptr, Dm, &pt.mem_mult);
  w.insert(w.end(), pt.rom_mult.begin(), pt.rom_mult.end());
  w.insert(w.end(), pt.rc_mult.begin(), pt.rc_mult.end());
  ch.observe_n(pt.rom_mult.data(), pt.rom_mult.size());
  ch.observe_n(pt.rc_mult.data(), pt.rc_mult.size());
  if (Dm >= 3) { w.insert(w.end(), pt.mem_mult.begin(), pt.mem_mult.end()); ch.observe_n(pt.mem_mult.data(), pt.mem_mult.size()); }
  pt.lp.alpha = ch.sample_ext();
  {
    const E lambda = ch.sample_ext();
    pt.lp.lam[0] = e_from(1);
    for (int j = 1; j <= N_TUPLE; j++) pt.lp.lam[j] = emul(pt.lp.lam[j - 1], lambda);
  }
  pt.lp.n_in = (F)(pub.n_in % P);
  {
    E T = lookup_table_sum(rom, pt.rom_mult.data(), pt.rc_mult.data(), pt.lp, Dm >= 3 ? pt.mem_mult.data() : nullptr);
    if (Dm >= 2) T = eadd(T, io_table_sum(pub, pt.lp));
    if (Dm >= 3) T = eadd(T, mem_table_sum(pub, pt.lp));
    if (Dm == 4) T = eadd(eadd(T, hash_table_sum(pub, pt.lp)), wide_table_sum(pub, pt.lp));
    pt.lp.t_over_n = emul_f(T, finv((F)(N % P)));
  }
  // ---- aux trace: helper columns + running sum; its own LDE and commitment ----
  aux_trace(pt.M, N, pt.lp, pt.A, Dm);
  pt.AL.assign((size_t)Wa * N2, 0);
  std::vector<std::vector<F>> acoeffs(Wa);
  for (int k = 0; k < Wa; k++) {
    std::vector<F> e(pt.A.begin() + (size_t)k * N, pt.A.begin() + (size_t)(k + 1) * N), o;
    lde(e, 1, acoeffs[k], o);
    memcpy(&pt.AL[(size_t)k * N2], o.data(), N2 * 4);
  }
  merkle_build(pt.AL, Wa, N2, pt.aux_tree);
  ch.observe_n(pt.aux_tree.layers.back().data(), 4);
  pt.alpha = ch.sample_ext();
  const int NC = num_constraints(Dm);
  std::vector<E> ap(NC); ap[0] = e_from(1); for (int c = 1; c < NC; c++) ap[c] = emul(ap[c - 1], pt.alpha);

  // ---- quotient on the LDE coset: x_j = g * w_2N^j, next row = position j + 2 ----
  const F w2n = root_of_unity(log_n + 1), wn = root_of_unity(log_n), wn_inv = finv(wn);
  const F w_last = fpow(wn, pub.n_real - 1);
  const F gN = fpow(GEN, N);
  pt.Qc.assign(4 * N2, 0);
  {
    F x = GEN;
    std::vector<E> ploc(Wm), pnxt(Wm), loc(W_MAX), nxt(W_MAX), aloc(W_AUX_MAX), anxt(W_AUX_MAX);
    for (size_t j = 0; j < N2; j++) {
      for (int k = 0; k < Wm; k++) { ploc[k] = e_from(pt.L[(size_t)k * N2 + j]); pnxt[k] = e_from(pt.L[(size_t)k * N2 + ((j + 2) & (N2 - 1))]); }
      to_logical_row(ploc.data(), Dm, e_from(0), loc.data()); to_logical_row(pnxt.data(), Dm, e_from(0), nxt.data());
      for (int k = 0; k < Wa; k++) { aloc[k] = e_from(pt.AL[(size_t)k * N2 + j]); anxt[k] = e_from(pt.AL[(size_t)k * N2 + ((j + 2) & (N2 - 1))]); }
      const F zh = fsub((j & 1) ? fneg(gN) : gN, 1);                      // x^N - 1, x^N = g^N (-1)^j
      const F inv_zh = finv(zh);
      const E is_first = e_from(fmul(zh, finv(fsub(x, 1))));
      const E is_last = e_from(fmul(zh, finv(fsub(x, w_last))));
      const E is_trans = e_from(fsub(x, wn_inv));
      E sum; constraints_sum(loc.data(), nxt.data(), aloc.data(), anxt.data(), is_first, is_last, is_trans, pub, pt.lp, ap.data(), sum);
      E q = emul_f(sum, inv_zh);
      for (int i = 0; i < 4; i++) pt.Qc[(size_t)i * N2 + j] = q.c[i];
      x = fmul(x, w2n);
    }
  }
  merkle_build(pt.Qc, 4, N2, pt.quot_tree);
  ch.observe_n(pt.quot_tree.layers.back().data(), 4);
  pt.zeta = ch.sample_ext();
  const E zeta_w = emul_f(pt.zeta, wn);

  // ---- openings (oracle: Horner on coefficient vectors) ----
  // column order everywhere below (openings, gamma powers): main columns, then aux columns = W_ALL "trace" columns
  const int Wt = Wm + Wa;
  std::vector<E> t_z(Wt), t_zw(Wt), q_z(4);
  for (int k = 0; k < Wm; k++) { t_z[k] = horner_base(coeffs[k], pt.zeta); t_zw[k] = horner_base(coeffs[k], zeta_w); }
  for (int k = 0; k < Wa; k++) { t_z[Wm + k] = horner_base(acoeffs[k], pt.zeta); t_zw[Wm + k] = horner_base(acoeffs[k], zeta_w); }
  for (int i = 0; i < 4; i++) {                                           // quotient columns: interpolate from the coset evaluations
    std::vector<F> ev(pt.Qc.begin() + (size_t)i * N2, pt.Qc.begin() + (size_t)(i + 1) * N2);
    ntt(ev, true);                                                        // coefficients of q_i(g x)
    F ginv = finv(GEN), sc = 1;
    for (size_t k2 = 0; k2 < N2; k2++) { ev[k2] = fmul(ev[k2], sc); sc = fmul(sc, ginv); }
    q_z[i] = horner_base(ev, pt.zeta);
  }
  for (int k = 0; k < Wt; k++) ch.observe_ext(t_z[k]);
  for (int k = 0; k < Wt; k++) ch.observe_ext(t_zw[k]);
  for (int i = 0; i < 4; i++) ch.observe_ext(q_z[i]);
  pt.gamma = ch.sample_ext();

  // ---- DEEP codeword over the LDE coset ----
  std::vector<E> gp(2 * Wt + 4); gp[0] = e_from(1); for (size_t k = 1; k < gp.size(); k++) gp[k] = emul(gp[k - 1], pt.gamma);
  E a0 = e_from(0), b0 = e_from(0);
  for (int k = 0; k < Wt; k++) { a0 = eadd(a0, emul(gp[k], t_z[k])); b0 = eadd(b0, emul(gp[Wt + k], t_zw[k])); }
  for (int i = 0; i < 4; i++) a0 = eadd(a0, emul(gp[2 * Wt + i], q_z[i]));
  std::vector<E> cw(N2);
  {
    F x = GEN;
    for (size_t j = 0; j < N2; j++) {
      E A = e_from(0), B = e_from(0);
      for (int k = 0; k < Wm; k++) { const F v = pt.L[(size_t)k * N2 + j]; A = eadd(A, emul_f(gp[k], v)); B = eadd(B, emul_f(gp[Wt + k], v)); }
      for (int k = 0; k < Wa; k++) { const F v = pt.AL[(size_t)k * N2 + j]; A = eadd(A, emul_f(gp[Wm + k], v)); B = eadd(B, emul_f(gp[Wt + Wm + k], v)); }
      for (int i = 0; i < 4; i++) A = eadd(A, emul_f(gp[2 * Wt + i], pt.Qc[(size_t)i * N2 + j]));
      const E d1 = einv(esub(e_from(x), pt.zeta)), d2 = einv(esub(e_from(x), zeta_w));
      cw[j] = eadd(emul(esub(A, a0), d1), emul(esub(B, b0), d2));
      x = fmul(x, w2n);
    }
  }

  // ---- FRI commit phase ----
  pt.fri.clear(); pt.fri_trees.clear(); pt.betas.clear();
  pt.fri.push_back(cw);
  const std::vector<int> ks = fri_schedule(log_n);
  F shift = GEN; int log_m = log_n + 1;
  for (const int k : ks) {
    const std::vector<E>& c = pt.fri.back();
    const size_t m = c.size(), g = m >> k, nv = (size_t)1 << k;              // g leaves of 2^k values: leaf i = (c[i + t g])_t
    std::vector<F> mat(4 * nv * g);                                         // 4 * 2^k base elements per leaf, column-major [4 * 2^k][g]
    for (size_t i = 0; i < g; i++) for (size_t t = 0; t < nv; t++) for (int e = 0; e < 4; e++) mat[(t * 4 + e) * g + i] = c[i + t * g].c[e];
    Merkle tr; merkle_build(mat, (int)(4 * nv), g, tr);
    ch.observe_n(tr.layers.back().data(), 4);
    E beta = ch.sample_ext();
    pt.betas.push_back(beta);
    std::vector<E> cur = c;
    for (int f = 0; f < k; f++) {                                           // binary folds with beta^(2^f); the domain shift squares each time
      const size_t h = cur.size() / 2;
      std::vector<E> nx(h);
      const F wm = root_of_unity(log_m);
      F x = shift;
      for (size_t i = 0; i < h; i++) { nx[i] = fold_pair(cur[i], cur[i + h], x, beta); x = fmul(x, wm); }
      cur.swap(nx);
      shift = fmul(shift, shift); log_m--; beta = emul(beta, beta);
    }
    pt.fri_trees.push_back(std::move(tr));
    pt.fri.push_back(std::move(cur));
  }
  const std::vector<E>& fin = pt.fri.back();
  for (const E& e : fin) ch.observe_ext(e);
  pt.pow_nonce = ch.grind(pub.pow_bits());
  const bool pow_ok = ch.check_pow(pt.pow_nonce, pub.pow_bits()); (void)pow_ok;
  pt.queries.clear();
  for (int t = 0; t < pub.num_queries(); t++) pt.queries.push_back(ch.sample_bits(log_n));

  // ---- serialize: header | program (length, halfwords) | ROM multiplicities | range multiplicities (all pushed above) | trace root | aux root |
  //      quotient root | openings (main + aux at zeta, main + aux at zeta w, quotient) | FRI roots | final codeword | pow nonce | queries ----
  for (int i = 0; i < 4; i++) w.push_back(pt.trace_tree.layers.back()[i]);
  for (int i = 0; i < 4; i++) w.push_back(pt.aux_tree.layers.back()[i]);
  for (int i = 0; i < 4; i++) w.push_back(pt.quot_tree.layers.back()[i]);
  for (int k = 0; k < Wt; k++) put_e(w, t_z[k]);
  for (int k = 0; k < Wt; k++) put_e(w, t_zw[k]);
  for (int i = 0; i < 4; i++) put_e(w, q_z[i]);
  w.push_back((uint32_t)pt.fri_trees.size());
  for (auto& tr : pt.fri_trees) for (int i = 0; i < 4; i++) w.push_back(tr.layers.back()[i]);
  for (const E& e : fin) put_e(w, e);
  w.push_back(pt.pow_nonce);
  for (uint32_t q : pt.queries) {
    w.push_back(q);
    for (size_t pos : {(size_t)q, (size_t)q + N}) { for (int k = 0; k < Wm; k++) w.push_back(pt.L[(size_t)k * N2 + pos]); merkle_path(pt.trace_tree, pos, w); }
    for (size_t pos : {(size_t)q, (size_t)q + N}) { for (int k = 0; k < Wa; k++) w.push_back(pt.AL[(size_t)k * N2 + pos]); merkle_path(pt.aux_tree, pos, w); }
    for (size_t pos : {(size_t)q, (size_t)q + N}) { for (int i = 0; i < 4; i++) w.push_back(pt.Qc[(size_t)i * N2 + pos]); merkle_path(pt.quot_tree, pos, w); }
    for (size_t j = 0; j < pt.fri_trees.size(); j++) {
      const size_t g = pt.fri[j].size() >> ks[j], idx = q & (g - 1);
      for (size_t t = 0; t < ((size_t)1 << ks[j]); t++) put_e(w, pt.fri[j][idx + t * g]);
      merkle_path(pt.fri_trees[j], idx, w);
    }
  }
}

// ---- the same proof, MEMORY-LEAN and THREADED (round 6: the whole-proof golden at BASELINE configs[2]'s own size, 2^24 rows) ----
// `prove` above is the definition; this is the same sequence of the same functions (so::lde, hash_elems, compress, constraints_sum, fold_pair, the same Challenger) with
//  * nothing kept for inspection: no committed copy of the matrix (a column is extended straight from its logical column), no coefficient vectors (the opening of a column at
//    zeta / zeta w is Horner on the coefficients RE-DERIVED from the even positions of its LDE: p(g w_N^i) = the LDE at 2 i, an inverse NTT gives the coefficients of p(g x),
//    times g^-k — field arithmetic is exact, so these are the very words `lde` produced), the logical matrix freed once the aux trace exists;
//  * `threads` std::threads over independent units: columns (LDE, openings), leaves / tree nodes (Merkle), coset points (quotient, DEEP codeword), fold outputs.
// What each unit computes is untouched, so the proof is word for word `prove`'s (tests/test_stark_oracle.py::test_lean_prover_equals_the_plain_one: every mode, ragged runs).
// Peak memory at 2^24 rows, mode 0: the rows (6.2 GB) + the logical matrix (11.5 GB) + the main LDE (20.4 GB) = 38 GB, against ~70 GB for `prove`.
template <class Body>
static void par_for(int threads, size_t count, Body&& body) {                // body(lo, hi) over [0, count) split among the threads
  if (threads <= 1 || count < 2) { if (count) body((size_t)0, count); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) { const size_t lo = count * t / threads, hi = count * (t + 1) / threads; if (lo < hi) th.emplace_back([lo, hi, &body] { body(lo, hi); }); }
  for (auto& x : th) x.join();
}
static void merkle_build_par(const std::vector<F>& mat, int width, size_t n, Merkle& t, int threads) {
  t.n_leaves = n;
  t.layers.clear();
  t.layers.emplace_back(4 * n);
  par_for(threads, n, [&](size_t lo, size_t hi) {
    std::vector<F> row(width);
    for (size_t j = lo; j < hi; j++) { for (int k = 0; k < width; k++) row[k] = mat[(size_t)k * n + j]; hash_elems(row.data(), width, &t.layers[0][4 * j]); }
  });
  while (t.layers.back().size() > 4) {
    const std::vector<F>& prev = t.layers.back();
    const size_t m = prev.size() / 8;
    std::vector<F> cur(4 * m);
    par_for(m >= 1024 ? threads : 1, m, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) compress(&prev[8 * i], &prev[8 * i + 4], &cur[4 * i]); });
    t.layers.push_back(std::move(cur));
  }
}
// coefficients of the degree < n polynomial behind column `col` of an LDE matrix [.][2n] on GEN <w_2n>, from its even positions
static void coeffs_from_lde(const F* col, size_t n, std::vector<F>& c) {
  c.resize(n);
  for (size_t i = 0; i < n; i++) c[i] = col[2 * i];
  ntt(c, true);
  const F ginv = finv(GEN); F sc = 1;
  for (size_t k = 0; k < n; k++) { c[k] = fmul(c[k], sc); sc = fmul(sc, ginv); }
}
static void prove_lean(const PackedRow* rows, const Public& pub_in, Proof& proof, int threads) {
  Public pub = pub_in;
  const int log_n = padded_log_n(pub.n_real);
  const size_t N = (size_t)1 << log_n, N2 = 2 * N;
  const int Dm = pub.mode();
  const int Wm = phys_width(Dm), Wl = logical_width(Dm), Wa = aux_width(Dm);
  if (threads < 1) threads = 1;
  std::vector<F> M;
  main_trace(rows, pub.n_real, pub, M, &pub.cells, &pub.hcalls, &pub.wrecs);
  for (int c = 0; c < Wl; c++) if (is_virtual(c, Dm)) std::fill(M.begin() + (size_t)c * N, M.begin() + (size_t)(c + 1) * N, 0);
  for (int i = 0; i < N_STATE; i++) { pub.first[i] = M[(size_t)state_col(i) * N]; pub.last[i] = M[(size_t)state_col(i) * N + (pub.n_real - 1)]; }
  if (Dm >= 2) for (int k = 0; k < 2; k++) { pub.cnt_first[k] = M[(size_t)(C_OC + k) * N]; pub.cnt_last[k] = M[(size_t)(C_OC + k) * N + (pub.n_real - 1)]; }
  std::vector<int> logical_of(Wm, -1);
  for (int c = 0; c < Wl; c++) if (!is_virtual(c, Dm)) logical_of[phys_col(c, Dm)] = c;
  std::vector<F> L((size_t)Wm * N2);
  par_for(threads, (size_t)Wm, [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; k++) {
      const F* col = &M[(size_t)logical_of[k] * N];
      std::vector<F> e(col, col + N), cf, o;
      lde(e, 1, cf, o);
      memcpy(&L[k * N2], o.data(), N2 * 4);
    }
  });
  Merkle trace_tree, aux_tree, quot_tree;
  merkle_build_par(L, Wm, N2, trace_tree, threads);
  std::vector<uint32_t>& w = proof.w;
  header_words(log_n, pub, w);
  Challenger ch;
  ch.observe_n(w.data() + 2, w.size() - 2);
  ch.observe_n(trace_tree.layers.back().data(), 4);
  const Rom rom = rom_from_blob(pub.blob, pub.blob_len, Dm);
  w.push_back((uint32_t)pub.blob_len);
  for (size_t i = 0; i < pub.blob_len; i += 2) w.push_back((uint32_t)pub.blob[i] | (i + 1 < pub.blob_len ? (uint32_t)pub.blob[i + 1] << 8 : 0u));
  if (Dm >= 2) { const size_t at = w.size(); io_section(pub, w); observe_section(ch, w.data() + at, w.size() - at); }
  if (Dm >= 3) { const size_t at = w.size(); mem_section(pub, w); observe_section(ch, w.data() + at, w.size() - at); }
  if (Dm == 4) { const size_t at = w.size(); hash_section(pub, w); observe_section(ch, w.data() + at, w.size() - at); }
  if (Dm == 4) { const size_t at = w.size(); wide_section(pub, w); observe_section(ch, w.data() + at, w.size() - at); }
  std::vector<F> rom_mult, rc_mult, mem_mult;
  lookup_multiplicities(M, N, rom, rom_mult, rc_mult, nullptr, Dm, &mem_mult);
  w.insert(w.end(), rom_mult.begin(), rom_mult.end());
  w.insert(w.end(), rc_mult.begin(), rc_mult.end());
  ch.observe_n(rom_mult.data(), rom_mult.size());
  ch.observe_n(rc_mult.data(), rc_mult.size());
  if (Dm >= 3) { w.insert(w.end(), mem_mult.begin(), mem_mult.end()); ch.observe_n(mem_mult.data(), mem_mult.size()); }
  LookupParams lp;
  lp.alpha = ch.sample_ext();
  {
    const E lambda = ch.sample_ext();
    lp.lam[0] = e_from(1);
    for (int j = 1; j <= N_TUPLE; j++) lp.lam[j] = emul(lp.lam[j - 1], lambda);
  }
  lp.n_in = (F)(pub.n_in % P);
  {
    E T = lookup_table_sum(rom, rom_mult.data(), rc_mult.data(), lp, Dm >= 3 ? mem_mult.data() : nullptr);
    if (Dm >= 2) T = eadd(T, io_table_sum(pub, lp));
    if (Dm >= 3) T = eadd(T, mem_table_sum(pub, lp));
    if (Dm == 4) T = eadd(eadd(T, hash_table_sum(pub, lp)), wide_table_sum(pub, lp));
    lp.t_over_n = emul_f(T, finv((F)(N % P)));
  }
  std::vector<F> AL((size_t)Wa * N2);
  {
    std::vector<F> A;
    aux_trace(M, N, lp, A, Dm);
    M.clear(); M.shrink_to_fit();                                           // the logical matrix has done its work
    par_for(threads, (size_t)Wa, [&](size_t lo, size_t hi) {
      for (size_t k = lo; k < hi; k++) {
        std::vector<F> e(A.begin() + k * N, A.begin() + (k + 1) * N), cf, o;
        lde(e, 1, cf, o);
        memcpy(&AL[k * N2], o.data(), N2 * 4);
      }
    });
  }
  merkle_build_par(AL, Wa, N2, aux_tree, threads);
  ch.observe_n(aux_tree.layers.back().data(), 4);
  const E alpha = ch.sample_ext();
  const int NC = num_constraints(Dm);
  std::vector<E> ap(NC); ap[0] = e_from(1); for (int c = 1; c < NC; c++) ap[c] = emul(ap[c - 1], alpha);

  const F w2n = root_of_unity(log_n + 1), wn = root_of_unity(log_n), wn_inv = finv(wn);
  const F w_last = fpow(wn, pub.n_real - 1);
  const F gN = fpow(GEN, N);
  std::vector<F> Qc(4 * N2);
  par_for(threads, N2, [&](size_t lo, size_t hi) {
    F x = fmul(GEN, fpow(w2n, lo));
    std::vector<E> ploc(Wm), pnxt(Wm), loc(W_MAX), nxt(W_MAX), aloc(W_AUX_MAX), anxt(W_AUX_MAX);
    for (size_t j = lo; j < hi; j++) {
      for (int k = 0; k < Wm; k++) { ploc[k] = e_from(L[(size_t)k * N2 + j]); pnxt[k] = e_from(L[(size_t)k * N2 + ((j + 2) & (N2 - 1))]); }
      to_logical_row(ploc.data(), Dm, e_from(0), loc.data()); to_logical_row(pnxt.data(), Dm, e_from(0), nxt.data());
      for (int k = 0; k < Wa; k++) { aloc[k] = e_from(AL[(size_t)k * N2 + j]); anxt[k] = e_from(AL[(size_t)k * N2 + ((j + 2) & (N2 - 1))]); }
      const F zh = fsub((j & 1) ? fneg(gN) : gN, 1);
      const F inv_zh = finv(zh);
      const E is_first = e_from(fmul(zh, finv(fsub(x, 1))));
      const E is_last = e_from(fmul(zh, finv(fsub(x, w_last))));
      const E is_trans = e_from(fsub(x, wn_inv));
      E sum; constraints_sum(loc.data(), nxt.data(), aloc.data(), anxt.data(), is_first, is_last, is_trans, pub, lp, ap.data(), sum);
      const E q = emul_f(sum, inv_zh);
      for (int i = 0; i < 4; i++) Qc[(size_t)i * N2 + j] = q.c[i];
      x = fmul(x, w2n);
    }
  });
  merkle_build_par(Qc, 4, N2, quot_tree, threads);
  ch.observe_n(quot_tree.layers.back().data(), 4);
  const E zeta = ch.sample_ext();
  const E zeta_w = emul_f(zeta, wn);

  const int Wt = Wm + Wa;
  std::vector<E> t_z(Wt), t_zw(Wt), q_z(4);
  par_for(threads, (size_t)Wt, [&](size_t lo, size_t hi) {
    std::vector<F> cf;
    for (size_t k = lo; k < hi; k++) {
      coeffs_from_lde(k < (size_t)Wm ? &L[k * N2] : &AL[(k - Wm) * N2], N, cf);
      t_z[k] = horner_base(cf, zeta); t_zw[k] = horner_base(cf, zeta_w);
    }
  });
  par_for(std::min(threads, 4), 4, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      std::vector<F> ev(Qc.begin() + i * N2, Qc.begin() + (i + 1) * N2);
      ntt(ev, true);
      F ginv = finv(GEN), sc = 1;
      for (size_t k2 = 0; k2 < N2; k2++) { ev[k2] = fmul(ev[k2], sc); sc = fmul(sc, ginv); }
      q_z[i] = horner_base(ev, zeta);
    }
  });
  for (int k = 0; k < Wt; k++) ch.observe_ext(t_z[k]);
  for (int k = 0; k < Wt; k++) ch.observe_ext(t_zw[k]);
  for (int i = 0; i < 4; i++) ch.observe_ext(q_z[i]);
  const E gamma = ch.sample_ext();

  std::vector<E> gp(2 * Wt + 4); gp[0] = e_from(1); for (size_t k = 1; k < gp.size(); k++) gp[k] = emul(gp[k - 1], gamma);
  E a0 = e_from(0), b0 = e_from(0);
  for (int k = 0; k < Wt; k++) { a0 = eadd(a0, emul(gp[k], t_z[k])); b0 = eadd(b0, emul(gp[Wt + k], t_zw[k])); }
  for (int i = 0; i < 4; i++) a0 = eadd(a0, emul(gp[2 * Wt + i], q_z[i]));
  std::vector<std::vector<E>> fri; std::vector<Merkle> fri_trees;
  fri.emplace_back(N2);
  par_for(threads, N2, [&](size_t lo, size_t hi) {
    std::vector<E>& cw = fri[0];
    F x = fmul(GEN, fpow(w2n, lo));
    for (size_t j = lo; j < hi; j++) {
      E A = e_from(0), B = e_from(0);
      for (int k = 0; k < Wm; k++) { const F v = L[(size_t)k * N2 + j]; A = eadd(A, emul_f(gp[k], v)); B = eadd(B, emul_f(gp[Wt + k], v)); }
      for (int k = 0; k < Wa; k++) { const F v = AL[(size_t)k * N2 + j]; A = eadd(A, emul_f(gp[Wm + k], v)); B = eadd(B, emul_f(gp[Wt + Wm + k], v)); }
      for (int i = 0; i < 4; i++) A = eadd(A, emul_f(gp[2 * Wt + i], Qc[(size_t)i * N2 + j]));
      const E d1 = einv(esub(e_from(x), zeta)), d2 = einv(esub(e_from(x), zeta_w));
      cw[j] = eadd(emul(esub(A, a0), d1), emul(esub(B, b0), d2));
      x = fmul(x, w2n);
    }
  });

  const std::vector<int> ks = fri_schedule(log_n);
  F shift = GEN; int log_m = log_n + 1;
  for (const int k : ks) {
    const std::vector<E>& c = fri.back();
    const size_t m = c.size(), g = m >> k, nv = (size_t)1 << k;
    std::vector<F> mat(4 * nv * g);
    par_for(g >= 4096 ? threads : 1, g, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) for (size_t t = 0; t < nv; t++) for (int e = 0; e < 4; e++) mat[(t * 4 + e) * g + i] = c[i + t * g].c[e]; });
    Merkle tr; merkle_build_par(mat, (int)(4 * nv), g, tr, g >= 4096 ? threads : 1);
    mat.clear(); mat.shrink_to_fit();
    ch.observe_n(tr.layers.back().data(), 4);
    E beta = ch.sample_ext();
    std::vector<E> cur = c;
    for (int f = 0; f < k; f++) {
      const size_t h = cur.size() / 2;
      std::vector<E> nx(h);
      const F wm = root_of_unity(log_m);
      par_for(h >= 4096 ? threads : 1, h, [&](size_t lo, size_t hi) { F x = fmul(shift, fpow(wm, lo)); for (size_t i = lo; i < hi; i++) { nx[i] = fold_pair(cur[i], cur[i + h], x, beta); x = fmul(x, wm); } });
      cur.swap(nx);
      shift = fmul(shift, shift); log_m--; beta = emul(beta, beta);
    }
    fri_trees.push_back(std::move(tr));
    fri.push_back(std::move(cur));
  }
  const std::vector<E>& fin = fri.back();
  for (const E& e : fin) ch.observe_ext(e);
  const F pow_nonce = ch.grind(pub.pow_bits());
  (void)ch.check_pow(pow_nonce, pub.pow_bits());
  std::vector<uint32_t> queries;
  for (int t = 0; t < pub.num_queries(); t++) queries.push_back(ch.sample_bits(log_n));

  for (int i = 0; i < 4; i++) w.push_back(trace_tree.layers.back()[i]);
  for (int i = 0; i < 4; i++) w.push_back(aux_tree.layers.back()[i]);
  for (int i = 0; i < 4; i++) w.push_back(quot_tree.layers.back()[i]);
  for (int k = 0; k < Wt; k++) put_e(w, t_z[k]);
  for (int k = 0; k < Wt; k++) put_e(w, t_zw[k]);
  for (int i = 0; i < 4; i++) put_e(w, q_z[i]);
  w.push_back((uint32_t)fri_trees.size());
  for (auto& tr : fri_trees) for (int i = 0; i < 4; i++) w.push_back(tr.layers.back()[i]);
  for (const E& e : fin) put_e(w, e);
  w.push_back(pow_nonce);
  for (uint32_t q : queries) {
    w.push_back(q);
    for (size_t pos : {(size_t)q, (size_t)q + N}) { for (int k = 0; k < Wm; k++) w.push_back(L[(size_t)k * N2 + pos]); merkle_path(trace_tree, pos, w); }
    for (size_t pos : {(size_t)q, (size_t)q + N}) { for (int k = 0; k < Wa; k++) w.push_back(AL[(size_t)k * N2 + pos]); merkle_path(aux_tree, pos, w); }
    for (size_t pos : {(size_t)q, (size_t)q + N}) { for (int i = 0; i < 4; i++) w.push_back(Qc[(size_t)i * N2 + pos]); merkle_path(quot_tree, pos, w); }
    for (size_t j = 0; j < fri_trees.size(); j++) {
      const size_t g = fri[j].size() >> ks[j], idx = q & (g - 1);
      for (size_t t = 0; t < ((size_t)1 << ks[j]); t++) put_e(w, fri[j][idx + t * g]);
      merkle_path(fri_trees[j], idx, w);
    }
  }
}

// ---- verifier (N4): returns 0 if the proof is accepted, otherwise a non-zero code naming the failed check.  `expect` (nullable):
// the public inputs the caller expects (program / io digests, row count, mode, entry point); a mismatch with the header is code 6 ----
static bool check_path(const F* leaf_digest, size_t idx, const uint32_t* path, int depth, const F* root) {
  F node[4]; memcpy(node, leaf_digest, 16);
  for (int d = 0; d < depth; d++) { F nx[4]; if (idx & 1) compress(path + 4 * d, node, nx); else compress(node, path + 4 * d, nx); memcpy(node, nx, 16); idx >>= 1; }
  return !memcmp(node, root, 16);
}
// whole_run: the proof must start in the VM's initial state (check 7); otherwise it is a segment and `states_out` (nullable, 2 x 68
// words: first, last) is what verify_chain() links.
// (mode 2) the I/O section of a proof -> the tapes and the halt reason; false = malformed
struct IoSection { std::vector<uint64_t> in, out; uint32_t halt_kind = 2; uint64_t halt_code = 0; size_t words = 0; };
static bool parse_io_section(const uint32_t* w, size_t avail, IoSection& io) {
  size_t p = 0;
  auto u64 = [&](uint64_t& v) { if (p + 4 > avail) return false; v = 0; for (int i = 0; i < 4; i++) { if (w[p + i] > 0xFFFF) return false; v |= (uint64_t)w[p + i] << (16 * i); } p += 4; return true; };
  for (int tape = 0; tape < 2; tape++) {
    if (p >= avail) return false;
    const size_t n = w[p++];
    if (n > ((size_t)1 << 28) || p + 4 * n > avail) return false;
    std::vector<uint64_t>& t = tape ? io.out : io.in;
    t.resize(n);
    for (size_t k = 0; k < n; k++) if (!u64(t[k])) return false;
  }
  if (p >= avail) return false;
  io.halt_kind = w[p++];
  if (io.halt_kind > 2 || !u64(io.halt_code)) return false;
  io.words = p;
  return true;
}
static int halt_binding(const uint32_t* w, const F* last, int halt_kind, uint64_t halt_code);
static bool last_row_writes(const uint32_t* w, const F* last, int halt_kind, uint64_t* value);
static int verify(const uint32_t* w, size_t len, const Public* expect, bool whole_run = true, F* states_out = nullptr, F* counters_out = nullptr) {
  size_t p = 0;
  auto need = [&](size_t k) { return p + k <= len; };
  if (!need(HEADER_WORDS) || w[0] != PROOF_MAGIC || w[9] > 4 || w[1] != proof_version((int)w[9])) return 1;
  const int log_n = w[2], Wm = w[3], nq = w[4], log_final = w[5], pow_bits = (int)w[6];
  // the prover's parameters: what `expect` names (0 = the defaults) when there is one; otherwise anything from the defaults up (never fewer queries / bits than those)
  if (expect ? (nq != expect->num_queries() || pow_bits != expect->pow_bits()) : (nq < NUM_QUERIES || pow_bits < POW_BITS)) return 2;
  if (nq > MAX_QUERIES || pow_bits > MAX_POW_BITS || log_final != LOG_FINAL || log_n < LOG_FINAL || log_n > 26) return 2;
  Public pub;
  pub.fri = (uint32_t)nq | ((uint32_t)pow_bits << 16);
  if (w[9] > 4 || Wm != phys_width((int)w[9])) return 2;                 // the committed width is the mode's (0 default, 1 deferred, 2 default + I/O)
  const int mode = (int)w[9];
  const int HW = header_words_of(mode), Wa = aux_width(mode);
  if (!need(HW)) return 1;
  if (w[7] >= (1u << 30) || w[10] >= (1u << 20) || w[11] >= (1u << 20) || w[12] >= (1u << 24)) return 2;
  pub.n_real = (uint64_t)w[7] | ((uint64_t)w[8] << 30); pub.deferred = w[9];
  pub.entry = (uint64_t)w[10] | ((uint64_t)w[11] << 20) | ((uint64_t)w[12] << 40);
  memcpy(pub.prog, w + 13, 16); memcpy(pub.io, w + 17, 16);
  memcpy(pub.first, w + 21, N_STATE * 4); memcpy(pub.last, w + 21 + N_STATE, N_STATE * 4);
  for (int i = 0; i < 2 * N_STATE; i++) if (w[21 + i] >= P) return 3;
  if (pub.n_real == 0 || padded_log_n(pub.n_real) != log_n) return 2;
  if (mode >= 3 && !whole_run) return 2;                                  // the memory check spans the whole run: a mode-3 proof is never a segment
  if (mode >= 2) { for (int k = 0; k < 4; k++) if (w[HEADER_WORDS + k] >= P) return 3; memcpy(pub.cnt_first, w + HEADER_WORDS, 8); memcpy(pub.cnt_last, w + HEADER_WORDS + 2, 8); }
  if (counters_out) { memcpy(counters_out, pub.cnt_first, 8); memcpy(counters_out + 2, pub.cnt_last, 8); }
  if (expect && (expect->n_real != pub.n_real || expect->deferred != pub.deferred || expect->entry != pub.entry ||
                 memcmp(expect->prog, pub.prog, 16) || memcmp(expect->io, pub.io, 16))) return 6;
  if (whole_run) { F init[N_STATE]; initial_state(pub.entry, init); if (memcmp(init, pub.first, sizeof init)) return 7; }
  if (states_out) { memcpy(states_out, pub.first, N_STATE * 4); memcpy(states_out + N_STATE, pub.last, N_STATE * 4); }
  p = HW;
  const size_t N = (size_t)1 << log_n;
  for (size_t i = 2; i < len; i++) if (w[i] >= P) return 3;   // every payload word must be canonical (query indices are < N < p)
  // the program: [byte length][16-bit halfwords]; its digest must be the header's, its entry point the header's (check 8)
  if (!need(1)) return 4;
  const size_t blob_len = w[p++];
  if (blob_len > ((size_t)1 << 30) || !need((blob_len + 1) / 2)) return 4;
  std::vector<uint8_t> blob(blob_len);
  for (size_t i = 0; i < blob_len; i += 2) {
    const uint32_t h = w[p + i / 2];
    if (h > 0xFFFF || (i + 1 >= blob_len && h > 0xFF)) return 8;
    blob[i] = (uint8_t)(h & 0xFF); if (i + 1 < blob_len) blob[i + 1] = (uint8_t)(h >> 8);
  }
  p += (blob_len + 1) / 2;
  { F dg[DIGEST]; digest_bytes(blob.data(), blob_len, dg); if (memcmp(dg, pub.prog, 16)) return 8; }
  const Rom rom = rom_from_blob(blob.data(), blob_len, mode);
  if (!rom.ok || rom.entry != pub.entry) return 8;
  // (mode 2) the tapes and the halt reason the io digest is a digest of (check 50: with the cycle count — a whole run's is its row count; a chain checks the
  // digest over the total), the counters' ends (51), and the halt row named by the halt reason (52 / 53)
  IoSection io;
  const uint32_t* io_words = w + p;
  if (mode >= 2) {
    if (!parse_io_section(w + p, len - p, io)) return 4;
    p += io.words;
    pub.inputs = io.in.data(); pub.n_in = io.in.size(); pub.outputs = io.out.data(); pub.n_out = io.out.size(); pub.halt_kind = io.halt_kind; pub.halt_code = io.halt_code;
    if (pub.cnt_first[0] > pub.cnt_last[0] || pub.cnt_first[1] > pub.cnt_last[1] || pub.cnt_last[0] > pub.n_out || pub.cnt_last[1] > pub.n_in) return 51;
    if (whole_run) {
      std::vector<uint64_t> b;
      b.push_back(pub.n_in); b.insert(b.end(), io.in.begin(), io.in.end()); b.push_back(pub.n_out); b.insert(b.end(), io.out.begin(), io.out.end());
      b.push_back(pub.halt_kind); b.push_back(pub.halt_kind == 1 ? pub.halt_code : 0); b.push_back(pub.n_real);
      F dg[DIGEST]; digest_bytes((const uint8_t*)b.data(), b.size() * 8, dg);
      if (memcmp(dg, pub.io, 16)) return 50;
      // every output was written, none before the run began.  A run that stops at its CYCLE LIMIT has executed its last row too (vm.rs:211-214, :302-347: cycles == rows), and the
      // AIR's counters describe what happened BEFORE a row: if that row is a WRITE ecall, the last output is the R11 of the public last state and is not counted yet
      uint64_t tail_value = 0;
      const bool tail = last_row_writes(w, pub.last, (int)pub.halt_kind, &tail_value);
      if (pub.cnt_first[0] || pub.cnt_first[1] || pub.cnt_last[0] + (tail ? 1 : 0) != pub.n_out) return 51;
      if (tail && pub.outputs[pub.n_out - 1] != tail_value) return 51;
      const int hb = halt_binding(w, pub.last, (int)pub.halt_kind, pub.halt_code);
      if (hb) return hb;
    }
  }
  // (mode 3) the touched cells: canonical (20-bit limbs, a multiple of 8), strictly increasing — every cell has ONE initial tuple (check 54)
  const uint32_t* mem_words = nullptr; size_t mem_len = 0;
  if (mode >= 3) {
    if (!need(1)) return 4;
    const size_t nc = w[p];
    if (nc > ((size_t)1 << 28) || !need(1 + 7 * nc)) return 4;
    mem_words = w + p; mem_len = 1 + 7 * nc;
    pub.blob = blob.data(); pub.blob_len = blob_len;
    pub.cells.resize(nc);
    for (size_t k = 0; k < nc; k++) {
      const uint32_t* c = w + p + 1 + 7 * k;
      if (c[0] >= (1u << 20) || (c[0] & 7) || c[1] >= (1u << 20)) return 54;
      uint64_t bytes = 0;
      for (int i = 0; i < 4; i++) { if (c[3 + i] > 0xFFFF) return 54; bytes |= (uint64_t)c[3 + i] << (16 * i); }
      pub.cells[k] = Public::Cell{(uint64_t)c[0] | ((uint64_t)c[1] << 20), bytes, c[2]};
      if (k && pub.cells[k].addr <= pub.cells[k - 1].addr) return 54;
      // (v11) no access to the CODE: instruction fetch is tied to the program's words (the ROM), so a store into [0x1000, 0x1000 + code_size) would change what the VM
      // executes next (vm.rs:175: strict protection is off) but not what the AIR lets through — a mode-3 proof is of a run whose loads and stores stay off the cells
      // that overlap the code segment, and every accessed cell is in this list (the memory check does not balance otherwise)
      // (mode 4) .. except the BOUNDARY cell (code_size % 8 == 4: the last code word and the first four data bytes): the AIR states there that no store writes its low half
      if (pub.cells[k].addr + 8 > 0x1000 && pub.cells[k].addr < 0x1000 + 4 * (uint64_t)rom.n && !(mode == 4 && (rom.n & 1) && pub.cells[k].addr == boundary_cell(blob.data(), blob_len))) return 55;
    }
    p += mem_len;
  }
  // (mode 4) the hash calls: records in increasing cycle order, ranges in the clear, the previous access of every touched cell before the call (the Blum condition: check 56);
  // the output never lands on code bytes (55)
  const uint32_t* hash_words = nullptr; size_t hash_len = 0;
  if (mode == 4) {
    if (!need(1)) return 4;
    const size_t nh = w[p];
    if (nh > pub.n_real) return 56;
    hash_words = w + p;
    size_t q = p + 1;
    pub.hcalls.reserve(std::min(nh, (len - p) / 28));                                         // (a record is at least 28 words: a forged count allocates nothing beyond the proof)
    std::vector<uint64_t> addrs;
    const uint64_t code_end = 0x1000 + 4 * (uint64_t)rom.n;
    for (size_t k = 0; k < nh; k++) {
      if (q + 8 > len) return 4;
      const uint32_t* c = w + q;
      pub.hcalls.emplace_back();
      Public::HashCall& hc = pub.hcalls[k];
      if (c[1] >= (1u << 20) || c[2] >= (1u << 20) || c[4] >= (1u << 20) || c[5] >= (1u << 20)) return 56;
      hc.cycle = c[0]; hc.in_ptr = (uint64_t)c[1] | ((uint64_t)c[2] << 20); hc.len = c[3]; hc.out_ptr = (uint64_t)c[4] | ((uint64_t)c[5] << 20); hc.kind = c[6];
      if (hc.cycle >= pub.n_real || (k && hc.cycle <= pub.hcalls[k - 1].cycle) || !hash_call_in_range(hc.in_ptr, hc.len, hc.out_ptr, hc.kind)) return 56;
      if (hc.out_ptr < code_end && hc.out_ptr + 32 > 0x1000) return 55;
      hash_call_cells(hc.in_ptr, hc.len, hc.out_ptr, addrs);
      if (c[7] != addrs.size()) return 56;
      if (q + 8 + 5 * addrs.size() > len) return 4;                                            // (truncated: like every other section that ends early)
      q += 8;
      hc.cells.resize(addrs.size());
      for (size_t j = 0; j < addrs.size(); j++, q += 5) {
        uint64_t bytes = 0;
        for (int i = 0; i < 4; i++) { if (w[q + 1 + i] > 0xFFFF) return 56; bytes |= (uint64_t)w[q + 1 + i] << (16 * i); }
        if (w[q] > hc.cycle) return 56;                                                     // the time read (previous access: its cycle + 1) is smaller than the time written (cycle + 1)
        hc.cells[j] = Public::Cell{addrs[j], bytes, w[q]};
      }
    }
    hash_len = q - p;
    p = q;
  }
  // (mode 4 d) the wide tape: records in increasing cycle order, limbs in range, an opcode 3..7, no zero divisor (check 57)
  const uint32_t* wide_words = nullptr; size_t wide_len = 0;
  if (mode == 4) {
    if (!need(1)) return 4;
    const size_t nw = w[p];
    if (nw > pub.n_real) return 57;
    if (!need(1 + 8 * nw)) return 4;
    wide_words = w + p; wide_len = 1 + 8 * nw;
    pub.wrecs.resize(nw);
    for (size_t k = 0; k < nw; k++) {
      const uint32_t* c = w + p + 1 + 8 * k;
      if (c[1] >= (1u << 20) || c[2] >= (1u << 20) || c[3] >= (1u << 24) || c[4] >= (1u << 20) || c[5] >= (1u << 20) || c[6] >= (1u << 24) || c[7] < 3 || c[7] > 7) return 57;
      Public::WideRec& r = pub.wrecs[k];
      r.cycle = c[0]; r.a = (uint64_t)c[1] | ((uint64_t)c[2] << 20) | ((uint64_t)c[3] << 40); r.b = (uint64_t)c[4] | ((uint64_t)c[5] << 20) | ((uint64_t)c[6] << 40); r.op = c[7];
      if (r.cycle >= pub.n_real || (k && r.cycle <= pub.wrecs[k - 1].cycle) || (r.op >= 4 && r.b == 0)) return 57;
    }
    p += wide_len;
  }
  if (!need(rom.n + RC_TABLE + (mode >= 3 ? MEM_MULT : 0))) return 4;
  const F* rom_mult = w + p; p += rom.n;
  const F* rc_mult = w + p; p += RC_TABLE;
  const F* mem_mult = nullptr;
  if (mode >= 3) { mem_mult = w + p; p += MEM_MULT; }
  if (!need(12)) return 4;
  const F* troot = w + p; p += 4; const F* aroot = w + p; p += 4; const F* qroot = w + p; p += 4;
  auto get_e = [&](size_t at) { E e; memcpy(e.c, w + at, 16); return e; };
  const int Wt = Wm + Wa;
  if (!need((size_t)(2 * Wt + 4) * 4)) return 4;
  std::vector<E> t_z(Wt), t_zw(Wt), q_z(4);
  for (int k = 0; k < Wt; k++) { t_z[k] = get_e(p); p += 4; }
  for (int k = 0; k < Wt; k++) { t_zw[k] = get_e(p); p += 4; }
  for (int i = 0; i < 4; i++) { q_z[i] = get_e(p); p += 4; }
  if (!need(1)) return 4;
  const int n_layers = w[p++];
  const std::vector<int> ks = fri_schedule(log_n);
  if (n_layers != (int)ks.size()) return 5;
  if (!need((size_t)4 * n_layers + 4 * ((size_t)1 << LOG_FINAL) + 1)) return 4;
  std::vector<const F*> lroots(n_layers);
  for (int j = 0; j < n_layers; j++) { lroots[j] = w + p; p += 4; }
  std::vector<E> fin((size_t)1 << LOG_FINAL);
  for (auto& e : fin) { e = get_e(p); p += 4; }
  const F pow_nonce = w[p++];
  // transcript
  Challenger ch;
  ch.observe_n(w + 2, HW - 2);
  ch.observe_n(troot, 4);
  if (mode >= 2) observe_section(ch, io_words, io.words);
  if (mode >= 3) observe_section(ch, mem_words, mem_len);
  if (mode == 4) { observe_section(ch, hash_words, hash_len); observe_section(ch, wide_words, wide_len); }
  ch.observe_n(rom_mult, rom.n);
  ch.observe_n(rc_mult, RC_TABLE);
  if (mode >= 3) ch.observe_n(mem_mult, MEM_MULT);
  LookupParams lp;
  lp.alpha = ch.sample_ext();
  {
    const E lambda = ch.sample_ext();
    lp.lam[0] = e_from(1);
    for (int j = 1; j <= N_TUPLE; j++) lp.lam[j] = emul(lp.lam[j - 1], lambda);
  }
  lp.n_in = (F)(pub.n_in % P);
  {
    E T = lookup_table_sum(rom, rom_mult, rc_mult, lp, mem_mult);           // the table side of the lookup identity, computed HERE
    if (mode >= 2) T = eadd(T, io_table_sum(pub, lp));                     // .. its I/O share from the tapes the proof carries
    if (mode >= 3) T = eadd(T, mem_table_sum(pub, lp));                    // .. and both ends of the memory check from the touched cells it carries
    if (mode == 4) T = eadd(eadd(T, hash_table_sum(pub, lp)), wide_table_sum(pub, lp));   // .. (mode 4) and the hash calls: every digest is computed HERE — and the wide tape: every result likewise
    lp.t_over_n = emul_f(T, finv((F)(N % P)));
  }
  ch.observe_n(aroot, 4);
  const E alpha = ch.sample_ext();
  ch.observe_n(qroot, 4);
  const E zeta = ch.sample_ext();
  for (int k = 0; k < Wt; k++) ch.observe_ext(t_z[k]);
  for (int k = 0; k < Wt; k++) ch.observe_ext(t_zw[k]);
  for (int i = 0; i < 4; i++) ch.observe_ext(q_z[i]);
  const E gamma = ch.sample_ext();
  std::vector<E> betas(n_layers);
  for (int j = 0; j < n_layers; j++) { ch.observe_n(lroots[j], 4); betas[j] = ch.sample_ext(); }
  for (const E& e : fin) ch.observe_ext(e);
  if (!ch.check_pow(pow_nonce, pow_bits)) return 12;
  // 1. constraints at zeta:  Σ alpha^c C_c(zeta) == Q(zeta) * Z_H(zeta)
  {
    const int NC = num_constraints(mode);
    std::vector<E> ap(NC); ap[0] = e_from(1); for (int c = 1; c < NC; c++) ap[c] = emul(ap[c - 1], alpha);
    const F wn = root_of_unity(log_n);
    const E zN = epow(zeta, N), zh = esub(zN, e_from(1));
    const E is_first = emul(zh, einv(esub(zeta, e_from(1))));
    const E is_last = emul(zh, einv(esub(zeta, e_from(fpow(wn, pub.n_real - 1)))));
    const E is_trans = esub(zeta, e_from(finv(wn)));
    std::vector<E> loc(W_MAX), nxt(W_MAX);                                // the logical rows at zeta and zeta w: uncommitted columns are the constant 0
    to_logical_row(t_z.data(), mode, e_from(0), loc.data()); to_logical_row(t_zw.data(), mode, e_from(0), nxt.data());
    E lhs; constraints_sum(loc.data(), nxt.data(), t_z.data() + Wm, t_zw.data() + Wm, is_first, is_last, is_trans, pub, lp, ap.data(), lhs);
    E qz = e_from(0);
    for (int i = 0; i < 4; i++) { E basis = e_from(0); basis.c[i] = 1; qz = eadd(qz, emul(basis, q_z[i])); }
    if (!eeq(lhs, emul(qz, zh))) return 10;
  }
  // 2. final codeword is low degree: interpolate over shift_f * <w_8>, top half of the coefficients must vanish
  {
    const size_t m = fin.size();
    for (int t = 0; t < 4; t++) {
      std::vector<F> ev(m); for (size_t i = 0; i < m; i++) ev[i] = fin[i].c[t];
      ntt(ev, true);                                                       // coefficients of f(shift * x): scaling by shift^-k keeps zeros zero
      for (size_t k2 = m / 2; k2 < m; k2++) if (ev[k2] != 0) return 11;
    }
  }
  // 3. queries
  std::vector<E> gp(2 * Wt + 4); gp[0] = e_from(1); for (size_t k = 1; k < gp.size(); k++) gp[k] = emul(gp[k - 1], gamma);
  E a0 = e_from(0), b0 = e_from(0);
  for (int k = 0; k < Wt; k++) { a0 = eadd(a0, emul(gp[k], t_z[k])); b0 = eadd(b0, emul(gp[Wt + k], t_zw[k])); }
  for (int i = 0; i < 4; i++) a0 = eadd(a0, emul(gp[2 * Wt + i], q_z[i]));
  const E zeta_w = emul_f(zeta, root_of_unity(log_n));
  const F w2n = root_of_unity(log_n + 1);
  const int depth0 = log_n + 1;
  for (int t = 0; t < nq; t++) {
    const uint32_t q = ch.sample_bits(log_n);
    if (!need(1) || w[p++] != q) return 20;
    E deep[2];
    const uint32_t* tl[2]; const uint32_t* al[2]; const uint32_t* ql[2];
    for (int s2 = 0; s2 < 2; s2++) {
      const size_t pos = (size_t)q + (s2 ? N : 0);
      if (!need((size_t)Wm + 4 * depth0)) return 4;
      tl[s2] = w + p; p += Wm;
      F dg[4]; hash_elems(tl[s2], Wm, dg);
      if (!check_path(dg, pos, w + p, depth0, troot)) return 21;
      p += 4 * depth0;
    }
    for (int s2 = 0; s2 < 2; s2++) {
      const size_t pos = (size_t)q + (s2 ? N : 0);
      if (!need((size_t)Wa + 4 * depth0)) return 4;
      al[s2] = w + p; p += Wa;
      F dg[4]; hash_elems(al[s2], Wa, dg);
      if (!check_path(dg, pos, w + p, depth0, aroot)) return 27;
      p += 4 * depth0;
    }
    for (int s2 = 0; s2 < 2; s2++) {
      const size_t pos = (size_t)q + (s2 ? N : 0);
      if (!need((size_t)4 + 4 * depth0)) return 4;
      ql[s2] = w + p; p += 4;
      F dg[4]; hash_elems(ql[s2], 4, dg);
      if (!check_path(dg, pos, w + p, depth0, qroot)) return 22;
      p += 4 * depth0;
    }
    for (int s2 = 0; s2 < 2; s2++) {
      const size_t pos = (size_t)q + (s2 ? N : 0);
      const F x = fmul(GEN, fpow(w2n, pos));
      E A = e_from(0), B = e_from(0);
      for (int k = 0; k < Wm; k++) { A = eadd(A, emul_f(gp[k], tl[s2][k])); B = eadd(B, emul_f(gp[Wt + k], tl[s2][k])); }
      for (int k = 0; k < Wa; k++) { A = eadd(A, emul_f(gp[Wm + k], al[s2][k])); B = eadd(B, emul_f(gp[Wt + Wm + k], al[s2][k])); }
      for (int i = 0; i < 4; i++) A = eadd(A, emul_f(gp[2 * Wt + i], ql[s2][i]));
      deep[s2] = eadd(emul(esub(A, a0), einv(esub(e_from(x), zeta))), emul(esub(B, b0), einv(esub(e_from(x), zeta_w))));
    }
    // FRI layers: the leaf of layer j holds the 2^k values that the next k binary folds combine into one
    E carried = e_from(0); size_t carried_idx = q;                         // value / position carried into the current layer
    F shift = GEN; int log_m = log_n + 1;
    for (int j = 0; j < n_layers; j++) {
      const int k = ks[j], depth = log_m - k;
      const size_t nv = (size_t)1 << k, g = (size_t)1 << depth, idx = carried_idx & (g - 1), slot = carried_idx >> depth;
      if (!need(4 * nv + 4 * (size_t)depth)) return 4;
      std::vector<E> v(nv);
      for (size_t t2 = 0; t2 < nv; t2++) v[t2] = get_e(p + 4 * t2);
      F dg[4]; hash_elems(w + p, 4 * nv, dg);
      p += 4 * nv;
      if (!check_path(dg, idx, w + p, depth, lroots[j])) return 23;
      p += 4 * (size_t)depth;
      if (j == 0) { if (!eeq(v[0], deep[0]) || !eeq(v[1], deep[1])) return 24; }   // both elements of the first leaf are known from the openings
      else if (!eeq(v[slot], carried)) return 25;
      E beta = betas[j];
      for (int f = 0; f < k; f++) {
        const size_t half = nv >> (f + 1);
        const F wm = root_of_unity(log_m);
        for (size_t t2 = 0; t2 < half; t2++) v[t2] = fold_pair(v[t2], v[t2 + half], fmul(shift, fpow(wm, idx + t2 * g)), beta);
        shift = fmul(shift, shift); log_m--; beta = emul(beta, beta);
      }
      carried = v[0]; carried_idx = idx;                                   // position in the next layer (size g)
    }
    if (!eeq(fin[carried_idx & (fin.size() - 1)], carried)) return 26;
  }
  if (p != len) return 30;
  return 0;
}

// ---- a run proven in SEGMENTS (row ranges that overlap by one row: the last row of segment i is row 0 of segment i + 1, labelled
// "halt" in the former — its instruction is executed by the latter).  Every proof must verify as a segment; the first starts in the
// VM's initial state; each later one starts in exactly the state its predecessor ended in (the cycle counter is part of the state);
// program digest, io digest, mode and entry point are the same everywhere; `expect` (nullable) carries the run's public inputs, its
// n_real being the run's TOTAL executed rows = sum(n_i - 1) + 1.  0 = accepted; 40-44 = chain checks; 1000 (i + 1) + c = check c of
// segment i failed.
static int verify_chain(const uint32_t* const* proofs, const size_t* lens, int n, const Public* expect) {
  if (n < 1) return 40;
  if (n == 1 && lens[0] > 9 && proofs[0][9] >= 3) return verify(proofs[0], lens[0], expect, true);   // a mode-3 proof is a whole run by itself (never a segment): a "chain" of one
  std::vector<F> st((size_t)n * 2 * N_STATE), cnt((size_t)n * 4, 0);
  uint64_t total = 1;
  for (int i = 0; i < n; i++) {
    const int rc = verify(proofs[i], lens[i], nullptr, false, &st[(size_t)i * 2 * N_STATE], &cnt[(size_t)i * 4]);
    if (rc) return 1000 * (i + 1) + rc;
    const uint32_t* w = proofs[i];
    total += ((uint64_t)w[7] | ((uint64_t)w[8] << 30)) - 1;
    if (memcmp(w + 9, proofs[0] + 9, 12 * 4)) return 43;                 // mode, entry, program digest, io digest
  }
  const uint32_t* w0 = proofs[0];
  const uint64_t entry = (uint64_t)w0[10] | ((uint64_t)w0[11] << 20) | ((uint64_t)w0[12] << 40);
  F init[N_STATE]; initial_state(entry, init);
  if (memcmp(init, &st[0], sizeof init)) return 41;
  for (int i = 1; i < n; i++)
    if (memcmp(&st[(size_t)i * 2 * N_STATE], &st[(size_t)(i - 1) * 2 * N_STATE + N_STATE], N_STATE * 4)) return 42;
  if (expect) {
    if (expect->deferred != w0[9] || expect->entry != entry || memcmp(expect->prog, w0 + 13, 16) || memcmp(expect->io, w0 + 17, 16)) return 43;
    for (int i = 0; i < n; i++) if (proofs[i][4] != (uint32_t)expect->num_queries() || proofs[i][6] != (uint32_t)expect->pow_bits()) return 43;   // every segment made with the parameters the caller expects
    if (expect->n_real != total) return 44;
  }
  if (w0[9] == 2) {
    // (mode 2) the I/O argument across segments: every segment carries the same tapes, the counters start at (0, 0), link up and end with every output written;
    // the tapes + halt reason + TOTAL cycle count hash to the io digest; the last segment's last row is the instruction the halt reason names
    const int HW = header_words_of(2);
    auto io_at = [&](const uint32_t* w) { return HW + 1 + (size_t)(w[HW] + 1) / 2; };
    IoSection io0;
    if (!parse_io_section(w0 + io_at(w0), lens[0] - io_at(w0), io0)) return 4;
    for (int i = 1; i < n; i++) { const uint32_t* w = proofs[i]; if (io_at(w) != io_at(w0) || io_at(w) + io0.words > lens[i] || memcmp(w + io_at(w), w0 + io_at(w0), io0.words * 4)) return 45; }
    if (cnt[0] || cnt[1]) return 51;
    for (int i = 1; i < n; i++) if (cnt[(size_t)i * 4] != cnt[(size_t)(i - 1) * 4 + 2] || cnt[(size_t)i * 4 + 1] != cnt[(size_t)(i - 1) * 4 + 3]) return 46;
    {
      uint64_t tail_value = 0;
      const bool tail = last_row_writes(proofs[n - 1], proofs[n - 1] + 21 + N_STATE, (int)io0.halt_kind, &tail_value);       // (see verify(): the last row of a run cut by its cycle limit)
      if (cnt[(size_t)(n - 1) * 4 + 2] + (tail ? 1 : 0) != io0.out.size() || (tail && io0.out.back() != tail_value)) return 51;
    }
    std::vector<uint64_t> b;
    b.push_back(io0.in.size()); b.insert(b.end(), io0.in.begin(), io0.in.end()); b.push_back(io0.out.size()); b.insert(b.end(), io0.out.begin(), io0.out.end());
    b.push_back(io0.halt_kind); b.push_back(io0.halt_kind == 1 ? io0.halt_code : 0); b.push_back(total);
    F dg[DIGEST]; digest_bytes((const uint8_t*)b.data(), b.size() * 8, dg);
    if (memcmp(dg, w0 + 17, 16)) return 50;
    const int hb = halt_binding(proofs[n - 1], proofs[n - 1] + 21 + N_STATE, (int)io0.halt_kind, io0.halt_code);
    if (hb) return hb;
  }
  return 0;
}
// ---- the claim in the clear (round 4): I/O tapes + halt reason hash to the io digest, and the halt row is the instruction the halt reason names ----
// (vm.rs:302-347: Ebreak halts on EBREAK; syscall.rs:101-107: Exit(code) halts on ECALL with R10 = 0, code = R11; vm.rs:211-214: CycleLimit names no instruction)
static int halt_binding(const uint32_t* w, const F* last, int halt_kind, uint64_t halt_code) {
  if (halt_kind == 2) return 0;
  if (halt_kind != 0 && halt_kind != 1) return 52;
  const int HW = header_words_of((int)w[9]);
  const size_t blob_len = w[HW];
  std::vector<uint8_t> blob(blob_len);
  for (size_t i = 0; i < blob_len; i += 2) { const uint32_t h = w[HW + 1 + i / 2]; blob[i] = (uint8_t)(h & 0xFF); if (i + 1 < blob_len) blob[i + 1] = (uint8_t)(h >> 8); }
  if (blob_len < 32) return 52;
  const uint32_t code_size = (uint32_t)blob[16] | ((uint32_t)blob[17] << 8) | ((uint32_t)blob[18] << 16) | ((uint32_t)blob[19] << 24);
  const uint64_t pc = (uint64_t)last[1] | ((uint64_t)last[2] << 20) | ((uint64_t)last[3] << 40);
  if (pc < 0x1000 || pc % 4 || pc - 0x1000 >= code_size || 32 + (pc - 0x1000) + 4 > blob_len) return 52;
  const size_t at = 32 + (size_t)(pc - 0x1000);
  const uint32_t word = (uint32_t)blob[at] | ((uint32_t)blob[at + 1] << 8) | ((uint32_t)blob[at + 2] << 16) | ((uint32_t)blob[at + 3] << 24);
  if ((word & 0x7F) != (halt_kind == 0 ? 0x51u : 0x50u)) return 52;
  if (halt_kind == 1) {
    uint64_t r[2];
    for (int k = 0; k < 2; k++) {
      const F* l = last + 4 + 3 * (10 + k); const int bits = last[52 + 10 + k] ? 30 : 20;
      r[k] = (uint64_t)l[0] | ((uint64_t)l[1] << bits) | ((uint64_t)l[2] << (2 * bits));
    }
    if (r[0] != 0 || r[1] != halt_code) return 53;
  }
  return 0;
}
// the last executed row of a run that stopped at its cycle limit, if it is a WRITE ecall (R10 = 2): the value it writes (R11 of the public last state)
static bool last_row_writes(const uint32_t* w, const F* last, int halt_kind, uint64_t* value) {
  if (halt_kind != 2) return false;
  const int HW = header_words_of((int)w[9]);
  const size_t blob_len = w[HW];
  if (blob_len < 32) return false;
  auto byte = [&](size_t i) { const uint32_t h = w[HW + 1 + i / 2]; return (uint32_t)((i & 1) ? (h >> 8) : (h & 0xFF)); };
  const uint32_t code_size = byte(16) | (byte(17) << 8) | (byte(18) << 16) | (byte(19) << 24);
  const uint64_t pc = (uint64_t)last[1] | ((uint64_t)last[2] << 20) | ((uint64_t)last[3] << 40);
  if (pc < 0x1000 || pc % 4 || pc - 0x1000 >= code_size || 32 + (pc - 0x1000) + 4 > blob_len) return false;
  const size_t at = 32 + (size_t)(pc - 0x1000);
  if ((byte(at) & 0x7F) != 0x50) return false;                                                 // opcode.rs:154-228: ECALL
  uint64_t r[2];
  for (int k = 0; k < 2; k++) { const F* l = last + 4 + 3 * (10 + k); const int bits = last[52 + 10 + k] ? 30 : 20; r[k] = (uint64_t)l[0] | ((uint64_t)l[1] << bits) | ((uint64_t)l[2] << (2 * bits)); }
  if (r[0] != 2) return false;
  *value = r[1];
  return true;
}
static bool io_claim_matches(const uint32_t* w, uint64_t cycles, const uint64_t* in, size_t n_in, const uint64_t* out, size_t n_out, int halt_kind, uint64_t halt_code) {
  std::vector<uint64_t> io;
  io.push_back(n_in); for (size_t i = 0; i < n_in; i++) io.push_back(in[i]);
  io.push_back(n_out); for (size_t i = 0; i < n_out; i++) io.push_back(out[i]);
  io.push_back((uint64_t)halt_kind); io.push_back(halt_kind == 1 ? halt_code : 0); io.push_back(cycles);
  F dg[DIGEST];
  digest_bytes((const uint8_t*)io.data(), io.size() * 8, dg);
  return !memcmp(dg, w + 17, 16);
}
static int verify_io(const uint32_t* w, size_t len, const Public* expect, const uint64_t* in, size_t n_in, const uint64_t* out, size_t n_out, int halt_kind, uint64_t halt_code) {
  F st[2 * N_STATE];
  const int rc = verify(w, len, expect, true, st);
  if (rc) return rc;
  if (!io_claim_matches(w, (uint64_t)w[7] | ((uint64_t)w[8] << 30), in, n_in, out, n_out, halt_kind, halt_code)) return 50;
  return halt_binding(w, st + N_STATE, halt_kind, halt_code);
}
static int verify_chain_io(const uint32_t* const* proofs, const size_t* lens, int n, const Public* expect, const uint64_t* in, size_t n_in, const uint64_t* out, size_t n_out,
                           int halt_kind, uint64_t halt_code) {
  const int rc = verify_chain(proofs, lens, n, expect);
  if (rc) return rc;
  uint64_t total = 1;
  for (int i = 0; i < n; i++) total += ((uint64_t)proofs[i][7] | ((uint64_t)proofs[i][8] << 30)) - 1;
  if (!io_claim_matches(proofs[0], total, in, n_in, out, n_out, halt_kind, halt_code)) return 50;
  return halt_binding(proofs[n - 1], proofs[n - 1] + 21 + N_STATE, halt_kind, halt_code);
}
}  // namespace so

// =================================================================================================
// C API (ctypes)
// =================================================================================================
extern "C" {
struct so_public { uint64_t n_real; uint32_t deferred; uint32_t fri /* prover parameters: num_queries | pow_bits << 16, 0 = the defaults */; uint64_t entry; uint32_t prog[4]; uint32_t io[4]; const uint8_t* blob; uint64_t blob_len;
                   // mode 2 (deferred == 2): the tapes and the halt reason in the clear; for a segment, the WRITE / READ ecalls executed before its first row
                   const uint64_t* inputs; uint64_t n_inputs; const uint64_t* outputs; uint64_t n_outputs; uint32_t halt_kind; uint32_t pad2; uint64_t halt_code;
                   uint64_t writes_before, reads_before; };
static so::Public to_pub(const so_public* p) {
  so::Public q; q.n_real = p->n_real; q.deferred = p->deferred; q.fri = p->fri; q.entry = p->entry; memcpy(q.prog, p->prog, 16); memcpy(q.io, p->io, 16);
  q.blob = p->blob; q.blob_len = (size_t)p->blob_len;
  q.inputs = p->inputs; q.n_in = (size_t)p->n_inputs; q.outputs = p->outputs; q.n_out = (size_t)p->n_outputs; q.halt_kind = p->halt_kind; q.halt_code = p->halt_code;
  q.writes_before = p->writes_before; q.reads_before = p->reads_before;
  return q;
}
uint32_t so_p() { return so::P; }
uint32_t so_fmul(uint32_t a, uint32_t b) { return so::fmul(a, b); }
uint32_t so_finv(uint32_t a) { return so::finv(a); }
uint32_t so_root_of_unity(int log_n) { return so::root_of_unity(log_n); }
void so_emul(const uint32_t* a, const uint32_t* b, uint32_t* out) { so::E x, y; memcpy(x.c, a, 16); memcpy(y.c, b, 16); so::E z = so::emul(x, y); memcpy(out, z.c, 16); }
void so_einv(const uint32_t* a, uint32_t* out) { so::E x; memcpy(x.c, a, 16); so::E z = so::einv(x); memcpy(out, z.c, 16); }
void so_poseidon2_permute(uint32_t* state12) { so::permute(state12); }
void so_poseidon2_constants(uint32_t* ext96, uint32_t* in22, uint32_t* diag12) {
  const so::P2Consts& c = so::p2consts();
  memcpy(ext96, c.ext, sizeof c.ext); memcpy(in22, c.in, sizeof c.in); memcpy(diag12, c.diag, sizeof c.diag);
}
void so_hash_elems(const uint32_t* in, size_t n, uint32_t* out4) { so::hash_elems(in, n, out4); }
void so_compress(const uint32_t* l, const uint32_t* r, uint32_t* out4) { so::compress(l, r, out4); }
void so_digest_bytes(const uint8_t* b, size_t n, uint32_t* out4) { so::digest_bytes(b, n, out4); }
void so_ntt(uint32_t* a, size_t n, int inverse) { std::vector<so::F> v(a, a + n); so::ntt(v, inverse != 0); memcpy(a, v.data(), n * 4); }
// evals[n] -> coeffs[n], lde[n << log_blowup]
void so_lde(const uint32_t* evals, size_t n, int log_blowup, uint32_t* coeffs, uint32_t* out) {
  std::vector<so::F> e(evals, evals + n), c, o;
  so::lde(e, log_blowup, c, o);
  if (coeffs) memcpy(coeffs, c.data(), n * 4);
  memcpy(out, o.data(), o.size() * 4);
}
int so_main_trace_width() { return so::W_MAIN; }                                   // LOGICAL columns (what so_main_trace writes and the constraints read)
int so_committed_width(int mode) { return so::phys_width(mode); }                  // columns of the committed matrix: 152 (mode 0: default) / 168 (1: deferred) / 160 (2: default + I/O)
int so_logical_width(int mode) { return so::logical_width(mode); }                 // 172 / 172 / 180
int so_aux_width_for(int mode) { return so::aux_width(mode); }                     // 40 / 40 / 48
int so_num_constraints_for(int mode) { return so::num_constraints(mode); }         // 398 / 398 / 430
// logical [so_logical_width][n] -> committed [so_committed_width][n]
void so_to_committed(const uint32_t* logical, size_t n, int mode, uint32_t* out) {
  std::vector<so::F> M(logical, logical + (size_t)so::logical_width(mode) * n), o;
  so::to_physical(M, n, mode, o);
  memcpy(out, o.data(), o.size() * 4);
}
int so_padded_log_n(uint64_t n_real) { return so::padded_log_n(n_real); }
int so_num_constraints() { return so::num_constraints(); }
void so_main_trace(const void* packed_rows, const so_public* pub, uint32_t* out /* [W_MAIN][N] */) {
  std::vector<so::F> m; so::main_trace((const so::PackedRow*)packed_rows, pub->n_real, to_pub(pub), m); memcpy(out, m.data(), m.size() * 4);
}
// lookup parameters as 52 words: alpha (4), lambda^0..lambda^10 (44), T / N (4)
static void lk_pack(const so::LookupParams& lp, uint32_t* out) {
  memcpy(out, lp.alpha.c, 16);
  for (int j = 0; j <= so::N_TUPLE; j++) memcpy(out + 4 + 4 * j, lp.lam[j].c, 16);
  memcpy(out + 4 + 4 * (so::N_TUPLE + 1), lp.t_over_n.c, 16);
}
static so::LookupParams lk_unpack(const uint32_t* in) {
  so::LookupParams lp;
  memcpy(lp.alpha.c, in, 16);
  for (int j = 0; j <= so::N_TUPLE; j++) memcpy(lp.lam[j].c, in + 4 + 4 * j, 16);
  memcpy(lp.t_over_n.c, in + 4 + 4 * (so::N_TUPLE + 1), 16);
  return lp;
}
int so_aux_width() { return so::W_AUX; }
int so_rc_table() { return so::RC_TABLE; }
// The lookup side of a main-trace matrix [W_MAIN][N] for GIVEN challenges (tests): multiplicities of both tables, T / N, the aux
// trace [W_AUX][N].  Returns the number of ROM rows (code words of pub->blob).  rom_mult (nullable) must hold that many words.
size_t so_lookup_setup(const uint32_t* matrix, const so_public* pub, const uint32_t* alpha4, const uint32_t* lambda4, uint32_t* aux_out, uint32_t* lk52, uint32_t* rom_mult,
                       uint32_t* rc_mult) {
  const so::Public q = to_pub(pub);
  const size_t N = (size_t)1 << so::padded_log_n(q.n_real);
  const so::Rom rom = so::rom_from_blob(q.blob, q.blob_len);
  std::vector<so::F> M(matrix, matrix + (size_t)so::W_MAIN * N), rm, cm, A;
  so::lookup_multiplicities(M, N, rom, rm, cm);
  so::LookupParams lp;
  memcpy(lp.alpha.c, alpha4, 16);
  so::E lambda; memcpy(lambda.c, lambda4, 16);
  lp.lam[0] = so::e_from(1);
  for (int j = 1; j <= so::N_TUPLE; j++) lp.lam[j] = so::emul(lp.lam[j - 1], lambda);
  lp.t_over_n = so::emul_f(so::lookup_table_sum(rom, rm.data(), cm.data(), lp), so::finv((so::F)(N % so::P)));
  so::aux_trace(M, N, lp, A);
  if (aux_out) memcpy(aux_out, A.data(), A.size() * 4);
  if (lk52) lk_pack(lp, lk52);
  if (rom_mult && rom.n) memcpy(rom_mult, rm.data(), rom.n * 4);
  if (rc_mult) memcpy(rc_mult, cm.data(), so::RC_TABLE * 4);
  return rom.n;
}
// Σ alpha^c C_c of one (local, next) row pair with base-field values and the given selector values (tests: which constraint fails)
int so_constraints_eval(const uint32_t* loc, const uint32_t* nxt, const uint32_t* aloc, const uint32_t* anxt, const uint32_t* lk52, uint32_t is_first, uint32_t is_last,
                        uint32_t is_trans, const so_public* pub, const uint32_t* alpha4, uint32_t* out4) {
  const int NC = so::num_constraints();
  so::E a; memcpy(a.c, alpha4, 16);
  std::vector<so::E> ap(NC); ap[0] = so::e_from(1); for (int c = 1; c < NC; c++) ap[c] = so::emul(ap[c - 1], a);
  std::vector<so::E> l(so::W_MAIN), x(so::W_MAIN), al(so::W_AUX), ax(so::W_AUX);
  for (int k = 0; k < so::W_MAIN; k++) { l[k] = so::e_from(loc[k]); x[k] = so::e_from(nxt[k]); }
  for (int k = 0; k < so::W_AUX; k++) { al[k] = so::e_from(aloc[k]); ax[k] = so::e_from(anxt[k]); }
  so::E r;
  so::Public q = to_pub(pub);
  so::initial_state(q.entry, q.first);                                                    // a whole run's first state;
  if (is_last) for (int i = 0; i < so::N_STATE; i++) q.last[i] = loc[so::state_col(i)];   // the last state is whatever the last row holds
  so::constraints_sum(l.data(), x.data(), al.data(), ax.data(), so::e_from(is_first), so::e_from(is_last), so::e_from(is_trans), q, lk_unpack(lk52), ap.data(), r);
  memcpy(out4, r.c, 16);
  return NC;
}
// the same with explicit boundary states (68 words each): segments
int so_constraints_eval_states(const uint32_t* loc, const uint32_t* nxt, const uint32_t* aloc, const uint32_t* anxt, const uint32_t* lk52, uint32_t is_first, uint32_t is_last,
                               uint32_t is_trans, const so_public* pub, const uint32_t* first68, const uint32_t* last68, const uint32_t* alpha4, uint32_t* out4) {
  const int NC = so::num_constraints();
  so::E a; memcpy(a.c, alpha4, 16);
  std::vector<so::E> ap(NC); ap[0] = so::e_from(1); for (int c = 1; c < NC; c++) ap[c] = so::emul(ap[c - 1], a);
  std::vector<so::E> l(so::W_MAIN), x(so::W_MAIN), al(so::W_AUX), ax(so::W_AUX);
  for (int k = 0; k < so::W_MAIN; k++) { l[k] = so::e_from(loc[k]); x[k] = so::e_from(nxt[k]); }
  for (int k = 0; k < so::W_AUX; k++) { al[k] = so::e_from(aloc[k]); ax[k] = so::e_from(anxt[k]); }
  so::Public q = to_pub(pub);
  memcpy(q.first, first68, sizeof q.first); memcpy(q.last, last68, sizeof q.last);
  so::E r;
  so::constraints_sum(l.data(), x.data(), al.data(), ax.data(), so::e_from(is_first), so::e_from(is_last), so::e_from(is_trans), q, lk_unpack(lk52), ap.data(), r);
  memcpy(out4, r.c, 16);
  return NC;
}
// the same in MODE 2 / 3 (pub->deferred; anything else reads as 2): 180 / 220 logical columns, 48 / 96 aux columns, lk = the 56 words + n_in, cnt4 = (oc, ic) of the first row, of the last row
int so_constraints_eval_io(const uint32_t* loc, const uint32_t* nxt, const uint32_t* aloc, const uint32_t* anxt, const uint32_t* lk57, uint32_t is_first, uint32_t is_last,
                           uint32_t is_trans, const so_public* pub, const uint32_t* first68, const uint32_t* last68, const uint32_t* cnt4, const uint32_t* alpha4, uint32_t* out4) {
  so::Public q = to_pub(pub);
  if (q.deferred < 3) q.deferred = 2;
  const int mode = q.mode(), NC = so::num_constraints(mode), Wl = so::logical_width(mode), Wa = so::aux_width(mode);
  so::E a; memcpy(a.c, alpha4, 16);
  std::vector<so::E> ap(NC); ap[0] = so::e_from(1); for (int c = 1; c < NC; c++) ap[c] = so::emul(ap[c - 1], a);
  std::vector<so::E> l(so::W_MAX, so::e_from(0)), x(so::W_MAX, so::e_from(0)), al(so::W_AUX_MAX, so::e_from(0)), ax(so::W_AUX_MAX, so::e_from(0));
  for (int k = 0; k < Wl; k++) { l[k] = so::e_from(loc[k]); x[k] = so::e_from(nxt[k]); }
  for (int k = 0; k < Wa; k++) { al[k] = so::e_from(aloc[k]); ax[k] = so::e_from(anxt[k]); }
  memcpy(q.first, first68, sizeof q.first); memcpy(q.last, last68, sizeof q.last);
  memcpy(q.cnt_first, cnt4, 8); memcpy(q.cnt_last, cnt4 + 2, 8);
  so::LookupParams lp = lk_unpack(lk57);
  lp.n_in = lk57[56];
  so::E r;
  so::constraints_sum(l.data(), x.data(), al.data(), ax.data(), so::e_from(is_first), so::e_from(is_last), so::e_from(is_trans), q, lp, ap.data(), r);
  memcpy(out4, r.c, 16);
  return NC;
}
// Merkle root (and optionally every layer, concatenated leaf-layer first) of a column-major matrix [width][n]
void so_merkle(const uint32_t* mat, int width, size_t n, uint32_t* root4, uint32_t* all_layers /* nullable, 4*(2n-1) */) {
  std::vector<so::F> m(mat, mat + (size_t)width * n);
  so::Merkle t; so::merkle_build(m, width, n, t);
  memcpy(root4, t.layers.back().data(), 16);
  if (all_layers) { size_t off = 0; for (auto& l : t.layers) { memcpy(all_layers + off, l.data(), l.size() * 4); off += l.size(); } }
}
// commit = main_trace -> committed columns -> per-column LDE -> Merkle over the LDE rows; returns root, optionally the LDE matrix [Wm][N<<lb]
void so_commit_trace(const void* packed_rows, const so_public* pub, int log_blowup, uint32_t* root4, uint32_t* lde_out /* nullable */) {
  std::vector<so::F> m; so::main_trace((const so::PackedRow*)packed_rows, pub->n_real, to_pub(pub), m);
  const size_t n = (size_t)1 << so::padded_log_n(pub->n_real);
  size_t big = n << log_blowup;
  const int Wm = so::phys_width((int)pub->deferred);
  std::vector<so::F> mp; so::to_physical(m, n, (int)pub->deferred, mp);
  std::vector<so::F> L((size_t)Wm * big);
  for (int k = 0; k < Wm; k++) {
    std::vector<so::F> e(mp.begin() + (size_t)k * n, mp.begin() + (size_t)(k + 1) * n), c, o;
    so::lde(e, log_blowup, c, o);
    memcpy(&L[(size_t)k * big], o.data(), big * 4);
  }
  so::Merkle t; so::merkle_build(L, Wm, big, t);
  memcpy(root4, t.layers.back().data(), 16);
  if (lde_out) memcpy(lde_out, L.data(), L.size() * 4);
}

// The same commitment, COLUMN-BLOCKED (round 5: the root at BASELINE configs[2]'s own size, 2^24 rows): the committed columns are extended eight at a time (one
// absorption of the rate-8 sponge) and every leaf keeps only its 12-word sponge state, so the 2^25 x 152 LDE matrix (20 GB) never exists: memory = the logical main
// trace + 8 LDE columns + 48 B per leaf.  Same functions as so_commit_trace (so::lde, so::permute, so::compress), same order of operations per leaf:
// hash_elems == zero state, for each chunk of 8 { overwrite s[0..8), permute }.  `threads` std::threads share the columns of a block / the leaves: the arithmetic
// is untouched.  tests/test_stark_oracle.py: equal to so_commit_trace on small sizes, every mode.
void so_commit_trace_blocked(const void* packed_rows, const so_public* pub, int log_blowup, uint32_t* root4, int threads) {
  std::vector<so::F> m; so::main_trace((const so::PackedRow*)packed_rows, pub->n_real, to_pub(pub), m);
  const int mode = (int)pub->deferred, Wm = so::phys_width(mode), Wl = so::logical_width(mode);
  const size_t n = (size_t)1 << so::padded_log_n(pub->n_real), big = n << log_blowup;
  std::vector<int> logical_of(Wm, -1);
  for (int c = 0; c < Wl; c++) if (!so::is_virtual(c, mode)) logical_of[so::phys_col(c, mode)] = c;
  if (threads < 1) threads = 1;
  std::vector<so::F> state((size_t)so::T * big, 0);                          // sponge state of leaf j: state[12 j .. 12 j + 12)
  std::vector<std::vector<so::F>> blk(so::RATE);
  auto par = [&](size_t count, auto&& body) {                                 // body(lo, hi) over [0, count) split among the threads
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) { size_t lo = count * t / threads, hi = count * (t + 1) / threads; if (lo < hi) th.emplace_back([=, &body] { body(lo, hi); }); }
    for (auto& x : th) x.join();
  };
  for (int k0 = 0; k0 < Wm; k0 += so::RATE) {
    const int len = std::min(so::RATE, Wm - k0);
    par((size_t)len, [&](size_t lo, size_t hi) {
      for (size_t q = lo; q < hi; q++) {
        const so::F* col = &m[(size_t)logical_of[k0 + (int)q] * n];
        std::vector<so::F> e(col, col + n), c;
        so::lde(e, log_blowup, c, blk[q]);
      }
    });
    par(big, [&](size_t lo, size_t hi) {
      for (size_t j = lo; j < hi; j++) { so::F* s = &state[(size_t)so::T * j]; for (int q = 0; q < len; q++) s[q] = blk[q][j]; so::permute(s); }
    });
  }
  m.clear(); m.shrink_to_fit();
  for (auto& b : blk) { b.clear(); b.shrink_to_fit(); }
  std::vector<so::F> cur(4 * big);
  for (size_t j = 0; j < big; j++) memcpy(&cur[4 * j], &state[(size_t)so::T * j], 16);
  state.clear(); state.shrink_to_fit();
  while (cur.size() > 4) {
    size_t cnt = cur.size() / 8;
    std::vector<so::F> nxt(4 * cnt);
    par(cnt, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) so::compress(&cur[8 * i], &cur[8 * i + 4], &nxt[4 * i]); });
    cur.swap(nxt);
  }
  memcpy(root4, cur.data(), 16);
}

// ---- stage B C API ----
static so::ProverTrace g_pt;   // last prover run (tests inspect intermediate objects)
size_t so_prove(const void* packed_rows, const so_public* pub, uint32_t* out, size_t cap) {
  so::Proof pr; so::prove((const so::PackedRow*)packed_rows, to_pub(pub), pr, g_pt);
  if (out && pr.w.size() <= cap) memcpy(out, pr.w.data(), pr.w.size() * 4);
  return pr.w.size();
}
// the memory-lean threaded restatement of so_prove (so::prove_lean): same proof words; `out` must hold the proof (cap words; the word count is returned either way)
size_t so_prove_lean(const void* packed_rows, const so_public* pub, uint32_t* out, size_t cap, int threads) {
  so::Proof pr; so::prove_lean((const so::PackedRow*)packed_rows, to_pub(pub), pr, threads);
  if (out && pr.w.size() <= cap) memcpy(out, pr.w.data(), pr.w.size() * 4);
  return pr.w.size();
}
size_t so_prove_matrix(const uint32_t* matrix, const so_public* pub, uint32_t* out, size_t cap) {
  so::Proof pr; so::prove(nullptr, to_pub(pub), pr, g_pt, matrix);
  if (out && pr.w.size() <= cap) memcpy(out, pr.w.data(), pr.w.size() * 4);
  return pr.w.size();
}
// (mode 3) the touched cells of a run, 7 words per cell as in the proof's memory section (address limbs, final time, final bytes as four 16-bit pieces); returns the cell count
size_t so_mem_cells(const void* packed_rows, const so_public* pub, uint32_t* out, size_t cap_cells) {
  std::vector<so::F> m; std::vector<so::Public::Cell> cells;
  so::main_trace((const so::PackedRow*)packed_rows, pub->n_real, to_pub(pub), m, &cells);
  if (out && cells.size() <= cap_cells) for (size_t k = 0; k < cells.size(); k++) {
    uint32_t* c = out + 7 * k;
    c[0] = (uint32_t)(cells[k].addr & 0xFFFFF); c[1] = (uint32_t)((cells[k].addr >> 20) & 0xFFFFF); c[2] = cells[k].t;
    for (int i = 0; i < 4; i++) c[3 + i] = (uint32_t)((cells[k].bytes >> (16 * i)) & 0xFFFF);
  }
  return cells.size();
}
// (mode 4) the hash calls of a run as the proof's hash section (so::hash_section); returns the word count
size_t so_hash_section(const void* packed_rows, const so_public* pub, uint32_t* out, size_t cap) {
  std::vector<so::F> m; so::Public q = to_pub(pub);
  so::main_trace((const so::PackedRow*)packed_rows, pub->n_real, q, m, &q.cells, &q.hcalls);
  std::vector<uint32_t> w; so::hash_section(q, w);
  if (out && w.size() <= cap) memcpy(out, w.data(), w.size() * 4);
  return w.size();
}
// (tests, mode 4) the hash calls so_prove_matrix_mem / so_failing_constraints are to use (a matrix brings no rows to replay): a hash section, possibly forged; n = 0 clears
static std::vector<so::Public::HashCall> g_hcalls;
void so_set_hash_calls(const uint32_t* w, size_t n) {
  g_hcalls.clear();
  if (!w || !n) return;
  size_t q = 1;
  std::vector<uint64_t> addrs;
  for (uint32_t k = 0; k < w[0] && q + 8 <= n; k++) {
    so::Public::HashCall hc{w[q], (uint64_t)w[q + 1] | ((uint64_t)w[q + 2] << 20), w[q + 3], (uint64_t)w[q + 4] | ((uint64_t)w[q + 5] << 20), w[q + 6], {}};
    const size_t nc = w[q + 7];
    so::hash_call_cells(hc.in_ptr, hc.len, hc.out_ptr, addrs);
    q += 8;
    for (size_t j = 0; j < nc && q + 5 <= n; j++, q += 5) {
      uint64_t b = 0; for (int i = 0; i < 4; i++) b |= (uint64_t)w[q + 1 + i] << (16 * i);
      hc.cells.push_back(so::Public::Cell{j < addrs.size() ? addrs[j] : 0, b, w[q]});
    }
    g_hcalls.push_back(std::move(hc));
  }
}
// (mode 3) proof of a GIVEN matrix with a GIVEN list of touched cells (tests: a cheating prover)
size_t so_prove_matrix_mem(const uint32_t* matrix, const so_public* pub, const uint32_t* cells7, size_t n_cells, uint32_t* out, size_t cap) {
  so::Public q = to_pub(pub);
  for (size_t k = 0; k < n_cells; k++) {
    const uint32_t* c = cells7 + 7 * k; uint64_t b = 0;
    for (int i = 0; i < 4; i++) b |= (uint64_t)c[3 + i] << (16 * i);
    q.cells.push_back(so::Public::Cell{(uint64_t)c[0] | ((uint64_t)c[1] << 20), b, c[2]});
  }
  q.hcalls = g_hcalls;
  so::Proof pr; so::prove(nullptr, q, pr, g_pt, matrix);
  if (out && pr.w.size() <= cap) memcpy(out, pr.w.data(), pr.w.size() * 4);
  return pr.w.size();
}
// (tests) which constraints does a main-trace matrix violate ON THE TRACE DOMAIN?  The lookup side is set up honestly for the given challenges (multiplicities, T, aux
// trace; mode 3: with the given cells); out receives up to cap (constraint index, row) pairs; returns the number of violations.  Boundary states = the matrix's own.
size_t so_failing_constraints(const uint32_t* matrix, const so_public* pub, const uint32_t* cells7, size_t n_cells, const uint32_t* alpha4, const uint32_t* lambda4, uint32_t* out, size_t cap) {
  so::Public q = to_pub(pub);
  const int mode = q.mode(), Wl = so::logical_width(mode), Wa = so::aux_width(mode);
  const size_t N = (size_t)1 << so::padded_log_n(q.n_real);
  for (size_t k = 0; k < n_cells; k++) {
    const uint32_t* c = cells7 + 7 * k; uint64_t b = 0;
    for (int i = 0; i < 4; i++) b |= (uint64_t)c[3 + i] << (16 * i);
    q.cells.push_back(so::Public::Cell{(uint64_t)c[0] | ((uint64_t)c[1] << 20), b, c[2]});
  }
  std::vector<so::F> M(matrix, matrix + (size_t)Wl * N), rm, cm, mm, A;
  for (int c = 0; c < Wl; c++) if (so::is_virtual(c, mode)) std::fill(M.begin() + (size_t)c * N, M.begin() + (size_t)(c + 1) * N, 0);
  for (int i = 0; i < so::N_STATE; i++) { q.first[i] = M[(size_t)so::state_col(i) * N]; q.last[i] = M[(size_t)so::state_col(i) * N + (q.n_real - 1)]; }
  if (mode >= 2) for (int k = 0; k < 2; k++) { q.cnt_first[k] = M[(size_t)(so::C_OC + k) * N]; q.cnt_last[k] = M[(size_t)(so::C_OC + k) * N + (q.n_real - 1)]; }
  const so::Rom rom = so::rom_from_blob(q.blob, q.blob_len, mode);
  so::lookup_multiplicities(M, N, rom, rm, cm, nullptr, mode, &mm);
  so::LookupParams lp;
  memcpy(lp.alpha.c, alpha4, 16);
  so::E lambda; memcpy(lambda.c, lambda4, 16);
  lp.lam[0] = so::e_from(1);
  for (int j = 1; j <= so::N_TUPLE; j++) lp.lam[j] = so::emul(lp.lam[j - 1], lambda);
  lp.n_in = (so::F)(q.n_in % so::P);
  so::E T = so::lookup_table_sum(rom, rm.data(), cm.data(), lp, mode >= 3 ? mm.data() : nullptr);
  if (mode >= 2) T = so::eadd(T, so::io_table_sum(q, lp));
  if (mode >= 3) T = so::eadd(T, so::mem_table_sum(q, lp));
  if (mode == 4) {                                                           // (the wide tape: read off the matrix's own ot rows, like so::prove does for an overridden matrix)
    q.hcalls = g_hcalls; T = so::eadd(T, so::hash_table_sum(q, lp));
    auto m = [&](int c, size_t i) { return (uint64_t)M[(size_t)c * N + i]; };
    for (size_t i = 0; i < q.n_real; i++) if (m(so::C_OT, i))
      q.wrecs.push_back(so::Public::WideRec{m(so::C_CYCLE, i), m(so::C_XB, i) | (m(so::C_XB + 1, i) << 20) | (m(so::C_XB + 2, i) << 40), m(so::C_XC, i) | (m(so::C_XC + 1, i) << 20) | (m(so::C_XC + 2, i) << 40), (uint32_t)m(so::C_OP, i)});
    T = so::eadd(T, so::wide_table_sum(q, lp));
  }
  lp.t_over_n = so::emul_f(T, so::finv((so::F)(N % so::P)));
  so::aux_trace(M, N, lp, A, mode);
  const int NC = so::num_constraints(mode);
  std::vector<so::E> ap(NC, so::e_from(0)), loc(so::W_MAX), nxt(so::W_MAX), al(so::W_AUX_MAX), ax(so::W_AUX_MAX), rec;
  size_t bad = 0;
  for (size_t i = 0; i < N; i++) {
    const size_t j = (i + 1) % N;
    for (int k = 0; k < so::W_MAX; k++) { loc[k] = so::e_from(k < Wl ? M[(size_t)k * N + i] : 0); nxt[k] = so::e_from(k < Wl ? M[(size_t)k * N + j] : 0); }
    for (int k = 0; k < so::W_AUX_MAX; k++) { al[k] = so::e_from(k < Wa ? A[(size_t)k * N + i] : 0); ax[k] = so::e_from(k < Wa ? A[(size_t)k * N + j] : 0); }
    rec.clear(); so::g_air_record = &rec;
    so::E r;
    so::constraints_sum(loc.data(), nxt.data(), al.data(), ax.data(), so::e_from(i == 0), so::e_from(i == q.n_real - 1), so::e_from(i + 1 < N), q, lp, ap.data(), r);
    so::g_air_record = nullptr;
    for (int c = 0; c < (int)rec.size(); c++) if (!so::eeq(rec[c], so::e_from(0))) { if (out && bad < cap) { out[2 * bad] = (uint32_t)c; out[2 * bad + 1] = (uint32_t)i; } bad++; }
  }
  return bad;
}
int so_verify(const uint32_t* proof, size_t len, const so_public* expect) {
  if (!expect) return so::verify(proof, len, nullptr);
  const so::Public e = to_pub(expect);
  return so::verify(proof, len, &e);
}
// segment verification: states136 (nullable) receives the header's first and last state
int so_verify_segment(const uint32_t* proof, size_t len, const so_public* expect, uint32_t* states136) {
  if (!expect) return so::verify(proof, len, nullptr, false, states136);
  const so::Public e = to_pub(expect);
  return so::verify(proof, len, &e, false, states136);
}
int so_verify_chain(const uint32_t* const* proofs, const size_t* lens, int n, const so_public* expect) {
  if (!expect) return so::verify_chain(proofs, lens, n, nullptr);
  const so::Public e = to_pub(expect);
  return so::verify_chain(proofs, lens, n, &e);
}
int so_verify_io(const uint32_t* proof, size_t len, const so_public* expect, const uint64_t* in, size_t n_in, const uint64_t* out, size_t n_out, int halt_kind, uint64_t halt_code) {
  so::Public q; if (expect) q = to_pub(expect);
  return so::verify_io(proof, len, expect ? &q : nullptr, in, n_in, out, n_out, halt_kind, halt_code);
}
int so_verify_chain_io(const uint32_t* const* proofs, const size_t* lens, int n, const so_public* expect, const uint64_t* in, size_t n_in, const uint64_t* out, size_t n_out,
                       int halt_kind, uint64_t halt_code) {
  so::Public q; if (expect) q = to_pub(expect);
  return so::verify_chain_io(proofs, lens, n, expect ? &q : nullptr, in, n_in, out, n_out, halt_kind, halt_code);
}
int so_state_words() { return so::N_STATE; }
void so_last_challenges(uint32_t* alpha, uint32_t* zeta, uint32_t* gamma) { memcpy(alpha, g_pt.alpha.c, 16); memcpy(zeta, g_pt.zeta.c, 16); memcpy(gamma, g_pt.gamma.c, 16); }
void so_last_quotient(uint32_t* out /* [4][2N] */) { memcpy(out, g_pt.Qc.data(), g_pt.Qc.size() * 4); }
size_t so_last_fri_layer(int j, uint32_t* out /* [m][4] */) { if (j < 0 || (size_t)j >= g_pt.fri.size()) return 0; if (out) memcpy(out, g_pt.fri[j].data(), g_pt.fri[j].size() * 16); return g_pt.fri[j].size(); }
void so_last_lookup(uint32_t* lk52) { lk_pack(g_pt.lp, lk52); }
void so_last_aux(uint32_t* out /* [W_AUX][N] */) { memcpy(out, g_pt.A.data(), g_pt.A.size() * 4); }
int so_num_queries() { return so::NUM_QUERIES; }
int so_log_final() { return so::LOG_FINAL; }
int so_pow_bits() { return so::POW_BITS; }
int so_header_words() { return so::HEADER_WORDS; }
}  // extern "C"
