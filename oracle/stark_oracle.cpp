// stark_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the self-defined prover stages.
//
// PARITY UNPINNED.  The reference (seceq/zkir) contains no prover: no AIR, NTT/LDE, Merkle tree, FRI or
// proof format, and its Plonky3 dependency is commented out with no rev (Cargo.toml:67-69).
// Everything in this file is therefore defined by THIS repository ("ZKIR-STARK v0") following
// BASELINE.json:north_star literally (Baby Bear, radix-2 NTT/LDE, Poseidon2 width 12, FRI), and is pinned only by
// algebraic self-checks (tests/test_stark_oracle.py: inverse-NTT round trips, naive DFT, Merkle path checks, a full
// verifier).  The GPU implementation (zkir_amd/csrc/stark.hip) is checked bit-for-bit against this file.
//
// Arithmetic here is deliberately the naive canonical form (u64 products reduced with %), independent of the
// Montgomery arithmetic used on the device.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

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
// main trace matrix: packed 372-byte reference rows -> W = 89 Baby Bear columns
// ---------------------------------------------------------------------------------------------
static const int W_MAIN = 89;
#pragma pack(push, 1)
struct PackedRow { uint64_t cycle, pc; uint32_t instruction; uint64_t registers[16]; uint32_t bound_bits[16]; uint8_t bound_tag[16]; uint64_t bound_payload[16]; uint8_t reg_state[16]; };
#pragma pack(pop)
// col-major out[W_MAIN][n]; `changed[r]` of row i = 1 iff the triple (value, bound, state) of register r differs... no:
// iff an instruction at row i wrote register r.  The oracle derives it from consecutive rows of the FULL triple, which is
// what a write does unless it rewrites identical contents; the product derives it from its event log.  To keep the two
// definitions identical the flag is defined on contents: changed_r[i] = [triple_r(row i+1) != triple_r(row i)], and 0 for the last row.
static void main_trace(const PackedRow* rows, size_t n, std::vector<F>& out) {
  out.assign((size_t)W_MAIN * n, 0);
  auto col = [&](int k) { return out.data() + (size_t)k * n; };
  for (size_t i = 0; i < n; i++) {
    const PackedRow& r = rows[i];
    col(0)[i] = (F)(r.cycle % P);
    col(1)[i] = (F)(r.pc & 0xFFFFF); col(2)[i] = (F)((r.pc >> 20) & 0xFFFFF); col(3)[i] = (F)(r.pc >> 40);
    uint32_t w = r.instruction;
    col(4)[i] = w & 0x7F; col(5)[i] = (w >> 7) & 0xF; col(6)[i] = (w >> 11) & 0xF; col(7)[i] = (w >> 15) & 0xF; col(8)[i] = w >> 19;
    for (int g = 0; g < 16; g++) {
      uint64_t v = r.registers[g];
      int bits = r.reg_state[g] ? 30 : 20;
      uint64_t mask = (1ull << bits) - 1;
      col(9 + 3 * g)[i] = (F)(v & mask); col(10 + 3 * g)[i] = (F)((v >> bits) & mask); col(11 + 3 * g)[i] = (F)(v >> (2 * bits));
      col(57 + g)[i] = r.reg_state[g];
      bool ch = false;
      if (i + 1 < n) {
        const PackedRow& q = rows[i + 1];
        ch = q.registers[g] != v || q.reg_state[g] != r.reg_state[g] || q.bound_bits[g] != r.bound_bits[g] || q.bound_tag[g] != r.bound_tag[g] ||
             q.bound_payload[g] != r.bound_payload[g];
      }
      col(73 + g)[i] = ch;
    }
  }
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


// =================================================================================================
// Stage B: AIR quotient, DEEP openings, FRI, proof bytes, verifier  (ZKIR-STARK v0)
// =================================================================================================
static const int NUM_QUERIES = 24, LOG_FINAL = 3, N_CONSTRAINTS = 103, LOG_ARITY = 3;
static const uint32_t PROOF_MAGIC = 0x46504B5Au, PROOF_VERSION = 2;   // "ZKPF"; v2: FRI layers are committed every LOG_ARITY folds

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
};

// ---- the AIR: Σ_c alpha^c C_c over one (local, next) row pair; generic over base / extension values ----
// columns: 0 cycle | 1-3 pc limbs | 4-8 instruction fields | 9+3r.. register limbs | 57+r state | 73+r changed
// row values as E (base-field rows are lifted); returns Σ alpha^c C_c
static E constraints_sum(const E* loc, const E* nxt, const E& is_first, const E& is_trans, const E* ap) {
  E acc = e_from(0); int c = 0;
  auto push = [&](const E& v) { acc = eadd(acc, emul(ap[c], v)); c++; };
  const E one = e_from(1);
  push(emul(esub(esub(nxt[0], loc[0]), one), is_trans));                 // c0: cycle' = cycle + 1
  push(emul(loc[0], is_first));                                          // c1: cycle[0] = 0
  for (int r = 0; r < 16; r++) {
    const E st = loc[57 + r], ch = loc[73 + r];
    push(emul(st, esub(st, one)));                                       // state boolean
    push(emul(ch, esub(ch, one)));                                       // changed boolean
    const E keep = esub(one, ch);
    for (int l = 0; l < 3; l++) push(emul(emul(keep, esub(nxt[9 + 3 * r + l], loc[9 + 3 * r + l])), is_trans));   // untouched registers keep their limbs
    push(emul(emul(keep, esub(nxt[57 + r], st)), is_trans));             // ... and their storage state
  }
  push(loc[9]); push(loc[10]); push(loc[11]); push(loc[57]); push(loc[73]);   // R0 is hard-wired zero, Normalized, never written
  return acc;
}

struct Proof { std::vector<uint32_t> w; };
static void put_e(std::vector<uint32_t>& w, const E& e) { for (int i = 0; i < 4; i++) w.push_back(e.c[i]); }

struct Tree { Merkle m; };
static void merkle_path(const Merkle& t, size_t leaf, std::vector<uint32_t>& w) {
  size_t j = leaf;
  for (size_t lv = 0; lv + 1 < t.layers.size(); lv++) { const F* sib = &t.layers[lv][4 * (j ^ 1)]; for (int i = 0; i < 4; i++) w.push_back(sib[i]); j >>= 1; }
}
static E fold_pair(const E& a, const E& b, F x, const E& beta) {          // (a+b)/2 + beta (a-b)/(2x)
  const F half = finv(2), inv2x = finv(fmul(2, x));
  return eadd(emul_f(eadd(a, b), half), emul(beta, emul_f(esub(a, b), inv2x)));
}
static E horner_base(const std::vector<F>& coeffs, const E& z) { E acc = e_from(0); for (size_t k = coeffs.size(); k-- > 0;) acc = eadd(emul(acc, z), e_from(coeffs[k])); return acc; }

struct ProverTrace {     // everything the oracle keeps for inspection by tests
  std::vector<F> M, L, Qc;            // [W][N], [W][2N], [4][2N]
  Merkle trace_tree, quot_tree;
  std::vector<std::vector<E>> fri;    // codewords per layer (layer 0 = DEEP codeword, size 2N)
  std::vector<Merkle> fri_trees;
  E alpha, zeta, gamma; std::vector<E> betas; std::vector<uint32_t> queries;
};

static void prove(const PackedRow* rows, size_t n, Proof& proof, ProverTrace& pt) {
  int log_n = 0; while (((size_t)1 << log_n) < n) log_n++;
  const size_t N = n, N2 = 2 * n;
  const int Wm = W_MAIN;
  main_trace(rows, n, pt.M);
  pt.L.assign((size_t)Wm * N2, 0);
  std::vector<std::vector<F>> coeffs(Wm);
  for (int k = 0; k < Wm; k++) {
    std::vector<F> e(pt.M.begin() + (size_t)k * N, pt.M.begin() + (size_t)(k + 1) * N), o;
    lde(e, 1, coeffs[k], o);
    memcpy(&pt.L[(size_t)k * N2], o.data(), N2 * 4);
  }
  merkle_build(pt.L, Wm, N2, pt.trace_tree);
  Challenger ch;
  ch.observe(log_n); ch.observe(Wm); ch.observe(NUM_QUERIES); ch.observe(LOG_FINAL);
  ch.observe_n(pt.trace_tree.layers.back().data(), 4);
  pt.alpha = ch.sample_ext();
  std::vector<E> ap(N_CONSTRAINTS); ap[0] = e_from(1); for (int c = 1; c < N_CONSTRAINTS; c++) ap[c] = emul(ap[c - 1], pt.alpha);

  // ---- quotient on the LDE coset: x_j = g * w_2N^j, next row = position j + 2 ----
  const F w2n = root_of_unity(log_n + 1), wn_inv = finv(root_of_unity(log_n));
  const F gN = fpow(GEN, N);
  pt.Qc.assign(4 * N2, 0);
  {
    F x = GEN;
    std::vector<E> loc(Wm), nxt(Wm);
    for (size_t j = 0; j < N2; j++) {
      for (int k = 0; k < Wm; k++) { loc[k] = e_from(pt.L[(size_t)k * N2 + j]); nxt[k] = e_from(pt.L[(size_t)k * N2 + ((j + 2) & (N2 - 1))]); }
      const F zh = fsub((j & 1) ? fneg(gN) : gN, 1);                      // x^N - 1, x^N = g^N (-1)^j
      const F inv_zh = finv(zh);
      const E is_first = e_from(fmul(zh, finv(fsub(x, 1))));
      const E is_trans = e_from(fsub(x, wn_inv));
      E q = emul_f(constraints_sum(loc.data(), nxt.data(), is_first, is_trans, ap.data()), inv_zh);
      for (int i = 0; i < 4; i++) pt.Qc[(size_t)i * N2 + j] = q.c[i];
      x = fmul(x, w2n);
    }
  }
  merkle_build(pt.Qc, 4, N2, pt.quot_tree);
  ch.observe_n(pt.quot_tree.layers.back().data(), 4);
  pt.zeta = ch.sample_ext();
  const E zeta_w = emul_f(pt.zeta, root_of_unity(log_n));

  // ---- openings (oracle: Horner on coefficient vectors) ----
  std::vector<E> t_z(Wm), t_zw(Wm), q_z(4);
  for (int k = 0; k < Wm; k++) { t_z[k] = horner_base(coeffs[k], pt.zeta); t_zw[k] = horner_base(coeffs[k], zeta_w); }
  for (int i = 0; i < 4; i++) {                                           // quotient columns: interpolate from the coset evaluations
    std::vector<F> ev(pt.Qc.begin() + (size_t)i * N2, pt.Qc.begin() + (size_t)(i + 1) * N2);
    ntt(ev, true);                                                        // coefficients of q_i(g x)
    F ginv = finv(GEN), sc = 1;
    for (size_t k2 = 0; k2 < N2; k2++) { ev[k2] = fmul(ev[k2], sc); sc = fmul(sc, ginv); }
    q_z[i] = horner_base(ev, pt.zeta);
  }
  for (int k = 0; k < Wm; k++) ch.observe_ext(t_z[k]);
  for (int k = 0; k < Wm; k++) ch.observe_ext(t_zw[k]);
  for (int i = 0; i < 4; i++) ch.observe_ext(q_z[i]);
  pt.gamma = ch.sample_ext();

  // ---- DEEP codeword over the LDE coset ----
  std::vector<E> gp(2 * Wm + 4); gp[0] = e_from(1); for (size_t k = 1; k < gp.size(); k++) gp[k] = emul(gp[k - 1], pt.gamma);
  E a0 = e_from(0), b0 = e_from(0);
  for (int k = 0; k < Wm; k++) { a0 = eadd(a0, emul(gp[k], t_z[k])); b0 = eadd(b0, emul(gp[Wm + k], t_zw[k])); }
  for (int i = 0; i < 4; i++) a0 = eadd(a0, emul(gp[2 * Wm + i], q_z[i]));
  std::vector<E> cw(N2);
  {
    F x = GEN;
    for (size_t j = 0; j < N2; j++) {
      E A = e_from(0), B = e_from(0);
      for (int k = 0; k < Wm; k++) { const F v = pt.L[(size_t)k * N2 + j]; A = eadd(A, emul_f(gp[k], v)); B = eadd(B, emul_f(gp[Wm + k], v)); }
      for (int i = 0; i < 4; i++) A = eadd(A, emul_f(gp[2 * Wm + i], pt.Qc[(size_t)i * N2 + j]));
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
  pt.queries.clear();
  for (int t = 0; t < NUM_QUERIES; t++) pt.queries.push_back(ch.sample_bits(log_n));

  // ---- serialize ----
  std::vector<uint32_t>& w = proof.w; w.clear();
  w.push_back(PROOF_MAGIC); w.push_back(PROOF_VERSION); w.push_back(log_n); w.push_back(Wm); w.push_back(NUM_QUERIES); w.push_back(LOG_FINAL);
  for (int i = 0; i < 4; i++) w.push_back(pt.trace_tree.layers.back()[i]);
  for (int i = 0; i < 4; i++) w.push_back(pt.quot_tree.layers.back()[i]);
  for (int k = 0; k < Wm; k++) put_e(w, t_z[k]);
  for (int k = 0; k < Wm; k++) put_e(w, t_zw[k]);
  for (int i = 0; i < 4; i++) put_e(w, q_z[i]);
  w.push_back((uint32_t)pt.fri_trees.size());
  for (auto& tr : pt.fri_trees) for (int i = 0; i < 4; i++) w.push_back(tr.layers.back()[i]);
  for (const E& e : fin) put_e(w, e);
  for (uint32_t q : pt.queries) {
    w.push_back(q);
    for (size_t pos : {(size_t)q, (size_t)q + N}) { for (int k = 0; k < Wm; k++) w.push_back(pt.L[(size_t)k * N2 + pos]); merkle_path(pt.trace_tree, pos, w); }
    for (size_t pos : {(size_t)q, (size_t)q + N}) { for (int i = 0; i < 4; i++) w.push_back(pt.Qc[(size_t)i * N2 + pos]); merkle_path(pt.quot_tree, pos, w); }
    for (size_t j = 0; j < pt.fri_trees.size(); j++) {
      const size_t g = pt.fri[j].size() >> ks[j], idx = q & (g - 1);
      for (size_t t = 0; t < ((size_t)1 << ks[j]); t++) put_e(w, pt.fri[j][idx + t * g]);
      merkle_path(pt.fri_trees[j], idx, w);
    }
  }
}

// ---- verifier (N4): returns 0 if the proof is accepted, otherwise a non-zero code naming the failed check ----
static bool check_path(const F* leaf_digest, size_t idx, const uint32_t* path, int depth, const F* root) {
  F node[4]; memcpy(node, leaf_digest, 16);
  for (int d = 0; d < depth; d++) { F nx[4]; if (idx & 1) compress(path + 4 * d, node, nx); else compress(node, path + 4 * d, nx); memcpy(node, nx, 16); idx >>= 1; }
  return !memcmp(node, root, 16);
}
static int verify(const uint32_t* w, size_t len) {
  size_t p = 0;
  auto need = [&](size_t k) { return p + k <= len; };
  if (!need(6) || w[0] != PROOF_MAGIC || w[1] != PROOF_VERSION) return 1;
  const int log_n = w[2], Wm = w[3], nq = w[4], log_final = w[5]; p = 6;
  if (Wm != W_MAIN || nq != NUM_QUERIES || log_final != LOG_FINAL || log_n < LOG_FINAL || log_n > 26) return 2;
  const size_t N = (size_t)1 << log_n;
  for (size_t i = 6; i < len; i++) if (w[i] >= P) return 3;   // every payload word must be canonical (query indices are < N < p)
  if (!need(8)) return 4;
  const F* troot = w + p; p += 4; const F* qroot = w + p; p += 4;
  auto get_e = [&](size_t at) { E e; memcpy(e.c, w + at, 16); return e; };
  if (!need((size_t)(2 * Wm + 4) * 4)) return 4;
  std::vector<E> t_z(Wm), t_zw(Wm), q_z(4);
  for (int k = 0; k < Wm; k++) { t_z[k] = get_e(p); p += 4; }
  for (int k = 0; k < Wm; k++) { t_zw[k] = get_e(p); p += 4; }
  for (int i = 0; i < 4; i++) { q_z[i] = get_e(p); p += 4; }
  if (!need(1)) return 4;
  const int n_layers = w[p++];
  const std::vector<int> ks = fri_schedule(log_n);
  if (n_layers != (int)ks.size()) return 5;
  if (!need((size_t)4 * n_layers + 4 * ((size_t)1 << LOG_FINAL))) return 4;
  std::vector<const F*> lroots(n_layers);
  for (int j = 0; j < n_layers; j++) { lroots[j] = w + p; p += 4; }
  std::vector<E> fin((size_t)1 << LOG_FINAL);
  for (auto& e : fin) { e = get_e(p); p += 4; }
  // transcript
  Challenger ch;
  ch.observe(log_n); ch.observe(Wm); ch.observe(NUM_QUERIES); ch.observe(LOG_FINAL);
  ch.observe_n(troot, 4);
  const E alpha = ch.sample_ext();
  ch.observe_n(qroot, 4);
  const E zeta = ch.sample_ext();
  for (int k = 0; k < Wm; k++) ch.observe_ext(t_z[k]);
  for (int k = 0; k < Wm; k++) ch.observe_ext(t_zw[k]);
  for (int i = 0; i < 4; i++) ch.observe_ext(q_z[i]);
  const E gamma = ch.sample_ext();
  std::vector<E> betas(n_layers);
  for (int j = 0; j < n_layers; j++) { ch.observe_n(lroots[j], 4); betas[j] = ch.sample_ext(); }
  for (const E& e : fin) ch.observe_ext(e);
  // 1. constraints at zeta:  Σ alpha^c C_c(zeta) == Q(zeta) * Z_H(zeta)
  {
    std::vector<E> ap(N_CONSTRAINTS); ap[0] = e_from(1); for (int c = 1; c < N_CONSTRAINTS; c++) ap[c] = emul(ap[c - 1], alpha);
    const E zN = epow(zeta, N), zh = esub(zN, e_from(1));
    const E is_first = emul(zh, einv(esub(zeta, e_from(1))));
    const E is_trans = esub(zeta, e_from(finv(root_of_unity(log_n))));
    const E lhs = constraints_sum(t_z.data(), t_zw.data(), is_first, is_trans, ap.data());
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
  std::vector<E> gp(2 * Wm + 4); gp[0] = e_from(1); for (size_t k = 1; k < gp.size(); k++) gp[k] = emul(gp[k - 1], gamma);
  E a0 = e_from(0), b0 = e_from(0);
  for (int k = 0; k < Wm; k++) { a0 = eadd(a0, emul(gp[k], t_z[k])); b0 = eadd(b0, emul(gp[Wm + k], t_zw[k])); }
  for (int i = 0; i < 4; i++) a0 = eadd(a0, emul(gp[2 * Wm + i], q_z[i]));
  const E zeta_w = emul_f(zeta, root_of_unity(log_n));
  const F w2n = root_of_unity(log_n + 1);
  const int depth0 = log_n + 1;
  for (int t = 0; t < nq; t++) {
    const uint32_t q = ch.sample_bits(log_n);
    if (!need(1) || w[p++] != q) return 20;
    E deep[2];
    const uint32_t* tl[2]; const uint32_t* ql[2];
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
      for (int k = 0; k < Wm; k++) { A = eadd(A, emul_f(gp[k], tl[s2][k])); B = eadd(B, emul_f(gp[Wm + k], tl[s2][k])); }
      for (int i = 0; i < 4; i++) A = eadd(A, emul_f(gp[2 * Wm + i], ql[s2][i]));
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
      for (size_t t = 0; t < nv; t++) v[t] = get_e(p + 4 * t);
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
        for (size_t t = 0; t < half; t++) v[t] = fold_pair(v[t], v[t + half], fmul(shift, fpow(wm, idx + t * g)), beta);
        shift = fmul(shift, shift); log_m--; beta = emul(beta, beta);
      }
      carried = v[0]; carried_idx = idx;                                   // position in the next layer (size g)
    }
    if (!eeq(fin[carried_idx & (fin.size() - 1)], carried)) return 26;
  }
  if (p != len) return 30;
  return 0;
}
}  // namespace so

// =================================================================================================
// C API (ctypes)
// =================================================================================================
extern "C" {
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
void so_ntt(uint32_t* a, size_t n, int inverse) { std::vector<so::F> v(a, a + n); so::ntt(v, inverse != 0); memcpy(a, v.data(), n * 4); }
// evals[n] -> coeffs[n], lde[n << log_blowup]
void so_lde(const uint32_t* evals, size_t n, int log_blowup, uint32_t* coeffs, uint32_t* out) {
  std::vector<so::F> e(evals, evals + n), c, o;
  so::lde(e, log_blowup, c, o);
  if (coeffs) memcpy(coeffs, c.data(), n * 4);
  memcpy(out, o.data(), o.size() * 4);
}
int so_main_trace_width() { return so::W_MAIN; }
void so_main_trace(const void* packed_rows, size_t n, uint32_t* out /* [W_MAIN][n] */) {
  std::vector<so::F> m; so::main_trace((const so::PackedRow*)packed_rows, n, m); memcpy(out, m.data(), m.size() * 4);
}
// Merkle root (and optionally every layer, concatenated leaf-layer first) of a column-major matrix [width][n]
void so_merkle(const uint32_t* mat, int width, size_t n, uint32_t* root4, uint32_t* all_layers /* nullable, 4*(2n-1) */) {
  std::vector<so::F> m(mat, mat + (size_t)width * n);
  so::Merkle t; so::merkle_build(m, width, n, t);
  memcpy(root4, t.layers.back().data(), 16);
  if (all_layers) { size_t off = 0; for (auto& l : t.layers) { memcpy(all_layers + off, l.data(), l.size() * 4); off += l.size(); } }
}
// commit = main_trace -> per-column LDE -> Merkle over the LDE rows; returns root, optionally the LDE matrix [W][n<<lb]
void so_commit_trace(const void* packed_rows, size_t n, int log_blowup, uint32_t* root4, uint32_t* lde_out /* nullable */) {
  std::vector<so::F> m; so::main_trace((const so::PackedRow*)packed_rows, n, m);
  size_t big = n << log_blowup;
  std::vector<so::F> L((size_t)so::W_MAIN * big);
  for (int k = 0; k < so::W_MAIN; k++) {
    std::vector<so::F> e(m.begin() + (size_t)k * n, m.begin() + (size_t)(k + 1) * n), c, o;
    so::lde(e, log_blowup, c, o);
    memcpy(&L[(size_t)k * big], o.data(), big * 4);
  }
  so::Merkle t; so::merkle_build(L, so::W_MAIN, big, t);
  memcpy(root4, t.layers.back().data(), 16);
  if (lde_out) memcpy(lde_out, L.data(), L.size() * 4);
}

// ---- stage B C API ----
static so::ProverTrace g_pt;   // last prover run (tests inspect intermediate objects)
size_t so_prove(const void* packed_rows, size_t n, uint32_t* out, size_t cap) {
  so::Proof pr; so::prove((const so::PackedRow*)packed_rows, n, pr, g_pt);
  if (out && pr.w.size() <= cap) memcpy(out, pr.w.data(), pr.w.size() * 4);
  return pr.w.size();
}
int so_verify(const uint32_t* proof, size_t len) { return so::verify(proof, len); }
void so_last_challenges(uint32_t* alpha, uint32_t* zeta, uint32_t* gamma) { memcpy(alpha, g_pt.alpha.c, 16); memcpy(zeta, g_pt.zeta.c, 16); memcpy(gamma, g_pt.gamma.c, 16); }
void so_last_quotient(uint32_t* out /* [4][2N] */) { memcpy(out, g_pt.Qc.data(), g_pt.Qc.size() * 4); }
size_t so_last_fri_layer(int j, uint32_t* out /* [m][4] */) { if (j < 0 || (size_t)j >= g_pt.fri.size()) return 0; if (out) memcpy(out, g_pt.fri[j].data(), g_pt.fri[j].size() * 16); return g_pt.fri[j].size(); }
int so_num_queries() { return so::NUM_QUERIES; }
int so_log_final() { return so::LOG_FINAL; }
}  // extern "C"
