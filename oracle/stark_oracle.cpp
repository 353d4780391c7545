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
}  // extern "C"
