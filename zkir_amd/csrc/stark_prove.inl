// stark_prove.inl — stage B of the self-defined prover (included by stark.hip): AIR quotient on the LDE coset, barycentric
// openings, DEEP combination, FRI commit/fold, query gathering, proof serialisation; host-side Fiat-Shamir transcript.
// Spec: oracle/stark_oracle.cpp (so::prove / so::verify).  Parity unpinned vs the reference (it has no prover).
//
// Every kernel is one thread per evaluation point, column reads coalesced across lanes.  The quotient / DEEP kernels stream the
// LDE matrix once each (HBM-bound: 4*W B read per point, 16 B written); FRI folds are geometric and negligible after layer 0.

namespace {

using bb::E4;

// WMX / WTX: the largest committed width (deferred mode) — array sizes; a proof's own widths are air::committed_width(deferred) and that + WA
constexpr int LOG_FINAL = air::LOG_FINAL, WMX = air::W_COMMITTED_MAX, WAX = air::W_AUX_MAX, WTX = WMX + WAX, LOG_ARITY = air::LOG_ARITY, NS = air::N_STATE, HEADER_WORDS = 21 + 2 * NS;   // (queries / grinding bits: air.h, per proof)
constexpr int N_CONSTRAINTS = air::N_CONSTRAINTS;
constexpr uint32_t PROOF_MAGIC = 0x46504B5Au;                  // (the version is the mode's: air::proof_version)

#ifndef DEEP_WAVES
#define DEEP_WAVES 4
#endif
#ifndef BARY_WAVES
#define BARY_WAVES 4
#endif
#ifndef QUOT_WAVES
#define QUOT_WAVES 3
#endif
#ifndef QUOT_WAVES_MEM
#define QUOT_WAVES_MEM 2            // mode 3 (224 + 96 columns, 559 constraints): at three waves per SIMD the kernel spills 528 B / lane
#endif
struct ProveParams {            // constants of one proof, Montgomery form; lives in the proof's workspace (device), uploaded per phase
  E4 alpha_seq[N_CONSTRAINTS + 2];   // alpha^c in the ORDER the quotient kernel consumes them (air::push_order), Montgomery; two spare entries: the kernel requests one ahead
  E4 gamma_pow[2 * WTX + 4];    // main columns, aux columns at zeta; the same at zeta w; the quotient (2 * (committed width + WA) + 4 used)
  E4 zeta, zeta_w, a0, b0;
  uint32_t first_m[NS], last_m[NS];   // public boundary states: rows 0 and n_real - 1 (Montgomery)
  uint32_t cnt_m[4];            // (mode 2) the counters (oc, ic) of row 0 and of row n_real - 1 (Montgomery)
  E4 cf, cl;                    // air::boundary_constants: the boundary words' share of the is_first / is_last sums
  uint32_t deferred;
  uint32_t lk[air::N_LK];       // lookup parameters (air.h LK_*): alpha, lambda powers, T / N — base-field coordinates, Montgomery
};

// Folds per committed FRI layer (so::fri_schedule): the DEEP codeword is folded once (its leaves are the pairs (q, q + N) the trace
// openings determine), later layers 8-to-1 (three binary folds with beta, beta^2, beta^4), the last one as needed to reach 2^LOG_FINAL.
// A commitment costs ~log2(leaves) SEQUENTIAL Poseidon2 levels (~10 us each once a level is small), so committing every third
// fold cuts the latency-bound part of FRI from ~200 levels to ~80 at 2^20 rows.
std::vector<int> fri_schedule(int log_n) {
  std::vector<int> ks;
  for (int log_m = log_n + 1; log_m > LOG_FINAL;) {
    const int k = ks.empty() ? 1 : (LOG_ARITY < log_m - LOG_FINAL ? LOG_ARITY : log_m - LOG_FINAL);
    ks.push_back(k); log_m -= k;
  }
  return ks;
}

__device__ __forceinline__ E4 e_from_base_m(uint32_t xm) { return E4{{xm, 0, 0, 0}}; }
__device__ __forceinline__ E4 e_fma_base(const E4& acc, const E4& coef, uint32_t vm) {          // acc + coef * v   (all Montgomery)
  return E4{{bb::add(acc.c[0], bb::mont_mul(coef.c[0], vm)), bb::add(acc.c[1], bb::mont_mul(coef.c[1], vm)), bb::add(acc.c[2], bb::mont_mul(coef.c[2], vm)),
             bb::add(acc.c[3], bb::mont_mul(coef.c[3], vm))}};
}

// Lazy accumulation of Σ coef_k * v_k (coef in E, v in F): each product is a Montgomery product WITHOUT its final subtraction
// (3 instructions, below 1.469p) added into a 64-bit sum (1 instruction); one Barrett reduction per coordinate at the end.
// Up to 136 terms fit the 200p bound of bb::reduce_wide<7>.
struct LazyE4 { uint64_t a[4] = {0, 0, 0, 0}; };
__device__ __forceinline__ void lz_fma(LazyE4& acc, const E4& coef, uint32_t v) {
#pragma unroll
  for (int t = 0; t < 4; t++) acc.a[t] = bb::acc_add(acc.a[t], bb::mont_mul_lazy(coef.c[t], v));
}
__device__ __forceinline__ E4 lz_reduce(const LazyE4& acc) {
  return E4{{bb::reduce_wide<7>(acc.a[0]), bb::reduce_wide<7>(acc.a[1]), bb::reduce_wide<7>(acc.a[2]), bb::reduce_wide<7>(acc.a[3])}};
}

// ---- quotient: Q(x_j) = (Σ_c alpha^c C_c(x_j)) / Z_H(x_j) on x_j = g w_2N^j;  next row = position j+2 -------------------
// The constraint list is air::eval (air.h), instantiated here on base-field Montgomery values read straight from the LDE matrices, which
// REST in Montgomery form inside a proof (round 4): no conversion on load.  Round 4 also changed HOW the list is evaluated, not what it says:
//   * the combination Σ alpha^c C_c is four exact 96-bit integer sums (one per extension coordinate) of alpha^c[t] * C_c — a 64-bit multiply-add
//     with carry-out + an add-with-carry per term (bb::mad96_s) — reduced once per point; round 3 formed a lazy Montgomery product per term
//     (3 multiplier-class instructions + the 64-bit add);
//   * one set of sums per row selector (none / is_trans / is_first / is_last): the selector multiplies its sum ONCE per point, and the public
//     boundary words leave the per-point work altogether (air::boundary_constants);
//   * 1 / (x_j - 1) comes from a per-context table (zkir_stark_ctx::d_inv_xm1; 1 / (x_j - w^last) is the same table read at j - 2 last, times
//     w^-last) — round 3 ran two 31-bit exponentiations per point;
//   * dot products (selected operands, extension-field products) accumulate in 96 bits as well; selector moments in 64;
//   * column reads are 16-byte loads of (block, row, half): the compiler merges the reads of neighbouring columns (the B8 layout keeps eight
//     columns of a row in 32 contiguous bytes) — round 3 issued one 4-byte load per column, 64 lanes x 32-byte stride each.
// Src supplies the words of one row pair: the device source reads the B8 matrices, tests/test_abi.py's host source reads plain arrays.
template <int DEF /* the MODE: 0 default, 1 deferred, 2 default + I/O */, class Src>
struct QuotientOps {
  using V = uint32_t;
  using AccP = bb::Acc96;
  using AccL = uint64_t;
  Src src;
  const E4* __restrict__ ap;            // alpha^c (Montgomery) in the order air::eval pushes them (air::push_order): read one after the other
  E4 pre;                               // the NEXT push's coefficient, requested while the current one is accumulated (scalar registers on the device)
  const uint32_t* __restrict__ lk;      // lookup parameters, Montgomery
  bb::Acc96 a0[4], at[4], af[4], al[4]; // Σ alpha^c C_c by selector: none, is_trans, is_first, is_last
  E4 sf, sl, st;                        // the two boundary sums and the transition sum, reduced as soon as they are complete (their accumulators' registers are then free)
  BB_HD void end_boundary() { sf = sum_of(af); sl = sum_of(al); }
  BB_HD void end_trans() { st = sum_of(at); }
  BB_HD void init() {
#pragma unroll
    for (int t = 0; t < 4; t++) a0[t] = at[t] = af[t] = al[t] = bb::acc96_zero();
    pre = *ap++;
  }
  BB_HD V loc(int k) const { return air::is_virtual(k, DEF) ? 0u : src.loc(air::phys_col(k, DEF)); }
  BB_HD V nxt(int k) const { return air::is_virtual(k, DEF) ? 0u : src.nxt(air::phys_col(k, DEF)); }
  BB_HD V loc_r(int k) const { return src.loc_r(air::phys_col(k, DEF)); }
  BB_HD V nxt_r(int k) const { return src.nxt_r(air::phys_col(k, DEF)); }
  BB_HD V aloc(int k) const { return src.aloc(k); }
  BB_HD V anxt(int k) const { return src.anxt(k); }
  BB_HD V par(int i) const { return lk[i]; }
  BB_HD V cst(uint32_t cm) const { return cm; }
  BB_HD V add(V a, V b) const { return bb::add(a, b); }
  BB_HD V sub(V a, V b) const { return bb::sub(a, b); }
  BB_HD V mul(V a, V b) const { return bb::mont_mul(a, b); }
  BB_HD V mulc(V a, uint32_t cm) const { return bb::mont_mul(a, cm); }
  BB_HD V lsub(V a, V b) const { return bb::sub_lazy(a, b); }
  BB_HD V ladd(V a, V b) const { return bb::add_lazy(a, b); }
  BB_HD V lmul(V a, V b) const { return bb::mont_mul_lazy(a, b); }
  BB_HD AccP accp() const { return bb::acc96_zero(); }
  BB_HD void acc_mul(AccP& a, V x, V y) const { bb::mad96(a, x, y); }
  BB_HD V acc_val(const AccP& a) const { return bb::acc96_div_R(a); }
  BB_HD AccL accl() const { return 0; }
  BB_HD void acc_lin(AccL& a, V x, uint32_t k) const { a += (uint64_t)k * x; }
  // Σ k x over reduced x with Σ k < 2^11: (acc >> 32) R + (acc mod 2^32) is below 2^38 and 200 p
  BB_HD V accl_val(const AccL& a) const { return bb::reduce_wide<6>((a >> 32) * bb::R1 + (uint32_t)a); }
  BB_HD void push_to(bb::Acc96* acc, int, V v) {           // (the constraint's index is implicit: the coefficients come in push order)
    const E4 c = pre;
    pre = *ap++;
    bb::mad96x4_s(acc, c.c[0], c.c[1], c.c[2], c.c[3], v);
  }
  BB_HD void push(int idx, V v) { push_to(a0, idx, v); }
  BB_HD void push_t(int idx, V v) { push_to(at, idx, v); }
  BB_HD void push_fc(int idx, V v, uint32_t) { push_to(af, idx, v); }
  BB_HD void push_lc(int idx, V v, uint32_t) { push_to(al, idx, v); }
  BB_HD void push_fc0(int, uint32_t) {}
  BB_HD void push_lc0(int, uint32_t) {}
  // the four sums as extension elements in Montgomery form: alpha^c and C_c both carry the factor R, the reduction divides by R once
  BB_HD static E4 sum_of(const bb::Acc96* a) { return E4{{bb::acc96_div_R(a[0]), bb::acc96_div_R(a[1]), bb::acc96_div_R(a[2]), bb::acc96_div_R(a[3])}}; }
};

// one (row, next row) pair of the LDE matrices, B8 layout, 16-byte reads
struct DeviceRowSrc {
  const uint4* __restrict__ L4; const uint4* __restrict__ A4; uint64_t N2, j, jn;
  static __device__ __forceinline__ uint32_t pick(const uint4& q, int c) { return c == 0 ? q.x : c == 1 ? q.y : c == 2 ? q.z : q.w; }
  __device__ __forceinline__ uint32_t at(const uint4* __restrict__ m, int p, uint64_t row) const { return pick(m[((uint64_t)(p >> 3) * N2 + row) * 2 + ((p >> 2) & 1)], p & 3); }
  __device__ __forceinline__ uint32_t at_r(const uint4* __restrict__ m, int p, uint64_t row) const { return reinterpret_cast<const uint32_t*>(m)[((uint64_t)(p >> 3) * N2 + row) * 8 + (p & 7)]; }
  __device__ __forceinline__ uint32_t loc_r(int p) const { return at_r(L4, p, j); }
  __device__ __forceinline__ uint32_t nxt_r(int p) const { return at_r(L4, p, jn); }
  __device__ __forceinline__ uint32_t loc(int p) const { return at(L4, p, j); }
  __device__ __forceinline__ uint32_t nxt(int p) const { return at(L4, p, jn); }
  __device__ __forceinline__ uint32_t aloc(int p) const { return at(A4, p, j); }
  __device__ __forceinline__ uint32_t anxt(int p) const { return at(A4, p, jn); }
};

template <int DEF /* the mode */>
__global__ __launch_bounds__(NT, DEF >= 3 ? QUOT_WAVES_MEM : QUOT_WAVES) void quotient_kernel(const uint32_t* __restrict__ L, const uint32_t* __restrict__ AL, uint32_t log_n, const uint32_t* __restrict__ tw_fwd,
                                                                  const uint32_t* __restrict__ inv_xm1, const ProveParams* __restrict__ pp, uint32_t wn_inv_m, uint32_t w_last_inv_m,
                                                                  uint32_t last_shift, uint32_t inv_zh_even_m, uint32_t inv_zh_odd_m, uint32_t* __restrict__ Q) {
  const uint32_t N2 = 2u << log_n;
  const uint32_t j = blockIdx.x * NT + threadIdx.x;
  if (j >= N2) return;
  // x = g * w_2N^j (Montgomery): w^j = tw[j] for j < N, -tw[j-N] otherwise
  const uint32_t N = N2 >> 1;
  const uint32_t wj = j < N ? tw_fwd[j] : bb::neg(tw_fwd[j - N]);
  const uint32_t x = bb::mont_mul(wj, bb::to_mont(bb::GEN));
  // Z_H takes two values on the coset (x^N = +-g^N): its inverses come from the host.  is_first = Z_H / (x - 1), is_last = Z_H / (x - w^last):
  // Q = (S0 + is_trans St) / Z_H + Sf / (x - 1) + Sl / (x - w^last), and x_j - w^last = w^last (x_(j - 2 last) - 1)
  const uint32_t inv_zh = (j & 1) ? inv_zh_odd_m : inv_zh_even_m;
  const uint32_t inv_first = inv_xm1[j], inv_last = bb::mont_mul(inv_xm1[(j + N2 - last_shift) & (N2 - 1)], w_last_inv_m);
  const uint32_t is_trans = bb::sub(x, wn_inv_m);
  QuotientOps<DEF, DeviceRowSrc> o{DeviceRowSrc{reinterpret_cast<const uint4*>(L), reinterpret_cast<const uint4*>(AL), N2, j, (j + 2) & (N2 - 1)}, pp->alpha_seq, {}, pp->lk};
  o.init();
  air::eval(o, pp->first_m, pp->last_m, DEF, pp->cnt_m);
  using QO = QuotientOps<DEF, DeviceRowSrc>;
  const E4 s0 = QO::sum_of(o.a0), st = o.st, sf = bb::e_sub(o.sf, pp->cf), sl = bb::e_sub(o.sl, pp->cl);
  const E4 q = bb::e_add(bb::e_add(bb::e_mul_fm(bb::e_add(s0, bb::e_mul_fm(st, is_trans)), inv_zh), bb::e_mul_fm(sf, inv_first)), bb::e_mul_fm(sl, inv_last));
  uint4* q4 = reinterpret_cast<uint4*>(Q);                                     // one B8 block: four coordinate columns (Montgomery, like every matrix of a proof) + four zero columns
  q4[(uint64_t)j * 2] = make_uint4(q.c[0], q.c[1], q.c[2], q.c[3]);
  q4[(uint64_t)j * 2 + 1] = make_uint4(0, 0, 0, 0);
}

// ---- lookup argument (AIR v2): per-row table indices + multiplicities, inverse tables, aux trace -------------------------------------
// The looked-up values of a row live in blocks 0, 1 and the last three of the main-trace matrix: tuple (pc limbs, op, fa, fb, fc | fhi | opclass | s | g)
// and the four range chunks.  One pass over them, BEFORE the LDE uses the matrix as scratch: side[i] = (chunk 0 | chunk 1 << 16,
// chunk 2 | chunk 3 << 16, ROM row u, 0) [v5: eight chunks, three 10-bit chunks per word: (c0 c1 c2, c3 c4 c5, c6 c7, u)], the range-table histogram (LDS-privatised) and the ROM histogram (LDS-privatised for the
// first ROM_LDS rows of the table — a program's hot code — global atomics beyond).  A row whose tuple is not the program's word at
// its pc (self-modified code, pc outside the code segment) or whose chunk is out of range has no proof: its index goes to bad_row.
constexpr uint32_t ROM_LDS = 4096;
// (mode 2) one entry per WRITE / live READ row: what the row sends into the tape lookup — filled here, turned into helper values once the challenges are drawn
struct IoEntry { uint32_t row, is_in, idx, v[3], pad[2]; };
__global__ __launch_bounds__(NT) void lookup_index_kernel(const uint32_t* __restrict__ M, uint64_t N, int deferred /* the mode */, const uint32_t* __restrict__ code, uint32_t n_code, uint4* __restrict__ side,
                                                           uint32_t* __restrict__ rc_mult, uint32_t* __restrict__ rom_mult, unsigned long long* __restrict__ bad_row,
                                                           IoEntry* __restrict__ io_list, uint32_t* __restrict__ io_count, uint32_t* __restrict__ mem_mult, uint4* __restrict__ mem_side,
                                                           uint2* __restrict__ wide_side /* (mode 4) the six extra range values of every row, ten bits each */,
                                                           uint32_t* __restrict__ wide_tape /* (mode 4) [N][8]: a record (cycle, rs1's limbs, rs2's, opcode) per row with ot = 1, appended in any order */) {
  __shared__ uint32_t h_rc[air::RC_TABLE];
  __shared__ uint32_t h_rom[ROM_LDS];
  __shared__ uint32_t h_mem[air::MEM_MULT];                    // (mode 3) LOW3 | BYTE | NIBBLE
  if (deferred >= 3) for (uint32_t k = threadIdx.x; k < (uint32_t)air::MEM_MULT; k += NT) h_mem[k] = 0;
  const uint32_t rom_lds = n_code < ROM_LDS ? n_code : ROM_LDS;
  for (uint32_t k = threadIdx.x; k < (uint32_t)air::RC_TABLE; k += NT) h_rc[k] = 0;
  for (uint32_t k = threadIdx.x; k < rom_lds; k += NT) h_rom[k] = 0;
  __syncthreads();
  const uint4* M4 = reinterpret_cast<const uint4*>(M);
  // committed positions of the columns read here (the instruction tuple's head is the same in both modes; opclass, the chunks and s move)
  const uint32_t p_opc = (uint32_t)air::phys_col(air::C_OPC, deferred), p_rc = (uint32_t)air::phys_col(air::C_RC, deferred), p_s = (uint32_t)air::phys_col(air::C_S, deferred);
  const uint32_t p_rc2 = (uint32_t)air::phys_col(air::C_RC2, deferred), p_g = (uint32_t)air::phys_col(air::C_G, deferred);
  static_assert(air::C_PC == 1 && air::C_OP == 4 && air::C_FHI == 8 && air::C_LIMB == 9, "the tuple's head sits before the first uncommitted column");
  for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < N; i += (uint64_t)gridDim.x * NT) {
    const uint4 b0l = M4[(0 * N + i) * 2], b0h = M4[(0 * N + i) * 2 + 1];                                  // cycle pc0 pc1 pc2 | op fa fb fc
    const uint32_t fhi_v = M[b8(air::C_FHI, i, N)], opc_v = M[b8(p_opc, i, N)], s_v = M[b8(p_s, i, N)], g_v = M[b8(p_g, i, N)];
    const uint32_t r[air::N_RC] = {M[b8(p_rc, i, N)], M[b8(p_rc + 1, i, N)], M[b8(p_rc + 2, i, N)], M[b8(p_rc + 3, i, N)],
                                   M[b8(p_rc2, i, N)], M[b8(p_rc2 + 1, i, N)], M[b8(p_rc2 + 2, i, N)], M[b8(p_rc2 + 3, i, N)]};
    bool ok = true;
    bool mem_row = false;
    int lg_which = -1;                                           // (mode 3) 0 / 1 / 2: the row is an AND / OR / XOR (immediate or not)
    if (deferred >= 3) {
      // (modes 3 / 4: the mode-3 columns sit at the same committed positions in both) the row's memory side, kept for the aux kernels (the LDE overwrites M): pieces, old bytes, old time, window; the piece lookups counted here
      const uint32_t p_kld = (uint32_t)air::phys_col(air::C_KLD, 3), p_e = (uint32_t)air::phys_col(air::C_E, 3), p_ob = (uint32_t)air::phys_col(air::C_OB, 3),
                     p_told = (uint32_t)air::phys_col(air::C_TOLD, 3), p_pc = (uint32_t)air::phys_col(air::C_PIECE, 3);
      const uint32_t kld = M[b8(p_kld, i, N)], kst = M[b8(p_kld + 1, i, N)];
      mem_row = (kld | kst) != 0;
      uint32_t win = 15;
      for (int v = 0; v < air::N_WIN; v++) if (M[b8(p_e + (uint32_t)v, i, N)]) win = (uint32_t)v;
      const uint32_t p_klg = (uint32_t)air::phys_col(air::C_KLG, 3), p_lb = (uint32_t)air::phys_col(air::C_LB, 3), p_lr = (uint32_t)air::phys_col(air::C_LR, 3);
      const uint32_t klg = M[b8(p_klg, i, N)];
      lg_which = klg ? (M[b8(p_klg + 1, i, N)] ? 0 : M[b8(p_klg + 2, i, N)] ? 1 : 2) : -1;
      uint32_t lbits_lo = 0, lbits_hi = 0;                       // b_0..7 | b_8, b_9: kept for the aux kernel
      auto logic_tuple = [&](uint32_t a, int k) {                // the nibble tuple (a, b_k, r_k) of a bitwise row: in its operation's table?
        const uint32_t b = M[b8(p_lb + (uint32_t)k, i, N)], r = M[b8(p_lr + (uint32_t)k, i, N)];
        if (k < 8) lbits_lo |= (b & 15) << (4 * k); else lbits_hi |= (b & 15) << (4 * (k - 8));
        if (a < 16 && b < 16 && r == air::logic_of(lg_which, a, b)) atomicAdd(&h_mem[air::LG_BASE + 256 * lg_which + 16 * a + b], 1u); else ok = false;
      };
      const uint32_t ksh = M[b8((uint32_t)air::phys_col(air::C_KSH, 3), i, N)], sh_reg = ksh && !M[b8((uint32_t)air::phys_col(air::C_SI, 3), i, N)];
      uint32_t kmu = M[b8((uint32_t)air::phys_col(air::C_KMU, 3), i, N)];
      if (deferred == 4) {                                       // (mode 4) a wide-arithmetic row reads the 10-bit table in every piece slot, like a MUL row; its operands must be below 2^40
        const uint32_t p_om = (uint32_t)air::phys_col(air::C_OM, 4);
        const uint32_t ot = M[b8((uint32_t)air::phys_col(air::C_OT, 4), i, N)];
        const uint32_t kwa = M[b8(p_om, i, N)] | M[b8(p_om + 1, i, N)] | M[b8(p_om + 2, i, N)] | ot;       // kwa = om + od + orr + ot
        if (M[b8((uint32_t)air::phys_col(air::C_FH, 4), i, N)]) atomicAdd(io_count + 1, 1u);                 // a hash syscall row: the proof needs its record (the hash tape)
        if (ot) {                                                // a wide row with an operand above 2^40: its record goes into the wide tape (the host sorts the records by cycle)
          uint32_t* r = wide_tape + 8 * (uint64_t)atomicAdd(io_count + 2, 1u);
          r[0] = M[b8((uint32_t)air::phys_col(air::C_CYCLE, 4), i, N)];
          for (int k = 0; k < 3; k++) { r[1 + k] = M[b8((uint32_t)air::phys_col(air::C_XB + k, 4), i, N)]; r[4 + k] = M[b8((uint32_t)air::phys_col(air::C_XC + k, 4), i, N)]; }
          r[7] = M[b8((uint32_t)air::phys_col(air::C_OP, 4), i, N)];
        }
        kmu |= kwa;
        uint32_t xs[air::N_X];
        for (int k = 0; k < air::N_X; k++) { xs[k] = M[b8((uint32_t)air::phys_col(air::C_X + k, 4), i, N)]; if (xs[k] < (uint32_t)air::RC_TABLE) atomicAdd(&h_rc[xs[k]], 1u); else ok = false; }
        wide_side[i] = make_uint2((xs[0] & 1023) | ((xs[1] & 1023) << 10) | ((xs[2] & 1023) << 20), (xs[3] & 1023) | ((xs[4] & 1023) << 10) | ((xs[5] & 1023) << 20));
      }
      uint32_t pc9[air::N_PIECE];
      for (int k = 0; k < air::N_PIECE; k++) {
        pc9[k] = M[b8(p_pc + (uint32_t)k, i, N)];
        if (lg_which >= 0) { logic_tuple(pc9[k], k); continue; }
        if (kmu) { if (pc9[k] < (uint32_t)air::RC_TABLE) atomicAdd(&h_rc[pc9[k]], 1u); else ok = false; continue; }   // a MUL row: every piece slot reads the 10-bit range table
        if (ksh) {                                               // a shift row: the slots are re-typed (air::shift_piece_tag); piece 8 of a register shift goes to LOW6 with the amount
          const int stag = air::shift_piece_tag(k, sh_reg != 0);
          if (stag == air::TAG_LOW6) {
            const uint32_t b = M[b8(p_lb + 8, i, N)];
            lbits_hi = b & 63;                                   // (kept for nothing but symmetry: the aux kernel needs the value only)
            if (pc9[k] < (uint32_t)air::RC_TABLE && b == (pc9[k] & 63)) atomicAdd(&h_mem[air::L6_BASE + pc9[k]], 1u); else ok = false;
          } else if (stag == air::TAG_NIB) { if (pc9[k] < 16u) atomicAdd(&h_mem[air::RC_TABLE + 256 + pc9[k]], 1u); else ok = false; }
          else { if (pc9[k] < (uint32_t)air::RC_TABLE) atomicAdd(&h_rc[pc9[k]], 1u); else ok = false; }
          continue;
        }
        const int tag = air::piece_tag(k);
        if (tag == air::TAG_BYTE) { if (pc9[k] < 256u) atomicAdd(&h_mem[air::RC_TABLE + pc9[k]], 1u); else ok = false; }
        else if (tag == air::TAG_NIB) { if (pc9[k] < 16u) atomicAdd(&h_mem[air::RC_TABLE + 256 + pc9[k]], 1u); else ok = false; }
        else { if (pc9[k] < (uint32_t)air::RC_TABLE) atomicAdd(&h_rc[pc9[k]], 1u); else ok = false; }
      }
      uint32_t obl = 0, obh = 0;
      for (int k = 0; k < 4; k++) { obl |= (M[b8(p_ob + (uint32_t)k, i, N)] & 0xFF) << (8 * k); obh |= (M[b8(p_ob + 4 + (uint32_t)k, i, N)] & 0xFF) << (8 * k); }
      // side: the nine pieces at ten bits each (a shift row's are chunks), the flags (kld | kst << 1 | window << 2 | (bitwise op + 1) << 6 | b_8, b_9 << 8 | ksh << 16 | register shift << 17 | kmu << 18)
      mem_side[2 * i] = make_uint4((pc9[0] & 1023) | ((pc9[1] & 1023) << 10) | ((pc9[2] & 1023) << 20), (pc9[3] & 1023) | ((pc9[4] & 1023) << 10) | ((pc9[5] & 1023) << 20),
                                   (pc9[6] & 1023) | ((pc9[7] & 1023) << 10) | ((pc9[8] & 1023) << 20), 0u);
      if (lg_which >= 0) logic_tuple(r[air::N_RC - 1], 9);       // the tenth tuple: a_9 = the last range chunk
      mem_side[2 * i].w = kld | (kst << 1) | (win << 2) | ((uint32_t)(lg_which + 1) << 6) | ((lg_which >= 0 ? lbits_hi : 0u) << 8) | (ksh << 16) | (sh_reg << 17) | (kmu << 18);
      mem_side[2 * i + 1] = make_uint4(obl, obh, M[b8(p_told, i, N)], lbits_lo);      // (the cycle of a whole run's row is its index)
      if (mem_row) {                                             // the first chunk goes to the LOW3 table, with the window's offset
        const uint32_t off = win < 15 ? (uint32_t)air::win_start((int)win) : 8u;
        if (r[0] < (uint32_t)air::RC_TABLE && (r[0] & 7) == off) atomicAdd(&h_mem[r[0]], 1u); else ok = false;
      }
    }
#pragma unroll
    for (int k = 0; k < air::N_RC; k++) { if ((k == 0 && mem_row) || (k == air::N_RC - 1 && lg_which >= 0)) continue; if (r[k] < (uint32_t)air::RC_TABLE) atomicAdd(&h_rc[r[k]], 1u); else ok = false; }
    const uint64_t pc = (uint64_t)b0l.y | ((uint64_t)b0l.z << 20) | ((uint64_t)b0l.w << 40);
    const uint64_t u = (pc - 0x1000) >> 2;
    uint32_t ui = 0;
    if (b0l.y < (1u << 20) && b0l.z < (1u << 20) && b0l.w < (1u << 24) && pc >= 0x1000 && !(pc & 3) && u < n_code) {
      const uint32_t w = code[u];
      ui = (uint32_t)u;
      if (b0h.x == (w & 0x7F) && b0h.y == ((w >> 7) & 0xF) && b0h.z == ((w >> 11) & 0xF) && b0h.w == ((w >> 15) & 0xF) && fhi_v == (w >> 19) && s_v == (w >> 31) &&
          opc_v == air::opclass_of(w & 0x7F, deferred) && g_v == air::variant_bit(w & 0x7F, deferred)) {
        if (ui < rom_lds) atomicAdd(&h_rom[ui], 1u); else atomicAdd(&rom_mult[ui], 1u);
      } else ok = false;
    } else ok = false;
    if (!ok) atomicMin(bad_row, (unsigned long long)i);
    // (a chunk outside the table has no proof anyway: masked here so that the packed word stays well-formed)
    auto c10 = [&](int k) { return r[k] & (uint32_t)(air::RC_TABLE - 1); };
    side[i] = make_uint4(c10(0) | (c10(1) << 10) | (c10(2) << 20), c10(3) | (c10(4) << 10) | (c10(5) << 20), c10(6) | (c10(7) << 10), ui);
    if (deferred >= 2) {                                       // the tape lookups of the row (rare: ecall rows only)
      const uint32_t f2 = M[b8((uint32_t)air::phys_col(air::C_F2, 2), i, N)], rl = M[b8((uint32_t)air::phys_col(air::C_RL, 2), i, N)];     // (the same committed positions in modes 2 and 3)
      if (f2 | rl) {
        IoEntry e;
        e.row = (uint32_t)i; e.is_in = rl ? 1u : 0u;
        e.idx = M[b8((uint32_t)air::phys_col(rl ? air::C_IC : air::C_OC, 2), i, N)];
        const int v0 = rl ? air::C_Y : air::C_LIMB + 33;       // a live READ sends what it writes to R10 (y), a WRITE sends R11's limbs
        for (int l = 0; l < 3; l++) e.v[l] = M[b8((uint32_t)air::phys_col(v0 + l, 2), i, N)];
        e.pad[0] = e.pad[1] = 0;
        io_list[atomicAdd(io_count, 1u)] = e;
      }
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < (uint32_t)air::RC_TABLE; k += NT) if (h_rc[k]) atomicAdd(&rc_mult[k], h_rc[k]);
  for (uint32_t k = threadIdx.x; k < rom_lds; k += NT) if (h_rom[k]) atomicAdd(&rom_mult[k], h_rom[k]);
  if (deferred >= 3) for (uint32_t k = threadIdx.x; k < (uint32_t)air::MEM_MULT; k += NT) if (h_mem[k]) atomicAdd(&mem_mult[k], h_mem[k]);
}
static_assert(air::phys_col(air::C_KLD, 3) == air::phys_col(air::C_KLD, 4) && air::phys_col(air::C_KMU, 3) == air::phys_col(air::C_KMU, 4) && air::phys_col(air::C_F2, 2) == air::phys_col(air::C_F2, 4), "mode 4 commits the mode-3 columns at mode 3's positions");
static_assert(air::phys_col(air::C_F2, 2) == air::phys_col(air::C_F2, 3) && air::phys_col(air::C_IC, 2) == air::phys_col(air::C_IC, 3) && air::phys_col(air::C_Y, 2) == air::phys_col(air::C_Y, 3),
              "modes 2 and 3 commit the mode-2 columns at the same positions");

__device__ __forceinline__ uint4 add4m(uint4 a, uint4 b);
// (mode 3) inverse tables of LOW3 | BYTE | NIBBLE for the drawn challenges: 1 / (alpha - v - lambda (v & 7) - 4 lambda^11), 1 / (alpha - v - 5 lambda^11), 1 / (alpha - v - 6 lambda^11)
__global__ __launch_bounds__(NT) void mem_tables_kernel(const ProveParams* __restrict__ pp, E4* __restrict__ inv_mem) {
  const uint32_t t = blockIdx.x * NT + threadIdx.x;
  if (t >= (uint32_t)air::MEM_MULT) return;
  const bool l6 = t >= (uint32_t)air::L6_BASE;                  // LOW6: entry v = the tuple (v, v & 63)
  const bool lg = !l6 && t >= (uint32_t)air::LG_BASE;           // AND | OR | XOR: entry 16 a + b = the tuple (a, b, a op b)
  const uint32_t which = lg ? (t - air::LG_BASE) >> 8 : 0, e = (t - air::LG_BASE) & 255;
  const uint32_t v = l6 ? t - air::L6_BASE : lg ? e >> 4 : t < (uint32_t)air::RC_TABLE ? t : t < (uint32_t)air::RC_TABLE + 256 ? t - air::RC_TABLE : t - air::RC_TABLE - 256;
  const uint32_t tag = l6 ? air::TAG_LOW6 : lg ? air::TAG_AND + which : t < (uint32_t)air::RC_TABLE ? air::TAG_LOW3 : t < (uint32_t)air::RC_TABLE + 256 ? air::TAG_BYTE : air::TAG_NIB;
  E4 d;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t fp = bb::mont_mul(pp->lk[air::LK_LAM + 4 * air::N_TUPLE + k], bb::to_mont(tag));
    if (tag == (uint32_t)air::TAG_LOW3) fp = bb::add(fp, bb::mont_mul(pp->lk[air::LK_LAM + 4 + k], bb::to_mont(v & 7)));
    if (l6) fp = bb::add(fp, bb::mont_mul(pp->lk[air::LK_LAM + 4 + k], bb::to_mont(v & 63)));
    if (lg) fp = bb::add(fp, bb::add(bb::mont_mul(pp->lk[air::LK_LAM + 4 + k], bb::to_mont(e & 15)), bb::mont_mul(pp->lk[air::LK_LAM + 8 + k], bb::to_mont(air::logic_of((int)which, e >> 4, e & 15)))));
    d.c[k] = bb::sub(pp->lk[air::LK_ALPHA + k], fp);
  }
  d.c[0] = bb::sub(d.c[0], bb::to_mont(v));
  inv_mem[t] = bb::e_inv_m(d);
}
// (mode 3) the chunk digests of a long proof section (so::observe_section: chunks of 512 words, each hashed on its own with the rate-8 overwrite sponge, so::hash_elems): a
// quad of lanes per chunk, 64 sequential permutations each — on the host the touched-cell list of a memory-heavy run (seven words per cell) cost more than the rest of the proof
constexpr uint32_t SECTION_CHUNK = 512;
__global__ __launch_bounds__(64) void section_hash_kernel(const p2::Consts* __restrict__ cp, const uint32_t* __restrict__ w, uint64_t n_words, uint32_t* __restrict__ digests) {
  // a QUAD of lanes per chunk (p2::permute_quad_scaled: the 64 permutations of a chunk are sequential and a section has few chunks — latency, not throughput): lane l
  // holds words l and 4 + l of the rate and capacity word 8 + l
  const uint64_t t = (uint64_t)blockIdx.x * 64 + threadIdx.x, c = t >> 2, at = c * SECTION_CHUNK;
  const int l = (int)(t & 3);
  if (at >= n_words) return;                                                  // whole quads leave together
  const uint64_t len = n_words - at < SECTION_CHUNK ? n_words - at : SECTION_CHUNK;
  uint32_t st[3] = {0, 0, 0};
  const uint32_t k_in = cp->in_scale, carry = cp->carry;
  for (uint64_t off = 0; off < len; off += p2::RATE) {                        // (a word the ragged last block does not overwrite stays: so::hash_elems)
    st[0] = off + l < len ? bb::mont_mul_lazy(w[at + off + l], k_in) : bb::mont_mul_lazy(st[0], carry);
    st[1] = off + 4 + l < len ? bb::mont_mul_lazy(w[at + off + 4 + l], k_in) : bb::mont_mul_lazy(st[1], carry);
    st[2] = bb::mont_mul_lazy(st[2], carry);
    p2::permute_quad_scaled(st, l, *cp);
  }
  digests[4 * c + l] = bb::mont_mul(st[0], cp->out_scale);
}
// (mode 3) the two ends of the memory check, the VERIFIER's share of the table side, formed on the device: per touched cell + 1 / (alpha - fp(cell, time 0, the program image's
// bytes)) - 1 / (alpha - fp(cell, final time, final bytes)), summed per workgroup (the host adds the partial sums).  image = the program's code + data bytes (loaded at 0x1000).
__global__ __launch_bounds__(NT) void mem_cells_sum_kernel(const uint64_t* __restrict__ addr, const uint64_t* __restrict__ bytes, const uint32_t* __restrict__ time, uint32_t n_cells,
                                                            const uint8_t* __restrict__ image, uint64_t image_len, const ProveParams* __restrict__ pp, E4* __restrict__ partial) {
  __shared__ E4 red[NT];
  const uint32_t k = blockIdx.x * NT + threadIdx.x;
  E4 term = bb::e_zero();
  if (k < n_cells) {
    const uint64_t a = addr[k];
    uint64_t img = 0;
    for (int b = 0; b < 8; b++) { const uint64_t x = a + b; if (x >= 0x1000 && x - 0x1000 < image_len) img |= (uint64_t)image[x - 0x1000] << (8 * b); }
    auto tuple_inv = [&](uint32_t t, uint64_t by) {
      const uint32_t g[11] = {(uint32_t)(a & 0xFFFFF), (uint32_t)((a >> 20) & 0xFFFFF), t, (uint32_t)(by & 0xFF), (uint32_t)((by >> 8) & 0xFF), (uint32_t)((by >> 16) & 0xFF), (uint32_t)((by >> 24) & 0xFF),
                              (uint32_t)((by >> 32) & 0xFF), (uint32_t)((by >> 40) & 0xFF), (uint32_t)((by >> 48) & 0xFF), (uint32_t)(by >> 56)};
      E4 d;
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++) {
        uint32_t f = bb::mont_mul(pp->lk[air::LK_LAM + 4 * air::N_TUPLE + c4], bb::to_mont((uint32_t)air::TAG_MEM));
#pragma unroll
        for (int j = 0; j < 11; j++) f = bb::add(f, bb::mont_mul(pp->lk[air::LK_LAM + 4 * j + c4], bb::to_mont(g[j])));
        d.c[c4] = bb::sub(pp->lk[air::LK_ALPHA + c4], f);
      }
      return bb::e_inv_m(d);
    };
    term = bb::e_sub(tuple_inv(0, img), tuple_inv(time[k], bytes[k]));
  }
  red[threadIdx.x] = term;
  __syncthreads();
  for (uint32_t off = NT / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] = bb::e_add(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// (mode 3) the memory columns of the aux trace, every row: P0..P8 (table reads), FPN = sum_k lambda^(3+k) (new byte k), and on load / store rows HMR = 1 / (alpha - fp(cell,
// told, old bytes)), HMW = 1 / (alpha - fp(cell, cycle + 1, new bytes)), H0 re-read from the LOW3 table; the row's running-sum increment (S slot, written by aux_rows_kernel)
// gains P0 + .. + P8 + HMR - HMW (and H0's correction).  Blocks A_P / 8 ..: P0 | P1, P2 | P3, P4 | P5, P6 | P7, P8 | HMR, HMW | FPN.
__global__ __launch_bounds__(NT) void mem_aux_kernel(const uint4* __restrict__ side, const uint4* __restrict__ mem_side, uint64_t N, const E4* __restrict__ inv_rc, const E4* __restrict__ inv_mem,
                                                      const ProveParams* __restrict__ pp, uint32_t* __restrict__ A, const uint2* __restrict__ wide_side /* mode 4, else null */) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= N) return;
  const uint4 m0 = mem_side[2 * i], m1 = mem_side[2 * i + 1];
  const uint32_t pc9[air::N_PIECE] = {m0.x & 1023, (m0.x >> 10) & 1023, m0.x >> 20, m0.y & 1023, (m0.y >> 10) & 1023, m0.y >> 20, m0.z & 1023, (m0.z >> 10) & 1023, m0.z >> 20};
  const uint32_t fl = m0.w;                                     // kld | kst << 1 | window << 2 | (bitwise op + 1) << 6 | b_8, b_9 << 8 | ksh << 16 | register shift << 17
  uint4* A4 = reinterpret_cast<uint4*>(A);
  auto put = [&](int col, const E4& c) { A4[((uint64_t)(col >> 3) * N + i) * 2 + ((col >> 2) & 1)] = make_uint4(c.c[0], c.c[1], c.c[2], c.c[3]); };
  E4 inc = bb::e_zero();
  const int lg_which = (int)((fl >> 6) & 3) - 1;                // 0 / 1 / 2: an AND / OR / XOR row — every piece slot then reads its nibble tuple's inverse from the operation's table
  const bool ksh = (fl >> 16) & 1, sh_reg = (fl >> 17) & 1;     // a shift row: the slots re-typed (air::shift_piece_tag)
  const bool kmu = (fl >> 18) & 1;                              // a MUL row: the 10-bit range table in every slot
  auto lg_b = [&](int k) { return k < 8 ? (m1.w >> (4 * k)) & 15 : (fl >> (8 + 4 * (k - 8))) & 15; };
#pragma unroll
  for (int k = 0; k < air::N_PIECE; k++) {
    const int tag = air::piece_tag(k);
    const int stag = air::shift_piece_tag(k, sh_reg);
    const E4 h = lg_which >= 0 ? inv_mem[air::LG_BASE + 256 * lg_which + 16 * (pc9[k] & 15) + lg_b(k)]
               : kmu ? inv_rc[pc9[k]]
               : ksh ? (stag == air::TAG_LOW6 ? inv_mem[air::L6_BASE + pc9[k]] : stag == air::TAG_NIB ? inv_mem[air::RC_TABLE + 256 + (pc9[k] & 15)] : inv_rc[pc9[k]])
               : tag == air::TAG_BYTE ? inv_mem[air::RC_TABLE + pc9[k]] : tag == air::TAG_NIB ? inv_mem[air::RC_TABLE + 256 + pc9[k]] : inv_rc[pc9[k]];
    put(air::A_P + 4 * k, h); inc = bb::e_add(inc, h);
  }
  if (lg_which >= 0) {                                          // the tenth tuple sits in the last range slot: H7 re-read from the operation's table
    const uint4 sd = side[i];
    const uint32_t c7 = sd.z >> 10;
    const E4 h7 = inv_mem[air::LG_BASE + 256 * lg_which + 16 * (c7 & 15) + lg_b(9)];
    inc = bb::e_add(inc, bb::e_sub(h7, inv_rc[c7]));
    put(air::A_H + 4 * (air::N_RC - 1), h7);
  }
  const uint32_t kld = fl & 1, kst = (fl >> 1) & 1, win = (fl >> 2) & 15;
  const uint64_t ob = (uint64_t)m1.x | ((uint64_t)m1.y << 32);
  uint64_t nb = ob;
  int off = 0;
  if (win < 15) {                                              // the window's bytes replaced by the pieces' (a load's pieces ARE the window's bytes: nb = ob)
    const uint64_t D = (uint64_t)pc9[0] | ((uint64_t)pc9[1] << 8) | ((uint64_t)(pc9[2] | (pc9[3] << 4)) << 16) | ((uint64_t)pc9[4] << 24) | ((uint64_t)pc9[5] << 32) | ((uint64_t)pc9[6] << 40) |
                       ((uint64_t)pc9[7] << 48) | ((uint64_t)pc9[8] << 56);
    const int w = air::win_width((int)win);
    off = air::win_start((int)win);
    const uint64_t mask = w == 8 ? ~0ull : ((1ull << (8 * w)) - 1);
    if (kst) nb = (ob & ~(mask << (8 * off))) | ((D & mask) << (8 * off));
  }
  auto bytes_fp = [&](uint64_t b) {
    E4 f = bb::e_zero();
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t bm = bb::to_mont((uint32_t)((b >> (8 * k)) & 0xFF));
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++) f.c[c4] = bb::add(f.c[c4], bb::mont_mul(pp->lk[air::LK_LAM + 4 * (3 + k) + c4], bm));
    }
    return f;
  };
  const E4 fpn = bytes_fp(nb);
  put(air::A_FPN, fpn);
  E4 hr = bb::e_zero(), hw = bb::e_zero();
  if (kld | kst) {
    const uint4 sd = side[i];
    const uint32_t c0 = sd.x & 1023, z0 = c0 | (((sd.x >> 10) & 1023) << 10), z1 = (sd.x >> 20) | ((sd.y & 1023) << 10);
    const uint32_t a0m = bb::to_mont(z0 - (uint32_t)off), a1m = bb::to_mont(z1);
    auto tuple_inv = [&](uint32_t t, const E4& bfp) {
      const uint32_t tm = bb::to_mont(t);
      E4 d;
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++) {
        uint32_t fp = bb::mont_mul(pp->lk[air::LK_LAM + 4 * air::N_TUPLE + c4], bb::to_mont((uint32_t)air::TAG_MEM));
        fp = bb::add(fp, bb::mont_mul(pp->lk[air::LK_LAM + c4], a0m));
        fp = bb::add(fp, bb::mont_mul(pp->lk[air::LK_LAM + 4 + c4], a1m));
        fp = bb::add(fp, bb::mont_mul(pp->lk[air::LK_LAM + 8 + c4], tm));
        d.c[c4] = bb::sub(pp->lk[air::LK_ALPHA + c4], bb::add(fp, bfp.c[c4]));
      }
      return bb::e_inv_m(d);
    };
    hr = tuple_inv(m1.z, bytes_fp(ob));
    hw = tuple_inv((uint32_t)((i + 1) % bb::P), fpn);            // time written = cycle + 1; a whole run's row i has cycle i
    inc = bb::e_add(inc, bb::e_sub(hr, hw));
    const E4 h0 = inv_mem[c0];                                  // the first chunk's helper comes from the LOW3 table on a memory row
    inc = bb::e_add(inc, bb::e_sub(h0, inv_rc[c0]));
    put(air::A_H, h0);
  }
  put(air::A_HMR, hr); put(air::A_HMW, hw);
  if (wide_side) {                                             // (mode 4) XH_k = 1 / (alpha - X_k): the six extra range slots, on every row
    const uint2 xs = wide_side[i];
    const uint32_t x6[air::N_X] = {xs.x & 1023, (xs.x >> 10) & 1023, xs.x >> 20, xs.y & 1023, (xs.y >> 10) & 1023, xs.y >> 20};
#pragma unroll
    for (int k = 0; k < air::N_X; k++) { const E4 h = inv_rc[x6[k]]; put(air::A_X + 4 * k, h); inc = bb::e_add(inc, h); }
  }
  uint4* S = A4 + ((uint64_t)(air::A_S / 8) * N + i) * 2 + 1;
  *S = add4m(*S, make_uint4(inc.c[0], inc.c[1], inc.c[2], inc.c[3]));
}

// inverse tables for the drawn challenges: inv_rc[t] = 1 / (alpha - t), inv_rom[u] = 1 / (alpha - fingerprint(ROM row u))  (Montgomery E4).
// Every lookup of the aux trace is then a table read, and the verifier-side sum T = sum m_t inv_rc[t] + sum r_u inv_rom[u] is formed
// from the same tables on the host.
__global__ __launch_bounds__(NT) void lookup_tables_kernel(const uint32_t* __restrict__ code, uint32_t n_code, const ProveParams* __restrict__ pp, E4* __restrict__ inv_rc,
                                                            E4* __restrict__ inv_rom, int mode) {
  const uint32_t t = blockIdx.x * NT + threadIdx.x;
  if (t >= (uint32_t)air::RC_TABLE + n_code) return;
  E4 d;
#pragma unroll
  for (int k = 0; k < 4; k++) d.c[k] = pp->lk[air::LK_ALPHA + k];
  if (t < (uint32_t)air::RC_TABLE) {
    d.c[0] = bb::sub(d.c[0], bb::to_mont(t));
    inv_rc[t] = bb::e_inv_m(d);
    return;
  }
  const uint32_t u = t - air::RC_TABLE, w = code[u];
  const uint64_t pc = 0x1000 + 4ull * u;
  const uint32_t f[air::N_TUPLE] = {(uint32_t)(pc & 0xFFFFF), (uint32_t)((pc >> 20) & 0xFFFFF), (uint32_t)(pc >> 40), w & 0x7F, (w >> 7) & 0xF, (w >> 11) & 0xF, (w >> 15) & 0xF,
                                    w >> 19, w >> 31, air::opclass_of(w & 0x7F, mode), air::variant_bit(w & 0x7F, mode)};
  // (fingerprint coordinates in Montgomery form: lk holds R * lambda^j_k, mont_mul(R a, to_mont(f)) = R a f)
  E4 fpm;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t fp = pp->lk[air::LK_LAM + 4 * air::N_TUPLE + k];
#pragma unroll
    for (int j = 0; j < air::N_TUPLE; j++) fp = bb::add(fp, bb::mont_mul(pp->lk[air::LK_LAM + 4 * j + k], bb::to_mont(f[j])));
    fpm.c[k] = fp;
  }
  E4 dd;
#pragma unroll
  for (int k = 0; k < 4; k++) dd.c[k] = bb::sub(pp->lk[air::LK_ALPHA + k], fpm.c[k]);
  inv_rom[u] = bb::e_inv_m(dd);
}

// aux rows: H0..H7, HR (Montgomery form) into blocks 0..4 of the aux matrix, and in the S slot (the second half of block 4) the row's increment
// d_i = H0 + .. + H7 + HR - T / N
// (the scan kernels below turn the increments into the running sum S_i = sum_{j < i} d_j)
__global__ __launch_bounds__(NT) void aux_rows_kernel(const uint4* __restrict__ side, uint64_t N, const E4* __restrict__ inv_rc, const E4* __restrict__ inv_rom,
                                                       const ProveParams* __restrict__ pp, uint32_t* __restrict__ A) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= N) return;
  const uint4 sd = side[i];
  const uint32_t ch[air::N_RC] = {sd.x & 1023, (sd.x >> 10) & 1023, sd.x >> 20, sd.y & 1023, (sd.y >> 10) & 1023, sd.y >> 20, sd.z & 1023, sd.z >> 10};
  uint4* A4 = reinterpret_cast<uint4*>(A);
  // (Montgomery words, as they come out of the inverse tables: the aux matrix rests in Montgomery form like every prover matrix — the LDE is linear)
  auto put = [&](uint32_t blk, uint32_t half, const E4& c) { A4[((uint64_t)blk * N + i) * 2 + half] = make_uint4(c.c[0], c.c[1], c.c[2], c.c[3]); };
  const E4 hr = inv_rom[sd.w];
  E4 d = hr;
#pragma unroll
  for (int k = 0; k < air::N_RC; k++) { const E4 h = inv_rc[ch[k]]; d = bb::e_add(d, h); put((uint32_t)k >> 1, (uint32_t)k & 1, h); }
#pragma unroll
  for (int k = 0; k < 4; k++) d.c[k] = bb::sub(d.c[k], pp->lk[air::LK_TN + k]);
  static_assert(air::A_HR == 4 * air::N_RC && air::A_S == air::A_HR + 4 && air::A_HR % 8 == 0, "aux layout: helpers, then HR | S in one block");
  put(air::A_HR / 8, 0, hr); put(air::A_HR / 8, 1, d);
}

__device__ __forceinline__ uint4 add4m(uint4 a, uint4 b);
// (mode 2) the tape helpers of the WRITE / live READ rows: h = 1 / (alpha - fingerprint(index, limbs; tag)) into HO / HI (aux block A_HO / 8: HO | HI) of the row, and added
// to the row's running-sum increment (the S slot written by aux_rows_kernel); every other row keeps the zeros the block was cleared with
__global__ __launch_bounds__(NT) void io_aux_kernel(const IoEntry* __restrict__ io_list, uint32_t n_io, uint64_t N, const ProveParams* __restrict__ pp, uint32_t* __restrict__ A) {
  const uint32_t t = blockIdx.x * NT + threadIdx.x;
  if (t >= n_io) return;
  const IoEntry e = io_list[t];
  const uint32_t g[4] = {bb::to_mont(e.idx), bb::to_mont(e.v[0]), bb::to_mont(e.v[1]), bb::to_mont(e.v[2])};
  E4 d;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t fp = bb::mont_mul(pp->lk[air::LK_LAM + 4 * air::N_TUPLE + k], bb::to_mont(e.is_in ? 3u : 2u));
#pragma unroll
    for (int j = 0; j < 4; j++) fp = bb::add(fp, bb::mont_mul(pp->lk[air::LK_LAM + 4 * j + k], g[j]));
    d.c[k] = bb::sub(pp->lk[air::LK_ALPHA + k], fp);
  }
  const E4 h = bb::e_inv_m(d);
  uint4* A4 = reinterpret_cast<uint4*>(A);
  static_assert(air::A_HO % 8 == 0 && air::A_HI == air::A_HO + 4, "aux layout: HO | HI in one block");
  A4[((uint64_t)(air::A_HO / 8) * N + e.row) * 2 + (e.is_in ? 1 : 0)] = make_uint4(h.c[0], h.c[1], h.c[2], h.c[3]);
  uint4* S = A4 + ((uint64_t)(air::A_S / 8) * N + e.row) * 2 + 1;
  *S = add4m(*S, make_uint4(h.c[0], h.c[1], h.c[2], h.c[3]));
}

// Exclusive prefix sum (coordinate-wise, mod p) over the S slot of the aux matrix (block A_S / 8, second half), three launches:
//   local: every workgroup scans SCAN_ROWS consecutive rows in place and leaves their total in sums[block]
//   sums:  one workgroup turns sums[] into exclusive offsets
//   add:   every row adds its workgroup's offset
constexpr uint32_t SCAN_PER = 4, SCAN_ROWS = NT * SCAN_PER;
__device__ __forceinline__ uint4 add4m(uint4 a, uint4 b) { return make_uint4(bb::add(a.x, b.x), bb::add(a.y, b.y), bb::add(a.z, b.z), bb::add(a.w, b.w)); }
// (mode 4) the hash-call helpers HH = 1 / (alpha - fp(call)) of the hash-syscall rows: computed on the host from the tape's records (the same values the table side is made
// of), scattered into the row's HH columns and added to its running-sum increment; every other row keeps the zeros the block was cleared with
struct HashAux { uint32_t row, pad[3]; E4 h; };
// `half` = 0: HH (the hash-call helper), 1: WW (the wide-tape helper, the other half of the same block of eight columns)
__global__ __launch_bounds__(NT) void hash_aux_kernel(const HashAux* __restrict__ list, uint32_t n, uint64_t N, uint32_t* __restrict__ A, uint32_t half) {
  const uint32_t t = blockIdx.x * NT + threadIdx.x;
  if (t >= n) return;
  const HashAux e = list[t];
  uint4* A4 = reinterpret_cast<uint4*>(A);
  static_assert(air::A_HH % 8 == 0 && air::A_WW == air::A_HH + 4, "aux layout: HH opens a block, WW is its second half");
  const uint4 h4 = make_uint4(e.h.c[0], e.h.c[1], e.h.c[2], e.h.c[3]);
  A4[((uint64_t)(air::A_HH / 8) * N + e.row) * 2 + half] = h4;
  uint4* S = A4 + ((uint64_t)(air::A_S / 8) * N + e.row) * 2 + 1;
  *S = add4m(*S, h4);
}
__device__ __forceinline__ uint4 block_exclusive_scan(uint4 v, uint4* lds /* NT */, uint4& total) {    // exclusive scan of one value per thread across the workgroup
  const uint32_t t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (uint32_t off = 1; off < NT; off <<= 1) {
    uint4 x = lds[t];
    if (t >= off) x = add4m(x, lds[t - off]);
    __syncthreads();
    lds[t] = x;
    __syncthreads();
  }
  total = lds[NT - 1];
  const uint4 ex = t ? lds[t - 1] : make_uint4(0, 0, 0, 0);
  __syncthreads();
  return ex;
}
__global__ __launch_bounds__(NT) void scan_local_kernel(uint32_t* __restrict__ A, uint64_t N, uint4* __restrict__ sums) {
  __shared__ uint4 lds[NT];
  uint4* S = reinterpret_cast<uint4*>(A) + (uint64_t)(air::A_S / 8) * N * 2 + 1;          // element i at S[2 i]
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_ROWS + (uint64_t)threadIdx.x * SCAN_PER;
  uint4 v[SCAN_PER], run = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER; k++) { v[k] = base + k < N ? S[2 * (base + k)] : make_uint4(0, 0, 0, 0); }
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER; k++) { const uint4 x = v[k]; v[k] = run; run = add4m(run, x); }     // thread-local exclusive
  uint4 total;
  const uint4 ex = block_exclusive_scan(run, lds, total);
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER; k++) if (base + k < N) S[2 * (base + k)] = add4m(v[k], ex);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(NT) void scan_sums_kernel(uint4* __restrict__ sums, uint32_t n) {
  __shared__ uint4 lds[NT];
  uint4 carry = make_uint4(0, 0, 0, 0);
  for (uint32_t b = 0; b < n; b += NT) {
    const uint32_t i = b + threadIdx.x;
    const uint4 v = i < n ? sums[i] : make_uint4(0, 0, 0, 0);
    uint4 total;
    const uint4 ex = block_exclusive_scan(v, lds, total);
    if (i < n) sums[i] = add4m(ex, carry);
    carry = add4m(carry, total);
  }
}
__global__ __launch_bounds__(NT) void scan_add_kernel(uint32_t* __restrict__ A, uint64_t N, const uint4* __restrict__ sums) {
  uint4* S = reinterpret_cast<uint4*>(A) + (uint64_t)(air::A_S / 8) * N * 2 + 1;
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= N) return;
  S[2 * i] = add4m(S[2 * i], sums[i / SCAN_ROWS]);
}

// ---- boundary states: the 68 state words of rows 0 and last_row of the main trace (B8 layout) -> out[136] ----------------------------
// (mode 2: + the counters (oc, ic) of both rows at out[2 NS .. 2 NS + 4))
__global__ void boundary_states_kernel(const uint32_t* __restrict__ M, uint64_t N, uint64_t last_row, int deferred /* the mode */, uint32_t* __restrict__ out) {
  const uint32_t i = threadIdx.x;
  if (deferred >= 2 && i >= 2 * NS && i < 2 * NS + 4) { const uint32_t k = i - 2 * NS; out[i] = M[b8((uint32_t)air::phys_col(air::C_OC + (k & 1), 2), k < 2 ? 0 : last_row, N)]; return; }
  if (i >= 2 * NS) return;
  const int k = air::state_col((int)(i % NS));                 // logical column; an uncommitted one (R0's limbs, default-mode storage states) is the constant 0
  out[i] = air::is_virtual(k, deferred) ? 0u : M[b8((uint32_t)air::phys_col(k, deferred), i < (uint32_t)NS ? 0 : last_row, N)];
}

// ---- barycentric weights over the LDE coset: e_j = x_j / (zeta - x_j)  (Montgomery E4, AoS) -------------------------------
// and dinv_j = 1 / (zeta - x_j), which the DEEP kernel reuses: 1 / (zeta w - x_j) = w^-1 / (zeta - x_{j-2}) on the 2N coset
__global__ __launch_bounds__(NT) void bary_weights_kernel(uint32_t log_n, const uint32_t* __restrict__ tw_fwd, const ProveParams* __restrict__ pp, E4* __restrict__ wts, E4* __restrict__ dinv) {
  const uint32_t N2 = 2u << log_n, N = N2 >> 1;
  const uint32_t j = blockIdx.x * NT + threadIdx.x;
  if (j >= N2) return;
  const uint32_t wj = j < N ? tw_fwd[j] : bb::neg(tw_fwd[j - N]);
  const uint32_t x = bb::mont_mul(wj, bb::to_mont(bb::GEN));
  E4 d = pp->zeta; d.c[0] = bb::sub(d.c[0], x);
  const E4 di = bb::e_inv_m(d);
  dinv[j] = di;
  wts[j] = bb::e_mul_fm(di, x);
}

// partial[col][chunk][2] = Σ_{j in chunk} v_j * e_j  and  Σ v_j * e_{j-2}  (MONTGOMERY E4).  ONE launch for the three matrices (round 5; round 4: three launches of a
// kernel that held 4 columns x 4 coordinates x 2 points = 96 accumulator registers per lane, spilled 16 of them at four waves per SIMD, read 16 of every 64 bytes a lane
// touched and re-read the weights once per four columns): grid = chunks x B8 blocks (main, aux, then the quotient's half block).
// A QUAD of lanes takes one row of a block: every lane of the quad reads the row's 32 bytes (eight columns — the whole row, so every byte of a fetched line that can be used is)
// and ONE coordinate t = lane & 3 of the two weights, and keeps 8 columns x 2 points = 16 exact 96-bit sums (bb::mad96: 2 instructions per term) = 48 registers.  The
// weights are read once per eight columns.  v and e both rest in Montgomery form: the reduction's division by R leaves R * Σ v e.
// step = 2 (trace matrices) sums over the EVEN positions only: x_(2i) = g w_N^i is the coset g H_N, on which a polynomial of degree < N (every trace column: the LDE of N values) is
// determined as well, with the same weights x_j / (zeta - x_j) and the scale ((zeta / g)^N - 1) / N — half the multiply-adds (the rows skipped share their 128-byte lines with the rows read:
// the HBM traffic is the whole matrix); j - 2 stays even, so the second opening point rides along.  The quotient's columns keep step = 1: an HONEST quotient has degree < N too, but the prover must
// make the same (worthless) proof of a false claim as the oracle does (tests: forged outputs), and there the quotient is whatever the division leaves on the 2 N points.
constexpr uint32_t BARY_ROWS = NT / 4;                         // rows a workgroup takes per iteration
constexpr uint64_t BARY_MAX_TERMS = 2048;                      // per 96-bit sum: 2^11 products below p^2 < 2^61.82 keep hi < 2^9, acc96_div_R's bound
template <int NCOL>
struct BaryRow { uint4 xa, xb; uint32_t w0, w1; };
template <int NCOL>
__device__ __forceinline__ BaryRow<NCOL> bary_load(const uint4* __restrict__ v4, const uint32_t* __restrict__ w32, uint64_t N2, uint64_t j, uint32_t t) {
  BaryRow<NCOL> r;
  r.w0 = w32[4 * j + t]; r.w1 = w32[4 * ((j + N2 - 2) & (N2 - 1)) + t];
  r.xa = v4[j * 2];
  if (NCOL == 8) r.xb = v4[j * 2 + 1];
  return r;
}
template <int NCOL>
__device__ __forceinline__ void bary_mads(const BaryRow<NCOL>& r, bb::Acc96 (*a)[2]) {
  const uint32_t x[8] = {r.xa.x, r.xa.y, r.xa.z, r.xa.w, NCOL == 8 ? r.xb.x : 0, NCOL == 8 ? r.xb.y : 0, NCOL == 8 ? r.xb.z : 0, NCOL == 8 ? r.xb.w : 0};
#pragma unroll
  for (int c = 0; c < NCOL; c++) { bb::mad96(a[c][0], r.w0, x[c]); bb::mad96(a[c][1], r.w1, x[c]); }
}
template <int NCOL>
__device__ __forceinline__ void bary_block(const uint4* __restrict__ v4, const uint32_t* __restrict__ w32, uint64_t N2, uint64_t lo, uint64_t per, uint32_t step, bb::Acc96 (*a)[2]) {
  const uint32_t t = threadIdx.x & 3, rq = threadIdx.x >> 2;
  const uint64_t stride = (uint64_t)step * BARY_ROWS;
  uint64_t j = lo + (uint64_t)step * rq;
  if (per >= 2 * stride) {
    // every lane makes the same per / stride iterations (powers of two): two rows in flight while two are multiplied — the loads of a row are 1-2 us away, its 32
    // multiply-adds ~130 cycles, and the registers hold five waves per SIMD
    const uint32_t n_it = (uint32_t)(per / stride);
    BaryRow<NCOL> r0 = bary_load<NCOL>(v4, w32, N2, j, t), r1 = bary_load<NCOL>(v4, w32, N2, j + stride, t);
    for (uint32_t it = 0; it < n_it; it += 2) {
      const uint64_t jn = it + 2 < n_it ? j + 2 * stride : j;          // (the last pair is loaded twice: nothing is read outside the chunk)
      const BaryRow<NCOL> n0 = bary_load<NCOL>(v4, w32, N2, jn, t), n1 = bary_load<NCOL>(v4, w32, N2, jn + stride, t);
      bary_mads<NCOL>(r0, a); bary_mads<NCOL>(r1, a);
      r0 = n0; r1 = n1; j = jn;
    }
  } else {
    for (; j < lo + per; j += stride) bary_mads<NCOL>(bary_load<NCOL>(v4, w32, N2, j, t), a);
  }
}
__global__ __launch_bounds__(NT, BARY_WAVES) void bary_dot_kernel(const uint32_t* __restrict__ main_m, const uint32_t* __restrict__ aux_m, const uint32_t* __restrict__ quot_m, uint32_t main_blocks,
                                                       uint32_t aux_blocks, uint64_t N2, const E4* __restrict__ wts, E4* __restrict__ partial, uint32_t n_chunks, uint32_t log_per) {
  __shared__ uint32_t red[NT / 64][8][2][4];
  const uint32_t blk = blockIdx.y, chunk = blockIdx.x;
  const bool is_q = blk >= main_blocks + aux_blocks;
  const uint32_t* mat = blk < main_blocks ? main_m + (uint64_t)blk * N2 * 8 : is_q ? quot_m : aux_m + (uint64_t)(blk - main_blocks) * N2 * 8;
  const uint32_t col0 = blk * 8, ncol = is_q ? 4 : 8;                // columns in the order used everywhere: main, aux, the quotient's four
  const uint64_t per = 1ull << log_per, lo = (uint64_t)chunk << log_per;             // N2 / n_chunks, a power of two
  bb::Acc96 a[8][2];
#pragma unroll
  for (int c = 0; c < 8; c++) a[c][0] = a[c][1] = bb::acc96_zero();
  if (is_q) bary_block<4>(reinterpret_cast<const uint4*>(mat), reinterpret_cast<const uint32_t*>(wts), N2, lo, per, 1u, a);
  else bary_block<8>(reinterpret_cast<const uint4*>(mat), reinterpret_cast<const uint32_t*>(wts), N2, lo, per, 2u, a);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 8; c++)
#pragma unroll
    for (int which = 0; which < 2; which++) {
      uint32_t r = bb::acc96_div_R(a[c][which]);
      for (int off = 32; off >= 4; off >>= 1) r = bb::add(r, __shfl_down(r, off, 64));      // lanes of one coordinate: lane & 3 is kept by every step
      if (lane < 4) red[wv][c][which][lane] = r;
    }
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t c = threadIdx.x >> 3, which = (threadIdx.x >> 2) & 1, t = threadIdx.x & 3;
    uint32_t r = red[0][c][which][t];
    for (int k = 1; k < NT / 64; k++) r = bb::add(r, red[k][c][which][t]);
    if (c < ncol) partial[((uint64_t)(col0 + c) * n_chunks + chunk) * 2 + which].c[t] = r;
  }
}

// the chunks' partial sums of a column, added up on the device: 2 (WT + 4) extension elements cross to the host instead of 2 (WT + 4) n_chunks.  One wave per sum:
// lane q takes chunks q, q + 64, .., then the wave adds its lanes up (one thread walking 256 chunks took 87 us).
__global__ __launch_bounds__(NT) void bary_sum_kernel(const E4* __restrict__ partial, uint32_t n_cols2, uint32_t n_chunks, E4* __restrict__ out) {
  const uint32_t idx = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;           // idx = 2 k + which
  if (idx >= n_cols2) return;
  E4 a = bb::e_zero();
  for (uint32_t q = lane; q < n_chunks; q += 64) a = bb::e_add(a, partial[((uint64_t)(idx >> 1) * n_chunks + q) * 2 + (idx & 1)]);
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int t = 0; t < 4; t++) a.c[t] = bb::add(a.c[t], __shfl_down(a.c[t], off, 64));
  if (lane == 0) out[idx] = a;
}

// ---- DEEP codeword: F(x) = (A(x) - a0)/(x - zeta) + (B(x) - b0)/(x - zeta w) ------------------------------------------------
// A = Σ gamma^k v_k, B = Σ gamma^(WT + k) v_k over the columns of a position (Montgomery words x Montgomery gamma powers): exact 96-bit sums
// per extension coordinate (bb::mad96_s, gamma^k in scalar registers), reduced once — the reduction's division by R leaves R A, R B.
__global__ __launch_bounds__(NT, DEEP_WAVES) void deep_kernel(const uint32_t* __restrict__ L, const uint32_t* __restrict__ AL, const uint32_t* __restrict__ Q, uint32_t log_n,
                                                   const E4* __restrict__ dinv, const ProveParams* __restrict__ pp, uint32_t wn_inv_m, int WM, int WA, uint32_t* __restrict__ cw) {
  const uint32_t N2 = 2u << log_n;
  const int WT = WM + WA;                                      // WM / WA = the proof's committed main-trace / aux widths (multiples of 8)
  const uint32_t j = blockIdx.x * NT + threadIdx.x;
  if (j >= N2) return;
  bb::Acc96 A[4], B[4];
#pragma unroll
  for (int t = 0; t < 4; t++) A[t] = B[t] = bb::acc96_zero();
  constexpr int UN = 8;
  static_assert(WMX % UN == 0 && air::W_COMMITTED_DEFAULT % UN == 0 && air::W_COMMITTED_IO % UN == 0 && air::W_AUX % UN == 0 && WAX % UN == 0 && WMX + WAX + 4 < 512, "column loop; each 96-bit sum (A: WT + 4 terms, B: WT) holds 2^9 terms");
  auto block = [&](const uint4* M4, int blk, int k0) {                         // one B8 block = eight columns, gamma indices k0 .. k0 + 7 (zeta) and WT + k0 .. (zeta w)
    const uint4 vlo = M4[((uint64_t)blk * N2 + j) * 2], vhi = M4[((uint64_t)blk * N2 + j) * 2 + 1];
    const uint32_t v[UN] = {vlo.x, vlo.y, vlo.z, vlo.w, vhi.x, vhi.y, vhi.z, vhi.w};
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const E4 ga = pp->gamma_pow[k0 + u], gb = pp->gamma_pow[WT + k0 + u];
#pragma unroll
      for (int t = 0; t < 4; t++) { bb::mad96_s(A[t], ga.c[t], v[u]); bb::mad96_s(B[t], gb.c[t], v[u]); }
    }
  };
#pragma unroll 1
  for (int k = 0; k < WM; k += UN) block(reinterpret_cast<const uint4*>(L), k >> 3, k);
#pragma unroll 1
  for (int k = 0; k < WA; k += UN) block(reinterpret_cast<const uint4*>(AL), k >> 3, WM + k);
  {
    const uint4 qv = reinterpret_cast<const uint4*>(Q)[(uint64_t)j * 2];
    const uint32_t v[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const E4 ga = pp->gamma_pow[2 * WT + u];
#pragma unroll
      for (int t = 0; t < 4; t++) bb::mad96_s(A[t], ga.c[t], v[u]);
    }
  }
  const E4 Am{{bb::acc96_div_R(A[0]), bb::acc96_div_R(A[1]), bb::acc96_div_R(A[2]), bb::acc96_div_R(A[3])}};
  const E4 Bm{{bb::acc96_div_R(B[0]), bb::acc96_div_R(B[1]), bb::acc96_div_R(B[2]), bb::acc96_div_R(B[3])}};
  const E4 i1 = dinv[j], i2 = bb::e_mul_fm(dinv[(j + N2 - 2) & (N2 - 1)], wn_inv_m);   // 1/(zeta - x), 1/(zeta w - x)
  const E4 t1 = bb::e_mul_m(bb::e_sub(pp->a0, Am), i1);                       // (A - a0)/(x - zeta) = (a0 - A)/(zeta - x)
  const E4 t2 = bb::e_mul_m(bb::e_sub(pp->b0, Bm), i2);
  const E4 f = bb::e_from_mont(bb::e_add(t1, t2));
#pragma unroll
  for (int i = 0; i < 4; i++) cw[(uint64_t)i * N2 + j] = f.c[i];
}

// ---- FRI: leaf i of a layer of size m folded k times = (c[i + t g])_{t < 2^k}, g = m >> k: 4 * 2^k base elements absorbed
// eight at a time (two extension values per permutation), as so::hash_elems does ------------------------------------------------
__global__ __launch_bounds__(NT) void fri_leaf_hash_kernel(const p2::Consts* __restrict__ cp, const uint32_t* __restrict__ c, uint64_t m, uint32_t k, uint32_t* __restrict__ digests) {
  const uint64_t g = m >> k, i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= g) return;
  uint32_t s[p2::T];
#pragma unroll
  for (int t = 0; t < p2::T; t++) s[t] = 0;
  const uint32_t in_scale = cp->in_scale, carry = cp->carry, ko = cp->out_scale;
  for (uint32_t u = 0; u < (1u << k); u += 2) {
#pragma unroll
    for (int t = 0; t < 4; t++) { s[t] = bb::mont_mul_lazy(c[(uint64_t)t * m + i + u * g], in_scale); s[4 + t] = bb::mont_mul_lazy(c[(uint64_t)t * m + i + (u + 1) * g], in_scale); }
#pragma unroll
    for (int t = p2::RATE; t < p2::T; t++) s[t] = bb::mont_mul_lazy(s[t], carry);
    p2::permute_scaled(s, *cp);
  }
  reinterpret_cast<uint4*>(digests)[i] = make_uint4(bb::mont_mul(s[0], ko), bb::mont_mul(s[1], ko), bb::mont_mul(s[2], ko), bb::mont_mul(s[3], ko));
}

// the same leaf hash with one permutation per quad of lanes (p2::permute_quad): small layers are latency-bound
__global__ __launch_bounds__(NT) void fri_leaf_hash_quad_kernel(const p2::Consts* __restrict__ cp, const uint32_t* __restrict__ c, uint64_t m, uint32_t k, uint32_t* __restrict__ digests) {
  const uint64_t g = m >> k, t = (uint64_t)blockIdx.x * NT + threadIdx.x, i = t >> 2;
  const int l = (int)(t & 3);
  if (i >= g) return;                                                         // whole quads leave together
  uint32_t s[3] = {0, 0, 0};
  const uint32_t k_in = cp->in_scale, carry = cp->carry;
  for (uint32_t u = 0; u < (1u << k); u += 2) {
    s[0] = bb::mont_mul_lazy(c[(uint64_t)l * m + i + u * g], k_in); s[1] = bb::mont_mul_lazy(c[(uint64_t)l * m + i + (u + 1) * g], k_in);
    s[2] = bb::mont_mul_lazy(s[2], carry);                                     // the capacity words stay: output factor -> input factor
    p2::permute_quad_scaled(s, l, *cp);
  }
  digests[4 * i + l] = bb::mont_mul(s[0], cp->out_scale);
}

// .. and with one permutation per ROW of 16 lanes (p2::permute_row16_scaled: the shortest chain) for the smallest layers, whose few leaves wait for each other's 2^k / 2 permutations
__global__ __launch_bounds__(NT) void fri_leaf_hash_row16_kernel(const p2::Consts* __restrict__ cp, const uint32_t* __restrict__ c, uint64_t m, uint32_t k, uint32_t* __restrict__ digests) {
  const uint64_t g = m >> k, t = (uint64_t)blockIdx.x * NT + threadIdx.x, i = t >> 4;
  const int l = (int)(t & 15);
  if (i >= g) return;                                                         // whole rows leave together
  uint32_t s = 0;
  const uint32_t k_in = cp->in_scale, carry = cp->carry;
  for (uint32_t u = 0; u < (1u << k); u += 2) {                               // words 0-3: the coordinates of value u, 4-7: of value u + 1, 8-11: the capacity (stays: output factor -> input factor)
    if (l < 4) s = bb::mont_mul_lazy(c[(uint64_t)l * m + i + u * g], k_in);
    else if (l < 8) s = bb::mont_mul_lazy(c[(uint64_t)(l - 4) * m + i + (u + 1) * g], k_in);
    else s = bb::mont_mul_lazy(s, carry);
    s = p2::permute_row16_scaled(s, l, *cp);
  }
  if (l < 4) digests[4 * i + l] = bb::mont_mul(s, cp->out_scale);
}

// one binary fold of a layer of m values: c'[i] = (c[i] + c[i+h])/2 + beta (c[i] - c[i+h]) / (2 x_i),  x_i = shift * w_m^i,  h = m / 2;
// w_m^-i = w_2N^-(i << (log_2n - log_m)), and w^-k = -w^(N-k) for 0 < k < N (w^N = -1): the table holds w^k for k < N = 2^(log_2n - 1)
// The K binary folds between two committed layers in ONE launch (round 4: three launches and two intermediate layers before): output i of the last fold depends on the 2^K
// values c[i + u g], g = m >> K, and fold f pairs v[u] with v[u + 2^(K-f-1)] — exactly the pairs (j, j + h_f) of a binary fold of the size-(m >> f) layer, the same field operations in
// the same order on every value, so the layer is the one the chain of binary folds leaves.
// beta_m: the layer's challenge and its squares, beta^(2^f) (Montgomery), written by fri_transcript_kernel on the device (round 5: no host round trip between the layers)
struct FoldParams { uint32_t half_shift_inv_m[3]; };
template <int K>
__global__ __launch_bounds__(NT) void fri_fold_k_kernel(const uint32_t* __restrict__ c, uint32_t log_m, uint32_t log_2n, const uint32_t* __restrict__ tw_fwd, FoldParams fp, const E4* __restrict__ beta_m, uint32_t* __restrict__ out) {
  const uint64_t m = 1ull << log_m, g = m >> K, i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= g) return;
  E4 v[1 << K];
#pragma unroll
  for (int u = 0; u < (1 << K); u++)
#pragma unroll
    for (int t = 0; t < 4; t++) v[u].c[t] = c[(uint64_t)t * m + i + (uint64_t)u * g];
  const uint32_t N = 1u << (log_2n - 1);
  constexpr uint32_t HALF_M = (uint32_t)(((uint64_t)((bb::P + 1) / 2) * bb::R1) % bb::P);
#pragma unroll
  for (int f = 0; f < K; f++) {
    const int cnt = 1 << (K - f - 1);
#pragma unroll
    for (int u = 0; u < cnt; u++) {
      const uint64_t j = i + (uint64_t)u * g;                                  // position in the half of the size-(m >> f) domain
      const uint32_t k = (uint32_t)(j << (log_2n - (log_m - f)));
      const uint32_t winv = k == 0 ? bb::R1 : bb::neg(tw_fwd[N - k]);
      const uint32_t inv2x = bb::mont_mul(winv, fp.half_shift_inv_m[f]);
      const E4 a = v[u], b = v[u + cnt];
      const E4 sum = bb::e_mul_fm(bb::e_add(a, b), HALF_M);
      const E4 dif = bb::e_mul_fm(bb::e_sub(a, b), inv2x);
      const E4 prod = bb::e_from_mont(bb::e_mul_m(bb::e_to_mont(dif), beta_m[f]));
      v[u] = bb::e_add(sum, prod);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; t++) out[(uint64_t)t * g + i] = v[0].c[t];
}

// ---- the Fiat-Shamir step of one FRI layer ON THE DEVICE: observe the layer's root, sample beta -------------------------------------------
// Challenger (below; so::Challenger) at this point of the transcript: nothing pending, so  observe(root[0..4)); sample_ext()  is ONE duplex — state words 0..3 overwritten
// with the root, one permutation, beta = (st[7], st[6], st[5], st[4]) (sample() pops from the back of the eight rate words).  One row of lanes runs it
// (p2::permute_row16_scaled: one row of 16 lanes, lane l holds word l) on the challenger state the host uploaded (Montgomery words, as Challenger::st), leaves the state for the next
// layer, and writes  out[0..4) = the root, out[4..16) = beta, beta^2, beta^4 (Montgomery: what fri_fold_k_kernel multiplies by).  The host replays the same steps from
// the roots once the whole commit phase is enqueued (the proof needs them anyway) and refuses to go on if its betas are not the device's.
__global__ void fri_transcript_kernel(const p2::Consts* __restrict__ cp, uint32_t* __restrict__ st, const uint32_t* __restrict__ root, uint32_t* __restrict__ out) {
  __shared__ uint32_t w[p2::T];
  const int l = (int)(threadIdx.x & 15);                                      // one row of 16 lanes (p2::permute_row16_scaled): lane l < 12 holds state word l
  const uint32_t k_in = bb::from_mont(cp->in_scale), ko_m = bb::to_mont(cp->out_scale);
  const uint32_t r = l < 4 ? root[l] : 0u;
  const uint32_t xm = l < 4 ? bb::to_mont(r) : l < p2::T ? st[l] : 0u;          // words 0-3 overwritten with the root, the rest of the state stays
  uint32_t s = p2::permute_row16_scaled(bb::mont_mul_lazy(xm, k_in), l, *cp);
  if (l < p2::T) { const uint32_t v = bb::mont_mul(s, ko_m); st[l] = v; w[l] = v; }
  if (l < 4) out[l] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    E4 beta{{w[7], w[6], w[5], w[4]}};
    E4* bo = reinterpret_cast<E4*>(out + 4);
    for (int f = 0; f < 3; f++) { bo[f] = beta; beta = bb::e_mul_m(beta, beta); }
  }
}

// ---- query gathering -------------------------------------------------------------------------------------------------------------
// A job (one workgroup of 64 lanes) is one of three shapes (round 5: ~750 jobs a proof instead of ~15,000 one-line copies — building and uploading that list was 90 us of host time with the GPU idle):
//   kind 0: `count` words src[k * stride] -> dst[k]                                    (a FRI leaf: the four coordinates of a value, one column apart)
//   kind 1: `count` ROW PIECES of `width` words, src[b * stride + w] -> dst[b * width + w]  (a matrix row: 8 words out of each B8 block; the quotient's 4)
//   kind 2: a MERKLE PATH — tree = src, `stride` = its number of leaves, `count` = the leaf index: the sibling digest (4 words) of every level, leaf level first
// mont: the source words rest in Montgomery form (rows of the LDE matrices) and enter the proof canonical
struct GatherJob { const uint32_t* src; uint64_t stride; uint32_t count; uint32_t dst; uint32_t mont; uint32_t kind_width; };   // kind_width = kind | width << 8
__global__ void gather_kernel(const GatherJob* __restrict__ jobs, uint32_t n_jobs, uint32_t* __restrict__ dst) {
  const uint32_t jb = blockIdx.x;
  if (jb >= n_jobs) return;
  const GatherJob g = jobs[jb];
  const uint32_t kind = g.kind_width & 0xFF, width = g.kind_width >> 8;
  if (kind == 0) {
    for (uint32_t k = threadIdx.x; k < g.count; k += blockDim.x) { const uint32_t v = g.src[(uint64_t)k * g.stride]; dst[g.dst + k] = g.mont ? bb::from_mont(v) : v; }
  } else if (kind == 1) {
    for (uint32_t k = threadIdx.x; k < g.count * width; k += blockDim.x) { const uint32_t v = g.src[(uint64_t)(k / width) * g.stride + k % width]; dst[g.dst + k] = g.mont ? bb::from_mont(v) : v; }
  } else {
    uint32_t depth = 0;
    for (uint64_t q = g.stride; q > 1; q >>= 1) depth++;
    for (uint32_t k = threadIdx.x; k < 4 * depth; k += blockDim.x) {
      const uint32_t lvl = k >> 2;                                             // level `lvl` starts 4 * (n + n/2 + .. ) = 4 * (2 n - (2 n >> lvl)) words into the tree
      const uint64_t n = g.stride, at = 4 * (2 * n - ((2 * n) >> lvl)), sib = ((uint64_t)g.count >> lvl) ^ 1;
      dst[g.dst + k] = g.src[at + 4 * sib + (k & 3)];
    }
  }
}

// ---- proof-of-work grinding: one candidate nonce per lane; the smallest hit wins (deterministic) ------------------------------
__global__ __launch_bounds__(NT) void pow_grind_kernel(const p2::Consts* __restrict__ cp, const uint32_t* __restrict__ state_m, uint32_t base, uint32_t pow_bits, uint32_t* __restrict__ best) {
  const uint32_t nonce = base + blockIdx.x * NT + threadIdx.x;
  if (nonce >= bb::P) return;
  uint32_t s[p2::T];
  const uint32_t k_in = bb::from_mont(cp->in_scale);                          // Montgomery words R v -> input words F_IN v of the throughput formulation (a fifth fewer instructions)
#pragma unroll
  for (int i = 1; i < p2::T; i++) s[i] = bb::mont_mul_lazy(state_m[i], k_in);
  s[0] = bb::mont_mul_lazy(nonce, cp->in_scale);                              // overwrite-absorb of the single pending element (canonical)
  p2::permute_scaled(s, *cp);
  if ((bb::mont_mul(s[p2::RATE - 1], cp->out_scale) & ((1u << pow_bits) - 1)) == 0) atomicMin(best, nonce);   // sample() takes the last rate element
}

// ---- host-side duplex challenger (Montgomery state; canonical in/out) -------------------------------------------------------
struct Challenger {
  const p2::Consts& c;
  uint32_t st[p2::T];
  std::vector<uint32_t> in, out;
  explicit Challenger(const p2::Consts& cc) : c(cc) { for (auto& v : st) v = 0; }
  // (round 5) the permutation in the hash kernels' formulation (p2::permute_scaled: a quarter fewer host instructions — a proof's transcript is ~330 permutations, all of
  // them on the critical path with the GPU idle); the state stays what it was: Montgomery words R v, canonical range
  void duplex() {
    for (size_t i = 0; i < in.size(); i++) st[i] = bb::to_mont(in[i]);
    in.clear();
    const uint32_t k_in = bb::from_mont(c.in_scale), ko_m = bb::to_mont(c.out_scale);
    uint32_t s[p2::T];
    for (int i = 0; i < p2::T; i++) s[i] = bb::mont_mul_lazy(st[i], k_in);
    p2::permute_scaled(s, c);
    for (int i = 0; i < p2::T; i++) st[i] = bb::mont_mul(s[i], ko_m);
    out.clear();
    for (int i = 0; i < p2::RATE; i++) out.push_back(bb::from_mont(st[i]));
  }
  void observe(uint32_t x) { out.clear(); in.push_back(x); if ((int)in.size() == p2::RATE) duplex(); }
  void observe_n(const uint32_t* x, size_t n) { for (size_t i = 0; i < n; i++) observe(x[i]); }
  uint32_t sample() { if (!in.empty() || out.empty()) duplex(); const uint32_t v = out.back(); out.pop_back(); return v; }
  E4 sample_ext() { E4 e; for (int i = 0; i < 4; i++) e.c[i] = sample(); return e; }
  uint32_t sample_bits(int b) { return sample() & ((1u << b) - 1); }
  void flush() { if (!in.empty()) duplex(); out.clear(); }                    // so::Challenger::flush
  bool check_pow(uint32_t nonce, int pow_bits) { flush(); observe(nonce); return (sample() & ((1u << pow_bits) - 1)) == 0; }
};

// so::hash_elems on the host (the overwrite sponge of the section digests)
void hash_elems_host(const p2::Consts& c, const uint32_t* in, size_t n, uint32_t out[4]) {
  uint32_t st[p2::T] = {0};
  for (size_t off = 0; off < n; off += p2::RATE) {
    const size_t len = n - off < (size_t)p2::RATE ? n - off : (size_t)p2::RATE;
    for (size_t i = 0; i < len; i++) st[i] = bb::to_mont(in[off + i]);
    p2::permute(st, c);
  }
  if (n == 0) p2::permute(st, c);
  for (int i = 0; i < 4; i++) out[i] = bb::from_mont(st[i]);
}

E4 h_e_mul(const E4& a, const E4& b) { return bb::e_from_mont(bb::e_mul_m(bb::e_to_mont(a), bb::e_to_mont(b))); }   // canonical in/out
E4 h_e_pow(E4 a, uint64_t e) { E4 r{{1, 0, 0, 0}}; while (e) { if (e & 1) r = h_e_mul(r, a); a = h_e_mul(a, a); e >>= 1; } return r; }

#define HIP_OK(expr)                                                                                         \
  do { hipError_t _e = (expr); if (_e != hipSuccess) { zkir::set_last_error({ZKIR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)}); return ZKIR_ERR_DEVICE; } } while (0)

// Bump allocation out of the context's persistent workspace (reset at the start of every zkir_prove; the context's mutex is held).
struct Arena {
  const zkir_stark_ctx* c;
  template <typename T> hipError_t take(T** out, size_t count) {
    const size_t need = (count * sizeof(T) + 255) & ~(size_t)255;
    if (c->arena_off + need > c->arena_size) return hipErrorOutOfMemory;
    *out = (T*)(c->arena + c->arena_off);
    c->arena_off += need;
    return hipSuccess;
  }
};
struct StageEvents {             // RAII: released on every return path
  hipEvent_t ev[10]; bool on = false;
  hipError_t create() { for (auto& e : ev) { const hipError_t r = hipEventCreate(&e); if (r != hipSuccess) return r; } on = true; return hipSuccess; }
  ~StageEvents() { if (on) for (auto& e : ev) (void)hipEventDestroy(e); }
};

// the header words of a v4 proof (so::header_words): parameters, public inputs, the two boundary states read off the main trace
void header_words(uint32_t log_n, const zkir_public_inputs& pub, const uint32_t* states, std::vector<uint32_t>& w) {      // states: 2 NS words (+ 4 counters in mode 2)
  w.clear();
  w.insert(w.end(), {PROOF_MAGIC, air::proof_version((int)pub.deferred), log_n, (uint32_t)air::committed_width((int)pub.deferred), (uint32_t)air::num_queries_of(pub.fri_params), (uint32_t)LOG_FINAL,
                     (uint32_t)air::pow_bits_of(pub.fri_params)});
  w.insert(w.end(), {(uint32_t)(pub.n_real & 0x3FFFFFFF), (uint32_t)(pub.n_real >> 30), pub.deferred});
  w.insert(w.end(), {(uint32_t)(pub.entry_point & 0xFFFFF), (uint32_t)((pub.entry_point >> 20) & 0xFFFFF), (uint32_t)(pub.entry_point >> 40)});
  w.insert(w.end(), pub.program_digest, pub.program_digest + 4);
  w.insert(w.end(), pub.io_digest, pub.io_digest + 4);
  w.insert(w.end(), states, states + 2 * NS + (pub.deferred >= 2 ? 4 : 0));
}

// The quotient kernel's evaluation of the constraint list (QuotientOps: lazy 32-bit arithmetic, 96-bit sums, one accumulator per selector), run on
// the HOST for one (row, next row) pair given as LOGICAL columns (canonical words): out4 = sum_c alpha^c C_c, canonical — what the oracle's
// constraints_sum gives for the same inputs (tests/test_abi.py).  A test entry point; nothing in the product calls it.
struct HostRowSrc {
  const uint32_t* l; const uint32_t* n; const uint32_t* a; const uint32_t* an;
  uint32_t loc(int p) const { return l[p]; }
  uint32_t nxt(int p) const { return n[p]; }
  uint32_t loc_r(int p) const { return l[p]; }
  uint32_t nxt_r(int p) const { return n[p]; }
  uint32_t aloc(int p) const { return a[p]; }
  uint32_t anxt(int p) const { return an[p]; }
};
template <int DEF /* the mode */>
static void air_eval_host(const uint32_t* loc, const uint32_t* nxt, const uint32_t* aloc, const uint32_t* anxt, const uint32_t* lk, const uint32_t* sel3, const uint32_t* first, const uint32_t* last,
                          const uint32_t* cnt4, const uint32_t* alpha4, uint32_t* out4) {
  uint32_t l[WMX], n[WMX], a[WAX], an[WAX], lkm[air::N_LK], fm[NS], lm[NS], cm[4] = {0, 0, 0, 0};
  for (int p = 0; p < air::committed_used(DEF); p++) { l[p] = bb::to_mont(loc[air::logical_col(p, DEF)]); n[p] = bb::to_mont(nxt[air::logical_col(p, DEF)]); }
  for (int k = 0; k < air::aux_width(DEF); k++) { a[k] = bb::to_mont(aloc[k]); an[k] = bb::to_mont(anxt[k]); }
  for (int i = 0; i < air::N_LK; i++) lkm[i] = (DEF >= 2 || i < air::LK_NIN) ? bb::to_mont(lk[i]) : 0u;       // (modes 0 / 1: the caller's lk has 56 words)
  for (int i = 0; i < NS; i++) { fm[i] = bb::to_mont(first[i]); lm[i] = bb::to_mont(last[i]); }
  if (DEF >= 2) for (int k = 0; k < 4; k++) cm[k] = bb::to_mont(cnt4[k]);
  std::vector<E4> ap(N_CONSTRAINTS);
  E4 al{{alpha4[0], alpha4[1], alpha4[2], alpha4[3]}}, cur{{1, 0, 0, 0}};
  for (int c = 0; c < N_CONSTRAINTS; c++) { ap[c] = bb::e_to_mont(cur); cur = h_e_mul(cur, al); }
  int order[N_CONSTRAINTS];
  const int n_push = air::push_order(DEF, order);
  std::vector<E4> seq((size_t)N_CONSTRAINTS + 2, bb::e_zero());
  for (int k = 0; k < n_push; k++) seq[k] = ap[order[k]];
  QuotientOps<DEF, HostRowSrc> o{HostRowSrc{l, n, a, an}, seq.data(), {}, lkm};
  o.init();
  air::eval(o, fm, lm, DEF, cm);
  E4 cf, cl;
  air::boundary_constants(ap.data(), fm, lm, cf, cl, DEF >= 2 ? cm : nullptr);
  using QO = QuotientOps<DEF, HostRowSrc>;
  const E4 s0 = QO::sum_of(o.a0), st = o.st, sf = bb::e_sub(o.sf, cf), sl = bb::e_sub(o.sl, cl);
  const E4 tot = bb::e_add(bb::e_add(s0, bb::e_mul_fm(st, bb::to_mont(sel3[2]))), bb::e_add(bb::e_mul_fm(sf, bb::to_mont(sel3[0])), bb::e_mul_fm(sl, bb::to_mont(sel3[1]))));
  const E4 r = bb::e_from_mont(tot);
  for (int t = 0; t < 4; t++) out4[t] = r.c[t];
}
}  // namespace

extern "C" {

void zkir_air_eval_host(const uint32_t* loc, const uint32_t* nxt, const uint32_t* aloc, const uint32_t* anxt, const uint32_t* lk, uint32_t is_first, uint32_t is_last, uint32_t is_trans,
                        const uint32_t* first68, const uint32_t* last68, const uint32_t* alpha4, uint32_t mode, const uint32_t* cnt4, uint32_t* out4) {
  const uint32_t sel3[3] = {is_first, is_last, is_trans};
  if (mode == 4) air_eval_host<4>(loc, nxt, aloc, anxt, lk, sel3, first68, last68, cnt4, alpha4, out4);
  else if (mode == 3) air_eval_host<3>(loc, nxt, aloc, anxt, lk, sel3, first68, last68, cnt4, alpha4, out4);
  else if (mode == 2) air_eval_host<2>(loc, nxt, aloc, anxt, lk, sel3, first68, last68, cnt4, alpha4, out4);
  else if (mode == 1) air_eval_host<1>(loc, nxt, aloc, anxt, lk, sel3, first68, last68, cnt4, alpha4, out4);
  else air_eval_host<0>(loc, nxt, aloc, anxt, lk, sel3, first68, last68, cnt4, alpha4, out4);
}

uint32_t zkir_proof_num_queries(void) { return (uint32_t)air::DEFAULT_NUM_QUERIES; }
uint32_t zkir_proof_version(void) { return air::proof_version(0); }
uint32_t zkir_proof_version_of_mode(uint32_t mode) { return air::proof_version((int)mode); }

void zkir_proof_free(uint32_t* proof) { free(proof); }

// Full proof of the run whose K1 output is `trace` (pub->n_real executed rows, padded to 2^ctx.log_n).  Phases are timed with HIP events
// when stage_ms != NULL (NINE floats): [0] main trace (+ lookup indices), [1] LDE, [2] trace Merkle, [3] lookup argument (inverse tables, aux trace, its LDE and
// Merkle tree), [4] quotient (+Merkle), [5] openings, [6] DEEP, [7] FRI (+grinding), [8] queries.
int zkir_prove(const zkir_stark_ctx* c, const zkir_trace_columns* trace, const zkir_public_inputs* pub, uint32_t** proof_out, uint64_t* proof_words, float* stage_ms,
               void* stream) {
  if (!c || !trace || !pub || !proof_out || !proof_words) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: null argument"}); return ZKIR_ERR_ARGUMENT; }
  const bool dbg_t = getenv("ZKIR_PROVE_TIMES") != nullptr;      // diagnostics: host wall time of the call's phases on stderr
  const auto t_entry = std::chrono::steady_clock::now();
  auto since = [&](const std::chrono::steady_clock::time_point& t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  const uint32_t log_n = c->log_n;
  const uint64_t N = 1ull << log_n, N2 = 2 * N;
  if (pub->deferred > 4) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: pub->deferred is the proof's mode: 0 default, 1 deferred model, 2 default + the I/O argument, 3 = 2 + the memory argument, 4 = 3 + MULH / DIVU / REMU / DIV / REM"}); return ZKIR_ERR_ARGUMENT; }
  const int MODE = (int)pub->deferred;                         // 0 default, 1 deferred carry model, 2 default + the I/O argument, 3 = 2 + the memory argument (round 4), 4 = 3 + the wide-arithmetic class (round 6)
  const bool IO = MODE >= 2, MEM = MODE >= 3, WIDE = MODE == 4;
  if (!air::fri_params_ok(pub->fri_params)) {
    zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: pub->fri_params (num_queries | pow_bits << 16, 0 = 50 queries + 12 bits) must name 50..128 queries and 12..24 grinding bits (zkir_public_inputs_set_params)"});
    return ZKIR_ERR_ARGUMENT;
  }
  const int NUM_QUERIES = air::num_queries_of(pub->fri_params), POW_BITS = air::pow_bits_of(pub->fri_params);     // this proof's (header words 4 and 6)
  const int WM = air::committed_width(MODE), WA = air::aux_width(MODE), WT = WM + WA;     // this proof's committed main-trace / aux columns
  // (mode 3) the memory witness: computed HERE on the device (memcheck.hip) unless the caller brings one (pub->mem_old != NULL: zkir_memcheck_witness_of's host replay — the
  // independent implementation the tests compare with — or a forged one)
  const bool MEM_HOST = MEM && pub->mem_old != nullptr;
  if (MEM && (pub->writes_before || pub->reads_before || (MEM_HOST && (!pub->mem_told || (pub->n_cells && (!pub->cell_addr || !pub->cell_bytes || !pub->cell_time)) || pub->n_cells >= (1u << 28))))) {
    zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: mode 3 proves a WHOLE run; a memory witness passed in the public inputs must be complete (zkir_memcheck_witness_of + zkir_public_inputs_set_memory)"});
    return ZKIR_ERR_ARGUMENT;
  }
  if (IO && ((!pub->inputs && pub->n_inputs) || (!pub->outputs && pub->n_outputs) || pub->n_inputs >= (1u << 28) || pub->n_outputs >= (1u << 28) || pub->halt_kind > 2)) {
    zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: mode 2 needs the I/O tapes and the halt reason in the public inputs (zkir_public_inputs_of fills them)"});
    return ZKIR_ERR_ARGUMENT;
  }
  if (pub->n_real == 0 || zkir_padded_log_n(pub->n_real) != log_n || pub->entry_point >= (1ull << 40)) {
    zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: the context's log_n must be zkir_padded_log_n(n_real) (n_real >= 1), and entry_point < 2^40"});
    return ZKIR_ERR_ARGUMENT;
  }
  for (int i = 0; i < 4; i++)
    if (pub->program_digest[i] >= bb::P || pub->io_digest[i] >= bb::P) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: digests must be canonical field elements"}); return ZKIR_ERR_ARGUMENT; }
  // the program: its code words are the instruction ROM every row is looked up in (Program::to_bytes layout, program.rs:170-214)
  const uint8_t* blob = pub->program_blob;
  const uint64_t blob_len = pub->program_blob_len;
  uint32_t n_code = 0;
  {
    auto le32 = [&](size_t at) { return (uint32_t)blob[at] | ((uint32_t)blob[at + 1] << 8) | ((uint32_t)blob[at + 2] << 16) | ((uint32_t)blob[at + 3] << 24); };
    bool ok = blob && blob_len >= 32 && blob_len <= (1ull << 30);
    if (ok) { const uint64_t code_size = le32(16); ok = code_size % 4 == 0 && 32 + code_size <= blob_len && le32(12) == pub->entry_point; n_code = (uint32_t)(code_size / 4); }
    if (ok) { uint32_t dg[4]; zkir_digest_bytes(blob, blob_len, dg); ok = !memcmp(dg, pub->program_digest, 16); }
    if (!ok) {
      zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: pub->program_blob must be the program that ran (Program::to_bytes layout; its digest and entry point are the public inputs'): "
                                               "use zkir_public_inputs_of, and keep the blob alive while proving"});
      return ZKIR_ERR_ARGUMENT;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  // One proof at a time per context (its workspace arena); proofs on different contexts run concurrently: the per-proof constants
  // live in the arena (no __constant__ / static state).
  std::lock_guard<std::mutex> ctx_lock(c->mu);
  *proof_out = nullptr; *proof_words = 0;
  const std::vector<int> ks = fri_schedule((int)log_n);
  const int n_layers = (int)ks.size();

  // (mode 4) a run with hash calls can touch more cells than it has rows (a 5000-byte BLAKE3 input is 626 cells): the cell arrays and the section buffer are sized by the
  // caller's witness when it brings one (the device witness sees loads and stores only: at most one cell per row)
  const uint64_t CELL_CAP = std::max<uint64_t>(N, MEM_HOST ? pub->n_cells + 1 : 0);
  const size_t SEC_WORDS = std::max<size_t>(8 * (size_t)CELL_CAP, WIDE ? (size_t)pub->hash_section_words + (size_t)pub->hash_section_words / 100 : 0) + 4096;   // (the memory section, then the hash section, each with its chunk digests behind it)
  {                                               // workspace: 12 W (M + L) + 440 (trees, quotient, weights, FRI) bytes per row, allocated once per context
    static_assert(WMX % 8 == 0 && air::W_COMMITTED_DEFAULT % 8 == 0, "the main trace fills whole B8 blocks");
    const size_t want = (size_t)(12 * WM + 12 * WA + 16 + 544 + (IO ? 48 : 0) + (MEM ? 48 : 0) + (WIDE ? 8 : 0)) * N + (size_t)air::MAX_NUM_QUERIES * 64 * 1024 + (8u << 20) + (size_t)n_code * 24 + (1u << 16) +
                        (IO ? (size_t)pub->n_inputs * 8 : 0) + (MEM ? (size_t)air::MEM_MULT * 24 + (size_t)CELL_CAP * 28 + SEC_WORDS * 4 + blob_len + (1u << 16) : 0);
    if (c->arena_size < want) {
      if (c->arena) (void)hipFree(c->arena);
      c->arena = nullptr; c->arena_size = 0;
      HIP_OK(hipMalloc((void**)&c->arena, want));
      c->arena_size = want;
    }
    c->arena_off = 0;
  }
  Arena ar{c};
  zkir::HostPin& pin = c->pin;                                 // every host block of unbounded size crosses through pinned staging (host.h)
  pin.reset();
  auto h2d = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    void* p = pin.take(bytes);
    if (!p) return hipErrorOutOfMemory;
    memcpy(p, src, bytes);
    return hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, s);
  };
  // small results the host waits for (roots, boundary states, partial sums ..): copied into PINNED staging — a copy into a pageable block is served synchronously, one
  // host round trip per copy (the kernel timeline showed 13-29 us between consecutive 16-byte copies) — and handed to their destinations after the ONE synchronisation that follows them
  struct Staged { void* dst; const void* src; size_t n; };
  std::vector<Staged> staged;
  auto d2h = [&](void* dst, const void* dsrc, size_t bytes) -> hipError_t {
    void* h = pin.take(bytes);
    if (!h) return hipErrorOutOfMemory;
    staged.push_back({dst, h, bytes});
    return hipMemcpyAsync(h, dsrc, bytes, hipMemcpyDeviceToHost, s);
  };
  auto sync_d2h = [&]() -> hipError_t {
    const hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) for (const Staged& x : staged) memcpy(x.dst, x.src, x.n);
    staged.clear();
    return e;
  };
  uint32_t *dM, *dL, *dTree, *dQ, *dQTree, *dState, *dBest, *dBound, *dA, *dAL, *dATree, *dCode, *dMult;
  unsigned long long* dBad;
  uint4 *dSide, *dSums;
  E4 *dW, *dDinv, *dPart, *dPartSum, *dInvRc, *dInvRom;
  ProveParams* dPP;
  IoEntry* dIo = nullptr; uint32_t* dIoCount = nullptr; uint64_t* dInputs = nullptr; uint32_t* dIoScratch = nullptr;
  uint64_t* dMemOld = nullptr; uint32_t* dMemTold = nullptr; uint4* dMemSide = nullptr; E4* dInvMem = nullptr; uint2* dWideSide = nullptr;
  uint64_t *dCellAddr = nullptr, *dCellBytes = nullptr; uint32_t* dCellTime = nullptr; uint8_t* dImage = nullptr; E4* dCellPart = nullptr; uint32_t* dSec = nullptr;
  HIP_OK(ar.take(&dPP, 1)); HIP_OK(ar.take(&dState, 16)); HIP_OK(ar.take(&dBest, 4)); HIP_OK(ar.take(&dBound, 2 * NS + 4)); HIP_OK(ar.take(&dBad, 1));
  if (IO) { HIP_OK(ar.take(&dIo, N)); HIP_OK(ar.take(&dIoCount, 4)); HIP_OK(ar.take(&dInputs, (size_t)pub->n_inputs + 1)); HIP_OK(ar.take(&dIoScratch, 2 * N + 2 * (N / 1024 + 2))); }
  if (MEM) {
    HIP_OK(ar.take(&dMemOld, N)); HIP_OK(ar.take(&dMemTold, N)); HIP_OK(ar.take(&dMemSide, 2 * N)); HIP_OK(ar.take(&dInvMem, air::MEM_MULT));
    HIP_OK(ar.take(&dCellAddr, CELL_CAP)); HIP_OK(ar.take(&dCellBytes, CELL_CAP)); HIP_OK(ar.take(&dCellTime, CELL_CAP)); HIP_OK(ar.take(&dImage, (size_t)blob_len + 1)); HIP_OK(ar.take(&dCellPart, CELL_CAP / NT + 1));
    HIP_OK(ar.take(&dSec, SEC_WORDS));                      // the memory section (seven words per touched cell) and its chunk digests
    if (WIDE) HIP_OK(ar.take(&dWideSide, N));
  }
  HIP_OK(ar.take(&dM, WM * N)); HIP_OK(ar.take(&dL, WM * N2)); HIP_OK(ar.take(&dTree, 4 * (2 * N2 - 1)));
  HIP_OK(ar.take(&dA, WA * N)); HIP_OK(ar.take(&dAL, WA * N2)); HIP_OK(ar.take(&dATree, 4 * (2 * N2 - 1)));
  HIP_OK(ar.take(&dSide, N)); HIP_OK(ar.take(&dSums, N / SCAN_ROWS + 1));
  const size_t n_mult = (size_t)n_code + air::RC_TABLE + (MEM ? air::MEM_MULT : 0);
  HIP_OK(ar.take(&dCode, (size_t)n_code + 1)); HIP_OK(ar.take(&dMult, n_mult));                   // ROM multiplicities, then range multiplicities (mode 3: then LOW3 | BYTE | NIBBLE)
  HIP_OK(ar.take(&dInvRc, air::RC_TABLE)); HIP_OK(ar.take(&dInvRom, (size_t)n_code + 1));
  HIP_OK(ar.take(&dQ, 8 * N2)); HIP_OK(ar.take(&dQTree, 4 * (2 * N2 - 1))); HIP_OK(ar.take(&dW, N2)); HIP_OK(ar.take(&dDinv, N2));
  // chunks of the barycentric sums: enough workgroups to fill the chip (64 x 25 blocks), and few enough terms per 96-bit sum (bary_dot_kernel: BARY_MAX_TERMS)
  const uint32_t n_chunks = N2 > 64 * BARY_ROWS * BARY_MAX_TERMS ? (uint32_t)(N2 / (BARY_ROWS * BARY_MAX_TERMS)) : N2 >= 1024 * NT ? 256 : N2 >= 64 * NT ? 64 : (N2 >= 16 * NT ? 16 : 1);
  HIP_OK(ar.take(&dPart, (size_t)(WT + 4) * n_chunks * 2)); HIP_OK(ar.take(&dPartSum, (size_t)(WT + 4) * 2));
  std::vector<uint32_t*> fri_trees(n_layers), fri_layers(n_layers + 1);

  StageEvents se;
  if (stage_ms) HIP_OK(se.create());
  double host_ms[10] = {0};                                     // (diagnostics) host wall time at every mark
  auto mark = [&](int i) { if (stage_ms) (void)hipEventRecord(se.ev[i], s); if (dbg_t) host_ms[i] = since(t_entry); };

  // ---- 1. main trace, lookup indices + multiplicities, LDE, trace commitment ---------------------------------------------------
  const double t_pre = since(t_entry);
  mark(0);
  int rc;
  std::vector<uint64_t> cell_addr_v, cell_bytes_v; std::vector<uint32_t> cell_time_v;     // (mode 3) the touched cells: the device witness's, or the caller's
  if (IO) {
    if (pub->n_inputs) HIP_OK(h2d(dInputs, pub->inputs, (size_t)pub->n_inputs * 8));
    const zkir_io_args io{dInputs, pub->n_inputs, pub->writes_before, pub->reads_before};
    if (MEM) {
      if (MEM_HOST) {
        HIP_OK(h2d(dMemOld, pub->mem_old, (size_t)pub->n_real * 8));
        HIP_OK(h2d(dMemTold, pub->mem_told, (size_t)pub->n_real * 4));
        cell_addr_v.assign(pub->cell_addr, pub->cell_addr + pub->n_cells); cell_bytes_v.assign(pub->cell_bytes, pub->cell_bytes + pub->n_cells); cell_time_v.assign(pub->cell_time, pub->cell_time + pub->n_cells);
      } else {
        // scratch = the LDE output buffer, which nothing has written yet (WM * 2N words: 1600 B per row against the ~70 B per row the witness needs)
        const size_t need = zkir::memcheck_scratch_bytes(pub->n_real, blob_len);
        if (need > (size_t)WM * N2 * 4) { zkir::set_last_error({ZKIR_ERR_OTHER, "zkir_prove: memcheck scratch does not fit the LDE buffer"}); return ZKIR_ERR_OTHER; }
        rc = zkir::memcheck_device(trace, pub->n_real, blob, blob_len, dL, (size_t)WM * N2 * 4, dMemOld, dMemTold, cell_addr_v, cell_bytes_v, cell_time_v, pin, s);
        if (rc) return rc;
      }
      rc = WIDE ? zkir_main_trace_wide_launch(trace, pub->n_real, &io, dMemOld, dMemTold, 4 * (uint64_t)n_code, dIoScratch, dM, s) : zkir_main_trace_mem_launch(trace, pub->n_real, &io, dMemOld, dMemTold, dIoScratch, dM, s);
    } else rc = zkir_main_trace_io_launch(trace, pub->n_real, &io, dIoScratch, dM, s);
  } else rc = zkir_main_trace_launch(trace, pub->n_real, pub->deferred, dM, s);
  if (rc) return rc;
  hipLaunchKernelGGL(boundary_states_kernel, dim3(1), dim3(256), 0, s, dM, N, pub->n_real - 1, MODE, dBound);   // before the LDE overwrites dM
  {
    if (n_code) HIP_OK(h2d(dCode, blob + 32, (size_t)n_code * 4));           // the code words, little-endian as the blob holds them
    HIP_OK(hipMemsetAsync(dMult, 0, n_mult * 4, s));
    HIP_OK(hipMemsetAsync(dBad, 0xFF, 8, s));
    unsigned g = grid_for(N); if (g > 2048) g = 2048;
    if (IO) HIP_OK(hipMemsetAsync(dIoCount, 0, 16, s));          // [0] the tape-lookup rows, [1] (mode 4) the hash-syscall rows
    hipLaunchKernelGGL(lookup_index_kernel, dim3(g), dim3(NT), 0, s, dM, N, MODE, dCode, n_code, dSide, dMult + n_code, dMult, dBad, dIo, dIoCount, dMult + n_code + air::RC_TABLE, dMemSide, dWideSide, dSec);
  }
  mark(1);
  rc = lde_launch(c, dM, WM, dL, /*mont_out=*/true, s); if (rc) return rc;        // canonical evaluations in, MONTGOMERY words out: the matrices of a proof rest in Montgomery form
  mark(2);
  rc = merkle_commit(c, dL, WM, N2, dTree, /*mont_in=*/true, s); if (rc) return rc;
  uint32_t troot[4], aroot[4], qroot[4], bound[2 * NS + 4] = {}, n_io = 0;
  unsigned long long bad_row = ~0ull;
  uint32_t* mult = pin.take_n<uint32_t>(n_mult);               // (pinned: read back below, and part of the proof)
  if (!mult) HIP_OK(hipErrorOutOfMemory);
  HIP_OK(d2h(troot, dTree + 4 * (2 * N2 - 2), 16));
  HIP_OK(d2h(bound, dBound, sizeof bound));
  uint32_t io_counts[4] = {0, 0, 0, 0};
  if (IO) HIP_OK(d2h(io_counts, dIoCount, 16));
  HIP_OK(hipMemcpyAsync(mult, dMult, n_mult * 4, hipMemcpyDeviceToHost, s));
  HIP_OK(d2h(&bad_row, dBad, 8));
  HIP_OK(sync_d2h());
  n_io = io_counts[0];
  if (bad_row != ~0ull) {
    char m[448];
    snprintf(m, sizeof m, "zkir_prove: row %llu of the trace has no proof in this AIR: its (pc, instruction word) is not in the program's code table (self-modified code, "
                          "a pc outside the code segment), a written limb is out of range, (mode 3) its memory witness is not the row's, or (mode 4) a wide-arithmetic chunk is out of range", bad_row);
    zkir::set_last_error({ZKIR_ERR_ARGUMENT, m});
    return ZKIR_ERR_ARGUMENT;
  }
  // (mode 4 d) the wide tape: the records of the wide rows that go through it (lookup_index_kernel appended them to the section buffer, which nothing else uses before this point), in cycle order
  struct WideRec { uint32_t w[8]; };
  std::vector<WideRec> wrecs;
  if (WIDE && io_counts[2]) {
    const uint32_t nw = io_counts[2];
    if ((size_t)nw * 8 > SEC_WORDS) { zkir::set_last_error({ZKIR_ERR_OTHER, "zkir_prove: more wide-tape rows than the workspace holds"}); return ZKIR_ERR_OTHER; }
    wrecs.resize(nw);
    HIP_OK(hipMemcpyAsync(wrecs.data(), dSec, (size_t)nw * 32, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    std::sort(wrecs.begin(), wrecs.end(), [](const WideRec& x, const WideRec& y) { return x.w[0] < y.w[0]; });
  }
  mark(3);
  std::vector<uint32_t> head;
  header_words(log_n, *pub, bound, head);
  Challenger ch(c->consts);
  ch.observe_n(head.data() + 2, head.size() - 2);
  ch.observe_n(troot, 4);
  std::vector<uint32_t> io_sec;                               // (modes 2 / 3) the tapes and the halt reason as the proof carries them — (v11) in the transcript BEFORE the lookup challenges (a segment's too: its digest only the chain checks)
  if (IO) {
    auto put_u64 = [&](uint64_t v) { for (int i = 0; i < 4; i++) io_sec.push_back((uint32_t)((v >> (16 * i)) & 0xFFFF)); };
    io_sec.push_back((uint32_t)pub->n_inputs); for (uint64_t i = 0; i < pub->n_inputs; i++) put_u64(pub->inputs[i]);
    io_sec.push_back((uint32_t)pub->n_outputs); for (uint64_t i = 0; i < pub->n_outputs; i++) put_u64(pub->outputs[i]);
    io_sec.push_back(pub->halt_kind); put_u64(pub->halt_kind == ZKIR_HALT_EXIT ? pub->halt_code : 0);
    for (size_t at = 0; at < io_sec.size(); at += SECTION_CHUNK) {          // so::observe_section: chunk digests (host: the tapes are short; a chunk is 64 permutations)
      uint32_t dg[4];
      hash_elems_host(c->consts, io_sec.data() + at, std::min((size_t)SECTION_CHUNK, io_sec.size() - at), dg);
      ch.observe_n(dg, 4);
    }
  }
  const uint64_t n_cells_v = cell_addr_v.size();
  std::vector<uint32_t> mem_sec;                              // (mode 3) the touched cells as the proof carries them: fixed before the lookup challenges like the multiplicities
  if (MEM) {
    mem_sec.push_back((uint32_t)n_cells_v);
    for (uint64_t k = 0; k < n_cells_v; k++) {
      const uint64_t a = cell_addr_v[k], b = cell_bytes_v[k];
      if ((a & 7) || (a >> 40) || (k && a <= cell_addr_v[k - 1])) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: the touched cells must be multiples of 8 below 2^40 in strictly increasing order"}); return ZKIR_ERR_ARGUMENT; }
      if (a + 8 > air::CODE_BASE && a < air::CODE_BASE + 4 * (uint64_t)n_code && !(WIDE && (n_code & 1) && a == air::boundary_cell(4 * (uint64_t)n_code))) {   // (mode 4 admits the boundary cell: I_BC)
        // (v11) instruction fetch is tied to the program's words: a store into the code would change what the VM executes next (vm.rs:175: strict protection is off) but not what
        // the AIR lets through — mode 3 proves runs whose loads and stores stay off the cells that overlap the code segment, and both verifiers check the list (55)
        char m[200];
        snprintf(m, sizeof m, "zkir_prove: the run accesses memory cell 0x%llx, which overlaps the code segment [0x1000, 0x%llx): such a run has no mode-3 proof", (unsigned long long)a,
                 (unsigned long long)(air::CODE_BASE + 4 * (uint64_t)n_code));
        zkir::set_last_error({ZKIR_ERR_ARGUMENT, m});
        return ZKIR_ERR_ARGUMENT;
      }
      mem_sec.push_back((uint32_t)(a & 0xFFFFF)); mem_sec.push_back((uint32_t)((a >> 20) & 0xFFFFF)); mem_sec.push_back(cell_time_v[k]);
      for (int i = 0; i < 4; i++) mem_sec.push_back((uint32_t)((b >> (16 * i)) & 0xFFFF));
    }
    // the section enters the transcript through its chunk digests (so::observe_section), hashed in parallel on the device
    const size_t n_chunks_sec = (mem_sec.size() + SECTION_CHUNK - 1) / SECTION_CHUNK;
    if (mem_sec.size() + 4 * n_chunks_sec + 64 > SEC_WORDS) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: more touched cells than the workspace holds"}); return ZKIR_ERR_ARGUMENT; }
    uint32_t* dSecDg = dSec + ((mem_sec.size() + 63) & ~(size_t)63);
    HIP_OK(h2d(dSec, mem_sec.data(), mem_sec.size() * 4));
    hipLaunchKernelGGL(section_hash_kernel, dim3((unsigned)((4 * n_chunks_sec + 63) / 64)), dim3(64), 0, s, c->d_p2, dSec, (uint64_t)mem_sec.size(), dSecDg);
    std::vector<uint32_t> sec_dg(4 * n_chunks_sec);
    HIP_OK(hipMemcpyAsync(sec_dg.data(), dSecDg, sec_dg.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    ch.observe_n(sec_dg.data(), sec_dg.size());
  }
  // (mode 4) the hash calls: the tape the proof carries (from the caller's host witness), checked as the verifier will check it, observed like the memory section
  std::vector<hashcall::Call> hcalls;
  static const uint32_t no_calls[1] = {0u};
  struct { const uint32_t* p; size_t n; const uint32_t* data() const { return p; } size_t size() const { return n; } } hash_sec{no_calls, 1};   // a VIEW of the caller's tape (a 2^22-cycle chain: 134 MB — not copied)
  const double t_hash0 = since(t_entry);
  if (WIDE) {
    if (pub->hash_section && pub->hash_section_words) {
      size_t used = 0;
      const int hrc = hashcall::parse_section(pub->hash_section, (size_t)pub->hash_section_words, pub->n_real, air::CODE_BASE + 4 * (uint64_t)n_code, hcalls, &used);
      if (hrc || used != pub->hash_section_words) {
        char m[160]; snprintf(m, sizeof m, "zkir_prove: the hash section of the public inputs is malformed (check %d): build it with zkir_memcheck_witness_of_mode(.., 4, ..)", hrc ? hrc : 4);
        zkir::set_last_error({ZKIR_ERR_ARGUMENT, m}); return ZKIR_ERR_ARGUMENT;
      }
      hash_sec.p = pub->hash_section; hash_sec.n = used;
    }
    if (hcalls.size() != io_counts[1]) {
      char m[256];
      snprintf(m, sizeof m, "zkir_prove: the run makes %u hash syscalls and the public inputs' hash section records %llu: a run with hash syscalls is proven (mode 4) from the host "
                            "witness (zkir_memcheck_witness_of_mode(.., 4, ..) + zkir_public_inputs_set_memory)", io_counts[1], (unsigned long long)hcalls.size());
      zkir::set_last_error({ZKIR_ERR_ARGUMENT, m}); return ZKIR_ERR_ARGUMENT;
    }
    const size_t n_chunks_sec = (hash_sec.size() + SECTION_CHUNK - 1) / SECTION_CHUNK;
    if (hash_sec.size() + 4 * n_chunks_sec + 64 <= SEC_WORDS) {           // chunk digests on the device (the buffer of the memory section, free again)
      uint32_t* dSecDg = dSec + ((hash_sec.size() + 63) & ~(size_t)63);
      HIP_OK(h2d(dSec, hash_sec.data(), hash_sec.size() * 4));
      hipLaunchKernelGGL(section_hash_kernel, dim3((unsigned)((4 * n_chunks_sec + 63) / 64)), dim3(64), 0, s, c->d_p2, dSec, (uint64_t)hash_sec.size(), dSecDg);
      std::vector<uint32_t> sec_dg(4 * n_chunks_sec);
      HIP_OK(hipMemcpyAsync(sec_dg.data(), dSecDg, sec_dg.size() * 4, hipMemcpyDeviceToHost, s));
      HIP_OK(hipStreamSynchronize(s));
      ch.observe_n(sec_dg.data(), sec_dg.size());
    } else {                                                                         // (more records than the workspace holds: the digests on the host)
      for (size_t at = 0; at < hash_sec.size(); at += SECTION_CHUNK) { uint32_t dg[4]; hash_elems_host(c->consts, hash_sec.data() + at, std::min((size_t)SECTION_CHUNK, hash_sec.size() - at), dg); ch.observe_n(dg, 4); }
    }
    if (dbg_t) fprintf(stderr, "zkir_prove mode 4: hash section (%zu calls, %zu words) parsed, hashed and observed in %.2f ms\n", hcalls.size(), hash_sec.size(), since(t_entry) - t_hash0);
  }
  // (mode 4 d) the wide tape as a proof section: [n] then the records, eight words each; checked as the verifier will check it, observed like the hash section
  std::vector<uint32_t> wide_sec;
  if (WIDE) {
    wide_sec.reserve(1 + 8 * wrecs.size());
    wide_sec.push_back((uint32_t)wrecs.size());
    for (size_t k = 0; k < wrecs.size(); k++) {
      const uint32_t* r = wrecs[k].w;
      const uint64_t b = (uint64_t)r[4] | ((uint64_t)r[5] << 20) | ((uint64_t)r[6] << 40);
      if (r[0] >= pub->n_real || (k && r[0] <= wrecs[k - 1].w[0]) || r[7] < 3 || r[7] > 7 || (r[7] >= 4 && b == 0)) {
        char m[160]; snprintf(m, sizeof m, "zkir_prove: the wide-tape record of row %u is not one of an executed MULH / DIVU / REMU / DIV / REM (the trace is not a run of the VM)", r[0]);
        zkir::set_last_error({ZKIR_ERR_ARGUMENT, m}); return ZKIR_ERR_ARGUMENT;
      }
      wide_sec.insert(wide_sec.end(), r, r + 8);
    }
    const size_t n_chunks_sec = (wide_sec.size() + SECTION_CHUNK - 1) / SECTION_CHUNK;
    if (wide_sec.size() > 4 * SECTION_CHUNK && wide_sec.size() + 4 * n_chunks_sec + 64 <= SEC_WORDS) {     // a long tape: the chunk digests on the device
      uint32_t* dSecDg = dSec + ((wide_sec.size() + 63) & ~(size_t)63);
      HIP_OK(h2d(dSec, wide_sec.data(), wide_sec.size() * 4));
      hipLaunchKernelGGL(section_hash_kernel, dim3((unsigned)((4 * n_chunks_sec + 63) / 64)), dim3(64), 0, s, c->d_p2, dSec, (uint64_t)wide_sec.size(), dSecDg);
      std::vector<uint32_t> sec_dg(4 * n_chunks_sec);
      HIP_OK(hipMemcpyAsync(sec_dg.data(), dSecDg, sec_dg.size() * 4, hipMemcpyDeviceToHost, s));
      HIP_OK(hipStreamSynchronize(s));
      ch.observe_n(sec_dg.data(), sec_dg.size());
    } else {
      for (size_t at = 0; at < wide_sec.size(); at += SECTION_CHUNK) { uint32_t dg[4]; hash_elems_host(c->consts, wide_sec.data() + at, std::min((size_t)SECTION_CHUNK, wide_sec.size() - at), dg); ch.observe_n(dg, 4); }
    }
  }
  ch.observe_n(mult, n_mult);                     // ROM multiplicities, range multiplicities (mode 3: LOW3 | BYTE | NIBBLE): fixed before the lookup challenges
  std::unique_ptr<ProveParams> pp(new ProveParams());         // host staging of this proof's constants
  {
    // ---- 1b. lookup challenges, inverse tables, T, aux trace (helper columns + running sum), its LDE and commitment ----
    const E4 alpha_l = ch.sample_ext(), lambda = ch.sample_ext();
    E4 lam{{1, 0, 0, 0}};
    for (int k = 0; k < 4; k++) pp->lk[air::LK_ALPHA + k] = bb::to_mont(alpha_l.c[k]);
    for (int j = 0; j <= air::N_TUPLE; j++) { for (int k = 0; k < 4; k++) pp->lk[air::LK_LAM + 4 * j + k] = bb::to_mont(lam.c[k]); lam = h_e_mul(lam, lambda); }
    for (int k = 0; k < 4; k++) pp->lk[air::LK_TN + k] = 0;
    pp->lk[air::LK_NIN] = IO ? bb::to_mont((uint32_t)(pub->n_inputs % bb::P)) : 0u;
    { const uint64_t Bc = air::boundary_cell(4 * (uint64_t)n_code); pp->lk[air::LK_B0] = bb::to_mont((uint32_t)(Bc & 0xFFFFF)); pp->lk[air::LK_B1] = bb::to_mont((uint32_t)((Bc >> 20) & 0xFFFFF)); }   // (mode 4) the boundary cell
    HIP_OK(h2d(dPP->lk, pp->lk, sizeof(pp->lk)));
    hipLaunchKernelGGL(lookup_tables_kernel, dim3(grid_for((uint64_t)air::RC_TABLE + n_code)), dim3(NT), 0, s, dCode, n_code, dPP, dInvRc, dInvRom, MODE);
    if (MEM) hipLaunchKernelGGL(mem_tables_kernel, dim3(grid_for(air::MEM_MULT)), dim3(NT), 0, s, dPP, dInvMem);
    E4* inv = pin.take_n<E4>((size_t)air::RC_TABLE + n_code + (MEM ? air::MEM_MULT : 0));
    if (!inv) HIP_OK(hipErrorOutOfMemory);
    HIP_OK(hipMemcpyAsync(inv, dInvRc, (size_t)air::RC_TABLE * sizeof(E4), hipMemcpyDeviceToHost, s));
    if (n_code) HIP_OK(hipMemcpyAsync(inv + air::RC_TABLE, dInvRom, (size_t)n_code * sizeof(E4), hipMemcpyDeviceToHost, s));
    if (MEM) HIP_OK(hipMemcpyAsync(inv + air::RC_TABLE + n_code, dInvMem, (size_t)air::MEM_MULT * sizeof(E4), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    E4 T = bb::e_zero();                                      // Montgomery: sum m_t / (alpha - t) + sum r_u / (alpha - fingerprint_u)
    for (int t = 0; t < air::RC_TABLE; t++) if (mult[n_code + t]) T = bb::e_add(T, bb::e_mul_fm(inv[t], bb::to_mont(mult[n_code + t] % bb::P)));
    for (uint32_t u = 0; u < n_code; u++) if (mult[u]) T = bb::e_add(T, bb::e_mul_fm(inv[air::RC_TABLE + u], bb::to_mont(mult[u] % bb::P)));
    if (MEM) {
      // LOW3 | BYTE | NIBBLE, then the two ends of the memory check, formed like the verifier will form them: per touched cell + 1 / (alpha - fp(cell, time 0, the program
      // image's bytes)) - 1 / (alpha - fp(cell, final time, final bytes)) (oracle: so::mem_table_sum) — on the device (mem_cells_sum_kernel), the host adds the partial sums
      const size_t m0 = (size_t)n_code + air::RC_TABLE;
      for (int t = 0; t < air::MEM_MULT; t++) if (mult[m0 + t]) T = bb::e_add(T, bb::e_mul_fm(inv[air::RC_TABLE + n_code + t], bb::to_mont(mult[m0 + t] % bb::P)));
      if (n_cells_v) {
        if (n_cells_v > CELL_CAP) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: more touched cells than the workspace holds"}); return ZKIR_ERR_ARGUMENT; }
        uint32_t code_size, data_size; memcpy(&code_size, blob + 16, 4); memcpy(&data_size, blob + 20, 4);
        const uint64_t image_len = 32 + (uint64_t)code_size + data_size <= blob_len ? (uint64_t)code_size + data_size : 0;
        if (image_len) HIP_OK(h2d(dImage, blob + 32, image_len));
        HIP_OK(h2d(dCellAddr, cell_addr_v.data(), n_cells_v * 8));
        HIP_OK(h2d(dCellBytes, cell_bytes_v.data(), n_cells_v * 8));
        HIP_OK(h2d(dCellTime, cell_time_v.data(), n_cells_v * 4));
        const unsigned nb = grid_for(n_cells_v);
        hipLaunchKernelGGL(mem_cells_sum_kernel, dim3(nb), dim3(NT), 0, s, dCellAddr, dCellBytes, dCellTime, (uint32_t)n_cells_v, dImage, image_len, dPP, dCellPart);
        std::vector<E4> part(nb);
        HIP_OK(hipMemcpyAsync(part.data(), dCellPart, (size_t)nb * sizeof(E4), hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        for (const E4& e : part) T = bb::e_add(T, e);
      }
    }
    if (IO) {
      // the tapes' share of the table side, formed like the verifier will form it: every output index in [oc_first, oc_last) and every input index in
      // [ic_first, ic_last) once (air.h: mode 2; oracle: so::io_table_sum) — if the trace's WRITE / READ rows do not send exactly these, the sums differ and the proof fails
      auto term = [&](uint64_t k, uint64_t v, uint32_t tag) {
        const uint32_t gl[4] = {(uint32_t)(k % bb::P), (uint32_t)(v & 0xFFFFF), (uint32_t)((v >> 20) & 0xFFFFF), (uint32_t)(v >> 40)};
        E4 fp;
        for (int c4 = 0; c4 < 4; c4++) {
          uint32_t f = bb::mont_mul(pp->lk[air::LK_LAM + 4 * air::N_TUPLE + c4], bb::to_mont(tag));
          for (int j = 0; j < 4; j++) f = bb::add(f, bb::mont_mul(pp->lk[air::LK_LAM + 4 * j + c4], bb::to_mont(gl[j])));
          fp.c[c4] = bb::sub(pp->lk[air::LK_ALPHA + c4], f);
        }
        T = bb::e_add(T, bb::e_inv_m(fp));
      };
      const uint32_t* cn = bound + 2 * NS;                      // (oc, ic) of the first row, of the last row
      if (cn[2] > pub->n_outputs || cn[3] > pub->n_inputs || cn[0] > cn[2] || cn[1] > cn[3]) {
        zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove: the trace writes more outputs / consumes more inputs than the public inputs' tapes hold (or writes_before / reads_before are wrong)"});
        return ZKIR_ERR_ARGUMENT;
      }
      for (uint64_t k = cn[0]; k < cn[2]; k++) term(k, pub->outputs[k], 2);
      for (uint64_t k = cn[1]; k < cn[3]; k++) term(k, pub->inputs[k], 3);
    }
    std::vector<HashAux> hash_aux;                             // (mode 4) the hash rows' helper values HH, to be scattered into the aux trace below
    if (WIDE && !hcalls.empty()) {
      // the hash calls' share of the table side, formed like the verifier will form it (verify.cpp; oracle: so::hash_table_sum): + 1 / (alpha - fp(call)) per call, and the call's
      // memory accesses, which no row states: per touched cell - 1 / (alpha - fp(cell, told, old bytes)) + 1 / (alpha - fp(cell, cycle + 1, new bytes)).  One batch inversion.
      E4 lamv[air::N_TUPLE + 1];
      for (int j = 0; j <= air::N_TUPLE; j++) for (int c4 = 0; c4 < 4; c4++) lamv[j].c[c4] = pp->lk[air::LK_LAM + 4 * j + c4];
      E4 alpha_lm; for (int c4 = 0; c4 < 4; c4++) alpha_lm.c[c4] = pp->lk[air::LK_ALPHA + c4];
      auto mem_d = [&](uint64_t addr, uint32_t t, uint64_t bytes) {
        E4 fp = bb::e_mul_fm(lamv[air::N_TUPLE], bb::to_mont((uint32_t)air::TAG_MEM));
        fp.c[0] = bb::add(fp.c[0], bb::to_mont((uint32_t)(addr & 0xFFFFF)));
        fp = bb::e_add(fp, bb::e_mul_fm(lamv[1], bb::to_mont((uint32_t)((addr >> 20) & 0xFFFFF))));
        fp = bb::e_add(fp, bb::e_mul_fm(lamv[2], bb::to_mont(t)));
        for (int k = 0; k < 8; k++) fp = bb::e_add(fp, bb::e_mul_fm(lamv[3 + k], bb::to_mont((uint32_t)((bytes >> (8 * k)) & 0xFF))));
        return bb::e_sub(alpha_lm, fp);
      };
      hash_aux.resize(hcalls.size());
      const unsigned parts = hashcall::parts_for(hcalls.size());
      std::vector<E4> Tpart(parts, bb::e_zero());
      hashcall::for_calls(hcalls.size(), parts, [&](unsigned part, size_t lo, size_t hi) {        // (host threads: 175 k calls at 2^20 rows of the SHA chain are ~1 s on one core)
        std::vector<E4> d; std::vector<int8_t> sign; std::vector<uint64_t> nb;
        for (size_t ci = lo; ci < hi; ci++) {
          const hashcall::Call& hc = hcalls[ci];
          const uint32_t e[11] = {(uint32_t)(hc.cycle % bb::P), (uint32_t)(hc.in_ptr & 0xFFFFF), (uint32_t)((hc.in_ptr >> 20) & 0xFFFFF), (uint32_t)(hc.in_ptr >> 40), (uint32_t)(hc.len & 0xFFFFF),
                                  (uint32_t)((hc.len >> 20) & 0xFFFFF), (uint32_t)(hc.len >> 40), (uint32_t)(hc.out_ptr & 0xFFFFF), (uint32_t)((hc.out_ptr >> 20) & 0xFFFFF), (uint32_t)(hc.out_ptr >> 40), hc.kind};
          E4 fp = bb::e_mul_fm(lamv[air::N_TUPLE], bb::to_mont((uint32_t)air::TAG_HASH));
          for (int j = 0; j < 11; j++) fp = bb::e_add(fp, bb::e_mul_fm(lamv[j], bb::to_mont(e[j])));
          d.push_back(bb::e_sub(alpha_lm, fp)); sign.push_back(2);                                // (2: a call's own entry — its inverse is also the row's HH)
          hashcall::new_bytes(hc, nb);
          for (size_t k = 0; k < hc.cells.size(); k++) {
            d.push_back(mem_d(hc.cells[k].addr, hc.cells[k].t, hc.cells[k].bytes)); sign.push_back(-1);
            d.push_back(mem_d(hc.cells[k].addr, (uint32_t)((hc.cycle + 1) % bb::P), nb[k])); sign.push_back(1);
          }
        }
        std::vector<E4> pre(d.size());
        E4 acc = bb::e_one_m();
        for (size_t i = 0; i < d.size(); i++) { pre[i] = acc; acc = bb::e_mul_m(acc, d[i]); }
        E4 inv = bb::e_inv_m(acc), Tp = bb::e_zero();
        size_t call = hi;
        for (size_t i = d.size(); i-- > 0;) {
          const E4 di = bb::e_mul_m(inv, pre[i]);
          inv = bb::e_mul_m(inv, d[i]);
          if (sign[i] < 0) Tp = bb::e_sub(Tp, di); else Tp = bb::e_add(Tp, di);
          if (sign[i] == 2) { call--; hash_aux[call].row = (uint32_t)hcalls[call].cycle; hash_aux[call].pad[0] = hash_aux[call].pad[1] = hash_aux[call].pad[2] = 0; hash_aux[call].h = di; }
        }
        Tpart[part] = Tp;
      });
      for (const E4& tp : Tpart) T = bb::e_add(T, tp);
      if (dbg_t) fprintf(stderr, "zkir_prove mode 4: the hash calls' table side (%u host threads) at %.2f ms\n", parts, since(t_entry));
    }
    std::vector<HashAux> wide_aux;                             // (mode 4 d) the wide-tape rows' helper values WW
    if (WIDE && !wrecs.empty()) {
      // the wide tape's share of the table side, formed like the verifier will form it (verify.cpp; oracle: so::wide_table_sum): + 1 / (alpha - fp(cycle, rs1, rs2, the
      // reference's result, opcode)) per record — the result is computed HERE (air::wide_result); the inverse is also the row's WW.  One batch inversion.
      E4 lamv[air::N_TUPLE + 1];
      for (int j = 0; j <= air::N_TUPLE; j++) for (int c4 = 0; c4 < 4; c4++) lamv[j].c[c4] = pp->lk[air::LK_LAM + 4 * j + c4];
      E4 alpha_lm; for (int c4 = 0; c4 < 4; c4++) alpha_lm.c[c4] = pp->lk[air::LK_ALPHA + c4];
      wide_aux.resize(wrecs.size());
      const unsigned parts = hashcall::parts_for(wrecs.size() / 4);
      std::vector<E4> Tpart(parts, bb::e_zero());
      hashcall::for_calls(wrecs.size(), parts, [&](unsigned part, size_t lo, size_t hi) {        // (host threads: a run with 2^18 tape rows is ~50 ms on one core)
        std::vector<E4> d(hi - lo), pre(hi - lo);
        for (size_t k = lo; k < hi; k++) {
          const uint32_t* r = wrecs[k].w;
          const uint64_t a = (uint64_t)r[1] | ((uint64_t)r[2] << 20) | ((uint64_t)r[3] << 40), b = (uint64_t)r[4] | ((uint64_t)r[5] << 20) | ((uint64_t)r[6] << 40);
          const uint64_t y = air::wide_result(r[7], a, b);
          const uint32_t e[11] = {r[0] % bb::P, r[1], r[2], r[3], r[4], r[5], r[6], (uint32_t)(y & 0xFFFFF), (uint32_t)((y >> 20) & 0xFFFFF), (uint32_t)(y >> 40), r[7]};
          E4 fp = bb::e_mul_fm(lamv[air::N_TUPLE], bb::to_mont((uint32_t)air::TAG_WIDE));
          for (int j = 0; j < 11; j++) fp = bb::e_add(fp, bb::e_mul_fm(lamv[j], bb::to_mont(e[j])));
          d[k - lo] = bb::e_sub(alpha_lm, fp);
        }
        E4 acc = bb::e_one_m(), Tp = bb::e_zero();
        for (size_t k = 0; k < d.size(); k++) { pre[k] = acc; acc = bb::e_mul_m(acc, d[k]); }
        E4 inv = bb::e_inv_m(acc);
        for (size_t k = d.size(); k-- > 0;) {
          const E4 dk = bb::e_mul_m(inv, pre[k]);
          inv = bb::e_mul_m(inv, d[k]);
          Tp = bb::e_add(Tp, dk);
          HashAux& x = wide_aux[lo + k];
          x.row = wrecs[lo + k].w[0]; x.pad[0] = x.pad[1] = x.pad[2] = 0; x.h = dk;
        }
        Tpart[part] = Tp;
      });
      for (const E4& tp : Tpart) T = bb::e_add(T, tp);
    }
    const E4 tn = bb::e_mul_fm(T, bb::to_mont(bb::inv((uint32_t)(N % bb::P))));
    for (int k = 0; k < 4; k++) pp->lk[air::LK_TN + k] = tn.c[k];
    HIP_OK(h2d(dPP->lk + air::LK_TN, pp->lk + air::LK_TN, 16));
    hipLaunchKernelGGL(aux_rows_kernel, dim3(grid_for(N)), dim3(NT), 0, s, dSide, N, dInvRc, dInvRom, dPP, dA);
    if (IO) {                                                 // the tape helpers HO | HI: zero but on the WRITE / live READ rows
      HIP_OK(hipMemsetAsync(dA + (size_t)(air::A_HO / 8) * N * 8, 0, (size_t)N * 32, s));
      if (n_io) hipLaunchKernelGGL(io_aux_kernel, dim3(grid_for(n_io)), dim3(NT), 0, s, dIo, n_io, N, dPP, dA);
    }
    if (MEM) hipLaunchKernelGGL(mem_aux_kernel, dim3(grid_for(N)), dim3(NT), 0, s, dSide, dMemSide, N, dInvRc, dInvMem, dPP, dA, dWideSide);   // P0..P8, HMR, HMW, FPN of every row
    if (WIDE) {                                               // the hash-call helpers HH (and the block's four padding columns): zero but on the hash-syscall rows
      HIP_OK(hipMemsetAsync(dA + (size_t)(air::A_HH / 8) * N * 8, 0, (size_t)N * 32, s));
      if (!hash_aux.empty()) {
        if (hash_aux.size() * sizeof(HashAux) > SEC_WORDS * 4) { zkir::set_last_error({ZKIR_ERR_OTHER, "zkir_prove: more hash calls than the workspace holds"}); return ZKIR_ERR_OTHER; }
        HashAux* dHashAux = reinterpret_cast<HashAux*>(dSec);   // (the section buffer is free again: 32 N bytes)
        HIP_OK(h2d(dHashAux, hash_aux.data(), hash_aux.size() * sizeof(HashAux)));
        hipLaunchKernelGGL(hash_aux_kernel, dim3(grid_for(hash_aux.size())), dim3(NT), 0, s, dHashAux, (uint32_t)hash_aux.size(), N, dA, 0u);
      }
      if (!wide_aux.empty()) {                                  // WW of the wide-tape rows (behind the hash rows' list in the same buffer: a row is never both)
        const size_t off = (hash_aux.size() * sizeof(HashAux) + 255) & ~(size_t)255;
        if (off + wide_aux.size() * sizeof(HashAux) > SEC_WORDS * 4) { zkir::set_last_error({ZKIR_ERR_OTHER, "zkir_prove: more wide-tape rows than the workspace holds"}); return ZKIR_ERR_OTHER; }
        HashAux* dWideAux = reinterpret_cast<HashAux*>(reinterpret_cast<char*>(dSec) + off);
        HIP_OK(h2d(dWideAux, wide_aux.data(), wide_aux.size() * sizeof(HashAux)));
        hipLaunchKernelGGL(hash_aux_kernel, dim3(grid_for(wide_aux.size())), dim3(NT), 0, s, dWideAux, (uint32_t)wide_aux.size(), N, dA, 1u);
      }
    }
    const uint32_t n_scan = (uint32_t)((N + SCAN_ROWS - 1) / SCAN_ROWS);
    hipLaunchKernelGGL(scan_local_kernel, dim3(n_scan), dim3(NT), 0, s, dA, N, dSums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(NT), 0, s, dSums, n_scan);
    hipLaunchKernelGGL(scan_add_kernel, dim3(grid_for(N)), dim3(NT), 0, s, dA, N, dSums);
    rc = lde_launch(c, dA, WA, dAL, /*mont_out=*/false, s); if (rc) return rc;      // the aux rows are written in Montgomery form already; the extension is linear
    rc = merkle_commit(c, dAL, WA, N2, dATree, /*mont_in=*/true, s); if (rc) return rc;
    HIP_OK(d2h(aroot, dATree + 4 * (2 * N2 - 2), 16));
    HIP_OK(sync_d2h());
    ch.observe_n(aroot, 4);
  }
  mark(4);
  const E4 alpha = ch.sample_ext();

  // ---- 2. quotient ------------------------------------------------------------------------------------------------------------
  {
    // (the powers run in Montgomery form: one extension product a step and no conversions — the GPU waits for this loop)
    E4 a = bb::e_one_m();
    const E4 alpha_m = bb::e_to_mont(alpha);
    std::vector<E4> alpha_pow(N_CONSTRAINTS);
    for (int k = 0; k < N_CONSTRAINTS; k++) { alpha_pow[k] = a; a = bb::e_mul_m(a, alpha_m); }
    int order[N_CONSTRAINTS];
    const int n_push = air::push_order(MODE, order);            // the order the quotient kernel consumes the coefficients in
    for (int k = 0; k < N_CONSTRAINTS + 2; k++) pp->alpha_seq[k] = k < n_push ? alpha_pow[order[k]] : bb::e_zero();
    for (int i = 0; i < NS; i++) { pp->first_m[i] = bb::to_mont(bound[i]); pp->last_m[i] = bb::to_mont(bound[NS + i]); }
    for (int k = 0; k < 4; k++) pp->cnt_m[k] = bb::to_mont(bound[2 * NS + k]);
    air::boundary_constants(alpha_pow.data(), pp->first_m, pp->last_m, pp->cf, pp->cl, IO ? pp->cnt_m : nullptr);
    pp->deferred = pub->deferred ? 1 : 0;
    HIP_OK(h2d(dPP, pp.get(), sizeof(ProveParams)));
  }
  const uint32_t wn = bb::root_of_unity((int)log_n);
  const uint32_t gN = bb::pow(bb::GEN, N), wn_inv_m = bb::to_mont(bb::inv(wn)), w_last_inv_m = bb::to_mont(bb::inv(bb::pow(wn, pub->n_real - 1)));
  const uint32_t inv_zh_even_m = bb::to_mont(bb::inv(bb::sub(gN, 1))), inv_zh_odd_m = bb::to_mont(bb::inv(bb::sub(bb::neg(gN), 1)));
  const uint32_t last_shift = (uint32_t)((2 * (pub->n_real - 1)) & (N2 - 1));   // x_j - w_N^last = w_N^last (x_(j - 2 last) - 1) on the 2N coset
  if (MODE == 1) hipLaunchKernelGGL(quotient_kernel<1>, dim3(grid_for(N2)), dim3(NT), 0, s, dL, dAL, log_n, c->d_tw_fwd, c->d_inv_xm1, dPP, wn_inv_m, w_last_inv_m, last_shift, inv_zh_even_m, inv_zh_odd_m, dQ);
  else if (MODE == 4) hipLaunchKernelGGL(quotient_kernel<4>, dim3(grid_for(N2)), dim3(NT), 0, s, dL, dAL, log_n, c->d_tw_fwd, c->d_inv_xm1, dPP, wn_inv_m, w_last_inv_m, last_shift, inv_zh_even_m, inv_zh_odd_m, dQ);
  else if (MODE == 3) hipLaunchKernelGGL(quotient_kernel<3>, dim3(grid_for(N2)), dim3(NT), 0, s, dL, dAL, log_n, c->d_tw_fwd, c->d_inv_xm1, dPP, wn_inv_m, w_last_inv_m, last_shift, inv_zh_even_m, inv_zh_odd_m, dQ);
  else if (MODE == 2) hipLaunchKernelGGL(quotient_kernel<2>, dim3(grid_for(N2)), dim3(NT), 0, s, dL, dAL, log_n, c->d_tw_fwd, c->d_inv_xm1, dPP, wn_inv_m, w_last_inv_m, last_shift, inv_zh_even_m, inv_zh_odd_m, dQ);
  else hipLaunchKernelGGL(quotient_kernel<0>, dim3(grid_for(N2)), dim3(NT), 0, s, dL, dAL, log_n, c->d_tw_fwd, c->d_inv_xm1, dPP, wn_inv_m, w_last_inv_m, last_shift, inv_zh_even_m, inv_zh_odd_m, dQ);
  rc = merkle_commit(c, dQ, 4, N2, dQTree, /*mont_in=*/true, s); if (rc) return rc;
  HIP_OK(d2h(qroot, dQTree + 4 * (2 * N2 - 2), 16));
  HIP_OK(sync_d2h());
  mark(5);
  ch.observe_n(qroot, 4);
  const E4 zeta = ch.sample_ext();
  const E4 zeta_w = bb::E4{{bb::mul(zeta.c[0], wn), bb::mul(zeta.c[1], wn), bb::mul(zeta.c[2], wn), bb::mul(zeta.c[3], wn)}};

  // ---- 3. openings by barycentric evaluation over the LDE coset ------------------------------------------------------------
  pp->zeta = bb::e_to_mont(zeta); pp->zeta_w = bb::e_to_mont(zeta_w);
  HIP_OK(h2d(&dPP->zeta, &pp->zeta, 2 * sizeof(E4)));
  hipLaunchKernelGGL(bary_weights_kernel, dim3(grid_for(N2)), dim3(NT), 0, s, log_n, c->d_tw_fwd, dPP, dW, dDinv);
  // columns in the order used everywhere below: main (WM), aux (WA) = WT "trace" columns, then the quotient's four
  hipLaunchKernelGGL(bary_dot_kernel, dim3(n_chunks, (WT + 8) / 8), dim3(NT), 0, s, dL, dAL, dQ, (uint32_t)WM / 8, (uint32_t)WA / 8, (uint64_t)N2, dW, dPart, n_chunks, (uint32_t)__builtin_ctzll(N2 / n_chunks));
  hipLaunchKernelGGL(bary_sum_kernel, dim3(((WT + 4) * 2 + NT / 64 - 1) / (NT / 64)), dim3(NT), 0, s, dPart, (uint32_t)(WT + 4) * 2, n_chunks, dPartSum);
  std::vector<E4> part((size_t)(WT + 4) * 2);
  HIP_OK(d2h(part.data(), dPartSum, part.size() * sizeof(E4)));
  HIP_OK(sync_d2h());
  std::vector<E4> t_z(WT), t_zw(WT), q_z(4);
  {
    // the quotient's columns, over the whole coset g H_2N: scale = ((z/g)^2N - 1) / (2N);  for z = zeta*w the factor is the same because w^(2N) = 1
    // the trace columns, over its even half g H_N: ((z/g)^N - 1) / N, likewise the same at zeta*w (w^N = 1)
    const uint32_t ginv = bb::inv(bb::GEN);
    const E4 zg = bb::E4{{bb::mul(zeta.c[0], ginv), bb::mul(zeta.c[1], ginv), bb::mul(zeta.c[2], ginv), bb::mul(zeta.c[3], ginv)}};
    const E4 zgn = h_e_pow(zg, N2 / 2);
    E4 sc = h_e_mul(zgn, zgn), sc_half = zgn; sc.c[0] = bb::sub(sc.c[0], 1); sc_half.c[0] = bb::sub(sc_half.c[0], 1);
    const uint32_t inv2n = bb::inv((uint32_t)(N2 % bb::P)), invn = bb::inv((uint32_t)((N2 / 2) % bb::P));
    for (int t = 0; t < 4; t++) { sc.c[t] = bb::mul(sc.c[t], inv2n); sc_half.c[t] = bb::mul(sc_half.c[t], invn); }
    for (int k = 0; k < WT + 4; k++) {
      const E4 a = part[2 * (size_t)k], b = part[2 * (size_t)k + 1];
      const E4& sk = k < WT ? sc_half : sc;                                                                   // (trace columns: the even half; the quotient: the whole coset)
      const E4 va = bb::e_mul_m(a, sk), vb = bb::e_mul_m(b, sk);                                              // Montgomery partial sums x canonical scale = canonical
      if (k < WT) { t_z[k] = va; t_zw[k] = vb; } else q_z[k - WT] = va;
    }
  }
  mark(6);
  for (int k = 0; k < WT; k++) ch.observe_n(t_z[k].c, 4);
  for (int k = 0; k < WT; k++) ch.observe_n(t_zw[k].c, 4);
  for (int i = 0; i < 4; i++) ch.observe_n(q_z[i].c, 4);
  const E4 gamma = ch.sample_ext();

  // ---- 4. DEEP codeword ------------------------------------------------------------------------------------------------------
  {
    // (the powers run in Montgomery form: Montgomery x canonical = canonical, Montgomery x Montgomery = Montgomery — two products a step and no conversions; the GPU waits for this loop)
    E4 g = bb::e_to_mont(bb::E4{{1, 0, 0, 0}}), a0 = bb::e_zero(), b0 = bb::e_zero();
    const E4 gamma_m = bb::e_to_mont(gamma);
    for (int k = 0; k < 2 * WT + 4; k++) {
      pp->gamma_pow[k] = g;
      if (k < WT) a0 = bb::e_add(a0, bb::e_mul_m(g, t_z[k]));
      else if (k < 2 * WT) b0 = bb::e_add(b0, bb::e_mul_m(g, t_zw[k - WT]));
      else a0 = bb::e_add(a0, bb::e_mul_m(g, q_z[k - 2 * WT]));
      g = bb::e_mul_m(g, gamma_m);
    }
    pp->a0 = bb::e_to_mont(a0); pp->b0 = bb::e_to_mont(b0);
    HIP_OK(h2d(dPP->gamma_pow, pp->gamma_pow, sizeof(pp->gamma_pow)));
    HIP_OK(h2d(&dPP->a0, &pp->a0, 2 * sizeof(E4)));
  }
  HIP_OK(ar.take(&fri_layers[0], 4 * N2));
  hipLaunchKernelGGL(deep_kernel, dim3(grid_for(N2)), dim3(NT), 0, s, dL, dAL, dQ, log_n, dDinv, dPP, wn_inv_m, WM, WA, fri_layers[0]);
  mark(7);

  // ---- 5. FRI commit phase ----------------------------------------------------------------------------------------------------
  // Enqueued as a whole (round 5): per layer  leaf hash -> tree -> fri_transcript_kernel (root -> beta on the device) -> fold;  ONE copy and ONE synchronisation at the
  // end bring the roots, the betas and the final layer to the host, which replays the transcript (round 4: a root copy, a synchronisation and a parameter upload per layer).
  std::vector<std::array<uint32_t, 4>> lroots(n_layers);
  std::vector<E4> betas(n_layers);
  uint32_t *dChSt, *dFri;
  HIP_OK(ar.take(&dChSt, 16)); HIP_OK(ar.take(&dFri, 16 * (size_t)(n_layers + 1)));
  if (!ch.in.empty()) { zkir::set_last_error({ZKIR_ERR_OTHER, "zkir_prove: the transcript has pending input at the FRI commit phase"}); return ZKIR_ERR_OTHER; }
  HIP_OK(h2d(dChSt, ch.st, sizeof(ch.st)));
  {
    uint32_t shift = bb::GEN;
    int log_m = (int)log_n + 1;
    for (int j = 0; j < n_layers; j++) {
      const int k = ks[j];
      const uint64_t m = 1ull << log_m, g = m >> k;
      HIP_OK(ar.take(&fri_trees[j], 4 * (2 * g - 1)));
      uint32_t* tree = fri_trees[j];
      if (g <= (1u << 11)) hipLaunchKernelGGL(fri_leaf_hash_row16_kernel, dim3(grid_for(16 * g)), dim3(NT), 0, s, c->d_p2, fri_layers[j], m, (uint32_t)k, tree);
      else if (g <= (1u << 14)) hipLaunchKernelGGL(fri_leaf_hash_quad_kernel, dim3(grid_for(4 * g)), dim3(NT), 0, s, c->d_p2, fri_layers[j], m, (uint32_t)k, tree);
      else hipLaunchKernelGGL(fri_leaf_hash_kernel, dim3(grid_for(g)), dim3(NT), 0, s, c->d_p2, fri_layers[j], m, (uint32_t)k, tree);
      launch_tree_levels(c->d_p2, tree, g, c->sync, s);
      hipLaunchKernelGGL(fri_transcript_kernel, dim3(1), dim3(16), 0, s, c->d_p2, dChSt, tree + 4 * (2 * g - 2), dFri + 16 * j);
      FoldParams fp{};                                                         // k binary folds: beta^(2^f) (device), shift^(2^f) — one launch for all of them
      for (int f = 0; f < k; f++) {
        fp.half_shift_inv_m[f] = bb::to_mont(bb::inv(bb::mul(2, shift)));
        shift = bb::mul(shift, shift);
      }
      uint32_t* dst;
      HIP_OK(ar.take(&dst, 4 * g));
      const E4* dBeta = reinterpret_cast<const E4*>(dFri + 16 * j + 4);
      if (k == 1) hipLaunchKernelGGL(fri_fold_k_kernel<1>, dim3(grid_for(g)), dim3(NT), 0, s, fri_layers[j], (uint32_t)log_m, log_n + 1, c->d_tw_fwd, fp, dBeta, dst);
      else if (k == 2) hipLaunchKernelGGL(fri_fold_k_kernel<2>, dim3(grid_for(g)), dim3(NT), 0, s, fri_layers[j], (uint32_t)log_m, log_n + 1, c->d_tw_fwd, fp, dBeta, dst);
      else hipLaunchKernelGGL(fri_fold_k_kernel<3>, dim3(grid_for(g)), dim3(NT), 0, s, fri_layers[j], (uint32_t)log_m, log_n + 1, c->d_tw_fwd, fp, dBeta, dst);
      fri_layers[j + 1] = dst;
      log_m -= k;
    }
  }
  const uint64_t fin_n = 1ull << LOG_FINAL;
  uint32_t fin_cols[4 * 8];
  std::vector<uint32_t> fri_out(16 * (size_t)n_layers);
  if (n_layers) HIP_OK(d2h(fri_out.data(), dFri, fri_out.size() * 4));
  HIP_OK(d2h(fin_cols, fri_layers[n_layers], 4 * fin_n * 4));
  HIP_OK(sync_d2h());
  for (int j = 0; j < n_layers; j++) {                                       // the host's replay of the device's transcript steps
    memcpy(lroots[j].data(), &fri_out[16 * (size_t)j], 16);
    ch.observe_n(lroots[j].data(), 4);
    betas[j] = ch.sample_ext();
    const E4 bm = bb::e_to_mont(betas[j]);
    if (memcmp(bm.c, &fri_out[16 * (size_t)j + 4], 16)) { zkir::set_last_error({ZKIR_ERR_OTHER, "zkir_prove: the device's FRI transcript diverged from the host's"}); return ZKIR_ERR_OTHER; }
  }
  for (uint64_t i = 0; i < fin_n; i++) for (int t = 0; t < 4; t++) ch.observe(fin_cols[t * fin_n + i]);
  // grinding: smallest nonce whose absorption makes the next squeezed element end in POW_BITS zero bits (so::Challenger::grind)
  uint32_t pow_nonce = 0xFFFFFFFFu;
  {
    ch.flush();
    HIP_OK(h2d(dState, ch.st, sizeof(ch.st)));
    const uint32_t batch = 1u << 18;
    for (uint64_t base = 0; base < bb::P && pow_nonce == 0xFFFFFFFFu; base += batch) {
      HIP_OK(hipMemsetAsync(dBest, 0xFF, 4, s));
      hipLaunchKernelGGL(pow_grind_kernel, dim3(batch / NT), dim3(NT), 0, s, c->d_p2, dState, (uint32_t)base, (uint32_t)POW_BITS, dBest);
      HIP_OK(d2h(&pow_nonce, dBest, 4));
      HIP_OK(sync_d2h());
    }
    if (pow_nonce == 0xFFFFFFFFu || !ch.check_pow(pow_nonce, POW_BITS)) { zkir::set_last_error({ZKIR_ERR_OTHER, "zkir_prove: proof-of-work search failed"}); return ZKIR_ERR_OTHER; }
  }
  mark(8);
  std::vector<uint32_t> queries(NUM_QUERIES);
  for (auto& q : queries) q = ch.sample_bits((int)log_n);

  // ---- 6. serialise: header + openings on the host, query section gathered on the device -----------------------------------
  head.reserve(head.size() + blob_len / 2 + io_sec.size() + mem_sec.size() + (WIDE ? hash_sec.size() + wide_sec.size() : 0) + n_mult + 8 * (size_t)WT + 4096);   // (one allocation: the tapes of mode 4 can be 100 MB)
  head.push_back((uint32_t)blob_len);                                         // the program: byte length, then 16-bit halfwords
  for (uint64_t i = 0; i < blob_len; i += 2) head.push_back((uint32_t)blob[i] | (i + 1 < blob_len ? (uint32_t)blob[i + 1] << 8 : 0u));
  if (IO) head.insert(head.end(), io_sec.begin(), io_sec.end());              // the I/O section: the tapes and the halt reason the io digest is a digest of
  if (MEM) head.insert(head.end(), mem_sec.begin(), mem_sec.end());           // (mode 3) the touched cells
  if (WIDE) { head.insert(head.end(), hash_sec.data(), hash_sec.data() + hash_sec.size()); head.insert(head.end(), wide_sec.begin(), wide_sec.end()); }   // (mode 4) the hash calls, the wide tape
  head.insert(head.end(), mult, mult + n_mult);                          // ROM multiplicities, range multiplicities (mode 3: LOW3 | BYTE | NIBBLE)
  head.insert(head.end(), troot, troot + 4); head.insert(head.end(), aroot, aroot + 4); head.insert(head.end(), qroot, qroot + 4);
  for (int k = 0; k < WT; k++) head.insert(head.end(), t_z[k].c, t_z[k].c + 4);
  for (int k = 0; k < WT; k++) head.insert(head.end(), t_zw[k].c, t_zw[k].c + 4);
  for (int i = 0; i < 4; i++) head.insert(head.end(), q_z[i].c, q_z[i].c + 4);
  head.push_back((uint32_t)n_layers);
  for (auto& r : lroots) head.insert(head.end(), r.begin(), r.end());
  for (uint64_t i = 0; i < fin_n; i++) for (int t = 0; t < 4; t++) head.push_back(fin_cols[t * fin_n + i]);
  head.push_back(pow_nonce);

  std::vector<GatherJob> jobs;
  uint32_t off = 0;                                                           // offsets inside the query section
  std::vector<uint32_t> qpos;                                                 // where each query's index word goes
  auto path_jobs = [&](const uint32_t* tree, uint64_t n_leaves, uint64_t leaf) {
    if (n_leaves > 1) { jobs.push_back({tree, n_leaves, (uint32_t)leaf, off, 0u, 2u}); for (uint64_t q = n_leaves; q > 1; q >>= 1) off += 4; }
  };
  for (uint32_t q : queries) {
    qpos.push_back(off); off += 1;
    for (uint64_t pos : {(uint64_t)q, (uint64_t)q + N}) {                     // a trace row = 8 consecutive words out of each of the WM/8 blocks
      jobs.push_back({dL + pos * 8, N2 * 8, (uint32_t)WM / 8, off, 1u, 1u | (8u << 8)}); off += (uint32_t)WM;
      path_jobs(dTree, N2, pos);
    }
    for (uint64_t pos : {(uint64_t)q, (uint64_t)q + N}) {                     // the aux row and its path
      jobs.push_back({dAL + pos * 8, N2 * 8, (uint32_t)WA / 8, off, 1u, 1u | (8u << 8)}); off += (uint32_t)WA;
      path_jobs(dATree, N2, pos);
    }
    for (uint64_t pos : {(uint64_t)q, (uint64_t)q + N}) { jobs.push_back({dQ + pos * 8, 1, 4, off, 1u, 0u}); off += 4; path_jobs(dQTree, N2, pos); }
    int log_m = (int)log_n + 1;
    for (int j = 0; j < n_layers; log_m -= ks[j], j++) {
      const uint64_t m = 1ull << log_m, g = m >> ks[j], idx = q & (g - 1);
      for (uint64_t t = 0; t < (1ull << ks[j]); t++) { jobs.push_back({fri_layers[j] + idx + t * g, m, 4, off, 0u, 0u}); off += 4; }
      path_jobs(fri_trees[j], g, idx);
    }
  }
  GatherJob* dJobs; uint32_t* dOut;
  HIP_OK(ar.take(&dJobs, jobs.size())); HIP_OK(ar.take(&dOut, (size_t)off));
  HIP_OK(h2d(dJobs, jobs.data(), jobs.size() * sizeof(GatherJob)));
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)jobs.size()), dim3(64), 0, s, dJobs, (uint32_t)jobs.size(), dOut);
  const uint64_t total = head.size() + off;
  uint32_t* out = (uint32_t*)malloc(total * 4);
  if (!out) { zkir::set_last_error({ZKIR_ERR_OTHER, "zkir_prove: out of host memory"}); return ZKIR_ERR_OTHER; }
  memcpy(out, head.data(), head.size() * 4);
  uint32_t* h_out = pin.take_n<uint32_t>((size_t)off);          // (the proof block is the caller's to free: it is never the target of a copy itself)
  hipError_t e = h_out ? hipMemcpyAsync(h_out, dOut, (size_t)off * 4, hipMemcpyDeviceToHost, s) : hipErrorOutOfMemory;
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess || check_launch("zkir_prove") != ZKIR_OK) { free(out); if (e != hipSuccess) zkir::set_last_error({ZKIR_ERR_DEVICE, hipGetErrorString(e)}); return ZKIR_ERR_DEVICE; }
  memcpy(out + head.size(), h_out, (size_t)off * 4);
  for (size_t t = 0; t < queries.size(); t++) out[head.size() + qpos[t]] = queries[t];
  mark(9);
  const double t_body = since(t_entry);
  if (stage_ms) {
    (void)hipEventSynchronize(se.ev[9]);
    for (int i = 0; i < 9; i++) (void)hipEventElapsedTime(&stage_ms[i], se.ev[i], se.ev[i + 1]);
  }
  if (dbg_t) {
    fprintf(stderr, "zkir_prove mode %d: %.2f ms before the first stage, %.2f ms to the last mark, %.2f ms at return; host ms per stage:", MODE, t_pre, t_body, since(t_entry));
    for (int i = 0; i < 9; i++) fprintf(stderr, " %.2f", host_ms[i + 1] - host_ms[i]);
    fprintf(stderr, "\n");
  }
  *proof_out = out; *proof_words = total;
  return ZKIR_OK;
}

}  // extern "C"
