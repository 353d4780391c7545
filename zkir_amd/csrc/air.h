// air.h — the AIR of ZKIR-STARK v1: column map of the 152-column main trace and the constraint list, written
// ONCE for the two places of the product that evaluate it: the quotient kernel (stark_prove.inl; base-field values at every point
// of the LDE coset, lazily accumulated) and the host verifier (verify.cpp; extension-field openings at zeta).  The oracle
// (oracle/stark_oracle.cpp, constraints_sum) states the same list independently in naive arithmetic; constraint c carries the
// coefficient alpha^c and the indices below are that list's order.
//
// What the constraints say (default VM mode; `deferred` public input = 0):
//   * the class one-hot follows the opcode: ADD / ADDI / BNE / JAL rows cannot hide as "other" (non-membership witness t5);
//   * wr = one-hot(rd) on ADD / ADDI / JAL rows, empty on BNE / halt / padding rows, at most one register otherwise;
//     selb = one-hot(field b), selc = one-hot(field c) (of field a on BNE rows), xb / xc = the selected registers' limbs;
//   * y = xb + xc, xb + sext(imm17) (mod 2^40, two 20-bit limbs with boolean carries), or pc + 4 — the register selected by wr shows
//     y in the next row, every other register keeps its limbs and storage state (execute.rs:43-63, :185-197, :639-647);
//   * pc' = pc + 4 | pc + sext(imm17) if the BNE operands differ in any limb | pc + sext(off21) for JAL, mod 2^64 (state.rs:131-133);
//   * cycle counts up; row 0 is in the public FIRST state and row n_real - 1 in the public LAST state (for a whole run the verifier
//     requires the first state to be the VM's initial one: cycle 0, entry point, zero registers); the row count is public: row
//     n_real - 1 is the halt row, only padding follows it, padding keeps everything.
// Not constrained yet (stated in DESIGN.md §8.5): limb / carry / field RANGES (need the lookup argument the range-check
// multiplicities of K2 are produced for), the instruction word at pc being the program's (same lookup), the other 46 opcodes'
// values, and deferred-mode arithmetic (deferred = 1 relaxes the write constraints to "unwritten registers keep their value").
#pragma once
#include "babybear.h"

namespace air {

constexpr int W = 152;
enum : int { C_CYCLE = 0, C_PC = 1, C_OP = 4, C_FA = 5, C_FB = 6, C_FC = 7, C_FHI = 8, C_LIMB = 9, C_STATE = 57, C_WR = 73, C_SELB = 88, C_SELC = 103,
             C_XB = 118, C_XC = 121, C_Y = 124, C_K = 127, C_T = 134, C_S = 139, C_SE = 140, C_C0 = 141, C_C1 = 142, C_D0 = 143, C_D1 = 144, C_D2 = 145,
             C_DL0 = 146, C_NE = 147, C_IV = 148, C_TK = 151 };
enum : int { K_ADD = 0, K_ADDI = 1, K_BNE = 2, K_JAL = 3, K_OTH = 4, K_HALT = 5, K_PAD = 6 };
constexpr uint32_t OP_ADD = 0x00, OP_ADDI = 0x08, OP_BNE = 0x41, OP_JAL = 0x48;

// constraint indices (the order of oracle/stark_oracle.cpp: constraints_sum)
enum : int { I_CYCLE = 0, I_CYCLE0 = 1, I_ENTRY = 2, I_ZERO0 = 5, I_HALT = 69, I_R0 = 70, I_BOOL_STATE = 74, I_BOOL_SEL = 90, I_BOOL_K = 135, I_BOOL_MISC = 142,
             I_ONE_CLASS = 150, I_CLASS_OP = 151, I_CHAIN = 155, I_OTH = 159, I_WR = 160, I_SELB = 163, I_SELC = 165, I_OPERAND = 167, I_VALUE = 173,
             I_NE = 182, I_TK = 186, I_DL0 = 188, I_SE = 189, I_PC = 190, I_PC_KEEP = 193, I_REGS = 196, I_TAIL = 256, I_LAST = 259, N_CONSTRAINTS = 327 };
// Boundary states (proof format v4): the 68 state words (cycle, 3 pc limbs, 48 register limbs, 16 storage states) of row 0 and of the
// last executed row are public; constraint 1 + i pins state word i of row 0, constraint I_LAST + i that of row n_real - 1.
constexpr int N_STATE = 68;
BB_HD constexpr int state_col(int i) { return i < 4 ? i : C_LIMB + (i - 4); }

// canonical constant -> Montgomery form at compile time
constexpr uint32_t M(uint64_t v) { return (uint32_t)(((v % bb::P) * (uint64_t)bb::R1) % bb::P); }
struct RegConsts { uint32_t r[16], r2[16]; };
constexpr RegConsts make_reg_consts() { RegConsts c{}; for (int i = 0; i < 16; i++) { c.r[i] = M((uint64_t)i); c.r2[i] = M((uint64_t)i * i); } return c; }

// `Ops` supplies the value type and its arithmetic:
//   using V;  V add(V,V), sub(V,V), mul(V,V);  V mulc(V, uint32_t montgomery_constant);  V cst(uint32_t montgomery_constant);
//   V loc(int column), nxt(int column);  void push(int constraint_index, V value)      (values in Montgomery form throughout)
// is_first / is_last / is_trans: the row selectors at the evaluation point; first_m / last_m: the public boundary states (Montgomery).
template <class Ops>
BB_HD void eval(Ops& o, typename Ops::V is_first, typename Ops::V is_last, typename Ops::V is_trans, const uint32_t* first_m, const uint32_t* last_m, bool deferred) {
  using V = typename Ops::V;
  const V one = o.cst(bb::R1), zero = o.cst(0);
  auto boolean = [&](int idx, V b) { o.push(idx, o.mul(b, o.sub(b, one))); };
  const V op = o.loc(C_OP), fa = o.loc(C_FA), fb = o.loc(C_FB), fc = o.loc(C_FC), fhi = o.loc(C_FHI), s = o.loc(C_S), se = o.loc(C_SE);
  V K[7];
#pragma unroll
  for (int k = 0; k < 7; k++) K[k] = o.loc(C_K + k);
  const V y[3] = {o.loc(C_Y), o.loc(C_Y + 1), o.loc(C_Y + 2)};
  const V pc[3] = {o.loc(C_PC), o.loc(C_PC + 1), o.loc(C_PC + 2)};
  const V npc[3] = {o.nxt(C_PC), o.nxt(C_PC + 1), o.nxt(C_PC + 2)};
  // 1. cycle counter, first row, last executed row
  const V cyc = o.loc(C_CYCLE);
  o.push(I_CYCLE, o.mul(o.sub(o.sub(o.nxt(C_CYCLE), cyc), one), is_trans));
  o.push(I_CYCLE0, o.mul(o.sub(cyc, o.cst(first_m[0])), is_first));
  o.push(I_LAST, o.mul(o.sub(cyc, o.cst(last_m[0])), is_last));
#pragma unroll
  for (int l = 0; l < 3; l++) {
    o.push(I_ENTRY + l, o.mul(o.sub(pc[l], o.cst(first_m[1 + l])), is_first));
    o.push(I_LAST + 1 + l, o.mul(o.sub(pc[l], o.cst(last_m[1 + l])), is_last));
  }
  o.push(I_HALT, o.mul(o.sub(K[K_HALT], one), is_last));
  // registers: first-row zero, R0, booleans, selector moments, operand sums, update — one pass per register
  V w0 = zero, w1 = zero, w2 = zero, b1 = zero, b2 = zero, c1s = zero, c2s = zero;
  V xbs[3] = {zero, zero, zero}, xcs[3] = {zero, zero, zero};
#pragma unroll 1
  for (int r = 0; r < 16; r++) {
    V limb[3];
#pragma unroll
    for (int l = 0; l < 3; l++) {
      limb[l] = o.loc(C_LIMB + 3 * r + l);
      o.push(I_ZERO0 + 3 * r + l, o.mul(o.sub(limb[l], o.cst(first_m[4 + 3 * r + l])), is_first));
      o.push(I_LAST + 4 + 3 * r + l, o.mul(o.sub(limb[l], o.cst(last_m[4 + 3 * r + l])), is_last));
    }
    const V st = o.loc(C_STATE + r);
    o.push(I_ZERO0 + 48 + r, o.mul(o.sub(st, o.cst(first_m[52 + r])), is_first));
    o.push(I_LAST + 52 + r, o.mul(o.sub(st, o.cst(last_m[52 + r])), is_last));
    boolean(I_BOOL_STATE + r, st);
    if (r == 0) {
#pragma unroll
      for (int l = 0; l < 3; l++) o.push(I_R0 + l, limb[l]);
      o.push(I_R0 + 3, st);
      continue;
    }
    const V wr = o.loc(C_WR + r - 1), sb = o.loc(C_SELB + r - 1), sc = o.loc(C_SELC + r - 1);
    boolean(I_BOOL_SEL + 3 * (r - 1), wr); boolean(I_BOOL_SEL + 3 * (r - 1) + 1, sb); boolean(I_BOOL_SEL + 3 * (r - 1) + 2, sc);
    constexpr RegConsts RC = make_reg_consts();
    const uint32_t rm = RC.r[r], r2m = RC.r2[r];
    w0 = o.add(w0, wr); w1 = o.add(w1, o.mulc(wr, rm)); w2 = o.add(w2, o.mulc(wr, r2m));
    b1 = o.add(b1, o.mulc(sb, rm)); b2 = o.add(b2, o.mulc(sb, r2m));
    c1s = o.add(c1s, o.mulc(sc, rm)); c2s = o.add(c2s, o.mulc(sc, r2m));
#pragma unroll
    for (int l = 0; l < 3; l++) {
      xbs[l] = o.add(xbs[l], o.mul(sb, limb[l])); xcs[l] = o.add(xcs[l], o.mul(sc, limb[l]));
      const V nx = o.nxt(C_LIMB + 3 * r + l);
      // nx - cur - wr * ((1 - D) y + D nx - cur)
      const V tgt = deferred ? nx : y[l];
      o.push(I_REGS + 4 * (r - 1) + l, o.mul(o.sub(o.sub(nx, limb[l]), o.mul(wr, o.sub(tgt, limb[l]))), is_trans));
    }
    const V nst = o.nxt(C_STATE + r);
    const V tgt = deferred ? nst : zero;
    o.push(I_REGS + 4 * (r - 1) + 3, o.mul(o.sub(o.sub(nst, st), o.mul(wr, o.sub(tgt, st))), is_trans));
  }
  // 3. remaining booleans
#pragma unroll
  for (int k = 0; k < 7; k++) boolean(I_BOOL_K + k, K[k]);
  const V c0 = o.loc(C_C0), c1 = o.loc(C_C1), d0 = o.loc(C_D0), d1 = o.loc(C_D1), d2 = o.loc(C_D2), ne = o.loc(C_NE), tk = o.loc(C_TK);
  boolean(I_BOOL_MISC, s); boolean(I_BOOL_MISC + 1, c0); boolean(I_BOOL_MISC + 2, c1); boolean(I_BOOL_MISC + 3, d0); boolean(I_BOOL_MISC + 4, d1);
  boolean(I_BOOL_MISC + 5, d2); boolean(I_BOOL_MISC + 6, ne); boolean(I_BOOL_MISC + 7, tk);
  // 4. classes and the opcode
  {
    V sum = K[0];
#pragma unroll
    for (int k = 1; k < 7; k++) sum = o.add(sum, K[k]);
    o.push(I_ONE_CLASS, o.sub(sum, one));
  }
  o.push(I_CLASS_OP, o.mul(K[K_ADD], op));
  o.push(I_CLASS_OP + 1, o.mul(K[K_ADDI], o.sub(op, o.cst(M(OP_ADDI)))));
  o.push(I_CLASS_OP + 2, o.mul(K[K_BNE], o.sub(op, o.cst(M(OP_BNE)))));
  o.push(I_CLASS_OP + 3, o.mul(K[K_JAL], o.sub(op, o.cst(M(OP_JAL)))));
  const V t1 = o.loc(C_T), t2 = o.loc(C_T + 1), t3 = o.loc(C_T + 2), tinv = o.loc(C_T + 3), t5 = o.loc(C_T + 4);
  o.push(I_CHAIN, o.sub(t1, o.mul(op, o.sub(op, o.cst(M(8))))));
  o.push(I_CHAIN + 1, o.sub(t2, o.mul(o.sub(op, o.cst(M(OP_BNE))), o.sub(op, o.cst(M(OP_JAL))))));
  o.push(I_CHAIN + 2, o.sub(t3, o.mul(t1, t2)));
  o.push(I_CHAIN + 3, o.sub(t5, o.mul(t3, tinv)));
  o.push(I_OTH, deferred ? zero : o.mul(K[K_OTH], o.sub(t5, one)));
  // 5. selectors
  o.push(I_WR, deferred ? zero : o.sub(o.mul(w1, w1), w2));
  o.push(I_WR + 1, o.mul(o.add(o.add(K[K_ADD], K[K_ADDI]), K[K_JAL]), o.sub(w1, fa)));
  o.push(I_WR + 2, o.mul(o.add(o.add(K[K_BNE], K[K_HALT]), K[K_PAD]), w0));
  o.push(I_SELB, o.sub(b1, fb)); o.push(I_SELB + 1, o.sub(o.mul(b1, b1), b2));
  o.push(I_SELC, o.sub(c1s, o.add(fc, o.mul(K[K_BNE], o.sub(fa, fc))))); o.push(I_SELC + 1, o.sub(o.mul(c1s, c1s), c2s));
  // 6. operands
  V xb[3], xc[3];
#pragma unroll
  for (int l = 0; l < 3; l++) {
    xb[l] = o.loc(C_XB + l); xc[l] = o.loc(C_XC + l);
    o.push(I_OPERAND + 2 * l, o.sub(xb[l], xbs[l])); o.push(I_OPERAND + 2 * l + 1, o.sub(xc[l], xcs[l]));
  }
  // 7. values written
  const V imm17 = o.add(fc, o.mulc(fhi, M(16)));
  const V im0 = o.add(o.sub(imm17, o.mulc(s, M(1u << 17))), o.mulc(s, M(1u << 20))), im1 = o.mulc(s, M(0xFFFFF));
  const V lo20 = o.sub(o.add(o.add(fb, o.mulc(fc, M(16))), o.mulc(fhi, M(256))), o.mulc(s, M(1u << 20)));
  const V c0s20 = o.mulc(c0, M(1u << 20)), c1s20 = o.mulc(c1, M(1u << 20));
  o.push(I_VALUE, o.mul(K[K_ADD], o.add(o.sub(o.sub(y[0], xb[0]), xc[0]), c0s20)));
  o.push(I_VALUE + 1, o.mul(K[K_ADD], o.add(o.sub(o.sub(o.sub(y[1], xb[1]), xc[1]), c0), c1s20)));
  o.push(I_VALUE + 2, o.mul(K[K_ADD], y[2]));
  o.push(I_VALUE + 3, o.mul(K[K_ADDI], o.add(o.sub(o.sub(y[0], xb[0]), im0), c0s20)));
  o.push(I_VALUE + 4, o.mul(K[K_ADDI], o.add(o.sub(o.sub(o.sub(y[1], xb[1]), im1), c0), c1s20)));
  o.push(I_VALUE + 5, o.mul(K[K_ADDI], y[2]));
  o.push(I_VALUE + 6, o.mul(K[K_JAL], o.add(o.sub(o.sub(y[0], pc[0]), o.cst(M(4))), c0s20)));
  o.push(I_VALUE + 7, o.mul(K[K_JAL], o.add(o.sub(o.sub(y[1], pc[1]), c0), c1s20)));
  o.push(I_VALUE + 8, o.mul(K[K_JAL], o.sub(o.sub(y[2], pc[2]), c1)));
  // 8. BNE operands differ?
  {
    V dot = zero;
#pragma unroll
    for (int l = 0; l < 3; l++) {
      const V d = o.sub(xb[l], xc[l]);
      dot = o.add(dot, o.mul(d, o.loc(C_IV + l)));
      o.push(I_NE + l, o.mul(o.sub(one, ne), d));
    }
    o.push(I_NE + 3, o.sub(ne, dot));
  }
  o.push(I_TK, o.mul(K[K_BNE], o.sub(tk, ne)));
  o.push(I_TK + 1, o.mul(o.sub(one, K[K_BNE]), tk));
  // 9. next pc
  const V four = o.cst(M(4)), dl0 = o.loc(C_DL0);
  o.push(I_DL0, o.sub(dl0, o.add(o.add(four, o.mul(tk, o.sub(im0, four))), o.mul(K[K_JAL], o.sub(lo20, four)))));
  o.push(I_SE, o.sub(se, o.mul(o.add(tk, K[K_JAL]), s)));
  const V kc = o.add(o.add(K[K_ADD], K[K_ADDI]), o.add(K[K_BNE], K[K_JAL])), hp = o.add(K[K_HALT], K[K_PAD]);
  o.push(I_PC, o.mul(o.mul(kc, o.add(o.sub(o.sub(npc[0], pc[0]), dl0), o.mulc(d0, M(1u << 20)))), is_trans));
  o.push(I_PC + 1, o.mul(o.mul(kc, o.add(o.sub(o.sub(o.sub(npc[1], pc[1]), o.mulc(se, M(0xFFFFF))), d0), o.mulc(d1, M(1u << 20)))), is_trans));
  o.push(I_PC + 2, o.mul(o.mul(kc, o.add(o.sub(o.sub(o.sub(npc[2], pc[2]), o.mulc(se, M(0xFFFFFF))), d1), o.mulc(d2, M(1u << 24)))), is_trans));
#pragma unroll
  for (int l = 0; l < 3; l++) o.push(I_PC_KEEP + l, o.mul(o.mul(hp, o.sub(npc[l], pc[l])), is_trans));
  // 11. executed rows, the halt row, padding
  const V npad = o.nxt(C_K + K_PAD);
  o.push(I_TAIL, o.mul(o.mul(K[K_HALT], o.sub(one, npad)), is_trans));
  o.push(I_TAIL + 1, o.mul(o.mul(K[K_PAD], o.sub(one, npad)), is_trans));
  o.push(I_TAIL + 2, o.mul(o.mul(o.sub(o.sub(one, K[K_PAD]), K[K_HALT]), npad), is_trans));
}

}  // namespace air
