// stark.hip — self-defined prover stages over Baby Bear on gfx950 ("ZKIR-STARK v1").
//
// The reference has none of this (parity unpinned); the spec is oracle/stark_oracle.cpp and every
// kernel here is checked bit-for-bit against it.  Stage A (this part): main-trace field columns, Poseidon2-12 Merkle commitment
// (the coset LDE between them lives in ntt.hip).
//
//   main_trace_kernel   372 B/row SoA trace -> the 172 logical Baby Bear columns of the AIR (152 committed by default, 168 deferred) (air.h: limbs of pc / instruction fields /
//                       registers, storage state, write and operand selectors, operands, result, opcode classes, range chunks, carries),
//                       padded to a power of two, written in the B8 layout (blocks of 8 columns, [rows][8]).  HBM-bound: ~170 B
//                       read (values + states of the row and the next) + 608 B written per row.
//   (NTT / coset LDE kernels: ntt.hip)
//   merkle kernels      Poseidon2 width-12 sponge over the rows of the LDE matrix (one lane per leaf, column reads coalesced
//                       across lanes) + 2-to-1 compression layers.  ALU-bound (≈740 Montgomery multiplications per permutation);
//                       no MFMA: 31-bit modular integer work, no dense contraction.
#include <hip/hip_runtime.h>

#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/zkir_amd.h"
#include "../../include/zkir_amd_experimental.h"
#include "air.h"
#include "babybear.h"
#include "host.h"
#include "hashcall.h"
#include "poseidon2.h"

namespace {

constexpr int NT = 256;

inline unsigned grid_for(uint64_t n, int per = NT) { return (unsigned)((n + per - 1) / per); }
int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { zkir::set_last_error({ZKIR_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)}); return ZKIR_ERR_DEVICE; }
  return ZKIR_OK;
}

__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return bits == 0 ? 0u : __brev(x) >> (32 - bits); }
// B8 matrix layout (include/zkir_amd.h): element (column k, row j) of a matrix with n rows
__host__ __device__ __forceinline__ uint64_t b8(uint32_t k, uint64_t j, uint64_t n) { return ((uint64_t)(k >> 3) * n + j) * 8 + (k & 7); }

// ------------------------------------------------------------------------------------------------
// main trace columns (oracle: so::main_trace)
// ------------------------------------------------------------------------------------------------
// Montgomery-free helpers for the witness columns (canonical values in and out)
BB_HD uint32_t f_inv(uint32_t a) {          // a^(p-2), canonical; 0 -> 0
  uint32_t r = bb::R1, b = bb::to_mont(a), e = bb::P - 2;
  while (e) { if (e & 1) r = bb::mont_mul(r, b); b = bb::mont_mul(b, b); e >>= 1; }
  return bb::from_mont(r);
}
BB_HD void reg_limbs(uint64_t v, uint32_t st, uint32_t out[3]) {
  const int bits = st ? 30 : 20;
  const uint64_t mask = (1ull << bits) - 1;
  out[0] = (uint32_t)(v & mask); out[1] = (uint32_t)((v >> bits) & mask); out[2] = (uint32_t)(v >> (2 * bits));
}

// One thread per (padded) row; every column write is coalesced across lanes.  Rows >= n_real are padding: they repeat the last
// executed row's state with class "pad" and keep counting cycles.  Mirrors so::main_trace of the oracle word for word.
// A lane builds the logical words of ITS row in registers (every column index below is a compile-time constant once the register loop
// is unrolled) and stores them as 40 16-byte vectors; the two halves of a 32-byte block position are written back to back, so the
// L2 merges them into full sectors.
#ifndef MT_WAVES
#define MT_WAVES 3
#endif
#ifndef MT_WAVES_MEM
#define MT_WAVES_MEM 2              // mode 3 builds 244 logical words per row: at three waves per SIMD (168 VGPRs) it spills 392 B / lane
#endif
// DEF = the deferred VM mode: decides which logical columns are committed (air.h: is_virtual) — 152 columns by default, 168 deferred.
// The row itself is a host + device function: the kernel below runs it once per lane, zkir_main_trace_host once per row on the CPU — the same
// code, so the CPU test suite (no GPU) checks it against the oracle column by column (tests/test_abi.py).
// SKIP: the first SKIP blocks are not stored (experiment, profiles/HISTORY.md round 4: the LDE's first pass generates them itself — zkir_lde_fused01_launch)
// MODE: 0 default, 1 deferred, 2 default + the I/O argument (air.h): there `io` carries the input tape and the ecall counts before the trace, `cnt` the prefix counts
// (WRITE ecalls, READ ecalls among rows < i of THIS trace) of every row.
// MODE 3 = mode 2 + the memory argument: mem_old / mem_told = per row the bytes of the accessed 8-byte cell before the access and the time of its previous access (zkir_memcheck_witness_of)
struct IoRowArgs { const uint64_t* inputs; uint64_t n_inputs, writes_before, reads_before; const uint32_t* cnt; /* [N][2] */ const uint64_t* mem_old; const uint32_t* mem_told; /* [n_real] */
                   uint64_t boundary; /* (mode 4) the boundary cell of the program's code segment (air::boundary_cell) */ };
template <int MODE, int SKIP = 0>
BB_HD void main_trace_row(const zkir_trace_columns& t, uint64_t n_real, uint64_t N, uint64_t i, uint32_t* __restrict__ out, const IoRowArgs* io = nullptr) {
  using namespace air;
  constexpr bool DEF = MODE == 1;
  constexpr uint32_t deferred = DEF ? 1u : 0u;
  uint32_t rowv[W];
  auto col = [&](int k) -> uint32_t& { return rowv[k]; };
  const bool pad = i >= n_real, last = i + 1 >= n_real;
  const uint64_t src = pad ? n_real - 1 : i;
  col(C_CYCLE) = (uint32_t)((t.cycle[src] + (i - src)) % bb::P);               // padding keeps counting from the last executed row
  const uint64_t pcv = t.pc[src];
  const uint32_t pc[3] = {(uint32_t)(pcv & 0xFFFFF), (uint32_t)((pcv >> 20) & 0xFFFFF), (uint32_t)(pcv >> 40)};
  col(C_PC) = pc[0]; col(C_PC + 1) = pc[1]; col(C_PC + 2) = pc[2];
  const uint32_t w = t.instruction[src];
  const uint32_t op = w & 0x7F, fa = (w >> 7) & 0xF, fb = (w >> 11) & 0xF, fc = (w >> 15) & 0xF, fhi = w >> 19, s = w >> 31;
  col(C_OP) = op; col(C_FA) = fa; col(C_FB) = fb; col(C_FC) = fc; col(C_FHI) = fhi; col(C_S) = s;
  int cls = pad ? K_PAD : last ? K_HALT : K_OTH;
  if (cls == K_OTH) {
    int wc = (int)opclass_of(op, MODE);
    if (wc == K_EBREAK) wc = K_OTH;                                               // (an EBREAK that is not the halt row: no honest run has one — the row runs as "other" and I_OPCLASS fails on it)
    if (!deferred) cls = wc;
    else if (wc == K_BRE || wc == K_JAL || wc == K_BRU || wc == K_JALR || wc == K_OJ) cls = K_OJ;     // deferred mode: no opcode semantics, but class "other" is sequential
  }
  const bool oth_like = cls == K_OTH || (cls == K_OJ && deferred);                 // what the row wrote is read off the next row
#pragma unroll
  for (int k = 0; k < N_CLASS; k++) col(kcol(k)) = cls == k;                     // (mode 2: an executed ecall row has none — its class is the sum of its syscall flags)
  if (MODE >= 3) { col(C_KLD) = cls == K_LD; col(C_KST) = cls == K_ST; col(C_KLG) = cls == K_LG; col(C_KSH) = cls == K_SH; col(C_KMU) = cls == K_MU; }
  col(C_OPC) = opclass_of(op, MODE);                                                  // of the WORD, whatever class the row runs as: part of the ROM tuple
  const bool branch = cls == K_BRE || cls == K_BRU;
  const uint32_t tc = (branch || (MODE >= 3 && cls == K_ST)) ? fa : fc;         // B-type and S-type words have rs1 in field a (rs2 in field b)
  uint32_t xb[3] = {0, 0, 0}, xc[3] = {0, 0, 0}, y[3] = {0, 0, 0};
  bool first = true;
#pragma unroll
  for (int g = 0; g < 16; g++) {
    const uint64_t o = (uint64_t)g * t.reg_stride + src;
    const uint64_t v = t.registers[o];
    const uint32_t st = t.reg_state[o];
    uint32_t limb[3];
    reg_limbs(v, st, limb);
    col(C_LIMB + 3 * g) = limb[0]; col(C_LIMB + 3 * g + 1) = limb[1]; col(C_LIMB + 3 * g + 2) = limb[2];
    col(C_STATE + g) = st;
    if (g == 0) continue;
    col(C_SELB + g - 1) = fb == (uint32_t)g; col(C_SELC + g - 1) = tc == (uint32_t)g;
    if (fb == (uint32_t)g) { xb[0] = limb[0]; xb[1] = limb[1]; xb[2] = limb[2]; }
    if (tc == (uint32_t)g) { xc[0] = limb[0]; xc[1] = limb[1]; xc[2] = limb[2]; }
    uint32_t wr = 0;
    if (cls == K_ADD || cls == K_ADDI || cls == K_JAL || cls == K_JALR || cls == K_SUB || cls == K_SE || cls == K_SU || cls == K_CMN || cls == K_CMZ || (MODE >= 3 && (cls == K_LD || cls == K_LG || cls == K_SH || cls == K_MU || cls == K_WA))) wr = fa == (uint32_t)g;   // (a conditional move: cleared below if its condition fails)
    else if (oth_like) {                                         // any other instruction: what it wrote is read off the next row
      uint32_t nl[3];
      const uint32_t nst = t.reg_state[o + 1];
      reg_limbs(t.registers[o + 1], nst, nl);
      if (nl[0] != limb[0] || nl[1] != limb[1] || nl[2] != limb[2] || nst != st) {
        wr = 1;
        if (first && !deferred) { y[0] = nl[0]; y[1] = nl[1]; y[2] = nl[2]; first = false; }
      }
    }
    col(C_WR + g - 1) = wr;
  }
  if (MODE >= 2) {
    // the counters every row shows — what happened BEFORE it — and, on an executed ecall, the dispatch on R10 (syscall.rs:94-177; R10 = 0 halts: the halt row)
    const uint64_t writes = io->writes_before + io->cnt[2 * i], reads = io->reads_before + io->cnt[2 * i + 1];
    col(C_OC) = (uint32_t)(writes % bb::P); col(C_IC) = (uint32_t)((reads < io->n_inputs ? reads : io->n_inputs) % bb::P);
    col(C_F2) = col(C_RL) = col(C_RE) = col(C_FH) = col(C_H0) = col(C_H1) = 0;
    if (cls == K_ECALL) {
      const uint64_t num = t.registers[(uint64_t)10 * t.reg_stride + src];
      if (num == 2) col(C_F2) = 1;
      else {
        if (num == 1) {
          if (reads < io->n_inputs) { col(C_RL) = 1; const uint64_t v = io->inputs[reads]; y[0] = (uint32_t)(v & 0xFFFFF); y[1] = (uint32_t)((v >> 20) & 0xFFFFF); y[2] = (uint32_t)(v >> 40); }
          else col(C_RE) = 1;
        } else { col(C_FH) = 1; col(C_H0) = (uint32_t)((num - 3) & 1); col(C_H1) = (uint32_t)(((num - 3) >> 1) & 1); }
#pragma unroll
        for (int r = 0; r < 15; r++) col(C_WR + r) = r == 9;                      // READ and the hash syscalls write R10 (and nothing else)
      }
    }
  }
  col(C_XB) = xb[0]; col(C_XB + 1) = xb[1]; col(C_XB + 2) = xb[2];
  col(C_XC) = xc[0]; col(C_XC + 1) = xc[1]; col(C_XC + 2) = xc[2];
  uint32_t ne = 0, iv[3] = {0, 0, 0};
#pragma unroll
  for (int l = 0; l < 3; l++) if (!ne && xb[l] != xc[l]) { ne = 1; iv[l] = f_inv(bb::sub(xb[l], xc[l])); }
  col(C_NE) = ne; col(C_IV) = iv[0]; col(C_IV + 1) = iv[1]; col(C_IV + 2) = iv[2];
  // the 40-bit difference of the masked operands and its borrows: xb - xc (SUB, SLTU / SGEU), xc - xb (BLTU / BGEU: rs1 = field a)
  // (v5) ordered comparisons, signed or not: the high limbs enter BIASED, t = limb + 2^19 sgn - 2^20 (sign bit) — the limb of value XOR 2^39 when
  // the comparison is signed (value.rs:710-716); u = (ta, tb) is the row's second range-checked pair, which forces the sign bits
  uint32_t z[2] = {0, 0}, c0 = 0, c1 = 0, u[2] = {0, 0}, sa = 0, sb = 0;
  const uint32_t g = variant_bit(op, MODE);
  col(C_G) = g;
  if (cls == K_SUB || cls == K_SU || cls == K_BRU) {
    const uint32_t* a = cls == K_BRU ? xc : xb; const uint32_t* b = cls == K_BRU ? xb : xc;
    const uint32_t sgn = cls == K_SU ? g : cls == K_BRU ? 1u - g : 0u;
    if (sgn) { sa = a[1] >> 19; sb = b[1] >> 19; }
    const int32_t ta = (int32_t)a[1] + (int32_t)(sgn << 19) - (int32_t)(sa << 20), tb = (int32_t)b[1] + (int32_t)(sgn << 19) - (int32_t)(sb << 20);
    const int32_t v0 = (int32_t)a[0] - (int32_t)b[0]; c0 = v0 < 0; z[0] = (uint32_t)(v0 + (int32_t)(c0 << 20));
    const int32_t v1 = ta - tb - (int32_t)c0; c1 = v1 < 0; z[1] = (uint32_t)(v1 + (int32_t)(c1 << 20));
    if (cls != K_SUB) { u[0] = (uint32_t)ta; u[1] = (uint32_t)tb; }
  }
  col(C_SB) = sb;
  uint32_t rc2[4] = {u[0] & (RC_TABLE - 1), u[0] >> RC_BITS, u[1] & (RC_TABLE - 1), u[1] >> RC_BITS};
  // (v6) nz = [rs2 != 0] over the raw 64 bits, for EVERY row, on the sum of xc's limbs (each in range, so the sum vanishes only if all do);
  // q = "a conditional move whose condition holds" (execute.rs:434-472)
  const uint32_t sx = (uint32_t)(((uint64_t)xc[0] + xc[1] + xc[2]) % bb::P), nz = sx != 0;
  col(C_NZ) = nz; col(C_IVZ) = nz ? f_inv(sx) : 0u;
  const uint32_t q = cls == K_CMN ? nz : cls == K_CMZ ? 1u - nz : 0u;
  col(C_Q) = q;
  if ((cls == K_CMN || cls == K_CMZ) && !q) {
#pragma unroll
    for (int r = 0; r < 15; r++) col(C_WR + r) = 0;
  }
  const uint32_t flag = (cls == K_BRE || cls == K_SE) ? 1u - ne : (cls == K_BRU || cls == K_SU) ? c1 : 0u;
  const uint32_t pol = op - family_base(cls) - 2u * g;                          // op = base + 2 g + pol: 0 / 1 inside a family; the opcode itself (minus 2 g) on other rows
  const uint32_t fx = flag ? 1u - pol : pol;                                    // flag XOR pol where it matters (flag = 0 outside the families)
  col(C_FLAG) = flag; col(C_FX) = fx;
  const uint32_t tk = branch ? fx : 0;
  col(C_TK) = tk;
  const uint32_t imm17 = fc + 16 * fhi, im0 = imm17 - (s << 17) + (s << 20), im1 = s * 0xFFFFFu;
  const uint32_t lo20 = fb + 16 * fc + 256 * fhi - (s << 20);
  bool mem_row = false, lg_row = false, sh_row = false;
  uint32_t mem_z[2] = {0, 0}, mem_dt = 0, lg_a9 = 0, sh_lo[4] = {0, 0, 0, 0}, sh_c[4] = {0, 0, 0, 0};
  if (MODE >= 3) {
    // loads and stores (execute.rs:477-575): address = rs1 + sext(imm17) mod 2^64 — below 2^40, or the run has no proof here — its aligned 8-byte cell's bytes before the
    // access and the time of the cell's previous access come with the row (the host's sequential memory replay); everything else is local
#pragma unroll
    for (int k = C_E; k < W; k++) if (k != C_KLG && k != C_KSH && k != C_KMU) col(k) = 0;
    if (MODE == 4 && cls == K_WA && (xb[2] | xc[2])) {
      // (mode 4 d) an operand with bits above 40: the row goes through the WIDE TAPE (air.h) — ot = 1, every chunk column zero; what is written is the reference's result on the
      // raw 64-bit registers, which the verifier recomputes from the record (cycle, rs1, rs2, opcode) the proof carries (stark_prove.inl gathers the records from these rows)
      // — written by wide_tape_fix_row AFTER this function (its own small kernel: with the 64-bit arithmetic inlined here, hipcc 7.2 at 256 VGPRs + 90 spilled SGPRs built a
      // row kernel that read registers r6..r9 from wrong addresses on gfx950; round 6, scripts/dbg/diff_mode4.py)
      sh_row = true;
      col(C_OT) = 1;
    } else
    if (MODE == 4 && cls == K_WA) {
      // (mode 4) MULH DIVU REMU DIV REM on operands below 2^40 (execute.rs:101-183): F1 F2 + ADD = LO + 2^40 HI in 10-bit chunks (air.h: the slots of a wide-arithmetic row).
      // The top limbs of the operands must be zero — I_WA_TOP says so; a run that breaks it has no proof (lookup_index_kernel reports the row)
      sh_row = true;                                             // R0..R3 = LO's chunks, R4..R7 = F1's
      const uint64_t M40 = (1ull << 40) - 1;
      const uint64_t a = (uint64_t)xb[0] | ((uint64_t)xb[1] << 20), b = (uint64_t)xc[0] | ((uint64_t)xc[1] << 20);
      const bool mulh = op == 0x03, quot = op == 0x04 || op == 0x06;
      col(C_OM) = mulh; col(C_OD) = !mulh && quot; col(C_ORR) = !mulh && !quot;
      uint64_t f1, addv, res;
      if (mulh) {                                                // bits 40..79 of the 80-bit product, by 20-bit limbs (no 128-bit type on the device)
        const uint64_t a0 = a & 0xFFFFF, a1 = a >> 20, b0 = b & 0xFFFFF, b1 = b >> 20;
        const uint64_t mid = a0 * b1 + a1 * b0 + ((a0 * b0) >> 20);            // < 2^42
        f1 = a; addv = 0; res = (a1 * b1 + (mid >> 20)) & M40;
      } else { const uint64_t qv = b ? a / b : 0, rv = b ? a % b : 0; f1 = qv; addv = rv; res = quot ? qv : rv; }     // (rs2 = 0 never is a row: the VM stops with DivisionByZero)
      uint32_t f1c[4], f2c[4], addc[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { f1c[k] = (uint32_t)((f1 >> (10 * k)) & 1023); f2c[k] = (uint32_t)((b >> (10 * k)) & 1023); addc[k] = (uint32_t)((addv >> (10 * k)) & 1023); sh_c[k] = f1c[k]; col(C_GF + k) = f1c[k]; col(C_PIECE + k) = f2c[k]; }
      uint32_t carry = 0, cs[7], outc[7];
#pragma unroll
      for (int k = 0; k < 7; k++) {                              // position k of F1 F2 + ADD: below 2^23
        uint32_t tsum = carry + (k < 4 ? addc[k < 4 ? k : 0] : 0u);
#pragma unroll
        for (int j = 0; j < 4; j++) if (k - j >= 0 && k - j < 4) tsum += f1c[j] * f2c[k - j];
        outc[k] = tsum & 1023; carry = tsum >> 10; cs[k] = carry;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) sh_lo[k] = outc[k];
      const uint32_t g4[4] = {mulh ? outc[4] : addc[0], mulh ? outc[5] : addc[1], mulh ? outc[6] : addc[2], mulh ? cs[6] : addc[3]};
      col(C_PIECE + 4) = cs[0];
      col(C_PIECE + 5) = cs[1] & 1023; col(C_WE) = cs[1] >> 10;
      col(C_PIECE + 6) = cs[2] & 1023; col(C_WE + 1) = (cs[2] >> 10) & 1; col(C_WE + 2) = cs[2] >> 11;
      col(C_PIECE + 7) = g4[0]; col(C_PIECE + 8) = g4[1]; col(C_X) = g4[2]; col(C_X + 1) = g4[3];
      if (mulh) {
#pragma unroll
        for (int k = 0; k < 3; k++) { col(C_X + 2 + k) = cs[3 + k] & 1023; col(C_WE + 3 + 2 * k) = (cs[3 + k] >> 10) & 1; col(C_WE + 4 + 2 * k) = cs[3 + k] >> 11; }
      } else {
        const uint64_t d = (b - addv - 1) & M40;
#pragma unroll
        for (int k = 0; k < 4; k++) col(C_X + 2 + k) = (uint32_t)((d >> (10 * k)) & 1023);
        col(C_WE + 3) = ((b & 0xFFFFF) < (addv & 0xFFFFF) + 1) ? 1u : 0u;
      }
      y[0] = (uint32_t)(res & 0xFFFFF); y[1] = (uint32_t)(res >> 20); y[2] = 0;
    }
    if (cls == K_MU) {
      // MUL (execute.rs:79-99): the product of the masked operands mod 2^40, schoolbook in 10-bit chunks; the range groups are filled like a shift row's (R0..R3 = the result's
      // chunks, R4..R7 = a's), b's chunks in pieces 0-3, the carries in pieces 4-8 and e_1..3
      sh_row = true;
      const uint64_t a = (uint64_t)xb[0] | ((uint64_t)xb[1] << 20), b = (uint64_t)xc[0] | ((uint64_t)xc[1] << 20);
      uint32_t bc[4], carry = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { sh_c[k] = (uint32_t)((a >> (10 * k)) & 1023); bc[k] = (uint32_t)((b >> (10 * k)) & 1023); col(C_MA + k) = sh_c[k]; col(C_PIECE + k) = bc[k]; }
      uint64_t res = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t tsum = carry;                                   // < 4 * 2^20 + 2^12
#pragma unroll
        for (int j = 0; j <= k; j++) tsum += sh_c[j] * bc[k - j];
        sh_lo[k] = tsum & 1023; carry = tsum >> 10;
        res |= (uint64_t)sh_lo[k] << (10 * k);
        col(C_PIECE + 4 + k) = carry & 1023;
        if (k == 1) col(C_ME) = carry >> 10;
        if (k == 2) { col(C_ME + 1) = (carry >> 10) & 1; col(C_ME + 2) = carry >> 11; }
        if (k == 3) col(C_PIECE + 8) = carry >> 10;
      }
      y[0] = (uint32_t)(res & 0xFFFFF); y[1] = (uint32_t)(res >> 20); y[2] = 0;
    }
    if (cls == K_SH) {
      // SLL SRL SRA SLLI SRLI SRAI (execute.rs:284-358) as a 2^t = H 2^40 + L: t = sh on a left shift, 40 - sh on a right shift (clamped at 40 / 0: d = the rest), t = 10 u + v
      sh_row = true;
      const uint32_t which = (op - 0x18) % 3, si = (op - 0x18) / 3;
      const uint64_t a = (uint64_t)xb[0] | ((uint64_t)xb[1] << 20);
      const uint32_t amount = si ? ((w >> 15) & 0xFF) : (xc[0] & 63);
      col(C_SA) = which == 2; col(C_SI) = si; col(C_SH) = amount;
      const uint32_t tt = which == 0 ? (amount < 40 ? amount : 40u) : (amount < 40 ? 40u - amount : 0u);
      const uint32_t dd = which == 0 ? amount - tt : amount - (40u - tt);
      const uint32_t u = tt / 10, v = tt % 10;
#pragma unroll
      for (int k = 0; k < 5; k++) { col(C_UL + k) = which == 0 && u == (uint32_t)k; col(C_UR + k) = which != 0 && u == (uint32_t)k; }
#pragma unroll
      for (int k = 0; k < 10; k++) col(C_V + k) = v == (uint32_t)k;
      uint32_t hi[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sh_c[k] = (uint32_t)((a >> (10 * k)) & 1023);
        const uint32_t pr = sh_c[k] << v;
        col(C_PR + k) = pr; sh_lo[k] = pr & 1023; hi[k] = pr >> 10;
        col(C_PIECE + k) = hi[k];
      }
      const uint32_t sb9 = sh_c[3] >> 9, sgn = which == 2 ? sb9 : 0u;
      col(C_SB9) = sb9; col(C_SGN) = sgn;
      col(C_PIECE + 4) = 2 * (sh_c[3] & 511);
      col(C_PIECE + 6) = dd;
      if (si) { col(C_PIECE + 7) = fhi & 15; col(C_PIECE + 5) = fhi >> 4; }
      else { col(C_PIECE + 8) = xc[0] & 1023; col(C_PIECE + 5) = xc[0] >> 10; col(C_LB + 8) = amount; }
      const uint32_t m[5] = {sh_lo[0], sh_lo[1] + hi[0], sh_lo[2] + hi[1], sh_lo[3] + hi[2], hi[3]};
      uint64_t res = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) { const int idx = which == 0 ? j - (int)u : j + 4 - (int)u; if (idx >= 0 && idx <= 4) res |= (uint64_t)m[idx] << (10 * j); }
      if (which != 0) {
        const uint64_t ones = ((1ull << 40) - 1) & ~((1ull << tt) - 1);
        col(C_ON) = (uint32_t)(ones & 0xFFFFF); col(C_ON + 1) = (uint32_t)(ones >> 20);
        if (sgn) res |= ones;
      }
      y[0] = (uint32_t)(res & 0xFFFFF); y[1] = (uint32_t)(res >> 20); y[2] = 0;
    }
    if (cls == K_LG) {
      // AND OR XOR ANDI ORI XORI on the 40-bit values (execute.rs:199-282), nibble by nibble: a's nibbles in the piece columns (the tenth in the last range chunk), b's and the result's beside them
      lg_row = true;
      const uint32_t which = (op - 0x10) % 3, li = (op - 0x10) / 3;
      col(C_OA) = which == 0; col(C_OO) = which == 1; col(C_LI) = li;
      const uint64_t a = (uint64_t)xb[0] | ((uint64_t)xb[1] << 20), b = li ? ((uint64_t)im0 | ((uint64_t)im1 << 20)) : ((uint64_t)xc[0] | ((uint64_t)xc[1] << 20));
      const uint64_t rr = which == 0 ? (a & b) : which == 1 ? (a | b) : (a ^ b);
#pragma unroll
      for (int k = 0; k < N_NIB; k++) {
        const uint32_t ak = (uint32_t)((a >> (4 * k)) & 15);
        if (k < N_PIECE) col(C_PIECE + k) = ak; else lg_a9 = ak;
        col(C_LB + k) = (uint32_t)((b >> (4 * k)) & 15); col(C_LR + k) = (uint32_t)((rr >> (4 * k)) & 15);
      }
      y[0] = (uint32_t)(rr & 0xFFFFF); y[1] = (uint32_t)(rr >> 20); y[2] = 0;
    }
    if (cls == K_LD || cls == K_ST) {
      mem_row = true;
      const uint32_t* base = cls == K_LD ? xb : xc;            // loads: rs1 = field b; stores: rs1 = field a (operand c), the stored register rs2 = field b (operand b)
      const uint64_t v0 = (uint64_t)base[0] + im0; c0 = (uint32_t)(v0 >> 20); mem_z[0] = (uint32_t)(v0 & 0xFFFFF);
      const uint64_t v1 = (uint64_t)base[1] + im1 + c0; c1 = (uint32_t)(v1 >> 20); mem_z[1] = (uint32_t)(v1 & 0xFFFFF);
      const uint64_t v2 = (uint64_t)base[2] + (uint64_t)s * 0xFFFFFF + c1;
      col(C_CM2) = (uint32_t)(v2 >> 24);
      const uint64_t ea = (uint64_t)mem_z[0] | ((uint64_t)mem_z[1] << 20);
      const int width = mem_width(op), off = (int)(ea & 7);
      const uint64_t mask = width == 8 ? ~0ull : ((1ull << (8 * width)) - 1);
      const int v = win_of(width, off - off % width);
#pragma unroll
      for (int k = 0; k < N_WIN; k++) col(C_E + k) = k == v;
      const uint64_t ob = io->mem_old[src];
      const uint32_t told = io->mem_told[src];
#pragma unroll
      for (int k = 0; k < 8; k++) col(C_OB + k) = (uint32_t)((ob >> (8 * k)) & 0xFF);
      col(C_TOLD) = told;
      mem_dt = (uint32_t)(t.cycle[src] - told);                 // the time written, cycle + 1, is larger than the time read
      uint64_t window;
      if (cls == K_LD) {
        window = (ob >> (8 * off)) & mask;
        uint64_t val = window;
        const bool sgb = op == OP_LB, sgh = op == OP_LH;
        const uint32_t tb = width <= 2 ? (uint32_t)((window >> (8 * width - 1)) & 1) : 0u;
        if ((sgb || sgh) && tb) val |= ~mask;                    // LB / LH sign-extend to 64 bits
        col(C_SGB) = sgb; col(C_SGH) = sgh; col(C_TB) = tb; col(C_SX) = (sgb || sgh) ? tb : 0u;
        y[0] = (uint32_t)(val & 0xFFFFF); y[1] = (uint32_t)((val >> 20) & 0xFFFFF); y[2] = (uint32_t)(val >> 40);
      } else {
        window = t.registers[(uint64_t)fb * t.reg_stride + src];  // the raw 64-bit register; the store keeps its low `width` bytes
        const uint64_t val = window & mask;                      // (nothing is written: y only satisfies the window equations)
        y[0] = (uint32_t)(val & 0xFFFFF); y[1] = (uint32_t)((val >> 20) & 0xFFFFF); y[2] = (uint32_t)(val >> 40);
      }
      col(C_PIECE) = (uint32_t)(window & 0xFF); col(C_PIECE + 1) = (uint32_t)((window >> 8) & 0xFF); col(C_PIECE + 2) = (uint32_t)((window >> 16) & 0xF);
      col(C_PIECE + 3) = (uint32_t)((window >> 20) & 0xF); col(C_PIECE + 4) = (uint32_t)((window >> 24) & 0xFF); col(C_PIECE + 5) = (uint32_t)((window >> 32) & 0xFF);
      col(C_PIECE + 6) = (uint32_t)((window >> 40) & 0xFF); col(C_PIECE + 7) = (uint32_t)((window >> 48) & 0xFF); col(C_PIECE + 8) = (uint32_t)((window >> 56) & 0xFF);
      if (cls == K_LD && width <= 2) col(C_PIECE + 7) = (uint32_t)(2 * ((window >> (8 * (width - 1))) & 0x7F));   // d6 = twice the low seven bits of the top byte
      if (MODE == 4 && cls == K_ST && is_low_window(v)) {        // (mode 4) a store into the low half of a cell: not the boundary cell's — nb = delta iws = 1, delta = cell - B as a field element != 0
        const uint64_t Bc = io->boundary;
        const uint32_t delta = bb::add(bb::sub(bb::sub(mem_z[0], (uint32_t)off), (uint32_t)(Bc & 0xFFFFF)), bb::mul(1u << 20, bb::sub(mem_z[1], (uint32_t)((Bc >> 20) & 0xFFFFF))));
        if (delta) { col(C_IWS) = f_inv(delta); col(C_NB) = 1; }
      }
    }
  }
  const uint32_t dl0 = cls == K_JAL ? lo20 : tk ? im0 : 4u;
  const uint32_t se = (tk || cls == K_JAL) ? s : 0u;
  col(C_DL0) = dl0; col(C_SE) = se;
  if (cls == K_ADD || cls == K_ADDI) {
    const uint32_t b0 = cls == K_ADD ? xc[0] : im0, b1 = cls == K_ADD ? xc[1] : im1;
    const uint64_t v0 = (uint64_t)xb[0] + b0; c0 = (uint32_t)(v0 >> 20); y[0] = (uint32_t)(v0 & 0xFFFFF);
    const uint64_t v1 = (uint64_t)xb[1] + b1 + c0; c1 = (uint32_t)(v1 >> 20); y[1] = (uint32_t)(v1 & 0xFFFFF);
  } else if (cls == K_JAL || cls == K_JALR) {                      // the link pc + 4
    const uint64_t v0 = (uint64_t)pc[0] + 4; c0 = (uint32_t)(v0 >> 20); y[0] = (uint32_t)(v0 & 0xFFFFF);
    const uint64_t v1 = (uint64_t)pc[1] + c0; c1 = (uint32_t)(v1 >> 20); y[1] = (uint32_t)(v1 & 0xFFFFF);
    y[2] = pc[2] + c1;
  } else if (cls == K_SUB) { y[0] = z[0]; y[1] = z[1]; }
  else if (cls == K_SE || cls == K_SU) y[0] = fx;
  else if (cls == K_CMN || cls == K_CMZ) { y[0] = xb[0]; y[1] = xb[1]; y[2] = xb[2]; }
  col(C_Y) = y[0]; col(C_Y + 1) = y[1]; col(C_Y + 2) = y[2];
  if (cls == K_OTH) {                                              // (v6) the bits above 40 of what an "other" row writes are range-checked: y2 = R4 + 2^10 R5 + 2^20 R6, R7 = 64 R6
    rc2[0] = y[2] & (RC_TABLE - 1); rc2[1] = (y[2] >> RC_BITS) & (RC_TABLE - 1); rc2[2] = y[2] >> (2 * RC_BITS); rc2[3] = 64u * rc2[2];
  }
  if (mem_row) { rc2[0] = mem_dt & (RC_TABLE - 1); rc2[1] = (mem_dt >> RC_BITS) & (RC_TABLE - 1); rc2[2] = mem_dt >> (2 * RC_BITS); rc2[3] = 0; }   // (mode 3) cycle - told in three chunks
  if (lg_row) { rc2[0] = rc2[1] = rc2[2] = 0; rc2[3] = lg_a9; }   // (mode 3) a bitwise row's tenth nibble tuple sits in the last range slot
  if (sh_row) { rc2[0] = sh_c[0]; rc2[1] = sh_c[1]; rc2[2] = sh_c[2]; rc2[3] = sh_c[3]; }   // (mode 3) a shift row: the chunks of the shifted value
  col(C_RC2) = rc2[0]; col(C_RC2 + 1) = rc2[1]; col(C_RC2 + 2) = rc2[2]; col(C_RC2 + 3) = rc2[3];
  if (cls == K_ADD || cls == K_ADDI || cls == K_JAL || cls == K_JALR || oth_like || (MODE >= 2 && cls == K_ECALL)) { z[0] = y[0]; z[1] = y[1]; }     // the written value's low limbs are the range-checked pair
  if (mem_row) { z[0] = mem_z[0]; z[1] = mem_z[1]; }             // (mode 3) the address's two low limbs are the range-checked pair
  col(C_RC) = z[0] & (RC_TABLE - 1); col(C_RC + 1) = z[0] >> RC_BITS; col(C_RC + 2) = z[1] & (RC_TABLE - 1); col(C_RC + 3) = z[1] >> RC_BITS;   // 10-bit chunks, looked up
  if (sh_row) { col(C_RC) = sh_lo[0]; col(C_RC + 1) = sh_lo[1]; col(C_RC + 2) = sh_lo[2]; col(C_RC + 3) = sh_lo[3]; }                           // (mode 3) a shift row: the low halves of c_i 2^v
  col(C_C0) = c0; col(C_C1) = c1;
  uint32_t d0 = 0, d1 = 0, d2 = 0, b0 = 0;
  if (cls != K_JALR && cls != K_OJ && cls != K_HALT && cls != K_PAD) b0 = sa;   // (v5) column b0 doubles as the sign bit of the first operand of an ordered comparison
  if (cls == K_JALR) {                                             // pc' + b0 = rs1 + sext(imm17) over (20, 20, 24)-bit limbs, mod 2^64
    const uint64_t v0 = (uint64_t)xb[0] + im0; d0 = (uint32_t)(v0 >> 20); b0 = (uint32_t)(v0 & 1);
    const uint64_t v1 = (uint64_t)xb[1] + im1 + d0; d1 = (uint32_t)(v1 >> 20);
    const uint64_t v2 = (uint64_t)xb[2] + (uint64_t)s * 0xFFFFFF + d1; d2 = (uint32_t)(v2 >> 24);
  } else if (cls != K_OJ && cls != K_HALT && cls != K_PAD) {
    const uint64_t v0 = (uint64_t)pc[0] + dl0; d0 = (uint32_t)(v0 >> 20);
    const uint64_t v1 = (uint64_t)pc[1] + (uint64_t)se * 0xFFFFF + d0; d1 = (uint32_t)(v1 >> 20);
    const uint64_t v2 = (uint64_t)pc[2] + (uint64_t)se * 0xFFFFFF + d1; d2 = (uint32_t)(v2 >> 24);
  }
  col(C_D0) = d0; col(C_D1) = d1; col(C_D2) = d2; col(C_B0) = b0;
  // the committed columns, packed: committed position p holds logical column logical_col(p); the tail of the last block is zero padding
  uint4* out4 = reinterpret_cast<uint4*>(out);
  auto at = [&](int p) -> uint32_t { return p < committed_used(MODE) ? rowv[logical_col(p < committed_used(MODE) ? p : 0, MODE)] : 0u; };   // (inner clamp: the index stays inside rowv for the padding positions too)
#pragma unroll
  for (int b = SKIP; b < committed_width(MODE) / 8; b++) {
    out4[((uint64_t)b * N + i) * 2] = make_uint4(at(8 * b), at(8 * b + 1), at(8 * b + 2), at(8 * b + 3));
    out4[((uint64_t)b * N + i) * 2 + 1] = make_uint4(at(8 * b + 4), at(8 * b + 5), at(8 * b + 6), at(8 * b + 7));
  }
}
template <int MODE, int SKIP = 0>
__global__ __launch_bounds__(NT, MODE >= 3 ? MT_WAVES_MEM : MT_WAVES) void main_trace_kernel(zkir_trace_columns t, uint64_t n_real, uint64_t N, uint32_t* __restrict__ out, IoRowArgs io = IoRowArgs{}) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= N) return;
  main_trace_row<MODE, SKIP>(t, n_real, N, i, out, &io);
}

// ---- (mode 2) prefix counts of the WRITE / READ ecalls over the rows: flags -> exclusive scan (three launches: per-block scan, scan of the block totals, add) ----
BB_HD void io_row_flags(const zkir_trace_columns& t, uint64_t n_real, uint64_t i, uint32_t f[2]) {
  f[0] = f[1] = 0;
  if (i + 1 >= n_real) return;                                 // the halt row and the padding execute nothing (an exit ECALL is the halt row)
  if ((t.instruction[i] & 0x7F) != air::OP_ECALL) return;
  const uint64_t num = t.registers[(uint64_t)10 * t.reg_stride + i];
  f[0] = num == 2; f[1] = num == 1;
}
constexpr uint32_t IOS_ROWS = 1024;                            // rows per workgroup of the scan (4 per lane)
__global__ __launch_bounds__(NT) void io_scan_local_kernel(zkir_trace_columns t, uint64_t n_real, uint64_t N, uint2* __restrict__ cnt, uint2* __restrict__ sums) {
  __shared__ uint2 lds[NT];
  const uint64_t base = (uint64_t)blockIdx.x * IOS_ROWS + (uint64_t)threadIdx.x * 4;
  uint2 v[4], run = make_uint2(0, 0);
#pragma unroll
  for (int k = 0; k < 4; k++) { uint32_t f[2] = {0, 0}; if (base + k < N) io_row_flags(t, n_real, base + k, f); v[k] = run; run.x += f[0]; run.y += f[1]; }
  lds[threadIdx.x] = run;
  __syncthreads();
  for (uint32_t off = 1; off < NT; off <<= 1) {
    uint2 x = lds[threadIdx.x];
    if (threadIdx.x >= off) { const uint2 y = lds[threadIdx.x - off]; x.x += y.x; x.y += y.y; }
    __syncthreads();
    lds[threadIdx.x] = x;
    __syncthreads();
  }
  const uint2 ex = threadIdx.x ? lds[threadIdx.x - 1] : make_uint2(0, 0);
#pragma unroll
  for (int k = 0; k < 4; k++) if (base + k < N) cnt[base + k] = make_uint2(v[k].x + ex.x, v[k].y + ex.y);
  if (threadIdx.x == NT - 1) sums[blockIdx.x] = lds[NT - 1];
}
// exclusive scan of the n = N / 1024 block totals (<= 65536), one workgroup: every lane sums its contiguous share, takes the totals of the lanes before it from LDS and
// writes its share's prefixes (one lane walking all of them took 118 us at 2^20 rows: a tenth of a mode-2 proof's main-trace stage)
__global__ __launch_bounds__(NT) void io_scan_sums_kernel(uint2* __restrict__ sums, uint32_t n) {
  __shared__ uint2 tot[NT];
  const uint32_t per = (n + NT - 1) / NT, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  uint2 a = make_uint2(0, 0);
  for (uint32_t b = lo; b < hi; b++) { const uint2 v = sums[b]; a.x += v.x; a.y += v.y; }
  tot[threadIdx.x] = a;
  __syncthreads();
  uint2 run = make_uint2(0, 0);
  for (uint32_t j = 0; j < threadIdx.x; j++) { run.x += tot[j].x; run.y += tot[j].y; }
  for (uint32_t b = lo; b < hi; b++) { const uint2 v = sums[b]; sums[b] = run; run.x += v.x; run.y += v.y; }
}
__global__ __launch_bounds__(NT) void io_scan_add_kernel(uint2* __restrict__ cnt, uint64_t N, const uint2* __restrict__ sums) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= N) return;
  const uint2 o = sums[i / IOS_ROWS];
  cnt[i].x += o.x; cnt[i].y += o.y;
}

// ------------------------------------------------------------------------------------------------
// NTT
// ------------------------------------------------------------------------------------------------
// tw[k] = w^k in Montgomery form, k < count; w canonical
__global__ __launch_bounds__(NT) void powers_kernel(uint32_t w, uint32_t scale_m, uint32_t* __restrict__ tw, uint32_t count) {
  const uint32_t k = blockIdx.x * NT + threadIdx.x;
  if (k >= count) return;
  uint32_t r = scale_m, b = bb::to_mont(w), e = k;
  while (e) { if (e & 1) r = bb::mont_mul(r, b); b = bb::mont_mul(b, b); e >>= 1; }
  tw[k] = r;
}


// 1 / (x_j - 1) over the LDE coset x_j = g w_2N^j (Montgomery), eight points per lane: one inversion per eight (Montgomery's trick).
// Depends on log_n only: a table of the context.  x_j = 1 cannot happen (g generates the whole group, so g w^j is never in the 2-power subgroup).
__global__ __launch_bounds__(NT) void selector_inverse_kernel(uint32_t log_n, const uint32_t* __restrict__ tw_fwd, uint32_t* __restrict__ inv_xm1) {
  const uint32_t N2 = 2u << log_n, N = N2 >> 1;
  const uint32_t j0 = (blockIdx.x * NT + threadIdx.x) * 8;
  if (j0 >= N2) return;
  uint32_t d[8], pre[8];
  uint32_t run = bb::R1;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t j = j0 + k;
    const uint32_t wj = j < N ? tw_fwd[j] : bb::neg(tw_fwd[j - N]);
    d[k] = bb::sub(bb::mont_mul(wj, bb::to_mont(bb::GEN)), bb::R1);
    pre[k] = run; run = bb::mont_mul(run, d[k]);
  }
  uint32_t inv = bb::R1, b = run, e = bb::P - 2;
  while (e) { if (e & 1) inv = bb::mont_mul(inv, b); b = bb::mont_mul(b, b); e >>= 1; }
#pragma unroll
  for (int k = 7; k >= 0; k--) { inv_xm1[j0 + k] = bb::mont_mul(inv, pre[k]); inv = bb::mont_mul(inv, d[k]); }
}


// ------------------------------------------------------------------------------------------------
// Poseidon2 Merkle
// ------------------------------------------------------------------------------------------------
// one lane per leaf (= row position); a B8 block is exactly one absorption of the rate-8 sponge: two 16-byte loads per permutation
// in_scale: cp->in_scale for canonical matrix words; from_mont(cp->in_scale) when the matrix rests in Montgomery form (the prover's LDE matrices): the
// factor R the words carry is divided out by the multiplication that brings them to the sponge's input scale — same instructions, same digests.
__global__ __launch_bounds__(NT) void leaf_hash_kernel(const p2::Consts* __restrict__ cp, const uint32_t* __restrict__ mat, uint32_t width, uint64_t n, uint32_t in_scale, uint32_t* __restrict__ digests) {
  const uint64_t j = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (j >= n) return;
  uint32_t s[p2::T];
#pragma unroll
  for (int i = 0; i < p2::T; i++) s[i] = 0;
  const uint4* m4 = reinterpret_cast<const uint4*>(mat);
  // p2::permute_scaled: absorbed values enter with the factor in_scale, words that stay (the capacity; the tail of the rate in a
  // ragged last block, which so::hash_elems leaves in place) are carried over from the previous output with `carry`
  const uint32_t carry = cp->carry, out_scale = cp->out_scale;
  for (uint32_t off = 0; off < width; off += p2::RATE) {
    const uint4 lo = m4[((uint64_t)(off >> 3) * n + j) * 2], hi = m4[((uint64_t)(off >> 3) * n + j) * 2 + 1];
    const uint32_t v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 0; i < p2::RATE; i++) s[i] = off + i < width ? bb::mont_mul_lazy(v[i], in_scale) : bb::mont_mul_lazy(s[i], carry);
#pragma unroll
    for (int i = p2::RATE; i < p2::T; i++) s[i] = bb::mont_mul_lazy(s[i], carry);
    p2::permute_scaled(s, *cp);
  }
  if (width == 0) p2::permute_scaled(s, *cp);
  uint4 d = make_uint4(bb::mont_mul(s[0], out_scale), bb::mont_mul(s[1], out_scale), bb::mont_mul(s[2], out_scale), bb::mont_mul(s[3], out_scale));
  reinterpret_cast<uint4*>(digests)[j] = d;
}

// EXPERIMENT (round 6): the leaf sponge BLOCK-WISE — blocks [b0, b1) of the matrix are absorbed into a 12-word sponge state kept per leaf in global memory
// (three 16-byte vectors per leaf, [3][n] so that a wave's accesses are contiguous), so that the absorption of a group of blocks can run while the NEXT group is still being
// extended (zkir_commit_overlapped_launch).  Same permutations on the same words in the same order as leaf_hash_kernel: same digests.  first: the state starts at zero;
// last: the digest is written instead of the state.  The state words are the kernel's scaled post-permutation words (the carry product is applied on the way back in).
__global__ __launch_bounds__(NT) void leaf_absorb_kernel(const p2::Consts* __restrict__ cp, const uint32_t* __restrict__ mat, uint32_t width, uint32_t b0, uint32_t b1, uint64_t n, uint32_t in_scale,
                                                         uint4* __restrict__ state, int first, int last, uint32_t* __restrict__ digests) {
  const uint64_t j = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (j >= n) return;
  uint32_t s[p2::T];
  if (first) {
#pragma unroll
    for (int i = 0; i < p2::T; i++) s[i] = 0;
  } else {
    const uint4 a = state[j], b = state[n + j], c = state[2 * n + j];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w; s[8] = c.x; s[9] = c.y; s[10] = c.z; s[11] = c.w;
  }
  const uint4* m4 = reinterpret_cast<const uint4*>(mat);
  const uint32_t carry = cp->carry, out_scale = cp->out_scale;
  for (uint32_t blk = b0; blk < b1; blk++) {
    const uint32_t off = blk * 8;
    const uint4 lo = m4[((uint64_t)blk * n + j) * 2], hi = m4[((uint64_t)blk * n + j) * 2 + 1];
    const uint32_t v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 0; i < p2::RATE; i++) s[i] = off + i < width ? bb::mont_mul_lazy(v[i], in_scale) : bb::mont_mul_lazy(s[i], carry);
#pragma unroll
    for (int i = p2::RATE; i < p2::T; i++) s[i] = bb::mont_mul_lazy(s[i], carry);
    p2::permute_scaled(s, *cp);
  }
  if (last) {
    reinterpret_cast<uint4*>(digests)[j] = make_uint4(bb::mont_mul(s[0], out_scale), bb::mont_mul(s[1], out_scale), bb::mont_mul(s[2], out_scale), bb::mont_mul(s[3], out_scale));
  } else {
    state[j] = make_uint4(s[0], s[1], s[2], s[3]); state[n + j] = make_uint4(s[4], s[5], s[6], s[7]); state[2 * n + j] = make_uint4(s[8], s[9], s[10], s[11]);
  }
}

__global__ __launch_bounds__(NT) void compress_kernel(const p2::Consts* __restrict__ cp, const uint32_t* __restrict__ in, uint64_t n_out, uint32_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n_out) return;
  const uint4 l = reinterpret_cast<const uint4*>(in)[2 * i], r = reinterpret_cast<const uint4*>(in)[2 * i + 1];
  const uint32_t k = cp->in_scale, ko = cp->out_scale;
  uint32_t s[p2::T] = {bb::mont_mul_lazy(l.x, k), bb::mont_mul_lazy(l.y, k), bb::mont_mul_lazy(l.z, k), bb::mont_mul_lazy(l.w, k), bb::mont_mul_lazy(r.x, k), bb::mont_mul_lazy(r.y, k),
                       bb::mont_mul_lazy(r.z, k), bb::mont_mul_lazy(r.w, k), 0, 0, 0, 0};
  p2::permute_scaled(s, *cp);
  reinterpret_cast<uint4*>(out)[i] = make_uint4(bb::mont_mul(s[0], ko), bb::mont_mul(s[1], ko), bb::mont_mul(s[2], ko), bb::mont_mul(s[3], ko));
}

// Subtrees in ONE launch: every workgroup takes `per_wg` (a power of two <= 1024) consecutive digests of the level `cur` (m digests)
// and walks log2(per_wg) levels up, keeping the running level in LDS (Montgomery form) and writing every level to its place in the
// tree buffer.  Small levels are latency-bound (one Poseidon2 permutation is ~1200 dependent instructions deep), so a launch and
// a global-memory round trip per level cost more than the hashing.  Round 5: the workgroup that FINISHES LAST (a device-scope counter, left at zero again) goes on with
// the m / per_wg (<= 256) digests the launch produced, so everything above the last full-occupancy level is one launch (round 4: two, 17 per proof).
constexpr uint32_t SUBTREE = 1024, NT_SUB = 1024;                 // digests / lanes of a workgroup: 64 rows of 16 lanes = 64 permutations a pass in the row formulation
__global__ __launch_bounds__(NT_SUB) void subtree_kernel(const p2::Consts* __restrict__ cp, uint32_t* cur, uint64_t m, uint32_t per_wg, uint32_t* counter) {
  __shared__ uint4 buf[SUBTREE];
  __shared__ uint32_t last_one;
  const uint32_t t = threadIdx.x;
  uint64_t pos0 = (uint64_t)blockIdx.x * per_wg;
  for (uint32_t e = t; e < per_wg; e += NT_SUB) {
    const uint4 v = reinterpret_cast<const uint4*>(cur)[pos0 + e];
    buf[e] = make_uint4(bb::to_mont(v.x), bb::to_mont(v.y), bb::to_mont(v.z), bb::to_mont(v.w));
  }
  __syncthreads();
 levels:
  for (uint32_t cnt = per_wg; cnt > 1; cnt >>= 1) {
    cur += 4 * m; m >>= 1; pos0 >>= 1;                         // level written by this iteration
    const uint32_t n_perm = cnt / 2;
    if (n_perm > NT_SUB / 16) {                                // more permutations than rows of lanes: one per LANE, the throughput formulation (p2::permute_scaled)
      const bool active = t < n_perm;
      uint32_t s[p2::T];
      if (active) {
        const uint4 l = buf[2 * t], r = buf[2 * t + 1];        // Montgomery words R v -> input words F_IN v: one product with F_IN
        const uint32_t k_in = bb::from_mont(cp->in_scale);
        s[0] = bb::mont_mul_lazy(l.x, k_in); s[1] = bb::mont_mul_lazy(l.y, k_in); s[2] = bb::mont_mul_lazy(l.z, k_in); s[3] = bb::mont_mul_lazy(l.w, k_in);
        s[4] = bb::mont_mul_lazy(r.x, k_in); s[5] = bb::mont_mul_lazy(r.y, k_in); s[6] = bb::mont_mul_lazy(r.z, k_in); s[7] = bb::mont_mul_lazy(r.w, k_in);
        s[8] = s[9] = s[10] = s[11] = 0;
      }
      __syncthreads();                                         // all inputs read before slot t is overwritten
      if (active) {
        p2::permute_scaled(s, *cp);
        const uint32_t ko = cp->out_scale, ko_m = bb::to_mont(ko);                     // output words F_OUT v -> canonical v (the tree) and R v (the next level)
        buf[t] = make_uint4(bb::mont_mul(s[0], ko_m), bb::mont_mul(s[1], ko_m), bb::mont_mul(s[2], ko_m), bb::mont_mul(s[3], ko_m));
        reinterpret_cast<uint4*>(cur)[pos0 + t] = make_uint4(bb::mont_mul(s[0], ko), bb::mont_mul(s[1], ko), bb::mont_mul(s[2], ko), bb::mont_mul(s[3], ko));
      }
    } else {                                                   // at most 64 permutations: one per ROW of 16 lanes (p2::permute_row16_scaled): the shortest chain — these levels wait for each other
      const uint32_t pi = t >> 4, l = t & 15;
      const bool active = pi < n_perm;                         // (whole rows)
      const uint32_t* words = reinterpret_cast<const uint32_t*>(buf);
      uint32_t s = 0;
      if (active && l < 8) s = bb::mont_mul_lazy(words[8 * pi + l], bb::from_mont(cp->in_scale));       // left digest = words 0-3, right digest = words 4-7, capacity zero
      __syncthreads();
      if (active) {
        s = p2::permute_row16_scaled(s, (int)l, *cp);
        if (l < 4) {
          const uint32_t ko = cp->out_scale;
          reinterpret_cast<uint32_t*>(buf)[4 * pi + l] = bb::mont_mul(s, bb::to_mont(ko));
          cur[4 * (pos0 + pi) + l] = bb::mont_mul(s, ko);
        }
      }
    }
    __syncthreads();
  }
  // `cur` is now the level this launch leaves (m digests, one per workgroup).  The last workgroup to get here takes all of them on.  No fence: a device-scope fence
  // on gfx950 writes back and invalidates the whole L2 of the XCD (buffer_wbl2 / buffer_inv; measured here: 512 workgroups doing so took the launch from 94 to 146 us).
  // Instead the ONE digest a workgroup hands over is re-stored with device-scope atomic stores (write-through), the counter is bumped once those have been
  // acknowledged (s_waitcnt vmcnt(0) in the same wave), and the workgroup that goes on reads the digests with device-scope atomic loads (past the L2 of its XCD).
  // Ordering, spelled out: the hardware side is the gfx9 rule that vmcnt counts stores as well as loads and that the immediate 0x0F70 is vmcnt(0) with every
  // other counter left alone — both true of gfx950 only, hence the #error below; the compiler side is the pair of `asm volatile("" ::: "memory")` barriers, which keep the
  // digest stores, the wait and the counter bump (and, on the winner's side, the counter read and the digest loads) in program order.  `counter` is THIS LAUNCH's slot
  // (launch_tree_levels hands out a fresh one of a ring per launch), so launches that overlap on two streams of one context never see each other's increments.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "subtree_kernel's hand-over relies on the gfx950 (gfx9) s_waitcnt encoding and on vmcnt covering stores"
#endif
  if (m > 1) {
    if (t < 4) {
      const uint32_t v = reinterpret_cast<const uint32_t*>(buf)[t];          // buf[0] = the workgroup's digest in Montgomery form; the tree holds it canonical
      __hip_atomic_store(&cur[4 * pos0 + t], bb::from_mont(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0)
      asm volatile("" ::: "memory");
    }
    if (t == 0) {
      const uint32_t done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::: "memory");
      last_one = done == (uint32_t)m - 1;
      if (last_one) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // every increment of this launch has happened: the slot is clean for its next turn of the ring
    }
    __syncthreads();
    if (!last_one) return;
    asm volatile("" ::: "memory");
    per_wg = (uint32_t)m; pos0 = 0;
    for (uint32_t e = t; e < 4 * per_wg; e += NT_SUB)
      reinterpret_cast<uint32_t*>(buf)[e] = bb::to_mont(__hip_atomic_load(&cur[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    goto levels;
  }
}

// ALU roofline probe: every lane runs `iters` rounds of 8 independent MINIMAL Montgomery products (64-bit multiply, low multiply,
// 64-bit multiply-add: three multiplier-pipe instructions, no final subtraction, no memory traffic): the rate at which the chip's
// integer multipliers can turn out 31-bit modular products at all.  The Poseidon2 kernels are priced against it; the share of their
// time that is NOT such products is the additions, 64-bit accumulations and reductions of the linear layers.
__global__ __launch_bounds__(NT) void modmul_peak_kernel(uint32_t* __restrict__ out, uint32_t iters) {
  uint32_t a[8], c[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { a[k] = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + k * 7919u) % bb::P; c[k] = (a[k] * 31u + 7u) % bb::P; }
  for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = bb::mont_mul_lazy(a[k], c[k]);    // the 3-instruction product (stays below 2p), as the hash kernels use it
  }
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) r ^= a[k];
  if (r == 0xFFFFFFFFu && iters == 0xFFFFFFFFu) out[0] = r;  // never true; keeps the chain live
}

// The "who finishes last" counters of subtree_kernel: a ring of zeroed words, ONE SLOT PER LAUNCH (one counter per context was shared by launches that
// overlap when a context's trees are built on two streams — service.py's commit_only pipeline, any caller of zkir_merkle_commit_launch / zkir_merkle_cap_launch with streams
// of its own — and a workgroup could then see "I am last" on another launch's increments).  A slot is left at zero by the launch that used it and comes round again after
// SYNC_SLOTS further subtree launches on the context; a context never has that many trees in flight (a proof enqueues ten, one after the other).
constexpr uint32_t SYNC_SLOTS = 1024;
struct SyncRing {
  uint32_t* d_slots = nullptr;
  std::atomic<uint32_t> next{0};
  uint32_t* take() { return d_slots + (next.fetch_add(1, std::memory_order_relaxed) % SYNC_SLOTS); }
};

// all levels above the leaf digests: wide levels (throughput-bound) one launch each, then subtree launches
void launch_tree_levels(const p2::Consts* cp, uint32_t* leaf_digests, uint64_t n_leaves, SyncRing& ring, hipStream_t s) {
  uint32_t* cur = leaf_digests;
  uint64_t m = n_leaves;
  for (; m > (1u << 18); m >>= 1) {
    uint32_t* nxt = cur + 4 * m;
    hipLaunchKernelGGL(compress_kernel, dim3(grid_for(m / 2)), dim3(NT), 0, s, cp, cur, m / 2, nxt);
    cur = nxt;
  }
  if (m > 1) {                                                   // m <= 2^18: per <= 1024 digests a workgroup, at most 256 workgroups, whose digests the last of them finishes
    const uint32_t per = m < SUBTREE ? (uint32_t)m : SUBTREE;
    hipLaunchKernelGGL(subtree_kernel, dim3((unsigned)(m / per)), dim3(NT_SUB), 0, s, cp, cur, m, per, ring.take());
  }
}

}  // namespace

// ================================================================================================
// context + C ABI
// ================================================================================================
struct zkir_stark_ctx {
  uint32_t log_n = 0, log_blowup = 1;
  uint32_t* d_tw_inv = nullptr;   // w_N^-k, k < N/2
  uint32_t* d_tw_fwd = nullptr;   // w_{2N}^k, k < N
  uint32_t* d_g_lo = nullptr;     // g^k / N, k < 1024
  uint32_t* d_g_lo_m = nullptr;   // R g^k / N: the same scale with the Montgomery factor folded in — the LDE then leaves its output in Montgomery form (zkir_prove)
  uint32_t* d_inv_xm1 = nullptr;  // 1 / (x_j - 1), j < 2N, over the LDE coset (Montgomery): the row selectors of the quotient (stark_prove.inl)
  uint32_t* d_g_hi = nullptr;     // g^(1024 k)
  uint32_t* d_small_inv = nullptr;  // w_{2^Bm}^-k, k < 2^(Bm-1)      (Bm = min(log_n, 10))
  uint32_t* d_small_fwd = nullptr;  // w_{2^(Bm+1)}^k, k < 2^Bm
  p2::Consts consts;              // host copy (transcript, verifier side)
  p2::Consts* d_p2 = nullptr;     // device copy: every hash kernel takes the pointer (no process-wide __constant__ state)
  // zkir_commit_overlapped_launch (experiment): a second stream, two events and the per-leaf sponge states, made on first use
  mutable hipStream_t ov_stream = nullptr; mutable hipEvent_t ov_ev[2] = {nullptr, nullptr}; mutable std::vector<hipEvent_t> ov_group_ev; mutable uint4* d_ov_state = nullptr; mutable size_t ov_state_bytes = 0;
  mutable SyncRing sync;          // subtree_kernel's "who finishes last" counters, one slot per launch (launches of one context may overlap on the caller's streams)
  mutable std::mutex mu;          // a context serves one proof at a time (its workspace arena); different contexts are independent
  // prover workspace: one device allocation made on the first zkir_prove and reused (hipMalloc of GBs costs more than the kernels)
  mutable unsigned char* arena = nullptr;
  mutable size_t arena_size = 0, arena_off = 0;
  mutable zkir::HostPin pin;      // pinned host staging of the proof's large copies (host.h: why no large pageable block is handed to a copy)
};

namespace {
// The LDE with the scale table chosen by the form its output is to rest in: mont_out = the words of `out` carry the Montgomery factor R (the
// prover's matrices; the scale g^k / N of the fused middle pass comes from the table that has R folded in — same kernels, same instruction count).
int lde_launch(const zkir_stark_ctx* c, uint32_t* in, uint32_t width, uint32_t* out, bool mont_out, hipStream_t s) {
  const zkir::LdeTables t{(int)c->log_n, c->d_tw_inv, c->d_tw_fwd, mont_out ? c->d_g_lo_m : c->d_g_lo, c->d_g_hi, c->d_small_inv, c->d_small_fwd};
  zkir::lde_run(t, in, (width + 7) / 8, out, s);               // ntt.hip
  return check_launch("lde");
}
// leaf layer + the levels above it; mont_in = the matrix words carry the Montgomery factor (the digests are those of the canonical words either way)
int merkle_commit(const zkir_stark_ctx* c, const uint32_t* mat, uint32_t width, uint64_t n_leaves, uint32_t* tree, bool mont_in, hipStream_t s) {
  hipLaunchKernelGGL(leaf_hash_kernel, dim3(grid_for(n_leaves)), dim3(NT), 0, s, c->d_p2, mat, width, n_leaves, mont_in ? bb::from_mont(c->consts.in_scale) : c->consts.in_scale, tree);
  launch_tree_levels(c->d_p2, tree, n_leaves, c->sync, s);
  return check_launch("merkle_commit");
}
}  // namespace

extern "C" {

uint32_t zkir_main_trace_width(void) { return air::committed_width(false); }
uint32_t zkir_main_trace_width_for(uint32_t deferred) { return air::committed_width((int)deferred); }     // `deferred` = the mode: 152 / 168 / 160

void zkir_poseidon2_permute(uint32_t state[12]) {
  static const p2::Consts consts = [] { p2::Consts c; p2::generate(c); return c; }();
  uint32_t s[p2::T];
  for (int i = 0; i < p2::T; i++) s[i] = bb::to_mont(state[i] % bb::P);
  p2::permute(s, consts);
  for (int i = 0; i < p2::T; i++) state[i] = bb::from_mont(s[i]);
}

// The throughput formulation the hash kernels run (p2::permute_scaled: scaled state words, Montgomery-reduced linear layers),
// host build, canonical words in and out, `rounds` chained applications with the carry step a sponge performs between them:
// lets the CPU suite pin it against the oracle without a device.
void zkir_poseidon2_permute_scaled(uint32_t state[12], uint32_t rounds) {
  static const p2::Consts consts = [] { p2::Consts c; p2::generate(c); return c; }();
  uint32_t s[p2::T];
  for (int i = 0; i < p2::T; i++) s[i] = bb::mont_mul_lazy(state[i] % bb::P, consts.in_scale);
  for (uint32_t r = 0; r < rounds; r++) {
    if (r) for (int i = 0; i < p2::T; i++) s[i] = bb::mont_mul_lazy(s[i], consts.carry);
    p2::permute_scaled(s, consts);
  }
  for (int i = 0; i < p2::T; i++) state[i] = rounds ? bb::mont_mul(s[i], consts.out_scale) : state[i] % bb::P;
}

// Diagnostic: measured peak Montgomery-multiplication rate (modmul/s) of the device, used as the ALU roofline of the Poseidon2 kernels.
// HBM copy probe: 16 bytes per lane (MI355X_MICROARCH.md's float4 copy: ~6.3 TB/s of the nominal 8).  Four loads in flight per lane before the first store; the best of a few
// grid sizes is reported (the figure is a property of the device, not of one launch geometry).
namespace { typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(NT) void copy16_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * NT;
  uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride), c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride); __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
  }
  for (; i < n; i += stride) dst[i] = src[i];
} }
double zkir_hbm_copy_peak_gbs(uint64_t bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint64_t n = (bytes >> 14) << 10;                      // uint4 elements, a multiple of 1024
  if (!n) return 0.0;
  u32x4 *a = nullptr, *b = nullptr;
  if (hipMalloc((void**)&a, n * 16) != hipSuccess) return 0.0;
  if (hipMalloc((void**)&b, n * 16) != hipSuccess) { (void)hipFree(a); return 0.0; }
  (void)hipMemsetAsync(a, 1, n * 16, s);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double best = 0.0;
  for (unsigned want : {1024u, 2048u, 4096u, 16384u, 65536u}) {
    const unsigned blocks = (unsigned)((n / NT) < want ? (n / NT) : want);
    hipLaunchKernelGGL(copy16_kernel, dim3(blocks), dim3(NT), 0, s, a, b, n);
    const int reps = 6;
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(copy16_kernel, dim3(blocks), dim3(NT), 0, s, a, b, n);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double gbs = ms > 0 ? (double)reps * 2.0 * (double)n * 16.0 / (ms * 1e-3) / 1e9 : 0.0;
    if (gbs > best) best = gbs;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(a); (void)hipFree(b);
  return best;
}

double zkir_modmul_peak_per_s(void* stream) {
  hipStream_t s = (hipStream_t)stream;
  uint32_t* d = nullptr;
  if (hipMalloc(&d, 256) != hipSuccess) return 0.0;
  const uint32_t blocks = 256 * 16, iters = 4096;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(modmul_peak_kernel, dim3(blocks), dim3(NT), 0, s, d, 64u);
  (void)hipEventRecord(a, s);
  hipLaunchKernelGGL(modmul_peak_kernel, dim3(blocks), dim3(NT), 0, s, d, iters);
  (void)hipEventRecord(b, s);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipFree(d);
  return ms > 0 ? (double)blocks * NT * iters * 8.0 / (ms * 1e-3) : 0.0;
}

int zkir_stark_ctx_create(uint32_t log_n, uint32_t log_blowup, zkir_stark_ctx** out) {
  if (!out || log_n < 1 || log_n > 26 || log_blowup != 1) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_stark_ctx_create: need 1 <= log_n <= 26 and log_blowup == 1"}); return ZKIR_ERR_ARGUMENT; }
  *out = nullptr;
  zkir_stark_ctx* c = new zkir_stark_ctx();
  c->log_n = log_n; c->log_blowup = log_blowup;
  const uint32_t N = 1u << log_n;
  const uint32_t n_inv = N >= 2 ? N / 2 : 1, n_hi = (N >> 10) + 1;
  p2::generate(c->consts);
  hipError_t e = hipMalloc((void**)&c->d_p2, sizeof(p2::Consts));
  if (e == hipSuccess) e = hipMemcpy(c->d_p2, &c->consts, sizeof(p2::Consts), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&c->sync.d_slots, SYNC_SLOTS * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(c->sync.d_slots, 0, SYNC_SLOTS * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&c->d_tw_inv, (size_t)n_inv * 4);
  if (e == hipSuccess) e = hipMalloc(&c->d_tw_fwd, (size_t)N * 4);
  if (e == hipSuccess) e = hipMalloc(&c->d_g_lo, 1024 * 4);
  if (e == hipSuccess) e = hipMalloc(&c->d_g_hi, (size_t)n_hi * 4);
  if (e == hipSuccess) e = hipMalloc(&c->d_g_lo_m, 1024 * 4);
  if (e == hipSuccess) e = hipMalloc(&c->d_inv_xm1, (size_t)2 * N * 4);
  const int Bm = log_n < 10 ? (int)log_n : 10;
  const uint32_t n_si = Bm >= 1 ? (1u << (Bm - 1)) : 1, n_sf = 1u << Bm;
  if (e == hipSuccess) e = hipMalloc(&c->d_small_inv, (size_t)n_si * 4);
  if (e == hipSuccess) e = hipMalloc(&c->d_small_fwd, (size_t)n_sf * 4);
  if (e != hipSuccess) { zkir::set_last_error({ZKIR_ERR_DEVICE, std::string("zkir_stark_ctx_create: ") + hipGetErrorString(e)}); zkir_stark_ctx_free(c); return ZKIR_ERR_DEVICE; }
  const uint32_t wN = bb::root_of_unity(log_n), w2N = bb::root_of_unity(log_n + 1);
  hipLaunchKernelGGL(powers_kernel, dim3(grid_for(n_inv)), dim3(NT), 0, 0, bb::inv(wN), bb::R1, c->d_tw_inv, n_inv);
  hipLaunchKernelGGL(powers_kernel, dim3(grid_for(N)), dim3(NT), 0, 0, w2N, bb::R1, c->d_tw_fwd, N);
  hipLaunchKernelGGL(powers_kernel, dim3(4), dim3(NT), 0, 0, bb::GEN, bb::to_mont(bb::inv(N % bb::P)), c->d_g_lo, 1024u);
  hipLaunchKernelGGL(powers_kernel, dim3(4), dim3(NT), 0, 0, bb::GEN, bb::to_mont(bb::to_mont(bb::inv(N % bb::P))), c->d_g_lo_m, 1024u);
  hipLaunchKernelGGL(powers_kernel, dim3(grid_for(n_hi)), dim3(NT), 0, 0, bb::pow(bb::GEN, 1024), bb::R1, c->d_g_hi, n_hi);
  hipLaunchKernelGGL(selector_inverse_kernel, dim3(grid_for((2ull * N + 7) / 8)), dim3(NT), 0, 0, log_n, c->d_tw_fwd, c->d_inv_xm1);
  hipLaunchKernelGGL(powers_kernel, dim3(grid_for(n_si)), dim3(NT), 0, 0, bb::inv(bb::root_of_unity(Bm)), bb::R1, c->d_small_inv, n_si);
  hipLaunchKernelGGL(powers_kernel, dim3(grid_for(n_sf)), dim3(NT), 0, 0, bb::root_of_unity(Bm + 1), bb::R1, c->d_small_fwd, n_sf);
  if (hipDeviceSynchronize() != hipSuccess || check_launch("stark ctx tables") != ZKIR_OK) { zkir_stark_ctx_free(c); return ZKIR_ERR_DEVICE; }
  *out = c;
  return ZKIR_OK;
}

void zkir_stark_ctx_free(zkir_stark_ctx* c) {
  if (!c) return;
  (void)hipFree(c->d_tw_inv); (void)hipFree(c->d_tw_fwd); (void)hipFree(c->d_g_lo); (void)hipFree(c->d_g_hi); (void)hipFree(c->d_g_lo_m); (void)hipFree(c->d_inv_xm1);
  (void)hipFree(c->d_small_inv); (void)hipFree(c->d_small_fwd); (void)hipFree(c->d_p2); (void)hipFree(c->sync.d_slots);
  if (c->ov_stream) (void)hipStreamDestroy(c->ov_stream);
  for (hipEvent_t e : c->ov_ev) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ov_group_ev) (void)hipEventDestroy(e);
  (void)hipFree(c->d_ov_state);
  if (c->arena) (void)hipFree(c->arena);
  delete c;
}

uint32_t zkir_padded_log_n(uint64_t n_real) { uint32_t k = 3; while (((uint64_t)1 << k) < n_real) k++; return k; }

int zkir_main_trace_launch(const zkir_trace_columns* trace, uint64_t n_real, uint32_t deferred, uint32_t* out, void* stream) {
  if (!trace || !out || n_real == 0 || deferred > 1) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_main_trace_launch: null argument, empty trace, or a mode other than 0 / 1 (mode 2: zkir_main_trace_io_launch)"}); return ZKIR_ERR_ARGUMENT; }
  const uint64_t N = (uint64_t)1 << zkir_padded_log_n(n_real);
  if (deferred) hipLaunchKernelGGL(main_trace_kernel<1>, dim3(grid_for(N)), dim3(NT), 0, (hipStream_t)stream, *trace, n_real, N, out, IoRowArgs{});
  else hipLaunchKernelGGL(main_trace_kernel<0>, dim3(grid_for(N)), dim3(NT), 0, (hipStream_t)stream, *trace, n_real, N, out, IoRowArgs{});
  return check_launch("main_trace");
}
// MODE 2: prefix counts of the WRITE / READ ecalls (three launches over the instruction and R10 columns), then the row kernel
int zkir_main_trace_io_launch(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, uint32_t* scratch, uint32_t* out, void* stream) {
  if (!trace || !out || !io || !scratch || n_real == 0 || (!io->inputs && io->n_inputs)) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_main_trace_io_launch: null argument or empty trace"}); return ZKIR_ERR_ARGUMENT; }
  hipStream_t s = (hipStream_t)stream;
  const uint64_t N = (uint64_t)1 << zkir_padded_log_n(n_real);
  uint2* cnt = reinterpret_cast<uint2*>(scratch);
  uint2* sums = cnt + N;
  const uint32_t n_blk = (uint32_t)((N + IOS_ROWS - 1) / IOS_ROWS);
  hipLaunchKernelGGL(io_scan_local_kernel, dim3(n_blk), dim3(NT), 0, s, *trace, n_real, N, cnt, sums);
  hipLaunchKernelGGL(io_scan_sums_kernel, dim3(1), dim3(NT), 0, s, sums, n_blk);
  hipLaunchKernelGGL(io_scan_add_kernel, dim3(grid_for(N)), dim3(NT), 0, s, cnt, N, sums);
  hipLaunchKernelGGL(main_trace_kernel<2>, dim3(grid_for(N)), dim3(NT), 0, s, *trace, n_real, N, out,
                     IoRowArgs{io->inputs, io->n_inputs, io->writes_before, io->reads_before, reinterpret_cast<const uint32_t*>(cnt)});
  return check_launch("main_trace_io");
}
// MODES 3 / 4: the same scan, then the row kernel with the memory witness (device arrays of n_real entries)
}  // extern "C" (the two helpers are templates)
namespace {
// (mode 4 d) what a wide-tape row writes: y = the reference's result on the raw 64-bit operands of the row (air::wide_result), filled into the committed matrix in place
BB_HD void wide_tape_fix_row(uint32_t* __restrict__ out, uint64_t N, uint64_t i) {
  using namespace air;
  auto at = [&](int c) -> uint32_t& { return out[b8((uint32_t)phys_col(c, 4), i, N)]; };
  if (!at(C_OT)) return;
  const uint64_t a = (uint64_t)at(C_XB) | ((uint64_t)at(C_XB + 1) << 20) | ((uint64_t)at(C_XB + 2) << 40), b = (uint64_t)at(C_XC) | ((uint64_t)at(C_XC + 1) << 20) | ((uint64_t)at(C_XC + 2) << 40);
  const uint64_t res = wide_result(at(C_OP), a, b);
  at(C_Y) = (uint32_t)(res & 0xFFFFF); at(C_Y + 1) = (uint32_t)((res >> 20) & 0xFFFFF); at(C_Y + 2) = (uint32_t)(res >> 40);
}
__global__ __launch_bounds__(NT) void wide_tape_fix_kernel(uint32_t* __restrict__ out, uint64_t N, uint64_t n_real) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i < n_real) wide_tape_fix_row(out, N, i);
}
template <int MODE>
int main_trace_mem_launch(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, const uint64_t* mem_old, const uint32_t* mem_told, uint32_t* scratch, uint32_t* out, void* stream, uint64_t boundary = 0x1000) {
  if (!trace || !out || !io || !scratch || !mem_old || !mem_told || n_real == 0 || (!io->inputs && io->n_inputs)) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_main_trace_mem_launch: null argument or empty trace"}); return ZKIR_ERR_ARGUMENT; }
  hipStream_t s = (hipStream_t)stream;
  const uint64_t N = (uint64_t)1 << zkir_padded_log_n(n_real);
  uint2* cnt = reinterpret_cast<uint2*>(scratch);
  uint2* sums = cnt + N;
  const uint32_t n_blk = (uint32_t)((N + IOS_ROWS - 1) / IOS_ROWS);
  hipLaunchKernelGGL(io_scan_local_kernel, dim3(n_blk), dim3(NT), 0, s, *trace, n_real, N, cnt, sums);
  hipLaunchKernelGGL(io_scan_sums_kernel, dim3(1), dim3(NT), 0, s, sums, n_blk);
  hipLaunchKernelGGL(io_scan_add_kernel, dim3(grid_for(N)), dim3(NT), 0, s, cnt, N, sums);
  hipLaunchKernelGGL(main_trace_kernel<MODE>, dim3(grid_for(N)), dim3(NT), 0, s, *trace, n_real, N, out,
                     IoRowArgs{io->inputs, io->n_inputs, io->writes_before, io->reads_before, reinterpret_cast<const uint32_t*>(cnt), mem_old, mem_told, boundary});
  if (MODE == 4) hipLaunchKernelGGL(wide_tape_fix_kernel, dim3(grid_for(n_real)), dim3(NT), 0, s, out, N, n_real);
  return check_launch("main_trace_mem");
}
template <int MODE>
int main_trace_mem_host(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, const uint64_t* mem_old, const uint32_t* mem_told, uint32_t* out, uint64_t boundary = 0x1000) {
  if (!trace || !out || !io || !mem_old || !mem_told || n_real == 0) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_main_trace_mem_host: null argument or empty trace"}); return ZKIR_ERR_ARGUMENT; }
  const uint64_t N = (uint64_t)1 << zkir_padded_log_n(n_real);
  std::vector<uint32_t> cnt(2 * N);
  uint32_t w = 0, r = 0;
  for (uint64_t i = 0; i < N; i++) { cnt[2 * i] = w; cnt[2 * i + 1] = r; uint32_t f[2] = {0, 0}; if (i < n_real) io_row_flags(*trace, n_real, i, f); w += f[0]; r += f[1]; }
  const IoRowArgs a{io->inputs, io->n_inputs, io->writes_before, io->reads_before, cnt.data(), mem_old, mem_told, boundary};
  for (uint64_t i = 0; i < N; i++) main_trace_row<MODE>(*trace, n_real, N, i, out, &a);
  if (MODE == 4) for (uint64_t i = 0; i < n_real; i++) wide_tape_fix_row(out, N, i);
  return ZKIR_OK;
}
}  // namespace
extern "C" {
int zkir_main_trace_mem_launch(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, const uint64_t* mem_old, const uint32_t* mem_told, uint32_t* scratch, uint32_t* out,
                               void* stream) { return main_trace_mem_launch<3>(trace, n_real, io, mem_old, mem_told, scratch, out, stream); }
int zkir_main_trace_mem_host(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, const uint64_t* mem_old, const uint32_t* mem_told, uint32_t* out) {
  return main_trace_mem_host<3>(trace, n_real, io, mem_old, mem_told, out);
}
// MODE 4 (round 6: mode 3 + the wide-arithmetic class MULH / DIVU / REMU / DIV / REM, hash syscalls, the boundary cell): 288 committed columns; code_size = the program's (its
// boundary cell: air::boundary_cell)
int zkir_main_trace_wide_launch(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, const uint64_t* mem_old, const uint32_t* mem_told, uint64_t code_size, uint32_t* scratch,
                                uint32_t* out, void* stream) { return main_trace_mem_launch<4>(trace, n_real, io, mem_old, mem_told, scratch, out, stream, air::boundary_cell(code_size)); }
int zkir_main_trace_wide_host(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, const uint64_t* mem_old, const uint32_t* mem_told, uint64_t code_size, uint32_t* out) {
  return main_trace_mem_host<4>(trace, n_real, io, mem_old, mem_told, out, air::boundary_cell(code_size));
}
int zkir_main_trace_io_host(const zkir_trace_columns* trace, uint64_t n_real, const zkir_io_args* io, uint32_t* out) {
  if (!trace || !out || !io || n_real == 0) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_main_trace_io_host: null argument or empty trace"}); return ZKIR_ERR_ARGUMENT; }
  const uint64_t N = (uint64_t)1 << zkir_padded_log_n(n_real);
  std::vector<uint32_t> cnt(2 * N);
  uint32_t w = 0, r = 0;
  for (uint64_t i = 0; i < N; i++) { cnt[2 * i] = w; cnt[2 * i + 1] = r; uint32_t f[2] = {0, 0}; if (i < n_real) io_row_flags(*trace, n_real, i, f); w += f[0]; r += f[1]; }
  const IoRowArgs a{io->inputs, io->n_inputs, io->writes_before, io->reads_before, cnt.data()};
  for (uint64_t i = 0; i < N; i++) main_trace_row<2>(*trace, n_real, N, i, out, &a);
  return ZKIR_OK;
}
// EXPERIMENT (profiles/HISTORY.md: round 4 against VERDICT r3 #5): main trace without its first two blocks + the extension whose first inverse pass generates them from the trace.
// Together they are zkir_main_trace_launch + zkir_lde_launch with 128 B/row less HBM traffic; same output.  Returns ZKIR_ERR_ARGUMENT where the fused pass
// does not apply (log_n < 20 or = 21, deferred mode).
int zkir_commit_fused01_launch(const zkir_stark_ctx* c, const zkir_trace_columns* trace, uint64_t n_real, uint32_t* m, uint32_t width, uint32_t* out, void* stream) {
  if (!c || !trace || !m || !out || n_real == 0 || zkir_padded_log_n(n_real) != c->log_n) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_commit_fused01_launch: bad argument"}); return ZKIR_ERR_ARGUMENT; }
  const uint64_t N = (uint64_t)1 << c->log_n;
  hipLaunchKernelGGL((main_trace_kernel<0, 2>), dim3(grid_for(N)), dim3(NT), 0, (hipStream_t)stream, *trace, n_real, N, m, IoRowArgs{});
  const zkir::LdeTables t{(int)c->log_n, c->d_tw_inv, c->d_tw_fwd, c->d_g_lo, c->d_g_hi, c->d_small_inv, c->d_small_fwd};
  if (!zkir::lde_run_fused01(t, trace, n_real, m, (width + 7) / 8, out, stream)) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_commit_fused01_launch: not applicable at this size"}); return ZKIR_ERR_ARGUMENT; }
  return check_launch("commit_fused01");
}

// EXPERIMENT: one strided NTT pass with a chosen tile geometry over `width` columns of 2^log_n (inverse) / 2^(log_n + 1) (forward) rows: timing only (ntt.hip: strided_variant_run)
int zkir_ntt_strided_variant_launch(const zkir_stark_ctx* c, uint32_t* data, uint32_t width, int variant, int forward, void* stream) {
  const zkir::LdeTables t{(int)c->log_n, c->d_tw_inv, c->d_tw_fwd, c->d_g_lo, c->d_g_hi, c->d_small_inv, c->d_small_fwd};
  if (!zkir::strided_variant_run(t, data, (width + 7) / 8, variant, forward != 0, stream)) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_ntt_strided_variant_launch: log_n < 20 or unknown variant"}); return ZKIR_ERR_ARGUMENT; }
  return check_launch("ntt_strided_variant");
}

// The same rows on the host (trace = HOST pointers, out = host buffer of zkir_main_trace_width_for(deferred) / 8 blocks [N][8]): main_trace_row is one
// host + device function, so what the kernel computes can be checked without a GPU.  A test / diagnostic entry point, not a fallback: nothing in the
// product calls it.
int zkir_main_trace_host(const zkir_trace_columns* trace, uint64_t n_real, uint32_t deferred, uint32_t* out) {
  if (!trace || !out || n_real == 0) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_main_trace_host: null argument or empty trace"}); return ZKIR_ERR_ARGUMENT; }
  const uint64_t N = (uint64_t)1 << zkir_padded_log_n(n_real);
  for (uint64_t i = 0; i < N; i++) { if (deferred) main_trace_row<1>(*trace, n_real, N, i, out); else main_trace_row<0>(*trace, n_real, N, i, out); }
  return ZKIR_OK;
}

// EXPERIMENT (round 6): zkir_lde_launch + zkir_merkle_commit_launch with the leaf sponge overlapped with the extension.  The matrix is extended in
// groups of `group` B8 blocks on `stream`; each group's absorption (leaf_absorb_kernel: the sponge state of every leaf read and written once per group) runs on the
// context's second stream behind the group's LDE, so group g + 1 is extended while group g is hashed — the hash leaves ~95 % of HBM idle, the LDE ~30 % of the VALU issue
// slots.  Same permutations in the same order per leaf: same tree.  Measured in profiles/r06_overlap_variants.txt (scripts/time_overlap.py).
int zkir_commit_overlapped_launch(const zkir_stark_ctx* c, uint32_t* in, uint32_t width, uint32_t* out, uint32_t* tree, uint32_t group, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!c || !in || !out || !tree || width == 0 || group == 0) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_commit_overlapped_launch: bad argument"}); return ZKIR_ERR_ARGUMENT; }
  std::lock_guard<std::mutex> lk(c->mu);
  const uint64_t N = (uint64_t)1 << c->log_n, N2 = N << c->log_blowup;
  const uint32_t nb = (width + 7) / 8, n_groups = (nb + group - 1) / group;
  if (!c->ov_stream) {
    if (hipStreamCreateWithFlags(&c->ov_stream, hipStreamNonBlocking) != hipSuccess) return check_launch("overlap stream");
    for (auto& e : c->ov_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return check_launch("overlap event");
  }
  while (c->ov_group_ev.size() < n_groups) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return check_launch("overlap event"); c->ov_group_ev.push_back(e); }
  if (c->ov_state_bytes < 48 * N2) { (void)hipFree(c->d_ov_state); c->d_ov_state = nullptr; c->ov_state_bytes = 0; if (hipMalloc(&c->d_ov_state, 48 * N2) != hipSuccess) return check_launch("overlap state"); c->ov_state_bytes = 48 * N2; }
  const zkir::LdeTables t{(int)c->log_n, c->d_tw_inv, c->d_tw_fwd, c->d_g_lo, c->d_g_hi, c->d_small_inv, c->d_small_fwd};
  (void)hipEventRecord(c->ov_ev[0], s); (void)hipStreamWaitEvent(c->ov_stream, c->ov_ev[0], 0);              // the second stream starts behind whatever `stream` holds
  for (uint32_t g = 0; g < n_groups; g++) {
    const uint32_t b0 = g * group, b1 = std::min(nb, b0 + group);
    zkir::lde_run(t, in + (uint64_t)b0 * N * 8, b1 - b0, out + (uint64_t)b0 * N2 * 8, s);
    (void)hipEventRecord(c->ov_group_ev[g], s); (void)hipStreamWaitEvent(c->ov_stream, c->ov_group_ev[g], 0);
    hipLaunchKernelGGL(leaf_absorb_kernel, dim3(grid_for(N2)), dim3(NT), 0, c->ov_stream, c->d_p2, out, width, b0, b1, N2, c->consts.in_scale, c->d_ov_state, g == 0 ? 1 : 0, g + 1 == n_groups ? 1 : 0, tree);
  }
  (void)hipEventRecord(c->ov_ev[1], c->ov_stream); (void)hipStreamWaitEvent(s, c->ov_ev[1], 0);
  launch_tree_levels(c->d_p2, tree, N2, c->sync, s);
  return check_launch("commit_overlapped");
}

// in: ceil(width/8) blocks [N][8] of canonical evaluations over H (natural order; used as scratch and overwritten!), out: blocks [2N][8]
int zkir_lde_launch(const zkir_stark_ctx* c, uint32_t* in, uint32_t width, uint32_t* out, void* stream) { return lde_launch(c, in, width, out, false, (hipStream_t)stream); }

// mat: B8 layout, ceil(width/8) blocks [n_leaves][8]; tree = [leaf digests (4*n)] [layer 1 (4*n/2)] ... [root (4)] = 4*(2n-1) words; n_leaves a power of two
int zkir_merkle_commit_launch(const zkir_stark_ctx* c, const uint32_t* mat, uint32_t width, uint64_t n_leaves, uint32_t* tree, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!c || n_leaves == 0 || (n_leaves & (n_leaves - 1))) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "merkle: null context, or n_leaves not a power of two"}); return ZKIR_ERR_ARGUMENT; }
  return merkle_commit(c, mat, width, n_leaves, tree, false, s);
}

// The leaf layer alone: digests[0 .. 4n) = sponge over the `width` real columns of every row (leaf_hash_kernel); zkir_merkle_cap_launch
// on the same buffer then adds the levels above — together they are zkir_merkle_commit_launch.  Exists so that a caller (bench.py)
// can bracket the dominant kernel of the commit step with its own events.
int zkir_merkle_leaves_launch(const zkir_stark_ctx* c, const uint32_t* mat, uint32_t width, uint64_t n_leaves, uint32_t* digests, void* stream) {
  if (!c || !mat || !digests || n_leaves == 0) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "merkle leaves: null argument or no leaves"}); return ZKIR_ERR_ARGUMENT; }
  hipLaunchKernelGGL(leaf_hash_kernel, dim3(grid_for(n_leaves)), dim3(NT), 0, (hipStream_t)stream, c->d_p2, mat, width, n_leaves, c->consts.in_scale, digests);
  return check_launch("merkle_leaves");
}

// Upper levels over already-computed digests (multi-GPU: the all-gathered subtree roots of the row shards are the leaves of the
// top log2(G) levels).  tree[0 .. 4n) must hold the n digests; the call fills the remaining 4(n-1) words, root = last 4.
int zkir_merkle_cap_launch(const zkir_stark_ctx* c, uint32_t* tree, uint64_t n_digests, void* stream) {
  if (!c || n_digests == 0 || (n_digests & (n_digests - 1))) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "merkle cap: null context, or n_digests not a power of two"}); return ZKIR_ERR_ARGUMENT; }
  launch_tree_levels(c->d_p2, tree, n_digests, c->sync, (hipStream_t)stream);
  return check_launch("merkle_cap");
}

}  // extern "C"

#include "stark_prove.inl"
