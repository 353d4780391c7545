// air.h — the AIR of ZKIR-STARK (v6): column map of the 172 LOGICAL main-trace columns (152 committed in default mode, 168 deferred) and of
// the 40-column aux trace, and the constraint list (398 constraints), written
// ONCE for the two places of the product that evaluate it: the quotient kernel (stark_prove.inl; base-field values at every point
// of the LDE coset, lazily accumulated) and the host verifier (verify.cpp; extension-field openings at zeta).  The oracle
// (oracle/stark_oracle.cpp, constraints_sum) states the same list independently in naive arithmetic; constraint c carries the
// coefficient alpha^c and the indices below are that list's order.  Every constraint has degree <= 2 in the columns (x is_trans, degree 1)
// or degree 1 (x is_first / is_last): the quotient has degree < N, which is what blow-up 2 affords — no helper column can be inlined.
//
// What the constraints say (default VM mode; `deferred` public input = 0) — the full semantics of 20 of the 50 opcodes, the control flow of all:
//   * an executed row runs as the class of ITS PROGRAM WORD (opclass, part of the instruction-ROM tuple): ADD, ADDI, SUB, JAL, JALR, the
//     comparison families BEQ / BNE, SEQ / SNE (pairs: op = base + pol), SLTU / SGEU / SLT / SGE and BLT / BGE / BLTU / BGEU (four members:
//     op = base + 2 g + pol, g = the word's variant bit, also in the ROM tuple) and the conditional moves CMOV / CMOVNZ (class cmn) and CMOVZ
//     (class cmz); everything else is class "other";
//   * wr = one-hot(rd) on rows that write (ADD / ADDI / SUB / JAL / JALR / SEQ.. / SLTU..), on a conditional move exactly when its condition
//     holds (q), empty on branch / halt / padding rows, at most one register otherwise; selb = one-hot(field b), selc = one-hot(field c) (of
//     field a on B-type rows), xb / xc = the selected limbs;
//   * z = R0 + 1024 R1, R2 + 1024 R3: the row's first RANGE-CHECKED pair of 20-bit limbs (no columns of its own since v6): xb + xc, xb + sext(imm17),
//     pc + 4 (mod 2^40, boolean carries), xb - xc on SUB rows; on ORDERED comparisons (SLTU.. / BLT.. rows) the difference of the operands with
//     their high limbs BIASED, t = limb + 2^19 sgn - 2^20 (sign bit) — the limbs of value XOR 2^39 when the comparison is signed (sgn = g on SLTU..
//     rows, 1 - g on BLT.. rows; Value40::signed_lt, value.rs:710-716) — u = (ta, tb) = R4 + 1024 R5, R6 + 1024 R7 being the row's SECOND
//     range-checked pair, which forces the sign bits (sa = column b0, sb); the borrow out of 40 bits IS the comparison (execute.rs:361-407, :598-636);
//     y = the value written: z on arithmetic rows, the comparison (0 / 1) on SEQ.. / SLTU.. rows, rs1's raw limbs on conditional moves — the
//     register selected by wr shows y in the next row, every other register keeps its limbs and storage state; on "other" rows y is a free
//     witness whose low limbs are z (in range) and whose bits above 40 are range-checked through the second pair: y2 = R4 + 2^10 R5 + 2^20 R6,
//     R7 = 64 R6 — so every limb of every register is in range by induction;
//   * flag = [xb == xc] over all three limbs (raw 64-bit compare, execute.rs:409-431, :578-596) on the equality families, the borrow on
//     the ordered ones; fx = flag XOR pol; a branch is taken iff fx; nz = [xc != 0] on every row (sum of the in-range limbs), q = K_cmn nz +
//     K_cmz (1 - nz) (execute.rs:434-472);
//   * pc' = pc + 4 | pc + sext(imm17) if taken | pc + sext(off21) for JAL | (rs1 + sext(imm17)) & ~1 for JALR, mod 2^64 (state.rs:131-133,
//     execute.rs:649-658): class "other" is sequential — the control flow of EVERY opcode is constrained;
//   * cycle counts up; row 0 is in the public FIRST state and row n_real - 1 in the public LAST state (for a whole run the verifier
//     requires the first state to be the VM's initial one: cycle 0, entry point, zero registers); the row count is public: row
//     n_real - 1 is the halt row, only padding follows it, padding keeps everything.
// ONE LogUp lookup argument used twice (aux trace: ten extension-field columns committed after the lookup challenges: H0..H7, HR, S):
//   * instruction ROM: the tuple (pc limbs, op, fa, fb, fc, fhi, s, opclass, g) of EVERY row is a row of the program's code table — the
//     verifier builds that table from the program carried in the proof (its digest is the public program_digest), so the instruction
//     word at pc is the program's (vm.rs:362-379), its fields are in range, and the class an executed row runs as is the class of that
//     word (opclass), not a free witness;
//   * ranges: EIGHT 10-bit chunks per row, every chunk a row of the 2^10 table (range_check.rs:175-192, config.rs:78-80), which makes the
//     boolean carries / borrows / sign bits the only solution;
//   the prover sends the multiplicities of both tables BEFORE the challenges (alpha, lambda) are drawn; the verifier computes the table
//   side T = sum m_t / (alpha - t) + sum r_u / (alpha - fingerprint_u) itself; the running-sum column closes over the cycle of N rows.
// Not constrained IN THE AIR in modes 0 / 1: WHICH instruction the halt row is — I_HALT only forces class "halt" onto the public last row, any row can be it —
// and the io digest (inputs, outputs, halt reason, cycle count: bound into the transcript, never opened by a constraint).  The halt half is closed OUTSIDE the AIR by
// zkir_verify_io (verify.cpp): given the claim in the clear it checks the digest and that the public last state sits on the EBREAK / exit-ECALL the claim names.
// Also not constrained there: the VALUES the other 30 opcodes write (MUL / DIV / logic / shifts / loads / ECALL: class "other", y is a free in-range witness),
// memory consistency, the SHA-256 chip, deferred-mode arithmetic (deferred = 1 relaxes the write constraints to "unwritten registers keep their value"; branches and jumps
// run as the free-pc class "oj" there, which no default-mode row can be).
// ROUND 4 — two opt-in MODES append columns and constraints to the default-mode list (modes 0 / 1 are unchanged):
//   MODE 2 = the I/O argument: ECALL's READ / WRITE / EXIT constrained, the outputs and consumed inputs tied to tapes the proof carries, the halt row bound (the verifier does
//            zkir_verify_io's checks itself);
//   MODE 3 = mode 2 + the memory argument + the bitwise opcodes + the shifts: the ten loads and stores constrained, every access one step of an offline memory check over
//            8-byte cells; AND OR XOR ANDI ORI XORI nibble by nibble through 256-entry tables; SLL SRL SRA SLLI SRLI SRAI as a 2^t = H 2^40 + L over 10-bit chunks; MUL
//            as a schoolbook product of 10-bit chunks (43 of 50 opcodes now carry their semantics).  Left free there: MULH / DIVU / REMU / DIV / REM (class "other": they work
//            on the raw 64-bit registers), hash syscalls (forbidden: fh = 0), the SHA-256 chip.
#pragma once
#include "babybear.h"

namespace air {

constexpr int W = 308;                                         // logical columns of MODE 4; mode 3 uses the first 284 (W_LOGICAL_MEM), mode 2 the first 180 (W_LOGICAL_IO), modes 0 / 1 the first 172 (W_LOGICAL_BASE)
constexpr int W_LOGICAL_BASE = 172, W_LOGICAL_IO = 180, W_LOGICAL_MEM = 284;
// MODES (the header's word 9, zkir_public_inputs::deferred): 0 = default VM mode, 1 = deferred carry model, 2 (round 4) = default mode WITH the I/O argument:
// ECALL is a class of its own there (id K_ECALL, no column: Kec = f2 + rl + re + fh), dispatched on R10's limbs — f2 = WRITE (R10 = 2), rl / re = READ (R10 = 1) on a
// non-empty / exhausted input tape, fh = a hash syscall (R10 = 3 + h0 + 2 h1) — oc / ic count the outputs written / inputs consumed before the row; WRITE rows send
// (oc, R11's limbs), live READ rows (ic, the limbs written to R10) into a LogUp relation whose table side the VERIFIER forms from the tapes the proof carries
// (syscall.rs:94-177).  A bool passed where a mode is expected reads as 0 / 1.
enum : int { C_F2 = 172, C_RL = 173, C_RE = 174, C_FH = 175, C_H0 = 176, C_H1 = 177, C_OC = 178, C_IC = 179 };
// MODE 3 (round 4) = mode 2 WITH the memory argument (oracle/stark_oracle.cpp "MODE 3"): the ten loads and stores (execute.rs:477-575) are classes of their
// own (ld = 16, st = 17, columns kld / kst) and every access is one step of an offline memory check over aligned 8-byte CELLS — the row READS the tuple (cell address, time of
// the previous access, the cell's eight bytes) and WRITES (cell address, its own cycle + 1, the new bytes), the time read is smaller than the time written, and the VERIFIER
// closes the multiset equation with the initial bytes (the program image, zero elsewhere) and the final (bytes, time) of every touched cell, which the proof carries.
//   e_v: one-hot of the accessed WINDOW v = (width, offset): 0-7 a byte at offset v, 8-11 a halfword at 2 (v - 8), 12-13 a word at 4 (v - 12), 14 the cell
//   ob_0..7 the cell's bytes before the access, told the time of the previous access (0: never)
//   pieces d0 d1 n0 n1 d3 d4 d5 d6 d7 of the 64-bit window value (byte 2 = n0 + 16 n1): register limbs d0 + 2^8 d1 + 2^16 n0 | n1 + 2^4 d3 + 2^12 d4 | d5 + 2^8 d6 + 2^16 d7;
//   the stored register on stores, the loaded window (zero-extended) on loads, where a byte / halfword load also keeps d6 = 2 x (the low seven bits of its top byte)
//   sgb / sgh: the row is LB / LH; tb: the top bit of what it loads; sx = (sgb + sgh) tb; cm2: the carry out of the address's third limb (the address stays below 2^40)
// .. and the six BITWISE opcodes AND OR XOR ANDI ORI XORI (execute.rs:199-282: on the 40-bit values, the immediate sign-extended and masked), class lg = 18, nibble by nibble: operands
// and result are ten nibbles each, nibble k a tuple (a_k, b_k, r_k) looked up in the 256-entry table of the row's operation — in the nine piece slots (a_k = piece k) and, the tenth, in
// the row's last range slot (a_9 = chunk R7); oa / oo: the operation is AND / OR (XOR = klg - oa - oo), li: the second operand is the immediate
enum : int { C_KLD = 180, C_KST = 181, C_E = 182, C_OB = 197, C_TOLD = 205, C_PIECE = 206, C_SGB = 215, C_SGH = 216, C_TB = 217, C_SX = 218, C_CM2 = 219,
             C_KLG = 220, C_OA = 221, C_OO = 222, C_LI = 223, C_LB = 224, C_LR = 234,
             // .. and the six SHIFTS SLL SRL SRA SLLI SRLI SRAI (execute.rs:284-358, value.rs:658-697), class sh = 19, as ONE relation a 2^t = H 2^40 + L (a left shift by sh is L at
             // t = sh, a right shift H at t = 40 - sh; 40 and more shift everything out): a in its four 10-bit chunks c_i (the row's second range group), t = 10 u + v by two
             // one-hots, c_i 2^v = lo_i + 2^10 hi_i (both range-checked: unique), the chunks of a 2^v are m_i = lo_i + hi_(i-1) — no carry — and the one-hot of u picks four of
             // them; SRA adds sign (2^40 - 2^t).  ksh | ul_0..4 / ur_0..4: shifts LEFT / RIGHT with chunk shift u | v_0..9: the bit shift | sa: SRA / SRAI | si: the amount is the
             // word's shamt | sb9: bit 39 of a | sgn = sa sb9 | pr_i = c_i 2^v | on_0, on_1: the limbs of 2^40 - 2^t on right shifts | sh: the amount.  Shared columns on a shift
             // row: R0..R3 = lo_i, R4..R7 = c_i, pieces 0-3 = hi_i, 4 = 2 (c_3 mod 2^9), 5 = the rest of rs2's low limb / of the word's field, 6 = d (what sh exceeds t's range
             // by), 7 = the shamt's high nibble, 8 = rs2's first chunk, looked up WITH sh in LOW6 = {(v, v & 63)}
             C_KSH = 244, C_UL = 245, C_UR = 250, C_V = 255, C_SA = 265, C_SI = 266, C_SB9 = 267, C_SGN = 268, C_PR = 269, C_ON = 273, C_SH = 275,
             // .. and MUL (execute.rs:79-99: Value40::wrapping_mul of the masked operands), class mu = 20, schoolbook in 10-bit chunks: a = rs1's low limbs in R4..R7, b = rs2's in
             // pieces 0-3, the result's chunks r_k = R0..R3 (= z), and for k = 0..3  sum_{i+j=k} a_i b_j + carry_(k-1) = r_k + 2^10 carry_k  with carry_0 = piece 4, carry_1 =
             // piece 5 + 2^10 e_1, carry_2 = piece 6 + 2^10 (e_2 + 2 e_3), carry_3 = piece 7 + 2^10 piece 8 (dropped) — every slot a 10-bit range lookup on such a row: both sides
             // stay below p, the equations hold over the integers.  The products have degree 2 already, so the class cannot gate them: ma_i = kmu a_i are columns (zero elsewhere).
             C_KMU = 276, C_MA = 277, C_ME = 281,
             // MODE 4 (round 6, proof format v12) = mode 3 WITH (a) the wide-arithmetic class, (b) hash syscalls as a tape, (c) the code segment's boundary cell
             // (oracle/stark_oracle.cpp "MODE 4").
             // (a) class wa = 22: MULH DIVU REMU DIV REM (opcodes 3..7, execute.rs:101-183) on operands BELOW 2^40.  The reference computes the five on the raw 64-bit registers
             // (quirks Q2, Q3); for registers below 2^40 an i64 is non-negative, so DIV = DIVU, REM = REMU, the product has 80 bits, and all five are ONE relation
             // F1 F2 + ADD = LO + 2^40 HI over 40-bit integers (MULH: a b = L + 2^40 y; DIVU / DIV: y b + r = a, r < b; REMU / REM: q b + y = a, y < b).  A wa row states
             // kwa xb2 = kwa xc2 = 0: a run that feeds one of the five a register with bits above 40 has no mode-4 proof (zkir_prove refuses it).  Schoolbook in 10-bit chunks like
             // MUL; the 80-bit product, the addend and the remainder's range check need 23 lookups where a mode-3 row has 17, so the mode adds SIX 10-bit range slots X0..X5 per row
             // (24 aux columns XH0..XH5) and 22 main columns (kwa = om + od + orr is an expression; DIV / REM are the word's variant bit g, from the ROM: op = 3 om + 4 od + 5 orr
             // + 2 g):  om (MULH) od (the quotient is written) orr (the remainder is) | gf_k = kwa F1_k (gated copies: the products have degree 2 already) | e_1..9 (the carries'
             // bits above their slot) | X0..5.  Slots of a wa row (all read the 10-bit table): R0..R3 = LO, R4..R7 = F1, pieces 0-3 = F2 (= rs2), pieces 4-6 = the low parts of c0 c1
             // c2 (c1 = p5 + 2^10 e1, c2 = p6 + 2^10 (e2 + 2 e3)), pieces 7, 8, X0, X1 = G4 (HI on MULH, ADD = the remainder on divisions), X2..X5 = G5 (MULH: c3 = X2 + 2^10 (e4 + 2
             // e5), c4 = X3 + .. (e6, e7), c5 = X4 + .. (e8, e9); divisions: the chunks of d = rs2 - r - 1 >= 0, borrow e4).  Position k: sum_{i+j=k} F1_i F2_j + ADD_k + c_(k-1) =
             // LO_k + 2^10 c_k (k <= 3), = HI_(k-4) + 2^10 c_k (k = 4, 5; k = 6: HI_2 + 2^10 HI_3); a division has HI = 0, c3 = 0 and no product above position 3.
             // (b) HASH SYSCALLS (SHA-256 = 3, Keccak-256 = 5, BLAKE3 = 6; syscall.rs:121-171) are a TAPE like mode 2's I/O: the proof carries one record per call (cycle, the
             // pointers, the length, the kind, and per touched 8-byte cell its bytes before the call and the time of its previous access); the VERIFIER computes every digest and
             // adds the call's memory accesses to the table side of the memory check.  The AIR ties the ECALL row to its record: HH (alpha - fp(cycle, R11, R12, R13, 3 + h0 +
             // 2 h1) - 12 lambda^11) = fh, four aux columns; mode 3's "no hash syscall" becomes "no syscall 4" (Poseidon2: an error in the reference).
             // (c) the BOUNDARY CELL (code_size % 8 == 4: the last code word and the first four data bytes share a cell): admitted here, with "no store writes its low half":
             // iws, nb with nb = delta iws, delta = the row's cell address minus B as ONE field element (B: LK_B0 / LK_B1, a constant of the program), and tl (kst - nb) = 0 with
             // tl = the windows that touch bytes 0..3.
             // (d) the WIDE TAPE: the five wide opcodes on operands with bits ABOVE 40 (the reference computes them on the raw 64-bit registers: `as i64`, 128-bit products — a
             // sign-extended LB result, an LD, a 64-bit input) are proven like the hash calls.  ot = "this wide row goes through the tape": the class is om + od + orr + ot, the
             // chunk relation is gated by om + od + orr alone; the proof carries one record (cycle, rs1's three limbs, rs2's, the opcode) per such row, the VERIFIER computes the
             // result with the reference's semantics (wide_result) and the row looks (cycle, rs1, rs2, rd's new value, opcode) up: WW (alpha - fp - 13 lambda^11) = ot, WW in the
             // four padding columns beside HH.  The honest prover takes the tape exactly when an operand has bits above 40 (canonical proofs); any wide row MAY (sound: the
             // verifier recomputes it).  Class "other" does not exist in mode 4 — no opcode is left in it — and ITS COLUMN is ot there: no column is added, no committed position moves.
             C_OM = 284, C_OD = 285, C_ORR = 286, C_GF = 287, C_WE = 291, C_X = 300, C_IWS = 306, C_NB = 307, C_OT = 131 /* = kcol(K_OTH), re-used */ };
constexpr int K_LD = 16, K_ST = 17, K_LG = 18, K_SH = 19, K_MU = 20, K_WA = 22, N_WIN = 15, N_PIECE = 9, N_NIB = 10, N_X = 6, N_WE = 9, TAG_HASH = 12, TAG_WIDE = 13;
BB_HD constexpr bool is_low_window(int v) { return v <= 3 || v == 8 || v == 9 || v == 12 || v == 14; }   // the windows that touch bytes 0..3 of their cell
BB_HD constexpr bool is_wide(uint32_t op) { return op >= 0x03 && op <= 0x07; }
// what MULH 3 / DIVU 4 / REMU 5 / DIV 6 / REM 7 write, on the raw 64-bit registers (execute.rs:101-183): MULH = bits 40..79 of the 128-bit product; DIVU / REMU unsigned; DIV /
// REM on `as i64` with wrapping_div / wrapping_rem (i64::MIN / -1 = i64::MIN, remainder 0).  b = 0 never is a row of a division (the VM stops with DivisionByZero).
// (A restoring division, one bit per step, instead of the `/` operator: in the row kernel — 256 registers, a 308-word row partly in scratch — the compiler's inline expansion of
// four 64-bit divisions left garbage in other columns of the row on gfx950, round 6; 64 short steps on the few rows that take the tape cost nothing.)
BB_HD void udivmod64(uint64_t a, uint64_t b, uint64_t& q, uint64_t& r) {
  q = 0; r = 0;
#pragma unroll 1
  for (int i = 63; i >= 0; i--) {
    const uint64_t carry = r >> 63;
    r = (r << 1) | ((a >> i) & 1);
    if (carry | (uint64_t)(r >= b)) { r -= b; q |= 1ull << i; }
  }
}
BB_HD uint64_t wide_result(uint32_t op, uint64_t a, uint64_t b) {
  if (op == 0x03) {                                             // (no 128-bit type on the device: the product's bits 40..79 from 32-bit halves)
    const uint64_t a0 = a & 0xFFFFFFFFull, a1 = a >> 32, b0 = b & 0xFFFFFFFFull, b1 = b >> 32;
    const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
    const uint64_t mid = (p00 >> 32) + (p01 & 0xFFFFFFFFull) + (p10 & 0xFFFFFFFFull);
    const uint64_t lo = (p00 & 0xFFFFFFFFull) | (mid << 32), hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
    return ((lo >> 40) | (hi << 24)) & ((1ull << 40) - 1);
  }
  if (b == 0) return 0;
  const bool sgn = op >= 0x06, na = sgn && (a >> 63), nb = sgn && (b >> 63);       // DIV / REM: |a| = |q| |b| + |r|, q negative iff the signs differ, r with the dividend's sign
  uint64_t q, r;
  udivmod64(na ? 0 - a : a, nb ? 0 - b : b, q, r);                                 // (|i64::MIN| = 2^63 as a u64: i64::MIN / -1 gives |q| = 2^63 = i64::MIN again — wrapping_div — and r = 0)
  if (op == 0x04 || op == 0x06) return (na != nb) ? 0 - q : q;
  return na ? 0 - r : r;
}
constexpr uint32_t OP_MUL_ = 0x02;
BB_HD constexpr bool is_logic(uint32_t op) { return op >= 0x10 && op <= 0x15; }
BB_HD constexpr bool is_shift(uint32_t op) { return op >= 0x18 && op <= 0x1D; }
BB_HD constexpr int win_width(int v) { return v < 8 ? 1 : v < 12 ? 2 : v < 14 ? 4 : 8; }
BB_HD constexpr int win_start(int v) { return v < 8 ? v : v < 12 ? 2 * (v - 8) : v < 14 ? 4 * (v - 12) : 0; }
BB_HD constexpr int win_of(int width, int off) { return width == 1 ? off : width == 2 ? 8 + off / 2 : width == 4 ? 12 + off / 4 : 14; }
constexpr uint32_t OP_LB = 0x30, OP_LH = 0x32, OP_LD = 0x35, OP_SB = 0x38, OP_SD = 0x3B;
BB_HD constexpr bool is_load(uint32_t op) { return op >= OP_LB && op <= OP_LD; }
BB_HD constexpr bool is_store(uint32_t op) { return op >= OP_SB && op <= OP_SD; }
BB_HD constexpr int mem_width(uint32_t op) { return is_store(op) ? 1 << (op - OP_SB) : op <= 0x31 ? 1 : op <= 0x33 ? 2 : op == 0x34 ? 4 : 8; }
// (mode 3) lookup tables beside the 10-bit range table (no tag), the ROM (tag 1) and the tapes (2, 3): LOW3 = {(v, v & 7) : v < 2^10} (tag 4: the FIRST range chunk of a memory
// row is looked up there, with the window's offset = the address's low three bits), BYTE = {v < 2^8} (5), NIBBLE = {v < 2^4} (6); memory tuples carry tag 7
// 8 / 9 / 10: the nibble tables {(a, b, a op b)} of AND / OR / XOR (256 entries each, entry 16 a + b); the multiplicities of LOW3 | BYTE | NIBBLE | AND | OR | XOR travel together
// 11: LOW6 = {(v, v & 63)} (1024 entries): the amount of a register shift
constexpr int TAG_LOW3 = 4, TAG_BYTE = 5, TAG_NIB = 6, TAG_MEM = 7, TAG_AND = 8, TAG_OR = 9, TAG_XOR = 10, TAG_LOW6 = 11, LG_BASE = 1024 + 256 + 16, L6_BASE = LG_BASE + 3 * 256,
              MEM_MULT = L6_BASE + 1024;
BB_HD constexpr int shift_piece_tag(int k, bool reg) { return k == 7 ? TAG_NIB : (k == 8 && reg) ? TAG_LOW6 : 0; }   // the table a piece slot reads on a SHIFT row
BB_HD constexpr uint32_t logic_of(int which, uint32_t a, uint32_t b) { return which == 0 ? (a & b) : which == 1 ? (a | b) : (a ^ b); }
BB_HD constexpr int piece_tag(int k) { return (k == 2 || k == 3) ? TAG_NIB : (k == 5 || k == 8) ? 0 : TAG_BYTE; }   // d0 d1 n0 n1 d3 d4 d5 d6 d7 (0: the 10-bit range table)
enum : int { C_CYCLE = 0, C_PC = 1, C_OP = 4, C_FA = 5, C_FB = 6, C_FC = 7, C_FHI = 8, C_LIMB = 9, C_STATE = 57, C_WR = 73, C_SELB = 88, C_SELC = 103,
             C_XB = 118, C_XC = 121, C_Y = 124, C_K = 127, C_OPC = 134, C_RC = 135, C_S = 139, C_SE = 140, C_C0 = 141, C_C1 = 142, C_D0 = 143, C_D1 = 144, C_D2 = 145,
             C_DL0 = 146, C_NE = 147, C_IV = 148, C_TK = 151, C_K2 = 152, C_NZ = 156, C_IVZ = 157, C_FLAG = 158, C_FX = 159, C_K3 = 160, C_B0 = 162, C_RC2 = 163, C_G = 167, C_SB = 168,
             C_K4 = 169, C_Q = 171, C_KOJ = C_K3 + 1 };
// LOGICAL vs COMMITTED columns (proof format v7).  W and the C_* map are the LOGICAL main trace: what the constraints talk about.  Columns
// that are identically zero by the constraints themselves are not committed: R0's three limbs and its storage state (R0 is hard-wired
// zero, state.rs:77-85) and, in the default VM mode — no register is ever Accumulated there (vm.rs:47) — all 16 storage states.  The
// committed matrix is the logical one with those columns removed, whole B8 blocks with no padding: 152 columns in default mode,
// 168 in deferred mode (W_COMMITTED_*); a removed column reads as the constant 0 wherever a constraint, a boundary state or a lookup mentions it.
constexpr int W_COMMITTED_DEFAULT = 152, W_COMMITTED_DEFERRED = 168, W_COMMITTED_IO = 160, W_COMMITTED_MEM = 264, W_COMMITTED_WIDE = 288, W_COMMITTED_MAX = 288;
// (AIR v6) The class column "other, jumps" (C_KOJ) is identically zero in the default mode as well — no opcode's class is oj there (constraint
// I_OPCLASS; deferred mode runs its branches and jumps as that class) — and is not committed either: 172 - 20 = 152 columns by default,
// 172 - 4 = 168 deferred, whole blocks of 8 with no padding.
BB_HD constexpr bool is_virtual(int c, int mode) { return (c >= C_LIMB && c < C_LIMB + 3) || (c >= C_F2 && mode < 2) || (c >= C_KLD && mode < 3) || (c >= C_OM && mode != 4) || (mode == 1 ? c == C_STATE : ((c >= C_STATE && c < C_STATE + 16) || c == C_KOJ)); }
BB_HD constexpr int phys_col(int c, int mode) { return c - (c >= C_LIMB + 3 ? 3 : 0) - (mode == 1 ? (c > C_STATE ? 1 : 0) : (c >= C_STATE + 16 ? 16 : 0) + (c > C_KOJ ? 1 : 0)); }   // of a committed column
BB_HD constexpr int committed_width(int mode) { return mode == 1 ? W_COMMITTED_DEFERRED : mode == 2 ? W_COMMITTED_IO : mode == 3 ? W_COMMITTED_MEM : mode == 4 ? W_COMMITTED_WIDE : W_COMMITTED_DEFAULT; }
BB_HD constexpr int committed_used(int mode) { return committed_width(mode); }               // (no padding since v6: 172 - 20, 172 - 4, 180 - 20, 276 - 20)
BB_HD constexpr int logical_width(int mode) { return mode == 4 ? W : mode == 3 ? W_LOGICAL_MEM : mode == 2 ? W_LOGICAL_IO : W_LOGICAL_BASE; }
// the logical column stored at committed position p (p < committed_used)
BB_HD constexpr int logical_col(int p, int mode) {
  int c = p;
  if (c >= C_LIMB) c += 3;                                     // R0's limbs
  if (mode == 1) { if (c >= C_STATE) c += 1; } else { if (c >= C_STATE) c += 16; if (c >= C_KOJ) c += 1; }
  return c;
}
// aux trace: H0..H7 (range helpers), HR (ROM helper), S (running sum), four coordinate columns each; mode 2: + HO (output-tape helper), HI (input-tape helper)
// mode 3: + P0..P8 (the piece lookups), HMR / HMW (the memory tuple read / written), FPN (the fingerprint of the new cell bytes: an aux column because it depends on lambda)
// mode 4: + XH0..XH5 (the helpers of the six extra range slots), HH (the hash-call helper), four columns of zero padding (whole blocks of 8)
constexpr int W_AUX = 40, W_AUX_IO = 48, W_AUX_MEM = 96, W_AUX_WIDE = 128, W_AUX_MAX = 128;
BB_HD constexpr int aux_width(int mode) { return mode == 4 ? W_AUX_WIDE : mode == 3 ? W_AUX_MEM : mode == 2 ? W_AUX_IO : W_AUX; }
enum : int { A_H = 0, A_HR = 32, A_S = 36, A_HO = 40, A_HI = 44, A_P = 48, A_HMR = 84, A_HMW = 88, A_FPN = 92, A_X = 96, A_HH = 120, A_WW = 124 };
constexpr int RC_BITS = 10, RC_TABLE = 1 << RC_BITS, N_TUPLE = 11, N_RC = 8;
BB_HD constexpr int rc_col(int k) { return k < 4 ? C_RC + k : C_RC2 + (k - 4); }     // the eight range lookups of a row: chunks of z, chunks of u
// per-proof lookup parameters (base-field words): alpha coordinates, the coordinates of lambda^0 .. lambda^N_TUPLE (= 11), T / N
enum : int { LK_ALPHA = 0, LK_LAM = 4, LK_TN = 4 + 4 * (N_TUPLE + 1), LK_NIN = LK_TN + 4 /* mode 2: the length of the input tape */,
             LK_B0 = LK_NIN + 1, LK_B1 = LK_B0 + 1 /* mode 4: the two 20-bit limbs of the boundary cell's address (a constant of the program) */, N_LK = LK_B1 + 1 };
// (mode 4) the code segment's BOUNDARY CELL: when code_size % 8 == 4 the last code word shares its cell with the first four data bytes; CODE_BASE (inside the code: refused anyway) when there is none
BB_HD constexpr uint64_t boundary_cell(uint64_t code_size) { return code_size % 8 == 4 ? 0x1000 + code_size - 4 : 0x1000; }
BB_HD constexpr int tuple_col(int j) { return j < 3 ? C_PC + j : j == 3 ? C_OP : j == 4 ? C_FA : j == 5 ? C_FB : j == 6 ? C_FC : j == 7 ? C_FHI : j == 8 ? C_S : j == 9 ? C_OPC : C_G; }
// AIR v3: class ids (= opclass values of the instruction word; halt / pad are row roles, not word classes).  A FAMILY is a pair of opcodes
// that differ in their low bit, the polarity of one comparison: bre = BEQ / BNE, bru = BLTU / BGEU, se = SEQ / SNE, su = SLTU / SGEU.
// AIR v4: jalr (JALR) and oj = "other, jumps" (BLT / BGE: the signed comparison is not stated yet — free next pc, nothing written); class
// "other" is SEQUENTIAL (pc + 4) like every instruction that is not a branch or a jump.
// AIR v6: cmn = CMOV / CMOVNZ (move if rs2 != 0), cmz = CMOVZ (move if rs2 == 0)
enum : int { K_ADD = 0, K_ADDI = 1, K_BRE = 2, K_JAL = 3, K_OTH = 4, K_HALT = 5, K_PAD = 6, K_SUB = 7, K_BRU = 8, K_SE = 9, K_SU = 10, K_JALR = 11, K_OJ = 12, K_CMN = 13, K_CMZ = 14, N_CLASS = 15,
             K_ECALL = 15 /* modes 2 / 3: the class id of the ECALL word; it has no column */,
             K_EBREAK = 21 /* modes 2 / 3 (format v11): the class id of the EBREAK word — NO selector carries it, so I_OPCLASS cannot hold on an executed EBREAK row: the VM halts there
                              (execute.rs:667), only the halt row (whose class is not the word's) can sit on it.  Modes 0 / 1: "other", as in v10 */ };
// ---- prover parameters (round 5; VERDICT r4 task 6): defined ONCE here for the prover (stark_prove.inl) and the verifier (verify.cpp).  DEFAULT_*: what a zero
// zkir_public_inputs.fri_params means; a proof may say more queries / grinding bits (up to MAX_*), never fewer than the defaults.  Conjectured FRI soundness at blow-up 2:
// one bit per query + the grinding bits (50 + 12 = 62; 84 + 16 = 100) — the capacity-4 Poseidon2 sponge caps collision resistance at ~62 bits either way (README).
constexpr int DEFAULT_NUM_QUERIES = 50, DEFAULT_POW_BITS = 12, MAX_NUM_QUERIES = 128, MAX_POW_BITS = 24, LOG_FINAL = 3, LOG_ARITY = 3;
BB_HD constexpr int num_queries_of(uint32_t fri_params) { return (fri_params & 0xFFFF) ? (int)(fri_params & 0xFFFF) : DEFAULT_NUM_QUERIES; }
BB_HD constexpr int pow_bits_of(uint32_t fri_params) { return (fri_params >> 16) ? (int)(fri_params >> 16) : DEFAULT_POW_BITS; }
BB_HD constexpr bool fri_params_ok(uint32_t fri_params) { return num_queries_of(fri_params) >= DEFAULT_NUM_QUERIES && num_queries_of(fri_params) <= MAX_NUM_QUERIES && pow_bits_of(fri_params) >= DEFAULT_POW_BITS && pow_bits_of(fri_params) <= MAX_POW_BITS; }
// proof format: modes 0 / 1 are v10 word for word; modes 2 / 3 are v11 (EBREAK class, the I/O section in the transcript, no access to the code segment in mode 3)
BB_HD constexpr uint32_t proof_version(int mode) { return mode == 4 ? 12u : mode >= 2 ? 11u : 10u; }     // (v12 = mode 4: the wide-arithmetic class)
constexpr uint64_t CODE_BASE = 0x1000;
BB_HD constexpr int kcol(int k) { return k < 7 ? C_K + k : k < 11 ? C_K2 + (k - 7) : k < 13 ? C_K3 + (k - 11) : C_K4 + (k - 13); }
BB_HD constexpr bool class_absent(int k, int mode) { return is_virtual(kcol(k), mode) || (mode == 4 && k == K_OTH); }   // a class no row of the mode has: its selector reads as zero (mode 4: "other" — its column carries ot)
static_assert(C_OT == kcol(K_OTH), "mode 4 re-uses the class column 'other' as the wide-tape selector");
constexpr uint32_t OP_ADD = 0x00, OP_SUB = 0x01, OP_ADDI = 0x08, OP_SLTU = 0x20, OP_SGEU = 0x21, OP_SLT = 0x22, OP_SGE = 0x23, OP_SEQ = 0x24, OP_CMOV = 0x26, OP_CMOVZ = 0x27, OP_CMOVNZ = 0x28, OP_SNE = 0x25, OP_BEQ = 0x40, OP_BNE = 0x41,
                   OP_BLT = 0x42, OP_BGE = 0x43, OP_BLTU = 0x44, OP_BGEU = 0x45, OP_JAL = 0x48, OP_JALR = 0x49, OP_ECALL = 0x50, OP_EBREAK = 0x51;
BB_HD constexpr uint32_t opclass_of(uint32_t op, int mode = 0) {
  return (op == OP_ECALL && mode >= 2) ? (uint32_t)K_ECALL : (op == OP_EBREAK && mode >= 2) ? (uint32_t)K_EBREAK : (mode >= 3 && is_load(op)) ? (uint32_t)K_LD : (mode >= 3 && is_store(op)) ? (uint32_t)K_ST : (mode >= 3 && is_logic(op)) ? (uint32_t)K_LG : (mode == 4 && is_wide(op)) ? (uint32_t)K_WA : (mode >= 3 && is_shift(op)) ? (uint32_t)K_SH : (mode >= 3 && op == OP_MUL_) ? (uint32_t)K_MU : op == OP_ADD ? K_ADD : op == OP_ADDI ? K_ADDI : (op == OP_BEQ || op == OP_BNE) ? K_BRE : op == OP_JAL ? K_JAL : op == OP_SUB ? K_SUB
       : (op == OP_BLTU || op == OP_BGEU || op == OP_BLT || op == OP_BGE) ? K_BRU : (op == OP_SEQ || op == OP_SNE) ? K_SE
       : (op == OP_SLTU || op == OP_SGEU || op == OP_SLT || op == OP_SGE) ? K_SU : op == OP_JALR ? K_JALR : (op == OP_CMOV || op == OP_CMOVNZ) ? K_CMN : op == OP_CMOVZ ? K_CMZ
       : (uint32_t)K_OTH;
}
BB_HD constexpr uint32_t family_base(int k) { return k == K_BRE ? OP_BEQ : k == K_BRU ? OP_BLT : k == K_SE ? OP_SEQ : k == K_SU ? OP_SLTU : 0u; }   // the lowest opcode of a family
// AIR v5: the families of ordered comparisons have FOUR members, op = base + 2 g + pol: SLTU SGEU SLT SGE (base 0x20, g = signed) and BLT BGE BLTU
// BGEU (base 0x42, g = unsigned).  g is the word's VARIANT BIT, the eleventh element of the ROM tuple (0 for every other opcode).
BB_HD constexpr uint32_t variant_bit(uint32_t op, int mode = 0) { return (op == OP_SLT || op == OP_SGE || op == OP_BLTU || op == OP_BGEU || (mode == 4 && (op == 0x06 || op == 0x07))) ? 1u : 0u; }   // (mode 4: DIV / REM are the signed variants of DIVU / REMU)

// constraint indices (the order of oracle/stark_oracle.cpp: constraints_sum)
enum : int { I_CYCLE = 0, I_CYCLE0 = 1, I_ENTRY = 2, I_ZERO0 = 5, I_HALT = 69, I_R0 = 70, I_BOOL_STATE = 74, I_BOOL_SEL = 90, I_BOOL_K = 135, I_BOOL_MISC = 150,
             I_ONE_CLASS = 161, I_OPCLASS = 162, I_WR = 163, I_SELB = 166, I_SELC = 168, I_CMOV = 170, I_OPERAND = 173, I_VALUE = 179, I_DIFF = 188, I_WRITTEN = 197,
             I_CMOV_Y = 202, I_Y2 = 205, I_NZ = 207, I_NE = 209, I_FLAG = 213, I_FX = 214, I_TK = 215, I_DL0 = 216, I_SE = 217, I_PC = 218, I_PC_KEEP = 221, I_JALR = 224, I_REGS = 227,
             I_TAIL = 287, I_LAST = 290, I_RANGE = 358, I_ROM = 390, I_SUM = 394, N_CONSTRAINTS_BASE = 398,
             // mode 2 (appended): syscall flags boolean (6), h-bits on hash rows only (2), the syscall number (3), what an ecall writes (3), zero results (3), the counters (2),
             // "exhausted" (1), the output lookup (4), the input lookup (4), the counters of the first / last row (2 + 2)
             I_IO_BOOL = 398, I_IO_H = 404, I_IO_R10 = 406, I_IO_WR = 409, I_IO_Y = 412, I_IO_CNT = 415, I_IO_END = 417, I_IO_OUT = 418, I_IO_IN = 422, I_IO_FIRST = 426, I_IO_LAST = 428,
             N_CONSTRAINTS_IO = 430,
             // mode 3 (appended): booleans (kld, kst, 15 windows, sgb, sgh, tb, cm2: 21), one window (1), no hash syscall (1), the opcode names width and sign (2), sgb / sgh
             // only on byte / halfword loads (2), what is written (2), the address (3), time order (1), the stored register's pieces (3), y (3), sign extension (3),
             // FPN (4), a load keeps the cell (4), the tuple read / written (4 + 4), the nine piece lookups (36)
             I_MEM_BOOL = 430, I_MEM_ONE = 451, I_MEM_NOHASH = 452, I_MEM_OP = 453, I_MEM_SG = 455, I_MEM_WR = 457, I_MEM_EA = 459, I_MEM_DT = 462, I_MEM_ST = 463, I_MEM_Y = 466,
             I_MEM_SX = 469, I_MEM_FPN = 472, I_MEM_KEEP = 476, I_MEM_RW = 480, I_MEM_PIECE = 488,
             // the bitwise opcodes (mode 3, appended): booleans klg oa oo ox li (5), the opcode (1), li only on bitwise rows (1), rd (1), rs1's nibbles (2), the second operand's (2),
             // the result (3), no b_k / r_k off the bitwise rows (20)
             I_LG_BOOL = 524, I_LG_OP = 529, I_LG_LI = 530, I_LG_WR = 531, I_LG_A = 532, I_LG_B = 534, I_LG_Y = 536, I_LG_ZERO = 539,
             // the shifts (mode 3, appended): booleans ksh ul ur v sa si sb9 (24), one chunk / bit shift (2), the opcode (3), rd (1), a's chunks (2), c_i 2^v (4) = lo + 2^10 hi (4),
             // the sign (2), the amount (5), sh vs t (1), d's guards (3), t <= 40 (1), 2^40 - 2^t (2), the result (3)
             I_SH_BOOL = 559, I_SH_ONE = 583, I_SH_OP = 585, I_SH_WR = 588, I_SH_A = 589, I_SH_PR = 591, I_SH_LOHI = 595, I_SH_SIGN = 599, I_SH_AMT = 601, I_SH_T = 606, I_SH_D = 607,
             I_SH_T40 = 610, I_SH_ON = 611, I_SH_Y = 613,
             // MUL (mode 3, appended): booleans kmu e_1..3 (4), rd (1), a's chunks (2), b's (2), ma_k = kmu a_k (4), the four chunk equations (4), the result (3)
             I_MU_BOOL = 616, I_MU_WR = 620, I_MU_A = 621, I_MU_B = 623, I_MU_MA = 625, I_MU_EQ = 629, I_MU_Y = 633,
             N_CONSTRAINTS_MEM = 636,
             // mode 4 (appended): the wide-arithmetic class — booleans kwa (= om + od + orr) om od orr e_1..9 (13), the opcode (1), rd (1), the operands' top limbs (2), F2 = rs2
             // (2), F1 = rs1 on MULH (2), LO = rs1 on divisions (2), gf_k = kwa F1_k (4), the seven positions of the product (7), r < rs2 (2), what is written (5), the six extra range
             // lookups (24); the boundary cell (2); the hash-call lookup (4)
             I_WA_BOOL = 636, I_WA_OP = 649, I_WA_WR = 650, I_WA_TOP = 651, I_WA_F2 = 653, I_WA_F1 = 655, I_WA_LO = 657, I_WA_GF = 659, I_WA_EQ = 663,
             I_WA_LT = 670, I_WA_Y = 672, I_WA_X = 677, I_BC = 701, I_HH = 703,
             I_WT_BOOL = 707, I_WW = 708,                              // the wide tape: ot boolean (1), the record lookup (4)
             N_CONSTRAINTS = 712 };
BB_HD constexpr int num_constraints(int mode) { return mode == 4 ? N_CONSTRAINTS : mode == 3 ? N_CONSTRAINTS_MEM : mode == 2 ? N_CONSTRAINTS_IO : N_CONSTRAINTS_BASE; }
// Boundary states (proof format v4): the 68 state words (cycle, 3 pc limbs, 48 register limbs, 16 storage states) of row 0 and of the
// last executed row are public; constraint 1 + i pins state word i of row 0, constraint I_LAST + i that of row n_real - 1.
constexpr int N_STATE = 68;
BB_HD constexpr int state_col(int i) { return i < 4 ? i : C_LIMB + (i - 4); }

// canonical constant -> Montgomery form at compile time
constexpr uint32_t M(uint64_t v) { return (uint32_t)(((v % bb::P) * (uint64_t)bb::R1) % bb::P); }
struct RegConsts { uint32_t r[16], r2[16]; };
constexpr RegConsts make_reg_consts() { RegConsts c{}; for (int i = 0; i < 16; i++) { c.r[i] = M((uint64_t)i); c.r2[i] = M((uint64_t)i * i); } return c; }

// `Ops` supplies the value type and its arithmetic (values in Montgomery form throughout):
//   using V;  V cst(uint32_t montgomery_constant);  V add(V,V), sub(V,V), mul(V,V);  V mulc(V, uint32_t montgomery_constant);
//   V loc(int column), nxt(int column);  V aloc(int aux column), anxt(int aux column);  V par(int lookup parameter: LK_*);
//   V loc_r(int column), nxt_r(int column): the same for a column index that is only known at run time (inside a rolled loop), never an uncommitted column;
//   void end_boundary(): no push_fc / push_lc follows (the quotient kernel folds its two boundary sums and frees their registers); end_trans(): no push_t follows;
//   LAZY values (round 4; the verifier's Ops treats them as plain values): V lsub(V,V), ladd(V,V) — difference / sum of two REDUCED values left
//   unreduced (below 2p); V lmul(V a, V b) — product of any 32-bit a with a reduced b, left below 2p.  A lazy value may only be (i) the FIRST
//   operand of mul / mulc / lmul, (ii) an operand of acc_mul / acc_lin, (iii) pushed.  air::BoundOps (below) runs this very function on value BOUNDS and
//   fails where a rule is broken (tests/test_abi.py), so the rules are checked, not trusted.
//   sums of products:  using AccP; AccP accp();  void acc_mul(AccP&, V a, V b)  (any 32-bit words);  V acc_val(const AccP&)  -> reduced
//   small-constant combinations:  using AccL; AccL accl();  void acc_lin(AccL&, V a, uint32_t k)  (adds k * a; k a small integer, NOT Montgomery);  V accl_val(const AccL&)
//   constraints (idx = position in the list = power of alpha; the value may be lazy):
//     push(idx, v)               C = v
//     push_t(idx, v)             C = v * is_trans
//     push_fc(idx, v, cm)        C = (v - cm) * is_first          push_lc(idx, v, cm)   C = (v - cm) * is_last       (cm a Montgomery constant: a public boundary word)
//     push_fc0(idx, cm)          the same for a column that is not committed (v = 0)                                 push_lc0(idx, cm)
//   The selectors live in the Ops: the quotient kernel keeps one accumulator per selector and multiplies ONCE per point; the boundary words'
//   share  - sum alpha^idx cm  is a per-proof constant the prover's host side forms (boundary_constant below).
// first_m / last_m: the public boundary states (Montgomery).
BB_HD constexpr int first_idx(int i) { return i == 0 ? I_CYCLE0 : i < 4 ? I_ENTRY + (i - 1) : i < 52 ? I_ZERO0 + (i - 4) : I_ZERO0 + 48 + (i - 52); }   // constraint pinning state word i of row 0
BB_HD constexpr int last_idx(int i) { return I_LAST + i; }                                                                                              // .. of row n_real - 1

// compile-time loop: the body sees its index as a constant (std::integral_constant)
template <int I> struct air_ic { static constexpr int value = I; };
template <int K, int N, class F>
BB_HD void air_static_for(F&& f) {
  if constexpr (K < N) { f(air_ic<K>{}); air_static_for<K + 1, N>(f); }
}

template <class Ops>
BB_HD void eval(Ops& o, const uint32_t* first_m, const uint32_t* last_m, int mode, const uint32_t* cnt_m = nullptr) {   // cnt_m (mode 2): (oc, ic) of the first row, of the last row (Montgomery)
  const bool deferred = mode == 1, io = mode >= 2, mem = mode >= 3, wide = mode == 4;
  using V = typename Ops::V;
  using AccP = typename Ops::AccP;
  using AccL = typename Ops::AccL;
  const V one = o.cst(bb::R1), zero = o.cst(0);
  constexpr uint32_t PM1 = bb::P - bb::R1;                     // -1 (Montgomery): b + PM1 is the lazy b - 1
  auto boolean = [&](int idx, V b) { o.push(idx, o.lmul(o.ladd(b, o.cst(PM1)), b)); };
  // ---- A. boundary rows: state word i of row 0 / of row n_real - 1 is the public one; the last executed row is the halt row -------------
  // (unrolled at compile time: the columns are constants, so the quotient kernel reads them as 16-byte vectors, all in flight at once)
  // and FIRST in program order, the pushes after them: one memory latency for the phase, not one per column)
  {
    V sv[N_STATE];
    air_static_for<0, N_STATE>([&](auto ic) { constexpr int i = decltype(ic)::value, col = state_col(i); sv[i] = is_virtual(col, mode) ? zero : o.loc(col); });
    const V khalt = o.loc(kcol(K_HALT));
    air_static_for<0, N_STATE>([&](auto ic) {
      constexpr int i = decltype(ic)::value, col = state_col(i);
      if (is_virtual(col, mode)) { o.push_fc0(first_idx(i), first_m[i]); o.push_lc0(last_idx(i), last_m[i]); return; }
      o.push_fc(first_idx(i), sv[i], first_m[i]); o.push_lc(last_idx(i), sv[i], last_m[i]);
    });
    o.push_lc(I_HALT, khalt, bb::R1);
    if (io) {                                                  // (mode 2) the counters of the first and of the last executed row are public too
      const V oc = o.loc(C_OC), ic = o.loc(C_IC);
      o.push_fc(I_IO_FIRST, oc, cnt_m[0]); o.push_fc(I_IO_FIRST + 1, ic, cnt_m[1]);
      o.push_lc(I_IO_LAST, oc, cnt_m[2]); o.push_lc(I_IO_LAST + 1, ic, cnt_m[3]);
    }
  }
  o.end_boundary();
  // ---- B. transitions (x is_trans): registers, cycle counter, next pc, the tail of the trace ------------------------------------------------
  // registers: unwritten ones keep their limbs and storage state, the written one shows y:  nx - cur - wr (tgt - cur);  in the same pass the
  // selector moments (sums over the registers with small-constant weights) and the selected operands (sums of products).
  // A ROLLED loop (the quotient kernel's code size), so the columns are run-time indices (loc_r / nxt_r: never an uncommitted column), software-
  // pipelined: register r + 1's words are requested before register r's are worked on.  It runs BEFORE the row's other columns are read: few values
  // are live across it.
  const V y[3] = {o.loc(C_Y), o.loc(C_Y + 1), o.loc(C_Y + 2)};
  AccL w0 = o.accl(), w1 = o.accl(), w2 = o.accl(), b1 = o.accl(), b2 = o.accl(), c1a = o.accl(), c2a = o.accl();
  AccP xbs[3] = {o.accp(), o.accp(), o.accp()}, xcs[3] = {o.accp(), o.accp(), o.accp()};
  struct RegIn { V wr, sb, sc, limb[3], nx[3], st, nst; };
  auto load_reg = [&](int r) {
    RegIn g;
    g.wr = o.loc_r(C_WR + r - 1); g.sb = o.loc_r(C_SELB + r - 1); g.sc = o.loc_r(C_SELC + r - 1);
#pragma unroll
    for (int l = 0; l < 3; l++) { g.limb[l] = o.loc_r(C_LIMB + 3 * r + l); g.nx[l] = o.nxt_r(C_LIMB + 3 * r + l); }
    if (deferred) { g.st = o.loc_r(C_STATE + r); g.nst = o.nxt_r(C_STATE + r); } else { g.st = zero; g.nst = zero; }
    return g;
  };
  RegIn cur = load_reg(1);
#pragma unroll 1
  for (int r = 1; r < 16; r++) {
    const RegIn ahead = load_reg(r + 1 < 16 ? r + 1 : r);
    const V wr = cur.wr, sb = cur.sb, sc = cur.sc;
    boolean(I_BOOL_SEL + 3 * (r - 1), wr); boolean(I_BOOL_SEL + 3 * (r - 1) + 1, sb); boolean(I_BOOL_SEL + 3 * (r - 1) + 2, sc);
    o.acc_lin(w0, wr, 1); o.acc_lin(w1, wr, (uint32_t)r); o.acc_lin(w2, wr, (uint32_t)(r * r));
    o.acc_lin(b1, sb, (uint32_t)r); o.acc_lin(b2, sb, (uint32_t)(r * r));
    o.acc_lin(c1a, sc, (uint32_t)r); o.acc_lin(c2a, sc, (uint32_t)(r * r));
#pragma unroll
    for (int l = 0; l < 3; l++) {
      const V limb = cur.limb[l], nx = cur.nx[l];
      o.acc_mul(xbs[l], sb, limb); o.acc_mul(xcs[l], sc, limb);
      // nx - cur - wr * ((1 - D) y + D nx - cur)
      const V tgt = deferred ? nx : y[l];
      o.push_t(I_REGS + 4 * (r - 1) + l, o.lsub(o.sub(nx, limb), o.mul(o.lsub(tgt, limb), wr)));
    }
    if (deferred) {                                              // (default mode: no storage-state columns, the constraint is 0 = 0)
      const V st = cur.st, nst = cur.nst;
      boolean(I_BOOL_STATE + r, st);
      o.push_t(I_REGS + 4 * (r - 1) + 3, o.lsub(o.sub(nst, st), o.mul(o.lsub(nst, st), wr)));
    }
    cur = ahead;
  }
  // every other column of the row pair, read HERE — together, before any of them is used (one memory latency)
  V K[N_CLASS];
#pragma unroll
  for (int k = 0; k < N_CLASS; k++) K[k] = class_absent(k, mode) ? zero : o.loc(kcol(k));          // (mode 4: no class "other" — its column is ot, the wide-tape selector)
  const V pc[3] = {o.loc(C_PC), o.loc(C_PC + 1), o.loc(C_PC + 2)};
  const V npc[3] = {o.nxt(C_PC), o.nxt(C_PC + 1), o.nxt(C_PC + 2)};
  const V cyc = o.loc(C_CYCLE), ncyc = o.nxt(C_CYCLE), npad = o.nxt(C_K + K_PAD);
  const V d0 = o.loc(C_D0), d1 = o.loc(C_D1), d2 = o.loc(C_D2), dl0 = o.loc(C_DL0), se = o.loc(C_SE), s = o.loc(C_S);
  const V op = o.loc(C_OP), fa = o.loc(C_FA), fb = o.loc(C_FB), fc = o.loc(C_FC), fhi = o.loc(C_FHI), opc = o.loc(C_OPC), g = o.loc(C_G);
  V xb[3], xc[3], iv[3];
#pragma unroll
  for (int l = 0; l < 3; l++) { xb[l] = o.loc(C_XB + l); xc[l] = o.loc(C_XC + l); iv[l] = o.loc(C_IV + l); }
  V R[4], R2[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { R[i] = o.loc(C_RC + i); R2[i] = o.loc(C_RC2 + i); }
  const V c0 = o.loc(C_C0), c1 = o.loc(C_C1), ne = o.loc(C_NE), tk = o.loc(C_TK), nz = o.loc(C_NZ), q = o.loc(C_Q), sa = o.loc(C_B0), sbit = o.loc(C_SB);
  const V flag = o.loc(C_FLAG), fx = o.loc(C_FX), ivz = o.loc(C_IVZ);
  const V hp = o.add(K[K_HALT], K[K_PAD]);
  const V imm17 = o.add(fc, o.mulc(fhi, M(16)));
  const V im0 = o.add(o.sub(imm17, o.mulc(s, M(1u << 17))), o.mulc(s, M(1u << 20))), im1 = o.mulc(s, M(0xFFFFF));
  o.push_t(I_CYCLE, o.lsub(o.sub(ncyc, cyc), one));
  // (R0 is hard-wired zero: its limbs and storage state are not committed in either mode — constraints I_R0 .. I_R0 + 3 and I_BOOL_STATE read 0 = 0)
  static_assert(is_virtual(C_LIMB, false) && is_virtual(C_LIMB + 2, true) && is_virtual(C_STATE, false) && is_virtual(C_STATE, true), "R0's columns are not committed");
  // next pc.  kc: pc' = pc + delta for every class but jalr, oj (free) and halt / pad (keep); class "other" is in it with delta 4 (tk = 0 there): sequential
  {
    const V kc = o.sub(o.sub(o.sub(one, K[K_JALR]), K[K_OJ]), hp);
    o.push_t(I_PC, o.lmul(o.add(o.sub(o.sub(npc[0], pc[0]), dl0), o.mulc(d0, M(1u << 20))), kc));
    o.push_t(I_PC + 1, o.lmul(o.add(o.sub(o.sub(o.sub(npc[1], pc[1]), o.mulc(se, M(0xFFFFF))), d0), o.mulc(d1, M(1u << 20))), kc));
    o.push_t(I_PC + 2, o.lmul(o.add(o.sub(o.sub(o.sub(npc[2], pc[2]), o.mulc(se, M(0xFFFFFF))), d1), o.mulc(d2, M(1u << 24))), kc));
#pragma unroll
    for (int l = 0; l < 3; l++) o.push_t(I_PC_KEEP + l, o.lmul(o.lsub(npc[l], pc[l]), hp));
    // JALR: pc' + b0 = rs1 + sext(imm17) mod 2^64 over (20, 20, 24)-bit limbs, b0 = the bit that is cleared (execute.rs:649-658); the limbs
    // of pc' are a code address (every row's pc is looked up in the ROM), so the carries and b0 are forced
    o.push_t(I_JALR, o.lmul(o.lsub(o.add(o.add(npc[0], sa), o.mulc(d0, M(1u << 20))), o.add(xb[0], im0)), K[K_JALR]));
    o.push_t(I_JALR + 1, o.lmul(o.lsub(o.add(npc[1], o.mulc(d1, M(1u << 20))), o.add(o.add(xb[1], im1), d0)), K[K_JALR]));
    o.push_t(I_JALR + 2, o.lmul(o.lsub(o.add(npc[2], o.mulc(d2, M(1u << 24))), o.add(o.add(xb[2], o.mulc(s, M(0xFFFFFF))), d1)), K[K_JALR]));
    // executed rows, the halt row, padding
    const V one_m_npad = o.sub(one, npad);
    o.push_t(I_TAIL, o.lmul(one_m_npad, K[K_HALT]));
    o.push_t(I_TAIL + 1, o.lmul(one_m_npad, K[K_PAD]));
    o.push_t(I_TAIL + 2, o.lmul(o.sub(o.sub(one, K[K_PAD]), K[K_HALT]), npad));
  }
  if (io) {                                                    // (mode 2) oc counts the WRITE ecalls, ic the inputs consumed (syscall.rs:110-121)
    o.push_t(I_IO_CNT, o.lsub(o.sub(o.nxt(C_OC), o.loc(C_OC)), o.loc(C_F2)));
    o.push_t(I_IO_CNT + 1, o.lsub(o.sub(o.nxt(C_IC), o.loc(C_IC)), o.loc(C_RL)));
  }
  o.end_trans();
  // ---- C. row-local constraints -----------------------------------------------------------------------------------------------------------
  const V Kbr = o.add(K[K_BRE], K[K_BRU]), Kcmp = o.add(K[K_SE], K[K_SU]);       // B-type rows; comparison rows (the flag is the value written)
  const V z[2] = {o.add(R[0], o.mulc(R[1], M(RC_TABLE))), o.add(R[2], o.mulc(R[3], M(RC_TABLE)))};   // (v6) z IS its chunks: no columns of its own
  // booleans
#pragma unroll
  for (int k = 0; k < N_CLASS; k++) { if (!class_absent(k, mode)) boolean(I_BOOL_K + k, K[k]); }
  boolean(I_BOOL_MISC, s); boolean(I_BOOL_MISC + 1, c0); boolean(I_BOOL_MISC + 2, c1); boolean(I_BOOL_MISC + 3, d0); boolean(I_BOOL_MISC + 4, d1);
  boolean(I_BOOL_MISC + 5, d2); boolean(I_BOOL_MISC + 6, ne); boolean(I_BOOL_MISC + 7, tk); boolean(I_BOOL_MISC + 8, sa); boolean(I_BOOL_MISC + 9, sbit); boolean(I_BOOL_MISC + 10, nz);
  // classes and the opcode
  V F2 = zero, RL = zero, RE = zero, FH = zero, H0 = zero, H1 = zero;   // (mode 2) the syscall flags of an ECALL row; Kec = their sum is the row's class
  if (io) { F2 = o.loc(C_F2); RL = o.loc(C_RL); RE = o.loc(C_RE); FH = o.loc(C_FH); H0 = o.loc(C_H0); H1 = o.loc(C_H1); }
  V Kld = zero, Kst = zero, Klg = zero, Ksh = zero, Kmu = zero, Kwa = zero;   // (mode 3) loads, stores, the bitwise opcodes, the shifts, MUL; (mode 4) MULH DIVU REMU DIV REM
  if (mem) { Kld = o.loc(C_KLD); Kst = o.loc(C_KST); Klg = o.loc(C_KLG); Ksh = o.loc(C_KSH); Kmu = o.loc(C_KMU); }
  V Kin = zero;                                                            // (mode 4) the wide rows proven by the chunk relation: om + od + orr
  if (wide) { Kin = o.add(o.add(o.loc(C_OM), o.loc(C_OD)), o.loc(C_ORR)); Kwa = o.add(Kin, o.loc(C_OT)); }   // the class: kwa = om + od + orr + ot (ot: through the wide tape); no column of its own
  {
    AccL sum = o.accl(), ks = o.accl();
#pragma unroll
    for (int k = 0; k < N_CLASS; k++) {
      if (class_absent(k, mode)) continue;
      o.acc_lin(sum, K[k], 1);
      if (k >= 1 && k != K_HALT && k != K_PAD) o.acc_lin(ks, K[k], (uint32_t)k);
    }
    if (io) {
      o.acc_lin(sum, F2, 1); o.acc_lin(sum, RL, 1); o.acc_lin(sum, RE, 1); o.acc_lin(sum, FH, 1);
      o.acc_lin(ks, F2, K_ECALL); o.acc_lin(ks, RL, K_ECALL); o.acc_lin(ks, RE, K_ECALL); o.acc_lin(ks, FH, K_ECALL);
    }
    if (mem) { o.acc_lin(sum, Kld, 1); o.acc_lin(sum, Kst, 1); o.acc_lin(sum, Klg, 1); o.acc_lin(sum, Ksh, 1); o.acc_lin(ks, Kld, K_LD); o.acc_lin(ks, Kst, K_ST); o.acc_lin(ks, Klg, K_LG); o.acc_lin(ks, Ksh, K_SH); o.acc_lin(sum, Kmu, 1); o.acc_lin(ks, Kmu, K_MU); }
    if (wide) { o.acc_lin(sum, Kwa, 1); o.acc_lin(ks, Kwa, K_WA); }
    o.push(I_ONE_CLASS, o.lsub(o.accl_val(sum), one));
    // an executed row runs as the class of its instruction word: (1 - halt - pad) opclass = sum_k k K_k; opclass comes with the ROM tuple
    if (!deferred) o.push(I_OPCLASS, o.lsub(o.mul(o.lsub(one, hp), opc), o.accl_val(ks)));
  }
  // selectors
  const V w0v = o.accl_val(w0), w1v = o.accl_val(w1), b1v = o.accl_val(b1), c1v = o.accl_val(c1a);
  if (!deferred) o.push(I_WR, o.lsub(o.mul(w1v, w1v), o.accl_val(w2)));
  o.push(I_WR + 1, o.lmul(o.lsub(w1v, fa), o.add(o.add(o.add(o.add(K[K_ADD], K[K_ADDI]), o.add(K[K_JAL], K[K_SUB])), Kcmp), K[K_JALR])));
  o.push(I_WR + 2, o.lmul(w0v, o.add(o.add(o.add(Kbr, deferred ? zero : K[K_OJ]), K[K_HALT]), K[K_PAD])));   // branches write nothing (BLT / BGE too: class oj in default mode)
  o.push(I_SELB, o.lsub(b1v, fb)); o.push(I_SELB + 1, o.lsub(o.mul(b1v, b1v), o.accl_val(b2)));
  o.push(I_SELC, o.lsub(c1v, o.add(fc, o.mul(o.lsub(fa, fc), mem ? o.add(Kbr, Kst) : Kbr)))); o.push(I_SELC + 1, o.lsub(o.mul(c1v, c1v), o.accl_val(c2a)));   // (S-type words: rs1 in field a, like B-type ones)
  // (v6) conditional moves CMOV / CMOVNZ (class cmn: the condition is rs2 != 0) and CMOVZ (class cmz: rs2 == 0), execute.rs:434-472: q = "this row is a
  //      conditional move whose condition holds"; it writes rd = field a exactly then (nothing at all otherwise)
  const V Kcm = o.add(K[K_CMN], K[K_CMZ]);
  o.push(I_CMOV, o.lsub(q, o.add(o.mul(K[K_CMN], nz), o.mul(K[K_CMZ], o.sub(one, nz)))));
  o.push(I_CMOV + 1, o.lmul(o.lsub(w1v, fa), q));
  o.push(I_CMOV + 2, o.lmul(w0v, o.sub(Kcm, q)));
  // operands
#pragma unroll
  for (int l = 0; l < 3; l++) { o.push(I_OPERAND + 2 * l, o.lsub(xb[l], o.acc_val(xbs[l]))); o.push(I_OPERAND + 2 * l + 1, o.lsub(xc[l], o.acc_val(xcs[l]))); }
  // values written
  const V lo20 = o.sub(o.add(o.add(fb, o.mulc(fc, M(16))), o.mulc(fhi, M(256))), o.mulc(s, M(1u << 20)));
  const V c0s20 = o.mulc(c0, M(1u << 20)), c1s20 = o.mulc(c1, M(1u << 20));
  // stated on z, the range-checked pair of limbs; y = z on the rows that write it and on "other" rows (whose y stays in range)
  o.push(I_VALUE, o.lmul(o.add(o.sub(o.sub(z[0], xb[0]), xc[0]), c0s20), K[K_ADD]));
  o.push(I_VALUE + 1, o.lmul(o.add(o.sub(o.sub(o.sub(z[1], xb[1]), xc[1]), c0), c1s20), K[K_ADD]));
  o.push(I_VALUE + 2, o.lmul(o.ladd(K[K_ADD], K[K_SUB]), y[2]));
  o.push(I_VALUE + 3, o.lmul(o.add(o.sub(o.sub(z[0], xb[0]), im0), c0s20), K[K_ADDI]));
  o.push(I_VALUE + 4, o.lmul(o.add(o.sub(o.sub(o.sub(z[1], xb[1]), im1), c0), c1s20), K[K_ADDI]));
  o.push(I_VALUE + 5, o.lmul(K[K_ADDI], y[2]));
  const V Kj = o.add(K[K_JAL], K[K_JALR]);                                    // both link pc + 4 (execute.rs:639-658)
  o.push(I_VALUE + 6, o.lmul(o.add(o.sub(o.sub(z[0], pc[0]), o.cst(M(4))), c0s20), Kj));
  o.push(I_VALUE + 7, o.lmul(o.add(o.sub(o.sub(z[1], pc[1]), c0), c1s20), Kj));
  o.push(I_VALUE + 8, o.lmul(o.lsub(o.sub(y[2], pc[2]), c1), Kj));
  // differences with borrows: z = xb - xc mod 2^40 on SUB and SLTU / SGEU rows (execute.rs:65-77, :373-407), z = xc - xb on BLTU / BGEU
  // rows (:618-636): c1 = 1 exactly when the minuend is the smaller 40-bit value
  {
    const V Ks = o.add(K[K_SUB], K[K_SU]);
    // (v5) ordered comparisons, signed or not (op = base + 2 g + pol; sgn = g on SLTU.. rows, 1 - g on BLT.. rows): the high limbs enter BIASED,
    // ta = a1 + 2^19 sgn - 2^20 sa, tb likewise with sb — the limbs of value XOR 2^39 when sgn = 1 (Value40::signed_lt, value.rs:710-716) — so
    // ta - tb = a1 - b1 - 2^20 (sa - sb); u = (ta, tb) is the row's second range-checked pair, which forces sa / sb to be the sign bits (0 if sgn = 0)
    const V sab = o.mulc(o.lsub(sa, sbit), M(1u << 20));
    o.push(I_DIFF, o.lmul(o.lsub(o.add(o.sub(z[0], xb[0]), xc[0]), c0s20), Ks));
    const V dsub = o.sub(o.add(o.add(o.sub(z[1], xb[1]), xc[1]), c0), c1s20);
    o.push(I_DIFF + 1, o.lmul(dsub, K[K_SUB]));
    o.push(I_DIFF + 2, o.lmul(o.ladd(dsub, sab), K[K_SU]));
    o.push(I_DIFF + 3, o.lmul(o.lsub(o.add(o.sub(z[0], xc[0]), xb[0]), c0s20), K[K_BRU]));
    o.push(I_DIFF + 4, o.lmul(o.ladd(o.sub(o.add(o.add(o.sub(z[1], xc[1]), xb[1]), c0), c1s20), sab), K[K_BRU]));
    const V u0 = o.add(R2[0], o.mulc(R2[1], M(RC_TABLE))), u1 = o.add(R2[2], o.mulc(R2[3], M(RC_TABLE)));
    const V gsu = o.mulc(g, M(1u << 19)), gbr = o.mulc(o.lsub(one, g), M(1u << 19));
    const V sa20 = o.mulc(sa, M(1u << 20)), sb20 = o.mulc(sbit, M(1u << 20));
    o.push(I_DIFF + 5, o.lmul(o.ladd(o.sub(o.sub(u0, xb[1]), gsu), sa20), K[K_SU]));
    o.push(I_DIFF + 6, o.lmul(o.ladd(o.sub(o.sub(u1, xc[1]), gsu), sb20), K[K_SU]));
    o.push(I_DIFF + 7, o.lmul(o.ladd(o.sub(o.sub(u0, xc[1]), gbr), sa20), K[K_BRU]));
    o.push(I_DIFF + 8, o.lmul(o.ladd(o.sub(o.sub(u1, xb[1]), gbr), sb20), K[K_BRU]));
  }
  {
    const V Ky = o.add(o.add(o.add(o.add(K[K_ADD], K[K_ADDI]), o.add(K[K_JAL], K[K_SUB])), o.add(K[K_OTH], K[K_JALR])), deferred ? K[K_OJ] : zero);   // (oj writes only in deferred mode)
    o.push(I_WRITTEN, o.lmul(o.lsub(y[0], z[0]), Ky)); o.push(I_WRITTEN + 1, o.lmul(o.lsub(y[1], z[1]), Ky));
    o.push(I_WRITTEN + 2, o.lmul(o.lsub(y[0], fx), Kcmp)); o.push(I_WRITTEN + 3, o.lmul(Kcmp, y[1])); o.push(I_WRITTEN + 4, o.lmul(Kcmp, y[2]));
#pragma unroll
    for (int l = 0; l < 3; l++) o.push(I_CMOV_Y + l, o.lmul(o.lsub(y[l], xb[l]), Kcm));        // (v6) a conditional move writes rs1's raw value (all three limbs)
    // (v6) the bits above 40 of what an "other" row writes: y2 = R4 + 2^10 R5 + 2^20 R6 with R7 = 64 R6 — all four in the 10-bit table, so y2 < 2^24.  With it
    // EVERY limb of every register is in range by induction (constrained classes write 0, pc2 + c1 or an operand's limb there)
    o.push(I_Y2, o.lmul(o.lsub(o.sub(o.sub(y[2], R2[0]), o.mulc(R2[1], M(RC_TABLE))), o.mulc(R2[2], M(RC_TABLE * RC_TABLE))), K[K_OTH]));
    o.push(I_Y2 + 1, o.lmul(o.lsub(R2[3], o.mulc(R2[2], M(64))), K[K_OTH]));
  }
  // (v6) nz = [xc != 0] on every row, on the sum of xc's limbs (in range, so the sum vanishes only if they all do)
  {
    const V sx = o.add(o.add(xc[0], xc[1]), xc[2]);
    o.push(I_NZ, o.lmul(o.lsub(one, nz), sx));
    o.push(I_NZ + 1, o.lsub(nz, o.mul(sx, ivz)));
  }
  // BNE operands differ?
  {
    AccP dot = o.accp();
#pragma unroll
    for (int l = 0; l < 3; l++) {
      const V d = o.sub(xb[l], xc[l]);
      o.acc_mul(dot, d, iv[l]);
      o.push(I_NE + l, o.lmul(o.lsub(one, ne), d));
    }
    o.push(I_NE + 3, o.lsub(ne, o.acc_val(dot)));
  }
  // the family's comparison and its polarity: flag = [xb == xc] on BEQ / BNE / SEQ / SNE rows, the borrow c1 on the unsigned comparisons,
  // 0 elsewhere; fx = flag XOR pol, pol = op - the family's even opcode (0 / 1: the ROM ties op to the class); a branch is taken iff fx
  o.push(I_FLAG, o.lsub(o.sub(flag, o.mul(o.lsub(one, ne), o.add(K[K_BRE], K[K_SE]))), o.mul(o.ladd(K[K_BRU], K[K_SU]), c1)));
  {
    V pol = op;
#pragma unroll
    for (int k = 0; k < N_CLASS; k++) if (family_base(k)) pol = o.sub(pol, o.mulc(K[k], M(family_base(k))));
    pol = o.sub(pol, o.mulc(g, M(2)));                                // (v5) op = base + 2 g + pol in the four-member families; g = 0 elsewhere
    o.push(I_FX, o.ladd(o.sub(o.sub(fx, flag), pol), o.mulc(o.mul(pol, flag), M(2))));
  }
  o.push(I_TK, o.lsub(tk, o.mul(Kbr, fx)));
  // next pc: the delta and its sign extension
  const V four = o.cst(M(4));
  o.push(I_DL0, o.lsub(dl0, o.add(o.add(four, o.mul(o.lsub(im0, four), tk)), o.mul(o.lsub(lo20, four), K[K_JAL]))));
  o.push(I_SE, o.lsub(se, o.mul(o.ladd(tk, K[K_JAL]), s)));
  // ---- the lookup argument.  An extension-field column is four base columns (coordinates in F[X]/(X^4 - 11)); a relation between
  //      extension values is stated coordinate by coordinate.  out_k = sum_{i+j=k} h_i d_j + 11 sum_{i+j=k+4} h_i d_j  (h11_i = 11 h_i)
  auto ext_mul = [&](const V* h, const V* d, V* out) {
    V h11[4];
#pragma unroll
    for (int i = 1; i < 4; i++) h11[i] = o.mulc(h[i], M(11));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      AccP a = o.accp();
#pragma unroll
      for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) { if (i + j == k) o.acc_mul(a, h[i], d[j]); else if (i + j == k + 4) o.acc_mul(a, h11[i], d[j]); }
      }
      out[k] = o.acc_val(a);
    }
  };
  // the aux row pair and the tuple columns, read together before use
  V H[N_RC][4], hr[4], S[4], nS[4], tup[N_TUPLE];
#pragma unroll
  for (int i = 0; i < N_RC; i++)
#pragma unroll
    for (int k = 0; k < 4; k++) H[i][k] = o.aloc(A_H + 4 * i + k);
#pragma unroll
  for (int k = 0; k < 4; k++) { hr[k] = o.aloc(A_HR + k); S[k] = o.aloc(A_S + k); nS[k] = o.anxt(A_S + k); }
  air_static_for<0, N_TUPLE>([&](auto jc) { tup[decltype(jc)::value] = o.loc(tuple_col(decltype(jc)::value)); });
  // range helpers: H_i (alpha - R_i) = 1, i = 0..7 (the chunks of z, the chunks of u)
  AccL hs[4] = {o.accl(), o.accl(), o.accl(), o.accl()};                         // H0 + .. + H7 + HR, coordinate by coordinate (the running sum's increment)
#pragma unroll
  for (int i = 0; i < N_RC; i++) {
    V d[4], pr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { d[k] = o.par(LK_ALPHA + k); o.acc_lin(hs[k], H[i][k], 1); }
    d[0] = o.sub(d[0], i < 4 ? R[i] : R2[i - 4]);
    if (mem && i == 0) {                                       // (mode 3) a memory row's first chunk goes to the LOW3 table with the window's offset: alpha - R0 - lambda off - 4 lambda^11 Kmem
      AccL offa = o.accl();
#pragma unroll
      for (int v = 1; v < N_WIN; v++) if (win_start(v)) o.acc_lin(offa, o.loc(C_E + v), (uint32_t)win_start(v));
      const V off = o.accl_val(offa), Kmem = o.add(Kld, Kst);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        AccP t = o.accp();
        o.acc_mul(t, off, o.par(LK_LAM + 4 + k)); o.acc_mul(t, Kmem, o.mulc(o.par(LK_LAM + 4 * N_TUPLE + k), M(TAG_LOW3)));
        d[k] = o.sub(d[k], o.acc_val(t));
      }
    }
    if (mem && i == N_RC - 1) {                                // (mode 3) a bitwise row's tenth nibble tuple: alpha - R7 - lambda b_9 - lambda^2 r_9 - (8 oa + 9 oo + 10 ox) lambda^11
      AccL tga = o.accl();
      o.acc_lin(tga, o.loc(C_OA), TAG_AND); o.acc_lin(tga, o.loc(C_OO), TAG_OR); o.acc_lin(tga, o.sub(o.sub(Klg, o.loc(C_OA)), o.loc(C_OO)), TAG_XOR);
      const V tg = o.accl_val(tga), b9 = o.loc(C_LB + 9), r9 = o.loc(C_LR + 9);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        AccP t = o.accp();
        o.acc_mul(t, b9, o.par(LK_LAM + 4 + k)); o.acc_mul(t, r9, o.par(LK_LAM + 8 + k)); o.acc_mul(t, tg, o.par(LK_LAM + 4 * N_TUPLE + k));
        d[k] = o.sub(d[k], o.acc_val(t));
      }
    }
    ext_mul(H[i], d, pr);
    o.push(I_RANGE + 4 * i, o.lsub(pr[0], one));
#pragma unroll
    for (int k = 1; k < 4; k++) o.push(I_RANGE + 4 * i + k, pr[k]);
  }
  // instruction ROM: HR (alpha - fingerprint(tuple)) = 1, fingerprint = sum_j lambda^j f_j + lambda^N_TUPLE
  {
    V d[4], pr[4];
    AccP fp[4] = {o.accp(), o.accp(), o.accp(), o.accp()};
#pragma unroll
    for (int k = 0; k < 4; k++) o.acc_lin(hs[k], hr[k], 1);
#pragma unroll
    for (int j = 0; j < N_TUPLE; j++) {
#pragma unroll
      for (int k = 0; k < 4; k++) o.acc_mul(fp[k], tup[j], o.par(LK_LAM + 4 * j + k));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) d[k] = o.sub(o.sub(o.par(LK_ALPHA + k), o.par(LK_LAM + 4 * N_TUPLE + k)), o.acc_val(fp[k]));
    ext_mul(hr, d, pr);
    o.push(I_ROM, o.lsub(pr[0], one));
#pragma unroll
    for (int k = 1; k < 4; k++) o.push(I_ROM + k, pr[k]);
  }
  // ---- (mode 2, round 4) ECALL rows and the I/O tapes (syscall.rs:94-177): constraints 398.. of the oracle's list ------------------------------------
  if (io) {
    const V Kec = o.add(o.add(F2, RL), o.add(RE, FH));
    boolean(I_IO_BOOL, F2); boolean(I_IO_BOOL + 1, RL); boolean(I_IO_BOOL + 2, RE); boolean(I_IO_BOOL + 3, FH); boolean(I_IO_BOOL + 4, H0); boolean(I_IO_BOOL + 5, H1);
    const V nfh = o.lsub(one, FH);
    o.push(I_IO_H, o.lmul(nfh, H0)); o.push(I_IO_H + 1, o.lmul(nfh, H1));             // the two bits of R10 - 3 live on hash rows only
    // the syscall number: R10 = 1 (READ), 2 (WRITE), 3 + h0 + 2 h1 (hash); 0 (EXIT) halts — that row is the halt row, not an executed ecall
    const V r10[3] = {o.loc(C_LIMB + 30), o.loc(C_LIMB + 31), o.loc(C_LIMB + 32)}, r11[3] = {o.loc(C_LIMB + 33), o.loc(C_LIMB + 34), o.loc(C_LIMB + 35)};
    o.push(I_IO_R10, o.lmul(r10[1], Kec)); o.push(I_IO_R10 + 1, o.lmul(r10[2], Kec));
    {
      AccL num = o.accl();
      o.acc_lin(num, RL, 1); o.acc_lin(num, RE, 1); o.acc_lin(num, F2, 2); o.acc_lin(num, FH, 3); o.acc_lin(num, H0, 1); o.acc_lin(num, H1, 2);
      o.push(I_IO_R10 + 2, o.lsub(o.mul(r10[0], Kec), o.accl_val(num)));
    }
    // what an ecall writes: READ and the hashes write R10 and only R10, WRITE writes nothing; hashes and an exhausted READ return 0
    const V wgrp = o.add(o.add(RL, RE), FH);
    o.push(I_IO_WR, o.lmul(o.lsub(w1v, o.cst(M(10))), wgrp)); o.push(I_IO_WR + 1, o.lmul(o.lsub(w0v, one), wgrp)); o.push(I_IO_WR + 2, o.lmul(w0v, F2));
    const V zgrp = o.add(FH, RE);
#pragma unroll
    for (int l = 0; l < 3; l++) o.push(I_IO_Y + l, o.lmul(y[l], zgrp));
    const V oc = o.loc(C_OC), ic = o.loc(C_IC);
    o.push(I_IO_END, o.lmul(o.lsub(ic, o.par(LK_NIN)), RE));                          // "exhausted": every input has been consumed
    // the tape lookups: HO (alpha - fp(oc, R11)) = f2, HI (alpha - fp(ic, y)) = rl; fp = sum_j lambda^j g_j + tag lambda^N_TUPLE, tag 2 / 3
    auto tape = [&](int a_col, int idx0, const V& index, const V* limbs, uint32_t tag, const V& flag) {
      V h[4], d[4], pr[4];
      AccP fp[4] = {o.accp(), o.accp(), o.accp(), o.accp()};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        h[k] = o.aloc(a_col + k); o.acc_lin(hs[k], h[k], 1);
        o.acc_mul(fp[k], index, o.par(LK_LAM + k));
#pragma unroll
        for (int j = 0; j < 3; j++) o.acc_mul(fp[k], limbs[j], o.par(LK_LAM + 4 * (1 + j) + k));
        d[k] = o.sub(o.sub(o.par(LK_ALPHA + k), o.mulc(o.par(LK_LAM + 4 * N_TUPLE + k), M(tag))), o.acc_val(fp[k]));
      }
      ext_mul(h, d, pr);
      o.push(idx0, o.lsub(pr[0], flag));
#pragma unroll
      for (int k = 1; k < 4; k++) o.push(idx0 + k, pr[k]);
    };
    tape(A_HO, I_IO_OUT, oc, r11, 2, F2);
    tape(A_HI, I_IO_IN, ic, y, 3, RL);
  }
  // ---- (mode 3, round 4) loads, stores and the memory check (execute.rs:477-575, memory.rs:86-505): constraints 430.. of the oracle's list ----------------------------
  V hmw[4] = {zero, zero, zero, zero};
  if (mem) {
    V Ev[N_WIN], ob[8], pcs[N_PIECE];
#pragma unroll
    for (int v = 0; v < N_WIN; v++) Ev[v] = o.loc(C_E + v);
#pragma unroll
    for (int j = 0; j < 8; j++) ob[j] = o.loc(C_OB + j);
#pragma unroll
    for (int j = 0; j < N_PIECE; j++) pcs[j] = o.loc(C_PIECE + j);
    const V sgb = o.loc(C_SGB), sgh = o.loc(C_SGH), tbit = o.loc(C_TB), sxv = o.loc(C_SX), cm2 = o.loc(C_CM2), told = o.loc(C_TOLD);
    const V Kmem = o.add(Kld, Kst);
    boolean(I_MEM_BOOL, Kld); boolean(I_MEM_BOOL + 1, Kst);
#pragma unroll
    for (int v = 0; v < N_WIN; v++) boolean(I_MEM_BOOL + 2 + v, Ev[v]);
    boolean(I_MEM_BOOL + 17, sgb); boolean(I_MEM_BOOL + 18, sgh); boolean(I_MEM_BOOL + 19, tbit); boolean(I_MEM_BOOL + 20, cm2);
    AccL esum = o.accl(), wba = o.accl(), wha = o.accl(), offa = o.accl();
#pragma unroll
    for (int v = 0; v < N_WIN; v++) {
      o.acc_lin(esum, Ev[v], 1);
      if (win_start(v)) o.acc_lin(offa, Ev[v], (uint32_t)win_start(v));
      if (win_width(v) == 1) o.acc_lin(wba, Ev[v], 1); else if (win_width(v) == 2) o.acc_lin(wha, Ev[v], 1);
    }
    const V WB = o.accl_val(wba), WH = o.accl_val(wha), WW = o.add(Ev[12], Ev[13]), WD = Ev[14], off = o.accl_val(offa);
    o.push(I_MEM_ONE, o.lsub(o.accl_val(esum), Kmem));                       // exactly one window on a memory row, none elsewhere
    if (wide) o.push(I_MEM_NOHASH, o.lmul(o.lsub(one, H1), H0));             // (mode 4) hash syscalls are a tape (below); syscall 4 — Poseidon2, an error in the reference — never is a row
    else o.push(I_MEM_NOHASH, FH);                                           // no hash syscall in this mode: its memory effect is not stated
    // the opcode names the width (and, for byte / halfword loads, whether the value is sign-extended): LB LBU LH LHU LW LD = 0x30.., SB SH SW SD = 0x38..
    o.push(I_MEM_OP, o.lmul(o.add(o.sub(o.sub(o.sub(o.sub(o.sub(op, o.cst(M(0x30))), o.mulc(WH, M(3))), o.mulc(WW, M(4))), o.mulc(WD, M(5))), WB), o.add(sgb, sgh)), Kld));
    o.push(I_MEM_OP + 1, o.lmul(o.sub(o.sub(o.sub(o.sub(op, o.cst(M(0x38))), WH), o.mulc(WW, M(2))), o.mulc(WD, M(3))), Kst));
    o.push(I_MEM_SG, o.lmul(o.sub(o.sub(o.cst(M(2)), Kld), WB), sgb)); o.push(I_MEM_SG + 1, o.lmul(o.sub(o.sub(o.cst(M(2)), Kld), WH), sgh));
    // what they write: a load rd = field a, a store nothing
    o.push(I_MEM_WR, o.lmul(o.lsub(w1v, fa), Kld)); o.push(I_MEM_WR + 1, o.lmul(w0v, Kst));
    // the address rs1 + sext(imm17) mod 2^64 (rs1 = operand b on loads, operand c on stores) = z, the first range-checked pair; its third limb must vanish
    o.push(I_MEM_EA, o.ladd(o.mul(o.add(o.sub(o.sub(z[0], xb[0]), im0), c0s20), Kld), o.mul(o.add(o.sub(o.sub(z[0], xc[0]), im0), c0s20), Kst)));
    o.push(I_MEM_EA + 1, o.ladd(o.mul(o.add(o.sub(o.sub(o.sub(z[1], xb[1]), im1), c0), c1s20), Kld), o.mul(o.add(o.sub(o.sub(o.sub(z[1], xc[1]), im1), c0), c1s20), Kst)));
    {
      const V hi = o.sub(o.add(o.mulc(s, M(0xFFFFFF)), c1), o.mulc(cm2, M(1u << 24)));
      o.push(I_MEM_EA + 2, o.ladd(o.mul(o.add(xb[2], hi), Kld), o.mul(o.add(xc[2], hi), Kst)));
    }
    // the time read is smaller than the time written (cycle + 1): cycle - told = R4 + 2^10 R5 + 2^20 R6
    o.push(I_MEM_DT, o.lmul(o.lsub(o.sub(o.sub(o.sub(cyc, told), R2[0]), o.mulc(R2[1], M(RC_TABLE))), o.mulc(R2[2], M(RC_TABLE * RC_TABLE))), Kmem));
    // stores: the pieces are the stored register's (rs2 = operand b), limb by limb
    const V lim0 = o.add(o.add(pcs[0], o.mulc(pcs[1], M(1u << 8))), o.mulc(pcs[2], M(1u << 16))), lim1 = o.add(o.add(pcs[3], o.mulc(pcs[4], M(1u << 4))), o.mulc(pcs[5], M(1u << 12))),
            lim2 = o.add(o.add(pcs[6], o.mulc(pcs[7], M(1u << 8))), o.mulc(pcs[8], M(1u << 16)));
    o.push(I_MEM_ST, o.lmul(o.lsub(xb[0], lim0), Kst)); o.push(I_MEM_ST + 1, o.lmul(o.lsub(xb[1], lim1), Kst)); o.push(I_MEM_ST + 2, o.lmul(o.lsub(xb[2], lim2), Kst));
    // y = the window value's limbs, zero-extended from the width, sign-extended when sx (a store writes nothing: its y merely satisfies this)
    {
      AccP a0 = o.accp(), a1 = o.accp();
      o.acc_mul(a0, WB, pcs[0]); o.acc_mul(a0, WH, o.add(pcs[0], o.mulc(pcs[1], M(1u << 8)))); o.acc_mul(a0, o.add(WW, WD), lim0);
      o.acc_mul(a0, sxv, o.sub(o.sub(o.cst(M(1u << 20)), o.mulc(WB, M(1u << 8))), o.mulc(WH, M(1u << 16))));
      o.push(I_MEM_Y, o.lsub(o.mul(y[0], Kmem), o.acc_val(a0)));
      o.acc_mul(a1, WW, o.add(pcs[3], o.mulc(pcs[4], M(1u << 4)))); o.acc_mul(a1, WD, lim1);
      o.push(I_MEM_Y + 1, o.lsub(o.mul(y[1], Kmem), o.add(o.acc_val(a1), o.mulc(sxv, M(0xFFFFF)))));
      o.push(I_MEM_Y + 2, o.lsub(o.mul(y[2], Kmem), o.add(o.mul(WD, lim2), o.mulc(sxv, M(0xFFFFFF)))));
    }
    // sign extension: sx = (sgb + sgh) tb; on LB rows d0 = 128 tb + d6 / 2, on LH rows d1 = 128 tb + d6 / 2 — d6 is in the byte table, so d6 / 2 has seven bits
    o.push(I_MEM_SX, o.lsub(sxv, o.mul(o.add(sgb, sgh), tbit)));
    {
      const V low7 = o.add(o.mulc(tbit, M(128)), o.mulc(pcs[7], M((bb::P + 1) / 2)));
      o.push(I_MEM_SX + 1, o.lmul(o.lsub(pcs[0], low7), sgb)); o.push(I_MEM_SX + 2, o.lmul(o.lsub(pcs[1], low7), sgh));
    }
    // the cell's new bytes as their fingerprint FPN (an aux column): the old bytes with the window replaced by the window value's bytes D_j; a load keeps the cell
    const V D[8] = {pcs[0], pcs[1], o.add(pcs[2], o.mulc(pcs[3], M(16))), pcs[4], pcs[5], pcs[6], pcs[7], pcs[8]};
    V fpn[4], obfp[4], hmr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { fpn[k] = o.aloc(A_FPN + k); hmr[k] = o.aloc(A_HMR + k); hmw[k] = o.aloc(A_HMW + k); }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      AccP a = o.accp(), t = o.accp();
#pragma unroll
      for (int j = 0; j < 8; j++) o.acc_mul(a, ob[j], o.par(LK_LAM + 4 * (3 + j) + k));
      obfp[k] = o.acc_val(a);
#pragma unroll
      for (int v = 0; v < N_WIN; v++) {
        AccP dl = o.accp();
#pragma unroll
        for (int j = 0; j < win_width(v); j++) o.acc_mul(dl, o.sub(D[j], ob[win_start(v) + j]), o.par(LK_LAM + 4 * (3 + win_start(v) + j) + k));
        o.acc_mul(t, Ev[v], o.acc_val(dl));
      }
      o.push(I_MEM_FPN + k, o.lsub(o.sub(fpn[k], obfp[k]), o.acc_val(t)));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) o.push(I_MEM_KEEP + k, o.lmul(o.lsub(fpn[k], obfp[k]), Kld));
    // the memory check: HMR (alpha - fp(cell, told, old bytes)) = Kmem, HMW (alpha - fp(cell, cycle + 1, new bytes)) = Kmem; cell = (z0 - off, z1); fp = a0 + lambda a1 + lambda^2 t + bytes + 7 lambda^11
    {
      const V a0 = o.sub(z[0], off), tnew = o.add(cyc, one);
#pragma unroll
      for (int rw = 0; rw < 2; rw++) {
        V d[4], pr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          AccP a = o.accp();
          o.acc_mul(a, a0, o.par(LK_LAM + k)); o.acc_mul(a, z[1], o.par(LK_LAM + 4 + k)); o.acc_mul(a, rw ? tnew : told, o.par(LK_LAM + 8 + k));
          d[k] = o.sub(o.sub(o.sub(o.par(LK_ALPHA + k), o.mulc(o.par(LK_LAM + 4 * N_TUPLE + k), M(TAG_MEM))), o.acc_val(a)), rw ? fpn[k] : obfp[k]);
        }
        ext_mul(rw ? hmw : hmr, d, pr);
        o.push(I_MEM_RW + 4 * rw, o.lsub(pr[0], Kmem));
#pragma unroll
        for (int k = 1; k < 4; k++) o.push(I_MEM_RW + 4 * rw + k, pr[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) o.acc_lin(hs[k], hmr[k], 1);
    }
    // the nine piece lookups: P_k (alpha - piece_k - tag_k lambda^11) = 1 — on a BITWISE row the slot looks up the nibble tuple (piece_k, b_k, r_k) in the operation's table
    // instead: P_k (alpha - piece_k - lambda b_k - lambda^2 r_k - (tag_k (1 - klg) + 8 oa + 9 oo + 10 ox) lambda^11) = 1
    const V oa = o.loc(C_OA), oo = o.loc(C_OO), lii = o.loc(C_LI), ox = o.sub(o.sub(Klg, oa), oo), nlg = o.sub(one, Klg);
    V lgtag;
    { AccL a = o.accl(); o.acc_lin(a, oa, TAG_AND); o.acc_lin(a, oo, TAG_OR); o.acc_lin(a, ox, TAG_XOR); lgtag = o.accl_val(a); }
    V lb[N_NIB], lr[N_NIB];
#pragma unroll
    for (int k = 0; k < N_NIB; k++) { lb[k] = o.loc(C_LB + k); lr[k] = o.loc(C_LR + k); }
#pragma unroll
    for (int i = 0; i < N_PIECE; i++) {
      V h[4], d[4], pr[4];
      // (.. and on a SHIFT row in the table shift_piece_tag names; piece 8's second element is then the amount, when it comes from a register)
      V tg = piece_tag(i) ? o.add(o.mulc(wide ? o.sub(o.sub(o.sub(nlg, Ksh), Kmu), Kwa) : o.sub(o.sub(nlg, Ksh), Kmu), M((uint32_t)piece_tag(i))), lgtag) : lgtag;   // (a MUL row, a wide-arithmetic row: the 10-bit range table in every slot)
      if (i == 7) tg = o.add(tg, o.mulc(Ksh, M((uint32_t)TAG_NIB)));
      if (i == 8) tg = o.add(tg, o.mulc(o.sub(Ksh, o.loc(C_SI)), M((uint32_t)TAG_LOW6)));
#pragma unroll
      for (int k = 0; k < 4; k++) {
        h[k] = o.aloc(A_P + 4 * i + k); o.acc_lin(hs[k], h[k], 1);
        AccP t = o.accp();
        o.acc_mul(t, lb[i], o.par(LK_LAM + 4 + k)); o.acc_mul(t, lr[i], o.par(LK_LAM + 8 + k)); o.acc_mul(t, tg, o.par(LK_LAM + 4 * N_TUPLE + k));
        d[k] = o.sub(o.par(LK_ALPHA + k), o.acc_val(t));
      }
      d[0] = o.sub(d[0], pcs[i]);
      ext_mul(h, d, pr);
      o.push(I_MEM_PIECE + 4 * i, o.lsub(pr[0], one));
#pragma unroll
      for (int k = 1; k < 4; k++) o.push(I_MEM_PIECE + 4 * i + k, pr[k]);
    }
    // ---- the bitwise opcodes AND OR XOR ANDI ORI XORI = 0x10 + (0 / 1 / 2) + 3 li (execute.rs:199-282), nibble by nibble: constraints 524.. ----
    boolean(I_LG_BOOL, Klg); boolean(I_LG_BOOL + 1, oa); boolean(I_LG_BOOL + 2, oo); boolean(I_LG_BOOL + 3, ox); boolean(I_LG_BOOL + 4, lii);   // (ox boolean: exactly one operation on a bitwise row, none elsewhere)
    o.push(I_LG_OP, o.ladd(o.add(o.sub(o.mulc(lii, M(3)), o.mul(o.lsub(op, o.cst(M(0x10))), Klg)), oo), o.mulc(ox, M(2))));                   // 3 li = klg (op - 0x10) - oo - 2 ox
    o.push(I_LG_LI, o.lmul(nlg, lii));
    o.push(I_LG_WR, o.lmul(o.lsub(w1v, fa), Klg));
    {
      // five nibbles -> a 20-bit limb: v0 + 16 v1 + 256 v2 + 4096 v3 + 65536 v4
      auto limb5 = [&](const V& v0, const V& v1, const V& v2, const V& v3, const V& v4) {
        return o.add(o.add(o.add(v0, o.mulc(v1, M(16))), o.add(o.mulc(v2, M(256)), o.mulc(v3, M(4096)))), o.mulc(v4, M(65536)));
      };
      const V alov = limb5(pcs[0], pcs[1], pcs[2], pcs[3], pcs[4]), ahiv = limb5(pcs[5], pcs[6], pcs[7], pcs[8], R2[3]);        // a_9 = the last range chunk
      const V blov = limb5(lb[0], lb[1], lb[2], lb[3], lb[4]), bhiv = limb5(lb[5], lb[6], lb[7], lb[8], lb[9]);
      const V rlov = limb5(lr[0], lr[1], lr[2], lr[3], lr[4]), rhiv = limb5(lr[5], lr[6], lr[7], lr[8], lr[9]);
      o.push(I_LG_A, o.lmul(o.lsub(xb[0], alov), Klg)); o.push(I_LG_A + 1, o.lmul(o.lsub(xb[1], ahiv), Klg));                          // rs1's 40 bits
      const V kr = o.sub(Klg, lii);
      o.push(I_LG_B, o.ladd(o.mul(o.lsub(xc[0], blov), kr), o.mul(o.lsub(im0, blov), lii)));                                              // rs2's, or the sign-extended immediate's
      o.push(I_LG_B + 1, o.ladd(o.mul(o.lsub(xc[1], bhiv), kr), o.mul(o.lsub(im1, bhiv), lii)));
      o.push(I_LG_Y, o.lmul(o.lsub(y[0], rlov), Klg)); o.push(I_LG_Y + 1, o.lmul(o.lsub(y[1], rhiv), Klg)); o.push(I_LG_Y + 2, o.lmul(y[2], Klg));   // the result: 40 bits
    }
#pragma unroll
    for (int k = 0; k < N_NIB; k++) {                          // no second / third tuple element off the bitwise rows (b_8: nor off the register shifts)
      o.push(I_LG_ZERO + 2 * k, o.lmul(lb[k], k == 8 ? o.add(o.sub(nlg, Ksh), o.loc(C_SI)) : nlg)); o.push(I_LG_ZERO + 2 * k + 1, o.lmul(lr[k], nlg));
    }
    // ---- the shifts SLL SRL SRA SLLI SRLI SRAI = 0x18 + (0 / 1 / 2) + 3 si (execute.rs:284-358): a 2^t = H 2^40 + L, t = 10 u + v: constraints 559.. ----
    {
      V UL[5], UR[5], Vb[10], PR[4];
#pragma unroll
      for (int u = 0; u < 5; u++) { UL[u] = o.loc(C_UL + u); UR[u] = o.loc(C_UR + u); }
#pragma unroll
      for (int v = 0; v < 10; v++) Vb[v] = o.loc(C_V + v);
#pragma unroll
      for (int k = 0; k < 4; k++) PR[k] = o.loc(C_PR + k);
      const V sa = o.loc(C_SA), si = o.loc(C_SI), sb9 = o.loc(C_SB9), sgn = o.loc(C_SGN), shv = o.loc(C_SH), dd = pcs[6], on0 = o.loc(C_ON), on1 = o.loc(C_ON + 1);
      boolean(I_SH_BOOL, Ksh);
#pragma unroll
      for (int u = 0; u < 5; u++) { boolean(I_SH_BOOL + 1 + u, UL[u]); boolean(I_SH_BOOL + 6 + u, UR[u]); }
#pragma unroll
      for (int v = 0; v < 10; v++) boolean(I_SH_BOOL + 11 + v, Vb[v]);
      boolean(I_SH_BOOL + 21, sa); boolean(I_SH_BOOL + 22, si); boolean(I_SH_BOOL + 23, sb9);
      AccL sula = o.accl(), sura = o.accl(), sva = o.accl(), tta = o.accl(), pva = o.accl(), vsa = o.accl();
#pragma unroll
      for (int u = 0; u < 5; u++) { o.acc_lin(sula, UL[u], 1); o.acc_lin(sura, UR[u], 1); if (u) { o.acc_lin(tta, UL[u], 10u * u); o.acc_lin(tta, UR[u], 10u * u); } }
#pragma unroll
      for (int v = 0; v < 10; v++) { o.acc_lin(sva, Vb[v], 1); o.acc_lin(pva, Vb[v], 1u << v); if (v) { o.acc_lin(tta, Vb[v], (uint32_t)v); o.acc_lin(vsa, Vb[v], (uint32_t)v); } }
      const V sUL = o.accl_val(sula), sUR = o.accl_val(sura), sV = o.accl_val(sva), tt = o.accl_val(tta), PV = o.accl_val(pva), vsum = o.accl_val(vsa);
      o.push(I_SH_ONE, o.lsub(o.add(sUL, sUR), Ksh)); o.push(I_SH_ONE + 1, o.lsub(sV, Ksh));      // one chunk shift (left or right) and one bit shift on a shift row, none elsewhere
      o.push(I_SH_OP, o.lsub(o.sub(o.sub(o.mul(o.lsub(op, o.cst(M(0x18))), Ksh), sUR), sa), o.mulc(si, M(3))));   // the opcode: 0x18 + [right] + [arithmetic] + 3 [immediate]
      o.push(I_SH_OP + 1, o.lmul(o.lsub(one, sUR), sa)); o.push(I_SH_OP + 2, o.lmul(o.lsub(one, Ksh), si));
      o.push(I_SH_WR, o.lmul(o.lsub(w1v, fa), Ksh));
      o.push(I_SH_A, o.lmul(o.lsub(o.sub(xb[0], R2[0]), o.mulc(R2[1], M(RC_TABLE))), Ksh)); o.push(I_SH_A + 1, o.lmul(o.lsub(o.sub(xb[1], R2[2]), o.mulc(R2[3], M(RC_TABLE))), Ksh));   // a's four chunks
#pragma unroll
      for (int k = 0; k < 4; k++) o.push(I_SH_PR + k, o.lsub(PR[k], o.mul(R2[k], PV)));           // c_k 2^v ..
#pragma unroll
      for (int k = 0; k < 4; k++) o.push(I_SH_LOHI + k, o.lmul(o.lsub(o.sub(PR[k], R[k]), o.mulc(pcs[k], M(RC_TABLE))), Ksh));   // .. = lo_k + 2^10 hi_k
      o.push(I_SH_SIGN, o.lmul(o.lsub(o.sub(R2[3], o.mulc(sb9, M(512))), o.mulc(pcs[4], M((bb::P + 1) / 2))), Ksh));   // bit 39 of a: c_3 = 512 sb9 + piece_4 / 2
      o.push(I_SH_SIGN + 1, o.lsub(sgn, o.mul(sa, sb9)));
      // the amount: rs2's low six bits (its first chunk piece_8 with sh in LOW6, the rest of the limb in piece_5), or the word's shamt fc + 16 (fhi mod 16)
      const V kreg = o.sub(Ksh, si);
      o.push(I_SH_AMT, o.lmul(o.lsub(o.sub(xc[0], pcs[8]), o.mulc(pcs[5], M(RC_TABLE))), kreg));
      o.push(I_SH_AMT + 1, o.lmul(o.lsub(lb[8], shv), kreg));
      o.push(I_SH_AMT + 2, o.lmul(o.lsub(o.sub(fhi, pcs[7]), o.mulc(pcs[5], M(16))), si));
      o.push(I_SH_AMT + 3, o.lmul(o.lsub(o.sub(shv, fc), o.mulc(pcs[7], M(16))), si));
      o.push(I_SH_AMT + 4, o.lmul(shv, o.sub(one, Ksh)));
      // sh = t + d on a left shift, 40 - t + d on a right shift; d (>= 0: a range lookup) only where t cannot say more: t = 40 resp. t = 0
      o.push(I_SH_T, o.ladd(o.mul(o.lsub(o.sub(shv, tt), dd), sUL), o.mul(o.lsub(o.add(o.sub(shv, o.cst(M(40))), tt), dd), sUR)));
      o.push(I_SH_D, o.lmul(dd, vsum)); o.push(I_SH_D + 1, o.lmul(o.lsub(sUL, UL[4]), dd)); o.push(I_SH_D + 2, o.lmul(o.lsub(sUR, UR[0]), dd));
      o.push(I_SH_T40, o.lmul(o.lsub(one, Vb[0]), UR[4]));                                       // a right shift keeps at most 40 bits: t <= 40
      // 2^40 - 2^t in limbs (right shifts): t < 20: (2^20 - 2^t, 2^20 - 1); 20 <= t < 40: (0, 2^20 - 2^(t-20)); t = 40: (0, 0)
      {
        const V low = o.add(UR[0], UR[1]);
        o.push(I_SH_ON, o.lsub(on0, o.sub(o.mulc(low, M(1u << 20)), o.mul(o.add(UR[0], o.mulc(UR[1], M(1u << 10))), PV))));
        o.push(I_SH_ON + 1, o.lsub(on1, o.sub(o.sub(o.sub(o.mulc(sUR, M(1u << 20)), low), o.mul(o.add(UR[2], o.mulc(UR[3], M(1u << 10))), PV)), o.mulc(UR[4], M(1u << 20)))));
      }
      // the chunks of a 2^v: m_0 = lo_0, m_i = lo_i + hi_(i-1), m_4 = hi_3; result chunk j = m_(j-u) on a left shift, m_(j+4-u) on a right shift
      const V m[5] = {R[0], o.add(R[1], pcs[0]), o.add(R[2], pcs[1]), o.add(R[3], pcs[2]), pcs[3]};
      V res[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        AccP a = o.accp();
#pragma unroll
        for (int u = 0; u < 5; u++) { if (j - u >= 0) o.acc_mul(a, UL[u], m[j - u]); if (j + 4 - u >= 0 && j + 4 - u <= 4) o.acc_mul(a, UR[u], m[j + 4 - u]); }
        res[j] = o.acc_val(a);
      }
      o.push(I_SH_Y, o.lsub(o.sub(o.mul(y[0], Ksh), o.add(res[0], o.mulc(res[1], M(RC_TABLE)))), o.mul(sgn, on0)));
      o.push(I_SH_Y + 1, o.lsub(o.sub(o.mul(y[1], Ksh), o.add(res[2], o.mulc(res[3], M(RC_TABLE)))), o.mul(sgn, on1)));
      o.push(I_SH_Y + 2, o.lmul(y[2], Ksh));
    }
    // ---- MUL (execute.rs:79-99): the product of the 40-bit operands mod 2^40, schoolbook in 10-bit chunks: constraints 616.. ----
    {
      V ma[4], me[3];
#pragma unroll
      for (int k = 0; k < 4; k++) ma[k] = o.loc(C_MA + k);
#pragma unroll
      for (int k = 0; k < 3; k++) me[k] = o.loc(C_ME + k);
      boolean(I_MU_BOOL, Kmu); boolean(I_MU_BOOL + 1, me[0]); boolean(I_MU_BOOL + 2, me[1]); boolean(I_MU_BOOL + 3, me[2]);
      o.push(I_MU_WR, o.lmul(o.lsub(w1v, fa), Kmu));
      o.push(I_MU_A, o.lmul(o.lsub(o.sub(xb[0], R2[0]), o.mulc(R2[1], M(RC_TABLE))), Kmu)); o.push(I_MU_A + 1, o.lmul(o.lsub(o.sub(xb[1], R2[2]), o.mulc(R2[3], M(RC_TABLE))), Kmu));   // a's four chunks
      o.push(I_MU_B, o.lmul(o.lsub(o.sub(xc[0], pcs[0]), o.mulc(pcs[1], M(RC_TABLE))), Kmu)); o.push(I_MU_B + 1, o.lmul(o.lsub(o.sub(xc[1], pcs[2]), o.mulc(pcs[3], M(RC_TABLE))), Kmu));   // b's
#pragma unroll
      for (int k = 0; k < 4; k++) o.push(I_MU_MA + k, o.lsub(ma[k], o.mul(R2[k], Kmu)));            // ma_k = kmu a_k
      const V carry[4] = {pcs[4], o.add(pcs[5], o.mulc(me[0], M(RC_TABLE))), o.add(pcs[6], o.mulc(o.add(me[1], o.add(me[2], me[2])), M(RC_TABLE))), o.add(pcs[7], o.mulc(pcs[8], M(RC_TABLE)))};
#pragma unroll
      for (int k = 0; k < 4; k++) {                                                                // sum_{i+j=k} a_i b_j + carry_(k-1) = r_k + 2^10 carry_k
        AccP t = o.accp();
#pragma unroll
        for (int j = 0; j <= k; j++) o.acc_mul(t, ma[j], pcs[k - j]);
        V lin = o.add(R[k], o.mulc(carry[k], M(RC_TABLE)));
        if (k) lin = o.sub(lin, carry[k - 1]);
        o.push(I_MU_EQ + k, o.lsub(o.acc_val(t), o.mul(lin, Kmu)));
      }
      o.push(I_MU_Y, o.lmul(o.lsub(y[0], z[0]), Kmu)); o.push(I_MU_Y + 1, o.lmul(o.lsub(y[1], z[1]), Kmu)); o.push(I_MU_Y + 2, o.lmul(y[2], Kmu));
    }
    // ---- (mode 4, round 6) MULH DIVU REMU DIV REM on operands below 2^40 (execute.rs:101-183): F1 F2 + ADD = LO + 2^40 HI, schoolbook in 10-bit chunks: constraints 636.. ----
    if (wide) {
      V gf[4], we[N_WE], X[N_X];
#pragma unroll
      for (int k = 0; k < 4; k++) gf[k] = o.loc(C_GF + k);
#pragma unroll
      for (int k = 0; k < N_WE; k++) we[k] = o.loc(C_WE + k);
#pragma unroll
      for (int k = 0; k < N_X; k++) X[k] = o.loc(C_X + k);
      const V om = o.loc(C_OM), od = o.loc(C_OD), orr = o.loc(C_ORR), kd = o.add(od, orr);
      boolean(I_WA_BOOL, Kwa); boolean(I_WA_BOOL + 1, om); boolean(I_WA_BOOL + 2, od); boolean(I_WA_BOOL + 3, orr);     // (kwa boolean: at most one of the three kinds)
#pragma unroll
      for (int k = 0; k < N_WE; k++) boolean(I_WA_BOOL + 4 + k, we[k]);
      { AccL a = o.accl(); o.acc_lin(a, om, 3); o.acc_lin(a, od, 4); o.acc_lin(a, orr, 5); o.push(I_WA_OP, o.lsub(o.mul(o.lsub(op, o.add(g, g)), Kin), o.accl_val(a))); }   // MULH 3, DIVU 4, REMU 5, DIV 6 = 4 + 2 g, REM 7 = 5 + 2 g
      o.push(I_WA_WR, o.lmul(o.lsub(w1v, fa), Kwa));                                               // rd = field a
      o.push(I_WA_TOP, o.lmul(xb[2], Kin)); o.push(I_WA_TOP + 1, o.lmul(xc[2], Kin));              // the chunk relation's operands are below 2^40 (wider ones: the tape, I_WW)
      constexpr uint32_t T10 = M(RC_TABLE);
      o.push(I_WA_F2, o.lmul(o.lsub(o.sub(xc[0], pcs[0]), o.mulc(pcs[1], T10)), Kin)); o.push(I_WA_F2 + 1, o.lmul(o.lsub(o.sub(xc[1], pcs[2]), o.mulc(pcs[3], T10)), Kin));   // F2 = rs2, always
      o.push(I_WA_F1, o.lmul(o.lsub(o.sub(xb[0], R2[0]), o.mulc(R2[1], T10)), om)); o.push(I_WA_F1 + 1, o.lmul(o.lsub(o.sub(xb[1], R2[2]), o.mulc(R2[3], T10)), om));         // MULH: F1 = rs1
      o.push(I_WA_LO, o.lmul(o.lsub(o.sub(xb[0], R[0]), o.mulc(R[1], T10)), kd)); o.push(I_WA_LO + 1, o.lmul(o.lsub(o.sub(xb[1], R[2]), o.mulc(R[3], T10)), kd));             // divisions: LO = rs1
#pragma unroll
      for (int k = 0; k < 4; k++) o.push(I_WA_GF + k, o.lsub(gf[k], o.mul(R2[k], Kin)));           // gf_k = (om + od + orr) F1_k
      const V G4[4] = {pcs[7], pcs[8], X[0], X[1]};
      auto two = [&](const V& a, const V& b) { return o.add(a, o.add(b, b)); };                    // a + 2 b
      const V cr[6] = {pcs[4], o.add(pcs[5], o.mulc(we[0], T10)), o.add(pcs[6], o.mulc(two(we[1], we[2]), T10)),
                       o.add(X[2], o.mulc(two(we[3], we[4]), T10)), o.add(X[3], o.mulc(two(we[5], we[6]), T10)), o.add(X[4], o.mulc(two(we[7], we[8]), T10))};
      air_static_for<0, 7>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        AccP t = o.accp();
#pragma unroll
        for (int j = 0; j < 4; j++) if (k - j >= 0 && k - j < 4) o.acc_mul(t, gf[j], pcs[k - j]);
        if constexpr (k < 4) {                                                                      // low half: + ADD_k + c_(k-1) = LO_k + 2^10 c_k; the carry OUT of position 3 exists on MULH rows only
          o.acc_mul(t, kd, G4[k]);
          V lin = R[k];
          if constexpr (k < 3) lin = o.add(lin, o.mulc(cr[k], T10));
          if constexpr (k > 0) lin = o.sub(lin, cr[k - 1]);
          if constexpr (k == 3) o.push(I_WA_EQ + k, o.lsub(o.sub(o.acc_val(t), o.mul(lin, Kin)), o.mul(o.mulc(cr[3], T10), om)));
          else o.push(I_WA_EQ + k, o.lsub(o.acc_val(t), o.mul(lin, Kin)));
        } else {                                                                                    // high half (MULH): + c_(k-1) = HI_(k-4) + 2^10 c_k (k = 6: 2^10 HI_3); a division has nothing there
          const V hi = k < 6 ? o.add(G4[k - 4], o.mulc(cr[k < 6 ? k : 5], T10)) : o.add(G4[2], o.mulc(G4[3], T10));
          o.push(I_WA_EQ + k, o.ladd(o.acc_val(t), o.mul(o.lsub(cr[k - 1], hi), om)));
        }
      });
      // divisions: the remainder is smaller than the divisor: d = rs2 - r - 1 >= 0 in chunks X2..X5, borrow e4 between the limbs (which also says rs2 != 0)
      const V r0 = o.add(G4[0], o.mulc(G4[1], T10)), r1 = o.add(G4[2], o.mulc(G4[3], T10)), dd0 = o.add(X[2], o.mulc(X[3], T10)), dd1 = o.add(X[4], o.mulc(X[5], T10));
      o.push(I_WA_LT, o.lmul(o.ladd(o.sub(o.sub(o.sub(xc[0], r0), one), dd0), o.mulc(we[3], M(1u << 20))), kd));
      o.push(I_WA_LT + 1, o.lmul(o.lsub(o.sub(o.sub(xc[1], r1), we[3]), dd1), kd));
      // what is written: HI (MULH) and the remainder (REMU / REM) sit in G4, the quotient (DIVU / DIV) in F1
      const V gk = o.add(om, orr);
      o.push(I_WA_Y, o.lmul(o.lsub(y[0], r0), gk)); o.push(I_WA_Y + 1, o.lmul(o.lsub(y[1], r1), gk));
      o.push(I_WA_Y + 2, o.lmul(o.lsub(o.sub(y[0], R2[0]), o.mulc(R2[1], T10)), od)); o.push(I_WA_Y + 3, o.lmul(o.lsub(o.sub(y[1], R2[2]), o.mulc(R2[3], T10)), od));
      o.push(I_WA_Y + 4, o.lmul(y[2], Kin));
      // the six extra range slots: XH_i (alpha - X_i) = 1, on every row
#pragma unroll
      for (int i = 0; i < N_X; i++) {
        V h[4], d[4], pr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { h[k] = o.aloc(A_X + 4 * i + k); o.acc_lin(hs[k], h[k], 1); d[k] = o.par(LK_ALPHA + k); }
        d[0] = o.sub(d[0], X[i]);
        ext_mul(h, d, pr);
        o.push(I_WA_X + 4 * i, o.lsub(pr[0], one));
#pragma unroll
        for (int k = 1; k < 4; k++) o.push(I_WA_X + 4 * i + k, pr[k]);
      }
      // the boundary cell: no store writes the low half of cell B — nb = delta iws, delta = the row's cell address minus B as ONE field element; tl (kst - nb) = 0
      {
        AccL tla = o.accl();
#pragma unroll
        for (int v = 0; v < N_WIN; v++) if (is_low_window(v)) o.acc_lin(tla, Ev[v], 1);
        const V nb = o.loc(C_NB), iws = o.loc(C_IWS);
        const V delta = o.add(o.sub(o.sub(z[0], off), o.par(LK_B0)), o.mulc(o.sub(z[1], o.par(LK_B1)), M(1u << 20)));
        o.push(I_BC, o.lsub(nb, o.mul(delta, iws)));
        o.push(I_BC + 1, o.lmul(o.lsub(Kst, nb), o.accl_val(tla)));
      }
      // hash syscalls: HH (alpha - fp(cycle, R11's limbs, R12's, R13's, 3 fh + h0 + 2 h1) - 12 lambda^11) = fh: the call's record is in the tape the proof carries
      {
        V h[4], d[4], pr[4], hl[9];
#pragma unroll
        for (int j = 0; j < 9; j++) hl[j] = o.loc(C_LIMB + 33 + j);
        AccL ka = o.accl();
        o.acc_lin(ka, FH, 3); o.acc_lin(ka, H0, 1); o.acc_lin(ka, H1, 2);
        const V kind = o.accl_val(ka), cycv = o.loc(C_CYCLE);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          h[k] = o.aloc(A_HH + k); o.acc_lin(hs[k], h[k], 1);
          AccP t = o.accp();
          o.acc_mul(t, cycv, o.par(LK_LAM + k));
#pragma unroll
          for (int j = 0; j < 9; j++) o.acc_mul(t, hl[j], o.par(LK_LAM + 4 * (1 + j) + k));
          o.acc_mul(t, kind, o.par(LK_LAM + 4 * 10 + k));
          d[k] = o.sub(o.sub(o.par(LK_ALPHA + k), o.mulc(o.par(LK_LAM + 4 * N_TUPLE + k), M((uint32_t)TAG_HASH))), o.acc_val(t));
        }
        ext_mul(h, d, pr);
        o.push(I_HH, o.lsub(pr[0], FH));
#pragma unroll
        for (int k = 1; k < 4; k++) o.push(I_HH + k, pr[k]);
      }
      // the wide tape: ot boolean; WW (alpha - fp(cycle, rs1's limbs, rs2's, y's, opcode) - 13 lambda^11) = ot: the row writes what the verifier computed from its record
      {
        const V ot = o.loc(C_OT), cycv = o.loc(C_CYCLE);
        boolean(I_WT_BOOL, ot);
        V h[4], d[4], pr[4];
        const V tup[10] = {cycv, xb[0], xb[1], xb[2], xc[0], xc[1], xc[2], y[0], y[1], y[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          h[k] = o.aloc(A_WW + k); o.acc_lin(hs[k], h[k], 1);
          AccP t = o.accp();
#pragma unroll
          for (int j = 0; j < 10; j++) o.acc_mul(t, tup[j], o.par(LK_LAM + 4 * j + k));
          o.acc_mul(t, op, o.par(LK_LAM + 4 * 10 + k));
          d[k] = o.sub(o.sub(o.par(LK_ALPHA + k), o.mulc(o.par(LK_LAM + 4 * N_TUPLE + k), M((uint32_t)TAG_WIDE))), o.acc_val(t));
        }
        ext_mul(h, d, pr);
        o.push(I_WW, o.lsub(pr[0], ot));
#pragma unroll
        for (int k = 1; k < 4; k++) o.push(I_WW + k, pr[k]);
      }
    }
  }
  // running sum over the cycle of all N rows (no selector): S(w x) - S(x) = H0 + .. + H7 + HR (+ HO + HI) - T / N
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (mem) o.push(I_SUM + k, o.ladd(o.add(o.sub(o.sub(nS[k], S[k]), o.accl_val(hs[k])), hmw[k]), o.par(LK_TN + k)));   // .. + P0 + .. + P8 + HMR - HMW: the tuple written is PROVIDED, not looked up
    else o.push(I_SUM + k, o.ladd(o.sub(o.sub(nS[k], S[k]), o.accl_val(hs[k])), o.par(LK_TN + k)));
  }
}

// - sum_i (alpha^first_idx(i) first_m[i]) and the same for the last row: the share of the public boundary words in the two boundary sums — per-proof
// constants (E4, Montgomery; alpha_pow in Montgomery form) the prover's host side hands the quotient kernel (push_fc / push_lc leave the words out)
inline void boundary_constants(const bb::E4* alpha_pow_m, const uint32_t* first_m, const uint32_t* last_m, bb::E4& cf, bb::E4& cl, const uint32_t* cnt_m = nullptr) {
  cf = bb::e_zero(); cl = bb::e_zero();
  for (int i = 0; i < N_STATE; i++) {
    cf = bb::e_add(cf, bb::e_mul_fm(alpha_pow_m[first_idx(i)], first_m[i]));
    cl = bb::e_add(cl, bb::e_mul_fm(alpha_pow_m[last_idx(i)], last_m[i]));
  }
  cl = bb::e_add(cl, bb::e_mul_fm(alpha_pow_m[I_HALT], bb::R1));
  if (cnt_m)                                                   // (mode 2) the counters of the first / last row
    for (int k = 0; k < 2; k++) { cf = bb::e_add(cf, bb::e_mul_fm(alpha_pow_m[I_IO_FIRST + k], cnt_m[k])); cl = bb::e_add(cl, bb::e_mul_fm(alpha_pow_m[I_IO_LAST + k], cnt_m[2 + k])); }
}

// The ORDER in which air::eval pushes the constraints a quotient kernel consumes (push_fc0 / push_lc0 are not consumed): the kernel reads its
// coefficients alpha^c from a table laid out in this order, one after the other, each requested while the previous constraint is being accumulated
// (QuotientOps: a scalar load issued at its point of use costs a scalar-cache round trip per constraint, with nothing to cover it).
struct OrderOps {
  struct V {};
  using AccP = V; using AccL = V;
  bool deferred;
  int order[N_CONSTRAINTS]; int n = 0;
  V loc(int) { return V{}; } V nxt(int) { return V{}; } V loc_r(int) { return V{}; } V nxt_r(int) { return V{}; } V aloc(int) { return V{}; } V anxt(int) { return V{}; }
  V par(int) { return V{}; } V cst(uint32_t) { return V{}; } V add(V, V) { return V{}; } V sub(V, V) { return V{}; } V mul(V, V) { return V{}; } V mulc(V, uint32_t) { return V{}; }
  V lsub(V, V) { return V{}; } V ladd(V, V) { return V{}; } V lmul(V, V) { return V{}; }
  AccP accp() { return V{}; } void acc_mul(AccP&, V, V) {} V acc_val(const AccP&) { return V{}; }
  AccL accl() { return V{}; } void acc_lin(AccL&, V, uint32_t) {} V accl_val(const AccL&) { return V{}; }
  void end_boundary() {} void end_trans() {}
  void rec(int idx) { if (n < N_CONSTRAINTS) order[n] = idx; n++; }
  void push(int idx, V) { rec(idx); } void push_t(int idx, V) { rec(idx); } void push_fc(int idx, V, uint32_t) { rec(idx); } void push_lc(int idx, V, uint32_t) { rec(idx); }
  void push_fc0(int, uint32_t) {} void push_lc0(int, uint32_t) {}
};
// order[k] = the constraint index of the k-th consumed push; returns their number (<= N_CONSTRAINTS)
inline int push_order(int mode, int* order) {
  OrderOps o{mode == 1};
  uint32_t st[N_STATE] = {}, cnt[4] = {};
  eval(o, st, st, mode, cnt);
  for (int k = 0; k < o.n && k < N_CONSTRAINTS; k++) order[k] = o.order[k];
  return o.n;
}
#if !defined(__HIP_DEVICE_COMPILE__)
// air::eval run on BOUNDS instead of values: every V carries the largest word it can hold under the quotient kernel's arithmetic (QuotientOps,
// stark_prove.inl), every operation checks the precondition that arithmetic needs (no 32 / 64 / 96-bit overflow, lazy values only where a
// reduction follows) and every constraint index is pushed at most once.  zkir_air_check_bounds (verify.cpp) runs it; tests/test_abi.py asserts it.
struct BoundOps {
  struct V { uint64_t max; };
  struct AccP { unsigned __int128 max; };
  struct AccL { uint64_t max; };
  int deferred;                                                // the MODE (0, 1, 2)
  const char* why = nullptr;
  unsigned __int128 tot[4] = {0, 0, 0, 0};
  int seen[N_CONSTRAINTS] = {};
  static constexpr uint64_t PM = bb::P - 1, U32 = 0xFFFFFFFFull;
  void need(bool c, const char* w) { if (!c && !why) why = w; }
  V red() const { return V{PM}; }
  V loc(int k) { need(k >= 0 && k < W, "loc: column out of range"); return is_virtual(k, deferred) ? V{0} : red(); }
  V nxt(int k) { return loc(k); }
  V loc_r(int k) { need(k >= 0 && k < W && !is_virtual(k, deferred), "loc_r: uncommitted column"); return red(); }
  V nxt_r(int k) { return loc_r(k); }
  V aloc(int k) { need(k >= 0 && k < aux_width(deferred), "aloc: column out of range"); return red(); }
  V anxt(int k) { return aloc(k); }
  V par(int i) { need(i >= 0 && i < N_LK, "par: index out of range"); return red(); }
  V cst(uint32_t cm) { need(cm < bb::P, "cst: constant not reduced"); return V{cm}; }
  V add(V a, V b) { need(a.max <= PM && b.max <= PM, "add: lazy operand"); return red(); }
  V sub(V a, V b) { need(a.max <= PM && b.max <= PM, "sub: lazy operand"); return red(); }
  V mul(V a, V b) { need(a.max <= U32, "mul: first operand above 32 bits"); need(b.max <= PM, "mul: second operand lazy"); return red(); }
  V mulc(V a, uint32_t cm) { need(cm < bb::P, "mulc: constant not reduced"); return mul(a, V{cm}); }
  V lsub(V a, V b) { need(a.max <= PM && b.max <= PM, "lsub: lazy operand"); return V{a.max + bb::P}; }
  V ladd(V a, V b) { need(a.max <= PM && b.max <= PM, "ladd: lazy operand"); return V{a.max + b.max}; }
  V lmul(V a, V b) { need(a.max <= U32, "lmul: first operand above 32 bits"); need(b.max <= PM, "lmul: second operand lazy"); return V{((a.max * b.max) >> 32) + bb::P}; }
  AccP accp() { return AccP{0}; }
  void acc_mul(AccP& a, V x, V y) { need(x.max <= U32 && y.max <= U32, "acc_mul: operand above 32 bits"); a.max += (unsigned __int128)x.max * y.max; need(a.max < ((unsigned __int128)1 << 73), "acc_mul: sum above 2^73"); }
  V acc_val(const AccP&) { return red(); }
  AccL accl() { return AccL{0}; }
  void acc_lin(AccL& a, V x, uint32_t k) { need(x.max <= PM, "acc_lin: lazy operand"); a.max += (uint64_t)k * x.max; }
  V accl_val(const AccL& a) { const uint64_t t = (a.max >> 32) * bb::R1 + U32; need(t < (1ull << 38) && t < 200ull * bb::P, "accl_val: sum outside reduce_wide<6>"); return red(); }
  void end_boundary() {}
  void end_trans() {}
  void count(int g, int idx, V v) {
    need(idx >= 0 && idx < N_CONSTRAINTS, "push: constraint index out of range");
    if (idx >= 0 && idx < N_CONSTRAINTS) need(seen[idx]++ == 0, "push: constraint index used twice");
    need(v.max <= U32, "push: value above 32 bits");
    tot[g] += (unsigned __int128)PM * v.max; need(tot[g] < ((unsigned __int128)1 << 73), "push: sum above 2^73");
  }
  void push(int idx, V v) { count(0, idx, v); }
  void push_t(int idx, V v) { count(1, idx, v); }
  void push_fc(int idx, V v, uint32_t) { count(2, idx, v); }
  void push_lc(int idx, V v, uint32_t) { count(3, idx, v); }
  void push_fc0(int idx, uint32_t) { count(2, idx, V{0}); }
  void push_lc0(int idx, uint32_t) { count(3, idx, V{0}); }
};
// nullptr = the quotient kernel's arithmetic is sound on air::eval; else the first broken rule
inline const char* check_bounds(int mode) {
  BoundOps o{mode};
  uint32_t st[N_STATE] = {}, cnt[4] = {};
  eval(o, st, st, mode, cnt);
  return o.why;
}
#endif

}  // namespace air
