// ntt.hip — Baby Bear NTT passes and the coset low-degree extension on gfx950 (part of the self-defined prover stages, DESIGN.md §8;
// oracle: so::lde / so::ntt in oracle/stark_oracle.cpp).  Split from stark.hip because the two want different instruction
// scheduling: these kernels are LDS/register-pressure sensitive (default scheduler), the hash kernels want maximum ILP.
//
// Matrix layout ("B8", include/zkir_amd.h): columns are grouped in blocks of 8; block b of a matrix with n rows is the array
// [n][8] of u32, i.e. every row position holds 32 contiguous bytes = two uint4 (columns 8b..8b+3 and 8b+4..8b+7).  Every kernel here
// moves and computes on uint4: one lane carries FOUR columns of one position, so the index arithmetic and — what matters, the
// kernels being VALU-bound — the twiddle bookkeeping (3 of the 7 Montgomery products per radix-4 quad were twiddle derivations
// when a lane carried one column) are shared by four (strided passes: eight) columns.
//
// Per block: inverse DIF NTT over H (natural -> bit-reversed), coset scale g^k / N, zero-interleave, forward DIT NTT over the
// 2N coset (bit-reversed -> natural).  LDS-staged passes: strided passes move tiles of 2^B rows x 2^C positions (2^C * 32 bytes
// consecutive per tile row keep loads coalesced); the last 10 inverse stages, the scaling and the first 11 forward stages are
// fused in one contiguous-chunk kernel.  HBM traffic per element and column: 8 B per strided pass, 12 B for the fused middle.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "../../include/zkir_amd.h"
#include "../../include/zkir_amd_experimental.h"
#include "babybear.h"
#include "host.h"

namespace {

constexpr int NT = 256;
__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return bits == 0 ? 0u : __brev(x) >> (32 - bits); }

// compile-time loop: the body sees its index as a constant, so small per-lane arrays stay in registers
template <int K, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (K < N) { f(std::integral_constant<int, K>{}); static_for<K + 1, N>(f); }
}

using u32x4 = uint32_t __attribute__((ext_vector_type(4)));          // plain vector for values that only travel (prefetch registers)
__device__ __forceinline__ u32x4 ld4(const uint4* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st4(uint4* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// LDS hand-off INSIDE a wave: the next round reads only what this wave itself wrote (LDS operations of one wave execute in issue
// order; the wait makes the writes complete, the clobber keeps the compiler from moving LDS accesses across).  Where it applies it
// replaces a workgroup barrier: the waves of a workgroup then drift apart, and one wave's LDS latency is covered by another's arithmetic
// instead of all of them waiting at the same instruction.
__device__ __forceinline__ void wave_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---- four columns at a time --------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 add4(uint4 a, uint4 b) { return make_uint4(bb::add(a.x, b.x), bb::add(a.y, b.y), bb::add(a.z, b.z), bb::add(a.w, b.w)); }
__device__ __forceinline__ uint4 sub4(uint4 a, uint4 b) { return make_uint4(bb::sub(a.x, b.x), bb::sub(a.y, b.y), bb::sub(a.z, b.z), bb::sub(a.w, b.w)); }
__device__ __forceinline__ uint4 addl4(uint4 a, uint4 b) { return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }                   // < 2p: only ever the lazy operand of a product
__device__ __forceinline__ uint4 subl4(uint4 a, uint4 b) { return make_uint4(a.x - b.x + bb::P, a.y - b.y + bb::P, a.z - b.z + bb::P, a.w - b.w + bb::P); }
__device__ __forceinline__ uint4 mul4(uint4 a, uint32_t w) { return make_uint4(bb::mont_mul(a.x, w), bb::mont_mul(a.y, w), bb::mont_mul(a.z, w), bb::mont_mul(a.w, w)); }

// inverse (DIF) radix-4 quad: two stages on x0..x3 (spans 2d, d) with twiddles wA (first stage, even pair), wB = wA * j (odd pair),
// w2 = wA^2 (second stage); differences only feed a product: no reduction
__device__ __forceinline__ void dif4(uint4& x0, uint4& x1, uint4& x2, uint4& x3, uint32_t wA, uint32_t wB, uint32_t w2) {
  const uint4 y0 = add4(x0, x2), y2 = mul4(subl4(x0, x2), wA);
  const uint4 y1 = add4(x1, x3), y3 = mul4(subl4(x1, x3), wB);
  x0 = add4(y0, y1); x1 = mul4(subl4(y0, y1), w2);
  x2 = add4(y2, y3); x3 = mul4(subl4(y2, y3), w2);
}
// forward (DIT) radix-4 quad: stages with twiddles w1 = w2^2 (first stage), w2 / w2i = w2 * j (second stage); outputs in positions
// (0, 2, 1, 3) order of the classic DIT flow: o0 -> i0, o1 -> i0 + d, o2 -> i0 + 2d, o3 -> i0 + 3d
__device__ __forceinline__ void dit4(uint4& x0, uint4& x1, uint4& x2, uint4& x3, uint32_t w1, uint32_t w2, uint32_t w2i) {
  const uint4 t1 = mul4(x1, w1), t3 = mul4(x3, w1);
  const uint4 y0 = add4(x0, t1), y1 = sub4(x0, t1), y2 = addl4(x2, t3), y3 = subl4(x2, t3);
  const uint4 u2 = mul4(y2, w2), u3 = mul4(y3, w2i);
  x0 = add4(y0, u2); x2 = sub4(y0, u2); x1 = add4(y1, u3); x3 = sub4(y1, u3);
}

// ---- one radix-2 stage straight through global memory (odd stage counts only; one lane = one butterfly of four columns) -----
template <bool DIT>
__global__ __launch_bounds__(NT) void ntt_stage_kernel(uint4* __restrict__ data, uint64_t blk_u4, int L, int s0, const uint32_t* __restrict__ tw) {
  uint4* x = data + (uint64_t)blockIdx.y * blk_u4;
  const uint64_t item = (uint64_t)blockIdx.x * NT + threadIdx.x;              // (pair t, half h)
  const uint32_t n = 1u << L;
  if (item >= (uint64_t)n) return;                                             // n/2 pairs x 2 halves
  const uint32_t t = (uint32_t)(item >> 1), h = (uint32_t)(item & 1);
  uint32_t p, stride, w;
  if (DIT) { stride = 1u << s0; const uint32_t hi = t >> s0, lo = t & (stride - 1); p = (hi << (s0 + 1)) + lo; w = tw[lo << (L - s0 - 1)]; }
  else { stride = n >> (s0 + 1); const uint32_t hi = t / stride, lo = t % stride; p = hi * (n >> s0) + lo; w = tw[lo << s0]; }
  const uint4 a = x[(uint64_t)p * 2 + h], b = x[(uint64_t)(p + stride) * 2 + h];
  if (DIT) { const uint4 tt = mul4(b, w); x[(uint64_t)p * 2 + h] = add4(a, tt); x[(uint64_t)(p + stride) * 2 + h] = sub4(a, tt); }
  else { x[(uint64_t)p * 2 + h] = add4(a, b); x[(uint64_t)(p + stride) * 2 + h] = mul4(subl4(a, b), w); }
}

// ---- EXPERIMENT (round 4, profiles/HISTORY.md): the first inverse pass GENERATING its input instead of reading it — blocks 0 and 1 of the main trace (cycle, pc limbs,
// the instruction's fields, the limbs of R1, R2 and R3's first: nothing but loads of the row's own trace words) are computed in the load stage from the
// 372-B trace, so main_trace_kernel need not write them and this pass need not read them back.  Words as stark.hip: main_trace_row writes them.
struct TraceSrc01 { zkir_trace_columns t; uint64_t n_real; };
__device__ __forceinline__ u32x4 trace_quad01(const TraceSrc01& q, uint32_t blk, uint32_t half, uint64_t i) {
  const bool pad = i >= q.n_real;
  const uint64_t src = pad ? q.n_real - 1 : i;
  u32x4 r;
  if (blk == 0 && half == 0) {
    const uint64_t pcv = q.t.pc[src];
    r.x = (uint32_t)((q.t.cycle[src] + (i - src)) % bb::P); r.y = (uint32_t)(pcv & 0xFFFFF); r.z = (uint32_t)((pcv >> 20) & 0xFFFFF); r.w = (uint32_t)(pcv >> 40);
  } else if (blk == 0) {
    const uint32_t w = q.t.instruction[src];
    r.x = w & 0x7F; r.y = (w >> 7) & 0xF; r.z = (w >> 11) & 0xF; r.w = (w >> 15) & 0xF;
  } else {
    auto limbs = [&](int g, uint32_t out[3]) {
      const uint64_t o = (uint64_t)g * q.t.reg_stride + src;
      const uint64_t v = q.t.registers[o];
      const int bits = q.t.reg_state[o] ? 30 : 20;
      const uint64_t mask = (1ull << bits) - 1;
      out[0] = (uint32_t)(v & mask); out[1] = (uint32_t)((v >> bits) & mask); out[2] = (uint32_t)(v >> (2 * bits));
    };
    uint32_t a[3], b[3];
    if (half == 0) { limbs(1, a); r.x = q.t.instruction[src] >> 19; r.y = a[0]; r.z = a[1]; r.w = a[2]; }
    else { limbs(2, a); limbs(3, b); r.x = a[0]; r.y = a[1]; r.z = a[2]; r.w = b[0]; }
  }
  return r;
}

// ---- strided radix-4 pass: 2R radix-2 stages (R register-resident radix-4 rounds) over tiles of 2^(2R) rows x 2^C positions, in
// place, one column block per blockIdx.y.  With R = 5 a single pass covers ten stages (tile 1024 x 2 positions x 32 B = 64 KiB of
// LDS), so a 2^20-row matrix needs ONE strided pass on each side of the fused middle kernel.  A lane owns one quad of positions and
// runs it for both halves of the block (eight columns) with one set of twiddles: a read of a compact table (root of order <= 1024 /
// 2048) times a per-lane running power; the other stage's twiddle is its square and the odd pair's is its product with a 4th root.
// LDS holds the two halves as separate planes so that consecutive lanes touch consecutive 16-byte words.
template <bool DIT, int R, int C, int NTH, bool FUSED = false>
__global__ __launch_bounds__(NTH) void ntt_strided_r4_kernel(uint4* __restrict__ data, uint64_t blk_u4, uint32_t tiles_per_block, uint32_t total, int L, int s0,
                                                              const uint32_t* __restrict__ tw, const uint32_t* __restrict__ small, int log_small, uint32_t j4_m, TraceSrc01 fsrc = TraceSrc01{}) {
  constexpr int B = 2 * R;
  constexpr uint32_t POS = 1u << (B + C), QUADS = POS / 4, CMASK = (1u << C) - 1, PLANE = POS + 4;       // +4: the two planes start in different banks
  constexpr uint32_t MOVES = 2 * POS / NTH;
  static_assert(QUADS % NTH == 0 && NTH % (1 << C) == 0 && (2 * POS) % NTH == 0, "tile / thread geometry");
  extern __shared__ uint4 lds4[];                                              // [2][PLANE]
  const uint32_t n = 1u << L;
  const uint32_t stride_mid = DIT ? (1u << s0) : (n >> (s0 + B));
  const uint32_t lo_tiles = stride_mid >> C;
  // Persistent workgroups: each walks the (block, tile) work list with a stride of the grid and keeps the NEXT tile's 16-byte loads
  // in flight (registers) while it runs the radix-4 rounds of the current one — with one or two workgroups per CU (64-128 KiB of LDS
  // each) nothing else would hide the global-memory latency of a tile load behind arithmetic.
  auto geometry = [&](uint32_t w, uint4*& x, uint32_t& base, uint32_t& lo0) {
    const uint32_t tile = w % tiles_per_block;
    x = data + (uint64_t)(w / tiles_per_block) * blk_u4;
    const uint32_t hi = tile / lo_tiles;
    lo0 = (tile % lo_tiles) << C;
    base = (DIT ? (hi << (s0 + B)) : hi * (n >> s0)) + lo0;
  };
  static_assert(QUADS == NTH, "one quad per lane and round");
  u32x4 pre[MOVES];
  uint4* x; uint32_t base, lo0;
  uint32_t w = blockIdx.x;
  if (w >= total) return;
  // vmcnt retires IN ORDER: a table read issued after the next tile's prefetch would make its s_waitcnt drain the whole prefetch
  // before the first round starts (rocm 7.2 emitted s_waitcnt vmcnt(0) in round 0 for exactly that) — so nothing is read from global
  // memory between the issue of a prefetch and its use.  The compact-table factors depend on (lane, round) only: read ONCE per
  // workgroup; the tile-dependent power tw[lo << ..] of the NEXT tile is fetched as the last load of that tile's prefetch.
  uint32_t sm[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int b = 2 * r;
    const uint32_t qq = threadIdx.x >> C;
    if (!DIT) { const int lg = B - 2 - b; sm[r] = small[(qq & ((1u << lg) - 1)) << (log_small - (B - b))]; }
    else sm[r] = small[(qq & ((1u << b) - 1)) << (log_small - (b + 2))];
  }
  geometry(w, x, base, lo0);
  static_for<0, (int)MOVES>([&](auto kc) {                                     // tile rows are 2^C consecutive positions = 2^(C+1) uint4
    constexpr uint32_t k = decltype(kc)::value;
    const uint32_t e = threadIdx.x + k * NTH, row = e >> (C + 1), wv = e & ((2u << C) - 1);
    if constexpr (FUSED) pre[k] = trace_quad01(fsrc, w / tiles_per_block, wv & 1, (uint64_t)base + (uint64_t)row * stride_mid + (wv >> 1));
    else pre[k] = ld4(&x[((uint64_t)base + (uint64_t)row * stride_mid) * 2 + wv]);
  });
  uint32_t tw_cur = DIT ? tw[(lo0 + (threadIdx.x & CMASK)) << (L - s0 - B)] : tw[(lo0 + (threadIdx.x & CMASK)) << s0];
  for (;;) {
    static_for<0, (int)MOVES>([&](auto kc) {
      constexpr uint32_t k = decltype(kc)::value;
      const uint32_t e = threadIdx.x + k * NTH, row = e >> (C + 1), wv = e & ((2u << C) - 1);
      st4(&lds4[(wv & 1) * PLANE + ((row << C) | (wv >> 1))], pre[k]);
    });
    uint32_t tp[R];                                            // per-lane power used by round r
    if (DIT) {                                                 // round r needs w^(lo << (L-1-s0-(2r+1))): finest at r = R-1, each earlier round is its 4th power
      uint32_t u = tw_cur;
#pragma unroll
      for (int r = R - 1; r >= 0; r--) { tp[r] = u; u = bb::mont_mul(u, u); u = bb::mont_mul(u, u); }
    } else {                                                   // round r needs w^-(lo << (s0+2r))
      uint32_t u = tw_cur;
#pragma unroll
      for (int r = 0; r < R; r++) { tp[r] = u; u = bb::mont_mul(u, u); u = bb::mont_mul(u, u); }
    }
    __syncthreads();
    const uint32_t wn = w + gridDim.x;
    uint4* xn = x; uint32_t base_n = base, lo0_n = lo0;
    if (wn < total) {                                          // next tile: loads issued now, consumed after this tile's rounds
      geometry(wn, xn, base_n, lo0_n);
      static_for<0, (int)MOVES>([&](auto kc) {
        constexpr uint32_t k = decltype(kc)::value;
        const uint32_t e = threadIdx.x + k * NTH, row = e >> (C + 1), wv = e & ((2u << C) - 1);
        if constexpr (FUSED) pre[k] = trace_quad01(fsrc, wn / tiles_per_block, wv & 1, (uint64_t)base_n + (uint64_t)row * stride_mid + (wv >> 1));
        else pre[k] = ld4(&xn[((uint64_t)base_n + (uint64_t)row * stride_mid) * 2 + wv]);
      });
      tw_cur = DIT ? tw[(lo0_n + (threadIdx.x & CMASK)) << (L - s0 - B)] : tw[(lo0_n + (threadIdx.x & CMASK)) << s0];
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int b = 2 * r;
#pragma unroll
      for (uint32_t k = 0; k < QUADS / NTH; k++) {
        const uint32_t q = threadIdx.x + k * NTH;
        const uint32_t lo_l = q & CMASK, qq = q >> C;
        uint32_t i0, d, wa, wb, wc;
        if (!DIT) {
          const int lg = B - 2 - b;                            // log2(h2) in row units
          const uint32_t h2 = 1u << lg, mid_lo = qq & (h2 - 1), mid_hi = qq >> lg;
          i0 = ((((mid_hi << (lg + 2)) | mid_lo)) << C) | lo_l; d = h2 << C;
          wa = bb::mont_mul(tp[r], sm[r]);
          wb = bb::mont_mul(wa, j4_m); wc = bb::mont_mul(wa, wa);
        } else {
          const uint32_t dm = 1u << b, mid_lo = qq & (dm - 1), mid_hi = qq >> b;
          i0 = ((((mid_hi << (b + 2)) | mid_lo)) << C) | lo_l; d = dm << C;
          wb = bb::mont_mul(tp[r], sm[r]);                                      // w2
          wa = bb::mont_mul(wb, wb); wc = bb::mont_mul(wb, j4_m);               // w1, w2i
        }
        // both halves' eight words are read before either quad is computed: one LDS latency per round instead of two
        uint4* pl0 = lds4; uint4* pl1 = lds4 + PLANE;
        uint4 x0 = pl0[i0], x1 = pl0[i0 + d], x2 = pl0[i0 + 2 * d], x3 = pl0[i0 + 3 * d];
        uint4 y0 = pl1[i0], y1 = pl1[i0 + d], y2 = pl1[i0 + 2 * d], y3 = pl1[i0 + 3 * d];
        if (!DIT) dif4(x0, x1, x2, x3, wa, wb, wc); else dit4(x0, x1, x2, x3, wa, wb, wc);
        pl0[i0] = x0; pl0[i0 + d] = x1; pl0[i0 + 2 * d] = x2; pl0[i0 + 3 * d] = x3;
        if (!DIT) dif4(y0, y1, y2, y3, wa, wb, wc); else dit4(y0, y1, y2, y3, wa, wb, wc);
        pl1[i0] = y0; pl1[i0 + d] = y1; pl1[i0 + 2 * d] = y2; pl1[i0 + 3 * d] = y3;
      }
      // A wave's 64 quads (2^(6-C) values of qq) cover one contiguous "home block" of 4 * 2^(6-C) rows in every round whose quad span
      // fits it: DIF rounds with log2(h2) <= 6 - C, DIT rounds with b <= 6 - C.  Between two such rounds the data never leaves the wave.
      const bool wave_local = !DIT ? (r + 1 < R && B - 2 - b <= 6 - C) : (r + 1 < R && b + 2 <= 6 - C);
      if (wave_local) wave_sync_lds(); else __syncthreads();
    }
#pragma unroll
    for (uint32_t k = 0; k < MOVES; k++) {
      const uint32_t e = threadIdx.x + k * NTH, row = e >> (C + 1), wv = e & ((2u << C) - 1);
      x[((uint64_t)base + (uint64_t)row * stride_mid) * 2 + wv] = lds4[(wv & 1) * PLANE + ((row << C) | (wv >> 1))];
    }
    if (wn >= total) break;
    w = wn; x = xn; base = base_n; lo0 = lo0_n;
    // (no barrier here: a lane refills exactly the LDS words it has just copied out — the copy-in and copy-out loops share one slot map —
    //  and the barrier after the refill orders everything before the first round)
  }
}

// ---- register-only pass of S = 2 or 3 stages (radix 4 / 8), no LDS: what is left over once the ten-stage LDS passes are taken
// (stage counts of 12 = 10 + 2, 13 = 10 + 3, 11 = 8 + 3 ...).  One lane = one (position, half-block) = four columns; it loads its 2^S
// rows (16 bytes each, consecutive lanes on consecutive 16-byte words), runs the S stages in registers and stores them back: 8 B per
// element of HBM traffic for S stages, where the tiny-tile LDS pass (2 stages) followed by the single-stage pass moved 16 B for 3.
// Twiddles: one table read (the finest stage's) and its squares, times the constant 4th / 8th roots.
template <bool DIT, int S>
__global__ __launch_bounds__(NT) void ntt_reg_kernel(uint4* __restrict__ data, uint64_t blk_u4, int L, int s0, const uint32_t* __restrict__ tw, uint32_t r4_m, uint32_t r8_m,
                                                       uint32_t r8_3_m) {
  constexpr int E = 1 << S;
  const uint32_t n = 1u << L;
  uint4* x = data + (uint64_t)blockIdx.y * blk_u4;
  const uint64_t item = (uint64_t)blockIdx.x * NT + threadIdx.x;               // (group t, half h)
  if (item >= ((uint64_t)n >> S) * 2) return;
  const uint32_t t = (uint32_t)(item >> 1), h = (uint32_t)(item & 1);
  const uint32_t d = DIT ? (1u << s0) : (n >> (s0 + S));                       // row distance between the lane's elements
  const uint32_t hi = t / d, lo = t % d;
  const uint64_t base = ((uint64_t)hi * d * E + lo) * 2 + h;
  uint4 v[E];
#pragma unroll
  for (int k = 0; k < E; k++) v[k] = x[base + (uint64_t)k * d * 2];
  if (!DIT) {
    // DIF: stage j pairs (k, k + E/2^(j+1)); twiddle of the pair with low index kl = T_j * rho_{2^(S-j)}^kl, T_j = T0^(2^j), T0 = w^-(lo << s0)
    const uint32_t T0 = tw[lo << s0];
    const uint32_t T1 = bb::mont_mul(T0, T0);
    if (S == 3) {
      const uint32_t T2 = bb::mont_mul(T1, T1);
      const uint32_t a1 = bb::mont_mul(T0, r8_m), a2 = bb::mont_mul(T0, r4_m), a3 = bb::mont_mul(T0, r8_3_m), b1 = bb::mont_mul(T1, r4_m);
      const uint32_t w0[4] = {T0, a1, a2, a3};
#pragma unroll
      for (int k = 0; k < 4; k++) { const uint4 a = v[k], b = v[k + 4]; v[k] = add4(a, b); v[k + 4] = mul4(subl4(a, b), w0[k]); }
#pragma unroll
      for (int g = 0; g < 8; g += 4) {
        { const uint4 a = v[g], b = v[g + 2]; v[g] = add4(a, b); v[g + 2] = mul4(subl4(a, b), T1); }
        { const uint4 a = v[g + 1], b = v[g + 3]; v[g + 1] = add4(a, b); v[g + 3] = mul4(subl4(a, b), b1); }
      }
#pragma unroll
      for (int g = 0; g < 8; g += 2) { const uint4 a = v[g], b = v[g + 1]; v[g] = add4(a, b); v[g + 1] = mul4(subl4(a, b), T2); }
    } else {
      const uint32_t a1 = bb::mont_mul(T0, r4_m);
      { const uint4 a = v[0], b = v[2]; v[0] = add4(a, b); v[2] = mul4(subl4(a, b), T0); }
      { const uint4 a = v[1], b = v[3]; v[1] = add4(a, b); v[3] = mul4(subl4(a, b), a1); }
#pragma unroll
      for (int g = 0; g < 4; g += 2) { const uint4 a = v[g], b = v[g + 1]; v[g] = add4(a, b); v[g + 1] = mul4(subl4(a, b), T1); }
    }
  } else {
    // DIT: stage j pairs (k, k + 2^j); twiddle of the pair with low index kl = U_j * rho_{2^(j+1)}^kl, U_(S-1) = w^(lo << (L - s0 - S)), U_(j-1) = U_j^2
    const uint32_t Uf = tw[lo << (L - s0 - S)];
    const uint32_t Um = bb::mont_mul(Uf, Uf);
    if (S == 3) {
      const uint32_t U0 = bb::mont_mul(Um, Um);
      const uint32_t m1 = bb::mont_mul(Um, r4_m), f1 = bb::mont_mul(Uf, r8_m), f2 = bb::mont_mul(Uf, r4_m), f3 = bb::mont_mul(Uf, r8_3_m);
#pragma unroll
      for (int g = 0; g < 8; g += 2) { const uint4 tt = mul4(v[g + 1], U0); const uint4 a = v[g]; v[g] = add4(a, tt); v[g + 1] = sub4(a, tt); }
#pragma unroll
      for (int g = 0; g < 8; g += 4) {
        { const uint4 tt = mul4(v[g + 2], Um); const uint4 a = v[g]; v[g] = add4(a, tt); v[g + 2] = sub4(a, tt); }
        { const uint4 tt = mul4(v[g + 3], m1); const uint4 a = v[g + 1]; v[g + 1] = add4(a, tt); v[g + 3] = sub4(a, tt); }
      }
      const uint32_t wf[4] = {Uf, f1, f2, f3};
#pragma unroll
      for (int k = 0; k < 4; k++) { const uint4 tt = mul4(v[k + 4], wf[k]); const uint4 a = v[k]; v[k] = add4(a, tt); v[k + 4] = sub4(a, tt); }
    } else {
      const uint32_t f1 = bb::mont_mul(Uf, r4_m);
#pragma unroll
      for (int g = 0; g < 4; g += 2) { const uint4 tt = mul4(v[g + 1], Um); const uint4 a = v[g]; v[g] = add4(a, tt); v[g + 1] = sub4(a, tt); }
      { const uint4 tt = mul4(v[2], Uf); const uint4 a = v[0]; v[0] = add4(a, tt); v[2] = sub4(a, tt); }
      { const uint4 tt = mul4(v[3], f1); const uint4 a = v[1]; v[1] = add4(a, tt); v[3] = sub4(a, tt); }
    }
  }
#pragma unroll
  for (int k = 0; k < E; k++) x[base + (uint64_t)k * d * 2] = v[k];
}

// CUs of the CURRENT device, cached per device id (one process may drive several GPUs)
inline unsigned cu_count() {
  static std::mutex mu;
  static unsigned cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256u;
  std::lock_guard<std::mutex> lk(mu);
  if (!cached[dev]) { hipDeviceProp_t p; cached[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? (unsigned)p.multiProcessorCount : 256u; }
  return cached[dev];
}
inline int persist() { static const int v = getenv("ZKIR_NTT_PERSIST") ? atoi(getenv("ZKIR_NTT_PERSIST")) : 1; return v; }

template <bool DIT, int R, int C, int NTH, bool FUSED = false>
void launch_strided_r4(uint32_t* data, uint64_t n, uint32_t n_blocks, int L, int s0, const uint32_t* tw, const uint32_t* small, int log_small, uint32_t j4_m, hipStream_t s,
                       const TraceSrc01* fsrc = nullptr) {
  constexpr size_t lds = 16u * 2 * ((1u << (2 * R + C)) + 4);
  auto k = ntt_strided_r4_kernel<DIT, R, C, NTH, FUSED>;
  static std::atomic<uint64_t> attr_done{0};                                   // one bit per device id: the attribute is per (function, device)
  int dev = 0; (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_relaxed) & bit)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done.fetch_or(bit); }
  const uint32_t tiles = (uint32_t)(n >> (2 * R + C)), total = tiles * n_blocks;
  const unsigned per_cu = (unsigned)((160u << 10) / lds) > 0 ? (unsigned)((160u << 10) / lds) : 1u;       // workgroups resident per CU (LDS-limited)
  unsigned grid = persist() ? cu_count() * (per_cu > 8 ? 8 : per_cu) : total;
  if (grid > total) grid = total;
  hipLaunchKernelGGL(k, dim3(grid), dim3(NTH), lds, s, (uint4*)data, 2 * n, tiles, total, L, s0, tw, small, log_small, j4_m, fsrc ? *fsrc : TraceSrc01{});
}

inline int strided_c() { static const int c = getenv("ZKIR_NTT_C") ? atoi(getenv("ZKIR_NTT_C")) : 2; return c; }

// `stages` radix-2 stages starting at s0 in as few passes over the data as possible: LDS radix-4 passes of 10 / 8 / 6 / 4 stages and
// register passes of 3 / 2 stages (13 = 10 + 3, 12 = 10 + 2, 11 = 8 + 3, 9 = 6 + 3, 7 = 4 + 3, 5 = 3 + 2); a lone stage only for 1
template <bool DIT>
void run_strided_stages(uint32_t* data, uint64_t n, uint32_t n_blocks, int L, int s0, int stages, const uint32_t* tw, const uint32_t* small, int log_small, uint32_t j4_m,
                        uint32_t r8_m, uint32_t r8_3_m, hipStream_t s) {
  while (stages > 0) {
    int take;
    if (stages >= 10 && stages != 11) take = 10;
    else if (stages == 11) take = 8;
    else if (stages == 9) take = 6;
    else if (stages == 7) take = 4;
    else if (stages == 5) take = 3;
    else take = stages;                                          // 8, 6, 4, 3, 2, 1
    switch (take) {
      case 10:
        // tile rows of 4 positions = 128 contiguous bytes (a full cache line; 128 KiB of LDS, one workgroup of 16 waves per CU) or of
        // 2 positions = 64 bytes (64 KiB, two workgroups of 8 waves): ZKIR_NTT_C picks (benchmarking), default from the measurements
        if (strided_c() == 2) launch_strided_r4<DIT, 5, 2, 1024>(data, n, n_blocks, L, s0, tw, small, log_small, j4_m, s);
        else launch_strided_r4<DIT, 5, 1, 512>(data, n, n_blocks, L, s0, tw, small, log_small, j4_m, s);
        break;
      case 8: launch_strided_r4<DIT, 4, 3, 512>(data, n, n_blocks, L, s0, tw, small, log_small, j4_m, s); break;
      case 6: launch_strided_r4<DIT, 3, 4, 256>(data, n, n_blocks, L, s0, tw, small, log_small, j4_m, s); break;
      case 4: launch_strided_r4<DIT, 2, 5, 128>(data, n, n_blocks, L, s0, tw, small, log_small, j4_m, s); break;
      case 3:
        hipLaunchKernelGGL((ntt_reg_kernel<DIT, 3>), dim3((unsigned)(((n >> 3) * 2 + NT - 1) / NT), n_blocks), dim3(NT), 0, s, (uint4*)data, 2 * n, L, s0, tw, j4_m, r8_m, r8_3_m);
        break;
      case 2:
        hipLaunchKernelGGL((ntt_reg_kernel<DIT, 2>), dim3((unsigned)(((n >> 2) * 2 + NT - 1) / NT), n_blocks), dim3(NT), 0, s, (uint4*)data, 2 * n, L, s0, tw, j4_m, r8_m, r8_3_m);
        break;
      default:
        hipLaunchKernelGGL(ntt_stage_kernel<DIT>, dim3((unsigned)((n + NT - 1) / NT), n_blocks), dim3(NT), 0, s, (uint4*)data, 2 * n, L, s0, tw);
        break;
    }
    s0 += take; stages -= take;
  }
}

// ---- fused middle for N < 1024 (tests, tiny traces): the whole transform of one column block in LDS, column by column, radix 2 ----
//   g_lo[k & 1023] * g_hi[k >> 10] = g^k * N^-1   (two-level power table, Montgomery form)
__global__ __launch_bounds__(NT) void lde_small_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int L, const uint32_t* __restrict__ small_inv,
                                                        const uint32_t* __restrict__ small_fwd, const uint32_t* __restrict__ g_lo, const uint32_t* __restrict__ g_hi) {
  extern __shared__ uint32_t lds[];                            // 2N words
  const uint32_t n = 1u << L;
  const uint32_t* x = in + (uint64_t)blockIdx.x * n * 8;
  uint32_t* y = out + (uint64_t)blockIdx.x * 2 * n * 8;
  for (int c = 0; c < 8; c++) {
    for (uint32_t e = threadIdx.x; e < n; e += NT) lds[e] = x[(uint64_t)e * 8 + c];
    __syncthreads();
    for (int b = 0; b < L; b++) {                              // inverse DIF stages, half = 2^(L-1-b)
      const int hb = L - 1 - b;
      const uint32_t half = 1u << hb;
      for (uint32_t q = threadIdx.x; q < (n >> 1); q += NT) {
        const uint32_t r_lo = q & (half - 1), r_hi = q >> hb;
        const uint32_t ia = (r_hi << (hb + 1)) | r_lo, ib = ia + half;
        const uint32_t a = lds[ia], bv = lds[ib];
        lds[ia] = bb::add(a, bv); lds[ib] = bb::mont_mul(bb::sub(a, bv), small_inv[r_lo << b]);
      }
      __syncthreads();
    }
    // scale + zero-interleave: position p holds coefficient k = bitrev_L(p); DIT stage 0 duplicates
    uint32_t v[2];
    int cnt = 0;
    for (uint32_t e = threadIdx.x; e < n; e += NT) { const uint32_t k = bitrev(e, L); v[cnt++] = bb::mont_mul(bb::mont_mul(lds[e], g_lo[k & 1023]), g_hi[k >> 10]); }
    __syncthreads();
    cnt = 0;
    for (uint32_t e = threadIdx.x; e < n; e += NT) { lds[2 * e] = v[cnt]; lds[2 * e + 1] = v[cnt]; cnt++; }
    __syncthreads();
    for (int s = 1; s <= L; s++) {                             // forward DIT stages 1..L of the size-2N transform
      const uint32_t half = 1u << s;
      for (uint32_t q = threadIdx.x; q < n; q += NT) {
        const uint32_t r_lo = q & (half - 1), r_hi = q >> s;
        const uint32_t ia = (r_hi << (s + 1)) | r_lo, ib = ia + half;
        const uint32_t a = lds[ia], t = bb::mont_mul(lds[ib], small_fwd[r_lo << (L - s)]);
        lds[ia] = bb::add(a, t); lds[ib] = bb::sub(a, t);
      }
      __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < 2 * n; e += NT) y[(uint64_t)e * 8 + c] = lds[e];
    __syncthreads();
  }
}

// ---- fused middle, N >= 1024: the last 10 inverse-DIF stages on a contiguous chunk of 1024 positions of the size-N block, the coset
// scale g^k / N (k = bit-reversal of the position), zero-interleave, and the first 11 forward-DIT stages of the size-2N transform,
// written as the 2048-position chunk of `out`.  Register-resident radix-4 rounds: 5 + 5 LDS round trips for the 21 stages, one
// twiddle read per quad (the others are its square and its product with a 4th root of unity), shared by the four columns a lane
// carries.  LDS: A = both halves of the chunk (32 KiB; later the forward buffer of half 1), Bf = the forward buffer of half 0 (32 KiB):
// 64 KiB per workgroup, two workgroups per CU.
constexpr int MID_NT = 512;
__global__ __launch_bounds__(MID_NT) void lde_middle_r4_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t chunks_per_block, uint32_t total, int L,
                                                                const uint32_t* __restrict__ small_inv, const uint32_t* __restrict__ small_fwd,
                                                                const uint32_t* __restrict__ g_lo, const uint32_t* __restrict__ g_hi, uint32_t j4_inv_m, uint32_t j4_fwd_m) {
  constexpr int Bm = 10;
  constexpr uint32_t APL = 1024;
  __shared__ uint4 A[2 * APL];
  __shared__ uint4 Bf[2048];
  const uint32_t n = 1u << L, t = threadIdx.x;
  // persistent workgroups with the next chunk's loads in flight during the 15 rounds of the current one (see the strided kernel)
  uint32_t w = blockIdx.x;
  if (w >= total) return;
  // vmcnt retires in order (see the strided kernel): no global read between the issue of the next chunk's prefetch and its use.
  // The twiddles of the ten rounds depend on (lane, round) only: read once per workgroup.  The coset scale of the lane's four
  // positions depends on the chunk: read at the top of the chunk, BEFORE the next prefetch is issued.
  uint32_t tw_i[5], tw_f[5];
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const int lg = 8 - 2 * r;
    tw_i[r] = small_inv[((t & 255) & ((1u << lg) - 1)) << (2 * r)];           // w_1024^-(lo << 2r)
    const int sft = 2 * r + 1;
    tw_f[r] = small_fwd[(t & ((1u << sft) - 1)) << (Bm - sft - 1)];            // w_2048^(lo << (9-s)): twiddle of stage s+1
  }
  u32x4 pre[4];
  uint32_t glo[4], ghi[4];                                     // factors of g^k / N for the lane's four positions 4q .. 4q + 3 of the last inverse round, k = bitrev_L(position)
  auto fetch = [&](uint32_t ww) {                              // one chunk's loads, in the order they are consumed: the 1024 positions, then the scale factors
    const uint4* x = in + ((uint64_t)(ww / chunks_per_block) * n + ((uint64_t)(ww % chunks_per_block) << Bm)) * 2;
#pragma unroll
    for (int k = 0; k < 4; k++) pre[k] = ld4(&x[t + k * MID_NT]);
    const uint32_t p0 = ((ww % chunks_per_block) << Bm) + 4 * (t & 255);
#pragma unroll
    for (int i = 0; i < 4; i++) { const uint32_t k = bitrev(p0 + i, L); glo[i] = g_lo[k & 1023]; ghi[i] = g_hi[k >> 10]; }
  };
  fetch(w);
  for (;;) {
    const uint32_t base = (w % chunks_per_block) << Bm;
    uint4* y = out + (uint64_t)(w / chunks_per_block) * n * 4;
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t e = t + k * MID_NT; st4(&A[(e & 1) * APL + (e >> 1)], pre[k]); }
    uint32_t gs[4];
#pragma unroll
    for (int i = 0; i < 4; i++) gs[i] = bb::mont_mul(glo[i], ghi[i]);
    __syncthreads();
    const uint32_t wn = w + gridDim.x;
    if (wn < total) fetch(wn);
    // ---- inverse DIF, rounds r = 0..4: stages (2r, 2r+1), spans h1 = 2^(9-2r), h2 = h1/2; lane = (quad q, half h) ----
    {
      const uint32_t q = t & 255;
      uint4* a = A + (t >> 8) * APL;
#pragma unroll
      for (int r = 0; r < 5; r++) {
        const int lg = 8 - 2 * r;                              // log2(h2)
        const uint32_t h2 = 1u << lg, lo = q & (h2 - 1), hi = q >> lg;
        const uint32_t i0 = (hi << (lg + 2)) | lo;
        const uint32_t wA = tw_i[r];                            // w_1024^-(lo << 2r)
        const uint32_t wB = bb::mont_mul(wA, j4_inv_m), w2 = bb::mont_mul(wA, wA);
        uint4 x0 = a[i0], x1 = a[i0 + h2], x2 = a[i0 + 2 * h2], x3 = a[i0 + 3 * h2];
        dif4(x0, x1, x2, x3, wA, wB, w2);
        if (r == 4) {                                          // last round (positions 4q..4q+3, i0 = 4q): the coset scale g^k / N
          x0 = mul4(x0, gs[0]); x1 = mul4(x1, gs[1]); x2 = mul4(x2, gs[2]); x3 = mul4(x3, gs[3]);
        }
        a[i0] = x0; a[i0 + h2] = x1; a[i0 + 2 * h2] = x2; a[i0 + 3 * h2] = x3;
        // rounds 1..4 stay inside the wave's own 256 positions of its plane (q = 64 v .. 64 v + 63): only round 0 hands data to other waves
        if (r >= 1 && r < 4) wave_sync_lds(); else __syncthreads();
      }
    }
    // ---- forward DIT of the zero-interleaved chunk (2048 positions), one half of the block at a time: stage 0 is a copy, rounds do
    //      stages (s, s+1), s = 1,3,5,7,9; lane = quad t of 512 ----
    // Half 0 runs its forward rounds in Bf; half 1 then runs them in A itself (A is dead once round 0 of half 1 has read it — one
    // extra barrier between that round's loads and stores), so that both halves of a position are stored TOGETHER: 32 contiguous
    // bytes per lane whose two 16-byte stores meet in L2.  Stored one half at a time, 10 us apart, every 32-byte sector reached HBM
    // twice, half-filled: WRITE_SIZE showed 2.40 GB per launch against 1.28 GB algorithmic (profiles/r02i_bench_commit_pmc_traffic.txt).
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
      const uint4* a = A + h * APL;
      uint4* F = h == 0 ? Bf : A;
#pragma unroll
      for (int r = 0; r < 5; r++) {
        const int s = 2 * r + 1;
        const uint32_t lo = t & ((1u << s) - 1), hi = t >> s;
        const uint32_t i0 = (hi << (s + 2)) | lo, d = 1u << s;
        const uint32_t w2 = tw_f[r];                            // w_2048^(lo << (9-s)): twiddle of stage s+1
        const uint32_t w1 = bb::mont_mul(w2, w2), w2i = bb::mont_mul(w2, j4_fwd_m);
        uint4 x0, x1, x2, x3;
        if (r == 0) {                                          // after stage 0: F[j] = A[j >> 1]
          x0 = a[i0 >> 1]; x1 = a[(i0 + d) >> 1]; x2 = a[(i0 + 2 * d) >> 1]; x3 = a[(i0 + 3 * d) >> 1];
          __syncthreads();                                     // half 1 overwrites what it has just read
        } else { x0 = F[i0]; x1 = F[i0 + d]; x2 = F[i0 + 2 * d]; x3 = F[i0 + 3 * d]; }
        dit4(x0, x1, x2, x3, w1, w2, w2i);
        F[i0] = x0; F[i0 + d] = x1; F[i0 + 2 * d] = x2; F[i0 + 3 * d] = x3;
        // rounds 0..2 (spans up to 128) stay inside the wave's own 256 positions [256 v, 256 v + 256) of F; rounds 3 and 4 cross waves
        if (r < 2) wave_sync_lds(); else __syncthreads();
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t e = t + k * MID_NT; y[((uint64_t)2 * base + e) * 2] = Bf[e]; y[((uint64_t)2 * base + e) * 2 + 1] = A[e]; }
    __syncthreads();
    if (wn >= total) break;
    w = wn;
  }
}

}  // namespace

namespace zkir {

// in: n_blocks x [N][8] canonical evaluations over H (natural order; used as scratch and overwritten!), out: n_blocks x [2N][8]
void lde_run(const LdeTables& t, uint32_t* in, uint32_t n_blocks, uint32_t* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int L = t.log_n;
  const uint32_t N = 1u << L;
  static const uint32_t j4_inv_m = bb::to_mont(bb::inv(bb::root_of_unity(2))), j4_fwd_m = bb::to_mont(bb::root_of_unity(2));
  static const uint32_t r8_fwd = bb::root_of_unity(3), r8_inv = bb::inv(r8_fwd);
  static const uint32_t r8_inv_m = bb::to_mont(r8_inv), r8_inv3_m = bb::to_mont(bb::mul(bb::mul(r8_inv, r8_inv), r8_inv));
  static const uint32_t r8_fwd_m = bb::to_mont(r8_fwd), r8_fwd3_m = bb::to_mont(bb::mul(bb::mul(r8_fwd, r8_fwd), r8_fwd));
  if (L < 10) {
    hipLaunchKernelGGL(lde_small_kernel, dim3(n_blocks), dim3(NT), (8u << L), s, in, out, L, t.small_inv, t.small_fwd, t.g_lo, t.g_hi);
    return;
  }
  // inverse DIF strided stages 0 .. L-11 (the compact table then has order 1024)
  run_strided_stages<false>(in, N, n_blocks, L, 0, L - 10, t.tw_inv, t.small_inv, 10, j4_inv_m, r8_inv_m, r8_inv3_m, s);
  {
    const uint32_t chunks = N >> 10, total = chunks * n_blocks;
    unsigned grid = persist() ? cu_count() * 2 : total;                     // 64 KiB of LDS: two workgroups per CU
    if (grid > total) grid = total;
    hipLaunchKernelGGL(lde_middle_r4_kernel, dim3(grid), dim3(MID_NT), 0, s, (const uint4*)in, (uint4*)out, chunks, total, L, t.small_inv, t.small_fwd, t.g_lo, t.g_hi,
                       j4_inv_m, j4_fwd_m);
  }
  // forward DIT strided stages 11 .. L of the size-2N transform
  run_strided_stages<true>(out, (uint64_t)2 * N, n_blocks, L + 1, 11, L - 10, t.tw_fwd, t.small_fwd, 11, j4_fwd_m, r8_fwd_m, r8_fwd3_m, s);
}


// EXPERIMENT (profiles/r04*_lde_tile_variants.txt): ONE strided pass at stage 0 over n_blocks blocks of 2^log_n rows with a chosen tile geometry — the timing of the
// tilings side by side on the same data (the values are a partial transform: only the time means anything).
//   0: 10 stages, 1024 rows x 4 positions (128-byte rows), 128 KiB, 1024 lanes: one workgroup per CU   (what lde_run uses)
//   1: 10 stages, 1024 rows x 2 positions (64-byte rows),   64 KiB,  512 lanes: two per CU
//   2:  8 stages,  256 rows x 8 positions (256-byte rows),  64 KiB,  512 lanes: two per CU
//   3:  8 stages,  256 rows x 4 positions (128-byte rows),  32 KiB,  256 lanes: four or five per CU
//   4:  6 stages,   64 rows x 16 positions,                 32 KiB,  256 lanes
bool strided_variant_run(const LdeTables& t, uint32_t* data, uint32_t n_blocks, int variant, bool dit, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int L = dit ? t.log_n + 1 : t.log_n;
  if (t.log_n < 20) return false;
  const uint64_t n = (uint64_t)1 << L;
  static const uint32_t j4_inv_m = bb::to_mont(bb::inv(bb::root_of_unity(2))), j4_fwd_m = bb::to_mont(bb::root_of_unity(2));
  const uint32_t* tw = dit ? t.tw_fwd : t.tw_inv; const uint32_t* sm = dit ? t.small_fwd : t.small_inv;
  const int ls = dit ? 11 : 10; const uint32_t j4 = dit ? j4_fwd_m : j4_inv_m; const int s0 = dit ? 11 : 0;
#define ZKIR_VARIANT(R, C, NTH) do { if (dit) launch_strided_r4<true, R, C, NTH>(data, n, n_blocks, L, s0, tw, sm, ls, j4, s); else launch_strided_r4<false, R, C, NTH>(data, n, n_blocks, L, s0, tw, sm, ls, j4, s); } while (0)
  switch (variant) {
    case 0: ZKIR_VARIANT(5, 2, 1024); break;
    case 1: ZKIR_VARIANT(5, 1, 512); break;
    case 2: ZKIR_VARIANT(4, 3, 512); break;
    case 3: ZKIR_VARIANT(4, 2, 256); break;
    case 4: ZKIR_VARIANT(3, 4, 256); break;
    default: return false;
  }
#undef ZKIR_VARIANT
  return true;
}

// EXPERIMENT: the same extension with blocks 0 and 1 of `in` NOT read by the first inverse pass — generated from the trace instead (trace_quad01).  Only for
// traces whose first inverse pass is the ten-stage one (log_n >= 20, log_n != 21); returns false (nothing launched) otherwise.
bool lde_run_fused01(const LdeTables& t, const zkir_trace_columns* trace, uint64_t n_real, uint32_t* in, uint32_t n_blocks, uint32_t* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int L = t.log_n;
  if (L < 20 || L == 21 || n_blocks < 3 || strided_c() != 2) return false;
  const uint32_t N = 1u << L;
  static const uint32_t j4_inv_m = bb::to_mont(bb::inv(bb::root_of_unity(2))), j4_fwd_m = bb::to_mont(bb::root_of_unity(2));
  static const uint32_t r8_fwd = bb::root_of_unity(3), r8_inv = bb::inv(r8_fwd);
  static const uint32_t r8_inv_m = bb::to_mont(r8_inv), r8_inv3_m = bb::to_mont(bb::mul(bb::mul(r8_inv, r8_inv), r8_inv));
  static const uint32_t r8_fwd_m = bb::to_mont(r8_fwd), r8_fwd3_m = bb::to_mont(bb::mul(bb::mul(r8_fwd, r8_fwd), r8_fwd));
  const TraceSrc01 src{*trace, n_real};
  launch_strided_r4<false, 5, 2, 1024, true>(in, N, 2, L, 0, t.tw_inv, t.small_inv, 10, j4_inv_m, s, &src);                         // blocks 0, 1: generated
  launch_strided_r4<false, 5, 2, 1024>(in + (uint64_t)2 * N * 8, N, n_blocks - 2, L, 0, t.tw_inv, t.small_inv, 10, j4_inv_m, s);     // the others: read
  if (L - 10 > 10) run_strided_stages<false>(in, N, n_blocks, L, 10, L - 20, t.tw_inv, t.small_inv, 10, j4_inv_m, r8_inv_m, r8_inv3_m, s);
  {
    const uint32_t chunks = N >> 10, total = chunks * n_blocks;
    unsigned grid = persist() ? cu_count() * 2 : total;
    if (grid > total) grid = total;
    hipLaunchKernelGGL(lde_middle_r4_kernel, dim3(grid), dim3(MID_NT), 0, s, (const uint4*)in, (uint4*)out, chunks, total, L, t.small_inv, t.small_fwd, t.g_lo, t.g_hi,
                       j4_inv_m, j4_fwd_m);
  }
  run_strided_stages<true>(out, (uint64_t)2 * N, n_blocks, L + 1, 11, L - 10, t.tw_fwd, t.small_fwd, 11, j4_fwd_m, r8_fwd_m, r8_fwd3_m, s);
  return true;
}

}  // namespace zkir
