// stark.hip — self-defined prover stages over Baby Bear on gfx950 ("ZKIR-STARK v0").
//
// The reference has none of this (parity unpinned); the spec is oracle/stark_oracle.cpp and every
// kernel here is checked bit-for-bit against it.  Stage A (this part): main-trace field columns, radix-2 NTT / coset LDE,
// Poseidon2-12 Merkle commitment.
//
//   main_trace_kernel   372 B/row SoA trace -> 89 Baby Bear columns (limbs of pc / instruction fields / registers, state and
//                       "changed" flags).  HBM-bound: ~744 B read (row + next row, second read L2-hot) + 356 B written per row.
//   ntt passes          per column: inverse DIF NTT over H (natural -> bit-reversed), coset scale, zero-interleave, forward DIT
//                       NTT over the 2N coset (bit-reversed -> natural).  LDS-staged radix-2^B passes: strided passes move
//                       tiles of 2^B x 2^C elements (2^C consecutive words per row of the tile keep loads coalesced); the last
//                       B_m inverse stages, the scaling and the first B_m+1 forward stages are fused in one contiguous-tile
//                       kernel.  HBM-bound: 8 B/element per strided pass, 12 B/element for the fused middle.
//   merkle kernels      Poseidon2 width-12 sponge over the rows of the LDE matrix (one lane per leaf, column reads coalesced
//                       across lanes) + 2-to-1 compression layers.  ALU-bound (≈740 Montgomery multiplications per permutation);
//                       no MFMA: 31-bit modular integer work, no dense contraction.
#include <hip/hip_runtime.h>

#include <array>
#include <cstddef>
#include <vector>

#include "../../include/zkir_amd.h"
#include "babybear.h"
#include "host.h"
#include "poseidon2.h"

namespace {

constexpr int NT = 256;
__constant__ p2::Consts d_p2;

inline unsigned grid_for(uint64_t n, int per = NT) { return (unsigned)((n + per - 1) / per); }
int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { zkir::set_last_error({ZKIR_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)}); return ZKIR_ERR_DEVICE; }
  return ZKIR_OK;
}

__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return bits == 0 ? 0u : __brev(x) >> (32 - bits); }

// ------------------------------------------------------------------------------------------------
// main trace columns (oracle: so::main_trace)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void main_trace_kernel(zkir_trace_columns t, uint64_t n, uint32_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  auto col = [&](int k) -> uint32_t& { return out[(uint64_t)k * n + i]; };
  col(0) = (uint32_t)(t.cycle[i] % bb::P);
  const uint64_t pc = t.pc[i];
  col(1) = (uint32_t)(pc & 0xFFFFF); col(2) = (uint32_t)((pc >> 20) & 0xFFFFF); col(3) = (uint32_t)(pc >> 40);
  const uint32_t w = t.instruction[i];
  col(4) = w & 0x7F; col(5) = (w >> 7) & 0xF; col(6) = (w >> 11) & 0xF; col(7) = (w >> 15) & 0xF; col(8) = w >> 19;
  const bool last = i + 1 >= n;
#pragma unroll 4
  for (int g = 0; g < 16; g++) {
    const uint64_t o = (uint64_t)g * t.reg_stride + i;
    const uint64_t v = t.registers[o];
    const uint32_t st = t.reg_state[o];
    const int bits = st ? 30 : 20;
    const uint64_t mask = (1ull << bits) - 1;
    col(9 + 3 * g) = (uint32_t)(v & mask); col(10 + 3 * g) = (uint32_t)((v >> bits) & mask); col(11 + 3 * g) = (uint32_t)(v >> (2 * bits));
    col(57 + g) = st;
    uint32_t ch = 0;
    if (!last) ch = (t.registers[o + 1] != v) | (t.reg_state[o + 1] != st) | (t.bound_bits[o + 1] != t.bound_bits[o]) | (t.bound_tag[o + 1] != t.bound_tag[o]) |
                    (t.bound_payload[o + 1] != t.bound_payload[o]);
    col(73 + g) = ch;
  }
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

// One strided pass of B (<= 5) radix-2 stages over tiles of 2^B x 2^C elements, in place, one column per blockIdx.y.
//   DIT == false: inverse DIF stages s0..s0+B-1 of a size-2^L transform; DIT == true: forward DIT stages s0..s0+B-1.
// Twiddles never come from a big strided table lookup per butterfly: the exponent splits into a per-lane part that only
// depends on `lo` (one table read per thread per tile, then repeated squaring across the stages) and a root of unity of
// order <= 2^B indexed by the position inside the tile (compact table `small` of order 2^log_small, L1-resident).
template <bool DIT>
__global__ __launch_bounds__(NT) void ntt_strided_kernel(uint32_t* __restrict__ data, uint64_t col_stride, int L, int s0, int B, int C, const uint32_t* __restrict__ tw,
                                                          const uint32_t* __restrict__ small, int log_small) {
  extern __shared__ uint32_t lds[];
  uint32_t* x = data + (uint64_t)blockIdx.y * col_stride;
  const uint32_t n = 1u << L;
  const uint32_t stride_mid = DIT ? (1u << s0) : (n >> (s0 + B));
  const uint32_t lo_tiles = stride_mid >> C;
  const uint32_t tile = blockIdx.x;
  const uint32_t hi = tile / lo_tiles, lo0 = (tile % lo_tiles) << C;
  const uint32_t base = (DIT ? (hi << (s0 + B)) : hi * (n >> s0)) + lo0;
  const uint32_t elems = 1u << (B + C), cmask = (1u << C) - 1;
  for (uint32_t e = threadIdx.x; e < elems; e += NT) lds[e] = x[base + (e >> C) * stride_mid + (e & cmask)];
  // per-lane twiddle powers (NT is a multiple of 2^C, so a thread always works on the same `lo`)
  const uint32_t lo = lo0 + (threadIdx.x & cmask);
  uint32_t tp[5];
  if (DIT) {                                                   // stage b needs w^(lo << (L-1-s0-b)): finest at b = B-1, each coarser stage squares it
    uint32_t u = tw[lo << (L - s0 - B)];
#pragma unroll
    for (int b = 4; b >= 0; b--) if (b < B) { tp[b] = u; u = bb::mont_mul(u, u); }
  } else {                                                     // stage b needs w^-(lo << (s0+b))
    uint32_t u = tw[lo << s0];
#pragma unroll
    for (int b = 0; b < 5; b++) if (b < B) { tp[b] = u; u = bb::mont_mul(u, u); }
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 5; b++) {
    if (b < B) {
      const int hb = DIT ? b : (B - 1 - b);                    // log2 of the half-span in `mid` units
      const uint32_t half_mid = 1u << hb;
      const int sh = log_small - (hb + 1);                     // small-root order 2^(hb+1)
      for (uint32_t q = threadIdx.x; q < (elems >> 1); q += NT) {
        const uint32_t lo_l = q & cmask, r = q >> C;
        const uint32_t mid_lo = r & (half_mid - 1), mid_hi = r >> hb;
        const uint32_t ia = (((mid_hi << (hb + 1)) | mid_lo) << C) | lo_l, ib = ia + (half_mid << C);
        const uint32_t w = bb::mont_mul(tp[b], small[mid_lo << sh]);
        const uint32_t a = lds[ia], bv = lds[ib];
        if (DIT) {
          const uint32_t t = bb::mont_mul(bv, w);
          lds[ia] = bb::add(a, t); lds[ib] = bb::sub(a, t);
        } else {
          lds[ia] = bb::add(a, bv); lds[ib] = bb::mont_mul(bb::sub(a, bv), w);
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t e = threadIdx.x; e < elems; e += NT) x[base + (e >> C) * stride_mid + (e & cmask)] = lds[e];
}

// Radix-4 strided pass: 2R radix-2 stages (R register-resident radix-4 rounds) over tiles of 2^(2R) x 2^C elements, in place.
// With R = 5 a single pass covers ten stages (tile 1024 x 16 words = 64 KiB of LDS), so a 2^20-point column needs ONE strided
// pass on each side of the fused middle kernel instead of two (36 B/element of HBM traffic per column instead of 60).
// Twiddles per quad: one read of a compact table (root of order <= 1024/2048) times a per-lane running power; the other
// stage's twiddle is its square and the odd pair's is its product with a 4th root of unity.
template <bool DIT, int R, int C, int NTH>
__global__ __launch_bounds__(NTH) void ntt_strided_r4_kernel(uint32_t* __restrict__ data, uint64_t col_stride, int L, int s0, const uint32_t* __restrict__ tw,
                                                              const uint32_t* __restrict__ small, int log_small, uint32_t j4_m) {
  constexpr int B = 2 * R;
  constexpr uint32_t ELEMS = 1u << (B + C), QUADS = ELEMS / 4, CMASK = (1u << C) - 1;
  static_assert(QUADS % NTH == 0 && NTH % (1 << C) == 0, "tile / thread geometry");
  extern __shared__ uint32_t lds[];
  uint32_t* x = data + (uint64_t)blockIdx.y * col_stride;
  const uint32_t n = 1u << L;
  const uint32_t stride_mid = DIT ? (1u << s0) : (n >> (s0 + B));
  const uint32_t lo_tiles = stride_mid >> C;
  const uint32_t tile = blockIdx.x;
  const uint32_t hi = tile / lo_tiles, lo0 = (tile % lo_tiles) << C;
  const uint32_t base = (DIT ? (hi << (s0 + B)) : hi * (n >> s0)) + lo0;
  for (uint32_t e = threadIdx.x; e < ELEMS; e += NTH) lds[e] = x[base + (e >> C) * stride_mid + (e & CMASK)];
  const uint32_t lo = lo0 + (threadIdx.x & CMASK);
  uint32_t tp[R];                                              // per-lane power used by round r
  if (DIT) {                                                   // round r needs w^(lo << (L-1-s0-(2r+1))): finest at r = R-1, each earlier round is its 4th power
    uint32_t u = tw[lo << (L - s0 - B)];
#pragma unroll
    for (int r = R - 1; r >= 0; r--) { tp[r] = u; u = bb::mont_mul(u, u); u = bb::mont_mul(u, u); }
  } else {                                                     // round r needs w^-(lo << (s0+2r))
    uint32_t u = tw[lo << s0];
#pragma unroll
    for (int r = 0; r < R; r++) { tp[r] = u; u = bb::mont_mul(u, u); u = bb::mont_mul(u, u); }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int b = 2 * r;
#pragma unroll
    for (uint32_t k = 0; k < QUADS / NTH; k++) {
      const uint32_t q = threadIdx.x + k * NTH;
      const uint32_t lo_l = q & CMASK, qq = q >> C;
      if (!DIT) {
        const int lg = B - 2 - b;                              // log2(h2) in mid units
        const uint32_t h2 = 1u << lg, mid_lo = qq & (h2 - 1), mid_hi = qq >> lg;
        const uint32_t i0 = ((((mid_hi << (lg + 2)) | mid_lo)) << C) | lo_l, d = h2 << C;
        const uint32_t x0 = lds[i0], x1 = lds[i0 + d], x2 = lds[i0 + 2 * d], x3 = lds[i0 + 3 * d];
        const uint32_t wA = bb::mont_mul(tp[r], small[mid_lo << (log_small - (B - b))]);
        const uint32_t wB = bb::mont_mul(wA, j4_m), w2 = bb::mont_mul(wA, wA);
        const uint32_t y0 = bb::add(x0, x2), y2 = bb::mont_mul(bb::sub(x0, x2), wA);
        const uint32_t y1 = bb::add(x1, x3), y3 = bb::mont_mul(bb::sub(x1, x3), wB);
        lds[i0] = bb::add(y0, y1); lds[i0 + d] = bb::mont_mul(bb::sub(y0, y1), w2);
        lds[i0 + 2 * d] = bb::add(y2, y3); lds[i0 + 3 * d] = bb::mont_mul(bb::sub(y2, y3), w2);
      } else {
        const uint32_t dm = 1u << b, mid_lo = qq & (dm - 1), mid_hi = qq >> b;
        const uint32_t i0 = ((((mid_hi << (b + 2)) | mid_lo)) << C) | lo_l, d = dm << C;
        const uint32_t x0 = lds[i0], x1 = lds[i0 + d], x2 = lds[i0 + 2 * d], x3 = lds[i0 + 3 * d];
        const uint32_t w2 = bb::mont_mul(tp[r], small[mid_lo << (log_small - (b + 2))]);
        const uint32_t w1 = bb::mont_mul(w2, w2), w2i = bb::mont_mul(w2, j4_m);
        const uint32_t t1 = bb::mont_mul(x1, w1), t3 = bb::mont_mul(x3, w1);
        const uint32_t y0 = bb::add(x0, t1), y1 = bb::sub(x0, t1), y2 = bb::add(x2, t3), y3 = bb::sub(x2, t3);
        const uint32_t u2 = bb::mont_mul(y2, w2), u3 = bb::mont_mul(y3, w2i);
        lds[i0] = bb::add(y0, u2); lds[i0 + 2 * d] = bb::sub(y0, u2); lds[i0 + d] = bb::add(y1, u3); lds[i0 + 3 * d] = bb::sub(y1, u3);
      }
    }
    __syncthreads();
  }
  for (uint32_t e = threadIdx.x; e < ELEMS; e += NTH) x[base + (e >> C) * stride_mid + (e & CMASK)] = lds[e];
}

template <bool DIT, int R, int C, int NTH>
void launch_strided_r4(uint32_t* data, uint64_t n, uint32_t width, int L, int s0, const uint32_t* tw, const uint32_t* small, int log_small, uint32_t j4_m, hipStream_t s) {
  constexpr size_t lds = 4u << (2 * R + C);
  auto k = ntt_strided_r4_kernel<DIT, R, C, NTH>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)(n >> (2 * R + C)), width), dim3(NTH), lds, s, data, n, L, s0, tw, small, log_small, j4_m);
}

// `stages` radix-2 stages starting at s0, as few passes as possible: radix-4 passes of 10/8/6/4/2 stages + a radix-2 pass for an odd one
template <bool DIT>
void run_strided_stages(uint32_t* data, uint64_t n, uint32_t width, int L, int s0, int stages, const uint32_t* tw, const uint32_t* small, int log_small, uint32_t j4_m,
                        hipStream_t s) {
  while (stages > 0) {
    int R = stages / 2 > 5 ? 5 : stages / 2;
    if (stages - 2 * R == 1 && R == 5) R = 4;                  // keep an even remainder (e.g. 11 = 8 + 2 + 1 is avoided: 11 -> 8 + ... )
    if (R == 0) {                                              // single leftover stage: radix-2 kernel
      const int C = 6;
      hipLaunchKernelGGL(ntt_strided_kernel<DIT>, dim3((unsigned)(n >> (1 + C)), width), dim3(NT), (4u << (1 + C)), s, data, n, L, s0, 1, C, tw, small, log_small);
      s0 += 1; stages -= 1;
      continue;
    }
    switch (R) {
      case 5: launch_strided_r4<DIT, 5, 4, 1024>(data, n, width, L, s0, tw, small, log_small, j4_m, s); break;
      case 4: launch_strided_r4<DIT, 4, 6, 1024>(data, n, width, L, s0, tw, small, log_small, j4_m, s); break;
      case 3: launch_strided_r4<DIT, 3, 6, 256>(data, n, width, L, s0, tw, small, log_small, j4_m, s); break;
      case 2: launch_strided_r4<DIT, 2, 6, 256>(data, n, width, L, s0, tw, small, log_small, j4_m, s); break;
      default: launch_strided_r4<DIT, 1, 6, 64>(data, n, width, L, s0, tw, small, log_small, j4_m, s); break;
    }
    s0 += 2 * R; stages -= 2 * R;
  }
}

// Fused middle: last Bm inverse-DIF stages on a contiguous 2^Bm chunk of the size-N array `in`, scale by g^k / N
// (k = bit-reversal of the position), zero-interleave, first Bm+1 forward-DIT stages, write the 2^(Bm+1) chunk of `out`.
//   g_lo[k & 1023] * g_hi[k >> 10] = g^k * N^-1   (two-level power table, Montgomery form)
__global__ __launch_bounds__(NT) void lde_middle_kernel(const uint32_t* __restrict__ in, uint64_t in_stride, uint32_t* __restrict__ out, uint64_t out_stride, int L,
                                                         int Bm, const uint32_t* __restrict__ small_inv, const uint32_t* __restrict__ small_fwd,
                                                         const uint32_t* __restrict__ g_lo, const uint32_t* __restrict__ g_hi) {
  extern __shared__ uint32_t lds[];                            // 2^(Bm+1) words
  const uint32_t* x = in + (uint64_t)blockIdx.y * in_stride;
  uint32_t* y = out + (uint64_t)blockIdx.y * out_stride;
  const uint32_t chunk = 1u << Bm, base = blockIdx.x << Bm;
  for (uint32_t e = threadIdx.x; e < chunk; e += NT) lds[e] = x[base + e];
  __syncthreads();
  for (int b = 0; b < Bm; b++) {                               // inverse DIF stages s = L-Bm+b, half = 2^(Bm-1-b)
    const int hb = Bm - 1 - b;
    const uint32_t half = 1u << hb;
    for (uint32_t q = threadIdx.x; q < (chunk >> 1); q += NT) {
      const uint32_t r_lo = q & (half - 1), r_hi = q >> hb;
      const uint32_t ia = (r_hi << (hb + 1)) | r_lo, ib = ia + half;
      const uint32_t a = lds[ia], bv = lds[ib];
      lds[ia] = bb::add(a, bv); lds[ib] = bb::mont_mul(bb::sub(a, bv), small_inv[r_lo << b]);   // w_N^-(r_lo << s) = w_{2^Bm}^-(r_lo << b)
    }
    __syncthreads();
  }
  // scale + zero-interleave (in registers, then one barrier): position p holds coefficient k = bitrev_L(p); DIT stage 0 duplicates
  uint32_t v[(1 << 11) / NT > 0 ? (1 << 11) / NT : 1];
  int cnt = 0;
  for (uint32_t e = threadIdx.x; e < chunk; e += NT) {
    const uint32_t k = bitrev(base + e, L);
    v[cnt++] = bb::mont_mul(bb::mont_mul(lds[e], g_lo[k & 1023]), g_hi[k >> 10]);
  }
  __syncthreads();
  cnt = 0;
  for (uint32_t e = threadIdx.x; e < chunk; e += NT) { lds[2 * e] = v[cnt]; lds[2 * e + 1] = v[cnt]; cnt++; }
  __syncthreads();
  for (int s = 1; s <= Bm; s++) {                              // forward DIT stages 1..Bm of the size-2N transform
    const uint32_t half = 1u << s;
    for (uint32_t q = threadIdx.x; q < chunk; q += NT) {
      const uint32_t r_lo = q & (half - 1), r_hi = q >> s;
      const uint32_t ia = (r_hi << (s + 1)) | r_lo, ib = ia + half;
      const uint32_t a = lds[ia], t = bb::mont_mul(lds[ib], small_fwd[r_lo << (Bm - s)]);    // w_2N^(r_lo << (L-s)) = w_{2^(Bm+1)}^(r_lo << (Bm-s))
      lds[ia] = bb::add(a, t); lds[ib] = bb::sub(a, t);
    }
    __syncthreads();
  }
  for (uint32_t e = threadIdx.x; e < 2 * chunk; e += NT) y[2 * base + e] = lds[e];
}

// Same computation for Bm = 10, with register-resident radix-4 butterflies: every round does TWO radix-2 stages on 4 values held
// in registers, so the 10 inverse + 10 forward stages need 10 LDS round trips / barriers instead of 21, each lane has 4-8
// independent multiplications in flight, and only one twiddle per quad is read (the others are its square and its product with
// a 4th root of unity).  A = LDS[0,1024): inverse part; Bf = LDS[1024, 3072): forward part (zero-interleaved, stage 0 = copy).
__global__ __launch_bounds__(NT) void lde_middle_r4_kernel(const uint32_t* __restrict__ in, uint64_t in_stride, uint32_t* __restrict__ out, uint64_t out_stride, int L,
                                                            const uint32_t* __restrict__ small_inv, const uint32_t* __restrict__ small_fwd,
                                                            const uint32_t* __restrict__ g_lo, const uint32_t* __restrict__ g_hi, uint32_t j4_inv_m, uint32_t j4_fwd_m) {
  constexpr int Bm = 10;
  __shared__ uint32_t A[1024];
  __shared__ uint32_t Bf[2048];
  const uint32_t* x = in + (uint64_t)blockIdx.y * in_stride;
  uint32_t* y = out + (uint64_t)blockIdx.y * out_stride;
  const uint32_t base = blockIdx.x << Bm, q = threadIdx.x;
  {
    const uint4 v = reinterpret_cast<const uint4*>(x + base)[q];
    reinterpret_cast<uint4*>(A)[q] = v;
  }
  __syncthreads();
  // ---- inverse DIF, rounds r = 0..4: stages (2r, 2r+1), spans h1 = 2^(9-2r), h2 = h1/2 ----
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const int lg = 8 - 2 * r;                                  // log2(h2)
    const uint32_t h2 = 1u << lg, lo = q & (h2 - 1), hi = q >> lg;
    const uint32_t i0 = (hi << (lg + 2)) | lo;
    const uint32_t x0 = A[i0], x1 = A[i0 + h2], x2 = A[i0 + 2 * h2], x3 = A[i0 + 3 * h2];
    const uint32_t wA = small_inv[lo << (2 * r)];              // w_1024^-(lo << 2r)
    const uint32_t wB = bb::mont_mul(wA, j4_inv_m), w2 = bb::mont_mul(wA, wA);
    const uint32_t y0 = bb::add(x0, x2), y2 = bb::mont_mul(bb::sub(x0, x2), wA);
    const uint32_t y1 = bb::add(x1, x3), y3 = bb::mont_mul(bb::sub(x1, x3), wB);
    uint32_t z0 = bb::add(y0, y1), z1 = bb::mont_mul(bb::sub(y0, y1), w2);
    uint32_t z2 = bb::add(y2, y3), z3 = bb::mont_mul(bb::sub(y2, y3), w2);
    if (r == 4) {                                              // last round (positions 4q..4q+3): fold in the coset scale g^k / N, k = bitrev_L(position)
      const uint32_t p0 = base + i0;
      const uint32_t k0 = bitrev(p0, L), k1 = bitrev(p0 + 1, L), k2 = bitrev(p0 + 2, L), k3 = bitrev(p0 + 3, L);
      z0 = bb::mont_mul(bb::mont_mul(z0, g_lo[k0 & 1023]), g_hi[k0 >> 10]);
      z1 = bb::mont_mul(bb::mont_mul(z1, g_lo[k1 & 1023]), g_hi[k1 >> 10]);
      z2 = bb::mont_mul(bb::mont_mul(z2, g_lo[k2 & 1023]), g_hi[k2 >> 10]);
      z3 = bb::mont_mul(bb::mont_mul(z3, g_lo[k3 & 1023]), g_hi[k3 >> 10]);
    }
    A[i0] = z0; A[i0 + h2] = z1; A[i0 + 2 * h2] = z2; A[i0 + 3 * h2] = z3;
    __syncthreads();
  }
  // ---- forward DIT of the zero-interleaved chunk (2048 points): stage 0 is a copy, rounds do stages (s, s+1), s = 1,3,5,7,9 ----
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const int s = 2 * r + 1;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const uint32_t qq = q + t * NT;                          // 512 quads
      const uint32_t lo = qq & ((1u << s) - 1), hi = qq >> s;
      const uint32_t i0 = (hi << (s + 2)) | lo, d = 1u << s;
      uint32_t x0, x1, x2, x3;
      if (r == 0) { x0 = A[i0 >> 1]; x1 = A[(i0 + d) >> 1]; x2 = A[(i0 + 2 * d) >> 1]; x3 = A[(i0 + 3 * d) >> 1]; }   // after stage 0: Bf[j] = A[j >> 1]
      else { x0 = Bf[i0]; x1 = Bf[i0 + d]; x2 = Bf[i0 + 2 * d]; x3 = Bf[i0 + 3 * d]; }
      const uint32_t w2 = small_fwd[lo << (Bm - s - 1)];       // w_2048^(lo << (9-s)): twiddle of stage s+1
      const uint32_t w1 = bb::mont_mul(w2, w2), w2i = bb::mont_mul(w2, j4_fwd_m);
      const uint32_t t1 = bb::mont_mul(x1, w1), t3 = bb::mont_mul(x3, w1);
      const uint32_t y0 = bb::add(x0, t1), y1 = bb::sub(x0, t1), y2 = bb::add(x2, t3), y3 = bb::sub(x2, t3);
      const uint32_t u2 = bb::mont_mul(y2, w2), u3 = bb::mont_mul(y3, w2i);
      Bf[i0] = bb::add(y0, u2); Bf[i0 + 2 * d] = bb::sub(y0, u2); Bf[i0 + d] = bb::add(y1, u3); Bf[i0 + 3 * d] = bb::sub(y1, u3);
    }
    __syncthreads();
  }
  {
    uint4* dst = reinterpret_cast<uint4*>(y + 2 * base);
    dst[q] = reinterpret_cast<const uint4*>(Bf)[q];
    dst[q + NT] = reinterpret_cast<const uint4*>(Bf)[q + NT];
  }
}

// ------------------------------------------------------------------------------------------------
// Poseidon2 Merkle
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void leaf_hash_kernel(const uint32_t* __restrict__ mat, uint32_t width, uint64_t n, uint64_t col_stride, uint32_t* __restrict__ digests) {
  const uint64_t j = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (j >= n) return;
  uint32_t s[p2::T];
#pragma unroll
  for (int i = 0; i < p2::T; i++) s[i] = 0;
  for (uint32_t off = 0; off < width; off += p2::RATE) {
#pragma unroll
    for (int i = 0; i < p2::RATE; i++)
      if (off + i < width) s[i] = bb::to_mont(mat[(uint64_t)(off + i) * col_stride + j]);
    p2::permute(s, d_p2);
  }
  if (width == 0) p2::permute(s, d_p2);
  uint4 d = make_uint4(bb::from_mont(s[0]), bb::from_mont(s[1]), bb::from_mont(s[2]), bb::from_mont(s[3]));
  reinterpret_cast<uint4*>(digests)[j] = d;
}

__global__ __launch_bounds__(NT) void compress_kernel(const uint32_t* __restrict__ in, uint64_t n_out, uint32_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n_out) return;
  const uint4 l = reinterpret_cast<const uint4*>(in)[2 * i], r = reinterpret_cast<const uint4*>(in)[2 * i + 1];
  uint32_t s[p2::T] = {bb::to_mont(l.x), bb::to_mont(l.y), bb::to_mont(l.z), bb::to_mont(l.w), bb::to_mont(r.x), bb::to_mont(r.y), bb::to_mont(r.z), bb::to_mont(r.w), 0, 0, 0, 0};
  p2::permute(s, d_p2);
  reinterpret_cast<uint4*>(out)[i] = make_uint4(bb::from_mont(s[0]), bb::from_mont(s[1]), bb::from_mont(s[2]), bb::from_mont(s[3]));
}

// Upper part of a Merkle tree in ONE launch: a single workgroup walks the levels from `m` digests down to the root
// (launch + tail latency of ~12 tiny kernels costs more than the hashing itself).  cur = level of m digests inside the tree buffer.
__global__ __launch_bounds__(NT) void compress_tail_kernel(uint32_t* __restrict__ cur, uint32_t m) {
  while (m > 1) {
    uint32_t* nxt = cur + 4 * (uint64_t)m;
    for (uint32_t i = threadIdx.x; i < m / 2; i += NT) {
      const uint4 l = reinterpret_cast<const uint4*>(cur)[2 * i], r = reinterpret_cast<const uint4*>(cur)[2 * i + 1];
      uint32_t s[p2::T] = {bb::to_mont(l.x), bb::to_mont(l.y), bb::to_mont(l.z), bb::to_mont(l.w), bb::to_mont(r.x), bb::to_mont(r.y), bb::to_mont(r.z), bb::to_mont(r.w), 0, 0, 0, 0};
      p2::permute(s, d_p2);
      reinterpret_cast<uint4*>(nxt)[i] = make_uint4(bb::from_mont(s[0]), bb::from_mont(s[1]), bb::from_mont(s[2]), bb::from_mont(s[3]));
    }
    __threadfence_block();
    __syncthreads();
    cur = nxt; m >>= 1;
  }
}

// ALU roofline probe: every lane runs `iters` rounds of 8 independent Montgomery multiplications (no memory traffic), i.e. the
// best modmul rate this formulation of mont_mul can reach on the chip.  The Poseidon2 kernels are priced against it.
__global__ __launch_bounds__(NT) void modmul_peak_kernel(uint32_t* __restrict__ out, uint32_t iters) {
  uint32_t a[8];
#pragma unroll
  for (int k = 0; k < 8; k++) a[k] = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + k * 7919u) % bb::P;
  for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = bb::mont_mul(a[k], a[(k + 1) & 7]);
  }
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) r ^= a[k];
  if (r == 0xFFFFFFFFu) out[0] = r;                     // never true (values < p); keeps the chain live
}

// all levels above the leaf digests: wide levels one launch each, the last <= 2048 digests in a single launch
void launch_tree_levels(uint32_t* leaf_digests, uint64_t n_leaves, hipStream_t s) {
  uint32_t* cur = leaf_digests;
  uint64_t m = n_leaves;
  for (; m > 2048; m >>= 1) {
    uint32_t* nxt = cur + 4 * m;
    hipLaunchKernelGGL(compress_kernel, dim3(grid_for(m / 2)), dim3(NT), 0, s, cur, m / 2, nxt);
    cur = nxt;
  }
  if (m > 1) hipLaunchKernelGGL(compress_tail_kernel, dim3(1), dim3(NT), 0, s, cur, (uint32_t)m);
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
  uint32_t* d_g_hi = nullptr;     // g^(1024 k)
  uint32_t* d_small_inv = nullptr;  // w_{2^Bm}^-k, k < 2^(Bm-1)      (Bm = min(log_n, 10))
  uint32_t* d_small_fwd = nullptr;  // w_{2^(Bm+1)}^k, k < 2^Bm
  p2::Consts consts;
  // prover workspace: one device allocation made on the first zkir_prove and reused (hipMalloc of GBs costs more than the kernels)
  mutable unsigned char* arena = nullptr;
  mutable size_t arena_size = 0, arena_off = 0;
};

extern "C" {

uint32_t zkir_main_trace_width(void) { return 89; }

// Diagnostic: measured peak Montgomery-multiplication rate (modmul/s) of the device, used as the ALU roofline of the Poseidon2 kernels.
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
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(d_p2), &c->consts, sizeof(p2::Consts));
  if (e == hipSuccess) e = hipMalloc(&c->d_tw_inv, (size_t)n_inv * 4);
  if (e == hipSuccess) e = hipMalloc(&c->d_tw_fwd, (size_t)N * 4);
  if (e == hipSuccess) e = hipMalloc(&c->d_g_lo, 1024 * 4);
  if (e == hipSuccess) e = hipMalloc(&c->d_g_hi, (size_t)n_hi * 4);
  const int Bm = log_n < 10 ? (int)log_n : 10;
  const uint32_t n_si = Bm >= 1 ? (1u << (Bm - 1)) : 1, n_sf = 1u << Bm;
  if (e == hipSuccess) e = hipMalloc(&c->d_small_inv, (size_t)n_si * 4);
  if (e == hipSuccess) e = hipMalloc(&c->d_small_fwd, (size_t)n_sf * 4);
  if (e != hipSuccess) { zkir::set_last_error({ZKIR_ERR_DEVICE, std::string("zkir_stark_ctx_create: ") + hipGetErrorString(e)}); zkir_stark_ctx_free(c); return ZKIR_ERR_DEVICE; }
  const uint32_t wN = bb::root_of_unity(log_n), w2N = bb::root_of_unity(log_n + 1);
  hipLaunchKernelGGL(powers_kernel, dim3(grid_for(n_inv)), dim3(NT), 0, 0, bb::inv(wN), bb::R1, c->d_tw_inv, n_inv);
  hipLaunchKernelGGL(powers_kernel, dim3(grid_for(N)), dim3(NT), 0, 0, w2N, bb::R1, c->d_tw_fwd, N);
  hipLaunchKernelGGL(powers_kernel, dim3(4), dim3(NT), 0, 0, bb::GEN, bb::to_mont(bb::inv(N % bb::P)), c->d_g_lo, 1024u);
  hipLaunchKernelGGL(powers_kernel, dim3(grid_for(n_hi)), dim3(NT), 0, 0, bb::pow(bb::GEN, 1024), bb::R1, c->d_g_hi, n_hi);
  hipLaunchKernelGGL(powers_kernel, dim3(grid_for(n_si)), dim3(NT), 0, 0, bb::inv(bb::root_of_unity(Bm)), bb::R1, c->d_small_inv, n_si);
  hipLaunchKernelGGL(powers_kernel, dim3(grid_for(n_sf)), dim3(NT), 0, 0, bb::root_of_unity(Bm + 1), bb::R1, c->d_small_fwd, n_sf);
  if (hipDeviceSynchronize() != hipSuccess || check_launch("stark ctx tables") != ZKIR_OK) { zkir_stark_ctx_free(c); return ZKIR_ERR_DEVICE; }
  *out = c;
  return ZKIR_OK;
}

void zkir_stark_ctx_free(zkir_stark_ctx* c) {
  if (!c) return;
  (void)hipFree(c->d_tw_inv); (void)hipFree(c->d_tw_fwd); (void)hipFree(c->d_g_lo); (void)hipFree(c->d_g_hi);
  (void)hipFree(c->d_small_inv); (void)hipFree(c->d_small_fwd);
  if (c->arena) (void)hipFree(c->arena);
  delete c;
}

int zkir_main_trace_launch(const zkir_trace_columns* trace, uint64_t n_rows, uint32_t* out, void* stream) {
  if (n_rows == 0) return ZKIR_OK;
  hipLaunchKernelGGL(main_trace_kernel, dim3(grid_for(n_rows)), dim3(NT), 0, (hipStream_t)stream, *trace, n_rows, out);
  return check_launch("main_trace");
}

// in: [width][N] canonical evaluations over H (natural order; used as scratch and overwritten!), out: [width][2N]
int zkir_lde_launch(const zkir_stark_ctx* c, uint32_t* in, uint32_t width, uint32_t* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int L = (int)c->log_n;
  const uint32_t N = 1u << L;
  const int Bm = L < 10 ? L : 10;
  static const uint32_t j4_inv_m = bb::to_mont(bb::inv(bb::root_of_unity(2))), j4_fwd_m = bb::to_mont(bb::root_of_unity(2));
  // inverse DIF strided stages 0 .. L-Bm-1 (only when L > 10; the compact table then has order 2^Bm = 1024)
  run_strided_stages<false>(in, N, width, L, 0, L - Bm, c->d_tw_inv, c->d_small_inv, Bm, j4_inv_m, s);
  if (Bm == 10) {
    hipLaunchKernelGGL(lde_middle_r4_kernel, dim3(N >> Bm, width), dim3(NT), 0, s, in, (uint64_t)N, out, (uint64_t)2 * N, L, c->d_small_inv, c->d_small_fwd, c->d_g_lo,
                       c->d_g_hi, j4_inv_m, j4_fwd_m);
  } else {
    hipLaunchKernelGGL(lde_middle_kernel, dim3(N >> Bm, width), dim3(NT), (8u << Bm), s, in, (uint64_t)N, out, (uint64_t)2 * N, L, Bm, c->d_small_inv, c->d_small_fwd, c->d_g_lo, c->d_g_hi);
  }
  // forward DIT strided stages Bm+1 .. L of the size-2N transform
  run_strided_stages<true>(out, (uint64_t)2 * N, width, L + 1, Bm + 1, L - Bm, c->d_tw_fwd, c->d_small_fwd, Bm + 1, j4_fwd_m, s);
  return check_launch("lde");
}

// tree = [leaf digests (4*n)] [layer 1 (4*n/2)] ... [root (4)]  = 4*(2n-1) words; n_leaves a power of two
int zkir_merkle_commit_launch(const zkir_stark_ctx* c, const uint32_t* mat, uint32_t width, uint64_t n_leaves, uint32_t* tree, void* stream) {
  (void)c;
  hipStream_t s = (hipStream_t)stream;
  if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "merkle: n_leaves must be a power of two"}); return ZKIR_ERR_ARGUMENT; }
  hipLaunchKernelGGL(leaf_hash_kernel, dim3(grid_for(n_leaves)), dim3(NT), 0, s, mat, width, n_leaves, n_leaves, tree);
  launch_tree_levels(tree, n_leaves, s);
  return check_launch("merkle_commit");
}

// Upper levels over already-computed digests (multi-GPU: the all-gathered subtree roots of the row shards are the leaves of the
// top log2(G) levels).  tree[0 .. 4n) must hold the n digests; the call fills the remaining 4(n-1) words, root = last 4.
int zkir_merkle_cap_launch(const zkir_stark_ctx* c, uint32_t* tree, uint64_t n_digests, void* stream) {
  (void)c;
  if (n_digests == 0 || (n_digests & (n_digests - 1))) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "merkle cap: n_digests must be a power of two"}); return ZKIR_ERR_ARGUMENT; }
  launch_tree_levels(tree, n_digests, (hipStream_t)stream);
  return check_launch("merkle_cap");
}

}  // extern "C"

#include "stark_prove.inl"
