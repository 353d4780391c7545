// ntt.hip — Baby Bear NTT passes and the coset low-degree extension on gfx950 (part of the self-defined prover stages, DESIGN.md §8;
// oracle: so::lde / so::ntt in oracle/stark_oracle.cpp).  Split from stark.hip because the two want different instruction
// scheduling: these kernels are LDS/register-pressure sensitive (default scheduler), the hash kernels want maximum ILP.
//
// Per column: inverse DIF NTT over H (natural -> bit-reversed), coset scale g^k / N, zero-interleave, forward DIT NTT over the
// 2N coset (bit-reversed -> natural).  LDS-staged passes: strided passes move tiles of 2^B x 2^C elements (2^C consecutive
// words per row of the tile keep loads coalesced); the last B_m inverse stages, the scaling and the first B_m+1 forward stages
// are fused in one contiguous-tile kernel.  8 B/element of HBM traffic per strided pass, 12 B/element for the fused middle.
#include <hip/hip_runtime.h>

#include "../../include/zkir_amd.h"
#include "babybear.h"
#include "host.h"

namespace {

constexpr int NT = 256;
__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return bits == 0 ? 0u : __brev(x) >> (32 - bits); }

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

// global <-> LDS move of a tile of ELEMS words whose rows (2^C consecutive words) are `stride` words apart, VEC words per lane
template <bool LOAD, int VEC, int C, uint32_t ELEMS, int NTH>
__device__ __forceinline__ void tile_move(uint32_t* __restrict__ g, uint32_t stride, uint32_t* __restrict__ lds) {
  constexpr uint32_t ROW = (1u << C) / VEC;                    // lanes per row
#pragma unroll
  for (uint32_t k = 0; k < ELEMS / VEC / NTH; k++) {
    const uint32_t e = threadIdx.x + k * NTH;
    uint32_t* gp = g + (e / ROW) * stride + (e % ROW) * VEC;
    if (VEC == 4) { if (LOAD) reinterpret_cast<uint4*>(lds)[e] = *reinterpret_cast<const uint4*>(gp); else *reinterpret_cast<uint4*>(gp) = reinterpret_cast<const uint4*>(lds)[e]; }
    else if (VEC == 2) { if (LOAD) reinterpret_cast<uint2*>(lds)[e] = *reinterpret_cast<const uint2*>(gp); else *reinterpret_cast<uint2*>(gp) = reinterpret_cast<const uint2*>(lds)[e]; }
    else { if (LOAD) lds[e] = *gp; else *gp = lds[e]; }
  }
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
  // Tile rows are 2^C consecutive words.  The inverse pass (rows 4 KiB apart at 2^20) moves them as uint4 — a quarter of the
  // memory instructions, measured 360 -> 301 us; the forward pass (rows 8 KiB apart) got SLOWER with wide or unrolled moves
  // (539 -> 610-675 us: bursts of requests on the same power-of-two stride), so it keeps a rolled loop of 4-byte moves.
  constexpr int VEC = DIT ? 1 : 4;
  static_assert(C >= 2 && (ELEMS / VEC) % NTH == 0, "tile moves");
  if (DIT) { for (uint32_t e = threadIdx.x; e < ELEMS; e += NTH) lds[e] = x[base + (e >> C) * stride_mid + (e & CMASK)]; }
  else tile_move<true, VEC, C, ELEMS, NTH>(x + base, stride_mid, lds);
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
        const uint32_t y0 = bb::add(x0, x2), y2 = bb::mont_mul(bb::sub_lazy(x0, x2), wA);      // differences only feed a product: no reduction
        const uint32_t y1 = bb::add(x1, x3), y3 = bb::mont_mul(bb::sub_lazy(x1, x3), wB);
        lds[i0] = bb::add(y0, y1); lds[i0 + d] = bb::mont_mul(bb::sub_lazy(y0, y1), w2);
        lds[i0 + 2 * d] = bb::add(y2, y3); lds[i0 + 3 * d] = bb::mont_mul(bb::sub_lazy(y2, y3), w2);
      } else {
        const uint32_t dm = 1u << b, mid_lo = qq & (dm - 1), mid_hi = qq >> b;
        const uint32_t i0 = ((((mid_hi << (b + 2)) | mid_lo)) << C) | lo_l, d = dm << C;
        const uint32_t x0 = lds[i0], x1 = lds[i0 + d], x2 = lds[i0 + 2 * d], x3 = lds[i0 + 3 * d];
        const uint32_t w2 = bb::mont_mul(tp[r], small[mid_lo << (log_small - (b + 2))]);
        const uint32_t w1 = bb::mont_mul(w2, w2), w2i = bb::mont_mul(w2, j4_m);
        const uint32_t t1 = bb::mont_mul(x1, w1), t3 = bb::mont_mul(x3, w1);
        const uint32_t y0 = bb::add(x0, t1), y1 = bb::sub(x0, t1), y2 = bb::add_lazy(x2, t3), y3 = bb::sub_lazy(x2, t3);
        const uint32_t u2 = bb::mont_mul(y2, w2), u3 = bb::mont_mul(y3, w2i);
        lds[i0] = bb::add(y0, u2); lds[i0 + 2 * d] = bb::sub(y0, u2); lds[i0 + d] = bb::add(y1, u3); lds[i0 + 3 * d] = bb::sub(y1, u3);
      }
    }
    __syncthreads();
  }
  if (DIT) { for (uint32_t e = threadIdx.x; e < ELEMS; e += NTH) x[base + (e >> C) * stride_mid + (e & CMASK)] = lds[e]; }
  else tile_move<false, VEC, C, ELEMS, NTH>(x + base, stride_mid, lds);
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
// in registers, so the 10 inverse + 10 forward stages need 10 LDS round trips / barriers instead of 21, and only one twiddle per
// quad is read (the others are its square and its product with a 4th root of unity).  A workgroup works on the same chunk of NC
// columns: the kernel is VALU-bound (rocprofv3: VALUBusy 92 % with one column), and the index arithmetic, twiddles and coset
// scale factors are the same for every column, so they are computed once per quad and reused NC times.
// A = inverse part (1024 words per column); Bf = forward part (2048 words per column, zero-interleaved, stage 0 = copy).
template <int NC>
__global__ __launch_bounds__(NT) void lde_middle_r4_kernel(const uint32_t* __restrict__ in, uint64_t in_stride, uint32_t* __restrict__ out, uint64_t out_stride,
                                                            uint32_t width, int L, const uint32_t* __restrict__ small_inv, const uint32_t* __restrict__ small_fwd,
                                                            const uint32_t* __restrict__ g_lo, const uint32_t* __restrict__ g_hi, uint32_t j4_inv_m, uint32_t j4_fwd_m) {
  constexpr int Bm = 10;
  __shared__ uint32_t A[NC][1024];
  __shared__ uint32_t Bf[NC][2048];
  const uint32_t col0 = blockIdx.y * NC;
  const uint32_t base = blockIdx.x << Bm, q = threadIdx.x;
#pragma unroll
  for (int c = 0; c < NC; c++)
    if (col0 + c < width) reinterpret_cast<uint4*>(A[c])[q] = reinterpret_cast<const uint4*>(in + (uint64_t)(col0 + c) * in_stride + base)[q];
  __syncthreads();
  // ---- inverse DIF, rounds r = 0..4: stages (2r, 2r+1), spans h1 = 2^(9-2r), h2 = h1/2 ----
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const int lg = 8 - 2 * r;                                  // log2(h2)
    const uint32_t h2 = 1u << lg, lo = q & (h2 - 1), hi = q >> lg;
    const uint32_t i0 = (hi << (lg + 2)) | lo;
    const uint32_t wA = small_inv[lo << (2 * r)];              // w_1024^-(lo << 2r)
    // (reading wB = table[e + 256] and w2 = table[2e] instead of computing them was tried: 610 -> 747 us, the loads cost more)
    const uint32_t wB = bb::mont_mul(wA, j4_inv_m), w2 = bb::mont_mul(wA, wA);
    uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0;
    if (r == 4) {                                              // last round (positions 4q..4q+3): the coset scale g^k / N, k = bitrev_L(position)
      const uint32_t p0 = base + i0;
      const uint32_t k0 = bitrev(p0, L), k1 = bitrev(p0 + 1, L), k2 = bitrev(p0 + 2, L), k3 = bitrev(p0 + 3, L);
      g0 = bb::mont_mul(g_lo[k0 & 1023], g_hi[k0 >> 10]); g1 = bb::mont_mul(g_lo[k1 & 1023], g_hi[k1 >> 10]);
      g2 = bb::mont_mul(g_lo[k2 & 1023], g_hi[k2 >> 10]); g3 = bb::mont_mul(g_lo[k3 & 1023], g_hi[k3 >> 10]);
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
      uint32_t* a = A[c];
      const uint32_t x0 = a[i0], x1 = a[i0 + h2], x2 = a[i0 + 2 * h2], x3 = a[i0 + 3 * h2];
      const uint32_t y0 = bb::add(x0, x2), y2 = bb::mont_mul(bb::sub_lazy(x0, x2), wA);
      const uint32_t y1 = bb::add(x1, x3), y3 = bb::mont_mul(bb::sub_lazy(x1, x3), wB);
      uint32_t z0 = bb::add(y0, y1), z1 = bb::mont_mul(bb::sub_lazy(y0, y1), w2);
      uint32_t z2 = bb::add(y2, y3), z3 = bb::mont_mul(bb::sub_lazy(y2, y3), w2);
      if (r == 4) { z0 = bb::mont_mul(z0, g0); z1 = bb::mont_mul(z1, g1); z2 = bb::mont_mul(z2, g2); z3 = bb::mont_mul(z3, g3); }
      a[i0] = z0; a[i0 + h2] = z1; a[i0 + 2 * h2] = z2; a[i0 + 3 * h2] = z3;
    }
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
      const uint32_t w2 = small_fwd[lo << (Bm - s - 1)];       // w_2048^(lo << (9-s)): twiddle of stage s+1
      const uint32_t w1 = bb::mont_mul(w2, w2), w2i = bb::mont_mul(w2, j4_fwd_m);
#pragma unroll
      for (int c = 0; c < NC; c++) {
        uint32_t* bf = Bf[c];
        uint32_t x0, x1, x2, x3;
        if (r == 0) { const uint32_t* a = A[c]; x0 = a[i0 >> 1]; x1 = a[(i0 + d) >> 1]; x2 = a[(i0 + 2 * d) >> 1]; x3 = a[(i0 + 3 * d) >> 1]; }   // after stage 0: Bf[j] = A[j >> 1]
        else { x0 = bf[i0]; x1 = bf[i0 + d]; x2 = bf[i0 + 2 * d]; x3 = bf[i0 + 3 * d]; }
        const uint32_t t1 = bb::mont_mul(x1, w1), t3 = bb::mont_mul(x3, w1);
        const uint32_t y0 = bb::add(x0, t1), y1 = bb::sub(x0, t1), y2 = bb::add_lazy(x2, t3), y3 = bb::sub_lazy(x2, t3);
        const uint32_t u2 = bb::mont_mul(y2, w2), u3 = bb::mont_mul(y3, w2i);
        bf[i0] = bb::add(y0, u2); bf[i0 + 2 * d] = bb::sub(y0, u2); bf[i0 + d] = bb::add(y1, u3); bf[i0 + 3 * d] = bb::sub(y1, u3);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < NC; c++) {
    if (col0 + c >= width) break;
    uint4* dst = reinterpret_cast<uint4*>(out + (uint64_t)(col0 + c) * out_stride + 2 * base);
    dst[q] = reinterpret_cast<const uint4*>(Bf[c])[q];
    dst[q + NT] = reinterpret_cast<const uint4*>(Bf[c])[q + NT];
  }
}

}  // namespace

namespace zkir {

// in: [width][N] canonical evaluations over H (natural order; used as scratch and overwritten!), out: [width][2N]
void lde_run(const LdeTables& t, uint32_t* in, uint32_t width, uint32_t* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int L = t.log_n;
  const uint32_t N = 1u << L;
  const int Bm = L < 10 ? L : 10;
  static const uint32_t j4_inv_m = bb::to_mont(bb::inv(bb::root_of_unity(2))), j4_fwd_m = bb::to_mont(bb::root_of_unity(2));
  // inverse DIF strided stages 0 .. L-Bm-1 (only when L > 10; the compact table then has order 2^Bm = 1024)
  run_strided_stages<false>(in, N, width, L, 0, L - Bm, t.tw_inv, t.small_inv, Bm, j4_inv_m, s);
  if (Bm == 10) {
    constexpr int NC = 1;
    hipLaunchKernelGGL(lde_middle_r4_kernel<NC>, dim3(N >> Bm, (width + NC - 1) / NC), dim3(NT), 0, s, in, (uint64_t)N, out, (uint64_t)2 * N, width, L, t.small_inv, t.small_fwd,
                       t.g_lo, t.g_hi, j4_inv_m, j4_fwd_m);
  } else {
    hipLaunchKernelGGL(lde_middle_kernel, dim3(N >> Bm, width), dim3(NT), (8u << Bm), s, in, (uint64_t)N, out, (uint64_t)2 * N, L, Bm, t.small_inv, t.small_fwd, t.g_lo, t.g_hi);
  }
  // forward DIT strided stages Bm+1 .. L of the size-2N transform
  run_strided_stages<true>(out, (uint64_t)2 * N, width, L + 1, Bm + 1, L - Bm, t.tw_fwd, t.small_fwd, Bm + 1, j4_fwd_m, s);
}

}  // namespace zkir
