// witness.hip — expansion of the host interpreter's side logs into the witness columns the reference's
// ExecutionResult carries (gfx950).  All kernels are elementwise / small-segment integer work, HBM-bound,
// written as coalesced SoA producers; none has a contraction (no MFMA).
//
//   memops_*      MemoryOp columns in row order + CSR row offsets (TraceRow.memory_ops, zkir-spec/src/trace.rs:149-167,
//                 bound = TypeWidth(8*width) zkir-runtime/src/memory.rs:245) and ExecutionResult::get_memory_trace()
//                 (zkir-runtime/src/vm.rs:85-94: stable sort by timestamp, address, Read<Write; Ord at trace.rs:210-223)
//   range_check   RangeCheckWitness entries (value, chunks[4], pc) zkir-runtime/src/range_check.rs:175-192,212 plus the
//                 lookup-table multiplicities of the chunks (table size 2^chunk_bits, config.rs:78-80)
//   norm          NormalizationEvent columns, zkir-runtime/src/normalization_witness.rs:19-43, normalize.rs:133-153
//   sha256_chip   Sha256Witness per single-block message, zkir-runtime/src/crypto.rs:142-207,223-297; trace.rs:236-256
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/zkir_amd.h"
#include "host.h"

namespace {

constexpr int NT = 256;
inline unsigned grid_for(uint64_t n, int per_block = NT) { return (unsigned)((n + per_block - 1) / per_block); }

// ------------------------------------------------------------------------------------------------
// memory ops
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_memop(const zkir_memop_columns& c, uint64_t at, const zkir_mem_event& e, uint64_t cycle_base) {
  c.address[at] = e.address;
  c.value[at] = e.value;
  c.timestamp[at] = cycle_base + e.row;
  c.is_write[at] = e.is_write;
  c.width[at] = e.width;
  c.bound_bits[at] = 8u * e.width;                 // ValueBound::from_type_width(width * 8)
  c.bound_tag[at] = ZKIR_BOUND_TYPE_WIDTH;
  c.bound_payload[at] = 8ull * e.width;
}

__global__ __launch_bounds__(NT) void memops_expand_kernel(const zkir_mem_event* __restrict__ ev, uint64_t n, uint64_t cycle_base, zkir_memop_columns c) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i < n) put_memop(c, i, ev[i], cycle_base);
}

// The same expansion in ONE pass over the events that also produces the CSR row offsets and the per-row shape flags of the sort:
// ops are ordered by row, so op i is the first op of its row iff ev[i-1].row != ev[i].row, and then offsets[r] = i for every row r in
// (ev[i-1].row, ev[i].row] (rows without ops in between included); the op after the last one closes the table with offsets = n for
// the remaining rows.  A lane fills short gaps itself; long ones (a program that touches memory rarely) are queued in LDS and filled by
// the whole workgroup, so no lane ever loops over more than GAP_INLINE rows alone.  The shape flag (memops_segment_check_kernel's
// test) needs the same neighbour.  Replaces a binary search per ROW over the event array (12x its algorithmic bytes in HBM traffic,
// profiles/r02k_config4_pmc_traffic.txt) and a separate check pass that re-read all events.
constexpr uint32_t GAP_INLINE = 32, GAP_QUEUE = 2 * NT;
__global__ __launch_bounds__(NT) void memops_expand_csr_kernel(const zkir_mem_event* __restrict__ ev, uint64_t n, uint64_t n_rows, uint64_t cycle_base, zkir_memop_columns c,
                                                                uint64_t* __restrict__ offsets, unsigned char* __restrict__ seg_bad) {
  __shared__ uint64_t q_lo[GAP_QUEUE], q_hi[GAP_QUEUE], q_val[GAP_QUEUE];
  __shared__ uint32_t q_n;
  if (threadIdx.x == 0) q_n = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  auto fill = [&](uint64_t lo, uint64_t hi, uint64_t val) {      // offsets[lo..hi] = val (inclusive)
    if (hi < lo) return;
    if (hi - lo < GAP_INLINE) { for (uint64_t r = lo; r <= hi; r++) offsets[r] = val; return; }
    const uint32_t k = atomicAdd(&q_n, 1u);                       // at most two gaps per lane (the last op also closes the table)
    q_lo[k] = lo; q_hi[k] = hi; q_val[k] = val;
  };
  // the neighbour's key comes from the previous LANE (one shuffle each); only lane 0 of a wave reads ev[i - 1] itself
  zkir_mem_event e{};
  if (i < n) e = ev[i];
  const int lane = threadIdx.x & 63;
  uint32_t p_row = __shfl_up(e.row, 1, 64);
  uint32_t p_wr = __shfl_up((uint32_t)e.is_write, 1, 64);
  uint64_t p_addr = __shfl_up(e.address, 1, 64);
  if (i < n) {
    put_memop(c, i, e, cycle_base);
    if (i == 0) fill(0, e.row, 0);
    else {
      if (lane == 0) { const zkir_mem_event p = ev[i - 1]; p_row = p.row; p_wr = p.is_write; p_addr = p.address; }
      if (p_row != e.row) fill((uint64_t)p_row + 1, e.row, i);
      else if ((e.is_write < p_wr) || (e.is_write == p_wr && e.address < p_addr)) seg_bad[e.row] = 1;
    }
    if (i == n - 1) fill((uint64_t)e.row + 1, n_rows, n);
  }
  __syncthreads();
  const uint32_t qn = q_n;
  for (uint32_t k = 0; k < qn; k++)
    for (uint64_t r = q_lo[k] + threadIdx.x; r <= q_hi[k]; r += NT) offsets[r] = q_val[k];
}

// offsets[r] = number of ops with row < r  (ops are ordered by row); one thread per row, binary search
__global__ __launch_bounds__(NT) void memops_row_offsets_kernel(const zkir_mem_event* __restrict__ ev, uint64_t n, uint64_t n_rows, uint64_t* __restrict__ offsets) {
  const uint64_t r = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (r > n_rows) return;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if ((uint64_t)ev[mid].row < r) lo = mid + 1; else hi = mid;
  }
  offsets[r] = lo;
}

// A row's ops come from one instruction: a single load/store, or a hash syscall = [byte reads, ascending][writes, ascending].
// Segments that keep that shape are merged by rank; any other shape (e.g. address wrap-around) is flagged here and
// sorted by an always-correct counting rank.
__global__ __launch_bounds__(NT) void memops_segment_check_kernel(const zkir_mem_event* __restrict__ ev, uint64_t n, unsigned char* __restrict__ seg_bad) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i == 0 || i >= n) return;
  const zkir_mem_event a = ev[i - 1], b = ev[i];
  if (a.row != b.row) return;
  const bool bad = (b.is_write < a.is_write) || (b.is_write == a.is_write && b.address < a.address);
  if (bad) seg_bad[b.row] = 1;
}

__device__ __forceinline__ bool memop_before(const zkir_mem_event& a, uint64_t ia, const zkir_mem_event& b, uint64_t ib) {
  if (a.address != b.address) return a.address < b.address;     // same row: timestamp ties
  if (a.is_write != b.is_write) return a.is_write < b.is_write;
  return ia < ib;                                                // stable
}

// One lane per op.  The rank of an op inside its row's segment takes three binary searches over the segment (first write; position
// of the op's address in the other run): ~11 DEPENDENT loads.  Done on global memory that was the whole cost of the kernel (0.8 ms
// against 0.3 ms for the same bytes in row order).  A workgroup therefore first copies the keys (address, is_write) of every segment
// its 256 ops touch into LDS — ops of a row are contiguous, so that is its own 256 ops plus the rest of the first and last segment —
// and searches there; a range that does not fit (a hash over kilobytes of input) keeps the global path.
constexpr uint32_t SORT_CAP = 2048;
struct GlobalKeys {
  const zkir_mem_event* ev;
  __device__ __forceinline__ uint64_t address(uint64_t i) const { return ev[i].address; }
  __device__ __forceinline__ bool is_write(uint64_t i) const { return ev[i].is_write != 0; }
};
struct LdsKeys {
  const uint64_t* addr; const unsigned char* wr; uint64_t base;
  __device__ __forceinline__ uint64_t address(uint64_t i) const { return addr[i - base]; }
  __device__ __forceinline__ bool is_write(uint64_t i) const { return wr[i - base] != 0; }
};
template <class Keys>
__device__ __forceinline__ uint64_t merge_rank(const Keys& k, uint64_t i, uint64_t s, uint64_t t, uint64_t address, bool is_write) {
  uint64_t lo = s, hi = t;                                       // first write in the segment
  while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (k.is_write(mid)) hi = mid; else lo = mid + 1; }
  const uint64_t w0 = lo;
  if (!is_write) {                                               // reads precede writes at equal address: count writes strictly below
    uint64_t a = w0, b = t;
    while (a < b) { const uint64_t mid = (a + b) >> 1; if (k.address(mid) < address) a = mid + 1; else b = mid; }
    return (i - s) + (a - w0);
  }
  uint64_t a = s, b = w0;                                        // count reads at or below
  while (a < b) { const uint64_t mid = (a + b) >> 1; if (k.address(mid) <= address) a = mid + 1; else b = mid; }
  return (i - w0) + (a - s);
}

__global__ __launch_bounds__(NT) void memops_sort_kernel(const zkir_mem_event* __restrict__ ev, uint64_t n, uint64_t cycle_base,
                                                          const uint64_t* __restrict__ offsets, const unsigned char* __restrict__ seg_bad, zkir_memop_columns c) {
  __shared__ uint64_t s_addr[SORT_CAP];
  __shared__ unsigned char s_wr[SORT_CAP];
  const uint64_t b0 = (uint64_t)blockIdx.x * NT, b1 = b0 + NT < n ? b0 + NT : n;        // this workgroup's ops
  const uint64_t lo = offsets[ev[b0].row], hi = offsets[(uint64_t)ev[b1 - 1].row + 1];    // every segment they belong to (uniform)
  const bool in_lds = hi - lo <= SORT_CAP;
  if (in_lds) {
    for (uint64_t j = lo + threadIdx.x; j < hi; j += NT) { const zkir_mem_event e = ev[j]; s_addr[j - lo] = e.address; s_wr[j - lo] = e.is_write; }
    __syncthreads();
  }
  const uint64_t i = b0 + threadIdx.x;
  if (i >= n) return;
  const zkir_mem_event e = ev[i];
  const uint64_t s = offsets[e.row], t = offsets[(uint64_t)e.row + 1];
  uint64_t rank;
  if (t - s == 1) {
    rank = 0;
  } else if (!seg_bad[e.row]) {
    rank = in_lds ? merge_rank(LdsKeys{s_addr, s_wr, lo}, i, s, t, e.address, e.is_write != 0) : merge_rank(GlobalKeys{ev}, i, s, t, e.address, e.is_write != 0);
  } else {
    rank = 0;
    for (uint64_t j = s; j < t; j++) rank += memop_before(ev[j], j, e, i) ? 1 : 0;
  }
  put_memop(c, s + rank, e, cycle_base);
}

// ------------------------------------------------------------------------------------------------
// range checks
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void range_check_kernel(const zkir_rc_event* __restrict__ ev, uint64_t n, uint32_t chunk_bits, uint64_t* __restrict__ value,
                                                          uint64_t* __restrict__ pc, uint16_t* __restrict__ chunks, uint64_t stride, uint32_t* __restrict__ mult) {
  extern __shared__ uint32_t hist[];                             // privatised multiplicity table
  const uint32_t table = 1u << chunk_bits, mask = table - 1;
  if (mult) { for (uint32_t k = threadIdx.x; k < table; k += NT) hist[k] = 0; __syncthreads(); }
  for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (uint64_t)gridDim.x * NT) {
    const zkir_rc_event e = ev[i];
    value[i] = e.value; pc[i] = e.pc;
    const uint32_t l0 = (uint32_t)(e.value & 0xFFFFF), l1 = (uint32_t)((e.value >> 20) & 0xFFFFF);      // Value40 limbs (value.rs:592-596)
    const uint32_t c0 = l0 & mask, c1 = (l0 >> chunk_bits) & mask, c2 = l1 & mask, c3 = (l1 >> chunk_bits) & mask;
    chunks[i] = (uint16_t)c0; chunks[stride + i] = (uint16_t)c1; chunks[2 * stride + i] = (uint16_t)c2; chunks[3 * stride + i] = (uint16_t)c3;
    if (mult) { atomicAdd(&hist[c0], 1u); atomicAdd(&hist[c1], 1u); atomicAdd(&hist[c2], 1u); atomicAdd(&hist[c3], 1u); }
  }
  if (mult) {
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < table; k += NT) if (hist[k]) atomicAdd(&mult[k], hist[k]);
  }
}

// ------------------------------------------------------------------------------------------------
// normalization events
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void norm_kernel(const zkir_norm_event* __restrict__ ev, uint64_t n, zkir_norm_columns c) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  const zkir_norm_event e = ev[i];
  const unsigned bits = e.state == 0 ? 20 : 30;                  // read_reg_limbs_extended, state.rs:202-220
  const uint64_t mask = (1ull << bits) - 1;
  const uint64_t a0 = e.raw_value & mask, a1 = (e.raw_value >> bits) & mask;
  const uint32_t c0 = (uint32_t)(a0 >> 20), n0 = (uint32_t)(a0 & 0xFFFFF);      // normalize.rs:138-145
  const uint64_t t = a1 + c0;
  const uint32_t c1 = (uint32_t)(t >> 20), n1 = (uint32_t)(t & 0xFFFFF);
  c.cycle[i] = e.cycle; c.pc[i] = e.pc; c.reg[i] = e.reg; c.opcode[i] = e.opcode;
  c.accumulated0[i] = a0; c.accumulated1[i] = a1;
  c.normalized0[i] = n0; c.normalized1[i] = n1;
  c.carry0[i] = c0; c.carry1[i] = c1;
}

// ------------------------------------------------------------------------------------------------
// SHA-256 chip: one lane per single-block message; 608 word-columns, column-major so every store is coalesced
// ------------------------------------------------------------------------------------------------
__constant__ uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_rotateright32(x, n); }

// column ids: [0,16) message_block, [16,24) initial_state, [24,88) message_schedule, [88,600) round_states[64][8], [600,608) final_state
__global__ __launch_bounds__(NT) void sha256_chip_kernel(const zkir_sha_block* __restrict__ blocks, uint64_t n, uint32_t* __restrict__ out, uint64_t stride,
                                                          uint64_t* __restrict__ timestamps) {
  const uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  uint32_t w[16];
  const uint4* src = reinterpret_cast<const uint4*>(blocks + i);               // 72-byte records: 8-byte aligned only
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(blocks + i);
  (void)src;
#pragma unroll
  for (int k = 0; k < 16; k++) { w[k] = sw[k]; out[(uint64_t)k * stride + i] = w[k]; }
  if (timestamps) timestamps[i] = blocks[i].timestamp;
  const uint32_t H0[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
#pragma unroll
  for (int k = 0; k < 8; k++) out[(uint64_t)(16 + k) * stride + i] = H0[k];
  uint32_t a = H0[0], b = H0[1], c = H0[2], d = H0[3], e = H0[4], f = H0[5], g = H0[6], h = H0[7];
#pragma unroll
  for (int t = 0; t < 64; t++) {
    uint32_t wt;
    if (t < 16) wt = w[t];
    else {                                                                     // crypto.rs:149-154, rolling 16-word window
      const uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
      wt = (rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10)) + w[(t + 9) & 15] + (rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3)) + w[t & 15];
      w[t & 15] = wt;
    }
    out[(uint64_t)(24 + t) * stride + i] = wt;
    const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[t] + wt;       // crypto.rs:173-179
    const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    uint32_t* o = out + (uint64_t)(88 + 8 * t) * stride + i;
    o[0] = a; o[stride] = b; o[2 * stride] = c; o[3 * stride] = d; o[4 * stride] = e; o[5 * stride] = f; o[6 * stride] = g; o[7 * stride] = h;
  }
  uint32_t* o = out + 600ull * stride + i;
  o[0] = H0[0] + a; o[stride] = H0[1] + b; o[2 * stride] = H0[2] + c; o[3 * stride] = H0[3] + d;
  o[4 * stride] = H0[4] + e; o[5 * stride] = H0[5] + f; o[6 * stride] = H0[6] + g; o[7 * stride] = H0[7] + h;
}

template <int K, int N, class F>
__device__ __forceinline__ void sha_static_for(F&& f) {
  if constexpr (K < N) { f(std::integral_constant<int, K>{}); sha_static_for<K + 1, N>(f); }
}

// The same with FOUR consecutive blocks per lane: every column store is one 16-byte non-temporal store per lane (1 KiB per wave
// instruction instead of 256 B) — the kernel is a pure HBM writer (2432 B out per 72 B in) and 4-byte stores left it at 3-4 TB/s.
// Needs stride % 4 == 0 and a 16-byte aligned `out`; the last lane of a ragged tail falls back to scalar stores for its blocks.
__global__ __launch_bounds__(NT) void sha256_chip_x4_kernel(const zkir_sha_block* __restrict__ blocks, uint64_t n, uint32_t* __restrict__ out, uint64_t stride,
                                                             uint64_t* __restrict__ timestamps) {
  using v4 = __attribute__((ext_vector_type(4))) unsigned int;
  const uint64_t i0 = ((uint64_t)blockIdx.x * NT + threadIdx.x) * 4;
  if (i0 >= n) return;
  const int cnt = n - i0 >= 4 ? 4 : (int)(n - i0);
  auto put = [&](uint64_t col, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    uint32_t* p = out + col * stride + i0;
    if (cnt == 4) { v4 v = {v0, v1, v2, v3}; __builtin_nontemporal_store(v, reinterpret_cast<v4*>(p)); }
    else { p[0] = v0; if (cnt > 1) p[1] = v1; if (cnt > 2) p[2] = v2; }
  };
  uint32_t w[4][16];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint64_t i = i0 + (q < cnt ? q : 0);                                 // lanes of a ragged tail re-read block i0: values unused
    const uint2* src = reinterpret_cast<const uint2*>(blocks + i);             // 72-byte records: 8-byte aligned
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint2 v = src[k]; w[q][2 * k] = v.x; w[q][2 * k + 1] = v.y; }
    if (timestamps && q < cnt) timestamps[i0 + q] = blocks[i0 + q].timestamp;
  }
#pragma unroll
  for (int k = 0; k < 16; k++) put(k, w[0][k], w[1][k], w[2][k], w[3][k]);
  const uint32_t H0[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
#pragma unroll
  for (int k = 0; k < 8; k++) put(16 + k, H0[k], H0[k], H0[k], H0[k]);
  uint32_t st[4][8];
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int k = 0; k < 8; k++) st[q][k] = H0[k];
  sha_static_for<0, 64>([&](auto tc) {                          // compile-time round index: w[][] and st[][] stay in registers
    constexpr int t = decltype(tc)::value;
    uint32_t wt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if constexpr (t < 16) wt[q] = w[q][t];
      else {
        const uint32_t w15 = w[q][(t + 1) & 15], w2 = w[q][(t + 14) & 15];
        wt[q] = (rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10)) + w[q][(t + 9) & 15] + (rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3)) + w[q][t & 15];
        w[q][t & 15] = wt[q];
      }
      const uint32_t a = st[q][0], b = st[q][1], c = st[q][2], d = st[q][3], e = st[q][4], f = st[q][5], g = st[q][6], h = st[q][7];
      const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[t] + wt[q];
      const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      st[q][7] = g; st[q][6] = f; st[q][5] = e; st[q][4] = d + t1; st[q][3] = c; st[q][2] = b; st[q][1] = a; st[q][0] = t1 + t2;
    }
    put(24 + t, wt[0], wt[1], wt[2], wt[3]);
#pragma unroll
    for (int k = 0; k < 8; k++) put(88 + 8 * t + k, st[0][k], st[1][k], st[2][k], st[3][k]);
  });
#pragma unroll
  for (int k = 0; k < 8; k++) put(600 + k, H0[k] + st[0][k], H0[k] + st[1][k], H0[k] + st[2][k], H0[k] + st[3][k]);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { zkir::set_last_error({ZKIR_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)}); return ZKIR_ERR_DEVICE; }
  return ZKIR_OK;
}

}  // namespace

extern "C" {

int zkir_memops_expand_launch(const zkir_mem_event* ev, uint64_t n_ops, uint64_t cycle_base, const zkir_memop_columns* out, void* stream) {
  if (n_ops == 0) return ZKIR_OK;
  hipLaunchKernelGGL(memops_expand_kernel, dim3(grid_for(n_ops)), dim3(NT), 0, (hipStream_t)stream, ev, n_ops, cycle_base, *out);
  return check_launch("memops_expand");
}

int zkir_memops_expand_csr_launch(const zkir_mem_event* ev, uint64_t n_ops, uint64_t n_rows, uint64_t cycle_base, const zkir_memop_columns* out, uint64_t* row_offsets,
                                  uint8_t* seg_flags, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!row_offsets || !seg_flags || (n_ops && (!ev || !out))) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_memops_expand_csr_launch: null argument"}); return ZKIR_ERR_ARGUMENT; }
  if (hipMemsetAsync(seg_flags, 0, n_rows ? n_rows : 1, s) != hipSuccess) return check_launch("memops_expand_csr memset");
  if (n_ops == 0) {                                              // no ops: every offset is 0
    if (hipMemsetAsync(row_offsets, 0, (n_rows + 1) * 8, s) != hipSuccess) return check_launch("memops_expand_csr memset");
    return ZKIR_OK;
  }
  hipLaunchKernelGGL(memops_expand_csr_kernel, dim3(grid_for(n_ops)), dim3(NT), 0, s, ev, n_ops, n_rows, cycle_base, *out, row_offsets, seg_flags);
  return check_launch("memops_expand_csr");
}

int zkir_memops_sort_prepared_launch(const zkir_mem_event* ev, uint64_t n_ops, uint64_t cycle_base, const uint64_t* row_offsets, const uint8_t* seg_flags,
                                     const zkir_memop_columns* out, void* stream) {
  if (n_ops == 0) return ZKIR_OK;
  hipLaunchKernelGGL(memops_sort_kernel, dim3(grid_for(n_ops)), dim3(NT), 0, (hipStream_t)stream, ev, n_ops, cycle_base, row_offsets, seg_flags, *out);
  return check_launch("memops_sort");
}

int zkir_memops_row_offsets_launch(const zkir_mem_event* ev, uint64_t n_ops, uint64_t n_rows, uint64_t* offsets, void* stream) {
  hipLaunchKernelGGL(memops_row_offsets_kernel, dim3(grid_for(n_rows + 1)), dim3(NT), 0, (hipStream_t)stream, ev, n_ops, n_rows, offsets);
  return check_launch("memops_row_offsets");
}

int zkir_memops_sort_launch(const zkir_mem_event* ev, uint64_t n_ops, uint64_t n_rows, uint64_t cycle_base, const uint64_t* row_offsets,
                            uint8_t* seg_scratch, const zkir_memop_columns* out, void* stream) {
  if (n_ops == 0) return ZKIR_OK;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(seg_scratch, 0, n_rows, s) != hipSuccess) return check_launch("memops_sort memset");
  hipLaunchKernelGGL(memops_segment_check_kernel, dim3(grid_for(n_ops)), dim3(NT), 0, s, ev, n_ops, seg_scratch);
  hipLaunchKernelGGL(memops_sort_kernel, dim3(grid_for(n_ops)), dim3(NT), 0, s, ev, n_ops, cycle_base, row_offsets, seg_scratch, *out);
  return check_launch("memops_sort");
}

int zkir_range_check_expand_launch(const zkir_rc_event* ev, uint64_t n, uint32_t chunk_bits, uint64_t* value, uint64_t* pc, uint16_t* chunks,
                                   uint64_t chunk_stride, uint32_t* multiplicity, void* stream) {
  if (chunk_bits < 8 || chunk_bits > 15) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "chunk_bits must be in 8..15"}); return ZKIR_ERR_ARGUMENT; }
  hipStream_t s = (hipStream_t)stream;
  if (multiplicity && hipMemsetAsync(multiplicity, 0, sizeof(uint32_t) << chunk_bits, s) != hipSuccess) return check_launch("range_check memset");
  if (n == 0) return ZKIR_OK;
  unsigned g = grid_for(n);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(range_check_kernel, dim3(g), dim3(NT), multiplicity ? (sizeof(uint32_t) << chunk_bits) : 0, s, ev, n, chunk_bits, value, pc, chunks,
                     chunk_stride, multiplicity);
  return check_launch("range_check");
}

int zkir_norm_expand_launch(const zkir_norm_event* ev, uint64_t n, const zkir_norm_columns* out, void* stream) {
  if (n == 0) return ZKIR_OK;
  hipLaunchKernelGGL(norm_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, ev, n, *out);
  return check_launch("norm_expand");
}

int zkir_sha256_chip_launch(const zkir_sha_block* blocks, uint64_t n, uint32_t* out, uint64_t stride, uint64_t* timestamps, void* stream) {
  if (n == 0) return ZKIR_OK;
  if (stride < n) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "sha256_chip: stride < n"}); return ZKIR_ERR_ARGUMENT; }
  if (stride % 4 == 0 && ((uintptr_t)out & 15) == 0)           // 16-byte column stores, four blocks per lane
    hipLaunchKernelGGL(sha256_chip_x4_kernel, dim3(grid_for((n + 3) / 4)), dim3(NT), 0, (hipStream_t)stream, blocks, n, out, stride, timestamps);
  else
    hipLaunchKernelGGL(sha256_chip_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, blocks, n, out, stride, timestamps);
  return check_launch("sha256_chip");
}

}  // extern "C"
