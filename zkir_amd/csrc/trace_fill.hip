// trace_fill.hip — K1: expand the register-write log into the wide SoA execution trace (gfx950).
//
// What it computes (reference: TraceRow, zkir-spec/src/trace.rs:24-50, built per cycle by
// VM::run, zkir-runtime/src/vm.rs:245-253,302-312): for every row i and register r the PRE-state triple
// (registers[r], bounds[r], register_states[r]) = the last write to r by an instruction at a cycle < i,
// plus the cycle column.  pc / instruction columns arrive in final form with the delta log.
//
// Formulation: a per-register "last writer" problem.  Rows are cut into tiles of T rows; the host
// interpreter's tile index gives, per tile, the slice of the (time-ordered) event log that becomes
// visible inside the tile and, per register, the last event visible at the tile's first row.  One
// workgroup owns one tile:
//   A. stage the tile's events (32 B each, coalesced 16-B loads) + the 16 snapshot events into LDS;
//   B. scatter event slot ids into an LDS marker table idx[16][T] (u16), recording which registers
//      are written inside the tile at all;
//   C. for those registers only, a wave-level inclusive max-scan along the rows turns markers into
//      "last writer" slot ids (slot ids grow with time, so last == max);
//   D. every column is written with 16-byte stores, each lane covering 2 (u64), 4 (u32) or 16 (u8)
//      consecutive rows; registers not written in the tile take a broadcast path with no LDS lookups.
// No inter-workgroup communication, no atomics on global memory, no re-reads: HBM traffic is the
// 360 B/row of columns written plus the 32 B/event log read once.  Bandwidth-bound by construction;
// MFMA is not applicable (no contraction, 40/64-bit integer payloads).
#include <hip/hip_runtime.h>

#include "../../include/zkir_amd.h"
#include "host.h"

namespace {

using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;

struct alignas(16) U64x2 { uint64_t a, b; };
struct alignas(16) U32x4 { uint32_t a, b, c, d; };

template <bool NTS, typename V>
__device__ __forceinline__ void store16(void* p, const V& v) {
  static_assert(sizeof(V) == 16, "16-byte store");
  using v4 = __attribute__((ext_vector_type(4))) unsigned int;
  if constexpr (NTS) __builtin_nontemporal_store(*reinterpret_cast<const v4*>(&v), reinterpret_cast<v4*>(p));
  else *reinterpret_cast<v4*>(p) = *reinterpret_cast<const v4*>(&v);
}

// LDS image of one event: same 32 bytes as zkir_reg_event, viewed as dwords.
//   dw0-1 value, dw2-3 payload, dw4 max_bits, dw5 vis, dw6 = reg | state<<8 | tag<<16
struct alignas(16) EvLds { uint32_t dw[8]; };

template <int T, int NT, int EVCAP, bool NTS>
__global__ __launch_bounds__(NT) void trace_fill_kernel(const zkir_reg_event* __restrict__ events, const uint32_t* __restrict__ tile_ev_off,
                                                          const uint32_t* __restrict__ tile_snap, uint32_t tile0, uint64_t cycle_base, uint64_t* __restrict__ col_cycle,
                                                          uint64_t* __restrict__ col_reg, uint32_t* __restrict__ col_bits, uint8_t* __restrict__ col_tag,
                                                          uint64_t* __restrict__ col_payload, uint8_t* __restrict__ col_state, uint64_t reg_stride) {
  static_assert(T % 128 == 0 && NT % 64 == 0, "tile geometry");
  constexpr int NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  EvLds* ev = reinterpret_cast<EvLds*>(smem);                                   // [EVCAP] slots: 0..15 snapshot, 16.. tile events
  unsigned short* idx = reinterpret_cast<unsigned short*>(smem + sizeof(EvLds) * EVCAP);   // [16][T]
  unsigned int* active = reinterpret_cast<unsigned int*>(smem + sizeof(EvLds) * EVCAP + 2 * 16 * T);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t tile = blockIdx.x + tile0;                                     // tile0: first tile of a partial launch (zkir_trace_fill_range_launch)
  const uint64_t row0 = (uint64_t)tile * T;
  const uint32_t ev_lo = tile_ev_off[tile], ev_hi = tile_ev_off[tile + 1];
  const uint32_t m = ev_hi - ev_lo;                                             // events becoming visible at rows row0+1 .. row0+T
  const uint32_t m_lds = m < (uint32_t)(EVCAP - 16) ? m : (uint32_t)(EVCAP - 16);

  // ---- A. stage events into LDS ------------------------------------------------------------------
  {
    const uint4* src = reinterpret_cast<const uint4*>(events + ev_lo);
    uint4* dst = reinterpret_cast<uint4*>(ev + 16);
    for (uint32_t i = tid; i < 2 * m_lds; i += NT) dst[i] = src[i];
    if (tid < 32) {                                                             // snapshot: event tile_snap[tile][r] -> slot r
      const uint32_t e = tile_snap[(uint64_t)tile * 16 + (tid >> 1)];
      reinterpret_cast<uint4*>(ev)[tid] = reinterpret_cast<const uint4*>(events + e)[tid & 1];
    }
    uint4* z = reinterpret_cast<uint4*>(idx);
    for (int i = tid; i < 16 * T * 2 / 16; i += NT) z[i] = make_uint4(0, 0, 0, 0);
    if (tid == 0) *active = 0;
  }
  __syncthreads();

  auto ev_dw = [&](uint32_t slot, int k) -> uint32_t {                          // slot: 1-based (0 = none is never looked up)
    const uint32_t s = slot - 1;
    if (s < (uint32_t)EVCAP) return ev[s].dw[k];
    return reinterpret_cast<const uint32_t*>(events + ev_lo + (s - 16))[k];     // overflow: straight from global (L2)
  };
  auto ev_u64 = [&](uint32_t slot, int k) -> uint64_t {
    const uint32_t s = slot - 1;
    if (s < (uint32_t)EVCAP) return *reinterpret_cast<const uint64_t*>(&ev[s].dw[k]);
    return *reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint32_t*>(events + ev_lo + (s - 16)) + k);
  };

  // ---- B. scatter markers ------------------------------------------------------------------------
  {
    unsigned int my = 0;
    for (uint32_t k = tid; k < m; k += NT) {
      const uint32_t slot = 17 + k;
      const uint32_t vis = ev_dw(slot, 5), meta = ev_dw(slot, 6);
      const uint32_t j = vis - (uint32_t)row0;                                  // >= 1 by construction of tile_ev_off
      if (j < (uint32_t)T) {
        const uint32_t r = meta & 0xF;
        idx[r * T + j] = (unsigned short)slot;                                  // at most one event per (reg, vis): plain store
        my |= 1u << r;
      }
    }
    if (my) atomicOr(active, my);
  }
  __syncthreads();
  const unsigned int act = *active;

  // ---- C. last-writer scan for the registers written inside this tile -----------------------------
  {
    int n = 0;
    for (int r = 1; r < 16; r++) {
      if (!((act >> r) & 1)) continue;
      if ((n++ % NW) != wave) continue;
      unsigned int carry = r + 1;                                               // snapshot slot of register r
      unsigned short* row = idx + r * T;
      for (int c = 0; c < T; c += 128) {                                        // 2 rows per lane per step
        const unsigned int pair = *reinterpret_cast<const unsigned int*>(row + c + 2 * lane);
        unsigned int lo = pair & 0xFFFF, hi = pair >> 16;
        hi = hi > lo ? hi : lo;                                                 // inclusive within the pair
        unsigned int v = hi;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const unsigned int u = __shfl_up(v, d, 64);
          if (lane >= d) v = v > u ? v : u;
        }
        unsigned int prev = __shfl_up(v, 1, 64);                                // exclusive prefix for this lane's pair
        if (lane == 0) prev = 0;
        prev = prev > carry ? prev : carry;
        lo = lo > prev ? lo : prev;
        hi = hi > prev ? hi : prev;
        *reinterpret_cast<unsigned int*>(row + c + 2 * lane) = lo | (hi << 16);
        carry = __shfl(hi, 63, 64);
      }
    }
  }
  __syncthreads();

  // ---- D. write the columns ------------------------------------------------------------------------
  // cycle column
  for (int q = tid; q < T / 2; q += NT) {
    const uint64_t c = cycle_base + row0 + 2 * (uint64_t)q;
    store16<NTS>(col_cycle + row0 + 2 * q, U64x2{c, c + 1});
  }
#pragma unroll 1
  for (int r = 0; r < 16; r++) {
    uint64_t* creg = col_reg + (uint64_t)r * reg_stride + row0;
    uint64_t* cpay = col_payload + (uint64_t)r * reg_stride + row0;
    uint32_t* cbits = col_bits + (uint64_t)r * reg_stride + row0;
    uint8_t* ctag = col_tag + (uint64_t)r * reg_stride + row0;
    uint8_t* cst = col_state + (uint64_t)r * reg_stride + row0;
    if (!((act >> r) & 1)) {
      // broadcast path: the whole tile sees the snapshot triple
      const uint64_t v = ev_u64(r + 1, 0), p = ev_u64(r + 1, 2);
      const uint32_t b = ev_dw(r + 1, 4), meta = ev_dw(r + 1, 6);
      const uint32_t st = (meta >> 8) & 0xFF, tg = (meta >> 16) & 0xFF;
      const U64x2 vv{v, v}, pp{p, p};
      const U32x4 bb{b, b, b, b};
      const uint32_t t4 = tg * 0x01010101u, s4 = st * 0x01010101u;
      const U32x4 tt{t4, t4, t4, t4}, ss{s4, s4, s4, s4};
      for (int q = tid; q < T / 2; q += NT) { store16<NTS>(creg + 2 * q, vv); store16<NTS>(cpay + 2 * q, pp); }
      for (int q = tid; q < T / 4; q += NT) store16<NTS>(cbits + 4 * q, bb);
      for (int q = tid; q < T / 16; q += NT) { store16<NTS>(ctag + 16 * q, tt); store16<NTS>(cst + 16 * q, ss); }
    } else {
      const unsigned short* row = idx + r * T;
      for (int q = tid; q < T / 2; q += NT) {
        const unsigned int pair = *reinterpret_cast<const unsigned int*>(row + 2 * q);
        const uint32_t s0 = pair & 0xFFFF, s1 = pair >> 16;
        store16<NTS>(creg + 2 * q, U64x2{ev_u64(s0, 0), ev_u64(s1, 0)});
        store16<NTS>(cpay + 2 * q, U64x2{ev_u64(s0, 2), ev_u64(s1, 2)});
      }
      for (int q = tid; q < T / 4; q += NT) {
        const uint2 quad = *reinterpret_cast<const uint2*>(row + 4 * q);
        store16<NTS>(cbits + 4 * q, U32x4{ev_dw(quad.x & 0xFFFF, 4), ev_dw(quad.x >> 16, 4), ev_dw(quad.y & 0xFFFF, 4), ev_dw(quad.y >> 16, 4)});
      }
      for (int q = tid; q < T / 16; q += NT) {
        const u16x8 s_lo = *reinterpret_cast<const u16x8*>(row + 16 * q), s_hi = *reinterpret_cast<const u16x8*>(row + 16 * q + 8);
        uint32_t tw[4] = {0, 0, 0, 0}, sw[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const uint32_t meta = ev_dw(k < 8 ? s_lo[k] : s_hi[k - 8], 6);
          sw[k >> 2] |= ((meta >> 8) & 0xFF) << (8 * (k & 3));
          tw[k >> 2] |= ((meta >> 16) & 0xFF) << (8 * (k & 3));
        }
        store16<NTS>(ctag + 16 * q, U32x4{tw[0], tw[1], tw[2], tw[3]});
        store16<NTS>(cst + 16 * q, U32x4{sw[0], sw[1], sw[2], sw[3]});
      }
    }
  }
}

template <int T, int NT, bool NTS>
int launch_tile(const zkir_trace_fill_args* a, uint32_t tile0, uint32_t n_tiles, hipStream_t stream) {
  constexpr int EVCAP = 16 + T;
  constexpr size_t lds = sizeof(EvLds) * EVCAP + 2 * 16 * T + 16;
  auto k = trace_fill_kernel<T, NT, EVCAP, NTS>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(k, dim3(n_tiles), dim3(NT), lds, stream, a->events, a->tile_ev_off, a->tile_snap, tile0, a->cycle_base, a->out.cycle, a->out.registers,
                     a->out.bound_bits, a->out.bound_tag, a->out.bound_payload, a->out.reg_state, a->out.reg_stride);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { zkir::set_last_error({ZKIR_ERR_DEVICE, std::string("trace_fill launch failed: ") + hipGetErrorString(e)}); return ZKIR_ERR_DEVICE; }
  return ZKIR_OK;
}

}  // namespace

extern "C" int zkir_trace_fill_range_launch(const zkir_trace_fill_args* a, uint64_t tile_begin, uint64_t tile_end, void* hip_stream);
extern "C" int zkir_trace_fill_launch(const zkir_trace_fill_args* a, void* hip_stream) {
  if (!a) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_trace_fill_launch: null args"}); return ZKIR_ERR_ARGUMENT; }
  return zkir_trace_fill_range_launch(a, 0, (a->n_rows + a->tile_rows - 1) / (a->tile_rows ? a->tile_rows : 1), hip_stream);
}

// tiles [tile_begin, tile_end) of the trace described by `a` (its arrays and columns are those of the WHOLE trace): what a caller
// that streams the delta log to the device while the interpreter is still running launches for the tiles it already has
extern "C" int zkir_trace_fill_range_launch(const zkir_trace_fill_args* a, uint64_t tile_begin, uint64_t tile_end, void* hip_stream) {
  if (!a) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_trace_fill_launch: null args"}); return ZKIR_ERR_ARGUMENT; }
  if (a->n_rows == 0 || tile_begin >= tile_end) return ZKIR_OK;
  const uint32_t T = a->tile_rows;
  const uint64_t n_tiles = (a->n_rows + T - 1) / T;
  if (tile_end > n_tiles) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_trace_fill_range_launch: tile range past the end of the trace"}); return ZKIR_ERR_ARGUMENT; }
  const uint32_t t0 = (uint32_t)tile_begin, cnt = (uint32_t)(tile_end - tile_begin);
  if (a->out.reg_stride < n_tiles * T || (a->out.reg_stride & 15)) {
    zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_trace_fill_launch: reg_stride must be a multiple of 16 and >= n_rows rounded up to tile_rows"});
    return ZKIR_ERR_ARGUMENT;
  }
  hipStream_t s = (hipStream_t)hip_stream;
  int rc;
  // tuning knobs (benchmarking only): ZKIR_TF_THREADS = 256|512, ZKIR_TF_NT = 0|1
  static const int threads_env = getenv("ZKIR_TF_THREADS") ? atoi(getenv("ZKIR_TF_THREADS")) : 0;
  const int threads = threads_env ? threads_env : (T >= 512 ? 512 : 256);    // measured best on MI355X (profiles/r01_sweep_trace_fill.txt)
  static const int nts = getenv("ZKIR_TF_NT") ? atoi(getenv("ZKIR_TF_NT")) : 1;
#define ZKIR_TF_CASE(TT)                                                                          \
  case TT:                                                                                        \
    if (threads == 512) rc = nts ? launch_tile<TT, 512, true>(a, t0, cnt, s) : launch_tile<TT, 512, false>(a, t0, cnt, s); \
    else rc = nts ? launch_tile<TT, 256, true>(a, t0, cnt, s) : launch_tile<TT, 256, false>(a, t0, cnt, s);               \
    break;
  switch (T) {
    ZKIR_TF_CASE(256)
    ZKIR_TF_CASE(512)
    ZKIR_TF_CASE(1024)
    ZKIR_TF_CASE(2048)
    default: zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_trace_fill_launch: tile_rows must be 256, 512, 1024 or 2048"}); return ZKIR_ERR_ARGUMENT;
  }
  return rc;
}

extern "C" uint64_t zkir_trace_fill_bytes(uint64_t n_rows, uint64_t n_events, uint64_t n_tiles) {
  return 360ull * n_rows + 32ull * n_events + 68ull * n_tiles;
}
