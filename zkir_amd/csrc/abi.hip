// abi.hip — extern "C" entry points of include/zkir_amd.h: delta-log accessors and the drop-in
// zkir_exec (VM::new + VM::run replacement: host interpreter -> H2D -> K1 trace fill).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "../../include/zkir_amd.h"
#include "host.h"

namespace zkir {
static thread_local std::string g_last_error;
void set_last_error(const Status& st) { g_last_error = st.msg; }

// host.h: the big blocks of the delta log are pinned when the process has a device (asked once; a host-only process — zkir_interpret / zkir_verify on a CPU box — never
// touches the runtime again after the first "no device").  hipHostMallocPortable: every device of the process may DMA out of the block.
void* pinned_alloc(size_t bytes) {
  static const bool have_device = [] { int n = 0; const bool ok = hipGetDeviceCount(&n) == hipSuccess && n > 0; if (!ok) (void)hipGetLastError(); return ok; }();
  if (!have_device) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void pinned_free(void* p) { if (p) (void)hipHostFree(p); }
}  // namespace zkir

struct zkir_result {
  zkir_delta_log* log = nullptr;
  zkir_trace_columns cols{};
  void* d_events = nullptr;
  void* d_tile_ev_off = nullptr;
  void* d_tile_snap = nullptr;
  void* d_block = nullptr;      // one allocation holding every trace column
  uint64_t cap_rows = 0;
  float stage_ms[4] = {0, 0, 0, 0};   // host interpretation | device allocation | H2D of the delta log | K1 + synchronisation
  std::vector<uint8_t> program;       // the program and the input tape of the run (zkir_exec / zkir_exec_window keep a copy: zkir_prove_result needs them for the public inputs)
  std::vector<uint64_t> inputs;
  bool whole_run = false;             // zkir_exec: rows [0, cycles); a window / shard is a SEGMENT of a run
  // witness streams of ExecutionResult (vm.rs:54-103), expanded on the device on first request and cached
  std::mutex wmu;
  bool mem_built = false, rc_built = false, norm_built = false, sha_built = false;
  void* d_mem = nullptr; zkir_memory_witness mem{};
  void* d_rc = nullptr; zkir_range_check_witness rc{};
  void* d_norm = nullptr; zkir_normalization_witness norm{};
  void* d_sha = nullptr; zkir_sha256_witness sha{};
};

namespace {
// carve `count` elements of T out of a device block (256-byte aligned pieces)
struct Carver {
  unsigned char* p; size_t off = 0;
  template <typename T> T* take(size_t count) { T* r = (T*)(p + off); off += (count * sizeof(T) + 255) & ~(size_t)255; return r; }
};
inline size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
}  // namespace

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      zkir::set_last_error({ZKIR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)});      \
      goto fail;                                                                                       \
    }                                                                                                  \
  } while (0)

extern "C" {

const char* zkir_last_error(void) { return zkir::g_last_error.c_str(); }
const char* zkir_version(void) { return "zkir_amd 0.1 (ZKIR v3.4, gfx950)"; }
uint32_t zkir_abi_version(void) { return ZKIR_AMD_ABI_VERSION; }

int zkir_interpret(const uint8_t* blob, size_t len, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config* cfg, uint32_t tile_rows,
                   zkir_delta_log** out) {
  if (!out || !cfg || (!blob && len)) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_interpret: null argument"}); return ZKIR_ERR_ARGUMENT; }
  *out = nullptr;
  zkir_delta_log* log = new zkir_delta_log();
  zkir::Status st;
  try {
    st = zkir::interpret(blob, len, inputs, n_inputs, *cfg, tile_rows, *log);
  } catch (const std::bad_alloc&) {
    st = {ZKIR_ERR_OTHER, "out of host memory while recording the delta log"};
  }
  if (!st.ok()) { zkir::set_last_error(st); delete log; return st.code; }
  *out = log;
  return ZKIR_OK;
}
void zkir_delta_log_free(zkir_delta_log* log) { delete log; }

int zkir_delta_log_shard(const zkir_delta_log* src, uint64_t row_begin, uint64_t row_end, zkir_delta_log** out) {
  if (!src || !out) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_delta_log_shard: null argument"}); return ZKIR_ERR_ARGUMENT; }
  *out = nullptr;
  const uint64_t T = src->tile_rows, B = src->cycle_base;                         // the source may itself be a window / shard: rows are absolute
  if (row_begin > row_end || row_begin < B || row_end > B + src->n_rows) {
    zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_delta_log_shard: need cycle_base <= row_begin <= row_end <= cycle_base + n_rows of the source log"});
    return ZKIR_ERR_ARGUMENT;
  }
  const uint64_t rb = row_begin - B, re = row_end - B;                            // positions inside the source
  zkir_delta_log* d = new zkir_delta_log();
  d->cycles = src->cycles; d->halt_kind = src->halt_kind; d->halt_code = src->halt_code; d->outputs = src->outputs; d->window_open = src->window_open;
  d->tile_rows = src->tile_rows; d->rc_chunk_bits = src->rc_chunk_bits;
  d->n_rows = row_end - row_begin; d->cycle_base = row_begin;
  d->rc_offsets.push_back(0);
  for (size_t k = 0; k < src->rc_cycles.size(); k++) {                            // a witness belongs to the shard that holds its checkpoint cycle (vm.rs:316-344)
    if (src->rc_cycles[k] < row_begin || src->rc_cycles[k] >= row_end) continue;
    d->rc_events.append(src->rc_events.data() + src->rc_offsets[k], src->rc_offsets[k + 1] - src->rc_offsets[k]);
    d->rc_offsets.push_back(d->rc_events.size());
    d->rc_cycles.push_back(src->rc_cycles[k]);
  }
  if (d->n_rows == 0) { d->tile_ev_off.push_back(0); *out = d; return ZKIR_OK; }
  // side logs: mem-event rows are relative to the log they sit in; normalization cycles and SHA timestamps are absolute
  auto side_logs = [&] {
    for (size_t m = 0; m < src->mem_events.size(); m++) {
      zkir_mem_event me = src->mem_events[m];
      if (me.row >= rb && me.row < re) { me.row -= (uint32_t)rb; d->mem_events.push(me); }
    }
    for (size_t m = 0; m < src->norm_events.size(); m++) if (src->norm_events[m].cycle >= row_begin && src->norm_events[m].cycle < row_end) d->norm_events.push(src->norm_events[m]);
    for (size_t m = 0; m < src->sha_blocks.size(); m++) if (src->sha_blocks[m].timestamp >= row_begin && src->sha_blocks[m].timestamp < row_end) d->sha_blocks.push(src->sha_blocks[m]);
  };
  d->pc.append(src->pc.data() + rb, d->n_rows);
  d->inst.append(src->inst.data() + rb, d->n_rows);
  if (rb % T != 0) {
    // A cut that is not on a tile boundary (segment proofs overlap by ONE row, so their first rows are at g (S - 1)): the shard gets
    // its own tiling from row_begin.  Snapshot = the source tile's snapshot advanced over the events visible by row_begin; then one
    // pass over the shard's events rebuilds tile_ev_off (first event with vis > tile start) and tile_snap (last event per register
    // with vis <= tile start), the conventions of include/zkir_amd.h.
    const uint64_t ts = rb / T, nt = (d->n_rows + T - 1) / T;
    const zkir_reg_event* ev = src->reg_events.data();
    const size_t n_ev = src->reg_events.size();
    uint32_t snap[16];
    for (int r = 0; r < 16; r++) snap[r] = src->tile_snap[ts * 16 + r];
    size_t e0 = src->tile_ev_off[ts];
    while (e0 < n_ev && ev[e0].vis <= rb) { snap[ev[e0].reg] = (uint32_t)e0; e0++; }
    const uint64_t end_vis = rb + nt * T;                                         // events up to the end of the shard's last tile
    size_t e1 = e0;
    while (e1 < n_ev && ev[e1].vis <= end_vis) e1++;
    for (int r = 0; r < 16; r++) { zkir_reg_event e = ev[snap[r]]; e.vis = 0; d->reg_events.push(e); }
    for (size_t k = e0; k < e1; k++) { zkir_reg_event e = ev[k]; e.vis -= (uint32_t)rb; d->reg_events.push(e); }
    uint32_t last[16];
    for (int r = 0; r < 16; r++) last[r] = (uint32_t)r;
    size_t k = e0;
    for (uint64_t j = 0; j < nt; j++) {
      const uint64_t start = rb + j * T;
      while (k < e1 && ev[k].vis <= start) { last[ev[k].reg] = (uint32_t)(k - e0 + 16); k++; }
      d->tile_ev_off.push_back((uint32_t)(k - e0 + 16));
      for (int r = 0; r < 16; r++) d->tile_snap.push_back(last[r]);
    }
    d->tile_ev_off.push_back((uint32_t)(e1 - e0 + 16));
    side_logs();
    *out = d;
    return ZKIR_OK;
  }
  const uint64_t t0 = rb / T, t1 = (re + T - 1) / T;                               // tiles [t0, t1)
  const uint32_t e0 = src->tile_ev_off[t0];
  // events up to the end of the last tile; for an interior cut that is tile_ev_off[t1], for the run's tail all remaining
  const uint32_t e1 = src->tile_ev_off[t1];
  for (int r = 0; r < 16; r++) {                                                   // snapshot at row_begin
    zkir_reg_event e = src->reg_events[src->tile_snap[t0 * 16 + r]];
    e.vis = 0;
    d->reg_events.push(e);
  }
  for (uint32_t k = e0; k < e1; k++) { zkir_reg_event e = src->reg_events[k]; e.vis -= (uint32_t)rb; d->reg_events.push(e); }
  for (uint64_t t = t0; t <= t1; t++) d->tile_ev_off.push_back(src->tile_ev_off[t] - e0 + 16);
  for (uint64_t t = t0; t < t1; t++)
    for (int r = 0; r < 16; r++) { const uint32_t i = src->tile_snap[t * 16 + r]; d->tile_snap.push_back(i < e0 ? (uint32_t)r : i - e0 + 16); }
  side_logs();
  *out = d;
  return ZKIR_OK;
}
uint64_t zkir_delta_log_cycle_base(const zkir_delta_log* l) { return l->cycle_base; }
int zkir_delta_log_window_open(const zkir_delta_log* l) { return l->window_open ? 1 : 0; }

int zkir_interpret_window(const uint8_t* blob, size_t len, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config* cfg, uint32_t tile_rows,
                          uint64_t row_begin, uint64_t row_end, zkir_delta_log** out) {
  if (!out || !cfg || (!blob && len)) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_interpret_window: null argument"}); return ZKIR_ERR_ARGUMENT; }
  *out = nullptr;
  if (!cfg->enable_execution_trace) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_interpret_window: a trace window needs enable_execution_trace"}); return ZKIR_ERR_ARGUMENT; }
  zkir_delta_log* log = new zkir_delta_log();
  zkir::Status st;
  try {
    st = zkir::interpret(blob, len, inputs, n_inputs, *cfg, tile_rows, *log, nullptr, row_begin, row_end);
  } catch (const std::bad_alloc&) {
    st = {ZKIR_ERR_OTHER, "out of host memory while recording the delta log"};
  }
  if (!st.ok()) { zkir::set_last_error(st); delete log; return st.code; }
  *out = log;
  return ZKIR_OK;
}

uint64_t zkir_delta_log_cycles(const zkir_delta_log* l) { return l->cycles; }
int zkir_delta_log_halt_kind(const zkir_delta_log* l) { return l->halt_kind; }
uint64_t zkir_delta_log_halt_code(const zkir_delta_log* l) { return l->halt_code; }
size_t zkir_delta_log_n_outputs(const zkir_delta_log* l) { return l->outputs.size(); }
const uint64_t* zkir_delta_log_outputs(const zkir_delta_log* l) { return l->outputs.data(); }
uint64_t zkir_delta_log_n_rows(const zkir_delta_log* l) { return l->n_rows; }
uint32_t zkir_delta_log_tile_rows(const zkir_delta_log* l) { return l->tile_rows; }
const uint64_t* zkir_delta_log_pc(const zkir_delta_log* l) { return l->pc.data(); }
const uint32_t* zkir_delta_log_inst(const zkir_delta_log* l) { return l->inst.data(); }
size_t zkir_delta_log_n_reg_events(const zkir_delta_log* l) { return l->reg_events.size(); }
const zkir_reg_event* zkir_delta_log_reg_events(const zkir_delta_log* l) { return l->reg_events.data(); }
size_t zkir_delta_log_n_tiles(const zkir_delta_log* l) { return l->tile_ev_off.empty() ? 0 : l->tile_ev_off.size() - 1; }
const uint32_t* zkir_delta_log_tile_ev_off(const zkir_delta_log* l) { return l->tile_ev_off.data(); }
const uint32_t* zkir_delta_log_tile_snap(const zkir_delta_log* l) { return l->tile_snap.data(); }
size_t zkir_delta_log_n_mem_events(const zkir_delta_log* l) { return l->mem_events.size(); }
const zkir_mem_event* zkir_delta_log_mem_events(const zkir_delta_log* l) { return l->mem_events.data(); }
size_t zkir_delta_log_n_rc_events(const zkir_delta_log* l) { return l->rc_events.size(); }
const zkir_rc_event* zkir_delta_log_rc_events(const zkir_delta_log* l) { return l->rc_events.data(); }
size_t zkir_delta_log_n_rc_witnesses(const zkir_delta_log* l) { return l->rc_offsets.size() - 1; }
const uint64_t* zkir_delta_log_rc_offsets(const zkir_delta_log* l) { return l->rc_offsets.data(); }
const uint64_t* zkir_delta_log_rc_cycles(const zkir_delta_log* l) { return l->rc_cycles.data(); }
uint32_t zkir_delta_log_rc_chunk_bits(const zkir_delta_log* l) { return l->rc_chunk_bits; }
size_t zkir_delta_log_n_norm_events(const zkir_delta_log* l) { return l->norm_events.size(); }
const zkir_norm_event* zkir_delta_log_norm_events(const zkir_delta_log* l) { return l->norm_events.data(); }
size_t zkir_delta_log_n_sha_blocks(const zkir_delta_log* l) { return l->sha_blocks.size(); }
const zkir_sha_block* zkir_delta_log_sha_blocks(const zkir_delta_log* l) { return l->sha_blocks.data(); }

// ---- drop-in layer ------------------------------------------------------------------------------
static inline uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// Device side of zkir_exec.  The columns, the event array and the tile index are sized for `cap_rows` rows / the given capacities.
static int alloc_trace(zkir_result* r, uint64_t cap, size_t ev_cap, size_t toff_cap, size_t snap_cap) {
  // column block: cycle 8 | pc 8 | regs 128 | payload 128 | bits 64 | inst 4 | tag 16 | state 16  = 372 B per (padded) row
  unsigned char* base = nullptr;
  r->cap_rows = cap;
  if (hipMalloc(&r->d_block, cap * 372) != hipSuccess || hipMalloc(&r->d_events, ev_cap * sizeof(zkir_reg_event)) != hipSuccess ||
      hipMalloc(&r->d_tile_ev_off, toff_cap * 4) != hipSuccess || hipMalloc(&r->d_tile_snap, snap_cap * 4) != hipSuccess) {
    zkir::set_last_error({ZKIR_ERR_DEVICE, std::string("zkir_exec: hipMalloc: ") + hipGetErrorString(hipGetLastError())});
    return ZKIR_ERR_DEVICE;
  }
  base = (unsigned char*)r->d_block;
  r->cols.cycle = (uint64_t*)base;                   base += cap * 8;
  r->cols.pc = (uint64_t*)base;                      base += cap * 8;
  r->cols.registers = (uint64_t*)base;               base += cap * 128;
  r->cols.bound_payload = (uint64_t*)base;           base += cap * 128;
  r->cols.bound_bits = (uint32_t*)base;              base += cap * 64;
  r->cols.instruction = (uint32_t*)base;             base += cap * 4;
  r->cols.bound_tag = (uint8_t*)base;                base += cap * 16;
  r->cols.reg_state = (uint8_t*)base;                base += cap * 16;
  r->cols.reg_stride = cap;
  return ZKIR_OK;
}
static void free_trace(zkir_result* r) {
  if (r->d_block) (void)hipFree(r->d_block);
  if (r->d_events) (void)hipFree(r->d_events);
  if (r->d_tile_ev_off) (void)hipFree(r->d_tile_ev_off);
  if (r->d_tile_snap) (void)hipFree(r->d_tile_snap);
  r->d_block = r->d_events = r->d_tile_ev_off = r->d_tile_snap = nullptr;
  r->cols = zkir_trace_columns{};
}

// Uploads what the running interpreter has finished (zkir::Progress) and fills those tiles, on its own thread and stream, so that
// the H2D copy of the delta log and K1 hide under the interpretation instead of following it.
// streams are recycled (creating and destroying one per call costs more than the copies it carries)
// The pool is keyed by DEVICE: a hipStream_t belongs to the device that was current when it was created, and one process may drive
// several GPUs (hipSetDevice + one zkir_exec per device); a stream of device 0 must never carry the copies and K1 of a run on device 1.
static std::mutex g_stream_mu;
static std::map<int, std::vector<hipStream_t>> g_streams;
static hipError_t acquire_stream(int device, hipStream_t* s) {
  { std::lock_guard<std::mutex> lk(g_stream_mu); auto& v = g_streams[device]; if (!v.empty()) { *s = v.back(); v.pop_back(); return hipSuccess; } }
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);      // created on the calling thread's current device (= `device`, set by the caller)
}
static void release_stream(int device, hipStream_t s) { std::lock_guard<std::mutex> lk(g_stream_mu); g_streams[device].push_back(s); }

struct ExecStreamer {
  zkir_result* r; const zkir_delta_log* log; zkir::Progress prog; uint64_t max_cycles; int device;
  std::thread th;
  hipStream_t stream = nullptr;
  bool active = false, abandoned = false, failed = false;
  uint64_t up_rows = 0, up_events = 0, up_tiles = 0, up_toff = 0;
  std::string error;

  zkir_trace_fill_args args() const {
    zkir_trace_fill_args a{};
    a.events = (const zkir_reg_event*)r->d_events; a.tile_ev_off = (const uint32_t*)r->d_tile_ev_off; a.tile_snap = (const uint32_t*)r->d_tile_snap;
    a.n_rows = r->cap_rows; a.cycle_base = log->cycle_base; a.tile_rows = log->tile_rows; a.n_events = 0; a.out = r->cols;
    return a;
  }
  // copy rows / events / tile index up to (tiles, rows, events) and fill the new complete tiles
  bool advance(uint64_t tiles, uint64_t rows, uint64_t events, uint64_t toff_entries) {
    auto ok = [&](hipError_t e, const char* what) { if (e != hipSuccess) { error = std::string(what) + ": " + hipGetErrorString(e); failed = true; } return e == hipSuccess; };
    if (rows > up_rows) {
      if (!ok(hipMemcpyAsync(r->cols.pc + up_rows, log->pc.data() + up_rows, (rows - up_rows) * 8, hipMemcpyHostToDevice, stream), "H2D pc")) return false;
      if (!ok(hipMemcpyAsync(r->cols.instruction + up_rows, log->inst.data() + up_rows, (rows - up_rows) * 4, hipMemcpyHostToDevice, stream), "H2D instruction")) return false;
    }
    if (events > up_events && !ok(hipMemcpyAsync((zkir_reg_event*)r->d_events + up_events, log->reg_events.data() + up_events, (events - up_events) * sizeof(zkir_reg_event),
                                                hipMemcpyHostToDevice, stream), "H2D events")) return false;
    if (toff_entries > up_toff && !ok(hipMemcpyAsync((uint32_t*)r->d_tile_ev_off + up_toff, log->tile_ev_off.data() + up_toff, (toff_entries - up_toff) * 4, hipMemcpyHostToDevice, stream),
                                      "H2D tile index")) return false;
    if (tiles > up_tiles && !ok(hipMemcpyAsync((uint32_t*)r->d_tile_snap + up_tiles * 16, log->tile_snap.data() + up_tiles * 16, (tiles - up_tiles) * 64, hipMemcpyHostToDevice, stream),
                                "H2D tile snapshots")) return false;
    const zkir_trace_fill_args a = args();
    if (tiles > up_tiles && zkir_trace_fill_range_launch(&a, up_tiles, tiles, stream) != ZKIR_OK) { error = zkir_last_error(); failed = true; return false; }
    up_rows = rows; up_events = events; up_tiles = tiles; up_toff = toff_entries;
    return true;
  }
  void body() {
    (void)hipSetDevice(device);
    const uint32_t T = log->tile_rows;
    for (;;) {
      const bool fin = prog.finished.load();
      if (!prog.stable.load()) { abandoned = true; break; }
      const uint64_t t = prog.tiles.load(std::memory_order_acquire);
      if (t > up_tiles && t * T >= (1u << 16)) {                           // short runs never start streaming (no device allocation for max_cycles rows)
        prog.consumer_idle.store(false);
        if (!prog.stable.load()) { prog.consumer_idle.store(true); abandoned = true; break; }
        const uint64_t rows = prog.rows.load(std::memory_order_relaxed), events = prog.events.load(std::memory_order_relaxed);
        if (!active) {
          const uint64_t cap = (max_cycles + T - 1) / T * T;
          // The streaming buffers are sized for max_cycles rows (the columns are strided by the capacity, they cannot grow in place):
          // 372 B/row + 40 B/row of log, 27 GB at 2^26.  If the device cannot give that much — several concurrent calls, a run
          // that will halt long before max_cycles — streaming is ABANDONED, not failed: zkir_exec then takes the plain path,
          // which allocates for the rows the run actually produced.
          if (acquire_stream(device, &stream) != hipSuccess) { stream = nullptr; abandoned = true; }
          else if (alloc_trace(r, cap, log->reg_events.capacity(), log->tile_ev_off.capacity(), log->tile_snap.capacity()) != ZKIR_OK) { (void)hipGetLastError(); free_trace(r); abandoned = true; }
          if (abandoned) { prog.consumer_idle.store(true); break; }
          active = true;
        }
        advance(t, rows, events, t + 1);
        prog.consumer_idle.store(true);
        if (failed) break;
      } else if (fin) {
        break;
      } else {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
      }
    }
    prog.consumer_idle.store(true);
  }
};

// zkir_exec and zkir_exec_window: rows [row_begin, row_end) of the run (the whole run for zkir_exec) interpreted, uploaded and filled
static int exec_impl(const uint8_t* blob, size_t len, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config* cfg, uint64_t row_begin, uint64_t row_end,
                     zkir_result** out) {
  if (!out || !cfg) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_exec: null argument"}); return ZKIR_ERR_ARGUMENT; }
  *out = nullptr;
  if (row_begin > row_end) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_exec_window: row_begin > row_end"}); return ZKIR_ERR_ARGUMENT; }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
    zkir::set_last_error({ZKIR_ERR_DEVICE, "zkir_exec: no usable HIP device (the product path has no CPU fallback)"});
    return ZKIR_ERR_DEVICE;
  }
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point t0) { return std::chrono::duration<float, std::milli>(clk::now() - t0).count(); };
  zkir_result* r = new zkir_result();
  zkir_delta_log* log = new zkir_delta_log();
  r->log = log;
  if (blob && len) r->program.assign(blob, blob + len);
  if (inputs && n_inputs) r->inputs.assign(inputs, inputs + n_inputs);
  r->whole_run = row_begin == 0 && row_end == ~0ull;
  // Long traced runs stream: a second host thread uploads the finished part of the delta log and launches K1 on it while the
  // interpreter keeps running (ZKIR_EXEC_STREAM=0 turns it off).  Everything else takes the plain path: interpret, upload, fill.
  static const bool stream_ok = !(getenv("ZKIR_EXEC_STREAM") && atoi(getenv("ZKIR_EXEC_STREAM")) == 0);
  const uint64_t win_hi = row_end < cfg->max_cycles ? row_end : cfg->max_cycles, win_rows = win_hi > row_begin ? win_hi - row_begin : 0;   // rows the trace can have at most
  const bool streaming = stream_ok && cfg->enable_execution_trace && win_rows >= (1ull << 17) && win_rows <= (1ull << 26);
  std::unique_ptr<ExecStreamer> st;
  if (streaming) {
    st.reset(new ExecStreamer());
    st->r = r; st->log = log; st->max_cycles = win_rows; st->device = 0;
    (void)hipGetDevice(&st->device);
  }
  auto t0 = clk::now();
  zkir::Status status;
  if (streaming) st->th = std::thread([&] { st->body(); });
  try {
    status = zkir::interpret(blob, len, inputs, n_inputs, *cfg, 0, *log, streaming ? &st->prog : nullptr, row_begin, row_end);
  } catch (const std::bad_alloc&) {
    status = {ZKIR_ERR_OTHER, "out of host memory while recording the delta log"};
  }
  if (streaming) { st->prog.finished.store(true); st->th.join(); }
  r->stage_ms[0] = ms_since(t0);
  int rc = ZKIR_OK;
  hipStream_t s = nullptr;
  if (!status.ok()) { zkir::set_last_error(status); rc = status.code; goto fail_rc; }
  if (streaming && st->failed) { zkir::set_last_error({ZKIR_ERR_DEVICE, "zkir_exec (streaming): " + st->error}); rc = ZKIR_ERR_DEVICE; goto fail_rc; }
  {
    const uint64_t n = log->n_rows;
    if (n > 0) {
      const uint32_t T = log->tile_rows;
      const uint64_t n_tiles = (n + T - 1) / T;
      if (streaming && st->active && !st->abandoned) {                     // the tail: what the interpreter produced after the last report
        t0 = clk::now();
        s = st->stream;
        if (!st->advance(n_tiles, n, log->reg_events.size(), n_tiles + 1)) { zkir::set_last_error({ZKIR_ERR_DEVICE, "zkir_exec (streaming): " + st->error}); rc = ZKIR_ERR_DEVICE; goto fail_rc; }
        r->stage_ms[2] = ms_since(t0); t0 = clk::now();
        HIP_TRY(hipStreamSynchronize(s));
        r->stage_ms[3] = ms_since(t0);
      } else {
        if (streaming && st->active) { HIP_TRY(hipStreamSynchronize(st->stream)); free_trace(r); }   // streaming was abandoned (a log buffer had to grow): start over
        zkir_trace_fill_args a{};
        t0 = clk::now();
        rc = alloc_trace(r, round_up(n, T), log->reg_events.size(), log->tile_ev_off.size(), log->tile_snap.size());
        if (rc != ZKIR_OK) goto fail_rc;
        r->stage_ms[1] = ms_since(t0); t0 = clk::now();
        HIP_TRY(hipMemcpyAsync(r->d_events, log->reg_events.data(), log->reg_events.size() * sizeof(zkir_reg_event), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(r->d_tile_ev_off, log->tile_ev_off.data(), log->tile_ev_off.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(r->d_tile_snap, log->tile_snap.data(), log->tile_snap.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(r->cols.pc, log->pc.data(), n * 8, hipMemcpyHostToDevice, s));           // pc / instruction columns arrive in final form
        HIP_TRY(hipMemcpyAsync(r->cols.instruction, log->inst.data(), n * 4, hipMemcpyHostToDevice, s));
        r->stage_ms[2] = ms_since(t0); t0 = clk::now();
        a.events = (const zkir_reg_event*)r->d_events;
        a.tile_ev_off = (const uint32_t*)r->d_tile_ev_off;
        a.tile_snap = (const uint32_t*)r->d_tile_snap;
        a.n_rows = n; a.cycle_base = log->cycle_base; a.tile_rows = T; a.n_events = (uint32_t)log->reg_events.size();
        a.out = r->cols;
        rc = zkir_trace_fill_launch(&a, s);
        if (rc != ZKIR_OK) goto fail_rc;
        HIP_TRY(hipStreamSynchronize(s));
        r->stage_ms[3] = ms_since(t0);
      }
    } else if (streaming && st->active) {
      HIP_TRY(hipStreamSynchronize(st->stream)); free_trace(r);
    }
  }
  if (streaming && st->stream) release_stream(st->device, st->stream);
  *out = r;
  return ZKIR_OK;
fail:
  rc = ZKIR_ERR_DEVICE;
fail_rc:
  if (streaming && st && st->stream) { (void)hipStreamSynchronize(st->stream); release_stream(st->device, st->stream); }
  zkir_result_free(r);
  return rc;
}

int zkir_exec(const uint8_t* blob, size_t len, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config* cfg, zkir_result** out) {
  return exec_impl(blob, len, inputs, n_inputs, cfg, 0, ~0ull, out);
}

// One GPU's share of a run executed on THIS rank: rows [0, row_begin) are executed untraced (the state at row_begin cannot be had any
// other way: execution is sequential), rows [row_begin, row_end) are traced, uploaded and filled while the interpreter runs.
int zkir_exec_window(const uint8_t* blob, size_t len, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config* cfg, uint64_t row_begin, uint64_t row_end,
                     zkir_result** out) {
  if (cfg && !cfg->enable_execution_trace) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_exec_window: a trace window needs enable_execution_trace"}); return ZKIR_ERR_ARGUMENT; }
  return exec_impl(blob, len, inputs, n_inputs, cfg, row_begin, row_end, out);
}

// The drop-in handle for a ROW SHARD of a finished interpretation (multi-GPU: every device takes one; segment proofs: shards that share
// one row): cut rows [row_begin, row_end) out of `log` (zkir_delta_log_shard), upload that shard to the current device and fill its
// trace.  The result owns the shard; its trace columns hold absolute cycles (cycle_base = row_begin), its witness accessors describe the
// shard's rows.  No host code of the caller needs HIP.
int zkir_exec_shard(const zkir_delta_log* log, uint64_t row_begin, uint64_t row_end, zkir_result** out) {
  if (!out || !log) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_exec_shard: null argument"}); return ZKIR_ERR_ARGUMENT; }
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
    zkir::set_last_error({ZKIR_ERR_DEVICE, "zkir_exec_shard: no usable HIP device (the product path has no CPU fallback)"});
    return ZKIR_ERR_DEVICE;
  }
  zkir_delta_log* sh = nullptr;
  int rc = zkir_delta_log_shard(log, row_begin, row_end, &sh);
  if (rc != ZKIR_OK) return rc;
  zkir_result* r = new zkir_result();
  r->log = sh;
  hipStream_t s = nullptr;
  const uint64_t n = sh->n_rows;
  if (n > 0) {
    const uint32_t T = sh->tile_rows;
    zkir_trace_fill_args a{};
    rc = alloc_trace(r, round_up(n, T), sh->reg_events.size(), sh->tile_ev_off.size(), sh->tile_snap.size());
    if (rc != ZKIR_OK) goto fail_rc;
    HIP_TRY(hipMemcpyAsync(r->d_events, sh->reg_events.data(), sh->reg_events.size() * sizeof(zkir_reg_event), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->d_tile_ev_off, sh->tile_ev_off.data(), sh->tile_ev_off.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->d_tile_snap, sh->tile_snap.data(), sh->tile_snap.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->cols.pc, sh->pc.data(), n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->cols.instruction, sh->inst.data(), n * 4, hipMemcpyHostToDevice, s));
    a.events = (const zkir_reg_event*)r->d_events;
    a.tile_ev_off = (const uint32_t*)r->d_tile_ev_off;
    a.tile_snap = (const uint32_t*)r->d_tile_snap;
    a.n_rows = n; a.cycle_base = sh->cycle_base; a.tile_rows = T; a.n_events = (uint32_t)sh->reg_events.size();
    a.out = r->cols;
    rc = zkir_trace_fill_launch(&a, s);
    if (rc != ZKIR_OK) goto fail_rc;
    HIP_TRY(hipStreamSynchronize(s));
  }
  *out = r;
  return ZKIR_OK;
fail:
  rc = ZKIR_ERR_DEVICE;
fail_rc:
  zkir_result_free(r);
  return rc;
}

void zkir_result_free(zkir_result* r) {
  if (!r) return;
  if (r->d_block) (void)hipFree(r->d_block);
  if (r->d_events) (void)hipFree(r->d_events);
  if (r->d_tile_ev_off) (void)hipFree(r->d_tile_ev_off);
  if (r->d_tile_snap) (void)hipFree(r->d_tile_snap);
  if (r->d_mem) (void)hipFree(r->d_mem);
  if (r->d_rc) (void)hipFree(r->d_rc);
  if (r->d_norm) (void)hipFree(r->d_norm);
  if (r->d_sha) (void)hipFree(r->d_sha);
  delete r->log;
  delete r;
}
const zkir_delta_log* zkir_result_delta_log(const zkir_result* r) { return r->log; }
void zkir_result_stage_ms(const zkir_result* r, float out[4]) { for (int i = 0; i < 4; i++) out[i] = r->stage_ms[i]; }
const zkir_trace_columns* zkir_result_trace(const zkir_result* r) { return &r->cols; }

int zkir_result_copy_column(const zkir_result* r, int field, int reg, void* dst) {
  const uint64_t n = r->log->n_rows;
  if (n == 0) return ZKIR_OK;
  if (field < 0 || field > 7 || (field >= 3 && (reg < 0 || reg > 15))) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_result_copy_column: bad field/reg"}); return ZKIR_ERR_ARGUMENT; }
  const void* src = nullptr; size_t elt = 0;
  const uint64_t off = (uint64_t)reg * r->cols.reg_stride;
  switch (field) {
    case 0: src = r->cols.cycle; elt = 8; break;
    case 1: src = r->cols.pc; elt = 8; break;
    case 2: src = r->cols.instruction; elt = 4; break;
    case 3: src = r->cols.registers + off; elt = 8; break;
    case 4: src = r->cols.bound_bits + off; elt = 4; break;
    case 5: src = r->cols.bound_tag + off; elt = 1; break;
    case 6: src = r->cols.bound_payload + off; elt = 8; break;
    case 7: src = r->cols.reg_state + off; elt = 1; break;
  }
  hipError_t e = hipMemcpy(dst, src, n * elt, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { zkir::set_last_error({ZKIR_ERR_DEVICE, std::string("hipMemcpy D2H: ") + hipGetErrorString(e)}); return ZKIR_ERR_DEVICE; }
  return ZKIR_OK;
}


// ---- ExecutionResult witness streams (vm.rs:54-103) behind the drop-in handle ---------------------------------------------------
// Each accessor uploads the compact side log once, runs the witness.hip kernels on the NULL stream, synchronises and caches the
// device columns in the handle (released by zkir_result_free).  Thread-safe per handle.
#define HIP_TRYW(expr)                                                                                 \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      zkir::set_last_error({ZKIR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)});      \
      if (staging) (void)hipFree(staging);                                                             \
      if (*owned) { (void)hipFree(*owned); *owned = nullptr; }   /* nothing half-built stays in the handle */ \
      return ZKIR_ERR_DEVICE;                                                                          \
    }                                                                                                  \
  } while (0)

static void carve_memops(Carver& cv, uint64_t n, zkir_memop_columns& c) {
  c.address = cv.take<uint64_t>(n); c.value = cv.take<uint64_t>(n); c.timestamp = cv.take<uint64_t>(n); c.bound_payload = cv.take<uint64_t>(n);
  c.bound_bits = cv.take<uint32_t>(n); c.is_write = cv.take<uint8_t>(n); c.width = cv.take<uint8_t>(n); c.bound_tag = cv.take<uint8_t>(n);
}

int zkir_result_memory_trace(zkir_result* r, zkir_memory_witness* out) {
  if (!r || !out) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_result_memory_trace: null argument"}); return ZKIR_ERR_ARGUMENT; }
  std::lock_guard<std::mutex> lk(r->wmu);
  if (!r->mem_built) {
    const zkir_delta_log* log = r->log;
    const uint64_t n = log->mem_events.size(), rows = log->n_rows;
    void* staging = nullptr;
    void** owned = &r->d_mem;
    r->mem = zkir_memory_witness{};
    r->mem.n_ops = n; r->mem.n_rows = rows;
    const size_t per_cols = 4 * padded(n * 8) + padded(n * 4) + 3 * padded(n);
    const size_t bytes = 2 * per_cols + padded((rows + 1) * 8) + 256;
    HIP_TRYW(hipMalloc(&r->d_mem, bytes));
    Carver cv{(unsigned char*)r->d_mem};
    carve_memops(cv, n, r->mem.row_order); carve_memops(cv, n, r->mem.sorted);
    uint64_t* offs = cv.take<uint64_t>(rows + 1);
    r->mem.row_offsets = offs;
    HIP_TRYW(hipMalloc(&staging, n * sizeof(zkir_mem_event) + rows + 256));
    zkir_mem_event* d_ev = (zkir_mem_event*)staging;
    uint8_t* scratch = (uint8_t*)staging + ((n * sizeof(zkir_mem_event) + 255) & ~(size_t)255);
    if (n) HIP_TRYW(hipMemcpyAsync(d_ev, log->mem_events.data(), n * sizeof(zkir_mem_event), hipMemcpyHostToDevice, nullptr));
    // two passes over the events: expansion + CSR offsets + shape flags, then the sort by rank
    int rc = zkir_memops_expand_csr_launch(d_ev, n, rows, log->cycle_base, &r->mem.row_order, offs, scratch, nullptr);
    if (rc == ZKIR_OK) rc = zkir_memops_sort_prepared_launch(d_ev, n, log->cycle_base, offs, scratch, &r->mem.sorted, nullptr);
    if (rc != ZKIR_OK) { (void)hipFree(staging); (void)hipFree(*owned); *owned = nullptr; return rc; }
    HIP_TRYW(hipStreamSynchronize(nullptr));
    (void)hipFree(staging);
    r->mem_built = true;
  }
  *out = r->mem;
  return ZKIR_OK;
}

int zkir_result_range_check_witnesses(zkir_result* r, zkir_range_check_witness* out) {
  if (!r || !out) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_result_range_check_witnesses: null argument"}); return ZKIR_ERR_ARGUMENT; }
  std::lock_guard<std::mutex> lk(r->wmu);
  if (!r->rc_built) {
    const zkir_delta_log* log = r->log;
    const uint64_t n = log->rc_events.size();
    void* staging = nullptr;
    void** owned = &r->d_rc;
    r->rc = zkir_range_check_witness{};
    r->rc.n_checks = n; r->rc.n_witnesses = log->rc_offsets.size() - 1; r->rc.witness_offsets = log->rc_offsets.data();
    r->rc.witness_cycles = log->rc_cycles.data();
    r->rc.chunk_bits = log->rc_chunk_bits; r->rc.chunk_stride = (n + 127) & ~(uint64_t)127;
    const size_t bytes = 2 * padded(n * 8) + padded(4 * r->rc.chunk_stride * 2) + padded(sizeof(uint32_t) << log->rc_chunk_bits) + 256;
    HIP_TRYW(hipMalloc(&r->d_rc, bytes));
    Carver cv{(unsigned char*)r->d_rc};
    uint64_t* value = cv.take<uint64_t>(n); uint64_t* pc = cv.take<uint64_t>(n);
    uint16_t* chunks = cv.take<uint16_t>(4 * r->rc.chunk_stride); uint32_t* mult = cv.take<uint32_t>((size_t)1 << log->rc_chunk_bits);
    r->rc.value = value; r->rc.pc = pc; r->rc.chunks = chunks; r->rc.multiplicity = mult;
    HIP_TRYW(hipMalloc(&staging, n * sizeof(zkir_rc_event) + 256));
    if (n) HIP_TRYW(hipMemcpyAsync(staging, log->rc_events.data(), n * sizeof(zkir_rc_event), hipMemcpyHostToDevice, nullptr));
    const int rc = zkir_range_check_expand_launch((const zkir_rc_event*)staging, n, log->rc_chunk_bits, value, pc, chunks, r->rc.chunk_stride, mult, nullptr);
    if (rc != ZKIR_OK) { (void)hipFree(staging); (void)hipFree(*owned); *owned = nullptr; return rc; }
    HIP_TRYW(hipStreamSynchronize(nullptr));
    (void)hipFree(staging);
    r->rc_built = true;
  }
  *out = r->rc;
  return ZKIR_OK;
}

int zkir_result_normalization_witnesses(zkir_result* r, zkir_normalization_witness* out) {
  if (!r || !out) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_result_normalization_witnesses: null argument"}); return ZKIR_ERR_ARGUMENT; }
  std::lock_guard<std::mutex> lk(r->wmu);
  if (!r->norm_built) {
    const zkir_delta_log* log = r->log;
    const uint64_t n = log->norm_events.size();
    void* staging = nullptr;
    void** owned = &r->d_norm;
    r->norm = zkir_normalization_witness{};
    r->norm.n_events = n;
    HIP_TRYW(hipMalloc(&r->d_norm, 4 * padded(n * 8) + 4 * padded(n * 4) + 2 * padded(n) + 256));
    Carver cv{(unsigned char*)r->d_norm};
    zkir_norm_columns& c = r->norm.columns;
    c.cycle = cv.take<uint64_t>(n); c.pc = cv.take<uint64_t>(n); c.accumulated0 = cv.take<uint64_t>(n); c.accumulated1 = cv.take<uint64_t>(n);
    c.normalized0 = cv.take<uint32_t>(n); c.normalized1 = cv.take<uint32_t>(n); c.carry0 = cv.take<uint32_t>(n); c.carry1 = cv.take<uint32_t>(n);
    c.reg = cv.take<uint8_t>(n); c.opcode = cv.take<uint8_t>(n);
    HIP_TRYW(hipMalloc(&staging, n * sizeof(zkir_norm_event) + 256));
    if (n) HIP_TRYW(hipMemcpyAsync(staging, log->norm_events.data(), n * sizeof(zkir_norm_event), hipMemcpyHostToDevice, nullptr));
    const int rc = zkir_norm_expand_launch((const zkir_norm_event*)staging, n, &c, nullptr);
    if (rc != ZKIR_OK) { (void)hipFree(staging); (void)hipFree(*owned); *owned = nullptr; return rc; }
    HIP_TRYW(hipStreamSynchronize(nullptr));
    (void)hipFree(staging);
    r->norm_built = true;
  }
  *out = r->norm;
  return ZKIR_OK;
}

int zkir_result_sha256_witnesses(zkir_result* r, zkir_sha256_witness* out) {
  if (!r || !out) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_result_sha256_witnesses: null argument"}); return ZKIR_ERR_ARGUMENT; }
  std::lock_guard<std::mutex> lk(r->wmu);
  if (!r->sha_built) {
    const zkir_delta_log* log = r->log;
    const uint64_t n = log->sha_blocks.size();
    void* staging = nullptr;
    void** owned = &r->d_sha;
    r->sha = zkir_sha256_witness{};
    r->sha.n_blocks = n; r->sha.stride = (n + 63) & ~(uint64_t)63;
    HIP_TRYW(hipMalloc(&r->d_sha, padded(608 * r->sha.stride * 4) + padded(n * 8) + 256));
    Carver cv{(unsigned char*)r->d_sha};
    uint32_t* cols = cv.take<uint32_t>(608 * r->sha.stride); uint64_t* ts = cv.take<uint64_t>(n);
    r->sha.columns = cols; r->sha.timestamps = ts;
    HIP_TRYW(hipMalloc(&staging, n * sizeof(zkir_sha_block) + 256));
    if (n) HIP_TRYW(hipMemcpyAsync(staging, log->sha_blocks.data(), n * sizeof(zkir_sha_block), hipMemcpyHostToDevice, nullptr));
    const int rc = zkir_sha256_chip_launch((const zkir_sha_block*)staging, n, cols, r->sha.stride, ts, nullptr);
    if (rc != ZKIR_OK) { (void)hipFree(staging); (void)hipFree(*owned); *owned = nullptr; return rc; }
    HIP_TRYW(hipStreamSynchronize(nullptr));
    (void)hipFree(staging);
    r->sha_built = true;
  }
  *out = r->sha;
  return ZKIR_OK;
}

int zkir_host_to_device(void* device_dst, const void* host_src, size_t bytes, void* stream) {
  if (bytes == 0) return ZKIR_OK;
  const hipError_t e = hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) { zkir::set_last_error({ZKIR_ERR_DEVICE, std::string("hipMemcpy H2D: ") + hipGetErrorString(e)}); return ZKIR_ERR_DEVICE; }
  return ZKIR_OK;
}

int zkir_device_to_host(void* host_dst, const void* device_src, size_t bytes) {
  if (bytes == 0) return ZKIR_OK;
  const hipError_t e = hipMemcpy(host_dst, device_src, bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { zkir::set_last_error({ZKIR_ERR_DEVICE, std::string("hipMemcpy D2H: ") + hipGetErrorString(e)}); return ZKIR_ERR_DEVICE; }
  return ZKIR_OK;
}

// ---- prove(result, params) -> proof bytes ------------------------------------------------------------------------------
// The whole-run handle of zkir_exec goes in, the proof comes out as BYTES (little-endian u32 words, malloc'ed: free with zkir_proof_bytes_free): the context for the
// run's padded size is made here, the public inputs come from the handle's own log / program / input tape, the mode and the FRI parameters from `params` (NULL =
// mode 0, 50 queries, 12 grinding bits).  A thin wrapper over zkir_public_inputs_of + zkir_public_inputs_set_params + zkir_prove: callers that prove many runs keep a
// context and call those (the context's tables and workspace are what a repeated proof reuses).
int zkir_prove_result(const zkir_result* res, const zkir_prover_params* params, uint8_t** proof, size_t* proof_len) {
  if (!res || !proof || !proof_len) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove_result: null argument"}); return ZKIR_ERR_ARGUMENT; }
  *proof = nullptr; *proof_len = 0;
  if (!res->whole_run || !res->log || res->program.empty() || !res->cols.cycle) {
    zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_prove_result: the handle must be a whole traced run made by zkir_exec (enable_execution_trace; a window / shard is a segment: zkir_prove)"});
    return ZKIR_ERR_ARGUMENT;
  }
  const zkir_prover_params dflt{0, 0, 0};
  const zkir_prover_params* pr = params ? params : &dflt;
  zkir_public_inputs pub;
  int rc = zkir_public_inputs_of(res->log, res->program.data(), res->program.size(), res->inputs.data(), res->inputs.size(), pr->mode, &pub);
  if (rc != ZKIR_OK) return rc;
  rc = zkir_public_inputs_set_params(&pub, pr);
  if (rc != ZKIR_OK) return rc;
  zkir_stark_ctx* ctx = nullptr;
  rc = zkir_stark_ctx_create(zkir_padded_log_n(pub.n_real), 1, &ctx);
  if (rc != ZKIR_OK) return rc;
  uint32_t* words = nullptr; uint64_t n_words = 0;
  rc = zkir_prove(ctx, &res->cols, &pub, &words, &n_words, nullptr, nullptr);
  zkir_stark_ctx_free(ctx);
  if (rc != ZKIR_OK) return rc;
  *proof = reinterpret_cast<uint8_t*>(words);                     // x86-64 / gfx950 hosts are little-endian: the words ARE the bytes
  *proof_len = (size_t)n_words * 4;
  return ZKIR_OK;
}
void zkir_proof_bytes_free(uint8_t* proof) { zkir_proof_free(reinterpret_cast<uint32_t*>(proof)); }

}  // extern "C"
