// abi.hip — extern "C" entry points of include/zkir_amd.h: delta-log accessors and the drop-in
// zkir_exec (VM::new + VM::run replacement: host interpreter -> H2D -> K1 trace fill).
#include <hip/hip_runtime.h>

#include "../../include/zkir_amd.h"
#include "host.h"

namespace zkir {
static thread_local std::string g_last_error;
void set_last_error(const Status& st) { g_last_error = st.msg; }
}  // namespace zkir

struct zkir_result {
  zkir_delta_log* log = nullptr;
  zkir_trace_columns cols{};
  void* d_events = nullptr;
  void* d_tile_ev_off = nullptr;
  void* d_tile_snap = nullptr;
  void* d_block = nullptr;      // one allocation holding every trace column
  uint64_t cap_rows = 0;
};

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
  const uint64_t T = src->tile_rows;
  if (row_begin > row_end || row_end > src->n_rows || row_begin % T != 0 || src->cycle_base != 0) {
    zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_delta_log_shard: need 0 <= row_begin <= row_end <= n_rows, row_begin % tile_rows == 0, unsharded source"});
    return ZKIR_ERR_ARGUMENT;
  }
  zkir_delta_log* d = new zkir_delta_log();
  d->cycles = src->cycles; d->halt_kind = src->halt_kind; d->halt_code = src->halt_code; d->outputs = src->outputs;
  d->tile_rows = src->tile_rows; d->rc_chunk_bits = src->rc_chunk_bits;
  d->n_rows = row_end - row_begin; d->cycle_base = row_begin;
  d->rc_offsets.push_back(0);
  if (d->n_rows == 0) { d->tile_ev_off.push_back(0); *out = d; return ZKIR_OK; }
  const uint64_t t0 = row_begin / T, t1 = (row_end + T - 1) / T;                   // tiles [t0, t1)
  const uint32_t e0 = src->tile_ev_off[t0];
  // events up to the end of the last tile; for an interior cut that is tile_ev_off[t1], for the run's tail all remaining
  const uint32_t e1 = src->tile_ev_off[t1];
  d->pc.append(src->pc.data() + row_begin, d->n_rows);
  d->inst.append(src->inst.data() + row_begin, d->n_rows);
  for (int r = 0; r < 16; r++) {                                                   // snapshot at row_begin
    zkir_reg_event e = src->reg_events[src->tile_snap[t0 * 16 + r]];
    e.vis = 0;
    d->reg_events.push(e);
  }
  for (uint32_t k = e0; k < e1; k++) { zkir_reg_event e = src->reg_events[k]; e.vis -= (uint32_t)row_begin; d->reg_events.push(e); }
  for (uint64_t t = t0; t <= t1; t++) d->tile_ev_off.push_back(src->tile_ev_off[t] - e0 + 16);
  for (uint64_t t = t0; t < t1; t++)
    for (int r = 0; r < 16; r++) { const uint32_t i = src->tile_snap[t * 16 + r]; d->tile_snap.push_back(i < e0 ? (uint32_t)r : i - e0 + 16); }
  for (size_t k = 0; k < src->mem_events.size(); k++) {
    zkir_mem_event m = src->mem_events[k];
    if (m.row >= row_begin && m.row < row_end) { m.row -= (uint32_t)row_begin; d->mem_events.push(m); }
  }
  for (size_t k = 0; k < src->norm_events.size(); k++) if (src->norm_events[k].cycle >= row_begin && src->norm_events[k].cycle < row_end) d->norm_events.push(src->norm_events[k]);
  for (size_t k = 0; k < src->sha_blocks.size(); k++) if (src->sha_blocks[k].timestamp >= row_begin && src->sha_blocks[k].timestamp < row_end) d->sha_blocks.push(src->sha_blocks[k]);
  *out = d;
  return ZKIR_OK;
}
uint64_t zkir_delta_log_cycle_base(const zkir_delta_log* l) { return l->cycle_base; }

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
uint32_t zkir_delta_log_rc_chunk_bits(const zkir_delta_log* l) { return l->rc_chunk_bits; }
size_t zkir_delta_log_n_norm_events(const zkir_delta_log* l) { return l->norm_events.size(); }
const zkir_norm_event* zkir_delta_log_norm_events(const zkir_delta_log* l) { return l->norm_events.data(); }
size_t zkir_delta_log_n_sha_blocks(const zkir_delta_log* l) { return l->sha_blocks.size(); }
const zkir_sha_block* zkir_delta_log_sha_blocks(const zkir_delta_log* l) { return l->sha_blocks.data(); }

// ---- drop-in layer ------------------------------------------------------------------------------
static inline uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

int zkir_exec(const uint8_t* blob, size_t len, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config* cfg, zkir_result** out) {
  if (!out) { zkir::set_last_error({ZKIR_ERR_ARGUMENT, "zkir_exec: null out"}); return ZKIR_ERR_ARGUMENT; }
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
    zkir::set_last_error({ZKIR_ERR_DEVICE, "zkir_exec: no usable HIP device (the product path has no CPU fallback)"});
    return ZKIR_ERR_DEVICE;
  }
  zkir_delta_log* log = nullptr;
  int rc = zkir_interpret(blob, len, inputs, n_inputs, cfg, 0, &log);
  if (rc != ZKIR_OK) return rc;
  zkir_result* r = new zkir_result();
  r->log = log;
  const uint64_t n = log->n_rows;
  if (n > 0) {
    const uint32_t T = log->tile_rows;
    const uint64_t cap = round_up(n, T);
    r->cap_rows = cap;
    // column block: cycle 8 | pc 8 | inst 4 | regs 128 | bits 64 | tag 16 | payload 128 | state 16  = 372 B per (padded) row
    const uint64_t bytes = cap * 372;
    unsigned char* base = nullptr;
    hipStream_t s = nullptr;
    zkir_trace_fill_args a{};
    HIP_TRY(hipMalloc(&r->d_block, bytes));
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
    HIP_TRY(hipMalloc(&r->d_events, log->reg_events.size() * sizeof(zkir_reg_event)));
    HIP_TRY(hipMalloc(&r->d_tile_ev_off, log->tile_ev_off.size() * 4));
    HIP_TRY(hipMalloc(&r->d_tile_snap, log->tile_snap.size() * 4));
    HIP_TRY(hipMemcpyAsync(r->d_events, log->reg_events.data(), log->reg_events.size() * sizeof(zkir_reg_event), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->d_tile_ev_off, log->tile_ev_off.data(), log->tile_ev_off.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->d_tile_snap, log->tile_snap.data(), log->tile_snap.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(r->cols.pc, log->pc.data(), n * 8, hipMemcpyHostToDevice, s));           // pc / instruction columns arrive in final form
    HIP_TRY(hipMemcpyAsync(r->cols.instruction, log->inst.data(), n * 4, hipMemcpyHostToDevice, s));
    a.events = (const zkir_reg_event*)r->d_events;
    a.tile_ev_off = (const uint32_t*)r->d_tile_ev_off;
    a.tile_snap = (const uint32_t*)r->d_tile_snap;
    a.n_rows = n; a.cycle_base = 0; a.tile_rows = T; a.n_events = (uint32_t)log->reg_events.size();
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
  delete r->log;
  delete r;
}
const zkir_delta_log* zkir_result_delta_log(const zkir_result* r) { return r->log; }
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

}  // extern "C"
