// host.h — internal host-side types of libzkir_amd (not part of the C ABI).
#pragma once

#include <sys/mman.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zkir_amd.h"

#define ZKIR_DEFAULT_TILE_ROWS 1024u

namespace zkir {

struct Status {
  int code = ZKIR_OK;
  std::string msg;
  bool ok() const { return code == ZKIR_OK; }
};

struct BoundT {            // ValueBound, zkir-spec/src/bound.rs:116-121
  uint32_t max_bits;
  uint8_t tag;
  uint64_t payload;
};

// Recycling of the big log blocks (interp.cpp): a 2^20-row run writes ~50 MB of logs once, front to back, and with 4 KiB pages
// (no THP in containers) first-touch page faults cost more than the interpretation itself (64 ms sys vs 53 ms user at 2^22 rows).
// Freed blocks >= 8 MiB are parked (at most 8 blocks / 4 GiB) and handed to the next run already mapped.
// Round 6: the big blocks are PINNED (hipHostMalloc) whenever the process has a device — the delta log is written once by the interpreter and read
// once by the DMA engine, and a copy out of pageable memory is staged by the runtime through its own pinned bounce buffers at ~4.5 GB/s (10 ms for the 46 MB of a 2^20-row
// run) where the link moves 50+.  Pinning costs a fraction of a millisecond per MB, ONCE: the pool recycles the blocks.  Without a device (the host-only entry points
// zkir_interpret / zkir_verify on a CPU box) and with ZKIR_PIN_LOG=0 the blocks are plain malloc memory as before.
struct Block { void* p = nullptr; size_t bytes = 0; bool pinned = false; };
Block block_acquire(size_t min_bytes);                        // a parked block of min_bytes .. 4 min_bytes, else a fresh one (pinned if possible); p == nullptr: out of memory
void block_release(const Block& b);                           // parks the block (or frees it: pool full)
void* pinned_alloc(size_t bytes);                             // abi.hip: hipHostMalloc; nullptr when there is no device (remembered: asked once) or no pinned memory left
void pinned_free(void* p);
constexpr size_t BLOCK_POOL_MIN = 8u << 20;

// Growable POD buffer: small ones live in realloc memory, big ones (>= BLOCK_POOL_MIN) in pool blocks.
template <typename T>
class Buf {
 public:
  Buf() = default;
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  ~Buf() { if (blk_.p) block_release(blk_); else free(p_); }
  void reserve(size_t n) {
    if (n <= cap_) return;
    if (n * sizeof(T) >= BLOCK_POOL_MIN) {
      const Block b = block_acquire(n * sizeof(T));
      if (!b.p) throw std::bad_alloc();
      if (n_) memcpy(b.p, p_, n_ * sizeof(T));
      if (blk_.p) block_release(blk_); else free(p_);
      blk_ = b; p_ = (T*)b.p; cap_ = b.bytes / sizeof(T);
      return;
    }
    void* q = realloc(p_, n * sizeof(T));
    if (!q) throw std::bad_alloc();
    p_ = (T*)q; cap_ = n;
  }
  inline void push(const T& v) {
    if (n_ == cap_) reserve(cap_ ? cap_ * 2 : 1024);
    p_[n_++] = v;
  }
  // bulk producers (the interpreter's hot loop): room for `extra` more elements at the end, written through the returned pointer
  // and made part of the buffer with commit(); growth is geometric
  inline T* grow(size_t extra) {
    if (n_ + extra > cap_) reserve(cap_ * 2 > n_ + extra ? cap_ * 2 : n_ + extra);
    return p_ + n_;
  }
  inline void commit(size_t k) { n_ += k; }
  void append(const T* src, size_t n) {
    if (n == 0) return;
    reserve(n_ + n);
    memcpy(p_ + n_, src, n * sizeof(T));
    n_ += n;
  }
  void clear() { n_ = 0; }
  size_t size() const { return n_; }
  size_t capacity() const { return cap_; }
  bool pinned() const { return blk_.p && blk_.pinned; }
  const T* data() const { return p_; }
  T* data() { return p_; }
  const T& operator[](size_t i) const { return p_[i]; }

 private:
  T* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
  Block blk_;                                                  // set when p_ is a pool block
};

// Pinned host memory for the prover's large host <-> device copies: bump-allocated per call (reset()), the blocks kept by the owner (a stark context) until it goes.
// WHY (measured, profiles/r04w_pageable_copy_stall.txt): the HIP runtime serves a pageable copy of more than 1 MiB by pinning the caller's pages in place.  When such a block
// is freed afterwards and glibc hands it back to the kernel (munmap of a large block, or a heap trim), the KFD's MMU notifier evicts the process's queues, and the next
// submission waits for their restore: +25 ms on every mode-3 proof (its memory section is 1.8 MB at 65536 cells), none on modes 0-2 (nothing above 1 MiB).  So no pageable
// block that can exceed that size is handed to a copy: it goes through here.  (hip calls inside: defined in memcheck.hip.)
struct HostPin {
  std::vector<std::pair<unsigned char*, size_t>> blocks;
  size_t cur = 0, off = 0;
  HostPin() = default;
  HostPin(const HostPin&) = delete;
  HostPin& operator=(const HostPin&) = delete;
  ~HostPin();
  void reset() { cur = 0; off = 0; }
  void* take(size_t bytes);                                    // 256-byte aligned; nullptr when the host has no more pinned memory to give
  template <typename T> T* take_n(size_t count) { return (T*)take(count * sizeof(T)); }
};

struct ProgramView {       // parsed Program blob (zkir-spec/src/program.rs:318-346); points into the caller's bytes
  uint8_t limb_bits = 20, data_limbs = 2, addr_limbs = 2;
  uint32_t entry_point = 0x1000;
  const uint8_t* code = nullptr; uint64_t n_code_words = 0;
  const uint8_t* data = nullptr; uint64_t n_data = 0;
};

}  // namespace zkir

// The opaque C-ABI handle: everything the sequential interpreter produced, in host memory.
struct zkir_delta_log {
  uint64_t cycles = 0;
  int halt_kind = ZKIR_HALT_EBREAK;
  uint64_t halt_code = 0;
  uint64_t n_rows = 0;
  uint64_t cycle_base = 0;       // TraceRow.cycle of row 0 (non-zero only for shards and trace windows)
  bool window_open = false;      // the interpretation stopped at the end of its trace window, not at a halt (zkir_interpret_window)
  uint32_t tile_rows = ZKIR_DEFAULT_TILE_ROWS;
  uint32_t rc_chunk_bits = 10;
  std::vector<uint64_t> outputs;
  zkir::Buf<uint64_t> pc;
  zkir::Buf<uint32_t> inst;
  zkir::Buf<zkir_reg_event> reg_events;
  std::vector<uint32_t> tile_ev_off;
  std::vector<uint32_t> tile_snap;
  zkir::Buf<zkir_mem_event> mem_events;
  zkir::Buf<zkir_rc_event> rc_events;
  std::vector<uint64_t> rc_offsets;
  std::vector<uint64_t> rc_cycles;   // cycle of the checkpoint that flushed witness k (vm.rs:316-344); row sharding cuts by it
  zkir::Buf<zkir_norm_event> norm_events;
  zkir::Buf<zkir_sha_block> sha_blocks;
};

namespace zkir {
using DeltaLog = ::zkir_delta_log;

// Progress of a running interpretation, for a consumer on another thread (zkir_exec's uploader).  The interpreter publishes, every
// few tiles, how many COMPLETE tiles / rows / register events the log holds; nothing it has published is ever moved or rewritten as
// long as `stable` stays true (buffers are pre-reserved; if one would have to grow, the interpreter clears `stable` and waits for
// `consumer_idle` before reallocating — the consumer then abandons streaming).
struct Progress {
  std::atomic<uint64_t> tiles{0}, rows{0}, events{0};
  std::atomic<bool> stable{true}, consumer_idle{true}, finished{false};
};

Status parse_program(const uint8_t* blob, size_t len, ProgramView& pv);
// win_begin / win_end: trace window in absolute rows (default: the whole run).  Rows before the window are executed untraced.
Status interpret(const uint8_t* blob, size_t len, const uint64_t* inputs, size_t n_inputs, const zkir_vm_config& cfg, uint32_t tile_rows, DeltaLog& log,
                 Progress* progress = nullptr, uint64_t win_begin = 0, uint64_t win_end = ~0ull);

// hashes.cpp — the digests behind syscalls 3/5/6 (zkir-runtime/src/crypto.rs uses sha2 / sha3::Keccak256 / blake3)
void sha256(const uint8_t* data, size_t len, uint32_t out_be_words[8]);
void keccak256(const uint8_t* data, size_t len, uint8_t out[32]);
void blake3(const uint8_t* data, size_t len, uint8_t out[32]);

void set_last_error(const Status& st);

// memcheck.hip — the memory witness of AIR mode 3 on the device: address-major sort of the accesses (rocPRIM's radix sort) + a segmented scan per cell (this repo's kernels)
size_t memcheck_scratch_bytes(uint64_t n_real, uint64_t image_len);
int memcheck_device(const zkir_trace_columns* trace, uint64_t n_real, const uint8_t* blob, size_t blob_len, void* scratch, size_t scratch_bytes, uint64_t* mem_old, uint32_t* mem_told,
                    std::vector<uint64_t>& cell_addr, std::vector<uint64_t>& cell_bytes, std::vector<uint32_t>& cell_time, HostPin& pin, void* hip_stream);

// ntt.hip — coset LDE of `width` columns (device pointers; tables owned by zkir_stark_ctx, all in Montgomery form)
struct LdeTables {
  int log_n;
  const uint32_t* tw_inv;     // w_N^-k, k < N/2
  const uint32_t* tw_fwd;     // w_{2N}^k, k < N
  const uint32_t* g_lo;       // g^k / N, k < 1024
  const uint32_t* g_hi;       // g^(1024 k)
  const uint32_t* small_inv;  // w_{2^Bm}^-k, k < 2^(Bm-1)      (Bm = min(log_n, 10))
  const uint32_t* small_fwd;  // w_{2^(Bm+1)}^k, k < 2^Bm
};
void lde_run(const LdeTables& t, uint32_t* in, uint32_t n_blocks, uint32_t* out, void* hip_stream);   // matrices in the B8 layout: n_blocks x [rows][8]
bool strided_variant_run(const LdeTables& t, uint32_t* data, uint32_t n_blocks, int variant, bool dit, void* hip_stream);   // experiment: one strided pass, chosen tile geometry (timing only)
// experiment (profiles/HISTORY.md, round 4): blocks 0 and 1 of `in` generated from the trace inside the first inverse pass; false = not applicable at this size (nothing launched)
bool lde_run_fused01(const LdeTables& t, const zkir_trace_columns* trace, uint64_t n_real, uint32_t* in, uint32_t n_blocks, uint32_t* out, void* hip_stream);
}  // namespace zkir
