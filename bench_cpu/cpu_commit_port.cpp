// cpu_commit_port.cpp — CPU PORT of the commit stage (LDE + Poseidon2 Merkle) for bench.py's `cpu_baseline` leg ONLY.
//
// BENCH INFRASTRUCTURE (kept outside oracle/: it compiles product headers).  The commit stage has no counterpart in the reference (no prover there), so this is not
// "the reference path": it is the same self-defined computation the GPU step performs, written the way a CPU implementation would
// be — Montgomery arithmetic (the product's babybear.h / poseidon2.h compiled for the host), iterative radix-2 NTTs with twiddle
// tables, std::thread over columns and leaves — so that the GPU number has a same-size, same-work CPU figure next to it.  The
// parity tests never use this file: their oracle is stark_oracle.cpp (naive arithmetic); tests/test_stark_oracle.py checks that the
// root computed here equals that oracle's.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../zkir_amd/csrc/babybear.h"
#include "../zkir_amd/csrc/poseidon2.h"

namespace {

void parallel_for(size_t n, int threads, const std::function<void(size_t, size_t)>& f) {
  if (threads <= 1 || n < 2) { f(0, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n + threads - 1) / threads;
  for (int t = 0; t < threads; t++) {
    const size_t a = (size_t)t * per, b = a + per < n ? a + per : n;
    if (a >= b) break;
    th.emplace_back(f, a, b);
  }
  for (auto& x : th) x.join();
}

// in-place NTT of Montgomery values, natural in / natural out; tw[k] = w^k (Montgomery), k < n/2
void ntt(uint32_t* a, int log_n, const uint32_t* tw) {
  const size_t n = (size_t)1 << log_n;
  for (size_t i = 0; i < n; i++) {                                           // bit reversal
    size_t j = 0;
    for (int b = 0; b < log_n; b++) j |= ((i >> b) & 1) << (log_n - 1 - b);
    if (i < j) { const uint32_t t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  for (int s = 1; s <= log_n; s++) {
    const size_t m = (size_t)1 << s, h = m >> 1, step = n >> s;
    for (size_t k = 0; k < n; k += m)
      for (size_t j = 0; j < h; j++) {
        const uint32_t t = bb::mont_mul(a[k + j + h], tw[j * step]), u = a[k + j];
        a[k + j] = bb::add(u, t); a[k + j + h] = bb::sub(u, t);
      }
  }
}
std::vector<uint32_t> powers_m(uint32_t w, size_t count) {                    // w^k in Montgomery form
  std::vector<uint32_t> t(count);
  uint32_t cur = bb::R1; const uint32_t wm = bb::to_mont(w);
  for (size_t k = 0; k < count; k++) { t[k] = cur; cur = bb::mont_mul(cur, wm); }
  return t;
}

}  // namespace

extern "C" {

// M: column-major canonical main-trace matrix [width][N] (N = 2^log_n).  Computes the LDE on 31 * <w_2N> and the Poseidon2 Merkle root
// over its 2N rows with `threads` host threads.  seconds[0] = LDE, seconds[1] = Merkle.  Returns nothing else: a timing harness.
void so_commit_port(const uint32_t* M, int width, int log_n, int threads, uint32_t root4[4], double seconds[2]) {
  using clk = std::chrono::steady_clock;
  const size_t N = (size_t)1 << log_n, N2 = 2 * N;
  static const p2::Consts consts = [] { p2::Consts c; p2::generate(c); return c; }();
  const uint32_t wn = bb::root_of_unity(log_n), w2n = bb::root_of_unity(log_n + 1);
  const std::vector<uint32_t> tw_inv = powers_m(bb::inv(wn), N / 2 ? N / 2 : 1), tw_fwd = powers_m(w2n, N), shift = powers_m(bb::GEN, N);
  const uint32_t ninv_m = bb::to_mont(bb::inv((uint32_t)(N % bb::P)));
  std::vector<uint32_t> L((size_t)width * N2);
  auto t0 = clk::now();
  parallel_for((size_t)width, threads, [&](size_t a, size_t b) {
    std::vector<uint32_t> buf(N2);
    for (size_t k = a; k < b; k++) {
      for (size_t i = 0; i < N; i++) buf[i] = bb::to_mont(M[k * N + i]);
      ntt(buf.data(), log_n, tw_inv.data());                                // coefficients * N
      for (size_t i = 0; i < N; i++) buf[i] = bb::mont_mul(bb::mont_mul(buf[i], ninv_m), shift[i]);
      memset(buf.data() + N, 0, N * 4);
      ntt(buf.data(), log_n + 1, tw_fwd.data());
      for (size_t i = 0; i < N2; i++) L[k * N2 + i] = bb::from_mont(buf[i]);
    }
  });
  seconds[0] = std::chrono::duration<double>(clk::now() - t0).count();
  t0 = clk::now();
  std::vector<uint32_t> cur(4 * N2);
  parallel_for(N2, threads, [&](size_t a, size_t b) {
    for (size_t j = a; j < b; j++) {
      uint32_t s[p2::T] = {0};
      for (int off = 0; off < width; off += p2::RATE) {
        for (int i = 0; i < p2::RATE && off + i < width; i++) s[i] = bb::to_mont(L[(size_t)(off + i) * N2 + j]);
        p2::permute(s, consts);
      }
      for (int i = 0; i < 4; i++) cur[4 * j + i] = bb::from_mont(s[i]);
    }
  });
  for (size_t m = N2; m > 1; m >>= 1) {
    std::vector<uint32_t> nxt(4 * (m / 2));
    parallel_for(m / 2, m / 2 >= 4096 ? threads : 1, [&](size_t a, size_t b) {
      for (size_t i = a; i < b; i++) {
        uint32_t s[p2::T] = {0};
        for (int t = 0; t < 8; t++) s[t] = bb::to_mont(cur[8 * i + t]);
        p2::permute(s, consts);
        for (int t = 0; t < 4; t++) nxt[4 * i + t] = bb::from_mont(s[t]);
      }
    });
    cur.swap(nxt);
  }
  seconds[1] = std::chrono::duration<double>(clk::now() - t0).count();
  memcpy(root4, cur.data(), 16);
}

}  // extern "C"
