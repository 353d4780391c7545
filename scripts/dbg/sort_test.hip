#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)1 << atoi(argv[1]) : 1 << 18;
  const int begin_bit = argc > 2 ? atoi(argv[2]) : 26;
  std::vector<uint64_t> k(n);
  for (size_t i = 0; i < n; i++) { const size_t it = i / 16, pos = i % 16; k[i] = (pos >= 1 && pos <= 6 && i + 1 < n) ? (((uint64_t)(0x20000 + it % 8192)) << 26) | i : ~0ull; }
  uint64_t *a, *b; void* tmp; size_t tb = 0;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
  hipMemcpy(a, k.data(), n * 8, hipMemcpyHostToDevice);
  rocprim::radix_sort_keys(nullptr, tb, a, b, n, begin_bit, 64, 0);
  hipMalloc(&tmp, tb);
  hipError_t e = rocprim::radix_sort_keys(tmp, tb, a, b, n, begin_bit, 64, 0);
  hipDeviceSynchronize();
  std::vector<uint64_t> o(n);
  hipMemcpy(o.data(), b, n * 8, hipMemcpyDeviceToHost);
  size_t bad_order = 0, bad_stable = 0;
  for (size_t i = 1; i < n; i++) { if ((o[i] >> 26) < (o[i - 1] >> 26)) bad_order++; if ((o[i] >> 26) == (o[i - 1] >> 26) && o[i] != ~0ull && (o[i] & 0x3FFFFFF) < (o[i - 1] & 0x3FFFFFF)) bad_stable++; }
  printf("n=2^%d begin_bit=%d err=%d tmp=%zu: order violations %zu, stability violations %zu; first keys %llx %llx %llx\n", argc > 1 ? atoi(argv[1]) : 18, begin_bit, (int)e, tb, bad_order, bad_stable,
         (unsigned long long)o[0], (unsigned long long)o[1], (unsigned long long)o[2]);
  return 0;
}
