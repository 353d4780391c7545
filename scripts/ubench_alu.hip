// ubench_alu.hip — issue rate of the integer VALU instructions the field arithmetic is made of, on gfx950.
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_alu scripts/ubench_alu.hip && /tmp/ubench_alu
// Each kernel runs ITER x 8 independent copies of one instruction per lane (8 chains hide the result latency), with enough
// waves per SIMD to keep the issue port busy; the report is SIMD cycles per wave64 instruction at the measured clock.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr int ITER = 4096;

#define CHAIN8(STMT) \
  for (int i = 0; i < ITER; i++) { STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) }

__global__ void k_add(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(c));
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_min(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[j]) : "v"(c));
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_mul_lo(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[j]) : "v"(c));
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_mul_hi(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[j]) : "v"(c));
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_mul_u24(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[j]) : "v"(c));
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_mad64(uint32_t* out, uint32_t c) {
  uint64_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
  uint32_t a = threadIdx.x | 1;
#define S(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[j]) : "v"(a), "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint64_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(r ^ (r >> 32));
}
__global__ void k_lshl_add64(uint32_t* out, uint32_t c) {
  uint64_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
  uint64_t a = threadIdx.x | 1;
#define S(j) asm volatile("v_lshl_add_u64 %0, %1, 1, %0" : "+v"(x[j]) : "v"(a));
  CHAIN8(S)
#undef S
  uint64_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(r ^ (r >> 32)) + c;
}
__global__ void k_add3(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(c));
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_alignbit(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(x[j]) : "v"(c));
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void k_sub(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_and(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_xor(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_lshl(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_max(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_mov(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_mov_b32 %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_addco(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_subco(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_cndmask(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_cmp(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_mini(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_bfe(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_addf(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_fmaf(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_minf(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_lshladd(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_addlshl(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_mad24(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_sad(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_sad_u32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_med3(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_med3_u32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_min3(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_pkadd(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_pkmin(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_subrev(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// ---- 64-bit (register pair) operand kernels: double-precision and packed-f32 instructions ----
#define K64(NAME, ASM) \
__global__ void NAME(uint32_t* out, uint32_t c) { \
  double x[8]; \
  for (int j = 0; j < 8; j++) x[j] = 1.0 + (double)(threadIdx.x + j) * 1e-9; \
  const double cc = 1.0 + (double)c * 1e-12; \
  for (int i = 0; i < ITER; i++) { \
    asm volatile(ASM : "+v"(x[0]) : "v"(cc)); asm volatile(ASM : "+v"(x[1]) : "v"(cc)); asm volatile(ASM : "+v"(x[2]) : "v"(cc)); asm volatile(ASM : "+v"(x[3]) : "v"(cc)); \
    asm volatile(ASM : "+v"(x[4]) : "v"(cc)); asm volatile(ASM : "+v"(x[5]) : "v"(cc)); asm volatile(ASM : "+v"(x[6]) : "v"(cc)); asm volatile(ASM : "+v"(x[7]) : "v"(cc)); \
  } \
  double r = 0; for (int j = 0; j < 8; j++) r += x[j]; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(r); \
}
K64(k_add_f64, "v_add_f64 %0, %0, %1")
K64(k_mul_f64, "v_mul_f64 %0, %0, %1")
K64(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
K64(k_rndne_f64, "v_rndne_f64 %0, %0")
K64(k_max_f64, "v_max_f64 %0, %0, %1")
K64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
K64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
K64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
K64(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
K64(k_lshl_add_u64b, "v_lshl_add_u64 %0, %0, 0, %1")
// conversions: 64 <-> 32 (one side a pair)
__global__ void k_cvt_f64_u32(uint32_t* out, uint32_t c) {
  double x[8]; uint32_t a = threadIdx.x + c;
  for (int j = 0; j < 8; j++) x[j] = 0;
#define S(j) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(x[j]) : "v"(a));
  CHAIN8(S)
#undef S
  double r = 0; for (int j = 0; j < 8; j++) r += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(r);
}
__global__ void k_cvt_f64_i32(uint32_t* out, uint32_t c) {
  double x[8]; uint32_t a = threadIdx.x + c;
  for (int j = 0; j < 8; j++) x[j] = 0;
#define S(j) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(x[j]) : "v"(a));
  CHAIN8(S)
#undef S
  double r = 0; for (int j = 0; j < 8; j++) r += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(r);
}
__global__ void k_cvt_u32_f64(uint32_t* out, uint32_t c) {
  uint32_t x[8]; double a = 3.0 + threadIdx.x + c;
  for (int j = 0; j < 8; j++) x[j] = 0;
#define S(j) asm volatile("v_cvt_u32_f64 %0, %1" : "+v"(x[j]) : "v"(a));
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_cvt_i32_f64(uint32_t* out, uint32_t c) {
  uint32_t x[8]; double a = 3.0 + threadIdx.x + c;
  for (int j = 0; j < 8; j++) x[j] = 0;
#define S(j) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(x[j]) : "v"(a));
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_mul_f32(uint32_t* out, uint32_t c) {
  uint32_t x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
#define S(j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c) : "vcc");
  CHAIN8(S)
#undef S
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename K>
void run(const char* name, K kernel, uint32_t* d_out, double clk_ghz, int n_simd) {
  const int blocks = 256 * 8, threads = 512;                  // 8 workgroups of 8 waves per CU -> 16 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, 3u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, 3u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = 5.0 * blocks * (threads / 64) * (double)ITER * 8;
  const double cycles = ms * 1e-3 * clk_ghz * 1e9 * n_simd;
  printf("%-16s %8.3f ms   %.2f SIMD-cycles per wave64 instruction\n", name, ms / 5, cycles / wave_instr);
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const double clk = prop.clockRate * 1e-6;
  const int n_simd = prop.multiProcessorCount * 4;
  printf("%s: %d CUs, %.2f GHz\n", prop.name, prop.multiProcessorCount, clk);
  uint32_t* d_out;
  hipMalloc(&d_out, 256 * 8 * 512 * 4);
  run("v_add_u32 (warm)", k_add, d_out, clk, n_simd);
  run("v_add_u32", k_add, d_out, clk, n_simd);
  run("v_min_u32", k_min, d_out, clk, n_simd);
  run("v_add3_u32", k_add3, d_out, clk, n_simd);
  run("v_alignbit_b32", k_alignbit, d_out, clk, n_simd);
  run("v_mul_u32_u24", k_mul_u24, d_out, clk, n_simd);
  run("v_mul_lo_u32", k_mul_lo, d_out, clk, n_simd);
  run("v_mul_hi_u32", k_mul_hi, d_out, clk, n_simd);
  run("v_mad_u64_u32", k_mad64, d_out, clk, n_simd);
  run("v_lshl_add_u64", k_lshl_add64, d_out, clk, n_simd);
  run("v_sub_u32", k_sub, d_out, clk, n_simd);
  run("v_and_b32", k_and, d_out, clk, n_simd);
  run("v_xor_b32", k_xor, d_out, clk, n_simd);
  run("v_lshlrev_b32", k_lshl, d_out, clk, n_simd);
  run("v_max_u32", k_max, d_out, clk, n_simd);
  run("v_mov_b32", k_mov, d_out, clk, n_simd);
  run("v_add_co_u32", k_addco, d_out, clk, n_simd);
  run("v_sub_co_u32", k_subco, d_out, clk, n_simd);
  run("v_cndmask_b32", k_cndmask, d_out, clk, n_simd);
  run("v_cmp_lt_u32", k_cmp, d_out, clk, n_simd);
  run("v_min_i32", k_mini, d_out, clk, n_simd);
  run("v_bfe_u32", k_bfe, d_out, clk, n_simd);
  run("v_add_f32", k_addf, d_out, clk, n_simd);
  run("v_fma_f32", k_fmaf, d_out, clk, n_simd);
  run("v_min_f32", k_minf, d_out, clk, n_simd);
  run("v_lshl_add_u32", k_lshladd, d_out, clk, n_simd);
  run("v_add_lshl_u32", k_addlshl, d_out, clk, n_simd);
  run("v_mad_u32_u24", k_mad24, d_out, clk, n_simd);
  run("v_sad_u32", k_sad, d_out, clk, n_simd);
  run("v_med3_u32", k_med3, d_out, clk, n_simd);
  run("v_min3_u32", k_min3, d_out, clk, n_simd);
  run("v_pk_add_u16", k_pkadd, d_out, clk, n_simd);
  run("v_pk_min_u16", k_pkmin, d_out, clk, n_simd);
  run("v_subrev_u32", k_subrev, d_out, clk, n_simd);
  run("v_mul_f32", k_mul_f32, d_out, clk, n_simd);
  run("v_add_f64", k_add_f64, d_out, clk, n_simd);
  run("v_mul_f64", k_mul_f64, d_out, clk, n_simd);
  run("v_fma_f64", k_fma_f64, d_out, clk, n_simd);
  run("v_rndne_f64", k_rndne_f64, d_out, clk, n_simd);
  run("v_max_f64", k_max_f64, d_out, clk, n_simd);
  run("v_cmp_lt_f64", k_cmp_f64, d_out, clk, n_simd);
  run("v_pk_fma_f32", k_pk_fma_f32, d_out, clk, n_simd);
  run("v_pk_add_f32", k_pk_add_f32, d_out, clk, n_simd);
  run("v_pk_mul_f32", k_pk_mul_f32, d_out, clk, n_simd);
  run("v_lshl_add_u64 (acc)", k_lshl_add_u64b, d_out, clk, n_simd);
  run("v_cvt_f64_u32", k_cvt_f64_u32, d_out, clk, n_simd);
  run("v_cvt_f64_i32", k_cvt_f64_i32, d_out, clk, n_simd);
  run("v_cvt_u32_f64", k_cvt_u32_f64, d_out, clk, n_simd);
  run("v_cvt_i32_f64", k_cvt_i32_f64, d_out, clk, n_simd);
  hipFree(d_out);
  return 0;
}
