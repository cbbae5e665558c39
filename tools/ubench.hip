// ubench.hip -- instruction-issue micro-benchmarks for gfx950 ("the integer-MAC rate ...
// peak to be micro-benchmarked on the box, not assumed").  Each kernel runs a long unrolled loop of ONE
// instruction form over 8 independent register chains; reports wave-instructions per ns per SIMD and
// cycles per wave-instruction at the measured clock, for 1 / 2 / 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip ; run: ./ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define ITERS 2048
#define REP 8     // instructions per chain per iteration -> 64 instrs / iteration

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define BODY(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS)

#define KERNEL32(NAME, ASMSTR)                                                                       \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                        \
    uint32_t r[8], a = seed * 2654435761u + threadIdx.x, b = seed ^ 0x9e3779b9u;                     \
    for (int i = 0; i < 8; i++) r[i] = a + i * 7919u;                                                \
    for (int it = 0; it < ITERS; it++) {                                                             \
      _Pragma("unroll") for (int k = 0; k < REP; k++) {                                              \
        _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(b)); \
      }                                                                                              \
    }                                                                                                \
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= r[i];                                           \
    if (s == 0x12345678u) out[0] = s;                                                                \
  }

#define KERNEL64(NAME, ASMSTR)                                                                       \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                        \
    uint64_t r[8]; uint32_t a = seed * 2654435761u + threadIdx.x, b = seed ^ 0x9e3779b9u;            \
    for (int i = 0; i < 8; i++) r[i] = ((uint64_t)(a + i) << 32) | (b + i);                          \
    for (int it = 0; it < ITERS; it++) {                                                             \
      _Pragma("unroll") for (int k = 0; k < REP; k++) {                                              \
        _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(b) : "vcc"); \
      }                                                                                              \
    }                                                                                                \
    uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= r[i];                                           \
    if (s == 0x12345678u) out[0] = (uint32_t)s;                                                      \
  }

#define KERNELF64(NAME, ASMSTR)                                                                      \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                        \
    double r[8]; double a = 1.0 + seed * 1e-9 + threadIdx.x * 1e-12, b = 1e-30;                      \
    for (int i = 0; i < 8; i++) r[i] = 1.0 + i * 1e-6;                                               \
    for (int it = 0; it < ITERS; it++) {                                                             \
      _Pragma("unroll") for (int k = 0; k < REP; k++) {                                              \
        _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(b)); \
      }                                                                                              \
    }                                                                                                \
    double s = 0; for (int i = 0; i < 8; i++) s += r[i];                                             \
    if (s == 0.12345) out[0] = 1;                                                                    \
  }

KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL32(k_cmp_cndmask, "v_cmp_gt_u32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_sub_co_chain, "v_sub_co_u32 %0, vcc, %0, %1\n\tv_subb_co_u32 %0, vcc, %0, %2, vcc")
KERNEL32(k_addc_pair, "v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 13")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %0 bitop3:0xc")
KERNEL32(k_sad_u32, "v_sad_u32 %0, %0, %1, %2")
KERNEL32(k_not_b32, "v_not_b32 %0, %0")
KERNEL32(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
KERNEL32(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL32(k_sub_u32, "v_sub_u32 %0, %1, %0")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_mad_u64_u32_addc, "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL64(k_lshrrev_b64, "v_lshrrev_b64 %0, 3, %0")
KERNELF64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNELF64(k_add_f64, "v_add_f64 %0, %0, %2")
KERNELF64(k_mul_f64, "v_mul_f64 %0, %0, %1")

typedef void (*kern_t)(uint32_t*, uint32_t);
struct entry { const char* name; kern_t k; int instrs_per_slot; };

int main() {
  std::vector<entry> es = {
    {"v_add_u32", k_add_u32, 1}, {"v_mov_b32", k_mov_b32, 1}, {"v_add3_u32", k_add3_u32, 1}, {"v_fma_f32", k_fma_f32, 1},
    {"v_cndmask_b32", k_cndmask, 1}, {"v_cndmask_b32_e64 (sgpr)", k_cndmask_e64, 1}, {"v_cmp+v_cndmask (pair)", k_cmp_cndmask, 2}, {"v_sub_co+v_subb_co (pair)", k_sub_co_chain, 2}, {"v_alignbit_b32", k_alignbit, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_sad_u32", k_sad_u32, 1}, {"v_not_b32", k_not_b32, 1}, {"v_lshrrev_b32", k_lshrrev_b32, 1}, {"v_and_b32", k_and_b32, 1}, {"v_sub_u32", k_sub_u32, 1},
    {"v_add_co+v_addc_co (pair)", k_addc_pair, 2},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1}, {"v_mad_u64_u32", k_mad_u64_u32, 1},
    {"v_mad_u64_u32+v_addc_co (pair)", k_mad_u64_u32_addc, 2},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1},
    {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1},
  };
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  uint32_t* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-34s %28s %28s %28s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
  printf("%-34s %28s %28s %28s\n", "", "cyc/instr@2.4GHz (ns/winstr)", "", "");
  for (auto& e : es) {
    printf("%-34s", e.name);
    for (int wps : {1, 2, 4}) {
      const int blocks = cus * wps;                      // 256 threads = 4 waves = 1 wave per SIMD of a CU
      for (int w = 0; w < 2; w++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int reps = 5;
      for (int w = 0; w < reps; w++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double slots = (double)ITERS * REP * 8 * wps * reps;      // wave-instruction slots per SIMD
      const double ns_per = ms * 1e6 / slots / e.instrs_per_slot;
      printf("   %10.2f cyc (%7.3f ns)", ns_per * 2.4, ns_per);
    }
    printf("\n");
  }
  return 0;
}
