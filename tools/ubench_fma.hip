// ubench_fma.hip -- ONE Montgomery product of the Twisted-Edwards-BLS12 base field in double-precision FMAs (5 limbs of 52
// bits, R = 2^260) against the engine's 153-mad product (9 limbs of 29 bits, csrc/fp.hpp), on gfx950.  Round-2 verdict, item 6:
// v_fma_f64 issues about as fast as v_mad_u64_u32 and carries a wider multiplier -- does a field product get cheaper?
//
// The FMA form is the standard one (Emmart et al.): with round-toward-zero, for integers a, b < 2^52 held in doubles
//     hi = fma(a, b, 2^104)              = 2^104 + floor(ab / 2^52) * 2^52     (its mantissa bits ARE floor(ab / 2^52))
//     lo = fma(a, b, (2^104 + 2^52) - hi) = 2^52 + (ab mod 2^52)               (its mantissa bits ARE ab mod 2^52)
// and the bit patterns are summed as 64-bit integers, column by column (the exponent fields add up to a constant per column).
// Per 52x52 partial product: 2 FMAs + 1 f64 subtraction + 2 64-bit integer additions.  The Montgomery quotient digit needs a
// real multiplication here (p = 1 mod 2^47 only: with 52-bit limbs -p^-1 is not -1), i.e. one more lo-product per column.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_fma tools/ubench_fma.hip ; run: ./ubench_fma
// Output: ns per product and wave at 4 waves per SIMD for both forms, instruction counts come from the ISA (hipcc -S), and
// limbs of a few products for tools/check_ubench_fma.py (bigint check of both forms).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../webgpu-msm-twisted-edwards_amd/csrc/fp.hpp"

#define ITERS 512
#define CHAINS 4

// p in 52-bit limbs, and -p^-1 mod 2^52 (filled by the host from the 32-bit words of fp.hpp)
struct fma_consts { double p[5]; double ninv; };

__device__ __forceinline__ uint64_t bits(double x) { return (uint64_t)__double_as_longlong(x); }
__device__ __forceinline__ double dbl(uint64_t x) { return __longlong_as_double((long long)x); }

// a, b: 5 doubles holding integers < 2^52 (value < 2^256); returns a*b/2^260 mod p (+ a multiple of p), limbs < 2^52
__device__ __forceinline__ void mont_mul_fma(const double (&a)[5], const double (&b)[5], const fma_consts& k, double (&r)[5]) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
  const uint64_t EXP_HI = 0x467ull << 52, EXP_LO = 0x433ull << 52, M52 = (1ull << 52) - 1ull;
  uint64_t col[11];
#pragma unroll
  for (int i = 0; i < 11; i++) col[i] = 0;
  // a * b: lo parts into column i + j, hi parts into column i + j + 1 (exponent fields included, removed below)
#pragma unroll
  for (int i = 0; i < 5; i++) {
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const double hi = __builtin_fma(a[i], b[j], C1);
      const double lo = __builtin_fma(a[i], b[j], C2 - hi);
      col[i + j] += bits(lo); col[i + j + 1] += bits(hi);
    }
  }
  // exponent fields contributed so far: column c got (number of lo terms) * EXP_LO + (number of hi terms) * EXP_HI
#pragma unroll
  for (int c = 0; c < 10; c++) {
    const int nlo = c < 5 ? c + 1 : 9 - c, nhi = c == 0 ? 0 : (c - 1 < 5 ? c : 10 - c);
    col[c] -= (uint64_t)(nlo > 0 ? nlo : 0) * EXP_LO + (uint64_t)(nhi > 0 ? nhi : 0) * EXP_HI;
  }
  // Montgomery reduction, radix 2^52: q_i = (col_i * ninv) mod 2^52, col += q_i * p * 2^(52 i), carry col_i >> 52 upwards
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const double x = dbl((col[i] & M52) | EXP_LO) - 0x1p52;                 // low 52 bits of the column as a double
    const double qh = __builtin_fma(x, k.ninv, C1);
    const double ql = __builtin_fma(x, k.ninv, C2 - qh);                     // 2^52 + (x * ninv mod 2^52)
    const double q = ql - 0x1p52;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const double hi = __builtin_fma(q, k.p[j], C1);
      const double lo = __builtin_fma(q, k.p[j], C2 - hi);
      col[i + j] += bits(lo) - EXP_LO; col[i + j + 1] += bits(hi) - EXP_HI;
    }
    col[i + 1] += col[i] >> 52;                                               // the low 52 bits of col_i are zero now
  }
  // columns 5..9 hold the result; normalise to 52-bit limbs in doubles
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const uint64_t v = col[5 + i] + c;
    r[i] = dbl((v & M52) | EXP_LO) - 0x1p52;
    c = v >> 52;
  }
}

// NC of the thread's CHAINS values are worked on (independent products per iteration): 4 as in the engine's closing products
// (spills at the 128 registers that four waves per SIMD allow), or 2 (no spills)
template <int NC>
__global__ void __launch_bounds__(256, 4) k_fma(double* io, fma_consts k, int sample) {
  // MODE.FP_ROUND[3:2] (f64) = toward zero.  As inline asm: after the s_setreg builtin the compiler's mode-register pass puts
  // the default (nearest-even) back in front of the first FMA.
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
  double x[NC][5], y[5];
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 5 * (CHAINS + 1);
  for (int c = 0; c < NC; c++) for (int i = 0; i < 5; i++) x[c][i] = io[base + 5 * c + i];
  for (int i = 0; i < 5; i++) y[i] = io[base + 5 * CHAINS + i];
  const int iters = sample ? 1 : ITERS * (CHAINS / NC);                      // the same number of products per thread
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < NC; c++) { double r[5]; mont_mul_fma(x[c], y, k, r); for (int i = 0; i < 5; i++) x[c][i] = r[i]; }
  }
  for (int c = 0; c < NC; c++) for (int i = 0; i < 5; i++) io[base + 5 * c + i] = x[c][i];
}

__global__ void __launch_bounds__(256, 4) k_mad(uint32_t* io, int sample) {
  te::fp x[CHAINS], y;
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 9 * (CHAINS + 1);
  for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 9; i++) x[c].v[i] = io[base + 9 * c + i];
  for (int i = 0; i < 9; i++) y.v[i] = io[base + 9 * CHAINS + i];
  const int iters = sample ? 1 : ITERS;
  for (int it = 0; it < iters; it++) {
    te::fp a[CHAINS], b[CHAINS], r[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { a[c] = x[c]; b[c] = y; }
    te::mont_mul_x<CHAINS>(a, b, r);                                         // the engine's lockstep product, 4 chains as in ete_close
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = r[c];
  }
  for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 9; i++) io[base + 9 * c + i] = x[c].v[i];
}

static uint64_t splitmix(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

int main() {
  const int blocks = 256 * 4, threads = blocks * 256;      // four blocks of four waves per CU: 4 waves per SIMD
  // p as a 256-bit integer -> 52-bit limbs; ninv = -p^-1 mod 2^52 by Newton iteration
  unsigned __int128 lo = ((unsigned __int128)te::P_W32[3] << 96) | ((unsigned __int128)te::P_W32[2] << 64) | ((unsigned __int128)te::P_W32[1] << 32) | te::P_W32[0];
  unsigned __int128 hi = ((unsigned __int128)te::P_W32[7] << 96) | ((unsigned __int128)te::P_W32[6] << 64) | ((unsigned __int128)te::P_W32[5] << 32) | te::P_W32[4];
  auto limb52 = [&](int i) -> uint64_t {
    const int bit = 52 * i; uint64_t v;
    if (bit < 128) { v = (uint64_t)(lo >> bit); if (bit + 52 > 128) v |= (uint64_t)(hi << (128 - bit)); } else v = (uint64_t)(hi >> (bit - 128));
    return v & ((1ull << 52) - 1);
  };
  fma_consts k;
  for (int i = 0; i < 5; i++) k.p[i] = (double)limb52(i);
  uint64_t p0 = (uint64_t)lo, inv = 1;
  for (int i = 0; i < 6; i++) inv *= 2 - p0 * inv;          // p0^-1 mod 2^64
  k.ninv = (double)((0 - inv) & ((1ull << 52) - 1));
  // random operands below 2^252 in both limb forms (same values)
  std::vector<double> hf((size_t)threads * 5 * (CHAINS + 1)); std::vector<uint32_t> hm((size_t)threads * 9 * (CHAINS + 1));
  uint64_t seed = 12345;
  for (size_t e = 0; e < (size_t)threads * (CHAINS + 1); e++) {
    uint64_t w[4] = {splitmix(seed), splitmix(seed), splitmix(seed), splitmix(seed) >> 12};
    auto bitsat = [&](int bit, int n) -> uint64_t { uint64_t v = w[bit >> 6] >> (bit & 63); if ((bit & 63) + n > 64 && (bit >> 6) + 1 < 4) v |= w[(bit >> 6) + 1] << (64 - (bit & 63)); return v & ((1ull << n) - 1); };
    for (int i = 0; i < 5; i++) hf[e * 5 + i] = (double)(52 * i < 256 ? bitsat(52 * i, 52 * i + 52 > 256 ? 256 - 52 * i : 52) : 0);
    for (int i = 0; i < 9; i++) hm[e * 9 + i] = (uint32_t)(29 * i < 256 ? bitsat(29 * i, 29 * i + 29 > 256 ? 256 - 29 * i : 29) : 0);
  }
  double* df; uint32_t* dm;
  (void)hipMalloc(&df, hf.size() * 8); (void)hipMalloc(&dm, hm.size() * 4);
  // --- samples for the bigint check: one product per chain
  (void)hipMemcpy(df, hf.data(), hf.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dm, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_fma<4>, dim3(blocks), dim3(256), 0, 0, df, k, 1);
  hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, dm, 1);
  std::vector<double> of(hf.size()); std::vector<uint32_t> om(hm.size());
  (void)hipMemcpy(of.data(), df, hf.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(om.data(), dm, hm.size() * 4, hipMemcpyDeviceToHost);
  for (int s = 0; s < 6; s++) {                              // thread s * 1000, chain s % CHAINS
    const size_t t = (size_t)s * 1000, c = s % CHAINS;
    printf("SAMPLE fma a"); for (int i = 0; i < 5; i++) printf(" %.0f", hf[(t * (CHAINS + 1) + c) * 5 + i]);
    printf(" b"); for (int i = 0; i < 5; i++) printf(" %.0f", hf[(t * (CHAINS + 1) + CHAINS) * 5 + i]);
    printf(" r"); for (int i = 0; i < 5; i++) printf(" %.0f", of[(t * (CHAINS + 1) + c) * 5 + i]);
    printf("\nSAMPLE mad a"); for (int i = 0; i < 9; i++) printf(" %u", hm[(t * (CHAINS + 1) + c) * 9 + i]);
    printf(" b"); for (int i = 0; i < 9; i++) printf(" %u", hm[(t * (CHAINS + 1) + CHAINS) * 9 + i]);
    printf(" r"); for (int i = 0; i < 9; i++) printf(" %u", om[(t * (CHAINS + 1) + c) * 9 + i]);
    printf("\n");
  }
  // --- timing
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeit = [&](auto launch) { float best = 1e9f; for (int rep = 0; rep < 5; rep++) { (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; };
  const float tf4 = timeit([&] { hipLaunchKernelGGL(k_fma<4>, dim3(blocks), dim3(256), 0, 0, df, k, 0); });
  const float tf2 = timeit([&] { hipLaunchKernelGGL(k_fma<2>, dim3(blocks), dim3(256), 0, 0, df, k, 0); });
  const float tf = tf2 < tf4 ? tf2 : tf4;
  const float tm = timeit([&] { hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, dm, 0); });
  const double products_per_wave = (double)ITERS * CHAINS, waves_per_simd = 4.0;
  // a SIMD executes waves_per_simd waves; time per wave-product on a SIMD = kernel time / (products per wave * waves per SIMD)
  printf("RESULT fma, 4 products in lockstep (spills): %.3f ms   2 products (no spills): %.3f ms\n", tf4, tf2);
  printf("RESULT fma: %.3f ms  -> %.2f ns per product and wave (SIMD issue time)\n", tf, tf * 1e6 / (products_per_wave * waves_per_simd));
  printf("RESULT mad: %.3f ms  -> %.2f ns per product and wave (SIMD issue time)\n", tm, tm * 1e6 / (products_per_wave * waves_per_simd));
  printf("RESULT ratio fma / mad = %.3f\n", tf / tm);
  return 0;
}
