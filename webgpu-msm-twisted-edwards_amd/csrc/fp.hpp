// fp.hpp -- base field of the Twisted-Edwards-BLS12 curve for CDNA4 (gfx950).
//
// Replaces the reference's 20 x 13-bit-limb field (wgsl/bigint/bigint.template.wgsl:1-45,
// wgsl/field/field.template.wgsl:1-35, wgsl/montgomery/mont_pro_product.template.wgsl:15-57,
// wgsl/cuzk/barrett.template.wgsl:16-78).  That limb width exists only because WGSL has no 64-bit
// integers; here a field element is 8 x 32-bit limbs in Montgomery form with R = 2^256 and the
// product is an operand-scanning CIOS built on v_mad_u64_u32 (32x32+64 -> 64).
//
// LAZY REDUCTION.  p < 2^253 and R = 2^256, so p/R < 0.0730 and any value below 13.7p fits in a limb
// vector.  Nothing in the hot loop is reduced to [0, p): every function states the bound it needs
// and the bound it returns, in multiples of p.  With operands a < ka*p, b < kb*p,
//        mont_mul(a, b) < (ka*kb*0.0730 + 1) * p           (no final subtraction, ever)
// and additions / subtractions are plain 256-bit carries (subtraction adds a multiple of p first).
//
// The same header compiles for the host (plain g++) so tests/ can check the device arithmetic
// bit-for-bit on the CPU build box; it is not a CPU fallback of the product.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TE_HD __host__ __device__ __forceinline__
#else
#define TE_HD inline
#endif

namespace te {

struct fp { uint32_t v[8]; };   // little-endian limbs

// p = 0x12ab655e 9a2ca556 60b44d1e 5c37b001 59aa76fe d0000001 0a118000 00000001  (params.ts:11-13)
#define TE_P0 0x00000001u
#define TE_P1 0x0a118000u
#define TE_P2 0xd0000001u
#define TE_P3 0x59aa76feu
#define TE_P4 0x5c37b001u
#define TE_P5 0x60b44d1eu
#define TE_P6 0x9a2ca556u
#define TE_P7 0x12ab655eu

TE_HD uint32_t p_limb(int i) {
  switch (i) {
    case 0: return TE_P0; case 1: return TE_P1; case 2: return TE_P2; case 3: return TE_P3;
    case 4: return TE_P4; case 5: return TE_P5; case 6: return TE_P6; default: return TE_P7;
  }
}
// limb i of K*p (K*p < 2^256 for K <= 13), a compile-time constant
constexpr uint32_t kp_limb_c(int K, int i) {
  constexpr uint32_t P[8] = {TE_P0, TE_P1, TE_P2, TE_P3, TE_P4, TE_P5, TE_P6, TE_P7};
  uint64_t acc = 0;
  for (int j = 0; j <= i; j++) acc = (acc >> 32) + (uint64_t)P[j] * (uint32_t)K;
  return (uint32_t)acc;
}
template <int K> TE_HD uint32_t kp_limb(int i) {
  switch (i) {
    case 0: return kp_limb_c(K, 0); case 1: return kp_limb_c(K, 1); case 2: return kp_limb_c(K, 2); case 3: return kp_limb_c(K, 3);
    case 4: return kp_limb_c(K, 4); case 5: return kp_limb_c(K, 5); case 6: return kp_limb_c(K, 6); default: return kp_limb_c(K, 7);
  }
}

// Constants in Montgomery form (R = 2^256); values are re-derived from bigint arithmetic in
// tests/test_host_logic.py::test_field_constants.
//   R mod p, R^2 mod p, d*R mod p (d = 3021, AleoConstants.ts:2-4)
TE_HD fp fp_const(const uint32_t (&w)[8]) { fp r; for (int i = 0; i < 8; i++) r.v[i] = w[i]; return r; }

TE_HD fp fp_zero() { fp r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }

// ---------------------------------------------------------------------------------------------
// Montgomery product, CIOS over 32-bit limbs.
//   requires a < 8p (any b < 2^256); returns a*b/R + (something < p), i.e. < (ka*kb*0.073 + 1) p.
// p = 1 (mod 2^32), so -p^-1 mod 2^32 = 0xffffffff: the per-row quotient digit is m = -t[0], and
// t[0] + m*p[0] is either 0 or 2^32 -- no multiplication for the lowest limb.
// Row bound: t < a + p + eps < 9p < 2^256 after every row, and t + a*b_i + m*p < 2^288 inside a
// row, so one transient top word suffices and never overflows.
TE_HD fp mont_mul_ref(const fp& a, const fp& b) {
  uint32_t t[8];
#pragma unroll
  for (int j = 0; j < 8; j++) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t bi = b.v[i];
    uint64_t s;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      s = (uint64_t)a.v[j] * bi + t[j] + c;
      t[j] = (uint32_t)s;
      c = (uint32_t)(s >> 32);
    }
    const uint32_t top = c;
    const uint32_t m = 0u - t[0];
    c = (t[0] != 0u) ? 1u : 0u;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      s = (uint64_t)m * p_limb(j) + t[j] + c;
      t[j - 1] = (uint32_t)s;
      c = (uint32_t)(s >> 32);
    }
    t[7] = top + c;
  }
  fp r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.v[j] = t[j];
  return r;
}

// a + b, no reduction.  requires a + b < 2^256.
TE_HD fp fp_add_ref(const fp& a, const fp& b) {
  fp r; uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  return r;
}
// a - b + K*p, no reduction.  requires b <= K*p and a + K*p < 2^256.
template <int K> TE_HD fp fp_sub_ref(const fp& a, const fp& b) {
  fp r; int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (int64_t)a.v[i] + (int64_t)kp_limb<K>(i) - (int64_t)b.v[i];
    r.v[i] = (uint32_t)c; c >>= 32;
  }
  return r;
}
// K*p - a.  requires a <= K*p.
template <int K> TE_HD fp fp_neg_ref(const fp& a) {
  fp r; int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (int64_t)kp_limb<K>(i) - (int64_t)a.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  return r;
}
// if a >= K*p then a - K*p else a.
template <int K> TE_HD fp fp_csub_ref(const fp& a) {
  fp d; int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (int64_t)a.v[i] - (int64_t)kp_limb<K>(i); d.v[i] = (uint32_t)c; c >>= 32; }
  const bool borrow = c < 0;
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = borrow ? a.v[i] : d.v[i];
  return r;
}

// ---------------------------------------------------------------------------------------------
// gfx950 forms.  Measured on MI355X (profiles/r01_ubench_instruction_rates.txt): v_mad_u64_u32 issues in
// ~4.6 cycles per wave, the same as any VOP3 / carry instruction (v_addc_co_u32 4.2, v_lshl_add_u64
// 4.2) and twice a plain VOP2 (2.5) -- 32-bit integer multiplies are NOT quarter rate on CDNA4, so the
// cost of a field product is its instruction COUNT.  The portable CIOS above compiles to 120 mads +
// 118 64-bit adds + ~370 moves (zero-extensions the compiler needs to feed 64-bit addends); the
// product-scanning form below is 120 x (v_mad_u64_u32 + v_addc_co_u32) with a three-word column
// accumulator and no zero-extension at all.  Same value, same bounds as mont_mul_ref.
#if defined(__HIP_DEVICE_COMPILE__)
#define TE_ASM_FIELD 1
#include "fp_montmul_gfx950.inc"

#define TE_V8(x) "v"(x.v[0]), "v"(x.v[1]), "v"(x.v[2]), "v"(x.v[3]), "v"(x.v[4]), "v"(x.v[5]), "v"(x.v[6]), "v"(x.v[7])
#define TE_O8(x) "=&v"(x.v[0]), "=&v"(x.v[1]), "=&v"(x.v[2]), "=&v"(x.v[3]), "=&v"(x.v[4]), "=&v"(x.v[5]), "=&v"(x.v[6]), "=&v"(x.v[7])
#define TE_K8(K) "v"(kp_limb_c(K, 0)), "v"(kp_limb_c(K, 1)), "v"(kp_limb_c(K, 2)), "v"(kp_limb_c(K, 3)), \
                 "v"(kp_limb_c(K, 4)), "v"(kp_limb_c(K, 5)), "v"(kp_limb_c(K, 6)), "v"(kp_limb_c(K, 7))

__device__ __forceinline__ fp fp_add(const fp& a, const fp& b) {
  fp r;
  asm("v_add_co_u32 %0, vcc, %8, %16\n\tv_addc_co_u32 %1, vcc, %9, %17, vcc\n\tv_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\tv_addc_co_u32 %4, vcc, %12, %20, vcc\n\tv_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\tv_addc_co_u32 %7, vcc, %15, %23, vcc"
      : TE_O8(r) : TE_V8(a), TE_V8(b) : "vcc");
  return r;
}
// a - b + K*p: borrow chain, then carry chain.  The limbs of K*p sit in VGPRs: a carry-in through VCC already
// uses the one constant-bus read gfx9 allows per instruction, so neither an SGPR nor a literal fits beside it.
template <int K> __device__ __forceinline__ fp fp_sub(const fp& a, const fp& b) {
  fp d, r;
  asm("v_sub_co_u32 %0, vcc, %8, %16\n\tv_subb_co_u32 %1, vcc, %9, %17, vcc\n\tv_subb_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %11, %19, vcc\n\tv_subb_co_u32 %4, vcc, %12, %20, vcc\n\tv_subb_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %14, %22, vcc\n\tv_subb_co_u32 %7, vcc, %15, %23, vcc"
      : TE_O8(d) : TE_V8(a), TE_V8(b) : "vcc");
  asm("v_add_co_u32 %0, vcc, %16, %8\n\tv_addc_co_u32 %1, vcc, %17, %9, vcc\n\tv_addc_co_u32 %2, vcc, %18, %10, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %19, %11, vcc\n\tv_addc_co_u32 %4, vcc, %20, %12, vcc\n\tv_addc_co_u32 %5, vcc, %21, %13, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %22, %14, vcc\n\tv_addc_co_u32 %7, vcc, %23, %15, vcc"
      : TE_O8(r) : TE_V8(d), TE_K8(K) : "vcc");
  return r;
}
// K*p - a
template <int K> __device__ __forceinline__ fp fp_neg(const fp& a) {
  fp r;
  asm("v_sub_co_u32 %0, vcc, %16, %8\n\tv_subb_co_u32 %1, vcc, %17, %9, vcc\n\tv_subb_co_u32 %2, vcc, %18, %10, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %19, %11, vcc\n\tv_subb_co_u32 %4, vcc, %20, %12, vcc\n\tv_subb_co_u32 %5, vcc, %21, %13, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %22, %14, vcc\n\tv_subb_co_u32 %7, vcc, %23, %15, vcc"
      : TE_O8(r) : TE_V8(a), TE_K8(K) : "vcc");
  return r;
}
// a >= K*p ? a - K*p : a   (d = a - K*p by v_subrev, select on the final borrow left in vcc)
template <int K> __device__ __forceinline__ fp fp_csub(const fp& a) {
  fp d, r;
  asm("v_subrev_co_u32 %0, vcc, %24, %16\n\tv_subbrev_co_u32 %1, vcc, %25, %17, vcc\n\tv_subbrev_co_u32 %2, vcc, %26, %18, vcc\n\t"
      "v_subbrev_co_u32 %3, vcc, %27, %19, vcc\n\tv_subbrev_co_u32 %4, vcc, %28, %20, vcc\n\tv_subbrev_co_u32 %5, vcc, %29, %21, vcc\n\t"
      "v_subbrev_co_u32 %6, vcc, %30, %22, vcc\n\tv_subbrev_co_u32 %7, vcc, %31, %23, vcc\n\t"
      "v_cndmask_b32 %8, %0, %16, vcc\n\tv_cndmask_b32 %9, %1, %17, vcc\n\tv_cndmask_b32 %10, %2, %18, vcc\n\tv_cndmask_b32 %11, %3, %19, vcc\n\t"
      "v_cndmask_b32 %12, %4, %20, vcc\n\tv_cndmask_b32 %13, %5, %21, vcc\n\tv_cndmask_b32 %14, %6, %22, vcc\n\tv_cndmask_b32 %15, %7, %23, vcc"
      : TE_O8(d), TE_O8(r) : TE_V8(a), TE_K8(K) : "vcc");
  return r;
}
#else
TE_HD fp mont_mul(const fp& a, const fp& b) { return mont_mul_ref(a, b); }
TE_HD fp fp_add(const fp& a, const fp& b) { return fp_add_ref(a, b); }
template <int K> TE_HD fp fp_sub(const fp& a, const fp& b) { return fp_sub_ref<K>(a, b); }
template <int K> TE_HD fp fp_neg(const fp& a) { return fp_neg_ref<K>(a); }
template <int K> TE_HD fp fp_csub(const fp& a) { return fp_csub_ref<K>(a); }
#endif

// canonical representative in [0, p) of any a < 2^256 (< 14p): conditional subtractions of 8p, 4p, 2p, p
TE_HD fp fp_reduce_full(const fp& a) { return fp_csub<1>(fp_csub<2>(fp_csub<4>(fp_csub<8>(a)))); }

// a/2 mod p for canonical a (< p): (a + (a odd ? p : 0)) >> 1
TE_HD fp fp_half(const fp& a) {
  const uint32_t odd = a.v[0] & 1u;
  uint32_t t[8]; uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + (odd ? p_limb(i) : 0u); t[i] = (uint32_t)c; c >>= 32; }
  fp r;
#pragma unroll
  for (int i = 0; i < 7; i++) r.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
  r.v[7] = t[7] >> 1;      // a + p < 2p < 2^254: no carry out of limb 7
  return r;
}

TE_HD bool fp_is_zero_canonical(const fp& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}

// ---------------------------------------------------------------------------------------------
// Constants (Montgomery form, R = 2^256), generated by tools/gen_constants.py and re-derived in
// tests/test_host_logic.py::test_field_constants.
#include "fp_constants.inc"

}  // namespace te
