// fp.hpp -- base field of the Twisted-Edwards-BLS12 curve for CDNA4 (gfx950).
//
// Replaces the reference's 20 x 13-bit-limb field (wgsl/bigint/bigint.template.wgsl:1-45,
// wgsl/field/field.template.wgsl:1-35, wgsl/montgomery/mont_pro_product.template.wgsl:15-57,
// wgsl/cuzk/barrett.template.wgsl:16-78).  That limb width exists only because WGSL has no 64-bit
// integers.  Here a field element is 9 limbs of 29 bits (one per u32) in Montgomery form, R = 2^261.
//
// WHY 9 x 29 AND NOT 8 x 32.  Measured on MI355X (profiles/r01_ubench_instruction_rates.txt): v_mad_u64_u32
// (32x32+64 -> 64) issues in ~4.6 cycles per wave -- the same as v_addc_co_u32 (4.2) or any other VOP3 /
// carry instruction; 32-bit integer multiplies are NOT quarter rate on CDNA4.  A field product therefore costs
// its instruction COUNT.  With saturated 32-bit limbs every multiply-accumulate needs a carry instruction
// (120 x (mad + addc) + column shuffles = ~345 instructions; that version is in the git history).  With 29-bit
// limbs a 64-bit accumulator absorbs a whole column of up to 17 partial products with no carry at all:
// 153 mads + ~50 column instructions = ~205, plain C++, no inline asm, and additions / subtractions become
// limb-wise VOP2 adds with no carry chain.
//
// LIMB BOUNDS (magnitude classes; every function states what it takes and returns):
//   N   "normalised": limbs 0..7 < 2^29 (limb 8, the top, is small: value / 2^232)
//   S   N + N                        : limbs < 2^30
//   D   N - N + offset(K*p)          : limbs < 2^30.6 (the offset form of K*p has limbs in [2^29, 2^30))
//   mont_mul(a, b) is exact while  9 * max(a_i) * max(b_j) + 8 * 2^58 + 2^36 < 2^64:
//   N x anything <= 2^31, S x S, S x D are fine;  D x D is NOT -- normalise one operand (fp_norm, 25 ops).
// VALUE BOUNDS never bind: R = 2^261 is 446 p, so mont_mul(a, b) < a*b/R + p stays below 2p for any operands
// below ~20p, and nothing in the hot loop is ever reduced modulo p.
//
// The same header compiles for the host (plain g++) so tests/ check the exact device arithmetic bit-for-bit on
// the CPU build box; it is not a CPU fallback of the product.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TE_HD __host__ __device__ __forceinline__
#else
#define TE_HD inline
#endif

namespace te {

constexpr int NL = 9;                     // limbs
constexpr uint32_t LB = 29;               // bits per limb
constexpr uint32_t LM = (1u << LB) - 1u;  // limb mask

// A field element in limbs: N little-endian 29-bit limbs, one per u32 word.  Both base fields of the engine use it --
// fp (this file, N = 9) and te377::fq (fq377.hpp, N = 14); the curve and the kernels are written once for any N
// (field.hpp gives the two fields one vocabulary).
template <int N> struct fel { uint32_t v[N]; };
using fp = fel<9>;               // 36 bytes

// p = 0x12ab655e 9a2ca556 60b44d1e 5c37b001 59aa76fe d0000001 0a118000 00000001  (params.ts:11-13), 32-bit words
constexpr uint32_t P_W32[8] = {0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu, 0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu};
// limb i of p in radix 2^29 (compile-time)
constexpr uint32_t p29(int i) {
  const int bit = i * 29, w = bit >> 5, s = bit & 31;
  uint64_t two = P_W32[w];
  if (w + 1 < 8) two |= (uint64_t)P_W32[w + 1] << 32;
  return (uint32_t)(two >> s) & ((1u << 29) - 1u);
}
TE_HD uint32_t p_limb(int i) {
  switch (i) {
    case 0: return p29(0); case 1: return p29(1); case 2: return p29(2); case 3: return p29(3); case 4: return p29(4);
    case 5: return p29(5); case 6: return p29(6); case 7: return p29(7); default: return p29(8);
  }
}

TE_HD fp fp_zero() { fp r; for (int i = 0; i < NL; i++) r.v[i] = 0; return r; }

// K*p in "offset form": limbs 0..7 raised by 2^29, the next limb lowered by 1 (same value), so that
// a_i + off_i - b_i cannot underflow for normalised b.  Specialisations in fp_constants.inc (K = 2, 4, 8, 16).
template <int K> TE_HD fp fp_kp_offset();

// ---------------------------------------------------------------------------------------------
// Montgomery product a*b/R mod p (plus a multiple of p), product scanning, R = 2^261.
// p = 1 (mod 2^29), so -p^-1 mod 2^29 = 2^29 - 1: the quotient digit of a column with value V is q = -V mod 2^29, and
// adding q*p[0] = q only clears the low 29 bits and carries into the next column.  Takes operands whose limb magnitudes
// satisfy the rule above; returns class N with value <= a*b/R + p.
//
// CARRY-FOLDED QUOTIENT.  That q*p[0] term is never computed.  Column 0 takes q_0 = 2^29 - (V_0 mod 2^29), in [1, 2^29]
// (2^29 instead of 0 when V_0 = 0 mod 2^29: as good a multiple), so V_0 + q_0 carries exactly (V_0 >> 29) + 1.  From
// column 1 on the accumulator holds W_i = V_i - 1, i.e. it leaves that "+ 1" out:  q_i = ~W_i mod 2^29 is then the
// standard digit -V_i mod 2^29, and the carry of V_i + q_i is again exactly (W_i >> 29) + 1 (if the low bits of W_i
// are all ones, q_i = 0 and V_i itself carries; otherwise V_i + q_i rounds W_i up to the next multiple of 2^29).  So
// the next accumulator is W_{i+1} = (products of column i+1) + (W_i >> 29) with no correction, nine columns long, and
// the one outstanding "+ 1" is added when the first result column (9) is complete.  Per product: 9 mads and 9
// subtractions fewer (~W & mask is one v_bitop3), one 64-bit add more; the result is the standard Montgomery product
// except that 0 * b comes out as p rather than 0 (the same residue; nothing on the device compares representatives).
TE_HD fp mont_mul(const fp& a, const fp& b) {
  uint32_t q[NL];
  fp r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * p_limb(k - i);
    q[k] = k == 0 ? (1u << LB) - ((uint32_t)acc & LM) : ~(uint32_t)acc & LM;
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - (NL - 1); i < NL; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - (NL - 1); i < NL; i++) acc += (uint64_t)q[i] * p_limb(k - i);
    if (k == NL) acc += 1u;
    r.v[k - NL] = (uint32_t)acc & LM;
    acc >>= LB;
  }
  r.v[NL - 1] = (uint32_t)acc;
  return r;
}

// Keeps the compiler from re-associating a column sum: left alone (mont_mul above) it sums every column from zero and
// merges the carry with an extra 64-bit add, 17 per product -- a shorter dependency chain, which is right for a single
// latency-bound product, but 8 % more instructions.  With the marker the carry is the addend of the column's first mad.
TE_HD void chain(uint64_t& acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(acc));
#else
  (void)acc;
#endif
}

// M independent products in lockstep (r[m] = mont_mul(a[m], b[m])) for throughput-bound code: 153 mads + ~45 other
// instructions per product.  The M chains are interleaved in program order, so a chain() marker is never directly
// followed by a use of its register (the compiler pads that pattern with an s_nop).
template <int M, int K>
TE_HD void mont_mul_x_col(const fp (&a)[M], const fp (&b)[M], fp (&r)[M], uint64_t (&acc)[M], uint32_t (&q)[M][NL]) {
  constexpr int lo = K < NL ? 0 : K - (NL - 1), hi = K < NL ? K : NL - 1, qhi = K < NL ? K - 1 : NL - 1;
#pragma unroll
  for (int i = lo; i <= hi; i++) {
#pragma unroll
    for (int m = 0; m < M; m++) { acc[m] += (uint64_t)a[m].v[i] * b[m].v[K - i]; chain(acc[m]); }
  }
#pragma unroll
  for (int i = lo; i <= qhi; i++) {
#pragma unroll
    for (int m = 0; m < M; m++) { acc[m] += (uint64_t)q[m][i] * p_limb(K - i); chain(acc[m]); }
  }
  if constexpr (K < NL) {
    // q*p[0] stays implicit (CARRY-FOLDED QUOTIENT above)
#pragma unroll
    for (int m = 0; m < M; m++) q[m][K] = K == 0 ? (1u << LB) - ((uint32_t)acc[m] & LM) : ~(uint32_t)acc[m] & LM;
#pragma unroll
    for (int m = 0; m < M; m++) acc[m] >>= LB;
  } else {
#pragma unroll
    for (int m = 0; m < M; m++) {
      if constexpr (K == NL) acc[m] += 1u;          // the carry the quotient columns left implicit
      r[m].v[K - NL] = (uint32_t)acc[m] & LM; acc[m] >>= LB;
    }
  }
  if constexpr (K + 1 < 2 * NL - 1) mont_mul_x_col<M, K + 1>(a, b, r, acc, q);
}
template <int M> TE_HD void mont_mul_x(const fp (&a)[M], const fp (&b)[M], fp (&r)[M]) {
  uint32_t q[M][NL];
  uint64_t acc[M];
#pragma unroll
  for (int m = 0; m < M; m++) acc[m] = 0;
  mont_mul_x_col<M, 0>(a, b, r, acc, q);
#pragma unroll
  for (int m = 0; m < M; m++) r[m].v[NL - 1] = (uint32_t)acc[m];
}

// a * 2d mod p (plus at most one p) for the curve constant 2d = 6042 of add-2008-hwcd-3 -- NOT a Montgomery product: the constant
// is 13 bits wide, so  6042 a - q p  with q = floor(6042 a / p) estimated from a's top 32 bits costs two multiply-accumulates
// per limb instead of a 153-mad product (about 50 instructions against 190; every full addition of the bucket reduction has one,
// and in a team addition it is a whole round of the dependent chain).  The factor is applied to the Montgomery form directly
// (a stands for a' R, 6042 a for 6042 a' R): no constant in Montgomery form is involved.
//   a: class N with value below 2^254 (product outputs are below 1.1 p).  Returns class N, value below 1.0001 p + 0:
//   h = a >> 222 (32 bits);  q = (h * K2D_Q) >> 49 with K2D_Q = floor(6042 * 2^49 / (floor(p / 2^222) + 1))  never exceeds
//   6042 a / p and falls short of it by less than 1 + 2^-15, so 0 <= 6042 a - q p < 1.0001 p.  One signed pass over the limbs:
//   acc += 6042 a_i - q p_i (|acc| < 2^43), the subtraction as  q (2^32 - p_i) - q 2^32  in unsigned 64-bit arithmetic.
constexpr uint32_t K2D_SMALL = 6042u;            // 2 d, d = 3021 (reference/params/AleoConstants.ts:2-4)
constexpr uint32_t K2D_Q = 0xa1d088f6u;          // checked against bigints in tests/test_host_logic.py::test_small_constant_product
TE_HD fp fp_mul_k2d(const fp& a) {
  const uint32_t h = (a.v[8] << 10) | (a.v[7] >> 19);
  const uint32_t q = (uint32_t)(((uint64_t)h * K2D_Q) >> 49);
  fp r;
  uint64_t acc = 0;                              // a signed value in two's complement
#pragma unroll
  for (int i = 0; i < NL; i++) {
    acc += (uint64_t)a.v[i] * K2D_SMALL;
    acc += (uint64_t)q * (0u - p_limb(i));       // q (2^32 - p_i) ...
    acc -= (uint64_t)q << 32;                    // ... - q 2^32 = - q p_i
    if (i < NL - 1) { r.v[i] = (uint32_t)acc & LM; acc = (uint64_t)((int64_t)acc >> LB); }
    else r.v[i] = (uint32_t)acc;
  }
  return r;
}

// limb-wise a + b (N + N -> S).  No carries.
TE_HD fp fp_add(const fp& a, const fp& b) {
  fp r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
// a - b + K*p, limb-wise (a with limbs < 2^30, b of class N with value < K*p  ->  D).
template <int K> TE_HD fp fp_sub(const fp& a, const fp& b) {
  const fp o = fp_kp_offset<K>();
  fp r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + (o.v[i] - b.v[i]);
  return r;
}
// K*p - a (a of class N, value < K*p  ->  limbs < 2^30)
template <int K> TE_HD fp fp_neg(const fp& a) {
  const fp o = fp_kp_offset<K>();
  fp r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = o.v[i] - a.v[i];
  return r;
}
// carry propagation: any limbs < 2^32  ->  class N (same value)
TE_HD fp fp_norm(const fp& a) {
  fp r; uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; i++) { const uint32_t t = a.v[i] + c; r.v[i] = t & LM; c = t >> LB; }
  r.v[NL - 1] = a.v[NL - 1] + c;
  return r;
}
// 8 little-endian 32-bit words (any 256-bit value)  ->  9 x 29-bit limbs (class N)
TE_HD fp fp_from_words32(const uint32_t (&w)[8]) {
  fp r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int bit = i * 29, j = bit >> 5, s = bit & 31;
    uint32_t v = w[j] >> s;
    if (s + 29 > 32 && j + 1 < 8) v |= w[j + 1] << (32 - s);
    r.v[i] = v & LM;
  }
  return r;
}

// ---------------------------------------------------------------------------------------------
// Constants (Montgomery form, R = 2^261), generated by tools/gen_constants.py and re-derived from bigint
// arithmetic in tests/test_host_logic.py::test_field_constants.
#include "fp_constants.inc"

}  // namespace te
