// fq377.hpp -- base field of BLS12-377 G1 (377 bits) for CDNA4, BASELINE config 5.
//
// The reference's BLS12-377 variant differs from the Twisted-Edwards one only in "number of limbs per point coordinate"
// (30 x 13 bits) and in the group law (README.md:279-287); it is not in the reference tree.  Same scheme as fp.hpp:
// 29-bit limbs in u32 words, 64-bit column accumulators, Montgomery form, no reduction modulo q on the device --
// here 14 limbs, R = 2^406 (q needs 13 limbs; the 14th gives 29 bits of headroom: values may grow to thousands of q).
// q = 1 (mod 2^29) as well (2-adicity 46), so the quotient digit is again a negation.
//
// LIMB RULE.  A product is exact while 14 * max(a_i) * max(b_j) + 13 * 2^58 < 2^64, i.e. max(a_i) * max(b_j) < 2^59.8:
// one operand must be normalised (class N, limbs < 2^29), the other may have limbs up to 2^30.8.  Sums of two
// normalised values (limbs < 2^30) and differences in offset form (< 2^30.6) qualify; a product of two sums does not --
// curve377.hpp normalises one side.  The checker build verifies every column against 2^64 (tests/csrc/fq377check.cpp).
#pragma once
#include <stdint.h>
#include "fp.hpp"      // TE_HD, chain()

namespace te377 {

using te::chain;

constexpr int NL = 14;
constexpr uint32_t LB = 29;
constexpr uint32_t LM = (1u << LB) - 1u;

using fq = te::fel<14>;          // little-endian limbs, 56 bytes

// q, 32-bit words (README.md:65-67)
constexpr uint32_t Q_W32[12] = {0x00000001u, 0x8508c000u, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu,
                                0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu, 0xc63b05c0u, 0x17c510eau, 0x01ae3a46u};
constexpr uint32_t q29(int i) {
  const int bit = i * 29, w = bit >> 5, s = bit & 31;
  if (w >= 12) return 0u;
  uint64_t two = Q_W32[w];
  if (w + 1 < 12) two |= (uint64_t)Q_W32[w + 1] << 32;
  return (uint32_t)(two >> s) & ((1u << 29) - 1u);
}
template <int I> struct q_limb_c { static constexpr uint32_t value = q29(I); };
TE_HD uint32_t q_limb(int i) {
  switch (i) {
    case 0: return q29(0); case 1: return q29(1); case 2: return q29(2); case 3: return q29(3); case 4: return q29(4);
    case 5: return q29(5); case 6: return q29(6); case 7: return q29(7); case 8: return q29(8); case 9: return q29(9);
    case 10: return q29(10); case 11: return q29(11); case 12: return q29(12); default: return q29(13);
  }
}

TE_HD fq fq_zero() { fq r; for (int i = 0; i < NL; i++) r.v[i] = 0; return r; }
template <int K> TE_HD fq fq_kq_offset();     // K*q in offset form (limbs 0..12 raised by 2^29, the next lowered by 1)

#if defined(TE377_CHECK_COLUMNS)
// checker build (tests/csrc/fq377check.cpp): every column sum is checked against 2^64 -- a violated limb rule would wrap
// silently on the device
}  // namespace te377
extern "C" int g_fq377_overflow;
namespace te377 {
struct col_acc {
  unsigned __int128 wide = 0; uint64_t acc = 0;
  void mad(uint32_t a, uint32_t b) { wide += (unsigned __int128)a * b; acc += (uint64_t)a * b; if (wide >> 64) g_fq377_overflow = 1; }
  void shift() { wide >>= LB; acc >>= LB; }
};
#endif

// M independent Montgomery products in lockstep, r[m] = a[m] * b[m] / R (mod q, plus a multiple of q), class N, value
// < a*b/R + q.  See fp.hpp mont_mul_x for the scheduling rationale.
template <int M, int K>
TE_HD void mont_mul_x_col(const fq (&a)[M], const fq (&b)[M], fq (&r)[M], uint64_t (&acc)[M], uint32_t (&q)[M][NL]) {
  constexpr int lo = K < NL ? 0 : K - (NL - 1), hi = K < NL ? K : NL - 1, qhi = K < NL ? K - 1 : NL - 1;
#pragma unroll
  for (int i = lo; i <= hi; i++) {
#pragma unroll
    for (int m = 0; m < M; m++) { acc[m] += (uint64_t)a[m].v[i] * b[m].v[K - i]; chain(acc[m]); }
  }
#pragma unroll
  for (int i = lo; i <= qhi; i++) {
#pragma unroll
    for (int m = 0; m < M; m++) { acc[m] += (uint64_t)q[m][i] * q_limb(K - i); chain(acc[m]); }
  }
  if constexpr (K < NL) {
#pragma unroll
    for (int m = 0; m < M; m++) q[m][K] = K == 0 ? (1u << LB) - ((uint32_t)acc[m] & LM) : ~(uint32_t)acc[m] & LM;
#pragma unroll
    for (int m = 0; m < M; m++) acc[m] >>= LB;       // q*q_limb(0) = q stays implicit: fp.hpp, CARRY-FOLDED QUOTIENT
  } else {
#pragma unroll
    for (int m = 0; m < M; m++) {
      if constexpr (K == NL) acc[m] += 1u;
      r[m].v[K - NL] = (uint32_t)acc[m] & LM; acc[m] >>= LB;
    }
  }
  if constexpr (K + 1 < 2 * NL - 1) mont_mul_x_col<M, K + 1>(a, b, r, acc, q);
}
template <int M> TE_HD void mont_mul_x(const fq (&a)[M], const fq (&b)[M], fq (&r)[M]) {
#if defined(TE377_CHECK_COLUMNS)
  for (int m = 0; m < M; m++) {            // same arithmetic, product by product, with overflow detection
    uint32_t q[NL]; col_acc c;
    for (int k = 0; k < NL; k++) {
      for (int i = 0; i <= k; i++) c.mad(a[m].v[i], b[m].v[k - i]);
      for (int i = 0; i < k; i++) c.mad(q[i], q_limb(k - i));
      q[k] = k == 0 ? (1u << LB) - ((uint32_t)c.acc & LM) : ~(uint32_t)c.acc & LM;
      c.shift();
    }
    for (int k = NL; k < 2 * NL - 1; k++) {
      for (int i = k - (NL - 1); i < NL; i++) c.mad(a[m].v[i], b[m].v[k - i]);
      for (int i = k - (NL - 1); i < NL; i++) c.mad(q[i], q_limb(k - i));
      if (k == NL) c.mad(1u, 1u);
      r[m].v[k - NL] = (uint32_t)c.acc & LM;
      c.shift();
    }
    r[m].v[NL - 1] = (uint32_t)c.acc;
  }
#else
  uint32_t q[M][NL];
  uint64_t acc[M];
#pragma unroll
  for (int m = 0; m < M; m++) acc[m] = 0;
  mont_mul_x_col<M, 0>(a, b, r, acc, q);
#pragma unroll
  for (int m = 0; m < M; m++) r[m].v[NL - 1] = (uint32_t)acc[m];
#endif
}
TE_HD fq mont_mul(const fq& a, const fq& b) {
  const fq aa[1] = {a}, bb[1] = {b};
  fq r[1];
  mont_mul_x<1>(aa, bb, r);
  return r[0];
}

// limb-wise a + b.  No carries.
TE_HD fq fq_add(const fq& a, const fq& b) {
  fq r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
// a - b + K*q, limb-wise; b must be class N (limbs < 2^29) with value < K*q
template <int K> TE_HD fq fq_sub(const fq& a, const fq& b) {
  const fq o = fq_kq_offset<K>();
  fq r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + (o.v[i] - b.v[i]);
  return r;
}
template <int K> TE_HD fq fq_neg(const fq& a) {
  const fq o = fq_kq_offset<K>();
  fq r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = o.v[i] - a.v[i];
  return r;
}
// carry propagation: any limbs < 2^32 -> class N (same value; the top limb takes what is left)
TE_HD fq fq_norm(const fq& a) {
  fq r; uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; i++) { const uint32_t t = a.v[i] + c; r.v[i] = t & LM; c = t >> LB; }
  r.v[NL - 1] = a.v[NL - 1] + c;
  return r;
}
// a * 3, limb-wise (N -> limbs < 2^30.6)
TE_HD fq fq_mul3(const fq& a) {
  fq r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] * 3u;
  return r;
}
// 12 little-endian 32-bit words (any 384-bit value) -> 14 limbs (class N)
TE_HD fq fq_from_words32(const uint32_t (&w)[12]) {
  fq r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int bit = i * 29, j = bit >> 5, s = bit & 31;
    uint32_t v = j < 12 ? w[j] >> s : 0u;
    if (s + 29 > 32 && j + 1 < 12) v |= w[j + 1] << (32 - s);
    r.v[i] = v & LM;
  }
  return r;
}

#include "fq377_constants.inc"

}  // namespace te377
