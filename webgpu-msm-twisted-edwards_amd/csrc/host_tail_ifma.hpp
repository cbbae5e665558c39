// host_tail_ifma.hpp -- the accumulator of Horner's rule (host_tail.hpp, host_tail377.hpp) on AVX-512 IFMA: ONE extended point in the
// lanes of L 512-bit registers, lane k of register i holding limb i (52 bits) of coordinate k (X, Y, Z, T).  Written once for both
// base fields: L = 5 (Twisted-Edwards BLS12, 253 bits, radix 2^260) and L = 8 (BLS12-377, 377 bits, radix 2^416).
//
// Why: the tail of an MSM is 256 dependent doublings and ~80 additions on one host core (submission.ts:362-412 does the same with
// bigints).  A doubling is two ROUNDS of four independent field products -- [X^2, Y^2, Z^2, XY], then [EF, GH, FG, EH] -- and an
// addition likewise ([A, B, D, C], then the same closing round), so with one product per lane a doubling costs two vector
// products instead of seven scalar ones executed back to back: measured on the GPU box's EPYC 9575F ~60 ns against 146 ns with the
// mulx/adcx form (the tail of a 16 x 16-bit MSM in 35 us instead of 59) and, for BLS12-377, against 444 ns of portable code.
//
// Arithmetic: limbs of 52 bits, Montgomery radix R' = 2^(52 L) (vpmadd52luq / vpmadd52huq multiply the LOW 52 bits of their
// operands: operands must be normalised, limbs < 2^52).  Values are only kept below 2^(52 L - 4): a product of two such values comes
// out below 2 p, sums and offset differences of a few products stay below 8 p -- nothing is reduced modulo p until the result
// leaves the lanes.  The accumulator's coordinates may carry any common factor (a projective point does not change; T Z = X Y still
// holds), so device rows and host elements are taken over as plain integers; the curve constant 2 d is applied on the scalar side.
// Checked against the scalar accumulator once per process (tail_selftest) and by the host-logic tests; used when the CPU has
// AVX-512 IFMA (TE_MSM_HOST_TAIL=scalar forces the scalar form for A/B measurements).
#pragma once
#if defined(__x86_64__)
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>

namespace te_ifma {

#define TE_IFMA __attribute__((target("avx512f,avx512ifma,avx512dq,avx512vl"))) static inline

template <int L> struct V { __m512i l[L]; };     // lane k: coordinate k of the point (X, Y, Z, T; lanes 4..7 mirror 0..3)
static const uint64_t M52 = (1ull << 52) - 1;

// constants of a field, 52-bit limbs: the modulus, -p^-1 mod 2^52, 2 p and 4 p
template <int L> struct field52 { uint64_t p[L]; uint64_t ninv; uint64_t p2[L], p4[L]; };

template <int L> TE_IFMA V<L> splat(const uint64_t* c) { V<L> r; for (int i = 0; i < L; i++) r.l[i] = _mm512_set1_epi64((long long)c[i]); return r; }
template <int L> TE_IFMA V<L> vadd(const V<L>& a, const V<L>& b) { V<L> r; for (int i = 0; i < L; i++) r.l[i] = _mm512_add_epi64(a.l[i], b.l[i]); return r; }
template <int L> TE_IFMA V<L> vsub(const V<L>& a, const V<L>& b) { V<L> r; for (int i = 0; i < L; i++) r.l[i] = _mm512_sub_epi64(a.l[i], b.l[i]); return r; }
template <int L> TE_IFMA V<L> vperm(const V<L>& a, __m512i idx) { V<L> r; for (int i = 0; i < L; i++) r.l[i] = _mm512_permutexvar_epi64(idx, a.l[i]); return r; }
template <int L> TE_IFMA V<L> vblend(__mmask8 k, const V<L>& a, const V<L>& b) { V<L> r; for (int i = 0; i < L; i++) r.l[i] = _mm512_mask_blend_epi64(k, a.l[i], b.l[i]); return r; }   // bit set: b
// carry propagation over signed limbs (differences may leave negative limbs; the value itself is never negative): limbs 0..L-2 -> [0, 2^52)
template <int L> TE_IFMA V<L> vnorm(V<L> a) {
  const __m512i m = _mm512_set1_epi64((long long)M52);
  for (int i = 0; i < L - 1; i++) {
    const __m512i c = _mm512_srai_epi64(a.l[i], 52);
    a.l[i] = _mm512_and_si512(a.l[i], m);
    a.l[i + 1] = _mm512_add_epi64(a.l[i + 1], c);
  }
  return a;
}
// lane-wise a * b / 2^(52 L) mod p (+ a multiple of p): normalised operands with values below 2^(52 L - 4) -> normalised, below 2 p
template <int L> TE_IFMA V<L> vmul(const V<L>& a, const V<L>& b, const field52<L>& f) {
  const __m512i z = _mm512_setzero_si512();
  __m512i t[2 * L];
  for (int i = 0; i < 2 * L; i++) t[i] = z;
  for (int i = 0; i < L; i++)
    for (int j = 0; j < L; j++) {
      t[i + j] = _mm512_madd52lo_epu64(t[i + j], a.l[j], b.l[i]);
      t[i + j + 1] = _mm512_madd52hi_epu64(t[i + j + 1], a.l[j], b.l[i]);
    }
  const __m512i ninv = _mm512_set1_epi64((long long)f.ninv);
  for (int i = 0; i < L; i++) {
    const __m512i q = _mm512_madd52lo_epu64(z, t[i], ninv);          // (t_i mod 2^52) * (-p^-1) mod 2^52
    for (int j = 0; j < L; j++) {
      const __m512i pj = _mm512_set1_epi64((long long)f.p[j]);
      t[i + j] = _mm512_madd52lo_epu64(t[i + j], q, pj);
      t[i + j + 1] = _mm512_madd52hi_epu64(t[i + j + 1], q, pj);
    }
    t[i + 1] = _mm512_add_epi64(t[i + 1], _mm512_srli_epi64(t[i], 52));     // t_i is 0 mod 2^52 now
  }
  V<L> r;
  for (int i = 0; i < L; i++) r.l[i] = t[L + i];
  return vnorm<L>(r);
}

// ---- field elements (NW little-endian 64-bit words, canonical) <-> lanes
template <int L, int NW> static inline void words_to_limbs(const uint64_t* w, uint64_t* o) {
  for (int i = 0; i < L; i++) {
    const int bit = 52 * i, j = bit >> 6, s = bit & 63;
    uint64_t v = j < NW ? w[j] >> s : 0;
    if (s > 12 && j + 1 < NW) v |= w[j + 1] << (64 - s);
    o[i] = v & M52;
  }
}
template <int L, int NW> static inline void limbs_to_words(const uint64_t* l, uint64_t* w) {      // normalised limbs, value below 2^(64 NW)
  for (int j = 0; j < NW; j++) w[j] = 0;
  for (int i = 0; i < L; i++) {
    const int bit = 52 * i, j = bit >> 6, s = bit & 63;
    if (j < NW) w[j] |= l[i] << s;
    if (s > 12 && j + 1 < NW) w[j + 1] |= l[i] >> (64 - s);
  }
}
template <int L, int NW> TE_IFMA V<L> from_words(const uint64_t* c0, const uint64_t* c1, const uint64_t* c2, const uint64_t* c3) {
  uint64_t q[4][L];
  words_to_limbs<L, NW>(c0, q[0]); words_to_limbs<L, NW>(c1, q[1]); words_to_limbs<L, NW>(c2, q[2]); words_to_limbs<L, NW>(c3, q[3]);
  alignas(64) uint64_t buf[L][8];
  for (int i = 0; i < L; i++) for (int k = 0; k < 8; k++) buf[i][k] = q[k & 3][i];
  V<L> r; for (int i = 0; i < L; i++) r.l[i] = _mm512_load_si512(buf[i]);
  return r;
}
// the four coordinates as NW-word integers (values below 2 p: the caller reduces)
template <int L, int NW> TE_IFMA void to_words(const V<L>& a, uint64_t out[4][NW]) {
  alignas(64) uint64_t buf[L][8];
  for (int i = 0; i < L; i++) _mm512_store_si512(buf[i], a.l[i]);
  for (int k = 0; k < 4; k++) { uint64_t l[L]; for (int i = 0; i < L; i++) l[i] = buf[i][k]; limbs_to_words<L, NW>(l, out[k]); }
}

// the closing round shared by doubling and addition: with E, F, G, H in all lanes, (X3, Y3, Z3, T3) = (E F, G H, F G, E H)
template <int L> TE_IFMA V<L> close_round(const V<L>& E, const V<L>& F, const V<L>& G, const V<L>& H, const field52<L>& f) {
  const V<L> a = vblend<L>(0x44, vblend<L>(0x22, E, G), F);            // [E, G, F, E]
  const V<L> b = vblend<L>(0x44, vblend<L>(0x11, H, F), G);            // [F, H, G, H]
  return vmul<L>(vnorm<L>(a), vnorm<L>(b), f);
}
template <int L> TE_IFMA V<L> lane(const V<L>& a, int k) { return vperm<L>(a, _mm512_set1_epi64(k)); }

// dbl-2008-hwcd, a = -1:  [A, B, C0, XY] = [X^2, Y^2, Z^2, X Y];  E = 2 XY, G = B - A, F = G - 2 C0, H = -A - B
template <int L> TE_IFMA V<L> vdbl(const V<L>& P, const field52<L>& f) {
  const V<L> r1 = vmul<L>(vperm<L>(P, _mm512_setr_epi64(0, 1, 2, 0, 0, 1, 2, 0)), vperm<L>(P, _mm512_setr_epi64(0, 1, 2, 1, 0, 1, 2, 1)), f);
  const V<L> A = lane<L>(r1, 0), B = lane<L>(r1, 1), C0 = lane<L>(r1, 2), XY = lane<L>(r1, 3);
  const V<L> p2 = splat<L>(f.p2), p4 = splat<L>(f.p4);
  const V<L> E = vadd<L>(XY, XY);                                          // < 4 p
  const V<L> G = vadd<L>(vsub<L>(B, A), p2);                               // (0, 4 p)
  const V<L> F = vadd<L>(vsub<L>(G, vadd<L>(C0, C0)), p4);                 // (0, 8 p)
  const V<L> H = vsub<L>(p4, vadd<L>(A, B));                               // (0, 4 p]
  return close_round<L>(E, F, G, H, f);
}
// add-2008-hwcd-3, a = -1, with the operand prepared on the scalar side: Q = lanes [Y2 - X2, Y2 + X2, 2 Z2, 2 d T2] (canonical)
//   [A, B, D, C] = [(Y1-X1)(Y2-X2), (Y1+X1)(Y2+X2), Z1 2Z2, T1 2dT2];  E = B - A, H = B + A, F = D - C, G = D + C
template <int L> TE_IFMA V<L> vaddp(const V<L>& P, const V<L>& Q, const field52<L>& f) {
  // first operand: lanes [Y - X + 2p, Y + X, Z, T]
  const V<L> pY = vperm<L>(P, _mm512_setr_epi64(1, 1, 2, 3, 1, 1, 2, 3));
  const V<L> p2 = splat<L>(f.p2);
  V<L> sx;
  for (int i = 0; i < L; i++) {
    const __m512i pX = _mm512_maskz_permutexvar_epi64(0x33, _mm512_setzero_si512(), P.l[i]);       // [X, X, 0, 0]
    sx.l[i] = _mm512_mask_sub_epi64(pX, 0x11, p2.l[i], pX);                                        // [2p - X, X, 0, 0]
  }
  const V<L> r1 = vmul<L>(vnorm<L>(vadd<L>(pY, sx)), Q, f);
  const V<L> A = lane<L>(r1, 0), B = lane<L>(r1, 1), D = lane<L>(r1, 2), C = lane<L>(r1, 3);
  const V<L> E = vadd<L>(vsub<L>(B, A), p2), H = vadd<L>(B, A), F = vadd<L>(vsub<L>(D, C), p2), G = vadd<L>(D, C);
  return close_round<L>(E, F, G, H, f);
}

static inline bool cpu_has_ifma() {       // TE_MSM_HOST_TAIL=scalar forces the scalar accumulators (A/B measurements)
  static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512dq") &&
                        __builtin_cpu_supports("avx512vl") && !(getenv("TE_MSM_HOST_TAIL") && getenv("TE_MSM_HOST_TAIL")[0] == 's');
  return v;
}

}  // namespace te_ifma
#endif
