// host_tail_ifma.hpp -- the accumulator of Horner's rule (host_tail.hpp) on AVX-512 IFMA: ONE extended point in the lanes of five
// 512-bit registers, lane k of register i holding limb i (52 bits) of coordinate k (X, Y, Z, T).
//
// Why: the tail of an MSM is 256 dependent doublings and ~80 additions on one host core (submission.ts:362-412 does the same with
// bigints).  A doubling is two ROUNDS of four independent field products -- [X^2, Y^2, Z^2, XY], then [EF, GH, FG, EH] -- and an
// addition likewise ([A, B, D, C], then the same closing round), so with one product per lane a doubling costs two vector
// products instead of seven scalar ones executed back to back: measured ~45-60 ns against 110-150 ns with the mulx/adcx form,
// i.e. the tail of a 16 x 16-bit MSM in ~25 us instead of 45-65 (4 % of the latency of a 2^20-point MSM, 15 % at 2^16).
//
// Arithmetic: 5 limbs of 52 bits, Montgomery radix R' = 2^260 (vpmadd52luq / vpmadd52huq multiply the LOW 52 bits of their operands:
// operands must be normalised, limbs < 2^52).  Values are only kept below 2^256 = 13.7 p: a product of two such values comes out
// below 2 p, sums and offset differences of a few products stay below 8 p -- nothing is reduced modulo p until the result leaves
// the lanes.  The accumulator's coordinates may carry any common factor (a projective point does not change; T Z = X Y still
// holds), so device rows and host elements are taken over as plain integers; only the curve constant 2 d is in Montgomery form for R'.
// Checked against the scalar accumulator at start-up (tail_selftest) and by the host-logic tests; used when the CPU has AVX-512 IFMA
// (TE_MSM_HOST_TAIL=scalar forces the scalar form for A/B measurements).
#pragma once
#if defined(__x86_64__)
#include <immintrin.h>

namespace te_host {
namespace ifma {

#define TE_IFMA __attribute__((target("avx512f,avx512ifma,avx512dq,avx512vl"))) static inline

struct V { __m512i l[5]; };                      // lane k: coordinate k of the point (X, Y, Z, T; lanes 4..7 mirror 0..3)

static const uint64_t M52 = (1ull << 52) - 1;
static const uint64_t P52[5] = {0x1800000000001ULL, 0xfed00000010a1ULL, 0xc37b00159aa76ULL, 0xa55660b44d1e5ULL, 0x12ab655e9a2cULL};
static const uint64_t NINV52 = 0x17fffffffffffULL;                     // -p^-1 mod 2^52
static const uint64_t K2D52[5] = {0x67fffffebc5efULL, 0xf42feafa485ddULL, 0x252004720d01dULL, 0xe544b42ecbbd8ULL, 0x242ca4cbad3ULL};   // 2 d 2^260 mod p
static const uint64_t P2_52[5] = {0x3000000000002ULL, 0xfda0000002142ULL, 0x86f6002b354edULL, 0x4aacc1689a3cbULL, 0x2556cabd3459ULL};   // 2 p
static const uint64_t P4_52[5] = {0x6000000000004ULL, 0xfb40000004284ULL, 0xdec00566a9dbULL, 0x955982d134797ULL, 0x4aad957a68b2ULL};    // 4 p

TE_IFMA V splat(const uint64_t c[5]) { V r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_set1_epi64((long long)c[i]); return r; }
TE_IFMA V vadd(const V& a, const V& b) { V r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_add_epi64(a.l[i], b.l[i]); return r; }
TE_IFMA V vsub(const V& a, const V& b) { V r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_sub_epi64(a.l[i], b.l[i]); return r; }
TE_IFMA V vperm(const V& a, __m512i idx) { V r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_permutexvar_epi64(idx, a.l[i]); return r; }
TE_IFMA V vblend(__mmask8 k, const V& a, const V& b) { V r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_mask_blend_epi64(k, a.l[i], b.l[i]); return r; }   // bit set: b
// carry propagation over signed limbs (differences may leave negative limbs; the value itself is never negative): limbs 0..3 -> [0, 2^52)
TE_IFMA V vnorm(V a) {
  const __m512i m = _mm512_set1_epi64((long long)M52);
  for (int i = 0; i < 4; i++) {
    const __m512i c = _mm512_srai_epi64(a.l[i], 52);
    a.l[i] = _mm512_and_si512(a.l[i], m);
    a.l[i + 1] = _mm512_add_epi64(a.l[i + 1], c);
  }
  return a;
}
// lane-wise a * b / 2^260 mod p (+ a multiple of p): normalised operands with values below 2^256 -> normalised, below 2 p
TE_IFMA V vmul(const V& a, const V& b) {
  const __m512i z = _mm512_setzero_si512();
  __m512i t[10] = {z, z, z, z, z, z, z, z, z, z};
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) {
      t[i + j] = _mm512_madd52lo_epu64(t[i + j], a.l[j], b.l[i]);
      t[i + j + 1] = _mm512_madd52hi_epu64(t[i + j + 1], a.l[j], b.l[i]);
    }
  const __m512i ninv = _mm512_set1_epi64((long long)NINV52);
  __m512i p[5];
  for (int j = 0; j < 5; j++) p[j] = _mm512_set1_epi64((long long)P52[j]);
  for (int i = 0; i < 5; i++) {
    const __m512i q = _mm512_madd52lo_epu64(z, t[i], ninv);          // (t_i mod 2^52) * (-p^-1) mod 2^52
    for (int j = 0; j < 5; j++) {
      t[i + j] = _mm512_madd52lo_epu64(t[i + j], q, p[j]);
      t[i + j + 1] = _mm512_madd52hi_epu64(t[i + j + 1], q, p[j]);
    }
    t[i + 1] = _mm512_add_epi64(t[i + 1], _mm512_srli_epi64(t[i], 52));     // t_i is 0 mod 2^52 now
  }
  V r;
  for (int i = 0; i < 5; i++) r.l[i] = t[5 + i];
  return vnorm(r);
}

// ---- one field element <-> lanes
static inline void fe_to_limbs(const Fe& a, uint64_t o[5]) {
  o[0] = a.l[0] & M52; o[1] = ((a.l[0] >> 52) | (a.l[1] << 12)) & M52; o[2] = ((a.l[1] >> 40) | (a.l[2] << 24)) & M52;
  o[3] = ((a.l[2] >> 28) | (a.l[3] << 36)) & M52; o[4] = a.l[3] >> 16;
}
// limbs of a value below 2 p (normalised) -> canonical element
static inline Fe limbs_to_fe(const uint64_t l[5]) {
  Fe r;
  r.l[0] = l[0] | (l[1] << 52); r.l[1] = (l[1] >> 12) | (l[2] << 40); r.l[2] = (l[2] >> 24) | (l[3] << 28); r.l[3] = (l[3] >> 36) | (l[4] << 16);
  return reduce_once(r);
}
TE_IFMA V from_fes(const Fe& c0, const Fe& c1, const Fe& c2, const Fe& c3) {
  uint64_t q[4][5]; fe_to_limbs(c0, q[0]); fe_to_limbs(c1, q[1]); fe_to_limbs(c2, q[2]); fe_to_limbs(c3, q[3]);
  alignas(64) uint64_t buf[5][8];
  for (int i = 0; i < 5; i++) for (int k = 0; k < 8; k++) buf[i][k] = q[k & 3][i];
  V r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_load_si512(buf[i]);
  return r;
}
TE_IFMA void to_fes(const V& a, Fe out[4]) {
  alignas(64) uint64_t buf[5][8];
  for (int i = 0; i < 5; i++) _mm512_store_si512(buf[i], a.l[i]);
  for (int k = 0; k < 4; k++) { const uint64_t l[5] = {buf[0][k], buf[1][k], buf[2][k], buf[3][k], buf[4][k]}; out[k] = limbs_to_fe(l); }
}

// the closing round shared by doubling and addition: with E, F, G, H in all lanes, (X3, Y3, Z3, T3) = (E F, G H, F G, E H)
TE_IFMA V close_round(const V& E, const V& F, const V& G, const V& H) {
  const V a = vblend(0x44, vblend(0x22, E, G), F);            // [E, G, F, E]
  const V b = vblend(0x44, vblend(0x11, H, F), G);            // [F, H, G, H]
  return vmul(vnorm(a), vnorm(b));
}
TE_IFMA V lane(const V& a, int k) { return vperm(a, _mm512_set1_epi64(k)); }

// dbl-2008-hwcd, a = -1:  [A, B, C0, XY] = [X^2, Y^2, Z^2, X Y];  E = 2 XY, G = B - A, F = G - 2 C0, H = -A - B
TE_IFMA V vdbl(const V& P) {
  const V r1 = vmul(vperm(P, _mm512_setr_epi64(0, 1, 2, 0, 0, 1, 2, 0)), vperm(P, _mm512_setr_epi64(0, 1, 2, 1, 0, 1, 2, 1)));
  const V A = lane(r1, 0), B = lane(r1, 1), C0 = lane(r1, 2), XY = lane(r1, 3);
  const V p2 = splat(P2_52), p4 = splat(P4_52);
  const V E = vadd(XY, XY);                                    // < 4 p
  const V G = vadd(vsub(B, A), p2);                            // (0, 4 p)
  const V F = vadd(vsub(G, vadd(C0, C0)), p4);                 // (0, 8 p)
  const V H = vsub(p4, vadd(A, B));                            // (0, 4 p]
  return close_round(E, F, G, H);
}
// the operand of an addition, prepared on the scalar side: lanes [Y - X, Y + X, 2 Z, 2 d T] of the point to add
struct Prepared { V v; };
TE_IFMA Prepared prepare(const Pt& q, const Fe& k2d_host /* 2 d in the host's Montgomery form (R = 2^256) */) {
  Prepared r;
  r.v = from_fes(sub(q.y, q.x), add(q.y, q.x), add(q.z, q.z), mul(q.t, k2d_host));
  return r;
}
// add-2008-hwcd-3, a = -1:  [A, B, D, C] = [(Y1-X1)(Y2-X2), (Y1+X1)(Y2+X2), Z1 2Z2, T1 2dT2];  E = B - A, H = B + A, F = D - C, G = D + C
TE_IFMA V vaddp(const V& P, const Prepared& Q) {
  // first operand: lanes [Y - X + 2p, Y + X, Z, T]
  const V pY = vperm(P, _mm512_setr_epi64(1, 1, 2, 3, 1, 1, 2, 3));
  V pX;
  for (int i = 0; i < 5; i++) pX.l[i] = _mm512_maskz_permutexvar_epi64(0x33, _mm512_setzero_si512(), P.l[i]);        // [X, X, 0, 0]
  const V p2 = splat(P2_52);
  V sx;
  for (int i = 0; i < 5; i++) sx.l[i] = _mm512_mask_sub_epi64(pX.l[i], 0x11, p2.l[i], pX.l[i]);                      // [2p - X, X, 0, 0]
  const V r1 = vmul(vnorm(vadd(pY, sx)), Q.v);
  const V A = lane(r1, 0), B = lane(r1, 1), D = lane(r1, 2), C = lane(r1, 3);
  const V E = vadd(vsub(B, A), p2), H = vadd(B, A), F = vadd(vsub(D, C), p2), G = vadd(D, C);
  return close_round(E, F, G, H);
}
TE_IFMA V videntity() { const Fe zero = {{0, 0, 0, 0}}, one = {{1, 0, 0, 0}}; return from_fes(zero, one, one, zero); }

static inline bool available() {          // TE_MSM_HOST_TAIL=scalar forces the scalar accumulator (A/B measurements)
  static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512dq") &&
                        __builtin_cpu_supports("avx512vl") && have_adx() && !(getenv("TE_MSM_HOST_TAIL") && getenv("TE_MSM_HOST_TAIL")[0] == 's');
  return v;
}

}  // namespace ifma
}  // namespace te_host
#endif
