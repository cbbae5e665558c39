// host_tail377.hpp -- the CPU tail for BLS12-377 G1: the same Horner fold over per-window rows [T | W0 | W1 | W2 | W3] as
// host_tail.hpp (which follows submission.ts:362-412).  The device works in the twisted-Edwards form of the curve
// (curve.hpp), so the fold uses the same extended-coordinate formulas as the other curve; the result is mapped back to
// the short-Weierstrass affine point the caller expects.  Host arithmetic: 6 x 64-bit limbs, Montgomery form R = 2^384;
// the device writes 14 x 29-bit limbs in Montgomery form R' = 2^406, lazily reduced.
#pragma once
#include <stdint.h>
#include <string.h>
#include "fq377.hpp"     // TE377_HOST_* constants (fq377_constants.inc)
#include "host_tail_ifma.hpp"

namespace te377_host {

typedef unsigned __int128 u128;
struct Fe { uint64_t l[6]; };
struct Pt { Fe x, y, z, t; };    // extended twisted Edwards (X : Y : Z : T) of the curve's Edwards form; identity (0 : 1 : 1 : 0)

static const uint64_t MOD[6] = {0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL,
                                0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x01ae3a4617c510eaULL};      // README.md:65-67
static const uint64_t MOD_NEG_INV = 0x8508bfffffffffffULL;   // -q^-1 mod 2^64 (checked in tail_selftest)
static const Fe ONE_M = {{0x02cdffffffffff68ULL, 0x51409f837fffffb1ULL, 0x9f7db3a98a7d3ff2ULL, 0x7b4e97b76e7c6305ULL, 0x4cf495bf803c84e8ULL, 0x008d6661e2fdf49aULL}};   // R mod q
static inline bool ge_mod(const Fe& a) { for (int i = 5; i >= 0; i--) { if (a.l[i] != MOD[i]) return a.l[i] > MOD[i]; } return true; }
static inline void sub_mod_raw(Fe& a) { uint64_t br = 0; for (int i = 0; i < 6; i++) { u128 d = (u128)a.l[i] - MOD[i] - br; a.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
// r < 2q -> r mod q without a branch (the comparison with q is data-dependent: a mispredicted branch per field operation costs
// more than the operation -- host_tail.hpp measured 240 -> 130 ns per doubling with the same change)
static inline Fe reduce_once(const Fe& r) {
  Fe d; uint64_t br = 0;
  for (int i = 0; i < 6; i++) { const u128 x = (u128)r.l[i] - MOD[i] - br; d.l[i] = (uint64_t)x; br = (uint64_t)(x >> 64) & 1; }
  const uint64_t keep = (uint64_t)0 - br;            // all ones when r < q
  Fe o;
  for (int i = 0; i < 6; i++) o.l[i] = (r.l[i] & keep) | (d.l[i] & ~keep);
  return o;
}
static inline Fe add(const Fe& a, const Fe& b) {
  Fe r; u128 c = 0;
  for (int i = 0; i < 6; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  return reduce_once(r);             // a, b < q < 2^377: no carry out
}
static inline Fe sub(const Fe& a, const Fe& b) {
  Fe r; uint64_t br = 0;
  for (int i = 0; i < 6; i++) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  const uint64_t m = (uint64_t)0 - br;               // borrow: add q back
  u128 c = 0;
  for (int i = 0; i < 6; i++) { c += (u128)r.l[i] + (MOD[i] & m); r.l[i] = (uint64_t)c; c >>= 64; }
  return r;
}
// a * b / R mod q for b < q and ANY a < 2^384 (result < 2q before the final subtraction)
static inline Fe mul(const Fe& a, const Fe& b) {
  uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 6; i++) {
    u128 c = 0;
    for (int j = 0; j < 6; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * MOD_NEG_INV;
    c = ((u128)m * MOD[0] + t[0]) >> 64;
    for (int j = 1; j < 6; j++) { c += (u128)m * MOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
  }
  Fe r; memcpy(r.l, t, 48);
  if (t[6] || ge_mod(r)) sub_mod_raw(r);
  return r;
}
static inline Fe inv(const Fe& a) {            // a^(q-2)
  uint64_t e[6]; memcpy(e, MOD, 48); e[0] -= 2;
  Fe acc = ONE_M, base = a;
  for (int i = 0; i < 377; i++) { if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base); base = mul(base, base); }
  return acc;
}
static inline bool is_zero(const Fe& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
static inline Pt identity() { Pt r; memset(&r, 0, sizeof r); r.y = ONE_M; r.z = ONE_M; return r; }
static inline bool all_zero_bytes(const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i]) return false; return true; }

#define TE377_TAIL_COORD_BYTES 56
#define TE377_TAIL_POINT_BYTES 224
#define TE377_TAIL_ROW_BYTES 1120
// one coordinate: 14 u32 words holding 29-bit limbs (possibly unnormalised), value < 2^410
static inline Fe load_coord(const uint8_t* src) {
  uint32_t l[14]; memcpy(l, src, 56);
  uint64_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 14; i++) {
    const int bit = 29 * i, j = bit >> 6, s = bit & 63;
    const u128 a = (u128)l[i] << s;
    u128 c = (u128)w[j] + (uint64_t)a; w[j] = (uint64_t)c; c >>= 64;
    c += (u128)w[j + 1] + (uint64_t)(a >> 64); w[j + 1] = (uint64_t)c; c >>= 64;
    for (int q = j + 2; q < 8 && c; q++) { c += w[q]; w[q] = (uint64_t)c; c >>= 64; }
  }
  // v = lo + hi 2^384 is taken over up to a COMMON factor of the point's four coordinates (a projective point does not change,
  // T Z = X Y still holds): v / 2^384 = mul(lo, 1) + hi, one product instead of the two of an exact domain change (hi < 2^26)
  Fe lo, hi, one_raw; memcpy(lo.l, w, 48); memset(&hi, 0, sizeof hi); hi.l[0] = w[6]; hi.l[1] = w[7];
  memset(&one_raw, 0, sizeof one_raw); one_raw.l[0] = 1;
  return add(mul(lo, one_raw), hi);
}
// device point: x | y | z | t, 56 bytes each
static inline Pt load_point(const uint8_t* src) { Pt r; r.x = load_coord(src); r.y = load_coord(src + 56); r.z = load_coord(src + 112); r.t = load_coord(src + 168); return r; }

static const Fe K2D = {TE377_HOST_K2D};        // 2 d of the Edwards form (fq377_constants.inc)
static const Fe F_M = {TE377_HOST_F};          // f = sqrt(-(A+2)/B)
static const Fe SQRT3_M = {TE377_HOST_SQRT3};  // 1 / s
// unified addition, a = -1, k = 2d (add-2008-hwcd-3)
static inline Pt padd(const Pt& a, const Pt& b) {
  const Fe A = mul(sub(a.y, a.x), sub(b.y, b.x));
  const Fe B = mul(add(a.y, a.x), add(b.y, b.x));
  const Fe C = mul(mul(a.t, b.t), K2D);
  const Fe zz = mul(a.z, b.z);
  const Fe D = add(zz, zz);
  const Fe E = sub(B, A), F = sub(D, C), G = add(D, C), H = add(B, A);
  Pt r; r.x = mul(E, F); r.y = mul(G, H); r.t = mul(E, H); r.z = mul(F, G);
  return r;
}
// dbl-2008-hwcd, a = -1
// (the input's T is not used; with_t = false leaves r.t zero: a doubling that is followed by another needs only X : Y : Z)
static inline Pt pdbl(const Pt& a, bool with_t = true) {
  const Fe A = mul(a.x, a.x), B = mul(a.y, a.y);
  Fe C = mul(a.z, a.z); C = add(C, C);
  Fe zero; memset(&zero, 0, sizeof zero);
  const Fe D = sub(zero, A);
  Fe E = mul(a.x, a.y); E = add(E, E);               // (x + y)^2 - x^2 - y^2
  const Fe G = add(D, B), F = sub(G, C), H = sub(D, B);
  Pt r; r.x = mul(E, F); r.y = mul(G, H); r.z = mul(F, G);
  if (with_t) r.t = mul(E, H); else r.t = zero;
  return r;
}
static inline Pt pdbl_n(Pt a, int k) { for (int i = 0; i < k; i++) a = pdbl(a, i + 1 == k); return a; }

// rows: W x 1120 B = [T | W0 | W1 | W2 | W3]; see host_tail.hpp for the identity behind the fold.  `sets` row buffers are
// summed on the fly (an MSM computed in pieces).
// Edwards (X : Y : Z) -> Montgomery u = (Z + Y)/(Z - Y), v = f u Z / X -> Weierstrass x = sqrt(3) u - 1, y = sqrt(3) v.
// X = 0: the neutral element (Y = Z; the point at infinity, 96 zero bytes) or the point of order two (Y = -Z; (-1, 0)).
static inline void weierstrass_out(const Pt& acc, uint8_t out_xy_le[96]) {
  Fe one_raw; memset(&one_raw, 0, sizeof one_raw); one_raw.l[0] = 1;
  if (is_zero(acc.x)) {
    memset(out_xy_le, 0, 96);
    if (!is_zero(sub(acc.y, acc.z))) { const Fe m1 = mul(sub(Fe{{0, 0, 0, 0, 0, 0}}, ONE_M), one_raw); memcpy(out_xy_le, m1.l, 48); }
    return;
  }
  const Fe zpy = add(acc.z, acc.y), zmy = sub(acc.z, acc.y);
  const Fe di = inv(mul(zmy, acc.x));                           // 1 / ((Z - Y) X)
  const Fe u = mul(mul(zpy, acc.x), di);                        // (Z + Y) / (Z - Y)
  const Fe v = mul(mul(mul(zpy, acc.z), di), F_M);              // f (Z + Y) Z / ((Z - Y) X)
  const Fe x = mul(sub(mul(u, SQRT3_M), ONE_M), one_raw), y = mul(mul(v, SQRT3_M), one_raw);
  memcpy(out_xy_le, x.l, 48); memcpy(out_xy_le + 48, y.l, 48);
}
// the accumulator of Horner's rule in two forms (host_tail.hpp): scalar products one after the other, or the point's coordinates
// in the lanes of AVX-512 registers (host_tail_ifma.hpp, L = 8 limbs of 52 bits, radix 2^416)
struct ScalarAcc {
  Pt acc;
  ScalarAcc() : acc(identity()) {}
  void dbl_n(int k) { acc = pdbl_n(acc, k); }
  void add_point(const Pt& q) { acc = padd(acc, q); }
  void to_affine(uint8_t out_xy_le[96]) const { weierstrass_out(acc, out_xy_le); }
};
#if defined(__x86_64__)
static const te_ifma::field52<8> BLS_F52 = {
  {0x8c00000000001ULL, 0x4430000000850ULL, 0xa094800170b5dULL, 0x138f1ef3622fbULL, 0xb1a22d9f300f5ULL, 0x3b05c06ca1493ULL, 0xa4617c510eac6ULL, 0x1ae3ULL}, 0x8bfffffffffffULL,
  {0x1800000000002ULL, 0x88600000010a1ULL, 0x41290002e16baULL, 0x271e3de6c45f7ULL, 0x63445b3e601eaULL, 0x760b80d942927ULL, 0x48c2f8a21d58cULL, 0x35c7ULL},
  {0x3000000000004ULL, 0x10c0000002142ULL, 0x82520005c2d75ULL, 0x4e3c7bcd88beeULL, 0xc688b67cc03d4ULL, 0xec1701b28524eULL, 0x9185f1443ab18ULL, 0x6b8eULL}};
static inline bool have_ifma() { return te_ifma::cpu_has_ifma(); }
#define TE377_IFMA_M __attribute__((target("avx512f,avx512ifma,avx512dq,avx512vl")))
struct IfmaAcc {
  te_ifma::V<8> acc;
  TE377_IFMA_M IfmaAcc() { Fe zero, one; memset(&zero, 0, sizeof zero); one = zero; one.l[0] = 1; acc = te_ifma::from_words<8, 6>(zero.l, one.l, one.l, zero.l); }
  TE377_IFMA_M void dbl_n(int k) { for (int i = 0; i < k; i++) acc = te_ifma::vdbl<8>(acc, BLS_F52); }
  TE377_IFMA_M void add_point(const Pt& q) {           // operand lanes [Y - X, Y + X, 2 Z, 2 d T], prepared with the scalar arithmetic
    const Fe a = sub(q.y, q.x), b = add(q.y, q.x), c = add(q.z, q.z), d = mul(q.t, K2D);
    acc = te_ifma::vaddp<8>(acc, te_ifma::from_words<8, 6>(a.l, b.l, c.l, d.l), BLS_F52);
  }
  TE377_IFMA_M void to_affine(uint8_t out_xy_le[96]) const {
    uint64_t w[4][6]; te_ifma::to_words<8, 6>(acc, w);
    Pt r; Fe* c[4] = {&r.x, &r.y, &r.z, &r.t};
    for (int k = 0; k < 4; k++) { Fe v; memcpy(v.l, w[k], 48); *c[k] = reduce_once(v); }       // below 2 q in the lanes
    weierstrass_out(r, out_xy_le);
  }
};
#endif
// rows: W x 1120 B = [T | W0 | W1 | W2 | W3]; see host_tail.hpp for the identity behind the fold.  points_of(w, slot, emit) calls
// emit(point) for every point of that window and slot (several row buffers are summed on the fly: an MSM computed in pieces).
template <typename Acc, typename F> static inline void horner_with(F&& points_of, int c, int bucket_bits, int W, uint8_t out_xy_le[96]) {
  int dw[4];
  for (int k = 0; k < 4; k++) dw[k] = (bucket_bits + 3 - k) / 4;
  const int s3 = dw[0] + dw[1] + dw[2];
  Acc acc;
  auto emit = [&](const Pt& q) { acc.add_point(q); };
  for (int w = W - 1; w >= 0; w--) {
    acc.dbl_n(c - s3);
    points_of(w, 4, emit);                             // W3
    acc.dbl_n(dw[2]);
    points_of(w, 3, emit);                             // W2
    acc.dbl_n(dw[1]);
    points_of(w, 2, emit);                             // W1
    acc.dbl_n(dw[0]);
    points_of(w, 1, emit);                             // W0
    points_of(w, 0, emit);                             // T
  }
  acc.to_affine(out_xy_le);
}
template <typename F> static inline void horner_core(F&& points_of, int c, int bucket_bits, int W, uint8_t out_xy_le[96]) {
#if defined(__x86_64__)
  if (have_ifma()) { horner_with<IfmaAcc>(points_of, c, bucket_bits, W, out_xy_le); return; }
#endif
  horner_with<ScalarAcc>(points_of, c, bucket_bits, W, out_xy_le);
}
static inline void horner_to_affine_multi(const uint8_t* const* partials, int sets, int c, int bucket_bits, int W, uint8_t out_xy_le[96]) {
  horner_core([&](int w, int slot, auto& emit) {
    for (int s = 0; s < sets; s++) {
      const uint8_t* row = partials[s] + (size_t)w * TE377_TAIL_ROW_BYTES;
      if (!all_zero_bytes(row, TE377_TAIL_ROW_BYTES)) emit(load_point(row + (size_t)slot * TE377_TAIL_POINT_BYTES));
    }
  }, c, bucket_bits, W, out_xy_le);
}
// two-step form for the multi-device te_msm_run (see host_tail.hpp)
static inline void merge_window_rows(const uint8_t* const* partials, int sets, int w, Pt* merged, uint8_t* present) {
  present[w] = 0;
  for (int s = 0; s < sets; s++) {
    const uint8_t* row = partials[s] + (size_t)w * TE377_TAIL_ROW_BYTES;
    if (all_zero_bytes(row, TE377_TAIL_ROW_BYTES)) continue;
    for (int slot = 0; slot < 5; slot++) {
      const Pt p = load_point(row + (size_t)slot * TE377_TAIL_POINT_BYTES);
      merged[(size_t)w * 5 + slot] = present[w] ? padd(merged[(size_t)w * 5 + slot], p) : p;
    }
    present[w] = 1;
  }
}
static inline void horner_to_affine_points(const Pt* merged, const uint8_t* present, int c, int bucket_bits, int W, uint8_t out_xy_le[96]) {
  horner_core([&](int w, int slot, auto& emit) { if (present[w]) emit(merged[(size_t)w * 5 + slot]); }, c, bucket_bits, W, out_xy_le);
}
static inline void horner_to_affine(const uint8_t* partials, int c, int bucket_bits, int W, uint8_t out_xy_le[96]) {
  horner_to_affine_multi(&partials, 1, c, bucket_bits, W, out_xy_le);
}

static inline bool tail_selftest_run() {
  if ((uint64_t)(MOD[0] * MOD_NEG_INV) != ~0ULL) return false;
  Fe one_raw; memset(&one_raw, 0, sizeof one_raw); one_raw.l[0] = 1;
  const Fe t = mul(ONE_M, one_raw);
  if (!(t.l[0] == 1 && !(t.l[1] | t.l[2] | t.l[3] | t.l[4] | t.l[5]))) return false;
#if defined(__x86_64__)
  if (have_ifma()) {
    // the two accumulators over the same doublings and additions: the formulas are polynomial identities, so any field elements
    // serve as "points" (the operands here are not on the curve); the affine outputs must be the same 96 bytes
    Pt g; g.x = K2D; g.y = F_M; g.z = SQRT3_M; g.t = mul(K2D, F_M);
    ScalarAcc a; IfmaAcc b;
    Pt q = g;
    for (int round = 0; round < 10; round++) {
      a.add_point(q); b.add_point(q);
      a.dbl_n(1 + round % 5); b.dbl_n(1 + round % 5);
      if (round == 3) { const Pt id = identity(); a.add_point(id); b.add_point(id); }
      if (round == 6) { Pt w; Fe top; memcpy(top.l, MOD, 48); top.l[0] -= 1; w.x = w.y = w.z = w.t = top; a.add_point(w); b.add_point(w); }
      q = a.acc;
      uint8_t oa[96], ob[96];
      a.to_affine(oa); b.to_affine(ob);
      if (memcmp(oa, ob, 96) != 0) return false;
    }
  }
#endif
  return true;
}
// once per process (the context-free entry points ask on every call)
static inline bool tail_selftest() { static const bool ok = tail_selftest_run(); return ok; }

}  // namespace te377_host
