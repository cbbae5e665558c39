// host_tail377.hpp -- the CPU tail for BLS12-377 G1: the same Horner fold over per-window rows [T | W0 | W1 | W2 | W3] as
// host_tail.hpp (which follows submission.ts:362-412).  The device works in the twisted-Edwards form of the curve
// (curve.hpp), so the fold uses the same extended-coordinate formulas as the other curve; the result is mapped back to
// the short-Weierstrass affine point the caller expects.  Host arithmetic: 6 x 64-bit limbs, Montgomery form R = 2^384;
// the device writes 14 x 29-bit limbs in Montgomery form R' = 2^406, lazily reduced.
#pragma once
#include <stdint.h>
#include <string.h>
#include "fq377.hpp"     // TE377_HOST_* constants (fq377_constants.inc)

namespace te377_host {

typedef unsigned __int128 u128;
struct Fe { uint64_t l[6]; };
struct Pt { Fe x, y, z, t; };    // extended twisted Edwards (X : Y : Z : T) of the curve's Edwards form; identity (0 : 1 : 1 : 0)

static const uint64_t MOD[6] = {0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL,
                                0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x01ae3a4617c510eaULL};      // README.md:65-67
static const uint64_t MOD_NEG_INV = 0x8508bfffffffffffULL;   // -q^-1 mod 2^64 (checked in tail_selftest)
static const Fe ONE_M = {{0x02cdffffffffff68ULL, 0x51409f837fffffb1ULL, 0x9f7db3a98a7d3ff2ULL, 0x7b4e97b76e7c6305ULL, 0x4cf495bf803c84e8ULL, 0x008d6661e2fdf49aULL}};   // R mod q
// device -> host domain: an integer v = lo + hi * 2^384 read from the device limbs stands for v / 2^406; its host
// Montgomery form is v * 2^-406 * 2^384 = mul(lo, 2^362) + mul(hi, 2^746)
static const Fe CONV_LO = {{0, 0, 0, 0, 0, 0x0000040000000000ULL}};
static const Fe CONV_HI = {{0x9425202a73a1b251ULL, 0xe35334ac0ffcb140ULL, 0x3d2ee6b284c44cfbULL, 0xbe1ba36ddcc0f814ULL, 0x2a6979709f82dbecULL, 0x01a8d750983edd8aULL}};

static inline bool ge_mod(const Fe& a) { for (int i = 5; i >= 0; i--) { if (a.l[i] != MOD[i]) return a.l[i] > MOD[i]; } return true; }
static inline void sub_mod_raw(Fe& a) { uint64_t br = 0; for (int i = 0; i < 6; i++) { u128 d = (u128)a.l[i] - MOD[i] - br; a.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
static inline Fe add(const Fe& a, const Fe& b) {
  Fe r; u128 c = 0;
  for (int i = 0; i < 6; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  if (ge_mod(r)) sub_mod_raw(r);
  return r;
}
static inline Fe sub(const Fe& a, const Fe& b) {
  Fe r; uint64_t br = 0;
  for (int i = 0; i < 6; i++) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 6; i++) { c += (u128)r.l[i] + MOD[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
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
  Fe lo, hi; memcpy(lo.l, w, 48); memset(&hi, 0, sizeof hi); hi.l[0] = w[6]; hi.l[1] = w[7];
  return add(mul(lo, CONV_LO), mul(hi, CONV_HI));
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
static inline Pt pdbl(const Pt& a) {
  const Fe A = mul(a.x, a.x), B = mul(a.y, a.y);
  Fe C = mul(a.z, a.z); C = add(C, C);
  Fe zero; memset(&zero, 0, sizeof zero);
  const Fe D = sub(zero, A);
  const Fe xy = add(a.x, a.y);
  const Fe E = sub(sub(mul(xy, xy), A), B);
  const Fe G = add(D, B), F = sub(G, C), H = sub(D, B);
  Pt r; r.x = mul(E, F); r.y = mul(G, H); r.t = mul(E, H); r.z = mul(F, G);
  return r;
}

// rows: W x 1120 B = [T | W0 | W1 | W2 | W3]; see host_tail.hpp for the identity behind the fold.  `sets` row buffers are
// summed on the fly (an MSM computed in pieces).
template <typename F> static inline void horner_core(F&& add_slot, int c, int bucket_bits, int W, uint8_t out_xy_le[96]) {
  int dw[4];
  for (int k = 0; k < 4; k++) dw[k] = (bucket_bits + 3 - k) / 4;
  const int s3 = dw[0] + dw[1] + dw[2];
  Pt acc = identity();
  for (int w = W - 1; w >= 0; w--) {
    for (int k = 0; k < c - s3; k++) acc = pdbl(acc);
    add_slot(w, 4, acc);                               // W3
    for (int k = 0; k < dw[2]; k++) acc = pdbl(acc);
    add_slot(w, 3, acc);                               // W2
    for (int k = 0; k < dw[1]; k++) acc = pdbl(acc);
    add_slot(w, 2, acc);                               // W1
    for (int k = 0; k < dw[0]; k++) acc = pdbl(acc);
    add_slot(w, 1, acc);                               // W0
    add_slot(w, 0, acc);                               // T
  }
  // Edwards (X : Y : Z) -> Montgomery u = (Z + Y)/(Z - Y), v = f u Z / X -> Weierstrass x = sqrt(3) u - 1, y = sqrt(3) v.
  // X = 0: the neutral element (Y = Z; the point at infinity, 96 zero bytes) or the point of order two (Y = -Z; (-1, 0)).
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
static inline void horner_to_affine_multi(const uint8_t* const* partials, int sets, int c, int bucket_bits, int W, uint8_t out_xy_le[96]) {
  horner_core([&](int w, int slot, Pt& acc) {
    for (int s = 0; s < sets; s++) {
      const uint8_t* row = partials[s] + (size_t)w * TE377_TAIL_ROW_BYTES;
      if (!all_zero_bytes(row, TE377_TAIL_ROW_BYTES)) acc = padd(acc, load_point(row + (size_t)slot * TE377_TAIL_POINT_BYTES));
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
  horner_core([&](int w, int slot, Pt& acc) { if (present[w]) acc = padd(acc, merged[(size_t)w * 5 + slot]); }, c, bucket_bits, W, out_xy_le);
}
static inline void horner_to_affine(const uint8_t* partials, int c, int bucket_bits, int W, uint8_t out_xy_le[96]) {
  horner_to_affine_multi(&partials, 1, c, bucket_bits, W, out_xy_le);
}

static inline bool tail_selftest_run() {
  if ((uint64_t)(MOD[0] * MOD_NEG_INV) != ~0ULL) return false;
  Fe one_raw; memset(&one_raw, 0, sizeof one_raw); one_raw.l[0] = 1;
  const Fe t = mul(ONE_M, one_raw);
  return t.l[0] == 1 && !(t.l[1] | t.l[2] | t.l[3] | t.l[4] | t.l[5]);
}
// once per process (the context-free entry points ask on every call)
static inline bool tail_selftest() { static const bool ok = tail_selftest_run(); return ok; }

}  // namespace te377_host
