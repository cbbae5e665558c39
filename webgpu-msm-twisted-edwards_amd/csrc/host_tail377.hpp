// host_tail377.hpp -- the CPU tail for BLS12-377 G1: the same Horner fold over per-window rows [T | W0 | W1 | W2 | W3] as
// host_tail.hpp (which follows submission.ts:362-412), with this curve's group law.  Host arithmetic: 6 x 64-bit limbs,
// Montgomery form R = 2^384; the device writes 14 x 29-bit limbs in Montgomery form R' = 2^406, lazily reduced.
#pragma once
#include <stdint.h>
#include <string.h>

namespace te377_host {

typedef unsigned __int128 u128;
struct Fe { uint64_t l[6]; };
struct Pt { Fe x, y, z; };       // projective (X : Y : Z); identity (0 : 1 : 0)

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
static inline Pt identity() { Pt r; memset(&r, 0, sizeof r); r.y = ONE_M; return r; }
static inline bool all_zero_bytes(const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i]) return false; return true; }

#define TE377_TAIL_COORD_BYTES 56
#define TE377_TAIL_POINT_BYTES 168
#define TE377_TAIL_ROW_BYTES 840
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
static inline Pt load_point(const uint8_t* src) { Pt r; r.x = load_coord(src); r.y = load_coord(src + 56); r.z = load_coord(src + 112); return r; }

// complete projective addition (Renes-Costello-Batina 2016, Algorithm 7, a = 0, b3 = 3); also doubles
static inline Fe mul3(const Fe& a) { return add(add(a, a), a); }
static inline Pt padd(const Pt& p, const Pt& q) {
  const Fe t0 = mul(p.x, q.x), t1 = mul(p.y, q.y), t2 = mul(p.z, q.z);
  const Fe t3 = sub(sub(mul(add(p.x, p.y), add(q.x, q.y)), t0), t1);
  const Fe t4 = sub(sub(mul(add(p.y, p.z), add(q.y, q.z)), t1), t2);
  const Fe y3 = mul3(sub(sub(mul(add(p.x, p.z), add(q.x, q.z)), t0), t2));
  const Fe t0x3 = mul3(t0), t2x3 = mul3(t2);
  const Fe z3 = add(t1, t2x3), t1m = sub(t1, t2x3);
  Pt r;
  r.x = sub(mul(t3, t1m), mul(t4, y3));
  r.y = add(mul(t1m, z3), mul(y3, t0x3));
  r.z = add(mul(z3, t4), mul(t0x3, t3));
  return r;
}

// rows: W x 840 B = [T | W0 | W1 | W2 | W3]; see host_tail.hpp for the identity behind the fold
static inline void horner_to_affine(const uint8_t* partials, int c, int bucket_bits, int W, uint8_t out_xy_le[96]) {
  int dw[4];
  for (int k = 0; k < 4; k++) dw[k] = (bucket_bits + 3 - k) / 4;
  const int s3 = dw[0] + dw[1] + dw[2];
  Pt acc = identity();
  for (int w = W - 1; w >= 0; w--) {
    const uint8_t* row = partials + (size_t)w * TE377_TAIL_ROW_BYTES;
    const bool present = !all_zero_bytes(row, TE377_TAIL_ROW_BYTES);
    for (int k = 0; k < c - s3; k++) acc = padd(acc, acc);
    if (present) acc = padd(acc, load_point(row + 4 * TE377_TAIL_POINT_BYTES));      // W3
    for (int k = 0; k < dw[2]; k++) acc = padd(acc, acc);
    if (present) acc = padd(acc, load_point(row + 3 * TE377_TAIL_POINT_BYTES));      // W2
    for (int k = 0; k < dw[1]; k++) acc = padd(acc, acc);
    if (present) acc = padd(acc, load_point(row + 2 * TE377_TAIL_POINT_BYTES));      // W1
    for (int k = 0; k < dw[0]; k++) acc = padd(acc, acc);
    if (present) { acc = padd(acc, load_point(row + TE377_TAIL_POINT_BYTES)); acc = padd(acc, load_point(row)); }   // W0, T
  }
  if (is_zero(acc.z)) { memset(out_xy_le, 0, 96); return; }            // point at infinity
  const Fe zi = inv(acc.z);
  Fe one_raw; memset(&one_raw, 0, sizeof one_raw); one_raw.l[0] = 1;
  const Fe x = mul(mul(acc.x, zi), one_raw), y = mul(mul(acc.y, zi), one_raw);
  memcpy(out_xy_le, x.l, 48); memcpy(out_xy_le + 48, y.l, 48);
}

static inline bool tail_selftest() {
  if ((uint64_t)(MOD[0] * MOD_NEG_INV) != ~0ULL) return false;
  Fe one_raw; memset(&one_raw, 0, sizeof one_raw); one_raw.l[0] = 1;
  const Fe t = mul(ONE_M, one_raw);
  return t.l[0] == 1 && !(t.l[1] | t.l[2] | t.l[3] | t.l[4] | t.l[5]);
}

}  // namespace te377_host
