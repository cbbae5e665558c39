// host_tail.hpp -- the CPU tail of compute_msm (submission.ts:362-412): fold the per-window partial
// sums with Horner's rule and convert to affine.  The reference does this with @noble/curves bigints
// over up to 4096 points; here the device has already reduced every window to three points, so the
// host executes ~256 doublings, ~50 additions and one inversion (tens of microseconds).
// Host arithmetic: 4 x 64-bit limbs, Montgomery form R = 2^256.  The device writes 9 x 29-bit limbs in Montgomery
// form R' = 2^261, lazily reduced; load_point() reassembles the integer, reduces it and multiplies by
// 2^-261 * 2^512 (one host product) to land in the host's Montgomery domain.
#pragma once
#include <stdint.h>
#include <string.h>

namespace te_host {

typedef unsigned __int128 u128;
struct Fe { uint64_t l[4]; };
struct Pt { Fe x, y, z, t; };   // device row layout: x | y | z | t, 36 bytes each

static const uint64_t MOD[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};
static const uint64_t MOD_NEG_INV = 0x0a117fffffffffffULL;   // -p^-1 mod 2^64 (checked in tail_selftest)
static const Fe ONE_M = {{0x7d1c7ffffffffff3ULL, 0x7257f50f6ffffff2ULL, 0x16d81575512c0feeULL, 0x0d4bda322bbb9a9dULL}};   // R mod p

static inline bool ge_mod(const Fe& a) {
  for (int i = 3; i >= 0; i--) { if (a.l[i] != MOD[i]) return a.l[i] > MOD[i]; }
  return true;
}
static inline void sub_mod_raw(Fe& a) {
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - MOD[i] - br; a.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
static inline Fe canon(Fe a) { while (ge_mod(a)) sub_mod_raw(a); return a; }   // any 256-bit value -> [0, p)

static inline Fe add(const Fe& a, const Fe& b) {
  Fe r; u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  if (ge_mod(r)) sub_mod_raw(r);     // a, b < p < 2^253: no carry out
  return r;
}
static inline Fe sub(const Fe& a, const Fe& b) {
  Fe r; uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + MOD[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
  return r;
}
static inline Fe mul(const Fe& a, const Fe& b) {
  uint64_t t[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; const uint64_t t5 = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * MOD_NEG_INV;
    c = ((u128)m * MOD[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * MOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t5 + (uint64_t)(c >> 64);
  }
  Fe r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || ge_mod(r)) sub_mod_raw(r);
  return r;
}
// a * a / R mod p: the 6 cross products once, doubled, then the same interleaved reduction (about 3/4 of mul's multiplications)
static inline Fe sqr(const Fe& a) {
  uint64_t w[8];
  {
    u128 c = (u128)a.l[0] * a.l[1]; w[1] = (uint64_t)c; c >>= 64;
    c += (u128)a.l[0] * a.l[2]; w[2] = (uint64_t)c; c >>= 64;
    c += (u128)a.l[0] * a.l[3]; w[3] = (uint64_t)c; w[4] = (uint64_t)(c >> 64);
    c = (u128)a.l[1] * a.l[2] + w[3]; w[3] = (uint64_t)c; c >>= 64;
    c += (u128)a.l[1] * a.l[3] + w[4]; w[4] = (uint64_t)c; w[5] = (uint64_t)(c >> 64);
    c = (u128)a.l[2] * a.l[3] + w[5]; w[5] = (uint64_t)c; w[6] = (uint64_t)(c >> 64);
    w[7] = w[6] >> 63; w[6] = (w[6] << 1) | (w[5] >> 63); w[5] = (w[5] << 1) | (w[4] >> 63); w[4] = (w[4] << 1) | (w[3] >> 63);
    w[3] = (w[3] << 1) | (w[2] >> 63); w[2] = (w[2] << 1) | (w[1] >> 63); w[1] <<= 1;
    c = (u128)a.l[0] * a.l[0]; w[0] = (uint64_t)c; c >>= 64;
    c += w[1]; w[1] = (uint64_t)c; c >>= 64;
    c += (u128)a.l[1] * a.l[1] + w[2]; w[2] = (uint64_t)c; c >>= 64;
    c += w[3]; w[3] = (uint64_t)c; c >>= 64;
    c += (u128)a.l[2] * a.l[2] + w[4]; w[4] = (uint64_t)c; c >>= 64;
    c += w[5]; w[5] = (uint64_t)c; c >>= 64;
    c += (u128)a.l[3] * a.l[3] + w[6]; w[6] = (uint64_t)c; c >>= 64;
    w[7] += (uint64_t)c;
  }
  uint64_t top = 0;                                  // carry out of word i + 4 of the previous round
  for (int i = 0; i < 4; i++) {
    const uint64_t m = w[i] * MOD_NEG_INV;
    u128 c = ((u128)m * MOD[0] + w[i]) >> 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * MOD[j] + w[i + j]; w[i + j] = (uint64_t)c; c >>= 64; }
    c += (u128)w[i + 4] + top; w[i + 4] = (uint64_t)c; top = (uint64_t)(c >> 64);
  }
  Fe r = {{w[4], w[5], w[6], w[7]}};
  if (top || ge_mod(r)) sub_mod_raw(r);
  return r;
}
static inline Fe inv(const Fe& a) {            // a^(p-2)
  uint64_t e[4] = {MOD[0] - 2, MOD[1], MOD[2], MOD[3]};
  Fe acc = ONE_M, base = a;
  for (int i = 0; i < 253; i++) {
    if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
    base = sqr(base);
  }
  return acc;
}
static inline bool is_zero(const Fe& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }

static inline Pt identity() { Pt r; memset(&r, 0, sizeof r); r.y = ONE_M; r.z = ONE_M; return r; }
static inline bool all_zero_bytes(const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i]) return false; return true; }

#define TE_TAIL_POINT_BYTES 144
#define TE_TAIL_ROW_BYTES 720
// one coordinate: 9 u32 words holding 29-bit limbs (possibly unnormalised), value < 2^262.
// The device's Montgomery radix is 2^261, the host's 2^256: read as a host residue, the integer stands for 2^5 times the
// coordinate.  A projective point is not changed by a common factor of (X : Y : Z : T) -- T Z = X Y still holds -- so the four
// coordinates of a device point are taken over as they are (reduced below p), without a conversion product each.
static inline Fe load_coord(const uint8_t* src) {
  uint32_t l[9]; memcpy(l, src, 36);
  uint64_t w[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 9; i++) {                      // w += l[i] << (29 i), with carries (limbs may exceed 29 bits)
    const int bit = 29 * i, j = bit >> 6, s = bit & 63;
    const u128 add = (u128)l[i] << s;
    u128 c = (u128)w[j] + (uint64_t)add; w[j] = (uint64_t)c; c >>= 64;
    c += (u128)w[j + 1] + (uint64_t)(add >> 64); w[j + 1] = (uint64_t)c; c >>= 64;
    for (int q = j + 2; q < 6 && c; q++) { c += w[q]; w[q] = (uint64_t)c; c >>= 64; }
  }
  // reduce below p: the device keeps values under ~2p, so this loop runs a couple of times at most
  Fe r = {{w[0], w[1], w[2], w[3]}};
  uint64_t top = w[4];
  for (int guard = 0; guard < 4096 && (top || ge_mod(r)); guard++) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)r.l[i] - MOD[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    top -= br;
  }
  return r;
}
static inline Pt load_point(const uint8_t* src) {      // 144 B device extended point
  Pt r; r.x = load_coord(src); r.y = load_coord(src + 36); r.z = load_coord(src + 72); r.t = load_coord(src + 108);
  return r;
}
// unified addition, a = -1, k = 2d (add-2008-hwcd-3)
static inline Pt padd(const Pt& a, const Pt& b, const Fe& k2d) {
  const Fe A = mul(sub(a.y, a.x), sub(b.y, b.x));
  const Fe B = mul(add(a.y, a.x), add(b.y, b.x));
  const Fe C = mul(mul(a.t, b.t), k2d);
  const Fe zz = mul(a.z, b.z);
  const Fe D = add(zz, zz);
  const Fe E = sub(B, A), F = sub(D, C), G = add(D, C), H = add(B, A);
  Pt r; r.x = mul(E, F); r.y = mul(G, H); r.t = mul(E, H); r.z = mul(F, G);
  return r;
}
// dbl-2008-hwcd, a = -1 (the input's T is not used).  with_t = false leaves r.t unset: a doubling that is followed by
// another doubling needs only (X : Y : Z) -- 7 products instead of 8.
static inline Pt pdbl(const Pt& a, bool with_t = true) {
  const Fe A = sqr(a.x), B = sqr(a.y);
  Fe C = sqr(a.z); C = add(C, C);
  const Fe zero = {{0, 0, 0, 0}};
  const Fe D = sub(zero, A);
  const Fe xy = add(a.x, a.y);
  const Fe E = sub(sub(sqr(xy), A), B);
  const Fe G = add(D, B), F = sub(G, C), H = sub(D, B);
  Pt r; r.x = mul(E, F); r.y = mul(G, H); r.z = mul(F, G);
  if (with_t) r.t = mul(E, H); else r.t = zero;
  return r;
}
// k doublings; only the last one produces T (the next operation is an addition)
static inline Pt pdbl_n(Pt a, int k) {
  for (int i = 0; i < k; i++) a = pdbl(a, i + 1 == k);
  return a;
}

// partials: W rows of 720 B = [T | W0 | W1 | W2 | W3]: T = sum of the window's buckets, Wk = sum_v v * M_k[v] for digit k of
// the bucket index (bucket_bits = c - 1 for signed digits, c for unsigned; digit widths w_k = (bucket_bits + 3 - k) / 4,
// e.g. 4,4,4,3 for 15 bits).  Window value
//     V = T + W0 + 2^w0 W1 + 2^(w0+w1) W2 + 2^(w0+w1+w2) W3,      result = sum_w 2^(c*w) V_w,
// evaluated top-down; the c doublings per window are split around the digit terms, so no doubling is added.
// `sets` row buffers are summed on the fly: the rows are linear in the bucket contents, so an MSM computed in pieces
// (te_msm_run uploads and processes a large host buffer in chunks) folds as the sum of the pieces' rows.
static inline void horner_to_affine_multi(const uint8_t* const* partials, int sets, int c, int bucket_bits, int W, uint8_t out_xy_le[64]) {
  const Fe d2 = {{2 * 3021, 0, 0, 0}};
  const Fe R2 = {{0x25d577bab861857bULL, 0xcc2c27b58860591fULL, 0xa7cc008fe5dc8593ULL, 0x011fdae7eff1c939ULL}};
  const Fe k2d = mul(d2, R2);
  int dw[4];
  for (int k = 0; k < 4; k++) dw[k] = (bucket_bits + 3 - k) / 4;
  const int s3 = dw[0] + dw[1] + dw[2];
  Pt acc = identity();
  auto add_slot = [&](int w, int slot) {
    for (int s = 0; s < sets; s++) {
      const uint8_t* row = partials[s] + (size_t)w * TE_TAIL_ROW_BYTES;
      if (!all_zero_bytes(row, TE_TAIL_ROW_BYTES)) acc = padd(acc, load_point(row + (size_t)slot * TE_TAIL_POINT_BYTES), k2d);
    }
  };
  for (int w = W - 1; w >= 0; w--) {
    acc = pdbl_n(acc, c - s3);
    add_slot(w, 4);                                    // W3
    acc = pdbl_n(acc, dw[2]);
    add_slot(w, 3);                                    // W2
    acc = pdbl_n(acc, dw[1]);
    add_slot(w, 2);                                    // W1
    acc = pdbl_n(acc, dw[0]);
    add_slot(w, 1);                                    // W0
    add_slot(w, 0);                                    // T
  }
  const Fe zi = inv(acc.z);
  const Fe one_raw = {{1, 0, 0, 0}};
  const Fe x = mul(mul(acc.x, zi), one_raw), y = mul(mul(acc.y, zi), one_raw);
  memcpy(out_xy_le, x.l, 32); memcpy(out_xy_le + 32, y.l, 32);
}
static inline void horner_to_affine(const uint8_t* partials, int c, int bucket_bits, int W, uint8_t out_xy_le[64]) {
  horner_to_affine_multi(&partials, 1, c, bucket_bits, W, out_xy_le);
}

static inline bool tail_selftest() {
  if ((uint64_t)(MOD[0] * MOD_NEG_INV) != ~0ULL) return false;     // p * (-p^-1) = -1 mod 2^64
  const Fe one_raw = {{1, 0, 0, 0}};
  const Fe t = mul(ONE_M, one_raw);                                // R * 1 / R = 1
  if (!(t.l[0] == 1 && !t.l[1] && !t.l[2] && !t.l[3])) return false;
  Fe v = {{0x243f6a8885a308d3ULL, 0x13198a2e03707344ULL, 0xa4093822299f31d0ULL, 0x082efa98ec4e6c89ULL}};     // below p
  for (int i = 0; i < 8; i++) {                                    // the dedicated squaring against the general product
    const Fe a = sqr(v), b = mul(v, v);
    if (memcmp(a.l, b.l, 32) != 0) return false;
    v = add(a, ONE_M);
  }
  return true;
}

}  // namespace te_host
