// fpcheck.cpp -- TEST SHIM: compiles the product's device arithmetic headers (csrc/fp.hpp, csrc/curve.hpp)
// for the host so the exact limb code that runs on gfx950 can be checked bit-for-bit on the CPU box,
// and emulates the device stages (digits -> buckets -> row/column marginals -> weighted sums) with that
// same arithmetic to produce the 384-byte partial rows the host tail and the sharding code consume.
// Not part of the product; not a fallback.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../webgpu-msm-twisted-edwards_amd/csrc/curve.hpp"

using namespace te;

// value of a limb vector as a 320-bit integer, compared with k*p
static void to_words(const fp& a, uint64_t w[6]) {
  for (int i = 0; i < 6; i++) w[i] = 0;
  for (int i = 0; i < NL; i++) {
    const int bit = 29 * i, j = bit >> 6, s = bit & 63;
    unsigned __int128 add = (unsigned __int128)a.v[i] << s, c = (unsigned __int128)w[j] + (uint64_t)add;
    w[j] = (uint64_t)c; c >>= 64; c += (unsigned __int128)w[j + 1] + (uint64_t)(add >> 64); w[j + 1] = (uint64_t)c; c >>= 64;
    for (int q = j + 2; q < 6 && c; q++) { c += w[q]; w[q] = (uint64_t)c; c >>= 64; }
  }
}
static bool lt_kp(const fp& a, int k) {   // a < k*p ?
  static const uint64_t P64[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};
  uint64_t kp[6] = {0, 0, 0, 0, 0, 0}; unsigned __int128 c = 0;
  for (int i = 0; i < 4; i++) { c += (unsigned __int128)P64[i] * (unsigned)k; kp[i] = (uint64_t)c; c >>= 64; }
  kp[4] = (uint64_t)c;
  uint64_t w[6]; to_words(a, w);
  for (int i = 5; i >= 0; i--) { if (w[i] != kp[i]) return w[i] < kp[i]; }
  return false;
}
static bool class_n(const fp& a) { for (int i = 0; i < NL - 1; i++) if (a.v[i] >= (1u << 29)) return false; return a.v[NL - 1] < (1u << 24); }
static int g_bound_violations = 0;
static void chk2(const ete& e) {
  if (!(lt_kp(e.x, 2) && lt_kp(e.y, 2) && lt_kp(e.z, 2) && lt_kp(e.t, 2))) g_bound_violations++;
  if (!(class_n(e.x) && class_n(e.y) && class_n(e.z) && class_n(e.t))) g_bound_violations++;
}

extern "C" {

int fpc_bound_violations() { return g_bound_violations; }

void fpc_mont_mul(const uint32_t a[9], const uint32_t b[9], uint32_t out[9]) {
  fp x, y; memcpy(x.v, a, 36); memcpy(y.v, b, 36); fp r = mont_mul(x, y); memcpy(out, r.v, 36);
}
void fpc_mul_k2d(const uint32_t a[9], uint32_t out[9]) { fp x; memcpy(x.v, a, 36); fp r = fp_mul_k2d(x); memcpy(out, r.v, 36); }
uint32_t fpc_k2d_q() { return K2D_Q; }
void fpc_norm(const uint32_t a[9], uint32_t out[9]) { fp x; memcpy(x.v, a, 36); fp r = fp_norm(x); memcpy(out, r.v, 36); }
void fpc_sub2(const uint32_t a[9], const uint32_t b[9], uint32_t out[9]) { fp x, y; memcpy(x.v, a, 36); memcpy(y.v, b, 36); fp r = fp_sub<2>(x, y); memcpy(out, r.v, 36); }
void fpc_from_words32(const uint32_t w[8], uint32_t out[9]) { uint32_t t[8]; memcpy(t, w, 32); fp r = fp_from_words32(t); memcpy(out, r.v, 36); }
void fpc_constants(uint32_t out[9 * 9]) {
  fp c[9] = {fp_R1(), fp_R2(), fp_D_MONT(), fp_K2D_MONT(), fp_ONE_RAW(), fp_P(), fp_kp_offset<2>(), fp_kp_offset<4>(), fp_kp_offset<8>()};
  memcpy(out, c, sizeof c);
}
void fpc_constants2(uint32_t out[3 * 9]) {
  fp c[3] = {fp_R2_HALF(), fp_NEG_D_R3(), fp_kp_offset<16>()};
  memcpy(out, c, sizeof c);
}
// body of k_prep_points: record in a 128-byte slot
void fpc_prep_point(const uint8_t xy_le[64], uint8_t rec[128]) {
  uint32_t xw[8], yw[8]; memcpy(xw, xy_le, 32); memcpy(yw, xy_le + 32, 32);
  const pnt r = pnt_from_affine_raw(fp_from_words32(xw), fp_from_words32(yw));
  memset(rec, 0, 128); memcpy(rec, &r, 108);
  if (!(class_n(r.hm) && class_n(r.hp) && class_n(r.dt) && lt_kp(r.hm, 2) && lt_kp(r.hp, 2) && lt_kp(r.dt, 2))) g_bound_violations++;
}
void fpc_identity(uint8_t out[144]) { ete e = ete_identity(); memcpy(out, &e, 144); }
void fpc_madd(const uint8_t acc[144], const uint8_t rec[128], int neg, uint8_t out[144]) {
  ete a; pnt b; memcpy(&a, acc, 144); memcpy(&b, rec, 108);
  ete r = ete_madd(a, pnt_cneg(b, neg != 0)); chk2(r); memcpy(out, &r, 144);
}
void fpc_add(const uint8_t a_[144], const uint8_t b_[144], uint8_t out[144]) {
  ete a, b; memcpy(&a, a_, 144); memcpy(&b, b_, 144); ete r = ete_add(a, b); chk2(r); memcpy(out, &r, 144);
}

// Emulation of the device stages for the windows w = first + k*step.  partials: W x 720 B (rows of other
// windows untouched).  Returns 0, or -3 on a final carry.
int fpc_partial_rows(const uint8_t* points, const uint8_t* scalars, uint64_t n, int c, int first, int step, uint8_t* partials) {
  const int W = (256 + c - 1) / c;
  const uint32_t B = 1u << (c - 1);
  std::vector<pnt> recs(n);
  for (uint64_t i = 0; i < n; i++) { uint8_t slot[128]; fpc_prep_point(points + 64 * i, slot); memcpy(&recs[i], slot, 108); }
  // half = sum_w 2^(c*w + c-1)
  uint32_t half[10] = {0};
  for (int w = 0; w < W; w++) { int bit = w * c + c - 1; if (bit < 320) half[bit >> 5] |= 1u << (bit & 31); }
  std::vector<std::vector<uint32_t>> dig(W, std::vector<uint32_t>(n));
  for (uint64_t i = 0; i < n; i++) {
    uint32_t s[11] = {0}; memcpy(s, scalars + 32 * i, 32);
    uint64_t cy = 0;
    for (int j = 0; j < 10; j++) { cy += (uint64_t)s[j] + half[j]; s[j] = (uint32_t)cy; cy >>= 32; }
    for (int w = 0; w <= W; w++) {
      const int bit = w * c; if (bit >= 320) break;
      const int word = bit >> 5, off = bit & 31;
      uint64_t two = (uint64_t)s[word] | ((uint64_t)s[word + 1] << 32);
      uint32_t v = (uint32_t)(two >> off) & ((1u << c) - 1u);
      if (w == W) { if (v) return -3; } else dig[w][i] = v;
    }
  }
  for (int w = first; w < W; w += step) {
    std::vector<ete> bk(B, ete_identity());
    for (uint64_t i = 0; i < n; i++) {
      const int d = (int)dig[w][i] - (int)B;
      if (d == 0) continue;
      const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
      bk[b] = ete_madd(bk[b], pnt_cneg(recs[i], d < 0)); chk2(bk[b]);
    }
    // digit marginals M_k[v] and their weighted sums, straight from the definition
    uint32_t dw[4], sh[4]; uint32_t acc_sh = 0;
    for (int k = 0; k < 4; k++) { dw[k] = (uint32_t)(c - 1 + 3 - k) / 4u; sh[k] = acc_sh; acc_sh += dw[k]; }
    ete T = ete_identity(), Wk[4];
    for (uint32_t j = 0; j < B; j++) { T = ete_add(T, bk[j]); chk2(T); }
    for (int k = 0; k < 4; k++) {
      const uint32_t N = 1u << dw[k];
      std::vector<ete> M(N, ete_identity());
      for (uint32_t j = 0; j < B; j++) { ete& m = M[(j >> sh[k]) & (N - 1u)]; m = ete_add(m, bk[j]); chk2(m); }
      ete run = ete_identity(); Wk[k] = ete_identity();
      for (uint32_t v = N; v-- > 1;) { run = ete_add(run, M[v]); Wk[k] = ete_add(Wk[k], run); }
      chk2(Wk[k]);
    }
    uint8_t* row = partials + (size_t)w * 720;
    memcpy(row, &T, 144);
    for (int k = 0; k < 4; k++) memcpy(row + 144 * (1 + k), &Wk[k], 144);
  }
  return 0;
}

}  // extern "C"
