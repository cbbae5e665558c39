// Host build of the product's BLS12-377 device arithmetic (csrc/fq377.hpp + the generic curve code of csrc/curve.hpp at
// 14 limbs) for the CPU test-suite: the same code the GPU runs, with every column sum checked against 2^64
// (g_fq377_overflow).  Not a CPU fallback.
#include <string.h>
#include <vector>
#define TE377_CHECK_COLUMNS 1
#include "../../webgpu-msm-twisted-edwards_amd/csrc/curve.hpp"

int g_fq377_overflow = 0;
extern "C" {
using namespace te;
using te377::fq;

int f377_overflow_and_reset() { const int v = g_fq377_overflow; g_fq377_overflow = 0; return v; }
void f377_mont_mul(const uint32_t a[14], const uint32_t b[14], uint32_t out[14]) {
  fq x, y; memcpy(x.v, a, 56); memcpy(y.v, b, 56); const fq r = te377::mont_mul(x, y); memcpy(out, r.v, 56);
}
void f377_norm(const uint32_t a[14], uint32_t out[14]) { fq x; memcpy(x.v, a, 56); const fq r = te377::fq_norm(x); memcpy(out, r.v, 56); }
void f377_from_words32(const uint32_t w[12], uint32_t out[14]) { uint32_t t[12]; memcpy(t, w, 48); const fq r = te377::fq_from_words32(t); memcpy(out, r.v, 56); }
// [R, R^2, q, 2q, 4q, 8q, 16q (offset forms), 2d R, -2d R, s R^2, f R^2, f R, (s+1) R, (s-1) R]
void f377_constants(uint32_t out[14 * 14]) {
  using namespace te377;
  const fq c[14] = {fq_R1(), fq_R2(), fq_Q(), fq_kq_offset<2>(), fq_kq_offset<4>(), fq_kq_offset<8>(), fq_kq_offset<16>(),
                    fq_K2D_MONT(), fq_NEG_2D_MONT(), fq_S_R2(), fq_F_R2(), fq_F_MONT(), fq_SP1_MONT(), fq_SM1_MONT()};
  memcpy(out, c, sizeof c);
}
// body of k_prep_points377: short-Weierstrass (x, y) -> projective Edwards record, 224 bytes (hm | hp | dt | z)
void f377_prep_point(const uint8_t xy_le[96], uint8_t rec[224]) {
  uint32_t xw[12], yw[12]; memcpy(xw, xy_le, 48); memcpy(yw, xy_le + 48, 48);
  const pnt_t<14> r = pnt_from_sw377(te377::fq_from_words32(xw), te377::fq_from_words32(yw));
  memcpy(rec, &r, 224);
}
void f377_identity(uint8_t out[224]) { const ete_t<14> e = ete_identity_t<14>(); memcpy(out, &e, 224); }
void f377_madd(const uint8_t acc[224], const uint8_t rec[224], int neg, uint8_t out[224]) {
  ete_t<14> a; pnt_t<14> b; memcpy(&a, acc, 224); memcpy(&b, rec, 224);
  const ete_t<14> r = ete_madd(a, pnt_cneg(b, neg != 0)); memcpy(out, &r, 224);
}
void f377_add(const uint8_t a_[224], const uint8_t b_[224], uint8_t out[224]) {
  ete_t<14> a, b; memcpy(&a, a_, 224); memcpy(&b, b_, 224); const ete_t<14> r = ete_add<14>(a, b); memcpy(out, &r, 224);
}

// Emulation of the device stages with the device arithmetic, for the windows w = first + k*step: digits -> buckets ->
// digit marginals -> weighted sums -> rows [T | W0 | W1 | W2 | W3] of 1120 bytes (rows of other windows untouched): what the
// host tail te377_host::horner_to_affine consumes.  Scalars are 48-byte records; returns -3 on a final carry.
int f377_partial_rows(const uint8_t* points, const uint8_t* scalars, uint64_t n, int c, int first, int step, uint8_t* partials) {
  const int W = (256 + c - 1) / c;
  const uint32_t B = 1u << (c - 1);
  std::vector<pnt_t<14>> recs(n);
  for (uint64_t i = 0; i < n; i++) f377_prep_point(points + 96 * i, reinterpret_cast<uint8_t*>(&recs[i]));
  uint32_t half[10] = {0};
  for (int w = 0; w < W; w++) { int bit = w * c + c - 1; if (bit < 320) half[bit >> 5] |= 1u << (bit & 31); }
  std::vector<std::vector<uint32_t>> dig(W, std::vector<uint32_t>(n));
  for (uint64_t i = 0; i < n; i++) {
    uint32_t s[11] = {0}; memcpy(s, scalars + 48 * i, 32);
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
    std::vector<ete_t<14>> bk(B, ete_identity_t<14>());
    for (uint64_t i = 0; i < n; i++) {
      const int d = (int)dig[w][i] - (int)B;
      if (d == 0) continue;
      const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
      bk[b] = ete_madd(bk[b], pnt_cneg(recs[i], d < 0));
    }
    uint32_t dw[4], sh[4]; uint32_t acc_sh = 0;
    for (int k = 0; k < 4; k++) { dw[k] = (uint32_t)(c - 1 + 3 - k) / 4u; sh[k] = acc_sh; acc_sh += dw[k]; }
    ete_t<14> T = ete_identity_t<14>(), Wk[4];
    for (uint32_t j = 0; j < B; j++) T = ete_add<14>(T, bk[j]);
    for (int k = 0; k < 4; k++) {
      const uint32_t N = 1u << dw[k];
      std::vector<ete_t<14>> M(N, ete_identity_t<14>());
      for (uint32_t j = 0; j < B; j++) { ete_t<14>& m = M[(j >> sh[k]) & (N - 1u)]; m = ete_add<14>(m, bk[j]); }
      ete_t<14> run = ete_identity_t<14>(); Wk[k] = ete_identity_t<14>();
      for (uint32_t v = N; v-- > 1;) { run = ete_add<14>(run, M[v]); Wk[k] = ete_add<14>(Wk[k], run); }
    }
    uint8_t* row = partials + (size_t)w * 1120;
    memcpy(row, &T, 224);
    for (int k = 0; k < 4; k++) memcpy(row + 224 * (1 + k), &Wk[k], 224);
  }
  return 0;
}
}
