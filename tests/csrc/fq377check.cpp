// Host build of the product's BLS12-377 device arithmetic (csrc/fq377.hpp + the generic curve code of csrc/curve.hpp at
// 14 limbs) for the CPU test-suite: the same code the GPU runs, with every column sum checked against 2^64
// (g_fq377_overflow).  Not a CPU fallback.
#include <string.h>
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
}
