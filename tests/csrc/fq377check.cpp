// Host build of the product's BLS12-377 device arithmetic (csrc/fq377.hpp, csrc/curve377.hpp) for the CPU test-suite:
// the same code the GPU runs, with every column sum checked against 2^64 (g_fq377_overflow).  Not a CPU fallback.
#include <string.h>
#define TE377_CHECK_COLUMNS 1
#include "../../webgpu-msm-twisted-edwards_amd/csrc/curve377.hpp"

int g_fq377_overflow = 0;
extern "C" {
using namespace te377;

int f377_overflow_and_reset() { const int v = g_fq377_overflow; g_fq377_overflow = 0; return v; }
void f377_mont_mul(const uint32_t a[14], const uint32_t b[14], uint32_t out[14]) {
  fq x, y; memcpy(x.v, a, 56); memcpy(y.v, b, 56); const fq r = mont_mul(x, y); memcpy(out, r.v, 56);
}
void f377_norm(const uint32_t a[14], uint32_t out[14]) { fq x; memcpy(x.v, a, 56); const fq r = fq_norm(x); memcpy(out, r.v, 56); }
void f377_from_words32(const uint32_t w[12], uint32_t out[14]) { uint32_t t[12]; memcpy(t, w, 48); const fq r = fq_from_words32(t); memcpy(out, r.v, 56); }
void f377_constants(uint32_t out[7 * 14]) {
  const fq c[7] = {fq_R1(), fq_R2(), fq_Q(), fq_kq_offset<2>(), fq_kq_offset<4>(), fq_kq_offset<8>(), fq_kq_offset<16>()};
  memcpy(out, c, sizeof c);
}
// body of the point conversion kernel: record in a 128-byte slot
void f377_prep_point(const uint8_t xy_le[96], uint8_t rec[128]) {
  uint32_t xw[12], yw[12]; memcpy(xw, xy_le, 48); memcpy(yw, xy_le + 48, 48);
  const g1a r = g1a_from_raw(fq_from_words32(xw), fq_from_words32(yw));
  memset(rec, 0, 128); memcpy(rec, &r, 112);
}
void f377_identity(uint8_t out[168]) { const g1p e = g1_identity(); memcpy(out, &e, 168); }
void f377_madd(const uint8_t acc[168], const uint8_t rec[128], int neg, uint8_t out[168]) {
  g1p a; g1a b; memcpy(&a, acc, 168); memcpy(&b, rec, 112);
  const g1p r = g1_madd(a, g1a_cneg(b, neg != 0)); memcpy(out, &r, 168);
}
void f377_add(const uint8_t a_[168], const uint8_t b_[168], uint8_t out[168]) {
  g1p a, b; memcpy(&a, a_, 168); memcpy(&b, b_, 168); const g1p r = g1_add(a, b); memcpy(out, &r, 168);
}
}
