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

static bool lt_kp(const fp& a, int k) {   // a < k*p ?
  // compare a with k*p using 64-bit math
  uint32_t kp[9]; uint64_t c = 0;
  for (int i = 0; i < 8; i++) { c += (uint64_t)p_limb(i) * k; kp[i] = (uint32_t)c; c >>= 32; }
  kp[8] = (uint32_t)c;
  if (kp[8]) return true;
  for (int i = 7; i >= 0; i--) { if (a.v[i] != kp[i]) return a.v[i] < kp[i]; }
  return false;
}
static int g_bound_violations = 0;
static void chk2(const ete& e) { if (!(lt_kp(e.x, 2) && lt_kp(e.y, 2) && lt_kp(e.z, 2) && lt_kp(e.t, 2))) g_bound_violations++; }

extern "C" {

int fpc_bound_violations() { return g_bound_violations; }

void fpc_mont_mul(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
  fp x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32); fp r = mont_mul(x, y); memcpy(out, r.v, 32);
}
void fpc_reduce_full(const uint32_t a[8], uint32_t out[8]) { fp x; memcpy(x.v, a, 32); fp r = fp_reduce_full(x); memcpy(out, r.v, 32); }
void fpc_half(const uint32_t a[8], uint32_t out[8]) { fp x; memcpy(x.v, a, 32); fp r = fp_half(x); memcpy(out, r.v, 32); }
void fpc_sub2(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) { fp x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32); fp r = fp_sub<2>(x, y); memcpy(out, r.v, 32); }
void fpc_constants(uint32_t out[5 * 8]) {
  fp c[5] = {fp_R1(), fp_R2(), fp_D_MONT(), fp_K2D_MONT(), fp_ONE_RAW()};
  memcpy(out, c, sizeof c);
}
// body of k_prep_points
void fpc_prep_point(const uint8_t xy_le[64], uint8_t rec[96]) {
  fp x, y; memcpy(x.v, xy_le, 32); memcpy(y.v, xy_le + 32, 32);
  const fp xm = fp_csub<1>(mont_mul(fp_R2(), x)), ym = fp_csub<1>(mont_mul(fp_R2(), y));
  const pnt r = pnt_from_affine_mont(xm, ym);
  memcpy(rec, &r, 96);
}
void fpc_identity(uint8_t out[128]) { ete e = ete_identity(); memcpy(out, &e, 128); }
void fpc_madd(const uint8_t acc[128], const uint8_t rec[96], int neg, uint8_t out[128]) {
  ete a; pnt b; memcpy(&a, acc, 128); memcpy(&b, rec, 96);
  ete r = ete_madd(a, pnt_cneg(b, neg != 0)); chk2(r); memcpy(out, &r, 128);
}
void fpc_add(const uint8_t a_[128], const uint8_t b_[128], uint8_t out[128]) {
  ete a, b; memcpy(&a, a_, 128); memcpy(&b, b_, 128); ete r = ete_add(a, b); chk2(r); memcpy(out, &r, 128);
}

// Emulation of the device stages for the windows w = first + k*step.  partials: W x 384 B (rows of other
// windows untouched).  Returns 0, or -3 on a final carry.
int fpc_partial_rows(const uint8_t* points, const uint8_t* scalars, uint64_t n, int c, int first, int step, uint8_t* partials) {
  const int W = (256 + c - 1) / c;
  const uint32_t B = 1u << (c - 1), lo_bits = (uint32_t)(c / 2), RL = 1u << lo_bits, RH = B / RL;
  std::vector<pnt> recs(n);
  for (uint64_t i = 0; i < n; i++) fpc_prep_point(points + 64 * i, (uint8_t*)&recs[i]);
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
    std::vector<ete> R(RH, ete_identity()), C(RL, ete_identity());
    for (uint32_t j = 0; j < B; j++) { R[j / RL] = ete_add(R[j / RL], bk[j]); C[j % RL] = ete_add(C[j % RL], bk[j]); chk2(R[j / RL]); chk2(C[j % RL]); }
    ete T = ete_identity(), WR = ete_identity(), WC = ete_identity(), run = ete_identity();
    for (uint32_t v = RH; v-- > 1;) { run = ete_add(run, R[v]); WR = ete_add(WR, run); }
    T = ete_add(run, R[0]);
    run = ete_identity();
    for (uint32_t v = RL; v-- > 1;) { run = ete_add(run, C[v]); WC = ete_add(WC, run); }
    chk2(T); chk2(WR); chk2(WC);
    uint8_t* row = partials + (size_t)w * 384;
    memcpy(row, &T, 128); memcpy(row + 128, &WR, 128); memcpy(row + 256, &WC, 128);
  }
  return 0;
}

}  // extern "C"
