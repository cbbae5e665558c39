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
#include <stdlib.h>

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

// r < 2p  ->  r mod p, without a branch: the comparison with p is data-dependent and unpredictable, and a mispredicted
// branch per field operation costs more than the operation's arithmetic (it also empties the window the independent products
// of a doubling overlap in) -- measured: 240 -> 130 ns per doubling
static inline Fe reduce_once(const Fe& r) {
  Fe d; uint64_t br = 0;
  for (int i = 0; i < 4; i++) { const u128 x = (u128)r.l[i] - MOD[i] - br; d.l[i] = (uint64_t)x; br = (uint64_t)(x >> 64) & 1; }
  const uint64_t keep = (uint64_t)0 - br;            // all ones when r < p
  Fe o;
  for (int i = 0; i < 4; i++) o.l[i] = (r.l[i] & keep) | (d.l[i] & ~keep);
  return o;
}
static inline Fe add(const Fe& a, const Fe& b) {
  Fe r; u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  return reduce_once(r);             // a, b < p < 2^253: no carry out
}
static inline Fe sub(const Fe& a, const Fe& b) {
  Fe r; uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  const uint64_t m = (uint64_t)0 - br;               // borrow: add p back
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + (MOD[i] & m); r.l[i] = (uint64_t)c; c >>= 64; }
  return r;
}
// a * b / R mod p, portable form (CIOS; also the reference the assembly form below is checked against at start-up)
static inline Fe mul_c(const Fe& a, const Fe& b) {
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
#if defined(__x86_64__)
// The same product with mulx and the two independent carry chains of adcx / adox (BMI2 + ADX), four rounds of
// "t += a * b_i; t = (t + m p) / 2^64".  p < 2^253 leaves three spare bits, so the running value never needs a fifth word
// beyond the round's own carry (the "no-carry" interleaving of the gnark / blst field code).  About 2.5 times the speed of the
// compiler's code for mul_c, on the 256 dependent doublings of Horner's rule -- the one serial stretch of an MSM that runs on
// a host core.  Used when the CPU has both extensions (have_adx(), checked once; tail_selftest compares the two forms).
__attribute__((target("bmi2,adx"))) static inline Fe mul_adx(const Fe& a, const Fe& b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, A = 0, B = 0;
#define TE_HOST_MUL_ROUND(i)                                                                                          \
    "movq " #i "*8(%[b]), %%rdx\n\t"                                                                                 \
    "xorq %%rax, %%rax\n\t"                                                                                          \
    "mulxq 0(%[a]), %%rax, %[A]\n\t"  "adoxq %%rax, %[t0]\n\t" "adcxq %[A], %[t1]\n\t"                              \
    "mulxq 8(%[a]), %%rax, %[A]\n\t"  "adoxq %%rax, %[t1]\n\t" "adcxq %[A], %[t2]\n\t"                              \
    "mulxq 16(%[a]), %%rax, %[A]\n\t" "adoxq %%rax, %[t2]\n\t" "adcxq %[A], %[t3]\n\t"                              \
    "mulxq 24(%[a]), %%rax, %[A]\n\t" "adoxq %%rax, %[t3]\n\t"                                                      \
    "movl $0, %%eax\n\t" "adcxq %%rax, %[A]\n\t" "adoxq %%rax, %[A]\n\t"                                            \
    "movq %[t0], %%rdx\n\t" "imulq %[ninv], %%rdx\n\t"                                                              \
    "xorq %%rax, %%rax\n\t"                                                                                          \
    "mulxq %[q0], %%rax, %[B]\n\t" "adcxq %[t0], %%rax\n\t" "movq %[B], %[t0]\n\t"                                  \
    "adcxq %[t1], %[t0]\n\t" "mulxq %[q1], %%rax, %[t1]\n\t" "adoxq %%rax, %[t0]\n\t"                               \
    "adcxq %[t2], %[t1]\n\t" "mulxq %[q2], %%rax, %[t2]\n\t" "adoxq %%rax, %[t1]\n\t"                               \
    "adcxq %[t3], %[t2]\n\t" "mulxq %[q3], %%rax, %[t3]\n\t" "adoxq %%rax, %[t2]\n\t"                               \
    "movl $0, %%eax\n\t" "adcxq %%rax, %[t3]\n\t" "adoxq %[A], %[t3]\n\t"
  __asm__(TE_HOST_MUL_ROUND(0) TE_HOST_MUL_ROUND(1) TE_HOST_MUL_ROUND(2) TE_HOST_MUL_ROUND(3)
          : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [A] "+&r"(A), [B] "+&r"(B)
          : [a] "r"(a.l), [b] "r"(b.l), [ninv] "m"(MOD_NEG_INV), [q0] "m"(MOD[0]), [q1] "m"(MOD[1]), [q2] "m"(MOD[2]), [q3] "m"(MOD[3]),
            "m"(*(const uint64_t(*)[4])a.l), "m"(*(const uint64_t(*)[4])b.l)
          : "rax", "rdx", "cc");
#undef TE_HOST_MUL_ROUND
  const Fe r = {{t0, t1, t2, t3}};
  return reduce_once(r);
}
static inline bool have_adx() {           // TE_MSM_HOST_MUL=c forces the portable form (A/B measurements)
  static const bool v = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx") && !(getenv("TE_MSM_HOST_MUL") && getenv("TE_MSM_HOST_MUL")[0] == 'c');
  return v;
}
static inline Fe mul(const Fe& a, const Fe& b) { return have_adx() ? mul_adx(a, b) : mul_c(a, b); }
#else
static inline bool have_adx() { return false; }
static inline Fe mul(const Fe& a, const Fe& b) { return mul_c(a, b); }
#endif
// a * a / R mod p: the 6 cross products once, doubled, then the same interleaved reduction (about 3/4 of mul's multiplications)
static inline Fe sqr_c(const Fe& a) {
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
static inline Fe sqr(const Fe& a) { return have_adx() ? mul(a, a) : sqr_c(a); }
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

}  // namespace te_host
#include "host_tail_ifma.hpp"
namespace te_host {

// partials: W rows of 720 B = [T | W0 | W1 | W2 | W3]: T = sum of the window's buckets, Wk = sum_v v * M_k[v] for digit k of
// the bucket index (bucket_bits = c - 1 for signed digits, c for unsigned; digit widths w_k = (bucket_bits + 3 - k) / 4,
// e.g. 4,4,4,3 for 15 bits).  Window value
//     V = T + W0 + 2^w0 W1 + 2^(w0+w1) W2 + 2^(w0+w1+w2) W3,      result = sum_w 2^(c*w) V_w,
// evaluated top-down; the c doublings per window are split around the digit terms, so no doubling is added.
// `sets` row buffers are summed on the fly: the rows are linear in the bucket contents, so an MSM computed in pieces
// (te_msm_run uploads and processes a large host buffer in chunks) folds as the sum of the pieces' rows.
static inline Fe tail_k2d() {
  const Fe d2 = {{2 * 3021, 0, 0, 0}};
  const Fe R2 = {{0x25d577bab861857bULL, 0xcc2c27b58860591fULL, 0xa7cc008fe5dc8593ULL, 0x011fdae7eff1c939ULL}};
  return mul(d2, R2);
}
// The accumulator of Horner's rule in two forms with one interface (identity, k doublings, + a point, -> affine x | y):
// ScalarAcc: four 64-bit limbs, mulx / adcx products one after the other (any x86-64, and the reference for the other form);
// IfmaAcc (host_tail_ifma.hpp): the four coordinates in the lanes of AVX-512 registers, two vector products per doubling.
static inline void affine_out(const Pt& acc, uint8_t out_xy_le[64]) {
  const Fe zi = inv(acc.z);
  const Fe one_raw = {{1, 0, 0, 0}};
  const Fe x = mul(mul(acc.x, zi), one_raw), y = mul(mul(acc.y, zi), one_raw);
  memcpy(out_xy_le, x.l, 32); memcpy(out_xy_le + 32, y.l, 32);
}
struct ScalarAcc {
  Pt acc; Fe k2d;
  ScalarAcc() : acc(identity()), k2d(tail_k2d()) {}
  void dbl_n(int k) { acc = pdbl_n(acc, k); }
  void add_point(const Pt& q) { acc = padd(acc, q, k2d); }
  void to_affine(uint8_t out_xy_le[64]) const { affine_out(acc, out_xy_le); }
};
#if defined(__x86_64__)
// 52-bit limbs of p, -p^-1 mod 2^52, 2 p, 4 p (re-derived in tests/test_host_logic.py::test_host_tail_forms_agree via the self-test)
static const te_ifma::field52<5> TE_F52 = {
  {0x1800000000001ULL, 0xfed00000010a1ULL, 0xc37b00159aa76ULL, 0xa55660b44d1e5ULL, 0x12ab655e9a2cULL}, 0x17fffffffffffULL,
  {0x3000000000002ULL, 0xfda0000002142ULL, 0x86f6002b354edULL, 0x4aacc1689a3cbULL, 0x2556cabd3459ULL},
  {0x6000000000004ULL, 0xfb40000004284ULL, 0xdec00566a9dbULL, 0x955982d134797ULL, 0x4aad957a68b2ULL}};
static inline bool have_ifma() { return te_ifma::cpu_has_ifma(); }
#define TE_IFMA_M __attribute__((target("avx512f,avx512ifma,avx512dq,avx512vl")))
struct IfmaAcc {
  te_ifma::V<5> acc; Fe k2d;
  TE_IFMA_M IfmaAcc() : k2d(tail_k2d()) { const Fe zero = {{0, 0, 0, 0}}, one = {{1, 0, 0, 0}}; acc = te_ifma::from_words<5, 4>(zero.l, one.l, one.l, zero.l); }
  TE_IFMA_M void dbl_n(int k) { for (int i = 0; i < k; i++) acc = te_ifma::vdbl<5>(acc, TE_F52); }
  // the operand of an addition prepared on the scalar side: lanes [Y - X, Y + X, 2 Z, 2 d T] (the host's Montgomery form of 2 d
  // applied with the host's product: the result carries T's own factor, like the other three)
  TE_IFMA_M void add_point(const Pt& q) {
    const Fe a = sub(q.y, q.x), b = add(q.y, q.x), c = add(q.z, q.z), d = mul(q.t, k2d);
    acc = te_ifma::vaddp<5>(acc, te_ifma::from_words<5, 4>(a.l, b.l, c.l, d.l), TE_F52);
  }
  TE_IFMA_M void to_affine(uint8_t out_xy_le[64]) const {
    uint64_t w[4][4]; te_ifma::to_words<5, 4>(acc, w);
    Pt r; Fe* c[4] = {&r.x, &r.y, &r.z, &r.t};
    for (int k = 0; k < 4; k++) { Fe v; memcpy(v.l, w[k], 32); *c[k] = reduce_once(v); }       // below 2 p in the lanes
    affine_out(r, out_xy_le);
  }
};
#endif
// Horner over the W windows; points_of(w, slot, emit) calls emit(point) for every point of that window and slot
template <typename Acc, typename F> static inline void horner_with(F&& points_of, int c, int bucket_bits, int W, uint8_t out_xy_le[64]) {
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
template <typename F> static inline void horner_core(F&& points_of, int c, int bucket_bits, int W, uint8_t out_xy_le[64]) {
#if defined(__x86_64__)
  if (have_ifma()) { horner_with<IfmaAcc>(points_of, c, bucket_bits, W, out_xy_le); return; }
#endif
  horner_with<ScalarAcc>(points_of, c, bucket_bits, W, out_xy_le);
}
static inline void horner_to_affine_multi(const uint8_t* const* partials, int sets, int c, int bucket_bits, int W, uint8_t out_xy_le[64]) {
  horner_core([&](int w, int slot, auto& emit) {
    for (int s = 0; s < sets; s++) {
      const uint8_t* row = partials[s] + (size_t)w * TE_TAIL_ROW_BYTES;
      if (!all_zero_bytes(row, TE_TAIL_ROW_BYTES)) emit(load_point(row + (size_t)slot * TE_TAIL_POINT_BYTES));
    }
  }, c, bucket_bits, W, out_xy_le);
}
// The same in two steps, for the multi-device te_msm_run: the sets' rows of ONE window summed slot by slot (independent per
// window: the devices' host threads share the windows), then Horner over the merged points.
//   merged: W x 5 points, present[w] = 0 when no set holds the window
static inline void merge_window_rows(const uint8_t* const* partials, int sets, int w, Pt* merged, uint8_t* present) {
  const Fe k2d = tail_k2d();
  present[w] = 0;
  for (int s = 0; s < sets; s++) {
    const uint8_t* row = partials[s] + (size_t)w * TE_TAIL_ROW_BYTES;
    if (all_zero_bytes(row, TE_TAIL_ROW_BYTES)) continue;
    for (int slot = 0; slot < 5; slot++) {
      const Pt p = load_point(row + (size_t)slot * TE_TAIL_POINT_BYTES);
      merged[(size_t)w * 5 + slot] = present[w] ? padd(merged[(size_t)w * 5 + slot], p, k2d) : p;
    }
    present[w] = 1;
  }
}
static inline void horner_to_affine_points(const Pt* merged, const uint8_t* present, int c, int bucket_bits, int W, uint8_t out_xy_le[64]) {
  horner_core([&](int w, int slot, auto& emit) { if (present[w]) emit(merged[(size_t)w * 5 + slot]); }, c, bucket_bits, W, out_xy_le);
}
static inline void horner_to_affine(const uint8_t* partials, int c, int bucket_bits, int W, uint8_t out_xy_le[64]) {
  horner_to_affine_multi(&partials, 1, c, bucket_bits, W, out_xy_le);
}
// FIXED-BASE WINDOWS (kernels.hip.hpp, k_fb_digits): `rows` pseudo-windows of 2^bucket_bits buckets share one bucket set; row r
// holds the buckets whose index is hb(r) 2^bucket_bits + lo, hb(r) = r for r < rb and 0 for the extra rows r >= rb (the top window).
// The device rows are the usual [T | W0 | W1 | W2 | W3]; the result has no window doublings at all:
//     sum_r ( T_r + W0_r + 2^w0 W1_r + 2^(w0+w1) W2_r + 2^(w0+w1+w2) W3_r )  +  2^bucket_bits sum_r hb(r) T_r
// i.e. ONE window whose slots are the sums over the rows, with U = sum_r hb(r) T_r (a running sum from the top: 2 rb additions) on
// top of it -- bucket_bits doublings in all, against c per window of Horner's rule.
template <typename Acc> static inline void fixed_base_with(const uint8_t* partials, int rows, int rb, int bucket_bits, uint8_t out_xy_le[64]) {
  int dw[4];
  for (int k = 0; k < 4; k++) dw[k] = (bucket_bits + 3 - k) / 4;
  const Fe k2d = tail_k2d();
  auto row_of = [&](int r) { return partials + (size_t)r * TE_TAIL_ROW_BYTES; };
  auto present = [&](int r) { return !all_zero_bytes(row_of(r), TE_TAIL_ROW_BYTES); };
  // U = sum_{r < rb} r T_r:  S = T_{rb-1}; U = S; S += T_{rb-2}; U += S; ... down to r = 1
  Pt S = identity(), U = identity();
  for (int r = rb - 1; r >= 1; r--) {
    if (present(r)) S = padd(S, load_point(row_of(r)), k2d);
    U = padd(U, S, k2d);
  }
  Acc acc;
  acc.add_point(U);
  auto slot_sum = [&](int slot) { for (int r = 0; r < rows; r++) if (present(r)) acc.add_point(load_point(row_of(r) + (size_t)slot * TE_TAIL_POINT_BYTES)); };
  acc.dbl_n(dw[3]); slot_sum(4);
  acc.dbl_n(dw[2]); slot_sum(3);
  acc.dbl_n(dw[1]); slot_sum(2);
  acc.dbl_n(dw[0]); slot_sum(1);
  slot_sum(0);
  acc.to_affine(out_xy_le);
}
static inline void fixed_base_to_affine(const uint8_t* partials, int rows, int rb, int bucket_bits, uint8_t out_xy_le[64]) {
#if defined(__x86_64__)
  if (have_ifma()) { fixed_base_with<IfmaAcc>(partials, rows, rb, bucket_bits, out_xy_le); return; }
#endif
  fixed_base_with<ScalarAcc>(partials, rows, rb, bucket_bits, out_xy_le);
}

static inline bool tail_selftest_run() {
  if ((uint64_t)(MOD[0] * MOD_NEG_INV) != ~0ULL) return false;     // p * (-p^-1) = -1 mod 2^64
  const Fe one_raw = {{1, 0, 0, 0}};
  const Fe t = mul(ONE_M, one_raw);                                // R * 1 / R = 1
  if (!(t.l[0] == 1 && !t.l[1] && !t.l[2] && !t.l[3])) return false;
  Fe v = {{0x243f6a8885a308d3ULL, 0x13198a2e03707344ULL, 0xa4093822299f31d0ULL, 0x082efa98ec4e6c89ULL}};     // below p
  Fe u = ONE_M;
  for (int i = 0; i < 64; i++) {                                   // every form of the product against the portable one
    const Fe a = sqr_c(v), b = mul_c(v, v), c = mul(v, u), d = mul_c(v, u), e = sqr(v);
    if (memcmp(a.l, b.l, 32) != 0 || memcmp(c.l, d.l, 32) != 0 || memcmp(e.l, b.l, 32) != 0) return false;
    u = add(c, v); v = add(a, ONE_M);
  }
  const Fe top = {{MOD[0] - 1, MOD[1], MOD[2], MOD[3]}};           // p - 1: the widest operands
  { const Fe c = mul(top, top), d = mul_c(top, top); if (memcmp(c.l, d.l, 32) != 0) return false; }
#if defined(__x86_64__)
  if (have_ifma()) {
    // the two accumulators over the same sequence of doublings and additions (the generator, its multiples as they come out of the
    // scalar form, the neutral element, a point with the widest coordinates): the affine results must be the same 64 bytes
    const Fe R2 = {{0x25d577bab861857bULL, 0xcc2c27b58860591fULL, 0xa7cc008fe5dc8593ULL, 0x011fdae7eff1c939ULL}};
    Pt g; g.x = Fe{{0x137e82844bbe49c5ULL, 0xe7608833a9dd83f3ULL, 0x16b294b80d905006ULL, 0x036824eb02475007ULL}};
    g.y = Fe{{0xd50dce7d8bcda9d4ULL, 0x7f6758f4c08bc255ULL, 0x37c0a81e810abce5ULL, 0x11b1d8d5c1d897a3ULL}};
    g.z = one_raw; g.t = mul(mul(g.x, g.y), R2);                   // x y as a plain integer
    ScalarAcc a; IfmaAcc b;
    Pt q = g;
    for (int round = 0; round < 12; round++) {
      a.add_point(q); b.add_point(q);
      a.dbl_n(1 + round % 5); b.dbl_n(1 + round % 5);
      if (round == 3) { const Pt id = identity(); a.add_point(id); b.add_point(id); }
      if (round == 7) { Pt w = q; w.x = top; w.y = top; w.z = top; w.t = top; a.add_point(w); b.add_point(w); }     // not a curve point: the formulas are polynomial identities
      q = a.acc;                                                    // the next operand: a projective point with general Z
      uint8_t oa[64], ob[64];
      a.to_affine(oa); b.to_affine(ob);
      if (memcmp(oa, ob, 64) != 0) return false;
    }
  }
#endif
  return true;
}
// run once per process: the context-free entry points (te_msm_finalize_host*, every MSM of a window-sharded pipeline) ask every time
static inline bool tail_selftest() { static const bool ok = tail_selftest_run(); return ok; }

}  // namespace te_host
