/* bls377_oracle.c -- CPU oracle for BASELINE config 5: MSM on BLS12-377 G1 (y^2 = x^3 + 1, 377-bit base field).
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  PARITY UNPINNED: the reference
 * holds no code, test or vector for this curve -- only prose (README.md:57-73,279-287,325-349).  This file restates the
 * reference's cuZK pipeline (signed digits miscellaneous/utils.ts:52-95, bucket sums smvp.ts:37-102, running-sum bucket
 * reduction bpr.ts:4-131, Horner submission.ts:369-407) with the group law README.md:285-287 names ("projective
 * algorithms"), and is itself checked against the pure-Python bigint model oracle/model377.py and the group axioms
 * (tests/test_oracle_bls377.py).
 *
 * Field: 6 x 64-bit limbs, Montgomery R = 2^384.  Group law: complete projective addition (Renes-Costello-Batina 2016,
 * Algorithm 7 with a = 0, b3 = 3), valid for every pair of inputs including doubling and the point at infinity.
 * Wire format: points n x (x || y), 48-byte little-endian each; scalars n x 48-byte little-endian (values < r);
 * result x || y (96 bytes), the point at infinity as 96 zero bytes. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fe;
typedef struct { fe x, y, z; } pt;

static const uint64_t QM[6] = {0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL,
                               0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x01ae3a4617c510eaULL};
static uint64_t N0;            /* -q^-1 mod 2^64 */
static fe R1, R2m, ZERO;
static int g_init = 0;

static int ge_q(const uint64_t a[6]) { for (int i = 5; i >= 0; i--) { if (a[i] != QM[i]) return a[i] > QM[i]; } return 1; }
static void sub_q(uint64_t a[6]) { uint64_t br = 0; for (int i = 0; i < 6; i++) { u128 d = (u128)a[i] - QM[i] - br; a[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
static void fe_add(fe *r, const fe *a, const fe *b) {
  u128 c = 0; for (int i = 0; i < 6; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
  if (ge_q(r->l)) sub_q(r->l);
}
static void fe_sub(fe *r, const fe *a, const fe *b) {
  uint64_t br = 0; for (int i = 0; i < 6; i++) { u128 d = (u128)a->l[i] - b->l[i] - br; r->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 6; i++) { c += (u128)r->l[i] + QM[i]; r->l[i] = (uint64_t)c; c >>= 64; } }
}
static void fe_mul(fe *r, const fe *a, const fe *b) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 6; i++) {
    u128 c = 0;
    for (int j = 0; j < 6; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * N0;
    c = ((u128)m * QM[0] + t[0]) >> 64;
    for (int j = 1; j < 6; j++) { c += (u128)m * QM[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
  }
  memcpy(r->l, t, 48);
  if (t[6] || ge_q(r->l)) sub_q(r->l);
}
static int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0; }
static void fe_inv(fe *r, const fe *a) {            /* a^(q-2) */
  uint64_t e[6]; memcpy(e, QM, 48); e[0] -= 2;
  fe acc = R1, base = *a;
  for (int i = 0; i < 377; i++) { if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(&acc, &acc, &base); fe_mul(&base, &base, &base); }
  *r = acc;
}
static void init_once(void) {
  if (g_init) return;
  uint64_t inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - QM[0] * inv;      /* Newton: q^-1 mod 2^64 */
  N0 = (uint64_t)0 - inv;
  memset(&ZERO, 0, sizeof ZERO);
  fe one; memset(&one, 0, sizeof one); one.l[0] = 1;
  /* R mod q and R^2 mod q by doubling 1 (384 and 768 times) */
  fe t = one;
  for (int i = 0; i < 768; i++) { fe_add(&t, &t, &t); if (i == 383) R1 = t; }
  R2m = t;
  g_init = 1;
}
static void fe_from_le(fe *r, const uint8_t *b) { fe raw; memcpy(raw.l, b, 48); while (ge_q(raw.l)) sub_q(raw.l); fe_mul(r, &raw, &R2m); }
static void fe_to_le(uint8_t *b, const fe *a) { fe one; memset(&one, 0, sizeof one); one.l[0] = 1; fe raw; fe_mul(&raw, a, &one); memcpy(b, raw.l, 48); }

static void pt_identity(pt *p) { p->x = ZERO; p->y = R1; p->z = ZERO; }
static void pt_from_affine_le(pt *p, const uint8_t *xy) {      /* 96 zero bytes = infinity */
  int allz = 1; for (int i = 0; i < 96; i++) if (xy[i]) { allz = 0; break; }
  if (allz) { pt_identity(p); return; }
  fe_from_le(&p->x, xy); fe_from_le(&p->y, xy + 48); p->z = R1;
}
static void pt_neg(pt *r, const pt *a) { r->x = a->x; fe_sub(&r->y, &ZERO, &a->y); r->z = a->z; }
/* complete addition, RCB16 Algorithm 7 (a = 0, b3 = 3) */
static void pt_add(pt *r, const pt *p, const pt *q) {
  fe t0, t1, t2, t3, t4, x3, y3, z3, s;
  fe_mul(&t0, &p->x, &q->x); fe_mul(&t1, &p->y, &q->y); fe_mul(&t2, &p->z, &q->z);
  fe_add(&t3, &p->x, &p->y); fe_add(&t4, &q->x, &q->y); fe_mul(&t3, &t3, &t4); fe_add(&t4, &t0, &t1); fe_sub(&t3, &t3, &t4);
  fe_add(&t4, &p->y, &p->z); fe_add(&x3, &q->y, &q->z); fe_mul(&t4, &t4, &x3); fe_add(&x3, &t1, &t2); fe_sub(&t4, &t4, &x3);
  fe_add(&x3, &p->x, &p->z); fe_add(&y3, &q->x, &q->z); fe_mul(&x3, &x3, &y3); fe_add(&y3, &t0, &t2); fe_sub(&y3, &x3, &y3);
  fe_add(&x3, &t0, &t0); fe_add(&t0, &x3, &t0);
  fe_add(&s, &t2, &t2); fe_add(&t2, &s, &t2);                   /* t2 = 3 t2 */
  fe_add(&z3, &t1, &t2); fe_sub(&t1, &t1, &t2);
  fe_add(&s, &y3, &y3); fe_add(&y3, &s, &y3);                   /* y3 = 3 y3 */
  fe_mul(&x3, &t4, &y3); fe_mul(&t2, &t3, &t1); fe_sub(&x3, &t2, &x3);
  fe_mul(&y3, &y3, &t0); fe_mul(&t1, &t1, &z3); fe_add(&y3, &t1, &y3);
  fe_mul(&t0, &t0, &t3); fe_mul(&z3, &z3, &t4); fe_add(&z3, &z3, &t0);
  r->x = x3; r->y = y3; r->z = z3;
}
static void pt_to_affine_le(uint8_t out[96], const pt *p) {
  if (fe_is_zero(&p->z)) { memset(out, 0, 96); return; }
  fe zi, t; fe_inv(&zi, &p->z);
  fe_mul(&t, &p->x, &zi); fe_to_le(out, &t);
  fe_mul(&t, &p->y, &zi); fe_to_le(out + 48, &t);
}
static void pt_mul_le(pt *r, const pt *p, const uint8_t *k, int nbytes) {
  pt acc; pt_identity(&acc);
  for (int i = nbytes * 8 - 1; i >= 0; i--) { pt_add(&acc, &acc, &acc); if ((k[i >> 3] >> (i & 7)) & 1) pt_add(&acc, &acc, p); }
  *r = acc;
}

/* ---------------------------------------------------------------- pipeline */
static int decompose_signed(const uint8_t *s48, int c, int W, uint32_t *digits /* W */) {
  /* miscellaneous/utils.ts:52-95 on the low 256 bits; bytes 32..47 must be zero (values < r) */
  for (int i = 32; i < 48; i++) if (s48[i]) return -3;
  const uint32_t half = 1u << (c - 1), full = 1u << c; uint32_t carry = 0;
  for (int w = 0; w < W; w++) {
    uint32_t v = 0;
    for (int b = 0; b < c; b++) { const int bit = w * c + b; if (bit < 256 && ((s48[bit >> 3] >> (bit & 7)) & 1)) v |= 1u << b; }
    v += carry; carry = 0;
    if (v >= half) { digits[w] = v + half - full; carry = 1; } else digits[w] = v + half;      /* stored = digit + half */
  }
  return carry ? -3 : 0;
}
typedef struct { const pt *pts; const uint32_t *digits; uint64_t n; int c, W, w_lo, w_hi; pt *win; } job_t;
static void *window_job(void *arg) {
  job_t *j = (job_t *)arg;
  const uint32_t half = 1u << (j->c - 1);
  pt *buckets = (pt *)malloc(sizeof(pt) * half);
  for (int w = j->w_lo; w < j->w_hi; w++) {
    for (uint32_t b = 0; b < half; b++) pt_identity(&buckets[b]);
    for (uint64_t i = 0; i < j->n; i++) {
      const int d = (int)j->digits[i * j->W + w] - (int)half;
      if (!d) continue;
      pt p = j->pts[i]; if (d < 0) pt_neg(&p, &p);
      const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
      pt_add(&buckets[b], &buckets[b], &p);
    }
    pt run, tot; pt_identity(&run); pt_identity(&tot);
    for (int b = (int)half - 1; b >= 0; b--) { pt_add(&run, &run, &buckets[b]); pt_add(&tot, &tot, &run); }
    j->win[w] = tot;
  }
  free(buckets);
  return NULL;
}
int ora377_msm(const uint8_t *points, const uint8_t *scalars, uint64_t n, int c, int threads, uint8_t out[96]) {
  init_once();
  if (c < 2 || c > 16) return -1;
  const int W = (256 + c - 1) / c;
  pt *pts = (pt *)malloc(sizeof(pt) * (n ? n : 1));
  uint32_t *digits = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1) * W);
  for (uint64_t i = 0; i < n; i++) {
    pt_from_affine_le(&pts[i], points + 96 * i);
    if (decompose_signed(scalars + 48 * i, c, W, digits + i * W)) { free(pts); free(digits); return -3; }
  }
  pt *win = (pt *)malloc(sizeof(pt) * W);
  if (threads < 1) threads = 1;
  if (threads > W) threads = W;
  if (threads > 64) threads = 64;
  pthread_t th[64]; job_t jobs[64];
  for (int t = 0; t < threads; t++) {
    jobs[t] = (job_t){pts, digits, n, c, W, W * t / threads, W * (t + 1) / threads, win};
    pthread_create(&th[t], NULL, window_job, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  pt acc; pt_identity(&acc);
  for (int w = W - 1; w >= 0; w--) { for (int k = 0; k < c; k++) pt_add(&acc, &acc, &acc); pt_add(&acc, &acc, &win[w]); }
  pt_to_affine_le(out, &acc);
  free(win); free(pts); free(digits);
  return 0;
}
int ora377_msm_naive(const uint8_t *points, const uint8_t *scalars, uint64_t n, uint8_t out[96]) {
  init_once();
  pt acc; pt_identity(&acc);
  for (uint64_t i = 0; i < n; i++) { pt p, t; pt_from_affine_le(&p, points + 96 * i); pt_mul_le(&t, &p, scalars + 48 * i, 48); pt_add(&acc, &acc, &t); }
  pt_to_affine_le(out, &acc);
  return 0;
}
void ora377_add_affine(const uint8_t a[96], const uint8_t b[96], uint8_t out[96]) {
  init_once(); pt p, q, r; pt_from_affine_le(&p, a); pt_from_affine_le(&q, b); pt_add(&r, &p, &q); pt_to_affine_le(out, &r);
}
void ora377_scalar_mul_affine(const uint8_t a[96], const uint8_t k[48], uint8_t out[96]) {
  init_once(); pt p, r; pt_from_affine_le(&p, a); pt_mul_le(&r, &p, k, 48); pt_to_affine_le(out, &r);
}
int ora377_on_curve(const uint8_t a[96]) {
  init_once(); fe x, y, l, r; fe_from_le(&x, a); fe_from_le(&y, a + 48);
  fe_mul(&l, &y, &y); fe_mul(&r, &x, &x); fe_mul(&r, &r, &x); fe_add(&r, &r, &R1);
  return memcmp(&l, &r, sizeof l) == 0;
}

/* ---------------------------------------------------------------- seeded inputs (scheme of te_oracle.c / model377.py) */
static const uint64_t RM[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};
static uint64_t splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void rand_mod_r(uint64_t r[4], uint64_t *s) {
  for (int i = 0; i < 4; i++) r[i] = splitmix64(s);
  for (;;) {
    int ge = 1; for (int i = 3; i >= 0; i--) if (r[i] != RM[i]) { ge = r[i] > RM[i]; break; }
    if (!ge) break;
    uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)r[i] - RM[i] - br; r[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  }
}
void ora377_gen_scalars(uint64_t seed, uint64_t n, uint8_t *scalars) {
  uint64_t s = seed;
  for (uint64_t i = 0; i < n; i++) { uint64_t r[4]; rand_mod_r(r, &s); memset(scalars + 48 * i, 0, 48); memcpy(scalars + 48 * i, r, 32); }
}
int ora377_gen_points(uint64_t seed, uint64_t n, const uint8_t g_xy[96], uint8_t *points) {
  init_once();
  if (!n) return 0;
  uint64_t s = seed ^ 0xA5A5A5A55A5A5A5AULL, a[4], b[4];
  rand_mod_r(a, &s); rand_mod_r(b, &s);
  pt g, p0, q; pt_from_affine_le(&g, g_xy);
  pt_mul_le(&p0, &g, (const uint8_t *)a, 32); pt_mul_le(&q, &g, (const uint8_t *)b, 32);
  pt *pts = (pt *)malloc(sizeof(pt) * n);
  pts[0] = p0;
  for (uint64_t i = 1; i < n; i++) pt_add(&pts[i], &pts[i - 1], &q);
  fe *pre = (fe *)malloc(sizeof(fe) * n);
  fe acc = R1;
  for (uint64_t i = 0; i < n; i++) { pre[i] = acc; fe_mul(&acc, &acc, &pts[i].z); }      /* no infinity in the chain (order r) */
  fe inv; fe_inv(&inv, &acc);
  for (uint64_t i = n; i-- > 0;) {
    fe zi, t; fe_mul(&zi, &inv, &pre[i]); fe_mul(&inv, &inv, &pts[i].z);
    fe_mul(&t, &pts[i].x, &zi); fe_to_le(points + 96 * i, &t);
    fe_mul(&t, &pts[i].y, &zi); fe_to_le(points + 96 * i + 48, &t);
  }
  free(pts); free(pre);
  return 0;
}
