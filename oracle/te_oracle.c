/*
 * te_oracle.c -- CPU restatement of the reference's Twisted-Edwards-BLS12 MSM pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libtemsm.so, the HIP path) never links,
 * loads or calls anything in oracle/.
 *
 * What it restates (paths relative to /root/reference/src):
 *   - field / curve:      reference/utils/FieldMath.ts:7-10,104-137 (p, a = p-1, d = 3021, generator)
 *   - add_points:         submission/implementation/wgsl/curve/ec.template.wgsl:34-66
 *                         (add-2008-hwcd, a = -1), twin submission/miscellaneous/add_points.ts:4-53
 *   - double_point:       submission/implementation/wgsl/curve/ec.template.wgsl:7-30 (dbl-2008-hwcd)
 *   - signed digits:      submission/miscellaneous/utils.ts:52-95 (decompose_scalars_signed),
 *                         GPU twin wgsl/cuzk/convert_point_coords_and_decompose_scalars.template.wgsl:98-120
 *   - transpose:          submission/miscellaneous/transpose.ts:14-62 (cpu_transpose, counting sort)
 *   - smvp:               submission/miscellaneous/smvp.ts:37-102 (cpu_smvp_signed)
 *   - bucket reduction:   submission/miscellaneous/bpr.ts:4-131 (running sum; parallel stage 1 / 2)
 *   - tail:               submission/submission.ts:362-412 (sum of g points, Horner, toAffine)
 *   - whole pipeline:     submission/miscellaneous/tests/cuzk.test.ts:28-141
 *
 * Third-party arithmetic not in /root/reference: @noble/curves 1.0.0 (ExtendedPoint add / double /
 * multiplyUnsafe / toAffine).  Its published algorithm (complete unified extended-coordinates
 * addition on a twisted Edwards curve, double-and-add) is restated here; pinned by the KATs of
 * reference/utils/FieldMath.test.ts:5-95 and reference/utils/wasmFunctions.test.ts:28-49 (tests/).
 *
 * Parity pin: tests/test_oracle_kat.py checks this file against those KATs and against
 * tests/golden/msm_wasm_golden.json, produced in the build container by the reference's own CPU MSM
 * (Aleo WASM Address.msm, reference/reference.ts:29-39) via oracle/gen_golden.py + oracle/wasm_msm.js.
 *
 * Arithmetic: 4 x 64-bit limbs, Montgomery form with R = 2^256, values kept fully reduced (< p).
 * Nothing here is tuned; it is written to be obviously right.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;            /* field element, Montgomery form, < p */
typedef struct { fe x, y, t, z; } pt;            /* extended twisted Edwards (X:Y:T:Z), x=X/Z, y=Y/Z, T=XY/Z */

/* p = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001  (params.ts:11-13) */
static const uint64_t P[4] = { 0x0a11800000000001ULL, 0x59aa76fed0000001ULL,
                               0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL };

static uint64_t N0;                  /* -p^-1 mod 2^64 */
static fe R1, R2, FE_D, FE_ZERO;     /* R mod p, R^2 mod p, d = 3021 in Montgomery form */
static pt PT_ZERO;                   /* identity (0 : 1 : 0 : 1) */
static int g_init = 0;

/* ------------------------------------------------------------------ raw 256-bit helpers */
static int ge_p(const uint64_t a[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] > P[i]) return 1; if (a[i] < P[i]) return 0; }
  return 1;
}
static uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - b[i] - br; r[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
  }
  return br;
}

/* ------------------------------------------------------------------ field (Fp of FieldMath.ts:27) */
static void fe_add(fe *r, const fe *a, const fe *b) {
  uint64_t t[4]; uint64_t c = add4(t, a->l, b->l);
  if (c || ge_p(t)) sub4(t, t, P);
  memcpy(r->l, t, 32);
}
static void fe_sub(fe *r, const fe *a, const fe *b) {
  uint64_t t[4];
  if (sub4(t, a->l, b->l)) add4(t, t, P);
  memcpy(r->l, t, 32);
}
static void fe_neg(fe *r, const fe *a) { fe_sub(r, &FE_ZERO, a); }

/* Montgomery product a*b/R mod p (CIOS, 64-bit words). */
static void fe_mul(fe *r, const fe *a, const fe *b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64;
    }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * N0;
    c = (u128)m * P[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64;
    }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || ge_p(t)) sub4(t, t, P);
  memcpy(r->l, t, 32);
}
static void fe_sqr(fe *r, const fe *a) { fe_mul(r, a, a); }

static void fe_from_raw(fe *r, const uint64_t raw[4]) {     /* raw may be any 256-bit value */
  fe t; memcpy(t.l, raw, 32);
  /* bring below p by repeated subtraction (2^256 < 14p) */
  while (ge_p(t.l)) sub4(t.l, t.l, P);
  fe_mul(r, &t, &R2);
}
static void fe_to_raw(uint64_t raw[4], const fe *a) {
  fe one = {{1, 0, 0, 0}}, t; fe_mul(&t, a, &one); memcpy(raw, t.l, 32);
}
static int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int fe_eq(const fe *a, const fe *b) { return memcmp(a->l, b->l, 32) == 0; }

static void fe_pow(fe *r, const fe *a, const uint64_t e[4]) {
  fe acc = R1, base = *a;
  for (int i = 0; i < 256; i++) {
    if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(&acc, &acc, &base);
    fe_sqr(&base, &base);
  }
  *r = acc;
}
static void fe_inv(fe *r, const fe *a) {                    /* Fermat: a^(p-2) */
  uint64_t e[4]; uint64_t two[4] = {2, 0, 0, 0}; sub4(e, P, two); fe_pow(r, a, e);
}

/* ------------------------------------------------------------------ curve */
/* add_points, ec.template.wgsl:34-66: add-2008-hwcd with a = -1 (H = B + A), k = d */
static void pt_add(pt *r, const pt *p1, const pt *p2) {
  fe A, B, C, D, E, F, G, H, t0, t1;
  fe_mul(&A, &p1->x, &p2->x);
  fe_mul(&B, &p1->y, &p2->y);
  fe_mul(&t0, &p1->t, &p2->t); fe_mul(&C, &FE_D, &t0);
  fe_mul(&D, &p1->z, &p2->z);
  fe_add(&t0, &p1->x, &p1->y); fe_add(&t1, &p2->x, &p2->y);
  fe_mul(&E, &t0, &t1); fe_sub(&E, &E, &A); fe_sub(&E, &E, &B);
  fe_sub(&F, &D, &C);
  fe_add(&G, &D, &C);
  fe_add(&H, &B, &A);
  fe_mul(&r->x, &E, &F); fe_mul(&r->y, &G, &H); fe_mul(&r->t, &E, &H); fe_mul(&r->z, &F, &G);
}
/* double_point, ec.template.wgsl:7-30: dbl-2008-hwcd with a = -1 */
static void pt_dbl(pt *r, const pt *p1) {
  fe A, B, C, D, E, F, G, H, t0;
  fe_sqr(&A, &p1->x);
  fe_sqr(&B, &p1->y);
  fe_sqr(&C, &p1->z); fe_add(&C, &C, &C);
  fe_neg(&D, &A);
  fe_add(&t0, &p1->x, &p1->y); fe_sqr(&E, &t0); fe_sub(&E, &E, &A); fe_sub(&E, &E, &B);
  fe_add(&G, &D, &B);
  fe_sub(&F, &G, &C);
  fe_sub(&H, &D, &B);
  fe_mul(&r->x, &E, &F); fe_mul(&r->y, &G, &H); fe_mul(&r->t, &E, &H); fe_mul(&r->z, &F, &G);
}
/* negate_point, smvp.template.wgsl:47-56 */
static void pt_neg(pt *r, const pt *a) { fe_neg(&r->x, &a->x); r->y = a->y; fe_neg(&r->t, &a->t); r->z = a->z; }

static void pt_from_affine_raw(pt *r, const uint64_t x[4], const uint64_t y[4]) {
  fe_from_raw(&r->x, x); fe_from_raw(&r->y, y); fe_mul(&r->t, &r->x, &r->y); r->z = R1;
}
static void pt_to_affine_raw(uint64_t x[4], uint64_t y[4], const pt *a) {   /* toAffine(), submission.ts:412 */
  fe zi, t; fe_inv(&zi, &a->z);
  fe_mul(&t, &a->x, &zi); fe_to_raw(x, &t);
  fe_mul(&t, &a->y, &zi); fe_to_raw(y, &t);
}
/* k*P by left-to-right double-and-add over the integer k (noble's multiplyUnsafe semantics:
 * no reduction of k; FieldMath.ts:73-88). */
static void pt_mul_raw(pt *r, const pt *a, const uint64_t k[4]) {
  pt acc = PT_ZERO;
  for (int i = 255; i >= 0; i--) {
    pt_dbl(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) pt_add(&acc, &acc, a);
  }
  *r = acc;
}
static void pt_mul_small(pt *r, const pt *a, uint64_t k) { uint64_t kk[4] = {k, 0, 0, 0}; pt_mul_raw(r, a, kk); }

/* ------------------------------------------------------------------ init */
static void init_once(void) {
  if (g_init) return;
  /* N0 = -p^-1 mod 2^64 by Newton iteration */
  uint64_t inv = 1; for (int i = 0; i < 7; i++) inv *= 2 - P[0] * inv;
  N0 = (uint64_t)0 - inv;
  /* R mod p and R^2 mod p by modular doubling of 1 */
  uint64_t v[4] = {1, 0, 0, 0};
  for (int i = 0; i < 512; i++) {
    uint64_t c = add4(v, v, v);
    if (c || ge_p(v)) sub4(v, v, P);
    if (i == 255) memcpy(R1.l, v, 32);
  }
  memcpy(R2.l, v, 32);
  memset(&FE_ZERO, 0, sizeof FE_ZERO);
  uint64_t d[4] = {3021, 0, 0, 0}; fe_from_raw(&FE_D, d);        /* AleoConstants.ts:2-4 */
  PT_ZERO.x = FE_ZERO; PT_ZERO.y = R1; PT_ZERO.t = FE_ZERO; PT_ZERO.z = R1;
  g_init = 1;
}

/* ------------------------------------------------------------------ byte codecs (wire format) */
static void le_to_raw(uint64_t r[4], const uint8_t *b) { memcpy(r, b, 32); }   /* little-endian host */
static void raw_to_le(uint8_t *b, const uint64_t r[4]) { memcpy(b, r, 32); }

/* ================================================================== pipeline stages */

/* decompose_scalars_signed, miscellaneous/utils.ts:52-95.  chunks[w*n + i] = digit + 2^(c-1).
 * Returns 0, or -1 if a final carry remains (the reference throws "final carry is 1"). */
int ora_decompose_scalars_signed(const uint8_t *scalars_le, uint64_t n, int c, int num_words, uint32_t *chunks) {
  const uint32_t l = 1u << c, shift = l >> 1;
  for (uint64_t i = 0; i < n; i++) {
    uint64_t s[5]; le_to_raw(s, scalars_le + 32 * i); s[4] = 0;
    uint32_t carry = 0;
    for (int w = 0; w < num_words; w++) {
      int bit = w * c; uint32_t limb = 0;
      if (bit < 256) {
        limb = (uint32_t)(s[bit >> 6] >> (bit & 63));
        if ((bit & 63) + c > 64) limb |= (uint32_t)(s[(bit >> 6) + 1] << (64 - (bit & 63)));
        limb &= l - 1;
      }
      int64_t v = (int64_t)limb + carry;
      if (v >= (int64_t)(l / 2)) { v = -((int64_t)l - v); carry = 1; } else carry = 0;
      chunks[(uint64_t)w * n + i] = (uint32_t)(v + shift);
    }
    if (carry) return -1;
  }
  return 0;
}

/* cpu_transpose, miscellaneous/transpose.ts:14-62: per subtask, counting sort of point indices by chunk
 * value.  col_ptr has num_subtasks*(ncols+1) entries, val_idx has num_subtasks*n entries. */
void ora_transpose(const uint32_t *chunks, uint64_t n, uint32_t ncols, int num_subtasks,
                   uint32_t *col_ptr, uint32_t *val_idx) {
  uint32_t *curr = (uint32_t *)malloc(sizeof(uint32_t) * ncols);
  for (int st = 0; st < num_subtasks; st++) {
    uint32_t *cp = col_ptr + (uint64_t)st * (ncols + 1);
    const uint32_t *ch = chunks + (uint64_t)st * n;
    memset(cp, 0, sizeof(uint32_t) * (ncols + 1));
    memset(curr, 0, sizeof(uint32_t) * ncols);
    for (uint64_t j = 0; j < n; j++) cp[ch[j] + 1]++;
    for (uint32_t i = 1; i < ncols + 1; i++) cp[i] += cp[i - 1];
    for (uint64_t j = 0; j < n; j++) {
      uint32_t loc = cp[ch[j]] + curr[ch[j]]++;
      val_idx[(uint64_t)st * n + loc] = (uint32_t)j;
    }
  }
  free(curr);
}

/* cpu_smvp_signed, miscellaneous/smvp.ts:37-102: buckets[t], t in [0, ncols/2). */
static void smvp_signed(int st, uint64_t n, uint32_t ncols, const uint32_t *col_ptr,
                        const uint32_t *val_idx, const pt *points, pt *buckets) {
  const uint32_t h = ncols / 2;
  const uint32_t *cp = col_ptr + (uint64_t)st * (ncols + 1);
  for (uint32_t tid = 0; tid < h; tid++) {
    pt acc = PT_ZERO;
    for (int j = 0; j < 2; j++) {
      uint32_t row = (j == 0) ? tid + h : h - tid;
      if (tid == 0 && j == 0) row = 0;
      pt sum = PT_ZERO;
      for (uint32_t k = cp[row]; k < cp[row + 1]; k++) pt_add(&sum, &sum, &points[val_idx[(uint64_t)st * n + k]]);
      uint32_t bucket_idx;
      if (h > row) { bucket_idx = h - row; pt_neg(&sum, &sum); } else bucket_idx = row - h;
      if (bucket_idx > 0) pt_add(&acc, &acc, &sum); else pt_add(&acc, &acc, &PT_ZERO);
    }
    buckets[tid] = acc;
  }
}

/* running_sum_bucket_reduction, miscellaneous/bpr.ts:4-24 */
static void bpr_running_sum(pt *g_out, const pt *buckets, uint32_t nb) {
  pt m = buckets[0], g = m;
  for (uint32_t i = 0; i + 1 < nb; i++) { pt_add(&m, &m, &buckets[nb - 1 - i]); pt_add(&g, &g, &m); }
  *g_out = g;
}
/* parallel_bucket_reduction_1 / _2, miscellaneous/bpr.ts:73-131 (GPU twins bpr.template.wgsl:73-171):
 * nt simulated threads; returns the sum of the nt g points (submission.ts:369-393 does that sum on the CPU). */
static void bpr_parallel(pt *g_out, const pt *buckets, uint32_t nb, uint32_t nt) {
  uint32_t bpt = nb / nt; pt total = PT_ZERO;
  for (uint32_t tid = 0; tid < nt; tid++) {
    uint32_t idx = tid == 0 ? 0 : (nt - tid) * bpt;
    pt m = buckets[idx], g = m;
    for (uint32_t i = 0; i + 1 < bpt; i++) {
      pt_add(&m, &m, &buckets[(nt - tid) * bpt - 1 - i]); pt_add(&g, &g, &m);
    }
    uint64_t s = (uint64_t)bpt * (nt - tid - 1);
    if (s > 0) { pt sm; pt_mul_small(&sm, &m, s); pt_add(&g, &g, &sm); }
    pt_add(&total, &total, &g);
  }
  *g_out = total;
}

typedef struct {
  int st_begin, st_end; uint64_t n; uint32_t ncols; const uint32_t *col_ptr, *val_idx;
  const pt *points; pt *window_sums; int bpr_mode; uint32_t bpr_threads;
} win_job;

static void *win_worker(void *arg) {
  win_job *j = (win_job *)arg;
  uint32_t h = j->ncols / 2;
  pt *buckets = (pt *)malloc(sizeof(pt) * h);
  for (int st = j->st_begin; st < j->st_end; st++) {
    smvp_signed(st, j->n, j->ncols, j->col_ptr, j->val_idx, j->points, buckets);
    if (j->bpr_mode == 0 || h < j->bpr_threads || (h % j->bpr_threads) != 0) bpr_running_sum(&j->window_sums[st], buckets, h);
    else bpr_parallel(&j->window_sums[st], buckets, h, j->bpr_threads);
  }
  free(buckets);
  return NULL;
}

/*
 * The whole reference pipeline (cuzk.test.ts:28-141 / submission.ts:73-413) on the CPU.
 *   points_xy_le: n * 64 B (x || y little-endian, canonical affine); scalars_le: n * 32 B.
 *   c: window bits (reference: 16 if n >= 65536 else 4, submission.ts:80); num_subtasks = ceil(256/c).
 *   bpr_mode 0 = serial running sum, 1 = the 256-thread (or fewer) parallel split of bpr.ts:73-131.
 *   threads: pthreads over windows (1 = scalar port).
 *   window_sums_out (optional): num_subtasks * 128 B, each window's G_w as affine-independent
 *   raw (x, y, t, z) NOT exported; only the affine result is.
 * Returns 0 on success, -1 final carry, -2 bad args.
 */
int ora_msm(const uint8_t *points_xy_le, const uint8_t *scalars_le, uint64_t n, int c, int bpr_mode,
            int threads, uint8_t out_xy_le[64]) {
  init_once();
  if (c < 2 || c > 16) return -2;
  int num_subtasks = (256 + c - 1) / c;
  uint32_t ncols = 1u << c;
  uint64_t rx[4], ry[4];
  if (n == 0) { pt_to_affine_raw(rx, ry, &PT_ZERO); raw_to_le(out_xy_le, rx); raw_to_le(out_xy_le + 32, ry); return 0; }

  pt *points = (pt *)malloc(sizeof(pt) * n);
  for (uint64_t i = 0; i < n; i++) {
    uint64_t x[4], y[4]; le_to_raw(x, points_xy_le + 64 * i); le_to_raw(y, points_xy_le + 64 * i + 32);
    pt_from_affine_raw(&points[i], x, y);
  }
  uint32_t *chunks = (uint32_t *)malloc(sizeof(uint32_t) * n * num_subtasks);
  int rc = ora_decompose_scalars_signed(scalars_le, n, c, num_subtasks, chunks);
  if (rc) { free(points); free(chunks); return rc; }
  uint32_t *col_ptr = (uint32_t *)malloc(sizeof(uint32_t) * (uint64_t)num_subtasks * (ncols + 1));
  uint32_t *val_idx = (uint32_t *)malloc(sizeof(uint32_t) * (uint64_t)num_subtasks * n);
  ora_transpose(chunks, n, ncols, num_subtasks, col_ptr, val_idx);

  pt *wsum = (pt *)malloc(sizeof(pt) * num_subtasks);
  if (threads < 1) threads = 1;
  if (threads > num_subtasks) threads = num_subtasks;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
  win_job *jobs = (win_job *)malloc(sizeof(win_job) * threads);
  for (int t = 0; t < threads; t++) {
    jobs[t] = (win_job){ (int)((int64_t)num_subtasks * t / threads), (int)((int64_t)num_subtasks * (t + 1) / threads),
                         n, ncols, col_ptr, val_idx, points, wsum, bpr_mode, 256 };
    if (threads == 1) win_worker(&jobs[t]); else pthread_create(&th[t], NULL, win_worker, &jobs[t]);
  }
  if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);

  /* Horner, submission.ts:400-407: result = result * 2^c + G_w from the top window down */
  pt result = wsum[num_subtasks - 1];
  for (int w = num_subtasks - 2; w >= 0; w--) {
    for (int k = 0; k < c; k++) pt_dbl(&result, &result);
    pt_add(&result, &result, &wsum[w]);
  }
  pt_to_affine_raw(rx, ry, &result);
  raw_to_le(out_xy_le, rx); raw_to_le(out_xy_le + 32, ry);
  free(points); free(chunks); free(col_ptr); free(val_idx); free(wsum); free(th); free(jobs);
  return 0;
}

/* Naive sum_i k_i * P_i by double-and-add (the "expected" leg of cuzk.test.ts:127-137). */
int ora_msm_naive(const uint8_t *points_xy_le, const uint8_t *scalars_le, uint64_t n, uint8_t out_xy_le[64]) {
  init_once();
  pt acc = PT_ZERO;
  for (uint64_t i = 0; i < n; i++) {
    uint64_t x[4], y[4], k[4]; pt p, kp;
    le_to_raw(x, points_xy_le + 64 * i); le_to_raw(y, points_xy_le + 64 * i + 32); le_to_raw(k, scalars_le + 32 * i);
    pt_from_affine_raw(&p, x, y); pt_mul_raw(&kp, &p, k); pt_add(&acc, &acc, &kp);
  }
  uint64_t rx[4], ry[4]; pt_to_affine_raw(rx, ry, &acc);
  raw_to_le(out_xy_le, rx); raw_to_le(out_xy_le + 32, ry);
  return 0;
}

/* ------------------------------------------------------------------ small exported primitives for KATs */
void ora_point_add_affine(const uint8_t a_xy[64], const uint8_t b_xy[64], uint8_t out_xy[64]) {
  init_once();
  uint64_t x[4], y[4]; pt a, b, r;
  le_to_raw(x, a_xy); le_to_raw(y, a_xy + 32); pt_from_affine_raw(&a, x, y);
  le_to_raw(x, b_xy); le_to_raw(y, b_xy + 32); pt_from_affine_raw(&b, x, y);
  pt_add(&r, &a, &b); pt_to_affine_raw(x, y, &r); raw_to_le(out_xy, x); raw_to_le(out_xy + 32, y);
}
void ora_point_double_affine(const uint8_t a_xy[64], uint8_t out_xy[64]) {
  init_once();
  uint64_t x[4], y[4]; pt a, r;
  le_to_raw(x, a_xy); le_to_raw(y, a_xy + 32); pt_from_affine_raw(&a, x, y);
  pt_dbl(&r, &a); pt_to_affine_raw(x, y, &r); raw_to_le(out_xy, x); raw_to_le(out_xy + 32, y);
}
void ora_scalar_mul_affine(const uint8_t a_xy[64], const uint8_t k_le[32], uint8_t out_xy[64]) {
  init_once();
  uint64_t x[4], y[4], k[4]; pt a, r;
  le_to_raw(x, a_xy); le_to_raw(y, a_xy + 32); le_to_raw(k, k_le); pt_from_affine_raw(&a, x, y);
  pt_mul_raw(&r, &a, k); pt_to_affine_raw(x, y, &r); raw_to_le(out_xy, x); raw_to_le(out_xy + 32, y);
}
/* field ops on canonical little-endian values (for wasmFunctions.test.ts:4-26 style KATs) */
void ora_field_op(int op, const uint8_t a_le[32], const uint8_t b_le[32], uint8_t out_le[32]) {
  init_once();
  uint64_t a[4], b[4], r[4]; fe fa, fb, fr;
  le_to_raw(a, a_le); le_to_raw(b, b_le); fe_from_raw(&fa, a); fe_from_raw(&fb, b);
  switch (op) {
    case 0: fe_add(&fr, &fa, &fb); break;
    case 1: fe_sub(&fr, &fa, &fb); break;
    case 2: fe_mul(&fr, &fa, &fb); break;
    case 3: fe_inv(&fr, &fa); break;
    default: fr = FE_ZERO;
  }
  fe_to_raw(r, &fr); raw_to_le(out_le, r);
}
int ora_on_curve(const uint8_t a_xy[64]) {          /* -x^2 + y^2 == 1 + d x^2 y^2 */
  init_once();
  uint64_t x[4], y[4]; fe fx, fy, x2, y2, l, r, t;
  le_to_raw(x, a_xy); le_to_raw(y, a_xy + 32); fe_from_raw(&fx, x); fe_from_raw(&fy, y);
  fe_sqr(&x2, &fx); fe_sqr(&y2, &fy); fe_sub(&l, &y2, &x2);
  fe_mul(&t, &x2, &y2); fe_mul(&t, &t, &FE_D); fe_add(&r, &R1, &t);
  return fe_eq(&l, &r) && !fe_is_zero(&R1);
}

/* ------------------------------------------------------------------ deterministic synthetic inputs
 * (shared by tests, bench.py and oracle/gen_golden.py; mirrored in pure Python in oracle/model.py)
 *   splitmix64 stream; scalars = 256 random bits reduced mod p (the harness's distribution,
 *   reference/webgpu/utils.ts:81-88,118-124); points P_i = (a + i*b) * G built as a chain
 *   P_0 = a*G, P_{i+1} = P_i + b*G, normalised to affine with one batched inversion. */
static uint64_t splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void rand_mod_p(uint64_t r[4], uint64_t *s) {
  for (int i = 0; i < 4; i++) r[i] = splitmix64(s);
  while (ge_p(r)) sub4(r, r, P);
}
void ora_gen_scalars(uint64_t seed, uint64_t n, uint8_t *scalars_le) {
  init_once();
  uint64_t s = seed;
  for (uint64_t i = 0; i < n; i++) { uint64_t r[4]; rand_mod_p(r, &s); raw_to_le(scalars_le + 32 * i, r); }
}
static uint64_t g_gx[4], g_gy[4]; static int g_have_gen = 0;
/* generator (FieldMath.ts:108-109) is passed in by the caller as canonical LE bytes, so that the
 * decimal constants live in exactly one place (oracle/model.py) and are KAT-checked there. */
void ora_set_generator(const uint8_t g_xy[64]) { le_to_raw(g_gx, g_xy); le_to_raw(g_gy, g_xy + 32); g_have_gen = 1; }

int ora_gen_points(uint64_t seed, uint64_t n, uint8_t *points_xy_le) {
  init_once();
  if (!g_have_gen) return -2;
  if (n == 0) return 0;
  uint64_t s = seed ^ 0xA5A5A5A55A5A5A5AULL, a[4], b[4];
  rand_mod_p(a, &s); rand_mod_p(b, &s);
  pt g, p0, q; pt_from_affine_raw(&g, g_gx, g_gy);
  pt_mul_raw(&p0, &g, a); pt_mul_raw(&q, &g, b);
  pt *pts = (pt *)malloc(sizeof(pt) * n);
  pts[0] = p0;
  for (uint64_t i = 1; i < n; i++) pt_add(&pts[i], &pts[i - 1], &q);
  /* batched inversion of the z coordinates */
  fe *pre = (fe *)malloc(sizeof(fe) * n);
  fe acc = R1;
  for (uint64_t i = 0; i < n; i++) { pre[i] = acc; fe_mul(&acc, &acc, &pts[i].z); }
  fe inv; fe_inv(&inv, &acc);
  for (uint64_t i = n; i-- > 0;) {
    fe zi, t; uint64_t r[4];
    fe_mul(&zi, &inv, &pre[i]); fe_mul(&inv, &inv, &pts[i].z);
    fe_mul(&t, &pts[i].x, &zi); fe_to_raw(r, &t); raw_to_le(points_xy_le + 64 * i, r);
    fe_mul(&t, &pts[i].y, &zi); fe_to_raw(r, &t); raw_to_le(points_xy_le + 64 * i + 32, r);
  }
  free(pts); free(pre);
  return 0;
}

/* Set (R) of SURVEY.md 8d: n INDEPENDENT points P_i = a_i * G, a_i = the i-th value of the splitmix64 stream of
 * (seed ^ 0x5A5A5A5AA5A5A5A5), 256 bits reduced mod p.  Restated here with the plain double-and-add of pt_mul_raw (the
 * product generates the same points with a fixed-base table: tests compare the two); pthreads over ranges of i. */
typedef struct { const uint64_t (*ks)[4]; pt *pts; const pt *g; uint64_t lo, hi; } rnd_job;
static void *rnd_worker(void *arg) {
  rnd_job *j = (rnd_job *)arg;
  for (uint64_t i = j->lo; i < j->hi; i++) pt_mul_raw(&j->pts[i], j->g, j->ks[i]);
  return NULL;
}
int ora_gen_points_random(uint64_t seed, uint64_t n, uint8_t *points_xy_le, int threads) {
  init_once();
  if (!g_have_gen) return -2;
  if (n == 0) return 0;
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  uint64_t (*ks)[4] = (uint64_t (*)[4])malloc(sizeof(uint64_t[4]) * n);
  pt *pts = (pt *)malloc(sizeof(pt) * n);
  uint64_t s = seed ^ 0x5A5A5A5AA5A5A5A5ULL;
  for (uint64_t i = 0; i < n; i++) rand_mod_p(ks[i], &s);
  pt g; pt_from_affine_raw(&g, g_gx, g_gy);
  rnd_job jobs[64]; pthread_t th[64];
  const uint64_t per = (n + (uint64_t)threads - 1) / (uint64_t)threads;
  for (int t = 0; t < threads; t++) {
    uint64_t lo = per * (uint64_t)t; if (lo > n) lo = n;
    uint64_t hi = lo + per; if (hi > n) hi = n;
    jobs[t].ks = (const uint64_t (*)[4])ks; jobs[t].pts = pts; jobs[t].g = &g; jobs[t].lo = lo; jobs[t].hi = hi;
    if (threads == 1) rnd_worker(&jobs[t]); else pthread_create(&th[t], NULL, rnd_worker, &jobs[t]);
  }
  if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  for (uint64_t i = 0; i < n; i++) {
    uint64_t x[4], y[4];
    pt_to_affine_raw(x, y, &pts[i]);
    raw_to_le(points_xy_le + 64 * i, x); raw_to_le(points_xy_le + 64 * i + 32, y);
  }
  free(ks); free(pts);
  return 0;
}
