/* msm_example.c -- the C-ABI of libtemsm.so (include/te_msm.h) from plain C: what a reference-side binding does under the hood.
 *
 *   gcc -std=c99 -O2 -Wall -I include examples/msm_example.c -L webgpu-msm-twisted-edwards_amd -ltemsm \
 *       -Wl,-rpath,$PWD/webgpu-msm-twisted-edwards_amd -o /tmp/msm_example && /tmp/msm_example [log2n] [device,device,...]
 *
 * 1. compute_msm(bufferPoints, bufferScalars) of the reference (submission/submission.ts:73-78) = te_msm_run on host buffers;
 * 2. several calls in flight (concurrent compute_msm promises, ui/Benchmark.tsx:32 is an async call) = tickets:
 *    te_msm_submit_async ... te_msm_collect, on one device or -- one whole MSM per device -- on several;
 * 3. resident bases -- the harness passes ONE point buffer to six calls per size (full_benchmarks.ts:63-68,100-105): te_msm_bind_points
 *    once, then te_msm_run_scalars / te_msm_submit_scalars move the scalars only;
 * 4. errors are negative codes with text (the reference throws: cuzk/gpu.ts:19-22, miscellaneous/utils.ts:80-83).
 * Prints the affine result (x, y little-endian hex) and whether every path returned the same 64 bytes. */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "te_msm.h"

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static void hex(const uint8_t* p, int n) { for (int i = n - 1; i >= 0; i--) printf("%02x", p[i]); }

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 16;
  int ids[64], nd = 0;
  if (argc > 2) { char* s = argv[2]; while (*s && nd < 64) { ids[nd++] = (int)strtol(s, &s, 10); if (*s == ',') s++; } }
  if (!nd) ids[nd++] = 0;
  const uint64_t n = 1ull << lg;
  uint8_t* points = malloc(TE_MSM_POINT_BYTES * n);
  uint8_t* scalars = malloc(TE_MSM_SCALAR_BYTES * n);
  if (!points || !scalars || te_msm_synth_inputs(0x5EED0000u + (unsigned)lg, n, TE_MSM_SYNTH_CHAIN, points, scalars)) { fprintf(stderr, "inputs\n"); return 2; }

  te_ctx* ctx = NULL;
  int rc = te_msm_init(ids, nd, &ctx);
  if (rc) { fprintf(stderr, "te_msm_init: %d: %s\n", rc, te_msm_last_error(NULL)); return 1; }   /* no device: no CPU fallback */

  uint8_t out[TE_MSM_RESULT_BYTES_MAX], ref[TE_MSM_RESULT_BYTES_MAX];
  if ((rc = te_msm_run(ctx, points, scalars, n, ref))) { fprintf(stderr, "te_msm_run: %d: %s\n", rc, te_msm_last_error(ctx)); return 1; }
  double t0 = now_ms();
  rc = te_msm_run(ctx, points, scalars, n, out);
  printf("n = 2^%d on %d device(s): te_msm_run %.3f ms\n  x = 0x", lg, nd, now_ms() - t0); hex(out, 32); printf("\n  y = 0x"); hex(out + 32, 32); printf("\n");
  int same = rc == 0 && memcmp(out, ref, 64) == 0;

  /* tickets: 2 per device in flight, collected in reverse order */
  enum { MAXT = 128 };
  uint64_t ticket[MAXT]; const int k = 2 * nd < MAXT ? 2 * nd : MAXT;
  t0 = now_ms();
  for (int i = 0; i < k; i++)
    if ((rc = te_msm_submit_async(ctx, points, scalars, n, &ticket[i]))) { fprintf(stderr, "te_msm_submit_async: %d: %s\n", rc, te_msm_last_error(ctx)); return 1; }
  for (int i = k - 1; i >= 0; i--) {
    int dev_index = -1, dev_id = -1;
    te_msm_ticket_device(ctx, ticket[i], &dev_index, &dev_id);
    if ((rc = te_msm_collect(ctx, ticket[i], out))) { fprintf(stderr, "te_msm_collect: %d: %s\n", rc, te_msm_last_error(ctx)); return 1; }
    same = same && memcmp(out, ref, 64) == 0;
    if (i == k - 1) printf("  last ticket ran on device %d (entry %d of the list)\n", dev_id, dev_index);
  }
  printf("  %d tickets in flight: %.3f ms per MSM; all results equal: %s\n", k, (now_ms() - t0) / k, same ? "yes" : "NO");

  /* resident bases: the points are uploaded and converted once (on every device of the context); every MSM then moves 32 of its 96
   * bytes per point.  The lone call, then tickets in flight over the same set; the set is released when no ticket uses it any more */
  te_bases* bases = NULL;
  t0 = now_ms();
  if ((rc = te_msm_bind_points(ctx, points, n, &bases))) { fprintf(stderr, "te_msm_bind_points: %d: %s\n", rc, te_msm_last_error(ctx)); return 1; }
  const double bind_ms = now_ms() - t0;
  if ((rc = te_msm_run_scalars(ctx, bases, scalars, out))) { fprintf(stderr, "te_msm_run_scalars: %d: %s\n", rc, te_msm_last_error(ctx)); return 1; }
  same = same && memcmp(out, ref, 64) == 0;
  t0 = now_ms();
  rc = te_msm_run_scalars(ctx, bases, scalars, out);
  const double lone_ms = now_ms() - t0;
  same = same && rc == 0 && memcmp(out, ref, 64) == 0;
  t0 = now_ms();
  for (int i = 0; i < k; i++)
    if ((rc = te_msm_submit_scalars(ctx, bases, scalars, &ticket[i]))) { fprintf(stderr, "te_msm_submit_scalars: %d: %s\n", rc, te_msm_last_error(ctx)); return 1; }
  same = same && te_msm_release_points(ctx, bases) == TE_MSM_ESTATE;            /* tickets over the set are in flight */
  for (int i = 0; i < k; i++) {
    if ((rc = te_msm_collect(ctx, ticket[i], out))) { fprintf(stderr, "te_msm_collect: %d: %s\n", rc, te_msm_last_error(ctx)); return 1; }
    same = same && memcmp(out, ref, 64) == 0;
  }
  printf("  bound bases (%llu points, bind %.3f ms): te_msm_run_scalars %.3f ms, %d tickets in flight %.3f ms per MSM; all results equal: %s\n",
         (unsigned long long)te_msm_bases_count(bases), bind_ms, lone_ms, k, (now_ms() - t0) / k, same ? "yes" : "NO");
  same = same && te_msm_release_points(ctx, bases) == 0;

  /* a scalar that does not fit the signed windows is an error of ITS call */
  memset(scalars, 0xff, TE_MSM_SCALAR_BYTES);
  te_msm_set_option(ctx, "window_bits", 16);
  rc = te_msm_run(ctx, points, scalars, n, out);
  printf("  scalar 2^256 - 1: code %d (%s)\n", rc, te_msm_last_error(ctx));
  same = same && rc == TE_MSM_ESCALAR;
  te_msm_destroy(ctx);
  free(points); free(scalars);
  return same ? 0 : 3;
}
