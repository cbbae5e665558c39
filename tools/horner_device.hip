// horner_device.hip -- SURVEY 8f rank 4 / VERDICT r3 item 8: Horner's rule ON THE DEVICE (the reference's exploratory K6,
// miscellaneous/wgsl/horners_rule.template.wgsl:45-78: one thread, `double_and_add(result, 2^c)` + add per window) against the
// host tail the engine uses (te_host::horner_to_affine, ~42 us on one core).
// The device form here is the best this engine has for a serial chain: a QUAD of lanes per point (team addition, three dependent
// field products instead of nine), the doubling done by the complete unified addition (P + P), one wave, points in registers,
// rows read from device memory.  Work: W windows x (c doublings + 5 additions) -- 256 + 80 dependent team additions at c = 16.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/horner_device tools/horner_device.hip -Lwebgpu-msm-twisted-edwards_amd -ltemsm -Wl,-rpath,$PWD/webgpu-msm-twisted-edwards_amd
// Run on the GPU box:  tools/horner_device [log2n]     (prints both times and checks the device result against the host tail)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/te_msm.h"
#include "../webgpu-msm-twisted-edwards_amd/csrc/host_tail.hpp"
#include "../webgpu-msm-twisted-edwards_amd/csrc/kernels.hip.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// rows: W x [T | W0 | W1 | W2 | W3] (te_msm_partial_device); out: one extended point (projective result, before the inversion)
__global__ void __launch_bounds__(64) k_horner_team(const te::ete* __restrict__ rows, int W, int c, int bucket_bits, te::ete* __restrict__ out) {
  using namespace te;
  if (threadIdx.x >= 4u) return;                       // one quad
  const uint32_t q = threadIdx.x & 3u, wq = team_word<9>(q);
  int dw[4];
  for (int k = 0; k < 4; k++) dw[k] = (bucket_bits + 3 - k) / 4;
  const int s3 = dw[0] + dw[1] + dw[2];
  fel<9> acc = identity_coord<9>(q);
  auto dbl = [&](int k) { for (int i = 0; i < k; i++) acc = ete_add_team<9>(acc, acc, q); };
  auto add_slot = [&](int w, int slot) { acc = ete_add_team<9>(acc, load_coord<9>(words<9>(rows + (size_t)w * 5 + slot) + wq), q); };
  for (int w = W - 1; w >= 0; w--) {
    dbl(c - s3); add_slot(w, 4);
    dbl(dw[2]); add_slot(w, 3);
    dbl(dw[1]); add_slot(w, 2);
    dbl(dw[0]); add_slot(w, 1);
    add_slot(w, 0);
  }
  store_coord<9>(words<9>(out) + wq, acc);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 16;
  const uint64_t n = 1ull << lg;
  std::vector<uint8_t> pts(64 * n), sc(32 * n);
  te_msm_synth_inputs(0x5EED0000 + lg, n, 0, pts.data(), sc.data());
  te_ctx* ctx = nullptr; int dev = 0;
  if (te_msm_init(&dev, 1, &ctx)) { fprintf(stderr, "init: %s\n", te_msm_last_error(nullptr)); return 1; }
  int c = 0, W = 0; te_msm_plan(ctx, n, &c, &W);
  void *dp, *ds, *drows; te::ete* dout;
  CK(hipMalloc(&dp, pts.size())); CK(hipMalloc(&ds, sc.size())); CK(hipMalloc(&drows, (size_t)W * 720)); CK(hipMalloc((void**)&dout, sizeof(te::ete)));
  CK(hipMemcpy(dp, pts.data(), pts.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(ds, sc.data(), sc.size(), hipMemcpyHostToDevice));
  CK(hipMemset(drows, 0, (size_t)W * 720));
  if (te_msm_partial_device(ctx, dp, ds, n, drows, TE_MSM_OWN_STREAM) || te_msm_partial_wait(ctx, 0)) { fprintf(stderr, "msm: %s\n", te_msm_last_error(ctx)); return 1; }
  std::vector<uint8_t> rows((size_t)W * 720);
  CK(hipMemcpy(rows.data(), drows, rows.size(), hipMemcpyDeviceToHost));
  // host tail
  uint8_t want[64]; double best_host = 1e9;
  for (int r = 0; r < 50; r++) {
    const auto t0 = std::chrono::steady_clock::now();
    te_msm_finalize_host(rows.data(), c, W, want);
    best_host = std::min(best_host, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  // device chain
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best_dev = 1e9f;
  for (int r = 0; r < 10; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_horner_team, dim3(1), dim3(64), 0, 0, (const te::ete*)drows, W, c, c - 1, dout);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best_dev = std::min(best_dev, ms);
  }
  uint8_t raw[144]; CK(hipMemcpy(raw, dout, 144, hipMemcpyDeviceToHost));
  // affine conversion of the device's projective result on the host (one inversion), compared with the host tail's result
  using namespace te_host;
  const Pt p = load_point(raw);
  const Fe zi = inv(p.z); const Fe one_raw = {{1, 0, 0, 0}};
  const Fe x = mul(mul(p.x, zi), one_raw), y = mul(mul(p.y, zi), one_raw);
  uint8_t got[64]; memcpy(got, x.l, 32); memcpy(got + 32, y.l, 32);
  const bool ok = memcmp(got, want, 64) == 0;
  printf("n = 2^%d, c = %d, W = %d: %d dependent team additions on the device\n", lg, c, W, W * (c + 5));
  printf("  device Horner chain (one quad, team additions, k_horner_team): %.1f us   [result %s the host tail's]\n", best_dev * 1e3, ok ? "equals" : "DIFFERS FROM");
  printf("  host tail te_host::horner_to_affine (Horner + inversion, one core): %.1f us\n", best_host);
  printf("  -> per dependent team addition on an otherwise idle GPU: %.2f us\n", best_dev * 1e3 / (W * (c + 5)));
  te_msm_destroy(ctx);
  return ok ? 0 : 2;
}
