// clock_load.hip -- two pure-VALU loops that hold the GPU busy for a given time, for tools/clock_under_load.py (round-5 verdict,
// item 6: what holds the shader clock at 1.97-2.18 GHz under k_accumulate -- the power cap or the DPM state?):
//   fma32   v_fma_f32 chains, 16 independent per thread (full-rate VALU: the highest VALU power draw per cycle)
//   mad     the engine's own field product (csrc/fp.hpp, mont_mul_x<4>: v_mad_u64_u32 + carries), as in k_accumulate
// Every wave adds its shader-clock and wall-clock ticks to two counters (as k_accumulate does): the mean core clock over the run.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/clock_load tools/clock_load.hip
// run:   clock_load <fma32|mad> <seconds> [waves_per_simd=4]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include "../webgpu-msm-twisted-edwards_amd/csrc/fp.hpp"

__global__ void __launch_bounds__(256) k_fma32(float* out, int iters, unsigned long long* clk) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i);
  const float m = 1.0000001f, c = 1e-7f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 16; i++) a[i] = __builtin_fmaf(a[i], m, c);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i];
  if (s == 123.456f) out[0] = s;
  if ((threadIdx.x & 63) == 0) { atomicAdd(clk, clock64() - c0); atomicAdd(clk + 1, wall_clock64() - w0); }
}

__global__ void __launch_bounds__(256) k_mad(uint32_t* out, int iters, unsigned long long* clk) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  te::fp a[4], b[4];
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int i = 0; i < te::NL; i++) { a[m].v[i] = (threadIdx.x * 2654435761u + i * 40503u + m) & te::LM; b[m].v[i] = (blockIdx.x * 97u + i * 7919u + m * 13u) & te::LM; }
  for (int it = 0; it < iters; it++) {
    te::fp r[4];
    te::mont_mul_x<4>(a, b, r);
#pragma unroll
    for (int m = 0; m < 4; m++) a[m] = r[m];
  }
  uint32_t s = 0;
#pragma unroll
  for (int m = 0; m < 4; m++) s ^= a[m].v[0] ^ a[m].v[8];
  if (s == 0x12345u) out[0] = s;
  if ((threadIdx.x & 63) == 0) { atomicAdd(clk, clock64() - c0); atomicAdd(clk + 1, wall_clock64() - w0); }
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: clock_load <fma32|mad> <seconds> [waves_per_simd]\n"); return 2; }
  const bool mad = !strcmp(argv[1], "mad");
  const double secs = atof(argv[2]);
  const int wps = argc > 3 ? atoi(argv[3]) : 4;
  int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const int blocks = pr.multiProcessorCount * wps;          // 256 threads = 4 waves per block = 1 per SIMD: wps blocks per CU
  void* out; unsigned long long* clk;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&clk, 16); (void)hipMemset(clk, 0, 16);
  // calibrate: one launch of ~20 ms
  int iters = mad ? 200 : 2000;
  auto launch = [&](int it) { if (mad) hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, (uint32_t*)out, it, clk); else hipLaunchKernelGGL(k_fma32, dim3(blocks), dim3(256), 0, 0, (float*)out, it, clk); };
  launch(iters); (void)hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  launch(iters); (void)hipDeviceSynchronize();
  const double ms1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  iters = (int)(iters * 20.0 / (ms1 > 0.01 ? ms1 : 0.01)); if (iters < 1) iters = 1;
  (void)hipMemset(clk, 0, 16);
  t0 = std::chrono::steady_clock::now();
  int launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int k = 0; k < 4; k++) launch(iters);
    (void)hipDeviceSynchronize(); launches += 4;
  }
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = h[1] ? (double)h[0] / (double)h[1] * khz * 1e-6 : 0.0;
  // issue rate: instructions of the loop body per wave and second
  const double per_iter = mad ? 4.0 * (153.0 + 45.0) : 8.0 * 16.0;
  const double inst_per_s_per_simd = per_iter * iters * launches * wps / el;
  printf("RESULT kind=%s waves_per_simd=%d seconds=%.2f launches=%d iters=%d mean_core_clock_ghz=%.3f loop_instr_per_simd_per_s=%.3e cycles_per_instr_at_that_clock=%.2f\n",
         argv[1], wps, el, launches, iters, ghz, inst_per_s_per_simd, ghz * 1e9 / inst_per_s_per_simd);
  return 0;
}
