// ubench_gather.hip -- calibration of rocprofv3's FETCH_SIZE for k_accumulate's access pattern.
// MI355X_MICROARCH.md: "on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ...
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".
// Kernels with KNOWN byte counts, run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (tools/calibrate_fetch.sh):
//   k_stream        every lane reads 16 B of a 1 GiB buffer once, coalesced          -> bytes = 1 GiB
//   k_gather<T>     every lane reads the first 112 B (7 x 16 B) of a random 128-B slot of a table of T MiB, `passes`
//                   times over fresh random permutations -- k_accumulate's load_pnt   -> lines touched = lanes x passes x 128 B
// Reported: FETCH_SIZE x 1024 / known bytes per kernel = the factor to apply to this pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ src, uint32_t* out, size_t n16) {
  uint32_t x = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256ull) { const uint4 v = src[i]; x ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (x == 0x12345u) out[0] = x;
}
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// slots = power of two; every (pass, lane) pair picks slot = bijection(lane index) so that each pass touches every slot
// of the first `lanes` exactly once in a scrambled order (odd multiplier + xor = a permutation of [0, slots))
template <int TAG>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, uint32_t* out, uint32_t slots_mask, uint32_t lanes, int passes) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= lanes) return;
  uint32_t x = 0;
  for (int p = 0; p < passes; p++) {
    const uint32_t slot = ((gid * 2654435761u) ^ mix(0x9e3779b9u * (uint32_t)(p + 1))) & slots_mask;
    const uint4* q = table + (size_t)slot * 8u;
#pragma unroll
    for (int j = 0; j < 7; j++) { const uint4 v = q[j]; x ^= v.x ^ v.w; }
  }
  if (x == 0x12345u) out[0] = x;
}
__global__ void __launch_bounds__(256) k_fill(uint4* p, size_t n16) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256ull) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
template <int TAG> void run_gather(const uint4* table, uint32_t* out, size_t table_bytes, uint32_t lanes, int passes, const char* what) {
  const uint32_t slots = (uint32_t)(table_bytes / 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_gather<TAG>, dim3((lanes + 255) / 256), dim3(256), 0, 0, table, out, slots - 1u, lanes, passes);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)lanes * passes * 128.0;
  printf("CAL k_gather<%d> %-58s known_bytes %.0f  %8.1f us  %6.2f TB/s of lines\n", TAG, what, bytes, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}
int main() {
  const size_t big = 1ull << 30;
  uint4* buf; uint32_t* out;
  if (hipMalloc(&buf, big) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, big / 16);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, buf, out, big / 16);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("CAL k_stream %-62s known_bytes %.0f  %8.1f us  %6.2f TB/s\n", "1 GiB, 16 B per lane, coalesced", (double)big, ms * 1e3, big / (ms * 1e-3) / 1e12);
  const uint32_t lanes = 1u << 20;
  // <1>: 1 GiB table, one pass of 2^20 lanes: every line is cold (HBM)           <2>: 128 MiB table = k_accumulate's record
  // table at n = 2^20, 16 passes (one per window): after the first pass the table sits in the 256 MiB Infinity Cache
  // <3>: 128 MiB, one pass (first touch after the big fill has flushed it)        <4>: 16 MiB table, 16 passes (mostly L2)
  run_gather<1>(buf, out, big, lanes, 1, "1 GiB table, 1 pass (HBM)");
  hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, buf, out, big / 16);     // flush the caches with the big buffer
  run_gather<3>(buf, out, 128ull << 20, lanes, 1, "128 MiB table, 1 pass, caches flushed");
  run_gather<2>(buf, out, 128ull << 20, lanes, 16, "128 MiB table, 16 passes (Infinity Cache after pass 1)");
  run_gather<4>(buf, out, 16ull << 20, lanes, 16, "16 MiB table, 16 passes (L2-resident share)");
  hipDeviceSynchronize();
  return 0;
}
