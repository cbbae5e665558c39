// queue_probe.hip -- which HIP streams of a process share a hardware queue?  The ROCm runtime multiplexes streams onto
// GPU_MAX_HW_QUEUES (default 4) hardware queues; kernels of one queue run in order.  Two one-block spin kernels launched
// on two streams take T when the streams sit on different queues and 2T when they share one.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/queue_probe tools/queue_probe.hip ; run: ./tools/queue_probe [streams]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

__global__ void spin(unsigned long long ticks, unsigned* out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (out) out[0] = 1;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 16;
  int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  const unsigned long long ticks = (unsigned long long)khz * 2;      // 2 ms
  std::vector<hipStream_t> s(n);
  for (int i = 0; i < n; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
  unsigned* d; hipMalloc(&d, 4);
  auto pair_ms = [&](int a, int b) {
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[a], ticks, d);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[b], ticks, d);
    hipStreamSynchronize(s[a]); hipStreamSynchronize(s[b]);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  pair_ms(0, 1);
  printf("GPU_MAX_HW_QUEUES=%s, %d streams created in order; X = the two streams share a hardware queue\n", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)", n);
  std::vector<int> cls(n, -1); int ncls = 0;
  for (int i = 0; i < n; i++) {
    if (cls[i] >= 0) continue;
    cls[i] = ncls;
    for (int j = i + 1; j < n; j++) if (cls[j] < 0 && pair_ms(i, j) > 3.0) cls[j] = ncls;
    ncls++;
  }
  printf("queue class of stream 0..%d:", n - 1);
  for (int i = 0; i < n; i++) printf(" %d", cls[i]);
  printf("   (%d classes)\n", ncls);
  return 0;
}
