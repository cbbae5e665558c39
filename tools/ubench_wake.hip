// ubench_wake.hip -- how long after a kernel's end does the host know?  (a) hipEventSynchronize on an event recorded behind the kernel,
// (b) hipStreamSynchronize, (c) the host spins on a word of pinned host memory that the kernel's last instruction writes.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_wake tools/ubench_wake.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <algorithm>
#include <vector>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_work(unsigned long long ticks, volatile unsigned* host_flag, unsigned v) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (host_flag && threadIdx.x == 0) { __threadfence_system(); *host_flag = v; }
}
int main() {
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  unsigned* flag; (void)hipHostMalloc((void**)&flag, 64); *flag = 0;
  int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  const unsigned long long ticks = (unsigned long long)khz / 5;     // 200 us of device work: the host is waiting when it ends
  for (int mode = 0; mode < 3; mode++) {
    std::vector<double> v;
    for (int i = 0; i < 60; i++) {
      *flag = 0;
      const double t0 = now_us();
      hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, ticks, mode == 2 ? flag : nullptr, (unsigned)(i + 1));
      if (mode == 0) { (void)hipEventRecord(ev, s); (void)hipEventSynchronize(ev); }
      else if (mode == 1) (void)hipStreamSynchronize(s);
      else { while (*(volatile unsigned*)flag != (unsigned)(i + 1)) {} }
      v.push_back(now_us() - t0 - 200.0);
      (void)hipStreamSynchronize(s);
    }
    std::sort(v.begin(), v.end());
    printf("%s: launch + wake-up overhead beyond the kernel's 200 us: median %.1f us, best %.1f us\n", mode == 0 ? "hipEventSynchronize  " : mode == 1 ? "hipStreamSynchronize " : "spin on pinned flag  ", v[v.size() / 2], v[0]);
  }
  return 0;
}
