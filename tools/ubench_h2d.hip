// ubench_h2d.hip -- can an own staging path beat hipMemcpy from pageable memory for the 96 MB of a 2^20-point MSM?
//   (a) hipMemcpy(pageable -> device), one call and in K pieces (what te_msm_run does today)
//   (b) T host threads copy pageable -> pinned mirror in 4 MB chunks, the main thread issues hipMemcpyAsync(pinned -> device) per chunk
//       as soon as the chunk is in the mirror; reports when the LAST byte is on the device and when each third of the data is
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_h2d tools/ubench_h2d.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const size_t bytes = 96ull << 20, chunk = (argc > 1 ? atoi(argv[1]) : 4) * (1ull << 20);
  const int T = argc > 2 ? atoi(argv[2]) : 8;
  char* src = (char*)malloc(bytes); memset(src, 1, bytes);
  char *pin, *dev; CK(hipHostMalloc((void**)&pin, bytes)); CK(hipMalloc((void**)&dev, bytes)); memset(pin, 2, bytes);
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now_us(); CK(hipMemcpy(dev, src, bytes, hipMemcpyHostToDevice)); double t1 = now_us();
    printf("pageable hipMemcpy, 1 call : %7.1f us  %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e3);
    for (int K : {3, 6, 12}) {
      t0 = now_us();
      for (int k = 0; k < K; k++) CK(hipMemcpyAsync(dev + k * (bytes / K), src + k * (bytes / K), bytes / K, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s)); t1 = now_us();
      printf("pageable hipMemcpyAsync, %2d calls: %7.1f us  %.1f GB/s\n", K, t1 - t0, bytes / (t1 - t0) / 1e3);
    }
    t0 = now_us(); CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t1 = now_us();
    printf("pinned   hipMemcpyAsync, 1 call : %7.1f us  %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e3);
    // host memcpy alone
    {
      t0 = now_us();
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++) th.emplace_back([&, t] { const size_t per = bytes / T; memcpy(pin + t * per, src + t * per, per); });
      for (auto& x : th) x.join();
      t1 = now_us();
      printf("host memcpy pageable -> pinned, %d threads (incl. thread start): %7.1f us  %.1f GB/s\n", T, t1 - t0, bytes / (t1 - t0) / 1e3);
    }
    // staged: persistent threads take chunks in order (atomic counter); main thread issues the async copies in order
    {
      const size_t nch = (bytes + chunk - 1) / chunk;
      std::vector<std::atomic<int>> done(nch);
      for (auto& d : done) d.store(0);
      std::atomic<size_t> next{0};
      std::atomic<int> go{0};
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++) th.emplace_back([&] {
        while (!go.load(std::memory_order_acquire)) {}
        for (;;) { const size_t c = next.fetch_add(1); if (c >= nch) break; const size_t off = c * chunk, len = std::min(chunk, bytes - off); memcpy(pin + off, src + off, len); done[c].store(1, std::memory_order_release); }
      });
      hipEvent_t ev[3]; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      t0 = now_us(); go.store(1, std::memory_order_release);
      double tmark[3] = {0, 0, 0}; int mk = 0;
      for (size_t c = 0; c < nch; c++) {
        while (!done[c].load(std::memory_order_acquire)) {}
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        CK(hipMemcpyAsync(dev + off, pin + off, len, hipMemcpyHostToDevice, s));
        if (mk < 3 && (c + 1) * 3 >= nch * (size_t)(mk + 1)) { CK(hipEventRecord(ev[mk], s)); mk++; }
      }
      const double t_issued = now_us();
      for (int i = 0; i < 3; i++) { CK(hipEventSynchronize(ev[i])); tmark[i] = now_us(); }
      CK(hipStreamSynchronize(s)); t1 = now_us();
      for (auto& x : th) x.join();
      printf("staged (%zu MB chunks, %d threads): all issued %7.1f us, thirds on device at %7.1f / %7.1f / %7.1f us, done %7.1f us  %.1f GB/s\n",
             chunk >> 20, T, t_issued - t0, tmark[0] - t0, tmark[1] - t0, tmark[2] - t0, t1 - t0, bytes / (t1 - t0) / 1e3);
    }
  }
  return 0;
}
