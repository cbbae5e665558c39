// Host-core speed of the CPU tail's arithmetic (csrc/host_tail.hpp): g++ -O3 -std=c++17 tools/host_tail_bench.cpp -o /tmp/htb && /tmp/htb
#include "../webgpu-msm-twisted-edwards_amd/csrc/host_tail.hpp"
#include <chrono>
#include <stdio.h>
using namespace te_host;
int main() {
  printf("selftest %d, mulx/adcx/adox form in use: %d\n", (int)tail_selftest(), (int)have_adx());
  Pt p = identity(); p.x = ONE_M; p.y = add(ONE_M, ONE_M); p.t = mul(p.x, p.y);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 400000; i++) p = pdbl(p, true);
  auto t1 = std::chrono::steady_clock::now();
  Fe a = p.x, b = p.y;
  for (int i = 0; i < 2000000; i++) { a = mul(a, b); b = mul(b, a); }
  auto t2 = std::chrono::steady_clock::now();
  Fe c = a, d = b;
  for (int i = 0; i < 2000000; i++) { c = mul_c(c, d); d = mul_c(d, c); }
  auto t3 = std::chrono::steady_clock::now();
  uint8_t rows[16 * 720]; memset(rows, 0, sizeof rows);
  for (int w = 0; w < 16; w++) for (int s5 = 0; s5 < 5; s5++) { uint32_t* q = (uint32_t*)(rows + w * 720 + s5 * 144); q[0] = 3 + w; q[9] = 1; q[18] = 1; }   // not curve points: timing only
  uint8_t out[64];
  auto t4 = std::chrono::steady_clock::now();
  for (int i = 0; i < 2000; i++) horner_to_affine(rows, 16, 15, 16, out);
  auto t5 = std::chrono::steady_clock::now();
  {
    // the two accumulators of Horner's rule on the same rows (the product chooses IfmaAcc when the CPU has AVX-512 IFMA)
    auto pts = [&](int w, int slot, auto& emit) { emit(load_point(rows + (size_t)w * 720 + (size_t)slot * 144)); };
    uint8_t o1[64], o2[64];
    auto a0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; i++) horner_with<ScalarAcc>(pts, 16, 15, 16, o1);
    auto a1 = std::chrono::steady_clock::now();
    double ifma_us = -1;
#if defined(__x86_64__)
    if (have_ifma()) {
      for (int i = 0; i < 2000; i++) horner_with<IfmaAcc>(pts, 16, 15, 16, o2);
      ifma_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a1).count() / 2000.0;
    }
#endif
    printf("Horner over 16 x 16-bit rows: scalar accumulator %.1f us, AVX-512 IFMA accumulator %.1f us (results equal: %d)\n",
           std::chrono::duration<double, std::micro>(a1 - a0).count() / 2000.0, ifma_us, ifma_us < 0 ? -1 : (int)(memcmp(o1, o2, 64) == 0));
  }
  printf("doubling %.1f ns   product (dependent chain) %.1f ns, portable form %.1f ns   whole tail (16 windows of 16 bits) %.1f us   (%llu)\n",
         std::chrono::duration<double, std::nano>(t1 - t0).count() / 400000.0, std::chrono::duration<double, std::nano>(t2 - t1).count() / 4000000.0,
         std::chrono::duration<double, std::nano>(t3 - t2).count() / 4000000.0, std::chrono::duration<double, std::micro>(t5 - t4).count() / 2000.0,
         (unsigned long long)(p.x.l[0] ^ a.l[0] ^ c.l[0] ^ out[0]));
}
