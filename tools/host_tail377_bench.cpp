#include <stdio.h>
#include <chrono>
#include <vector>
#include "host_tail377.hpp"
using namespace te377_host;
int main() {
  std::vector<uint8_t> rows(16 * 1120, 0);
  for (int w = 0; w < 16; w++) for (int s5 = 0; s5 < 5; s5++) { uint32_t* q = (uint32_t*)(rows.data() + w * 1120 + s5 * 224); q[0] = 3 + w; q[14] = 1; q[28] = 1; }
  uint8_t out[96];
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; i++) horner_to_affine(rows.data(), 16, 15, 16, out);
    auto t1 = std::chrono::steady_clock::now();
    {
      auto pts = [&](int w, int slot, auto& emit) { emit(load_point(rows.data() + (size_t)w * 1120 + (size_t)slot * 224)); };
      uint8_t o1[96], o2[96];
      auto a0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 200; i++) horner_with<ScalarAcc>(pts, 16, 15, 16, o1);
      auto a1 = std::chrono::steady_clock::now();
      double ifma_us = -1;
#if defined(__x86_64__)
      if (have_ifma()) {
        for (int i = 0; i < 200; i++) horner_with<IfmaAcc>(pts, 16, 15, 16, o2);
        ifma_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a1).count() / 200;
      }
#endif
      printf("  selftest %d; scalar accumulator %.1f us, AVX-512 IFMA accumulator %.1f us (equal: %d)\n", (int)tail_selftest(),
             std::chrono::duration<double, std::micro>(a1 - a0).count() / 200, ifma_us, ifma_us < 0 ? -1 : (int)(memcmp(o1, o2, 96) == 0));
    }
    Pt p = identity(); p.x = ONE_M; p.t = ONE_M;
    for (int i = 0; i < 20000; i++) p = pdbl(p);
    auto t2 = std::chrono::steady_clock::now();
    Fe a = p.x, b = p.y;
    for (int i = 0; i < 200000; i++) { a = mul(a, b); b = mul(b, a); }
    auto t3 = std::chrono::steady_clock::now();
    printf("BLS12-377 tail %.1f us, doubling %.1f ns, product %.1f ns (%llu)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 200, std::chrono::duration<double, std::nano>(t2 - t1).count() / 20000, std::chrono::duration<double, std::nano>(t3 - t2).count() / 400000, (unsigned long long)(a.l[0] ^ out[0]));
  }
}
