// Does the AVX-512 accumulator of the host tail slow down the scalar code around it (or itself, when called now and then)?
// g++ -O3 -std=c++17 tools/host_tail_interleave.cpp -o /tmp/hti && /tmp/hti
#include "../webgpu-msm-twisted-edwards_amd/csrc/host_tail.hpp"
#include <chrono>
#include <stdio.h>
#include <vector>
using namespace te_host;
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  uint8_t rows[16 * 720]; memset(rows, 0, sizeof rows);
  for (int w = 0; w < 16; w++) for (int s5 = 0; s5 < 5; s5++) { uint32_t* q = (uint32_t*)(rows + w * 720 + s5 * 144); q[0] = 3 + w; q[9] = 1; q[18] = 1; }
  auto pts = [&](int w, int slot, auto& emit) { emit(load_point(rows + (size_t)w * 720 + (size_t)slot * 144)); };
  std::vector<uint8_t> a(1 << 20, 1), b(1 << 20);
  uint8_t out[64];
  for (int form = 0; form < 2; form++) {
    for (int gap = 0; gap < 3; gap++) {                       // other work between two tails: none / a 1 MB memcpy + scalar products / the same + 200 us sleep-spin
      double t_tail = 0, t_other = 0; Fe x = ONE_M, y = ONE_M;
      for (int i = 0; i < 300; i++) {
        double t0 = now_us();
        if (form == 0) horner_with<ScalarAcc>(pts, 16, 15, 16, out); else horner_with<IfmaAcc>(pts, 16, 15, 16, out);
        double t1 = now_us();
        if (gap >= 1) { memcpy(b.data(), a.data(), a.size()); for (int k = 0; k < 2000; k++) { x = mul(x, y); y = add(y, x); } a[i & 1023] = b[77] ^ (uint8_t)x.l[0]; }
        if (gap == 2) { const double until = now_us() + 200; while (now_us() < until) {} }
        double t2 = now_us();
        t_tail += t1 - t0; t_other += t2 - t1;
      }
      printf("%s accumulator, gap %d: tail %.1f us, other work %.1f us\n", form ? "AVX-512 IFMA" : "scalar      ", gap, t_tail / 300, t_other / 300);
    }
  }
  return out[0] == 255;
}
