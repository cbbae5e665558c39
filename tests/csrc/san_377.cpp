// san_377.cpp -- the BLS12-377 twin of san_te.cpp: fq377.hpp / curve.hpp at 14 limbs through fq377check.cpp's stage emulation,
// the host tail (csrc/host_tail377.hpp) and the input synthesis, against oracle/bls377_oracle.c, under
// -fsanitize=address,undefined (tests/sanitize/Makefile).  The oracle of this curve is unpinned by the reference.
#include "fq377check.cpp"
#include "../../webgpu-msm-twisted-edwards_amd/csrc/host_tail.hpp"
#include "../../webgpu-msm-twisted-edwards_amd/csrc/synth.hpp"
#include <stdio.h>

extern "C" int ora377_msm(const uint8_t* points, const uint8_t* scalars, uint64_t n, int c, int threads, uint8_t out[96]);
extern "C" int ora377_msm_naive(const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out[96]);

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fails++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

int main() {
  CHECK(te377_host::tail_selftest(), "host tail self-test");
  const struct { uint64_t n; int c; } cases[] = {{1, 4}, {40, 7}, {200, 9}, {129, 11}};
  for (const auto& cs : cases) {
    const uint64_t n = cs.n; const int c = cs.c, W = (256 + c - 1) / c;
    std::vector<uint8_t> pts(96 * n), sc(48 * n);
    te377_host::synth_points(31 + n, n, pts.data());
    te377_host::synth_scalars(41 + n, n, sc.data());
    uint8_t want[96], naive[96];
    CHECK(ora377_msm(pts.data(), sc.data(), n, n >= 64 ? 8 : 4, 2, want) == 0, "oracle n=%llu", (unsigned long long)n);
    if (n <= 40) { ora377_msm_naive(pts.data(), sc.data(), n, naive); CHECK(memcmp(want, naive, 96) == 0, "oracle pipeline vs naive"); }
    std::vector<uint8_t> rows((size_t)W * TE377_TAIL_ROW_BYTES, 0);
    CHECK(f377_partial_rows(pts.data(), sc.data(), n, c, 0, 1, rows.data()) == 0, "rows");
    uint8_t out[96];
    te377_host::horner_to_affine(rows.data(), c, c - 1, W, out);
    CHECK(memcmp(out, want, 96) == 0, "tail n=%llu c=%d", (unsigned long long)n, c);
    if (n >= 3) {
      std::vector<std::vector<uint8_t>> sets(3, std::vector<uint8_t>((size_t)W * TE377_TAIL_ROW_BYTES, 0));
      std::vector<const uint8_t*> ptrs;
      for (int s = 0; s < 3; s++) {
        const uint64_t lo = n * (uint64_t)s / 3, hi = n * (uint64_t)(s + 1) / 3;
        CHECK(f377_partial_rows(pts.data() + 96 * lo, sc.data() + 48 * lo, hi - lo, c, 0, 1, sets[(size_t)s].data()) == 0, "slice rows");
        ptrs.push_back(sets[(size_t)s].data());
      }
      te377_host::horner_to_affine_multi(ptrs.data(), 3, c, c - 1, W, out);
      CHECK(memcmp(out, want, 96) == 0, "point shards (on the fly) n=%llu c=%d", (unsigned long long)n, c);
      std::vector<te377_host::Pt> merged((size_t)W * 5); std::vector<uint8_t> present((size_t)W, 0);
      for (int w = 0; w < W; w++) te377_host::merge_window_rows(ptrs.data(), 3, w, merged.data(), present.data());
      te377_host::horner_to_affine_points(merged.data(), present.data(), c, c - 1, W, out);
      CHECK(memcmp(out, want, 96) == 0, "point shards (merged) n=%llu c=%d", (unsigned long long)n, c);
    }
    CHECK(f377_overflow_and_reset() == 0, "column overflow n=%llu c=%d", (unsigned long long)n, c);
    printf("ok n=%llu c=%d\n", (unsigned long long)n, c);
  }
  printf(fails ? "san_377: %d FAILURES\n" : "san_377: all checks passed\n", fails);
  return fails ? 1 : 0;
}
