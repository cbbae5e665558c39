// san_te.cpp -- AddressSanitizer + UndefinedBehaviorSanitizer self-check of the host-side code of the Twisted-Edwards path:
// the device arithmetic headers compiled for the host (fp.hpp / curve.hpp through fpcheck.cpp's stage emulation), the host
// tail in all its forms (csrc/host_tail.hpp: mulx/adcx, portable, AVX-512 IFMA accumulator, the multi-set merge of the
// multi-device te_msm_run), the input synthesis (csrc/synth.hpp) -- and the C oracle (oracle/te_oracle.c), against which
// every result is compared.  Test infrastructure: built by tests/sanitize/Makefile with -fsanitize=address,undefined.
// SURVEY.md section 5 "race detection / sanitizers" (the reference has none: jest only).
#include "fpcheck.cpp"
#include "../../webgpu-msm-twisted-edwards_amd/csrc/host_tail.hpp"
#include "../../webgpu-msm-twisted-edwards_amd/csrc/synth.hpp"
#include <stdio.h>

extern "C" int ora_msm(const uint8_t* points_xy_le, const uint8_t* scalars_le, uint64_t n, int c, int bpr_mode, int threads, uint8_t out_xy_le[64]);
extern "C" int ora_msm_naive(const uint8_t* points_xy_le, const uint8_t* scalars_le, uint64_t n, uint8_t out_xy_le[64]);

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fails++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

int main() {
  CHECK(te_host::tail_selftest(), "host tail self-test");
  // (window sizes up to 10 bits: the stage emulation adds up 5 W 2^(c-1) points per call, under the sanitizer)
  const struct { uint64_t n; int c; int mode; } cases[] = {{1, 4, 0}, {33, 5, 0}, {300, 9, 0}, {129, 10, 2}, {64, 7, 1}, {500, 8, 0}};
  for (const auto& cs : cases) {
    const uint64_t n = cs.n; const int c = cs.c, W = (256 + c - 1) / c;
    std::vector<uint8_t> pts(64 * n), sc(32 * n);
    if (cs.mode == 1) te_host::synth_points_fixed(n, pts.data());
    else if (cs.mode == 2) te_host::synth_points_random(77 + n, n, pts.data(), 2);
    else te_host::synth_points(77 + n, n, pts.data());
    te_host::synth_scalars(99 + n, n, sc.data());
    if (n > 3) { memset(&sc[0], 0, 32); memset(&sc[32], 0, 32); sc[32] = 1; }          // a zero and a one among them
    uint8_t want[64], naive[64];
    CHECK(ora_msm(pts.data(), sc.data(), n, n >= 64 ? 8 : 4, 1, 2, want) == 0, "oracle n=%llu", (unsigned long long)n);
    if (n <= 64) { ora_msm_naive(pts.data(), sc.data(), n, naive); CHECK(memcmp(want, naive, 64) == 0, "oracle pipeline vs naive n=%llu", (unsigned long long)n); }
    // whole rows -> every form of the tail
    std::vector<uint8_t> rows((size_t)W * 720, 0);
    CHECK(fpc_partial_rows(pts.data(), sc.data(), n, c, 0, 1, rows.data()) == 0, "rows");
    uint8_t out[64];
    te_host::horner_to_affine(rows.data(), c, c - 1, W, out);
    CHECK(memcmp(out, want, 64) == 0, "tail n=%llu c=%d", (unsigned long long)n, c);
    auto one_set = [&](int w, int slot, auto& emit) { emit(te_host::load_point(rows.data() + (size_t)w * 720 + (size_t)slot * 144)); };
    te_host::horner_with<te_host::ScalarAcc>(one_set, c, c - 1, W, out);
    CHECK(memcmp(out, want, 64) == 0, "scalar accumulator n=%llu c=%d", (unsigned long long)n, c);
#if defined(__x86_64__)
    if (te_host::have_ifma()) {
      te_host::horner_with<te_host::IfmaAcc>(one_set, c, c - 1, W, out);
      CHECK(memcmp(out, want, 64) == 0, "IFMA accumulator n=%llu c=%d", (unsigned long long)n, c);
    }
#endif
    // window shards of 3 "ranks", merged (what an all-gather leaves)
    {
      std::vector<uint8_t> merged((size_t)W * 720, 0);
      for (int r = 0; r < 3; r++) {
        std::vector<uint8_t> part((size_t)W * 720, 0);
        CHECK(fpc_partial_rows(pts.data(), sc.data(), n, c, r, 3, part.data()) == 0, "shard rows");
        for (int w = r; w < W; w += 3) memcpy(&merged[(size_t)w * 720], &part[(size_t)w * 720], 720);
      }
      te_host::horner_to_affine(merged.data(), c, c - 1, W, out);
      CHECK(memcmp(out, want, 64) == 0, "window shards n=%llu c=%d", (unsigned long long)n, c);
    }
    // point shards: 1..5 slices (one may be empty), summed on the fly and merged window by window
    for (int S : {1, 2, 5}) {
      if (n < 2) break;
      std::vector<std::vector<uint8_t>> sets((size_t)S, std::vector<uint8_t>((size_t)W * 720, 0));
      std::vector<const uint8_t*> ptrs;
      for (int s = 0; s < S; s++) {
        const uint64_t lo = n * (uint64_t)s / (uint64_t)S, hi = s == 2 ? lo : n * (uint64_t)(s + 1) / (uint64_t)S;      // slice 2 stays empty: its points go nowhere
        if (hi > lo) CHECK(fpc_partial_rows(pts.data() + 64 * lo, sc.data() + 32 * lo, hi - lo, c, 0, 1, sets[(size_t)s].data()) == 0, "slice rows");
        ptrs.push_back(sets[(size_t)s].data());
      }
      if (S >= 3) {                                                // the empty slice's points as a slice of their own
        const uint64_t lo = n * 2 / (uint64_t)S, hi = n * 3 / (uint64_t)S;
        sets.emplace_back((size_t)W * 720, 0);
        if (hi > lo) CHECK(fpc_partial_rows(pts.data() + 64 * lo, sc.data() + 32 * lo, hi - lo, c, 0, 1, sets.back().data()) == 0, "slice rows");
        ptrs.clear(); for (auto& v : sets) ptrs.push_back(v.data());
      }
      const int ns = (int)ptrs.size();
      te_host::horner_to_affine_multi(ptrs.data(), ns, c, c - 1, W, out);
      CHECK(memcmp(out, want, 64) == 0, "point shards (on the fly) n=%llu c=%d S=%d", (unsigned long long)n, c, S);
      std::vector<te_host::Pt> merged((size_t)W * 5); std::vector<uint8_t> present((size_t)W, 0);
      for (int w = 0; w < W; w++) te_host::merge_window_rows(ptrs.data(), ns, w, merged.data(), present.data());
      te_host::horner_to_affine_points(merged.data(), present.data(), c, c - 1, W, out);
      CHECK(memcmp(out, want, 64) == 0, "point shards (merged) n=%llu c=%d S=%d", (unsigned long long)n, c, S);
    }
    CHECK(fpc_bound_violations() == 0, "limb bounds n=%llu c=%d", (unsigned long long)n, c);
    printf("ok n=%llu c=%d\n", (unsigned long long)n, c);
  }
  // a scalar that leaves a final carry is an error of the emulation too
  {
    std::vector<uint8_t> pts(64 * 2), sc(32 * 2, 0xff), rows(16 * 720);
    te_host::synth_points(5, 2, pts.data());
    CHECK(fpc_partial_rows(pts.data(), sc.data(), 2, 16, 0, 1, rows.data()) == -3, "final carry");
  }
  printf(fails ? "san_te: %d FAILURES\n" : "san_te: all checks passed\n", fails);
  return fails ? 1 : 0;
}
