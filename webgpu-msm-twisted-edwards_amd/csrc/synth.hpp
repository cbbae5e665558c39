// synth.hpp -- deterministic synthetic inputs for the harness side of the path (host code only).
// The reference's harness feeds compute_msm either the ZPrize files or random inputs it generates itself
// (ui/AllBenchmarks.tsx:99-131: one fixed point replicated n times + random scalars; reference/webgpu/utils.ts:81-88,
// 118-124 for the scalar distribution).  bench.py, full_benchmarks.py and the tests need the same thing without the
// files: scalars = 256 random bits reduced mod p (splitmix64 stream), points P_i = (a + i*b)*G built as a chain
// P_0 = a*G, P_{i+1} = P_i + b*G and normalised with one batched inversion -- n DISTINCT subgroup points, which the
// harness's replicated point is not.
#pragma once
#include <vector>
#include <thread>
#include "host_tail.hpp"
#include "host_tail377.hpp"

namespace te_host {

static inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline Fe rand_mod_p(uint64_t& s) {          // plain integer in [0, p)
  Fe r; for (int i = 0; i < 4; i++) r.l[i] = splitmix64(s);
  return canon(r);
}
static const Fe R2_M = {{0x25d577bab861857bULL, 0xcc2c27b58860591fULL, 0xa7cc008fe5dc8593ULL, 0x011fdae7eff1c939ULL}};   // R^2 mod p
static inline Fe to_mont(const Fe& raw) { return mul(raw, R2_M); }
static inline Fe from_mont(const Fe& m) { const Fe one_raw = {{1, 0, 0, 0}}; return mul(m, one_raw); }

static inline void synth_scalars(uint64_t seed, uint64_t n, uint8_t* out) {
  uint64_t s = seed;
  for (uint64_t i = 0; i < n; i++) { const Fe r = rand_mod_p(s); memcpy(out + 32 * i, r.l, 32); }
}

// k * P, k a plain integer below p (double-and-add over its 253 bits)
static inline Pt pmul(const Pt& p, const Fe& k, const Fe& k2d) {
  Pt acc = identity();
  for (int i = 252; i >= 0; i--) {
    acc = pdbl(acc);
    if ((k.l[i >> 6] >> (i & 63)) & 1) acc = padd(acc, p, k2d);
  }
  return acc;
}

static inline void synth_points(uint64_t seed, uint64_t n, uint8_t* out) {
  if (n == 0) return;
  // generator of the prime-order subgroup, reference/utils/FieldMath.ts:108-109
  const Fe gx = {{0x137e82844bbe49c5ULL, 0xe7608833a9dd83f3ULL, 0x16b294b80d905006ULL, 0x036824eb02475007ULL}};
  const Fe gy = {{0xd50dce7d8bcda9d4ULL, 0x7f6758f4c08bc255ULL, 0x37c0a81e810abce5ULL, 0x11b1d8d5c1d897a3ULL}};
  const Fe d2 = {{2 * 3021, 0, 0, 0}};
  const Fe k2d = to_mont(d2);
  Pt g; g.x = to_mont(gx); g.y = to_mont(gy); g.z = ONE_M; g.t = mul(g.x, g.y);
  uint64_t s = seed ^ 0xA5A5A5A55A5A5A5AULL;
  const Fe a = rand_mod_p(s), b = rand_mod_p(s);
  const Pt q = pmul(g, b, k2d);
  std::vector<Pt> pts(n);
  pts[0] = pmul(g, a, k2d);
  for (uint64_t i = 1; i < n; i++) pts[i] = padd(pts[i - 1], q, k2d);
  std::vector<Fe> pre(n);
  Fe acc = ONE_M;
  for (uint64_t i = 0; i < n; i++) { pre[i] = acc; acc = mul(acc, pts[i].z); }
  Fe iv = inv(acc);
  for (uint64_t i = n; i-- > 0;) {
    const Fe zi = mul(iv, pre[i]);
    iv = mul(iv, pts[i].z);
    const Fe x = from_mont(mul(pts[i].x, zi)), y = from_mont(mul(pts[i].y, zi));
    memcpy(out + 64 * i, x.l, 32); memcpy(out + 64 * i + 32, y.l, 32);
  }
}

// Set (R) of SURVEY.md 8d: P_i = a_i * G with a_i seeded-random (the splitmix64 stream of seed ^ 0x5A5A..., 256 bits reduced
// mod p, as the scalars are) -- n INDEPENDENT subgroup points, what "random points" means for a benchmark of a general MSM;
// the chain above is an arithmetic progression.  Fixed-base method: a table of v * 2^(8 j) * G (j < 32, v < 256) built once,
// then 31 additions per point; points are independent, so they are spread over the host's threads (2^20 points: ~7 s of one
// core).  The a_i come from ONE sequential stream (thread t re-derives its share by skipping), so the output does not depend
// on the thread count.
static inline void synth_points_random(uint64_t seed, uint64_t n, uint8_t* out, unsigned threads = 0) {
  if (n == 0) return;
  const Fe gx = {{0x137e82844bbe49c5ULL, 0xe7608833a9dd83f3ULL, 0x16b294b80d905006ULL, 0x036824eb02475007ULL}};
  const Fe gy = {{0xd50dce7d8bcda9d4ULL, 0x7f6758f4c08bc255ULL, 0x37c0a81e810abce5ULL, 0x11b1d8d5c1d897a3ULL}};
  const Fe d2 = {{2 * 3021, 0, 0, 0}};
  const Fe k2d = to_mont(d2);
  Pt g; g.x = to_mont(gx); g.y = to_mont(gy); g.z = ONE_M; g.t = mul(g.x, g.y);
  std::vector<Pt> table(32 * 256);
  {
    Pt base = g;                                    // 2^(8 j) * G
    for (int j = 0; j < 32; j++) {
      table[(size_t)j * 256] = identity();
      for (int v = 1; v < 256; v++) table[(size_t)j * 256 + v] = padd(table[(size_t)j * 256 + v - 1], base, k2d);
      base = padd(table[(size_t)j * 256 + 255], base, k2d);
    }
  }
  std::vector<Fe> ks(n);
  { uint64_t s = seed ^ 0x5A5A5A5AA5A5A5A5ULL; for (uint64_t i = 0; i < n; i++) ks[i] = rand_mod_p(s); }
  std::vector<Pt> pts(n);
  if (!threads) { threads = std::thread::hardware_concurrency(); if (threads > 32) threads = 32; }
  if (threads < 1 || n < 4096) threads = 1;
  auto work = [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) {
      const uint8_t* kb = reinterpret_cast<const uint8_t*>(ks[i].l);
      Pt acc = table[kb[0]];
      for (int j = 1; j < 32; j++) acc = padd(acc, table[(size_t)j * 256 + kb[j]], k2d);
      pts[i] = acc;
    }
  };
  {
    std::vector<std::thread> th;
    const uint64_t per = (n + threads - 1) / threads;
    for (unsigned t = 1; t < threads; t++) { const uint64_t lo = std::min<uint64_t>(n, per * t), hi = std::min<uint64_t>(n, lo + per); if (hi > lo) th.emplace_back(work, lo, hi); }
    work(0, std::min<uint64_t>(n, per));
    for (auto& t : th) t.join();
  }
  // one batched inversion of the z coordinates (a_i = 0 mod l would give the neutral element, z != 0 all the same)
  std::vector<Fe> pre(n);
  Fe acc = ONE_M;
  for (uint64_t i = 0; i < n; i++) { pre[i] = acc; acc = mul(acc, pts[i].z); }
  Fe iv = inv(acc);
  for (uint64_t i = n; i-- > 0;) {
    const Fe zi = mul(iv, pre[i]);
    iv = mul(iv, pts[i].z);
    const Fe x = from_mont(mul(pts[i].x, zi)), y = from_mont(mul(pts[i].y, zi));
    memcpy(out + 64 * i, x.l, 32); memcpy(out + 64 * i + 32, y.l, 32);
  }
}

// the harness's fixed point (ui/AllBenchmarks.tsx:107-109), replicated
static inline void synth_points_fixed(uint64_t n, uint8_t* out) {
  const Fe hx = {{0xd5d3b8c459b4076eULL, 0x1b5799eed0eb02d2ULL, 0x8e051509543ece5eULL, 0x062edc0d88e22612ULL}};
  const Fe hy = {{0xb7969948a31c10c6ULL, 0x96933076821a1429ULL, 0x803b36417a89f1a0ULL, 0x11fbd6ecd3449628ULL}};
  for (uint64_t i = 0; i < n; i++) { memcpy(out + 64 * i, hx.l, 32); memcpy(out + 64 * i + 32, hy.l, 32); }
}

}  // namespace te_host

// the same scheme on BLS12-377 G1: 48-byte scalar records (values < r), 96-byte points P_i = (a + i*b)*G
namespace te377_host {

static inline void synth_scalars(uint64_t seed, uint64_t n, uint8_t* out) {
  uint64_t s = seed;
  for (uint64_t i = 0; i < n; i++) { const te_host::Fe r = te_host::rand_mod_p(s); memset(out + 48 * i, 0, 48); memcpy(out + 48 * i, r.l, 32); }
}
// The harness points are generated on the short-Weierstrass model the wire format uses (the engine's own Edwards form is
// an internal matter): complete projective addition of Renes-Costello-Batina 2016, Algorithm 7, a = 0, b3 = 3; also doubles.
struct SwPt { Fe x, y, z; };     // (X : Y : Z); the point at infinity is (0 : 1 : 0)
static inline Fe mul3(const Fe& a) { return add(add(a, a), a); }
static inline SwPt sw_add(const SwPt& p, const SwPt& q) {
  const Fe t0 = mul(p.x, q.x), t1 = mul(p.y, q.y), t2 = mul(p.z, q.z);
  const Fe t3 = sub(sub(mul(add(p.x, p.y), add(q.x, q.y)), t0), t1);
  const Fe t4 = sub(sub(mul(add(p.y, p.z), add(q.y, q.z)), t1), t2);
  const Fe y3 = mul3(sub(sub(mul(add(p.x, p.z), add(q.x, q.z)), t0), t2));
  const Fe t0x3 = mul3(t0), t2x3 = mul3(t2);
  const Fe z3 = add(t1, t2x3), t1m = sub(t1, t2x3);
  SwPt r;
  r.x = sub(mul(t3, t1m), mul(t4, y3));
  r.y = add(mul(t1m, z3), mul(y3, t0x3));
  r.z = add(mul(z3, t4), mul(t0x3, t3));
  return r;
}
static inline SwPt sw_mul(const SwPt& p, const te_host::Fe& k) {      // k < r (253 bits)
  SwPt acc; memset(&acc, 0, sizeof acc); acc.y = ONE_M;
  for (int i = 252; i >= 0; i--) { acc = sw_add(acc, acc); if ((k.l[i >> 6] >> (i & 63)) & 1) acc = sw_add(acc, p); }
  return acc;
}
static inline void synth_points(uint64_t seed, uint64_t n, uint8_t* out) {
  if (n == 0) return;
  // generator of the order-r subgroup (the standard one of the BLS12-377 specification), plain integers
  const Fe gx = {{0xeab9b16eb21be9efULL, 0xd5481512ffcd394eULL, 0x188282c8bd37cb5cULL, 0x85951e2caa9d41bbULL, 0xc8fc6225bf87ff54ULL, 0x008848defe740a67ULL}};
  const Fe gy = {{0xfd82de55559c8ea6ULL, 0xc2fe3d3634a9591aULL, 0x6d182ad44fb82305ULL, 0xbd7fb348ca3e52d9ULL, 0x1f674f5d30afeec4ULL, 0x01914a69c5102effULL}};
  const Fe R2 = {{0xb786686c9400cd22ULL, 0x0329fcaab00431b1ULL, 0x22a5f11162d6b46dULL, 0xbfdf7d03827dc3acULL, 0x837e92f041790bf9ULL, 0x006dfccb1e914b88ULL}};
  SwPt g; g.x = mul(gx, R2); g.y = mul(gy, R2); g.z = ONE_M;
  uint64_t s = seed ^ 0xA5A5A5A55A5A5A5AULL;
  const te_host::Fe a = te_host::rand_mod_p(s), b = te_host::rand_mod_p(s);
  const SwPt q = sw_mul(g, b);
  std::vector<SwPt> pts(n);
  pts[0] = sw_mul(g, a);
  for (uint64_t i = 1; i < n; i++) pts[i] = sw_add(pts[i - 1], q);
  std::vector<Fe> pre(n);
  Fe acc = ONE_M;
  for (uint64_t i = 0; i < n; i++) { pre[i] = acc; acc = mul(acc, pts[i].z); }      // no infinity in the chain (order r)
  Fe iv = inv(acc);
  Fe one_raw; memset(&one_raw, 0, sizeof one_raw); one_raw.l[0] = 1;
  for (uint64_t i = n; i-- > 0;) {
    const Fe zi = mul(iv, pre[i]);
    iv = mul(iv, pts[i].z);
    const Fe x = mul(mul(pts[i].x, zi), one_raw), y = mul(mul(pts[i].y, zi), one_raw);
    memcpy(out + 96 * i, x.l, 48); memcpy(out + 96 * i + 48, y.l, 48);
  }
}

}  // namespace te377_host
