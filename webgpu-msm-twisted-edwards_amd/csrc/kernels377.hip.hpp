// kernels377.hip.hpp -- the curve-dependent stages for BLS12-377 G1 (BASELINE config 5).
//
// README.md:279-287 of the reference: the BLS12-377 variant is "structurally identical" to the Twisted-Edwards one and
// differs only in the limbs per coordinate and the group law.  Accordingly everything that does not touch a point --
// digits, the two-level counting sort, segments and their schedule (kernels.hip.hpp) -- is shared; this file holds the
// five stages that do: point conversion, bucket accumulation, recombination of split buckets, marginal sums and the
// weighted sums.  First version: one lane per point everywhere (no four-lane "team" additions, no LDS staging), i.e.
// correct and structurally final, not yet tuned like the Twisted-Edwards kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "curve377.hpp"
#include "kernels.hip.hpp"

namespace te377 {

using te::pnt_slot;                       // 128-byte record slot: x | y (28 limb words) + padding
struct g1p_slot { uint4 q[11]; };         // projective point, 42 limb words in a 176-byte slot (16-byte accesses)
constexpr uint32_t ROW_POINT_WORDS = 3 * NL;   // partial rows are packed: 5 points x 168 bytes = 840 bytes per window

__device__ __forceinline__ g1a load_rec(const pnt_slot* __restrict__ recs, uint32_t entry) {
  const uint4* q = recs[entry & 0x7fffffffu].q;
  uint4 u[7];
#pragma unroll
  for (int j = 0; j < 7; j++) u[j] = q[j];
  const uint32_t w[28] = {u[0].x, u[0].y, u[0].z, u[0].w, u[1].x, u[1].y, u[1].z, u[1].w, u[2].x, u[2].y, u[2].z, u[2].w,
                          u[3].x, u[3].y, u[3].z, u[3].w, u[4].x, u[4].y, u[4].z, u[4].w, u[5].x, u[5].y, u[5].z, u[5].w,
                          u[6].x, u[6].y, u[6].z, u[6].w};
  g1a r;
#pragma unroll
  for (int j = 0; j < NL; j++) { r.x.v[j] = w[j]; r.y.v[j] = w[NL + j]; }
  return r;
}
__device__ __forceinline__ void store_g1p(g1p_slot* dst, const g1p& a) {
  uint32_t w[44];
#pragma unroll
  for (int j = 0; j < NL; j++) { w[j] = a.x.v[j]; w[NL + j] = a.y.v[j]; w[2 * NL + j] = a.z.v[j]; }
  w[42] = 0u; w[43] = 0u;
#pragma unroll
  for (int j = 0; j < 11; j++) dst->q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}
__device__ __forceinline__ g1p load_g1p(const g1p_slot* src) {
  uint32_t w[44];
#pragma unroll
  for (int j = 0; j < 11; j++) { const uint4 u = src->q[j]; w[4 * j] = u.x; w[4 * j + 1] = u.y; w[4 * j + 2] = u.z; w[4 * j + 3] = u.w; }
  g1p a;
#pragma unroll
  for (int j = 0; j < NL; j++) { a.x.v[j] = w[j]; a.y.v[j] = w[NL + j]; a.z.v[j] = w[2 * NL + j]; }
  return a;
}

// K1a: affine (x, y), 48-byte little-endian each -> Montgomery record in a 128-byte slot
__global__ void __launch_bounds__(256) k377_prep_points(const uint4* __restrict__ pts, pnt_slot* __restrict__ recs, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint4 u[6];
#pragma unroll
  for (int j = 0; j < 6; j++) u[j] = pts[6 * (size_t)i + j];
  const uint32_t xw[12] = {u[0].x, u[0].y, u[0].z, u[0].w, u[1].x, u[1].y, u[1].z, u[1].w, u[2].x, u[2].y, u[2].z, u[2].w};
  const uint32_t yw[12] = {u[3].x, u[3].y, u[3].z, u[3].w, u[4].x, u[4].y, u[4].z, u[4].w, u[5].x, u[5].y, u[5].z, u[5].w};
  const g1a r = g1a_from_raw(fq_from_words32(xw), fq_from_words32(yw));
  uint32_t w[32];
#pragma unroll
  for (int j = 0; j < NL; j++) { w[j] = r.x.v[j]; w[NL + j] = r.y.v[j]; }
#pragma unroll
  for (int j = 2 * NL; j < 32; j++) w[j] = 0u;
#pragma unroll
  for (int j = 0; j < 8; j++) recs[i].q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}

// K3: one thread per segment (same schedule tables as k_accumulate), 11-product complete mixed additions
__global__ void __launch_bounds__(256) k377_accumulate(const pnt_slot* __restrict__ recs, const uint32_t* __restrict__ sorted,
                                                       const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_count,
                                                       const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ seg_bucket,
                                                       const uint32_t* __restrict__ seg_lenv, const uint32_t* __restrict__ order,
                                                       const uint32_t* __restrict__ num_segments, g1p_slot* __restrict__ buckets,
                                                       g1p_slot* __restrict__ seg_out, uint32_t n, uint32_t logB, uint32_t seg_len, uint32_t ids) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= (order ? *num_segments : ids)) return;         // see k_accumulate
  const uint32_t sgm = order ? order[gid] : gid;
  const uint32_t g = seg_bucket[sgm];
  if (g == TE_SEG_INVALID) return;
  const uint32_t k = g >> logB;
  const uint32_t part = sgm - seg_base[g];
  const uint32_t cnt = seg_lenv[sgm];
  const uint32_t* lst = sorted + (size_t)k * n + bucket_start[g] + part * seg_len;
  g1p acc = g1_identity();
  for (uint32_t j = 0; j < cnt; j++) {
    const uint32_t e = lst[j];
    acc = g1_madd(acc, g1a_cneg(load_rec(recs, e), (e >> 31) != 0u));
  }
  const bool whole = bucket_count[g] <= seg_len;
  store_g1p(whole ? buckets + g : seg_out + sgm, acc);
}

// sums the parts of every split bucket: one thread per bucket (a giant bucket is a serial chain here)
__global__ void __launch_bounds__(256) k377_seg_combine(const uint32_t* __restrict__ bucket_count, const uint32_t* __restrict__ seg_base,
                                                        const g1p_slot* __restrict__ seg_out, g1p_slot* __restrict__ buckets,
                                                        uint32_t total_buckets, uint32_t seg_len) {
  for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < total_buckets; g += gridDim.x * 256u) {
    const uint32_t cnt = bucket_count[g];
    if (cnt <= seg_len) continue;
    const uint32_t parts = (cnt + seg_len - 1u) / seg_len, s0 = seg_base[g];
    g1p acc = load_g1p(seg_out + s0);
    for (uint32_t p = 1; p < parts; p++) acc = g1_add(acc, load_g1p(seg_out + s0 + p));
    store_g1p(buckets + g, acc);
  }
}

// K4a: marginal sums, one fold level (see k_sum_groups): out[o] = sum_{t<K} in[(outer*K + t)*inner + q], o = outer*inner + q
struct sum_job { const g1p_slot* in; g1p_slot* out; uint32_t n_out, K, inner, in_per_window, out_per_window; };
struct sum_jobs { sum_job j[4]; };
__global__ void __launch_bounds__(256) k377_sum_groups(sum_jobs js, uint32_t nw) {
  const sum_job& j = js.j[blockIdx.y];
  const uint32_t total = j.n_out * nw;
  for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
    const uint32_t k = g / j.n_out, o = g - k * j.n_out;
    const uint32_t outer = o / j.inner, q = o - outer * j.inner;
    const g1p_slot* src = j.in + (size_t)k * j.in_per_window + (size_t)outer * j.K * j.inner + q;
    g1p acc = load_g1p(src);
    for (uint32_t t = 1; t < j.K; t++) acc = g1_add(acc, load_g1p(src + (size_t)t * j.inner));
    store_g1p(j.out + (size_t)k * j.out_per_window + o, acc);
  }
}

// K4b: per window and digit k, over N <= 16 points M[0..N): total = sum_v M[v] (digit 0 only) and
// weighted = sum_v v * M[v] = sum_{v>=1} S_v with suffix sums S_v.  One lane per point: log-step suffix scan, then a
// tree sum, points exchanged through LDS.  grid (4, nw), block 16.  Row written: [T | W0 | W1 | W2 | W3], packed 168-byte points.
struct wsum_jobs { const g1p_slot* in[4]; uint32_t N[4]; };
__global__ void __launch_bounds__(16) k377_weighted_sum(wsum_jobs js, uint32_t* __restrict__ rows, uint32_t row_stride_words) {
  __shared__ g1p_slot lds[16];
  const uint32_t dgt = blockIdx.x, k = blockIdx.y, t = threadIdx.x, N = js.N[dgt];
  g1p mine = t < N ? load_g1p(js.in[dgt] + (size_t)k * N + t) : g1_identity();
  for (uint32_t d = 1; d < N; d <<= 1) {            // inclusive suffix scan
    store_g1p(&lds[t], mine);
    __syncthreads();
    if (t + d < N) mine = g1_add(mine, load_g1p(&lds[t + d]));
    __syncthreads();
  }
  uint32_t* row = rows + (size_t)k * row_stride_words;
  auto write_point = [&](uint32_t slot, const g1p& p) {
    uint32_t* o = row + slot * ROW_POINT_WORDS;
    for (int j = 0; j < NL; j++) { o[j] = p.x.v[j]; o[NL + j] = p.y.v[j]; o[2 * NL + j] = p.z.v[j]; }
  };
  if (t == 0) { if (dgt == 0) write_point(0, mine); mine = g1_identity(); }
  for (uint32_t s = 8; s > 0; s >>= 1) {            // tree sum of S_1..S_{N-1} (slot 0 = identity)
    if (t >= s && t < 2 * s) store_g1p(&lds[t], mine);
    __syncthreads();
    if (t < s && t + s < N) mine = g1_add(mine, load_g1p(&lds[t + s]));
    __syncthreads();
  }
  if (t == 0) write_point(1 + dgt, mine);
}

}  // namespace te377
