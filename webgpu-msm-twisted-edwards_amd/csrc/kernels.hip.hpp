// kernels.hip.hpp -- the MSM stages as hand-written HIP kernels for gfx950 (wave64, 256 CUs, 160 KB LDS).
//
// Stage map (reference -> here); the reference's kernels are listed in SURVEY.md 2.1:
//   K1 convert_point_coords_and_decompose_scalars (wgsl/cuzk/convert_point_coords_and_decompose_scalars
//      .template.wgsl:37-123)                    -> k_prep_points + k_digits<C>
//   K2 transpose (wgsl/cuzk/transpose.wgsl:32-76; 16 threads in total)
//                                                -> k_hist / k_scan_a / k_scan_b / k_scatter
//                                                   (whole-window histogram held in LDS, one pass)
//   K3 smvp (wgsl/cuzk/smvp.template.wgsl:58-152) -> k_accumulate (7-product mixed additions)
//   K4/K5 bpr stage_1/2 (wgsl/cuzk/bpr.template.wgsl:73-171) + the CPU sum of 4096 points
//      (submission.ts:362-393)                   -> k_sum_groups (row / column marginals) + k_weighted_sum
// Window w of this context is  w = w_first + k * w_step  for local index k (multi-GPU window sharding).
#pragma once
#include <hip/hip_runtime.h>
#include "curve.hpp"

namespace te {

struct digits_params {
  uint32_t half[10];   // sum_w 2^(c*w + c-1) over ALL windows of the decomposition, 9 limbs (+1 zero)
  uint32_t n;
  int num_windows;     // total windows W of the decomposition
  int w_first, w_step, nw_local;
};

// ------------------------------------------------------------------------------------------------
// K1a: affine (x, y) little-endian canonical  ->  Montgomery record ((y-x)/2, (y+x)/2, d*x*y).
__global__ void __launch_bounds__(256) k_prep_points(const uint4* __restrict__ pts, pnt* __restrict__ recs, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint4 a0 = pts[4 * (size_t)i + 0], a1 = pts[4 * (size_t)i + 1];
  const uint4 b0 = pts[4 * (size_t)i + 2], b1 = pts[4 * (size_t)i + 3];
  fp x, y;
  x.v[0] = a0.x; x.v[1] = a0.y; x.v[2] = a0.z; x.v[3] = a0.w; x.v[4] = a1.x; x.v[5] = a1.y; x.v[6] = a1.z; x.v[7] = a1.w;
  y.v[0] = b0.x; y.v[1] = b0.y; y.v[2] = b0.z; y.v[3] = b0.w; y.v[4] = b1.x; y.v[5] = b1.y; y.v[6] = b1.z; y.v[7] = b1.w;
  // to Montgomery form: R2 * x / R.  R2 < p is the bounded operand, x may be any 256-bit value.
  const fp xm = fp_csub<1>(mont_mul(fp_R2(), x));
  const fp ym = fp_csub<1>(mont_mul(fp_R2(), y));
  const pnt r = pnt_from_affine_mont(xm, ym);
  uint4* o = reinterpret_cast<uint4*>(recs + i);
  o[0] = make_uint4(r.hm.v[0], r.hm.v[1], r.hm.v[2], r.hm.v[3]);
  o[1] = make_uint4(r.hm.v[4], r.hm.v[5], r.hm.v[6], r.hm.v[7]);
  o[2] = make_uint4(r.hp.v[0], r.hp.v[1], r.hp.v[2], r.hp.v[3]);
  o[3] = make_uint4(r.hp.v[4], r.hp.v[5], r.hp.v[6], r.hp.v[7]);
  o[4] = make_uint4(r.dt.v[0], r.dt.v[1], r.dt.v[2], r.dt.v[3]);
  o[5] = make_uint4(r.dt.v[4], r.dt.v[5], r.dt.v[6], r.dt.v[7]);
}

// ------------------------------------------------------------------------------------------------
// K1b: signed window digits.  The reference walks the windows with a carry
// (convert_point_coords...wgsl:98-120, miscellaneous/utils.ts:52-95):
//     v = chunk + carry; if v >= 2^(c-1): digit = v - 2^c, carry = 1; stored = digit + 2^(c-1).
// Equivalent closed form used here: stored_w = window w of (s + sum_w 2^(c*w + c-1)) -- adding half a
// window everywhere performs exactly those carries.  A non-zero bit at or above c*W is the
// reference's "final carry is 1" error (utils.ts:80-83) and sets *err.
// digits[k * n + i] (u16) for local window k.
template <int C>
__global__ void __launch_bounds__(256) k_digits(const uint4* __restrict__ scalars, uint16_t* __restrict__ digits,
                                                digits_params prm, uint32_t* __restrict__ err) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= prm.n) return;
  const uint4 s0 = scalars[2 * (size_t)i], s1 = scalars[2 * (size_t)i + 1];
  uint32_t s[10] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, 0u, 0u};
  uint64_t c = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) { c += (uint64_t)s[j] + prm.half[j]; s[j] = (uint32_t)c; c >>= 32; }
  // window extraction with compile-time bit positions (runtime-indexed register arrays would spill)
  constexpr int WMAX = (255 + C) / C + 1;
  bool bad = false;
#pragma unroll
  for (int w = 0; w < WMAX; w++) {
    const int bit = w * C;
    if (bit >= 288) break;
    const int word = bit >> 5, off = bit & 31;
    uint32_t v = s[word] >> off;
    if (off + C > 32 && word + 1 < 10) v |= s[word + 1] << (32 - off);
    v &= (1u << C) - 1u;
    if (w >= prm.num_windows) { bad |= (v != 0u); continue; }
    const int rel = w - prm.w_first;
    if (rel >= 0 && (rel % prm.w_step) == 0) {
      const int k = rel / prm.w_step;
      if (k < prm.nw_local) digits[(size_t)k * prm.n + i] = (uint16_t)v;
    }
  }
  if (bad) atomicOr(err, 1u);
}

// ------------------------------------------------------------------------------------------------
// bucket of a stored digit: digit = stored - B (B = 2^(c-1)); bucket = |digit| - 1 in [0, B);
// weight of bucket j is j + 1; digit 0 contributes nothing (smvp.template.wgsl:128).
__device__ __forceinline__ bool digit_bucket(uint32_t stored, uint32_t B, uint32_t& bucket, uint32_t& neg) {
  const int d = (int)stored - (int)B;
  if (d == 0) return false;
  neg = d < 0 ? 1u : 0u;
  bucket = (uint32_t)(d < 0 ? -d : d) - 1u;
  return true;
}

// K2a: per (chunk, window) histogram of bucket ids, the window's B counters live in LDS.
//   grid (CH, nw_local), block 1024, dynamic LDS = B * 4 bytes.  counts[(k*CH + ch)*B + b].
__global__ void __launch_bounds__(1024) k_hist(const uint16_t* __restrict__ digits, uint32_t* __restrict__ counts,
                                               uint32_t n, uint32_t B, uint32_t chunk_len) {
  extern __shared__ uint32_t lds_u32[];
  const uint32_t ch = blockIdx.x, k = blockIdx.y, CH = gridDim.x;
  for (uint32_t b = threadIdx.x; b < B; b += 1024u) lds_u32[b] = 0u;
  __syncthreads();
  const uint32_t lo = ch * chunk_len, hi = min(n, lo + chunk_len);
  const uint16_t* d = digits + (size_t)k * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024u) {
    uint32_t b, neg;
    if (digit_bucket(d[i], B, b, neg)) atomicAdd(&lds_u32[b], 1u);
  }
  __syncthreads();
  uint32_t* out = counts + ((size_t)k * CH + ch) * B;
  for (uint32_t b = threadIdx.x; b < B; b += 1024u) out[b] = lds_u32[b];
}

// block-wide exclusive scan of one value per thread (blockDim.x <= 1024, multiple of 64 or < 64)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds /*>= 17 words*/, uint32_t& block_total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if ((int)lane >= d) inc += o;
  }
  if (lane == 63u || threadIdx.x == blockDim.x - 1u) lds[wave] = inc;
  __syncthreads();
  const uint32_t nwaves = (blockDim.x + 63u) >> 6;
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (uint32_t w = 0; w < nwaves; w++) { const uint32_t t = lds[w]; lds[w] = run; run += t; }
    lds[16] = run;
  }
  __syncthreads();
  const uint32_t res = lds[wave] + inc - v;
  block_total = lds[16];
  __syncthreads();
  return res;
}

// K2b: for each bucket: exclusive prefix over chunks (written back into counts), bucket_count, and an
// exclusive scan over the buckets of one 1024-bucket segment.  grid (nseg, nw_local), block min(B,1024).
__global__ void __launch_bounds__(1024) k_scan_a(uint32_t* __restrict__ counts, uint32_t* __restrict__ bucket_count,
                                                 uint32_t* __restrict__ local_excl, uint32_t* __restrict__ seg_total,
                                                 uint32_t B, uint32_t CH) {
  __shared__ uint32_t sm[17];
  const uint32_t seg = blockIdx.x, k = blockIdx.y, nseg = gridDim.x;
  const uint32_t b = seg * blockDim.x + threadIdx.x;
  uint32_t total = 0;
  if (b < B) {
    for (uint32_t ch = 0; ch < CH; ch++) {
      uint32_t* p = counts + ((size_t)k * CH + ch) * B + b;
      const uint32_t v = *p; *p = total; total += v;
    }
    bucket_count[(size_t)k * B + b] = total;
  }
  uint32_t bt;
  const uint32_t ex = block_excl_scan(total, sm, bt);
  if (b < B) local_excl[(size_t)k * B + b] = ex;
  if (threadIdx.x == 0) seg_total[k * nseg + seg] = bt;
}

// K2c: bucket_start = (sum of the window's earlier segment totals) + local_excl.
__global__ void __launch_bounds__(1024) k_scan_b(const uint32_t* __restrict__ local_excl, const uint32_t* __restrict__ seg_total,
                                                 uint32_t* __restrict__ bucket_start, uint32_t B) {
  __shared__ uint32_t base_s;
  const uint32_t seg = blockIdx.x, k = blockIdx.y, nseg = gridDim.x;
  if (threadIdx.x == 0) {
    uint32_t base = 0;
    for (uint32_t s = 0; s < seg; s++) base += seg_total[k * nseg + s];
    base_s = base;
  }
  __syncthreads();
  const uint32_t b = seg * blockDim.x + threadIdx.x;
  if (b < B) bucket_start[(size_t)k * B + b] = base_s + local_excl[(size_t)k * B + b];
}

// K2d: scatter point indices into bucket order.  Entry = index | (negative digit ? 1<<31 : 0).
// Order inside a bucket is whatever the LDS atomics give: the group is commutative, so the bucket sum
// (and the final affine point) does not depend on it.
__global__ void __launch_bounds__(1024) k_scatter(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ counts,
                                                  const uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ sorted,
                                                  uint32_t n, uint32_t B, uint32_t chunk_len) {
  extern __shared__ uint32_t lds_u32[];
  const uint32_t ch = blockIdx.x, k = blockIdx.y, CH = gridDim.x;
  const uint32_t* cnt = counts + ((size_t)k * CH + ch) * B;
  const uint32_t* bs = bucket_start + (size_t)k * B;
  for (uint32_t b = threadIdx.x; b < B; b += 1024u) lds_u32[b] = bs[b] + cnt[b];
  __syncthreads();
  const uint32_t lo = ch * chunk_len, hi = min(n, lo + chunk_len);
  const uint16_t* d = digits + (size_t)k * n;
  uint32_t* out = sorted + (size_t)k * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024u) {
    uint32_t b, neg;
    if (digit_bucket(d[i], B, b, neg)) {
      const uint32_t pos = atomicAdd(&lds_u32[b], 1u);
      out[pos] = i | (neg << 31);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Bucket scheduling: the work of bucket accumulation is one thread per bucket; bucket sizes are
// Poisson(n/B), so a wave of 64 natural-order buckets idles ~30 % of its lanes.  k_order_* sort the
// (window, bucket) pairs by descending size (counting sort on min(count, 1023)) so that the 64
// buckets of a wave have (nearly) equal length and the biggest buckets start first.
__global__ void __launch_bounds__(256) k_order_hist(const uint32_t* __restrict__ bucket_count, uint32_t total, uint32_t* __restrict__ size_hist) {
  __shared__ uint32_t h[1024];
  for (uint32_t j = threadIdx.x; j < 1024u; j += 256u) h[j] = 0;
  __syncthreads();
  for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) atomicAdd(&h[min(bucket_count[g], 1023u)], 1u);
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < 1024u; j += 256u) if (h[j]) atomicAdd(&size_hist[j], h[j]);
}
// one block of 1024: size_cursor[s] = number of buckets with size > s  (descending order start)
__global__ void __launch_bounds__(1024) k_order_scan(const uint32_t* __restrict__ size_hist, uint32_t* __restrict__ size_cursor) {
  __shared__ uint32_t sm[17];
  const uint32_t s = 1023u - threadIdx.x;          // thread 0 handles the largest size
  uint32_t bt;
  const uint32_t ex = block_excl_scan(size_hist[s], sm, bt);
  size_cursor[s] = ex;
}
__global__ void __launch_bounds__(256) k_order_scatter(const uint32_t* __restrict__ bucket_count, uint32_t total,
                                                       uint32_t* __restrict__ size_cursor, uint32_t* __restrict__ order) {
  __shared__ uint32_t h[1024];
  __shared__ uint32_t base[1024];
  for (uint32_t j = threadIdx.x; j < 1024u; j += 256u) h[j] = 0;
  __syncthreads();
  // the block owns a contiguous slice so that each thread sees the same elements in both passes
  const uint32_t per = (total + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = min(total, lo + per);
  for (uint32_t g = lo + threadIdx.x; g < hi; g += 256u) atomicAdd(&h[min(bucket_count[g], 1023u)], 1u);
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < 1024u; j += 256u) { base[j] = h[j] ? atomicAdd(&size_cursor[j], h[j]) : 0u; h[j] = 0; }
  __syncthreads();
  for (uint32_t g = lo + threadIdx.x; g < hi; g += 256u) {
    const uint32_t s = min(bucket_count[g], 1023u);
    const uint32_t pos = base[s] + atomicAdd(&h[s], 1u);
    order[pos] = g;
  }
}

// ------------------------------------------------------------------------------------------------
// K3: bucket accumulation, one thread per (window, bucket), scheduled through order[] (or natural
// order when order == nullptr).  The next record is fetched while the current addition runs.
__device__ __forceinline__ pnt load_pnt(const pnt* __restrict__ recs, uint32_t entry) {
  const uint4* q = reinterpret_cast<const uint4*>(recs + (entry & 0x7fffffffu));
  const uint4 a0 = q[0], a1 = q[1], b0 = q[2], b1 = q[3], c0 = q[4], c1 = q[5];
  pnt r;
  r.hm.v[0] = a0.x; r.hm.v[1] = a0.y; r.hm.v[2] = a0.z; r.hm.v[3] = a0.w; r.hm.v[4] = a1.x; r.hm.v[5] = a1.y; r.hm.v[6] = a1.z; r.hm.v[7] = a1.w;
  r.hp.v[0] = b0.x; r.hp.v[1] = b0.y; r.hp.v[2] = b0.z; r.hp.v[3] = b0.w; r.hp.v[4] = b1.x; r.hp.v[5] = b1.y; r.hp.v[6] = b1.z; r.hp.v[7] = b1.w;
  r.dt.v[0] = c0.x; r.dt.v[1] = c0.y; r.dt.v[2] = c0.z; r.dt.v[3] = c0.w; r.dt.v[4] = c1.x; r.dt.v[5] = c1.y; r.dt.v[6] = c1.z; r.dt.v[7] = c1.w;
  return r;
}
__device__ __forceinline__ void store_ete(ete* dst, const ete& a) {
  uint4* o = reinterpret_cast<uint4*>(dst);
  o[0] = make_uint4(a.x.v[0], a.x.v[1], a.x.v[2], a.x.v[3]); o[1] = make_uint4(a.x.v[4], a.x.v[5], a.x.v[6], a.x.v[7]);
  o[2] = make_uint4(a.y.v[0], a.y.v[1], a.y.v[2], a.y.v[3]); o[3] = make_uint4(a.y.v[4], a.y.v[5], a.y.v[6], a.y.v[7]);
  o[4] = make_uint4(a.z.v[0], a.z.v[1], a.z.v[2], a.z.v[3]); o[5] = make_uint4(a.z.v[4], a.z.v[5], a.z.v[6], a.z.v[7]);
  o[6] = make_uint4(a.t.v[0], a.t.v[1], a.t.v[2], a.t.v[3]); o[7] = make_uint4(a.t.v[4], a.t.v[5], a.t.v[6], a.t.v[7]);
}
__device__ __forceinline__ ete load_ete(const ete* src) {
  const uint4* q = reinterpret_cast<const uint4*>(src);
  ete a; uint4 u;
  u = q[0]; a.x.v[0] = u.x; a.x.v[1] = u.y; a.x.v[2] = u.z; a.x.v[3] = u.w; u = q[1]; a.x.v[4] = u.x; a.x.v[5] = u.y; a.x.v[6] = u.z; a.x.v[7] = u.w;
  u = q[2]; a.y.v[0] = u.x; a.y.v[1] = u.y; a.y.v[2] = u.z; a.y.v[3] = u.w; u = q[3]; a.y.v[4] = u.x; a.y.v[5] = u.y; a.y.v[6] = u.z; a.y.v[7] = u.w;
  u = q[4]; a.z.v[0] = u.x; a.z.v[1] = u.y; a.z.v[2] = u.z; a.z.v[3] = u.w; u = q[5]; a.z.v[4] = u.x; a.z.v[5] = u.y; a.z.v[6] = u.z; a.z.v[7] = u.w;
  u = q[6]; a.t.v[0] = u.x; a.t.v[1] = u.y; a.t.v[2] = u.z; a.t.v[3] = u.w; u = q[7]; a.t.v[4] = u.x; a.t.v[5] = u.y; a.t.v[6] = u.z; a.t.v[7] = u.w;
  return a;
}

__global__ void __launch_bounds__(256) k_accumulate(const pnt* __restrict__ recs, const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_count,
                                                    const uint32_t* __restrict__ order, ete* __restrict__ buckets,
                                                    uint32_t n, uint32_t logB, uint32_t total) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= total) return;
  const uint32_t g = order ? order[gid] : gid;      // g = k * B + b
  const uint32_t k = g >> logB;
  const uint32_t cnt = bucket_count[g];
  const uint32_t* lst = sorted + (size_t)k * n + bucket_start[g];
  ete acc = ete_identity();
  if (cnt) {
    uint32_t e = lst[0];
    pnt cur = load_pnt(recs, e);
    for (uint32_t j = 0; j < cnt; j++) {
      const uint32_t e_cur = e;
      pnt nxt = cur;
      if (j + 1 < cnt) { e = lst[j + 1]; nxt = load_pnt(recs, e); }
      acc = ete_madd(acc, pnt_cneg(cur, (e_cur >> 31) != 0u));
      cur = nxt;
    }
  }
  store_ete(buckets + g, acc);
}

// ------------------------------------------------------------------------------------------------
// K4a: marginal sums of the bucket grid.  A window's B buckets form an RH x RL grid, bucket
// j = hi * RL + lo, weight j + 1.  With R_hi = sum_lo B[hi,lo] and C_lo = sum_hi B[hi,lo]:
//     sum_j (j+1) B_j = sum_hi R_hi + RL * sum_hi hi R_hi + sum_lo lo C_lo.
// Each level of k_sum_groups folds K (2 or 4) elements per thread with full additions, so every lane
// of every wave does the same amount of work (the earlier LDS tree left 3/4 of the lanes idle):
//     out[o] = sum_{k<K} in[(outer*K + k)*inner + q],   o = outer*inner + q.
// rows fold the contiguous (lo) dimension: inner = 1;  columns fold hi: inner = RL.
// Two independent jobs (rows, columns) share one launch: blockIdx.y selects the job.
struct sum_job {
  const ete* in; ete* out;
  uint32_t n_out;      // outputs per window
  uint32_t K, inner;
  uint32_t in_per_window, out_per_window;
};
__global__ void __launch_bounds__(256) k_sum_groups(sum_job j0, sum_job j1, uint32_t nw) {
  const sum_job& j = blockIdx.y == 0 ? j0 : j1;
  const uint32_t total = j.n_out * nw;
  for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
    const uint32_t k = g / j.n_out, o = g - k * j.n_out;
    const uint32_t outer = o / j.inner, q = o - outer * j.inner;
    const ete* src = j.in + (size_t)k * j.in_per_window + (size_t)outer * j.K * j.inner + q;
    ete acc = load_ete(src);
    for (uint32_t t = 1; t < j.K; t++) acc = ete_add(acc, load_ete(src + (size_t)t * j.inner));
    store_ete(j.out + (size_t)k * j.out_per_window + o, acc);
  }
}

// K4b: weighted sums over N points E_0..E_{N-1} (N a power of two <= blockDim.x):
//   total = sum_v E_v,  weighted = sum_v v * E_v = sum_{v >= 1} S_v,  S_v = sum_{u >= v} E_u.
// Suffix sums by a log-step scan in LDS, then a tree sum of S_1..S_{N-1}.  blockIdx.x selects the
// problem (0: the RH row sums -> total + weighted, 1: the RL column sums -> weighted), blockIdx.y the
// window; threads >= N hold the identity.  dynamic LDS = blockDim.x * 128 B.
struct wsum_job { const ete* in; ete* out_total; ete* out_weighted; uint32_t N; };
__global__ void __launch_bounds__(256) k_weighted_sum(wsum_job j0, wsum_job j1, uint32_t out_stride) {
  extern __shared__ uint4 lds_u4[];
  ete* sm = reinterpret_cast<ete*>(lds_u4);
  const wsum_job& j = blockIdx.x == 0 ? j0 : j1;
  const uint32_t k = blockIdx.y, t = threadIdx.x, T = blockDim.x, N = j.N;
  ete mine = t < N ? load_ete(j.in + (size_t)k * N + t) : ete_identity();
  for (uint32_t d = 1; d < N; d <<= 1) {            // inclusive suffix scan
    store_ete(&sm[t], mine);
    __syncthreads();
    if (t + d < N) mine = ete_add(mine, load_ete(&sm[t + d]));
    __syncthreads();
  }
  if (t == 0) { if (j.out_total) store_ete(j.out_total + (size_t)k * out_stride, mine); mine = ete_identity(); }
  for (uint32_t s = T >> 1; s > 0; s >>= 1) {       // tree sum of S_1..S_{N-1} (slot 0 = identity)
    if (t >= s && t < 2 * s) store_ete(&sm[t], mine);
    __syncthreads();
    if (t < s && t + s < N) mine = ete_add(mine, load_ete(&sm[t + s]));
    __syncthreads();
  }
  if (t == 0) store_ete(j.out_weighted + (size_t)k * out_stride, mine);
}

}  // namespace te
