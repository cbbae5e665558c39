// kernels.hip.hpp -- the MSM stages as hand-written HIP kernels for gfx950 (wave64, 256 CUs, 160 KB LDS).
//
// Stage map (reference -> here); the reference's kernels are listed in SURVEY.md 2.1:
//   K1 convert_point_coords_and_decompose_scalars (wgsl/cuzk/convert_point_coords_and_decompose_scalars
//      .template.wgsl:37-123)                    -> k_prep_points + k_digits<C>
//   K2 transpose (wgsl/cuzk/transpose.wgsl:32-76; 16 threads in total)
//                                                -> (histogram in k_digits) k_part_scatter / k_l2_local / k_l2_place_order
//                                                   (two-level counting sort, all stores coalesced; segment plan and schedule ride along)
//   K3 smvp (wgsl/cuzk/smvp.template.wgsl:58-152) -> k_accumulate (7-product mixed additions)
//   K4/K5 bpr stage_1/2 (wgsl/cuzk/bpr.template.wgsl:73-171) + the CPU sum of 4096 points
//      (submission.ts:362-393)                   -> k_sum_groups[_team] (fold levels) + k_reduce_tail (digit marginals, weighted sums)
// Window w of this context is  w = w_first + k * w_step  for local index k (multi-GPU window sharding).
#pragma once
#include <hip/hip_runtime.h>
#include "curve.hpp"

namespace te {

// A launch sequence can carry up to TE_BATCH_MAX MSMs of the same n (te_msm_partial_device_batch): one launch of k_digits
// and of the record conversion covers all of them (blockIdx.y = MSM / record slab), with the input pointers in a table.
#define TE_BATCH_MAX 8
struct batch_ptrs { const uint4* p[TE_BATCH_MAX]; };
// record slab of MSM m: MSMs of a call that name the SAME point buffer share one conversion (slab = first MSM naming it)
struct batch_slabs { uint32_t s[TE_BATCH_MAX]; };

struct digits_params {
  uint32_t half[10];   // signed digits: sum_w 2^(c*w + c-1) over ALL windows of the decomposition, 9 limbs (+1 zero); unsigned: 0
  uint32_t zero_digit; // stored code of digit 0: 2^(c-1) signed, 0 unsigned
  uint32_t sc_stride;  // scalar record size in 16-byte units: 2 (32 bytes) or 3 (48-byte records, upper 16 bytes must be zero)
  uint32_t n, nst;     // points, digit-row stride (n rounded up to a multiple of 8; pad entries hold digit 0)
  int num_windows;     // total windows W of the decomposition
  int w_first, w_step, nw_local;
  // level-1 histogram of the sort, fused into k_digits (see K2 below): partition = bucket >> logS, P partitions per window,
  // chunk = entry index / chunk_len (CH chunks per window)
  uint32_t half_code, logS, P, CH, chunk_len;
};

// ------------------------------------------------------------------------------------------------
// K1a: affine (x, y) little-endian canonical  ->  Montgomery record ((y-x)/2, (y+x)/2, d*x*y) in a 128-byte slot
// (27 limb words + padding: one gather touches exactly one 128-byte line).
struct pnt_slot { uint4 q[8]; };
// Global traffic is staged through LDS so that every load and store instruction covers 1 KB of consecutive addresses:
// with one point per lane straight to memory (16-B pieces at 64-B / 128-B stride) the kernel spent most of its 70 us
// issuing 12 million 16-byte requests.  Records are swizzled in LDS (16-B piece q of record r at piece q ^ (r & 7)) so
// that both the per-record writes and the per-piece reads are conflict-free.
// The first 256 threads of the block convert points [256 blk, 256 blk + 256) of one buffer; lds: 256 * 8 uint4 (32 KB).
// Threads beyond 256 (the fused launch below has 512-thread blocks) only take part in the barriers.
__device__ __forceinline__ void prep_points_block(uint32_t blk, uint4* __restrict__ lds, const uint4* __restrict__ pts, pnt_slot* __restrict__ recs, uint32_t n) {
  const uint32_t t = threadIdx.x, base = blk * 256u;
  const bool active = t < 256u;                          // whole waves: no divergence
  const uint32_t pieces_in = (n - base < 256u ? n - base : 256u) * 4u;     // valid 16-B input pieces of this block
  const uint4* __restrict__ src = pts + (size_t)base * 4u;
  if (active) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t g = (uint32_t)j * 256u + t;
      lds[g] = src[g < pieces_in ? g : 0u];             // clamped: unconditional loads
    }
  }
  __syncthreads();
  uint4 a0 = make_uint4(0u, 0u, 0u, 0u), a1 = a0, b0 = a0, b1 = a0;
  if (active) { a0 = lds[4 * t + 0]; a1 = lds[4 * t + 1]; b0 = lds[4 * t + 2]; b1 = lds[4 * t + 3]; }
  __syncthreads();
  if (active) {
    const uint32_t xw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const uint32_t yw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const pnt r = pnt_from_affine_raw(fp_from_words32(xw), fp_from_words32(yw));
    uint32_t w[32];
#pragma unroll
    for (int j = 0; j < NL; j++) { w[j] = r.hm.v[j]; w[NL + j] = r.hp.v[j]; w[2 * NL + j] = r.dt.v[j]; }
#pragma unroll
    for (int j = 3 * NL; j < 32; j++) w[j] = 0u;
#pragma unroll
    for (int q = 0; q < 8; q++) lds[8 * t + ((uint32_t)q ^ (t & 7u))] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
  }
  __syncthreads();
  if (active) {
    const uint32_t pieces_out = pieces_in * 2u;
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(recs + base);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t g = (uint32_t)j * 256u + t, rr = g >> 3, q = g & 7u;
      if (g < pieces_out) dst[g] = lds[8 * rr + (q ^ (rr & 7u))];
    }
  }
}
__global__ void __launch_bounds__(256) k_prep_points(batch_ptrs in, batch_slabs row_slab, pnt_slot* __restrict__ recs, uint32_t n) {
  __shared__ uint4 lds[256 * 8];                        // 32 KB: first the block's 256 points (16 KB), then its 256 records
  // grid row y converts point buffer y into record slab s = row_slab[y]: slots [s * n, (s + 1) * n)
  prep_points_block(blockIdx.x, lds, in.p[blockIdx.y], recs + (size_t)row_slab.s[blockIdx.y] * n, n);
}

// ------------------------------------------------------------------------------------------------
// K1b: signed window digits.  The reference walks the windows with a carry
// (convert_point_coords...wgsl:98-120, miscellaneous/utils.ts:52-95):
//     v = chunk + carry; if v >= 2^(c-1): digit = v - 2^c, carry = 1; stored = digit + 2^(c-1).
// Equivalent closed form used here: stored_w = window w of (s + sum_w 2^(c*w + c-1)) -- adding half a
// window everywhere performs exactly those carries.  A non-zero bit at or above c*W is the
// reference's "final carry is 1" error (utils.ts:80-83) and sets *err.
// digits[k * n + i] (u16) for local window k.
// Each thread decomposes TWO consecutive scalars per step and stores their digits as one packed u32 per window (nst is even).
// A block covers TE_DIG_BLOCK consecutive entries (it never straddles a level-1 chunk: chunk_len is a multiple of it; 1024 = ONE
// step of two scalars per thread -- measured at n = 2^20 / 2^16: 2048 entries (two steps) 24.3 / 13.8 us, 1024 21.5 / 9.3,
// 4096 28.3 / 23.4; smaller blocks lose to the flush below: 512 entries 30.6 us, 256 entries 54.5 us at 2^20 -- about 6 us per
// million global atomics) and
// also builds the level-1 histogram of the sort for them -- counts1[window][chunk][partition] += ... -- in LDS, flushed with
// one global atomic per non-zero counter.  (The first version re-read all digits in a separate histogram kernel.)
#ifndef TE_DIG_BLOCK
#define TE_DIG_BLOCK 1024u
#endif
#ifndef TE_DIG_THREADS
#define TE_DIG_THREADS 512u
#endif
__device__ __forceinline__ bool digit_bucket(uint32_t stored, uint32_t half, uint32_t& bucket, uint32_t& neg);
// (A/B builds, round 6: -DTE_DIG_WAVES / TE_L2_WAVES name the waves per SIMD the sort kernels are compiled for -- a footprint that fits
// beside three resident accumulation waves per SIMD, 104 VGPRs -- profiles/r06_pipelined_overlap.txt)
#ifndef TE_DIG_WAVES
#define TE_DIG_WAVES 1
#endif
#ifndef TE_L2_WAVES
#define TE_L2_WAVES 3
#endif
template <int C>
__global__ void __launch_bounds__(TE_DIG_THREADS, TE_DIG_WAVES) k_digits(batch_ptrs in, uint16_t* __restrict__ digits,
                                                digits_params prm, uint32_t* __restrict__ err, uint32_t* __restrict__ counts1) {
  __shared__ uint32_t hist[4096];                        // [local window][partition]: nw_local * P <= 4096 for every plan
  // MSM blockIdx.y of the launch sequence: its scalars, its digit rows and level-1 counts [y * nw_local, (y + 1) * nw_local)
  const uint4* __restrict__ scalars = in.p[blockIdx.y];
  digits += (size_t)blockIdx.y * prm.nw_local * prm.nst;
  counts1 += (size_t)blockIdx.y * prm.nw_local * prm.CH * prm.P;
  const uint32_t hn = (uint32_t)prm.nw_local * prm.P;
  for (uint32_t j = threadIdx.x; j < hn; j += TE_DIG_THREADS) hist[j] = 0u;
  __syncthreads();
  uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(digits);
  const uint32_t half_stride = prm.nst >> 1;
  const uint32_t ZERO_DIGIT = prm.zero_digit;
  bool bad = false;
  // every load of the block's two steps is issued before the first digit is extracted (one memory latency, not two)
  constexpr uint32_t STEPS = TE_DIG_BLOCK / (2u * TE_DIG_THREADS);
  const size_t st = prm.sc_stride;
  uint4 ld[STEPS][6];
#pragma unroll
  for (uint32_t step = 0; step < STEPS; step++) {
    const uint32_t i0 = 2u * ((blockIdx.x * STEPS + step) * TE_DIG_THREADS + threadIdx.x);
    const size_t ia = min(i0, prm.n - 1u), ib = min(i0 + 1u, prm.n - 1u);        // clamped: unconditional loads
    ld[step][0] = scalars[st * ia]; ld[step][1] = scalars[st * ia + 1]; ld[step][2] = scalars[st * ib]; ld[step][3] = scalars[st * ib + 1];
    ld[step][4] = ld[step][5] = make_uint4(0u, 0u, 0u, 0u);
    if (st == 3) { ld[step][4] = scalars[st * ia + 2]; ld[step][5] = scalars[st * ib + 2]; }
  }
#pragma unroll
  for (uint32_t step = 0; step < STEPS; step++) {
    const uint32_t pair = (blockIdx.x * STEPS + step) * TE_DIG_THREADS + threadIdx.x, i0 = 2u * pair;
    if (i0 >= prm.nst) break;
    if (i0 >= prm.n) {                                   // padding entries: digit 0
      for (int k = 0; k < prm.nw_local; k++) out[(size_t)k * half_stride + pair] = ZERO_DIGIT | (ZERO_DIGIT << 16);
      continue;
    }
    const bool second = i0 + 1u < prm.n;
    const uint4 a0 = ld[step][0], a1 = ld[step][1], b0 = ld[step][2], b1 = ld[step][3];
    {                                                     // 48-byte records hold values below 2^256
      const uint4 a2 = ld[step][4], b2 = ld[step][5];
      bad |= (a2.x | a2.y | a2.z | a2.w | b2.x | b2.y | b2.z | b2.w) != 0u;
    }
    uint32_t s[2][10] = {{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, 0u, 0u}, {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, 0u, 0u}};
#pragma unroll
    for (int t = 0; t < 2; t++) {
      uint64_t c = 0;
#pragma unroll
      for (int j = 0; j < 9; j++) { c += (uint64_t)s[t][j] + prm.half[j]; s[t][j] = (uint32_t)c; c >>= 32; }
    }
    // window extraction with compile-time bit positions (runtime-indexed register arrays would spill)
    constexpr int WMAX = (255 + C) / C + 1;
    int next = prm.w_first, k = 0;                       // next owned window and its local index
#pragma unroll
    for (int w = 0; w < WMAX; w++) {
      const int bit = w * C;
      if (bit >= 288) break;
      const int word = bit >> 5, off = bit & 31;
      if (w < prm.num_windows && w != next) continue;    // a window of another shard (uniform): nothing to extract
      uint32_t v[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        v[t] = s[t][word] >> off;
        if (off + C > 32 && word + 1 < 10) v[t] |= s[t][word + 1] << (32 - off);
        v[t] &= (1u << C) - 1u;
      }
      if (!second) v[1] = w < prm.num_windows ? ZERO_DIGIT : 0u;
      if (w >= prm.num_windows) { bad |= ((v[0] | v[1]) != 0u); continue; }
      if (w == next) {
        if (k < prm.nw_local) {
#if !defined(TE_EXP_NO_DIGIT_STORE)      // (timing experiment only, profiles/r05_sort_bytes_experiment.txt: what the digit rows cost k_digits)
          out[(size_t)k * half_stride + pair] = v[0] | (v[1] << 16);
#endif
#pragma unroll
          for (int t = 0; t < 2; t++) {
            uint32_t b, neg;
            if (digit_bucket(v[t], prm.half_code, b, neg)) atomicAdd(&hist[(uint32_t)k * prm.P + (b >> prm.logS)], 1u);
          }
        }
        k++; next += prm.w_step;
      }
    }
  }
  if (bad) atomicOr(err, 1u);
  __syncthreads();
  const uint32_t ch = (blockIdx.x * TE_DIG_BLOCK) / prm.chunk_len;
  for (uint32_t j = threadIdx.x; j < hn; j += TE_DIG_THREADS) {
    const uint32_t v = hist[j];
    if (v) { const uint32_t k = j / prm.P, p = j - k * prm.P; atomicAdd(&counts1[((size_t)k * prm.CH + ch) * prm.P + p], v); }
  }
}

// ------------------------------------------------------------------------------------------------
// bucket of a stored digit: digit = stored - half (half = 2^(c-1) for signed digits, 0 for unsigned ones);
// bucket = |digit| - 1 in [0, B); weight of bucket j is j + 1; digit 0 contributes nothing (smvp.template.wgsl:128).
__device__ __forceinline__ bool digit_bucket(uint32_t stored, uint32_t half, uint32_t& bucket, uint32_t& neg) {
  const int d = (int)stored - (int)half;
  if (d == 0) return false;
  neg = d < 0 ? 1u : 0u;
  bucket = (uint32_t)(d < 0 ? -d : d) - 1u;
  return true;
}

// Hand-offs between blocks of ONE launch ("the block that finishes last does the next step").  The L2 caches of the eight XCDs
// are not coherent with each other: a device-scope fence (__threadfence) writes back and invalidates the issuing XCD's whole
// L2 -- measured at ~60 us per fence when every thread of a streaming kernel issues one (a fused count + plan kernel built that
// way took 663 us instead of 20).  So the handed-over DATA travels through device-scope atomic accesses, every wave only waits
// for its own accesses to be acknowledged (s_waitcnt), and the fences of the memory model are issued by the ONE thread that
// counts the block's arrival (handoff_arrive below).
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every memory access this wave has issued is complete (acknowledged by the memory side for device-scope accesses)
__device__ __forceinline__ void wait_own_accesses() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
}
// The arrival itself, by ONE thread of the block, behind a block barrier that follows wait_own_accesses() in every wave;
// returns the number of arrivals before this one.  Two builds:
//   default (round 5)    the memory model's own form: the arriving thread RELEASES at agent scope before the atomic and ACQUIRES
//                        after it; the other waves of the block are ordered through the block barriers on either side
//                        (workgroup-scope synchronisation, cumulative into the agent-scope release).  The fences write back /
//                        invalidate the XCD's L2 (buffer_wbl2 sc1 / buffer_inv sc1) -- but only ONE wave of a block that hands
//                        over issues them, a few hundred times per MSM: k_l2_local 58.0 against 56.6 us at n = 2^20, the
//                        pipelined headline 1 037-1 050 against 1 039-1 044 MSM/s on one box, three rounds back to back
//                        (profiles/r05_handoff_fenced_twin.txt, r05_handoff_fenced_pipelined_ab.txt).  Rounds 1-4 feared
//                        60 us per fence: that was EVERY block of a streaming kernel fencing (__threadfence in each thread).
//   -DTE_HANDOFF_RELAXED the form of rounds 3-4 (libtemsm_relaxed.so, `make relaxed`): a relaxed device-scope atomic and nothing
//                        else.  It stands OUTSIDE the HIP / LLVM memory model -- it relies on gfx950 performing device-scope
//                        atomic accesses at the memory side and on s_waitcnt covering them -- and was correct on ROCm 7.2 /
//                        MI355X by test and soak only.  Kept as the A/B twin (TE_MSM_LIB): only the pathological
//                        many-parts case tells the two apart (its combine 1.10 against 1.49 ms).
// The two hand-offs of the engine and the tests that reach them (tests/test_gpu_handoffs.py runs them against BOTH builds):
//   k_l2_local         pieces of a partition with more than TE_L2_CAP entries: bucket_count by global atomics, part_ticket
//                      counts the pieces, the last one plans the partition -- every canonical scalar set has such partitions in
//                      its top window (test_full_size_2_20, test_wasm_golden_cases); skew makes more: test_witness_like_scalars,
//                      test_giant_buckets (all scalars equal: ONE partition holds every entry of a window)
//   k_seg_combine_all  runs of a giant bucket (more than TE_COMBINE_SMALL parts): store_coord_agent, bucket_cursor counts the runs,
//                      the last one sums them with block_sum_points<COHERENT> -- test_giant_buckets, test_many_parts_per_bucket,
//                      test_witness_like_scalars (bucket 0 of window 0: n / 4 entries)
__device__ __forceinline__ uint32_t handoff_arrive(uint32_t* counter) {
#if defined(TE_HANDOFF_RELAXED)
  return atomicAdd(counter, 1u);
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  const uint32_t v = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return v;
#endif
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

// Exclusive scan of in[0..n) (n <= 512, LDS) into out[0..n) by the FIRST WAVE of the block alone, total to *total: no block
// barrier inside -- the caller puts ONE barrier before (in[] complete) and one after.  (block_excl_scan above costs three.)
__device__ __forceinline__ void wave0_excl_scan(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total) {
  if (threadIdx.x >= 64u) return;
  const uint32_t lane = threadIdx.x, per = (n + 63u) >> 6;           // consecutive values per lane (<= 8)
  uint32_t v[8], sum = 0;
#pragma unroll
  for (uint32_t j = 0; j < 8u; j++) { const uint32_t i = lane * per + j; v[j] = (j < per && i < n) ? in[i] : 0u; sum += v[j]; }
  uint32_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
  uint32_t run = inc - sum;
#pragma unroll
  for (uint32_t j = 0; j < 8u; j++) { const uint32_t i = lane * per + j; if (j < per && i < n) { out[i] = run; run += v[j]; } }
  if (lane == 63u) *total = inc;
}

// ------------------------------------------------------------------------------------------------
// K2: counting sort of point indices by bucket, per window, in two levels so that every global store
// is coalesced.  (The first version scattered 4-byte indices straight to their final position: each
// store became a 32-byte sector write, 535 MB of write traffic for 67 MB of payload --
// profiles/r01_rocprofv3_v2_summary.txt.)
//   level 1  partition = bucket >> logS (P = B/S partitions per window, ~n/P entries each):
//            histogram (in k_digits) -> k_part_scatter (4096-entry tiles sorted by partition in LDS,
//            written out as contiguous runs: u16 key = bucket low bits | sign << 15, u32 index)
//   level 2  k_l2_local: one block per partition counts, plans and places it from one load (pieces of over-long partitions:
//            counted there, placed by k_l2_place_order) -- balanced for any digit distribution.
struct sort_geom {
  uint32_t n, nst;       // entries per window; row stride of digits / part_keys / part_idx (multiple of 8, >= n)
  uint32_t B, logS, S, P, CH, chunk_len;   // chunk_len is a multiple of TE_TILE
  uint32_t half;         // stored code of digit 0 (see digit_bucket)
  uint32_t packed;       // level-1 entries as ONE word: index (23 bits) | bucket low bits << 23 (8) | sign << 31 (n <= 2^23; else u16 key + u32 index)
};
#define TE_TILE 4096u
// level 2 takes a partition in pieces of at most TE_L2_CAP entries (see k_l2_local)
#define TE_L2_CAP 9208u          // entries per piece (multiple of 8); n/P = 8192 at n = 2^20: +11 sigma of its Poisson spread
#define TE_L2_LIST (TE_L2_CAP + 8u)

// All global loads in these kernels are 16 bytes per lane (8 u16 digits / keys, 4 u32 indices): with 2- or
// 4-byte loads the level-2 kernel spent 160 of its 197 us just fetching 100 MB (the memory pipeline is
// paid per load INSTRUCTION and per distinct line it touches).
__device__ __forceinline__ void unpack8(const uint4& v, uint32_t (&d)[8]) {
  d[0] = v.x & 0xffffu; d[1] = v.x >> 16; d[2] = v.y & 0xffffu; d[3] = v.y >> 16;
  d[4] = v.z & 0xffffu; d[5] = v.z >> 16; d[6] = v.w & 0xffffu; d[7] = v.w >> 16;
}

// grid (CH, nw), block 512: tiles of 4096 entries, 8 consecutive entries per thread (one 16-byte load).
// Every block first derives its write offsets from counts1[window][chunk][partition] (k_digits) itself -- partition totals
// over all chunks, their exclusive scan, plus the counts of the earlier chunks: P x CH coalesced loads per block instead of
// a scan kernel (one block per window, 12 us of dependent loads) between two launches.  The block of chunk 0 also writes
// what level 2 needs: part_start / part_count, and seg_part_base, the first segment id of each partition: window k owns the
// ids [k * capW, (k+1) * capW), capW = B + n / seg_len; inside, partition p starts at p * S + sum_{p' < p} floor(count_p' / seg_len)
// -- an upper bound on the segments of the earlier partitions that needs no bucket counts (ids left over are marked invalid
// by k_seg_plan).
struct scatter_args {
  const uint16_t* digits; const uint32_t* counts1; uint16_t* part_keys; uint32_t* part_idx;
  uint32_t *part_start, *part_count, *seg_part_base; uint32_t seg_len, cap_w, nw /* local windows of the launch sequence */; sort_geom g;
  unsigned long long* entries;   // += the non-zero digits of every window (one 64-bit atomic per window): what k_accumulate will gather -- bench.py's roofline
  // fixed-base windows (k_fb_digits): the entry at position i of row k is not point i -- remap[k * nst + i] = table index | sign << 31;
  // nullptr: the ordinary case (entry i of a window is point i).  Needs the general entry form (packed == 0).  row_fill[k]: the entries
  // row k holds -- what lies behind them was never written (the rows are not cleared) and is not read
  const uint32_t* remap; const uint32_t* row_fill;
};
// LDS of one level-1 block in words: one packed word per entry of the tile, four 512-entry tables, scan scratch.
// Packed entry: source slot in the tile (12 bits) | partition << 12 (8 bits: P <= 256) | bucket low bits << 20 (8) | sign << 28 --
// index, key and partition in ONE LDS store and ONE load per entry (round 3 staged a u32 index and two u16 words: three
// stores, and five loads in the copy-out with the two offset tables; LDS instructions per thread and tile 72 -> 32).
#define TE_SCATTER_LDS_WORDS (TE_TILE + 4u * 512u + 17u)
// chunk `ch` of local window `k` (block of 512 threads; lds: TE_SCATTER_LDS_WORDS words)
__device__ __forceinline__ void part_scatter_block(uint32_t ch, uint32_t k, uint32_t* __restrict__ lds, const scatter_args& a) {
  uint32_t* const st = lds;
  uint32_t* const tile_cnt = lds + TE_TILE; uint32_t* const tile_off = tile_cnt + 512; uint32_t* const run_base = tile_cnt + 1024;
  uint32_t* const gdelta = tile_cnt + 1536;            // run_base - tile_off of the current tile: global position of LDS slot s = s + gdelta[partition]
  uint32_t* const sm = tile_cnt + 2048;
  const uint16_t* __restrict__ digits = a.digits; const uint32_t* __restrict__ counts1 = a.counts1;
  uint16_t* __restrict__ part_keys = a.part_keys; uint32_t* __restrict__ part_idx = a.part_idx;
  uint32_t* __restrict__ part_start = a.part_start; uint32_t* __restrict__ part_count = a.part_count; uint32_t* __restrict__ seg_part_base = a.seg_part_base;
  const uint32_t seg_len = a.seg_len, cap_w = a.cap_w; const sort_geom& g = a.g;
  const uint32_t t = threadIdx.x;
  {
    // P is a power of two <= 256: 512 / P threads share a partition's column of counts (independent loads, eight in flight)
    tile_cnt[t] = 0u; tile_off[t] = 0u;
    __syncthreads();
    {
      const uint32_t pp = t & (g.P - 1u), grp = t / g.P, ngrp = 512u / g.P;
      const uint32_t* c = counts1 + (size_t)k * g.CH * g.P + pp;
      uint32_t tot = 0, pre = 0;
#pragma unroll 8
      for (uint32_t cc = grp; cc < g.CH; cc += ngrp) { const uint32_t v = c[(size_t)cc * g.P]; tot += v; pre += cc < ch ? v : 0u; }
      if (tot) atomicAdd(&tile_cnt[pp], tot);
      if (pre) atomicAdd(&tile_off[pp], pre);
    }
    __syncthreads();
    const uint32_t tot = t < g.P ? tile_cnt[t] : 0u, pre = t < g.P ? tile_off[t] : 0u;
    uint32_t bt;
    const uint32_t start = block_excl_scan(tot, sm, bt);
    if (t < g.P) run_base[t] = start + pre;
    if (ch == 0) {
      if (t == 0 && bt) atomicAdd(a.entries, (unsigned long long)bt);
      const uint32_t extra = block_excl_scan(tot / seg_len, sm, bt);
      if (t < g.P) {
        part_start[k * g.P + t] = start; part_count[k * g.P + t] = tot;
        seg_part_base[k * g.P + t] = k * cap_w + t * g.S + extra;
      }
      // overflow pieces of the window (pieces 1.. of the partitions with more than TE_L2_CAP entries), behind the nw * P counts:
      // an extra block of level 2 that has no piece leaves after ONE load instead of a scan over the partition counts
      uint32_t ovt;
      (void)block_excl_scan((t < g.P && tot) ? (tot + TE_L2_CAP - 1u) / TE_L2_CAP - 1u : 0u, sm, ovt);
      if (t == 0) part_count[a.nw * g.P + k] = ovt;
    }
  }
  const uint32_t lo = ch * g.chunk_len, hi = min(a.remap ? min(a.row_fill[k], g.nst) : g.nst, lo + g.chunk_len);
  const uint4* d4 = reinterpret_cast<const uint4*>(digits + (size_t)k * g.nst);
  const uint32_t last8 = (g.nst >> 3) - 1u;
  uint16_t* ok = part_keys + (size_t)k * g.nst;
  uint32_t* oi = part_idx + (size_t)k * g.nst;
  uint4 vnext = d4[min((lo >> 3) + t, last8)];
  for (uint32_t p = t; p < g.P; p += 512u) tile_cnt[p] = 0u;
  __syncthreads();
  // five barriers per tile (the first version had nine: the waves spent 60 % of their time parked): the tile's counters are
  // cleared in the interval that advances run_base, and the scan over the P partition counts is done by the first wave alone
  for (uint32_t base = lo; base < hi; base += TE_TILE) {
    const uint4 vcur = vnext;
    vnext = d4[min(((base + TE_TILE) >> 3) + t, last8)];          // prefetch the next tile during the LDS phases
    uint32_t dd[8], part[8], key[8], rank[8];
    unpack8(vcur, dd);
    const uint32_t i0 = base + t * 8u;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      part[e] = 0xffffffffu;
      uint32_t b, neg;
      if (i0 + (uint32_t)e < hi && digit_bucket(dd[e], g.half, b, neg)) {
        part[e] = b >> g.logS; key[e] = (b & (g.S - 1u)) | (neg << 15);
        rank[e] = atomicAdd(&tile_cnt[part[e]], 1u);
      }
    }
    __syncthreads();
    wave0_excl_scan(tile_cnt, tile_off, g.P, &sm[16]);
    __syncthreads();
    const uint32_t tile_total = sm[16];
    if (t < g.P) gdelta[t] = run_base[t] - tile_off[t];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if (part[e] != 0xffffffffu)
        st[tile_off[part[e]] + rank[e]] = (t * 8u + (uint32_t)e) | (part[e] << 12) | ((key[e] & 0xffu) << 20) | ((key[e] >> 15) << 28);
    }
    __syncthreads();
    if (g.packed) {                                       // uniform: key and index leave as one word (4 bytes per entry instead of 6)
      for (uint32_t s = t; s < tile_total; s += 512u) {
        const uint32_t w = st[s];
        oi[s + gdelta[(w >> 12) & 0xffu]] = (base + (w & 0xfffu)) | (((w >> 20) & 0xffu) << 23) | ((w >> 28) << 31);
      }
    } else if (a.remap) {                                 // fixed-base rows: index and sign come from the row's remap word (a 16 KB window per tile)
      const uint32_t* __restrict__ rm = a.remap + (size_t)k * g.nst + base;
      for (uint32_t s = t; s < tile_total; s += 512u) {
        const uint32_t w = st[s];
        const uint32_t gpos = s + gdelta[(w >> 12) & 0xffu], r = rm[w & 0xfffu];
        ok[gpos] = (uint16_t)(((w >> 20) & 0xffu) | ((r >> 31) << 15)); oi[gpos] = r & 0x7fffffffu;
      }
    } else {
      for (uint32_t s = t; s < tile_total; s += 512u) {
        const uint32_t w = st[s];
        const uint32_t gpos = s + gdelta[(w >> 12) & 0xffu];
        ok[gpos] = (uint16_t)(((w >> 20) & 0xffu) | ((w >> 28) << 15)); oi[gpos] = base + (w & 0xfffu);
      }
    }
    __syncthreads();
    for (uint32_t p = t; p < g.P; p += 512u) { run_base[p] += tile_cnt[p]; tile_cnt[p] = 0u; }
    __syncthreads();
  }
}

// grid (CH, nw), block 512
__global__ void __launch_bounds__(512) k_part_scatter(scatter_args a) {
  __shared__ uint32_t lds[TE_SCATTER_LDS_WORDS];
  part_scatter_block(blockIdx.x, blockIdx.y, lds, a);
}

// Level 1 of the sort and the record conversion in ONE launch (device-resident inputs): neither needs the other, the
// scatter is bound by its LDS rounds and barriers (2.3 TB/s of its 128 MB) and the conversion by memory and four products per
// point -- side by side they fill each other's gaps, back to back they cost 56 + 39 us on one MSM's critical path.
// 1-D grid of scatter_blocks + prep_blocks blocks of 512 threads, the two kinds interleaved evenly (Bresenham); a
// conversion block works with its first 256 threads (the other four waves only meet the barriers).
#ifndef TE_SCATTER_PREP_WAVES
#define TE_SCATTER_PREP_WAVES 8      // 64 VGPRs, nothing spilled: four 512-thread blocks per CU (80 VGPRs, three blocks: 81.1 -> 77.9 us at n = 2^20)
#endif
__global__ void __launch_bounds__(512, TE_SCATTER_PREP_WAVES) k_part_scatter_prep(scatter_args a, uint32_t scatter_blocks, batch_ptrs in, batch_slabs row_slab,
                                                           pnt_slot* __restrict__ recs, uint32_t n, uint32_t prep_blocks_per_row, uint32_t prep_blocks) {
  __shared__ uint4 lds4[(TE_SCATTER_LDS_WORDS + 3u) / 4u > 256u * 8u ? (TE_SCATTER_LDS_WORDS + 3u) / 4u : 256u * 8u];
  const uint64_t tot = (uint64_t)scatter_blocks + prep_blocks, b = blockIdx.x;
  const uint32_t s_before = (uint32_t)(b * scatter_blocks / tot), s_after = (uint32_t)((b + 1u) * scatter_blocks / tot);
  if (s_after > s_before) {                               // this block is scatter block number s_before
    part_scatter_block(s_before % a.g.CH, s_before / a.g.CH, reinterpret_cast<uint32_t*>(lds4), a);
  } else {
    const uint32_t pb = (uint32_t)b - s_before, row = pb / prep_blocks_per_row, blk = pb - row * prep_blocks_per_row;
    prep_points_block(blk, lds4, in.p[row], recs + (size_t)row_slab.s[row] * n, n);
  }
}

// level 2: one block per PARTITION of a window sorts it by bucket -- count, segment plan and placement in ONE pass over the
// partition's ~n/P entries, held in registers from the single load: no global atomics, no second read of the keys, no launch
// boundary between count, plan and placement (rounds 1-3: three launches -- k_l2_count / k_seg_plan / k_l2_place, blocks taking
// fixed-size slices of the level-1 output, every partition cut in two by a slice boundary, ranges reserved with a returning
// global atomic per touched bucket).  The sorted run of a partition occupies the same range [part_start, part_start + count)
// as its level-1 run, in bucket order: the copy-out is one coalesced stream.
// Work stays balanced for any digit distribution: a block takes at most TE_L2_CAP entries.  A partition with more (the top
// window of a 253-bit scalar has only ~4.8k of its 32k buckets occupied -- 19 partitions of 55k entries; skewed scalars in
// general) is cut into PIECES of TE_L2_CAP entries, piece 0 for the partition's own block, the others for the "extra" blocks
// behind the P partition blocks of the grid (block P + x takes the x-th overflow piece of the window: it finds it by scanning
// the P partition counts).  Pieces of such a partition are counted into bucket_count with global atomics, the block that
// counts the last piece plans the partition (hand-over through device-scope accesses, see ld_agent), and their placement --
// ranges reserved with one returning atomic per touched bucket -- happens in the next launch (k_l2_place_order), beside
// the segment schedule.  With well-spread digits only the top window has such partitions.
//   k_l2_local       : grid (P + X, nw), X = n / TE_L2_CAP + 1 extra blocks
//   k_l2_place_order : grid (order_cols + P + X, nw): segment schedule + placement of the pieces of multi-piece partitions

// loads the 16-byte groups covering entries [a, b) of a row (<= TE_L2_CAP + 8 entries), 5 groups of 8 entries per thread.
// PK: packed level-1 entries (one u32 each: the key lives in the index word, the key array does not exist)
struct piece_regs { uint4 k[5], ia[5], ib[5]; };
template <bool PK>
__device__ __forceinline__ void load_piece(const uint16_t* __restrict__ keys_row, const uint32_t* __restrict__ idx_row, uint32_t a, uint32_t b,
                                           uint32_t t, bool with_idx, piece_regs& r, uint32_t& head, uint32_t& total) {
  const uint32_t a_al = a & ~7u;
  head = a - a_al; total = head + (b - a);
  const uint32_t groups = (total + 7u) >> 3;
  const uint4* k4 = reinterpret_cast<const uint4*>(keys_row + a_al);
  const uint4* i4 = reinterpret_cast<const uint4*>(idx_row + a_al);
#pragma unroll
  for (int c = 0; c < 5; c++) {
    const uint32_t gi = min((uint32_t)c * 256u + t, groups - 1u);
    if (!PK) r.k[c] = k4[gi];
    if (PK || with_idx) { r.ia[c] = i4[2 * gi]; r.ib[c] = i4[2 * gi + 1]; }
  }
}
// the 8 entries of group c as (key = bucket low bits | sign << 15, index)
template <bool PK> __device__ __forceinline__ void piece_group(const piece_regs& r, int c, uint32_t (&kv)[8], uint32_t (&iv)[8]) {
  const uint32_t w[8] = {r.ia[c].x, r.ia[c].y, r.ia[c].z, r.ia[c].w, r.ib[c].x, r.ib[c].y, r.ib[c].z, r.ib[c].w};
  if (PK) {
#pragma unroll
    for (int e = 0; e < 8; e++) { kv[e] = ((w[e] >> 23) & 0xffu) | ((w[e] >> 31) << 15); iv[e] = w[e] & 0x7fffffu; }
  } else {
    unpack8(r.k[c], kv);
#pragma unroll
    for (int e = 0; e < 8; e++) iv[e] = w[e];
  }
}
static_assert(5u * 256u * 8u >= TE_L2_CAP + 8u, "five groups per thread hold a piece");

// Which piece does block bx of a window's grid row work on?  bx < P: piece 0 of partition bx.  bx >= P: the (bx - P)-th overflow
// piece of the window -- pieces 1.. of the partitions with more than TE_L2_CAP entries, in partition order; false when there
// are fewer (overflow_pieces = their number in this window).  All 256 threads call (block scan inside); pj: 2 words of LDS, sm: 17.
__device__ __forceinline__ bool l2_piece_of_block(uint32_t bx, const uint32_t* __restrict__ pc, uint32_t P, uint32_t overflow_pieces, uint32_t* sm, uint32_t* pj, uint32_t& p, uint32_t& j) {
  if (bx < P) { p = bx; j = 0u; return true; }
  const uint32_t t = threadIdx.x, x = bx - P;
  if (x >= overflow_pieces) return false;                    // uniform (the window's count, written by k_part_scatter's chunk-0 block)
  const uint32_t c = t < P ? pc[t] : 0u;
  const uint32_t ov = c ? (c + TE_L2_CAP - 1u) / TE_L2_CAP - 1u : 0u;
  uint32_t tot;
  const uint32_t ex = block_excl_scan(ov, sm, tot);
  if (x >= tot) return false;                              // uniform
  if (x >= ex && x < ex + ov) { pj[0] = t; pj[1] = x - ex + 1u; }
  __syncthreads();
  p = pj[0]; j = pj[1];
  __syncthreads();
  return true;
}

// ------------------------------------------------------------------------------------------------
// Work scheduling.  The unit of work of bucket accumulation is a SEGMENT: at most seg_len consecutive entries of one
// bucket, one thread each.  Two reasons:
//  * bucket sizes are Poisson(n/B): a wave of 64 natural-order buckets idles ~30 % of its lanes, so segments are
//    sorted by descending length (counting sort on the length) and a wave gets 64 segments of equal length;
//  * a bucket that is much larger than the rest would otherwise be ONE thread's serial chain and set the kernel's
//    duration on its own -- every 253-bit scalar does this: its top window has only ~4.8k occupied buckets of ~219
//    entries (1.13 ms of serial additions against ~1 ms for everything else); skewed scalars do it in general.
// Buckets split into several segments are summed afterwards by k_seg_combine*.
//   seg_plan_block : one block per level-1 partition, one thread per bucket, inside k_l2_local: bucket_start / bucket_cursor
//                    (= part_start + scan of the partition's counts -- no scan across blocks), seg_base (= seg_part_base +
//                    scan of the segments per bucket), the segment records (bucket, length), the histogram of the
//                    lengths and the lists of split buckets.  Segment ids are dense inside a partition; the ids between
//                    a partition's last segment and the next partition's base stay INVALID.
//                    (The first version needed a two-kernel scan over all buckets plus a thread-per-segment kernel with a
//                    binary search: three dependent launches.)
//   k_order_scatter: counting sort of the valid segment ids by descending length; every block scans the 1024-entry
//                    histogram itself.
#define TE_COMBINE_SMALL 16u
// a giant bucket's parts are summed in runs of TE_GIANT_RUN by one block each, then the runs by the block that finishes the last one.
// 256 (1024 until round 4): a block works a run off as 64 quads x (run / 64) serial team additions + a 6-level tree, so a prover's
// witness with a quarter of ones (bucket 0 of window 0: 4096 parts at n = 2^20) took 16 + 6 + 3 levels of ~2.6 us; now 4 + 6 + 5
#define TE_GIANT_RUN 256u
#define TE_SEG_INVALID 0xffffffffu
// The histogram of segment lengths is kept in TE_HIST_COPIES copies (block b adds to copy b mod copies; k_order_scatter sums
// them): 2048 blocks adding to the same ~60 hot addresses cost 28 us of serialised atomics with a single copy.
#define TE_HIST_COPIES 32u
// Segment plan of partition p of local window k by one block of 256 threads, thread t holding the count of bucket t of the
// partition (threads t >= S idle along: S < 256 only for windows of fewer than 9 bits).  Returns the bucket's start inside the
// partition (the exclusive scan of the counts).  L may alias memory the caller uses before and after: the function starts and
// ends with the block in step.
struct plan_args {
  const uint32_t* part_start; const uint32_t* part_count; const uint32_t* seg_part_base;
  uint32_t *bucket_start, *bucket_cursor, *seg_base, *seg_bucket, *seg_lenv, *size_hist, *split_list;
  uint32_t* split_count;    // [0] small, [1] large buckets, [2] large chunks
  uint32_t* chunk_list;     // pairs (bucket, first part)
  uint32_t seg_len, cap_w, chunk_cap;
};
struct plan_lds { uint32_t h[1024]; uint32_t sm[17]; uint32_t giant[256 * 3]; uint32_t n_giant; };
// pstart / pcount / sbase: part_start, part_count and seg_part_base of the partition, loaded by the caller together with its first
// loads (read here, behind the scans, they were one more memory round trip in every block's chain)
__device__ __forceinline__ uint32_t seg_plan_block(uint32_t p, uint32_t k, uint32_t cnt_in, const sort_geom& sg, const plan_args& a, plan_lds& L,
                                                   uint32_t pstart, uint32_t pcount, uint32_t sbase) {
  const uint32_t t = threadIdx.x, P = sg.P, S = sg.S, B = sg.B, seg_len = a.seg_len;
  uint32_t* const h = L.h; uint32_t* const sm = L.sm; uint32_t* const giant = L.giant;
  __syncthreads();                                       // the caller is done with whatever L aliases
  for (uint32_t j = t; j < 1024u; j += 256u) h[j] = 0u;
  if (t == 0) L.n_giant = 0u;
  const bool mine = t < S;
  const uint32_t g = k * B + p * S + (mine ? t : 0u);
  const uint32_t cnt = mine ? cnt_in : 0u;
  const uint32_t nparts = mine ? max(1u, (cnt + seg_len - 1u) / seg_len) : 0u;
  uint32_t bt, bt2;
  const uint32_t ex = block_excl_scan(cnt, sm, bt);
  const uint32_t ex2 = block_excl_scan(nparts, sm, bt2);
  const uint32_t bs = pstart + ex, sb = sbase + ex2;
  if (mine) {
    a.bucket_start[g] = bs; a.bucket_cursor[g] = bs; a.seg_base[g] = sb;
    if (nparts <= TE_COMBINE_SMALL) {
      for (uint32_t part = 0; part < nparts; part++) {
        const uint32_t len = cnt > part * seg_len ? min(seg_len, cnt - part * seg_len) : 0u;
        a.seg_bucket[sb + part] = g; a.seg_lenv[sb + part] = len;
        atomicAdd(&h[min(len, 1023u)], 1u);
      }
      if (nparts > 1u) a.split_list[atomicAdd(&a.split_count[0], 1u)] = g;
    } else {
      const uint32_t j = atomicAdd(&L.n_giant, 1u);
      giant[3 * j] = g; giant[3 * j + 1] = sb; giant[3 * j + 2] = cnt;
      atomicAdd(&a.split_count[1], 1u);
    }
  }
  // ids this partition does not use: [first + segments, first + S + floor(part_count / seg_len))
  {
    // (the last partition also covers the rest of the window's id range, up to (k + 1) * cap_w)
    const uint32_t first = sbase, used = bt2;
    const uint32_t cap = p + 1u == P ? (k + 1u) * a.cap_w - first : S + pcount / seg_len;
    for (uint32_t j = used + t; j < cap; j += 256u) { a.seg_bucket[first + j] = TE_SEG_INVALID; a.seg_lenv[first + j] = TE_SEG_INVALID; }
  }
  __syncthreads();
  // giant buckets: the whole block writes their segment records; one chunk entry per TE_GIANT_RUN parts (k_seg_combine_all)
  const uint32_t ng = L.n_giant;
  for (uint32_t j = 0; j < ng; j++) {
    const uint32_t gg = giant[3 * j], sb0 = giant[3 * j + 1], c0 = giant[3 * j + 2];
    const uint32_t np = (c0 + seg_len - 1u) / seg_len;
    for (uint32_t part = t; part < np; part += 256u) {
      const uint32_t len = min(seg_len, c0 - part * seg_len);
      a.seg_bucket[sb0 + part] = gg; a.seg_lenv[sb0 + part] = len;
      atomicAdd(&h[min(len, 1023u)], 1u);
      if ((part & (TE_GIANT_RUN - 1u)) == 0) { const uint32_t ci = atomicAdd(&a.split_count[2], 1u); if (ci < a.chunk_cap) { a.chunk_list[2 * ci] = gg; a.chunk_list[2 * ci + 1] = part; } }
    }
  }
  __syncthreads();
  uint32_t* my_hist = a.size_hist + ((k * P + p) % TE_HIST_COPIES) * 1024u;
  for (uint32_t j = t; j < 1024u; j += 256u) if (h[j]) atomicAdd(&my_hist[j], h[j]);
  __syncthreads();                                       // L is free again
  return ex;
}

struct l2_args {
  const uint16_t* part_keys; const uint32_t* part_idx; const uint32_t* part_start; const uint32_t* part_count;
  uint32_t* bucket_count; uint32_t* sorted;
  uint32_t* part_ticket;    // [window][partition], zeroed per MSM: pieces of a multi-piece partition counted so far
  sort_geom g; plan_args pa;
};
// LDS of a k_l2_local block, in words: three 256-entry tables, scan scratch, the piece's sorted entries (the plan's tables
// alias them: the plan runs between the count and the placement)
#define TE_L2A_LDS_WORDS (3u * 256u + 32u + TE_L2_LIST)
static_assert(sizeof(plan_lds) <= TE_L2_LIST * 4u, "the plan's LDS fits into the list area");
// grid (P + X, nw), block 256: see "level 2" above
template <bool PK>
__global__ void __launch_bounds__(256, TE_L2_WAVES) k_l2_local(l2_args a) {
  __shared__ uint32_t lds[TE_L2A_LDS_WORDS];
  uint32_t* const cnt_s = lds; uint32_t* const off_s = lds + 256; uint32_t* const sm = lds + 768; uint32_t* const pj = lds + 768 + 17;
  uint32_t* const list = lds + 768 + 32;
  plan_lds& PL = *reinterpret_cast<plan_lds*>(list);
  const sort_geom& g = a.g;
  // Dispatch order = chain length: the pieces of over-long partitions are the longest chains of the launch (count, global atomics,
  // ticket, and the last one plans the partition), and with canonical scalars they all sit in the TOP window -- so the grid is read
  // backwards: last window first, and inside a window the extra blocks before the P partition blocks.  (Blocks are dispatched in
  // grid order; with the pieces at the end of the grid their chain started when everything else was done.)
  const uint32_t k = gridDim.y - 1u - blockIdx.y, t = threadIdx.x;
  const uint32_t extra = gridDim.x - g.P, bx = blockIdx.x < extra ? g.P + blockIdx.x : blockIdx.x - extra;
  const uint32_t* ps = a.part_start + k * g.P; const uint32_t* pc = a.part_count + k * g.P;
  uint32_t p, j;
  if (!l2_piece_of_block(bx, pc, g.P, bx < g.P ? 0u : a.part_count[gridDim.y * g.P + k], sm, pj, p, j)) return;
  const uint32_t cntp = pc[p], pb = ps[p], sbp = a.pa.seg_part_base[k * g.P + p];
  if (cntp == 0u) { (void)seg_plan_block(p, k, 0u, g, a.pa, PL, pb, cntp, sbp); return; }       // an empty partition: its buckets' (empty) segments
  const bool single = cntp <= TE_L2_CAP;                                         // uniform
  const uint32_t a0 = pb + j * TE_L2_CAP, b0 = min(pb + cntp, a0 + TE_L2_CAP);
  const uint16_t* keys_row = a.part_keys + (size_t)k * g.nst; const uint32_t* idx_row = a.part_idx + (size_t)k * g.nst;
  piece_regs r; uint32_t head, total;
  load_piece<PK>(keys_row, idx_row, a0, b0, t, single, r, head, total);      // in flight across the barrier
  cnt_s[t] = 0u;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 5; c++) {
    uint32_t kv[8], iv[8]; piece_group<PK>(r, c, kv, iv);
    const uint32_t e0 = ((uint32_t)c * 256u + t) * 8u;
#pragma unroll
    for (int e = 0; e < 8; e++) { const uint32_t pos = e0 + (uint32_t)e; if (pos >= head && pos < total) atomicAdd(&cnt_s[kv[e] & 0x7fffu], 1u); }
  }
  __syncthreads();
  const uint32_t c0 = cnt_s[t];
  const size_t gb = (size_t)k * g.B + (size_t)p * g.S + t;        // bucket t of the partition (t < S)
  if (!single) {
    // a piece of a multi-piece partition: counts to memory, the block that counts the last piece plans the partition
    if (c0) atomicAdd(&a.bucket_count[gb], c0);
    wait_own_accesses();                                // the atomic adds have been performed before the piece is counted as done
    __syncthreads();
    if (t == 0) pj[2] = (handoff_arrive(&a.part_ticket[k * g.P + p]) + 1u == (cntp + TE_L2_CAP - 1u) / TE_L2_CAP) ? 1u : 0u;
    __syncthreads();
    if (pj[2]) (void)seg_plan_block(p, k, t < g.S ? ld_agent(a.bucket_count + gb) : 0u, g, a.pa, PL, pb, cntp, sbp);      // uniform
    return;
  }
  // the whole partition is in this block's registers: plan it from the counts at hand, then place it
  if (t < g.S) a.bucket_count[gb] = c0;
  const uint32_t ex = seg_plan_block(p, k, c0, g, a.pa, PL, pb, cntp, sbp);
  if (t < g.S) a.pa.bucket_cursor[gb] = pb + ex + c0;   // "everything placed" (the arrival counter of k_seg_combine_all starts from there)
  off_s[t] = ex;                                        // next free LDS slot of bucket t
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 5; c++) {
    uint32_t kv[8], iv[8]; piece_group<PK>(r, c, kv, iv);
    const uint32_t e0 = ((uint32_t)c * 256u + t) * 8u;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const uint32_t pos = e0 + (uint32_t)e;
      if (pos >= head && pos < total) list[atomicAdd(&off_s[kv[e] & 0x7fffu], 1u)] = iv[e] | ((kv[e] >> 15) << 31);
    }
  }
  __syncthreads();
  uint32_t* out = a.sorted + (size_t)k * g.n + pb;      // the partition's sorted run: same range as its level-1 run, in bucket order
  for (uint32_t s = t; s < cntp; s += 256u) out[s] = list[s];
}

// Counting sort of the valid segment ids by descending length: block `ob` of `nob` (256 threads; lds: 2 * 1024 + 17 words);
// `ids` = size of the segment id space (host-known); rel_cursor zeroed per MSM; max_len = largest length a segment can have
// (min(seg_len, 1023), host-known).
// order[] receives the valid segment ids by descending length; block 0 writes their number to *num_segments.
// The block is a chain of dependent memory round trips on one MSM's critical path (20 us as a launch of its own), so it is
// written to have as few of them as possible (round 4: 26 -> see profiles/r04_single_msm_timeline.txt):
//  * the block's slice of seg_lenv is loaded ONCE, all loads in flight together, and kept in registers for both passes
//    (TE_ORDER_REGS entries per thread; a larger slice takes the two-pass loop);
//  * of the four 256-entry chunks of the length histogram only those that can be non-empty are summed and scanned -- with
//    segments of at most 64 entries that is one chunk, not four (each costs 32 loads per thread and a block scan);
//  * the 32 copies of a histogram entry are fetched by one batch of loads.
#define TE_ORDER_REGS 32u
__device__ __forceinline__ void order_scatter_block(uint32_t ob, uint32_t nob, uint32_t* __restrict__ lds, const uint32_t* __restrict__ lenv, uint32_t ids,
                                                    const uint32_t* __restrict__ size_hist, uint32_t* __restrict__ rel_cursor,
                                                    uint32_t* __restrict__ order, uint32_t* __restrict__ num_segments, uint32_t max_len) {
  uint32_t* const h = lds; uint32_t* const base = lds + 1024; uint32_t* const sm = lds + 2048;
  // the block owns a contiguous slice so that each thread sees the same elements in both passes
  const uint32_t per = (ids + nob - 1) / nob;
  const uint32_t lo = min(ids, ob * per), hi = min(ids, lo + per);
  const bool in_regs = per <= TE_ORDER_REGS * 256u;                  // uniform
  uint32_t lv[TE_ORDER_REGS];
  if (in_regs) {
#pragma unroll
    for (uint32_t j = 0; j < TE_ORDER_REGS; j++) { const uint32_t g = lo + j * 256u + threadIdx.x; lv[j] = g < hi ? lenv[g] : TE_SEG_INVALID; }
  }
  for (uint32_t j = threadIdx.x; j < 1024u; j += 256u) h[j] = 0;
  // descending start of every length: number of segments longer than s (each block scans the histogram itself)
  {
    uint32_t run = 0;
    for (uint32_t c = (1023u - min(max_len, 1023u)) >> 8; c < 4u; c++) {      // chunk c holds the lengths 1023 - 256 c .. 768 - 256 c
      const uint32_t s = 1023u - (c * 256u + threadIdx.x);        // thread 0 of chunk 0 handles the largest size
      uint32_t part[TE_HIST_COPIES];
#pragma unroll
      for (uint32_t r = 0; r < TE_HIST_COPIES; r++) part[r] = size_hist[r * 1024u + s];
      uint32_t cnt_s = 0;
#pragma unroll
      for (uint32_t r = 0; r < TE_HIST_COPIES; r++) cnt_s += part[r];
      uint32_t bt;
      const uint32_t ex = block_excl_scan(cnt_s, sm, bt);
      base[s] = run + ex;
      run += bt;
    }
    if (ob == 0 && threadIdx.x == 0) *num_segments = run;
  }
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (uint32_t j = 0; j < TE_ORDER_REGS; j++) if (lv[j] != TE_SEG_INVALID) atomicAdd(&h[min(lv[j], 1023u)], 1u);
  } else {
    for (uint32_t g = lo + threadIdx.x; g < hi; g += 256u) { const uint32_t l = lenv[g]; if (l != TE_SEG_INVALID) atomicAdd(&h[min(l, 1023u)], 1u); }
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < 1024u; j += 256u) { base[j] += h[j] ? atomicAdd(&rel_cursor[j], h[j]) : 0u; h[j] = 0; }
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (uint32_t j = 0; j < TE_ORDER_REGS; j++) {
      if (lv[j] == TE_SEG_INVALID) continue;
      const uint32_t sz = min(lv[j], 1023u);
      order[base[sz] + atomicAdd(&h[sz], 1u)] = lo + j * 256u + threadIdx.x;
    }
  } else {
    for (uint32_t g = lo + threadIdx.x; g < hi; g += 256u) {
      const uint32_t l = lenv[g];
      if (l == TE_SEG_INVALID) continue;
      const uint32_t sz = min(l, 1023u);
      order[base[sz] + atomicAdd(&h[sz], 1u)] = g;
    }
  }
}

// Placement of ONE piece [a0, b0) of a multi-piece partition p of local window k (block of 256 threads; lds: TE_PLACE_LDS_WORDS
// words): LDS count, reserve [base, base + c) in every touched bucket with one returning atomicAdd on bucket_cursor, sort the
// piece by bucket in LDS and copy each run to its reserved range.
#define TE_PLACE_LDS_WORDS (4u * 256u + TE_L2_LIST + TE_L2_LIST / 4u + 2u + 17u)
template <bool PK>
__device__ __forceinline__ void l2_place_piece(uint32_t p, uint32_t k, uint32_t a0, uint32_t b0, uint32_t* __restrict__ lds,
                                               const uint16_t* __restrict__ part_keys, const uint32_t* __restrict__ part_idx,
                                               uint32_t* __restrict__ bucket_cursor, uint32_t* __restrict__ sorted, const sort_geom& g) {
  uint32_t* const cnt_s = lds; uint32_t* const lex_s = lds + 256; uint32_t* const off_s = lds + 512; uint32_t* const gbase_s = lds + 768;
  uint32_t* const list = lds + 1024;
  uint8_t* const list_b = reinterpret_cast<uint8_t*>(list + TE_L2_LIST);
  uint32_t* const sm = list + TE_L2_LIST + TE_L2_LIST / 4u + 2u;
  const uint32_t t = threadIdx.x, len = b0 - a0;
  const uint16_t* keys_row = part_keys + (size_t)k * g.nst; const uint32_t* idx_row = part_idx + (size_t)k * g.nst;
  uint32_t* out = sorted + (size_t)k * g.n;
  cnt_s[t] = 0u;
  __syncthreads();
  piece_regs r; uint32_t head, total;
  load_piece<PK>(keys_row, idx_row, a0, b0, t, true, r, head, total);
#pragma unroll
  for (int c = 0; c < 5; c++) {
    uint32_t kv[8], iv[8]; piece_group<PK>(r, c, kv, iv);
    const uint32_t e0 = ((uint32_t)c * 256u + t) * 8u;
#pragma unroll
    for (int e = 0; e < 8; e++) { const uint32_t pos = e0 + (uint32_t)e; if (pos >= head && pos < total) atomicAdd(&cnt_s[kv[e] & 0x7fffu], 1u); }
  }
  __syncthreads();
  wave0_excl_scan(cnt_s, lex_s, 256u, &sm[16]);        // the first wave scans the 256 bucket counts
  {
    const uint32_t c0 = cnt_s[t];                      // (reserving the output ranges needs only the counts: it overlaps the scan)
    gbase_s[t] = c0 ? atomicAdd(&bucket_cursor[(size_t)k * g.B + (size_t)p * g.S + t], c0) : 0u;
  }
  __syncthreads();
  off_s[t] = lex_s[t];
  gbase_s[t] -= lex_s[t];                               // global position of LDS slot s of bucket t: s + gbase_s[t]
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 5; c++) {
    uint32_t kv[8], iv[8]; piece_group<PK>(r, c, kv, iv);
    const uint32_t e0 = ((uint32_t)c * 256u + t) * 8u;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const uint32_t pos = e0 + (uint32_t)e;
      if (pos >= head && pos < total) {
        const uint32_t b = kv[e] & 0x7fffu;
        const uint32_t slot = atomicAdd(&off_s[b], 1u);
        list[slot] = iv[e] | ((kv[e] >> 15) << 31); list_b[slot] = (uint8_t)b;
      }
    }
  }
  __syncthreads();
  for (uint32_t s = t; s < len; s += 256u) out[gbase_s[list_b[s]] + s] = list[s];
}

// The segment schedule and the placement of the multi-piece partitions in ONE launch: both need the plans of k_l2_local, neither
// needs the other, and the schedule is latency-bound (a few dependent rounds over 3 MB: 20 us as a launch of its own).
// grid (order_cols + P + X, nw), block 256: blocks with blockIdx.x < order_cols sort segments (order_cols * nw of them,
// dispatched first); the others take the pieces of k_l2_local's grid and leave at once unless theirs belongs to a partition of
// more than TE_L2_CAP entries (with well-spread digits: the top window's ~115 pieces).
struct order_args { const uint32_t* lenv; uint32_t ids; const uint32_t* size_hist; uint32_t* rel_cursor; uint32_t* order; uint32_t* num_segments; uint32_t order_cols, max_len; };
template <bool PK>
__global__ void __launch_bounds__(256, TE_L2_WAVES) k_l2_place_order(const uint16_t* __restrict__ part_keys, const uint32_t* __restrict__ part_idx,
                                                           const uint32_t* __restrict__ part_start, const uint32_t* __restrict__ part_count,
                                                           uint32_t* __restrict__ bucket_cursor, uint32_t* __restrict__ sorted, sort_geom g, order_args oa) {
  __shared__ uint32_t lds[TE_PLACE_LDS_WORDS];
  if (blockIdx.x < oa.order_cols) {
    order_scatter_block(blockIdx.y * oa.order_cols + blockIdx.x, gridDim.y * oa.order_cols, lds, oa.lenv, oa.ids, oa.size_hist, oa.rel_cursor, oa.order, oa.num_segments, oa.max_len);
    return;
  }
  const uint32_t k = gridDim.y - 1u - blockIdx.y;        // as in k_l2_local: the top window's pieces first
  const uint32_t px = blockIdx.x - oa.order_cols, extra = gridDim.x - oa.order_cols - g.P, bx = px < extra ? g.P + px : px - extra;
  const uint32_t* pc = part_count + k * g.P;
  uint32_t p, j;
  if (!l2_piece_of_block(bx, pc, g.P, bx < g.P ? 0u : part_count[gridDim.y * g.P + k], lds, lds + 32, p, j)) return;
  const uint32_t cntp = pc[p], pb = part_start[k * g.P + p];
  if (cntp <= TE_L2_CAP) return;                         // sorted by its own block of k_l2_local (uniform)
  const uint32_t a0 = pb + j * TE_L2_CAP;
  l2_place_piece<PK>(p, k, a0, min(pb + cntp, a0 + TE_L2_CAP), lds, part_keys, part_idx, bucket_cursor, sorted, g);
}

// ================================================================================================
// From here on everything handles curve points and is written once for both base fields: N = 9 limbs (Twisted-Edwards
// BLS12) and N = 14 limbs (BLS12-377 G1 in twisted-Edwards form).  Memory layouts, in u32 words:
//   accumulator (ete_t<N>)   4 N words  x | y | z | t            144 B / 224 B, 16-byte accesses
//   record (pnt_t<N>)        N = 9: 27 words in a 128-byte slot (one gather = one 128-byte line)
//                            N = 14: 56 words = 224 bytes, slots back to back
template <int N> struct geo {
  static constexpr uint32_t PW = 4u * N;                    // words per accumulator
  static constexpr uint32_t PQ = PW / 4u;                   // 16-byte pieces per accumulator
  static constexpr uint32_t RW = N == 9 ? 27u : 4u * N;     // record words
  static constexpr uint32_t SQ = N == 9 ? 8u : N;           // 16-byte pieces per record slot
  static constexpr uint32_t LQ = (RW + 3u) / 4u;            // 16-byte pieces to load per record
};
template <int N> struct rec_slot { uint4 q[geo<N>::SQ]; };
static_assert(sizeof(rec_slot<9>) == sizeof(pnt_slot) && sizeof(rec_slot<14>) == 224, "record slots");

// ------------------------------------------------------------------------------------------------
// K3: bucket accumulation, one thread per segment, scheduled through order[] (or natural order when
// order == nullptr).  The next record is fetched while the current addition runs; the first entry of a segment is not
// added to the neutral element but converted (ete_from_pnt, 3 products instead of 7).  A bucket that is a single
// segment is written straight to buckets[]; parts of a split bucket go to seg_out[] for k_seg_combine*.
// (A variant that fused level 2 of the sort into this kernel -- one block per 256 buckets, lists consumed
// straight from LDS -- was measured at 2.8 ms against 1.4 ms: block-granular scheduling leaves < 1 wave per
// SIMD resident on average (SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE = 0.78), far too few to hide gather latency.)
template <int N> __device__ __forceinline__ pnt_t<N> load_pnt(const rec_slot<N>* __restrict__ recs, uint32_t entry) {
  const uint4* q = recs[entry & 0x7fffffffu].q;
  constexpr uint32_t LQ = geo<N>::LQ;
  uint4 u[LQ];
#pragma unroll
  for (uint32_t j = 0; j < LQ; j++) u[j] = q[j];
  uint32_t w[4 * LQ];
#pragma unroll
  for (uint32_t j = 0; j < LQ; j++) { w[4 * j] = u[j].x; w[4 * j + 1] = u[j].y; w[4 * j + 2] = u[j].z; w[4 * j + 3] = u[j].w; }
  pnt_t<N> r;
#pragma unroll
  for (int j = 0; j < N; j++) { r.hm.v[j] = w[j]; r.hp.v[j] = w[N + j]; r.dt.v[j] = w[2 * N + j]; }
  if constexpr (N == 14) {
#pragma unroll
    for (int j = 0; j < N; j++) r.z.v[j] = w[3 * N + j];
  }
  return r;
}
template <int N> __device__ __forceinline__ void store_pnt(rec_slot<N>* dst, const pnt_t<N>& r) {
  uint32_t w[4 * geo<N>::SQ];
#pragma unroll
  for (uint32_t j = 0; j < 4 * geo<N>::SQ; j++) w[j] = 0u;
#pragma unroll
  for (int j = 0; j < N; j++) { w[j] = r.hm.v[j]; w[N + j] = r.hp.v[j]; w[2 * N + j] = r.dt.v[j]; }
  if constexpr (N == 14) {
#pragma unroll
    for (int j = 0; j < N; j++) w[3 * N + j] = r.z.v[j];
  }
#pragma unroll
  for (uint32_t j = 0; j < geo<N>::SQ; j++) dst->q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}
template <int N> __device__ __forceinline__ void store_ete(ete_t<N>* dst, const ete_t<N>& a) {
  uint32_t w[4 * N];
#pragma unroll
  for (int j = 0; j < N; j++) { w[j] = a.x.v[j]; w[N + j] = a.y.v[j]; w[2 * N + j] = a.z.v[j]; w[3 * N + j] = a.t.v[j]; }
  uint4* o = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int j = 0; j < N; j++) o[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}
template <int N> __device__ __forceinline__ ete_t<N> load_ete(const ete_t<N>* src) {
  const uint4* q = reinterpret_cast<const uint4*>(src);
  uint32_t w[4 * N];
#pragma unroll
  for (int j = 0; j < N; j++) { const uint4 u = q[j]; w[4 * j] = u.x; w[4 * j + 1] = u.y; w[4 * j + 2] = u.z; w[4 * j + 3] = u.w; }
  ete_t<N> a;
#pragma unroll
  for (int j = 0; j < N; j++) { a.x.v[j] = w[j]; a.y.v[j] = w[N + j]; a.z.v[j] = w[2 * N + j]; a.t.v[j] = w[3 * N + j]; }
  return a;
}

// Affine BLS12-377 record of a bound point set (curve.hpp, pnt_aff377): 42 limb words = 168 bytes, slots back to back (8-byte
// aligned: ten 16-byte loads at 8-byte alignment -- gfx950 takes global_load_dwordx4 at any 4-byte alignment -- and one 8-byte load)
struct __attribute__((aligned(8))) rec_aff377 { uint32_t w[42]; };
struct __attribute__((aligned(8))) u4a8 { uint32_t v[4]; };
struct __attribute__((aligned(8))) u2a8 { uint32_t v[2]; };
static_assert(sizeof(rec_aff377) == 168, "affine record slot");
__device__ __forceinline__ pnt_aff377 load_pnt_aff377(const rec_aff377* __restrict__ recs, uint32_t entry) {
  const uint32_t* base = recs[entry & 0x7fffffffu].w;
  u4a8 u[10];
#pragma unroll
  for (int j = 0; j < 10; j++) u[j] = reinterpret_cast<const u4a8*>(base)[j];
  const u2a8 t = *reinterpret_cast<const u2a8*>(base + 40);
  uint32_t w[42];
#pragma unroll
  for (int j = 0; j < 10; j++) { w[4 * j] = u[j].v[0]; w[4 * j + 1] = u[j].v[1]; w[4 * j + 2] = u[j].v[2]; w[4 * j + 3] = u[j].v[3]; }
  w[40] = t.v[0]; w[41] = t.v[1];
  pnt_aff377 r;
#pragma unroll
  for (int j = 0; j < 14; j++) { r.hm.v[j] = w[j]; r.hp.v[j] = w[14 + j]; r.dt.v[j] = w[28 + j]; }
  return r;
}
__device__ __forceinline__ void store_pnt_aff377(rec_aff377* dst, const pnt_aff377& r) {
#pragma unroll
  for (int j = 0; j < 14; j++) { dst->w[j] = r.hm.v[j]; dst->w[14 + j] = r.hp.v[j]; dst->w[28 + j] = r.dt.v[j]; }
}
// The record a launch of k_accumulate gathers: RK = 0 the curve's own (converted per call: rec_slot<N>), RK = 1 the affine
// BLS12-377 record of a bound point set (N = 14 only)
template <int N, int RK> struct rec_kind;
template <int N> struct rec_kind<N, 0> {
  using slot = rec_slot<N>; using pnt = pnt_t<N>;
  static __device__ __forceinline__ pnt load(const slot* __restrict__ recs, uint32_t e) { return load_pnt<N>(recs, e); }
};
template <> struct rec_kind<14, 1> {
  using slot = rec_aff377; using pnt = pnt_aff377;
  static __device__ __forceinline__ pnt load(const slot* __restrict__ recs, uint32_t e) { return load_pnt_aff377(recs, e); }
};

struct __attribute__((aligned(4))) idx4 { uint32_t v[4]; };      // 16 bytes at 4-byte alignment: one global_load_dwordx4
#define TE_CLK_SLOTS 64u       // copies of k_accumulate's four profiling words (64-bit each), see the kernel
#define TE_IDX_STRIP 16u       // sorted indices a lane fetches at a time (k_accumulate); d_sorted is padded by as many words
// registers: N = 9 fits four waves per SIMD in 128 VGPRs (six of them spilled) and ran that way in rounds 1-3; round 4 asks for
// THREE (133 VGPRs, nothing spilled): the VALU is as busy with three waves per SIMD (0.93-0.94 of the issue estimate either way),
// and boxes whose clock sags under this kernel sustain a higher one with fewer waves in flight -- 2.11-2.19 GHz against 2.00-2.07,
// the kernel alone 0.77-0.79 ms against 0.82-0.85, +2-4 % MSM/s there, nothing lost on boxes that hold 2.15 GHz anyway
// (profiles/r04_accumulate_occupancy_experiment.txt).  N = 14 holds 56 + 2 x 56 words of points alone: two waves
template <int N, int RK = 0>
__global__ void __launch_bounds__(256, N == 9 ? 3 : 2) k_accumulate(const typename rec_kind<N, RK>::slot* __restrict__ recs, const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_count,
                                                    const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ seg_bucket,
                                                    const uint32_t* __restrict__ seg_lenv, const uint32_t* __restrict__ order,
                                                    const uint32_t* __restrict__ num_segments, ete_t<N>* __restrict__ buckets,
                                                    ete_t<N>* __restrict__ seg_out, uint32_t n, uint32_t logB, uint32_t seg_len, uint32_t ids, uint32_t onto,
                                                    uint32_t win_per_msm, batch_slabs slabs, unsigned long long* __restrict__ clk) {
  __shared__ uint32_t idx_strip[256 * TE_IDX_STRIP];
  using RT = rec_kind<N, RK>; using P = typename RT::pnt;
#if defined(TE_ACC_TWO_WAVES)
  // A/B builds only: naming a high register makes the kernel's VGPR allocation 176, i.e. two waves per SIMD without touching LDS
  if constexpr (N == 9) asm volatile("" ::: "v175");
#endif
  // profiling: ~clock of the first wave in and clock of the last wave out, by atomic max on zeroed words -- the kernel's
  // own duration on the device, which an event pair around the launch overstates when other streams' kernels hold the
  // CUs (te_msm_stage_ms "accumulate_on_device"); per-wave shader-clock and wall-clock ticks give the core clock it ran at.
  // TE_CLK_SLOTS copies of every word, chosen by block: thousands of waves ending together on ONE address serialise
  // (0.1 ms at n = 2^16, where all segments are short); the host folds the copies.
  unsigned long long* const slot = clk ? clk + (blockIdx.x & (TE_CLK_SLOTS - 1u)) : nullptr;
  const unsigned long long t0_wall = clk ? (unsigned long long)wall_clock64() : 0ull, t0_core = clk ? (unsigned long long)clock64() : 0ull;   // scalar registers
  if (clk && threadIdx.x == 0) atomicMax(slot, ~t0_wall);
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= (order ? *num_segments : ids)) return;
  const uint32_t sgm = order ? order[gid] : gid;
  const uint32_t g = seg_bucket[sgm];                // g = k * B + b
  if (g == TE_SEG_INVALID) return;
  const uint32_t k = g >> logB;
  recs += (size_t)slabs.s[k / win_per_msm] * n;      // a batch of MSMs (te_msm_partial_device_batch): window k belongs to MSM k / win_per_msm
  const uint32_t part = sgm - seg_base[g];
  const uint32_t cnt = seg_lenv[sgm];
  const uint32_t* lst = sorted + (size_t)k * n + bucket_start[g] + part * seg_len;
  // onto: the buckets already hold the sums of earlier pieces of the same MSM (te_msm_run uploads and processes large host
  // buffers in pieces): the first part of every bucket continues from that value instead of the neutral element
  ete_t<N> acc = (onto && part == 0u) ? load_ete<N>(buckets + g) : ete_identity_t<N>();
  if (cnt) {
    // Software pipeline: while the addition of entry j runs, the record of entry j+1 is in flight and the index of entry
    // j+2 is read from the LDS strip below -- nothing that is loaded in an iteration is waited for in the same iteration
    // except the strip's refill, once in TE_IDX_STRIP additions.  (The first version fetched index j+1 and then its
    // record inside one iteration: an s_waitcnt on the index in front of the record loads, one memory latency exposed
    // per addition.)
    const uint32_t last = cnt - 1u;
    uint32_t j0 = 0;
    uint32_t e = lst[0], e_n = lst[min(1u, last)];
    P cur = RT::load(recs, e);
    if (!(onto && part == 0u)) {
      // first entry: neutral element + P needs 3 (4) products, not 7 (8) -- except where the segment continues a bucket
      const P first = pnt_cneg(cur, (e >> 31) != 0u);
      e = e_n; e_n = lst[min(2u, last)];
      cur = RT::load(recs, e);
      acc = ete_from_pnt(first);
      j0 = 1;
    }
    // Indices come through a private LDS strip, TE_IDX_STRIP at a time: a lane's list is 4 B per addition, and between two
    // of its accesses the wave front has pulled megabytes of records through the L2 -- read one by one, every index access
    // re-fetched its 128-byte line (profiles/r02_pmc_l2_k_accumulate.txt: 6 M extra lines per launch; FETCH_SIZE 1.60 ->
    // 1.36 GB raw per launch with the strip).  Entries beyond the lane's segment belong to other lists (or to the 64 words
    // of padding behind `sorted`, ensure_buffers) and are never used.
    uint32_t* const strip = idx_strip + threadIdx.x * TE_IDX_STRIP;
    auto refill = [&](uint32_t pos) {                      // two rounds of eight: the second one finds the line in the L2
      const idx4* src = reinterpret_cast<const idx4*>(lst + pos);
      idx4* dst = reinterpret_cast<idx4*>(strip);
#pragma unroll
      for (uint32_t h = 0; h < TE_IDX_STRIP / 4u; h += 2u) {
        const idx4 v0 = src[h], v1 = src[h + 1u];
        dst[h] = v0; dst[h + 1u] = v1;
      }
    };
    if (((j0 + 2u) & (TE_IDX_STRIP - 1u)) != 0u) refill((j0 + 2u) & ~(TE_IDX_STRIP - 1u));
    for (uint32_t j = j0; j < cnt; j++) {
      const uint32_t e_cur = e;
      P nxt = cur;
      const uint32_t pn = j + 2u;
      if ((pn & (TE_IDX_STRIP - 1u)) == 0u) refill(pn);
      const uint32_t e_nn = strip[pn & (TE_IDX_STRIP - 1u)];
      if (j + 1 < cnt) { e = e_n; nxt = RT::load(recs, e); }
      acc = ete_madd(acc, pnt_cneg(cur, (e_cur >> 31) != 0u));
      cur = nxt; e_n = e_nn;
    }
  }
  const bool whole = bucket_count[g] <= seg_len;
  store_ete<N>(whole ? buckets + g : seg_out + sgm, acc);
  if (clk && (uint32_t)__lane_id() == (uint32_t)__ffsll((long long)__ballot(1)) - 1u) {
    const unsigned long long t1_wall = (unsigned long long)wall_clock64(), t1_core = (unsigned long long)clock64();
    atomicMax(slot + TE_CLK_SLOTS, t1_wall);
    // the shader clock counts per XCD and is not synchronised across them: every wave adds its OWN core ticks and wall ticks;
    // the ratio of the two sums is the mean core clock the kernel's waves ran at
    atomicAdd(slot + 2u * TE_CLK_SLOTS, t1_core - t0_core); atomicAdd(slot + 3u * TE_CLK_SLOTS, t1_wall - t0_wall);
  }
}

// ------------------------------------------------------------------------------------------------
// K4a: marginal sums.  Bucket j of a window has weight j + 1; write j in four digits j = (d3 d2 d1 d0) of w3..w0 bits
// (15 = 3+4+4+4 for c = 16).  With M_k[v] = sum of all buckets whose digit k equals v (<= 16 values per digit):
//     sum_j (j+1) B_j = sum_j B_j + sum_k 2^(w0+..+w(k-1)) * sum_v v * M_k[v].
// The powers of two cost nothing: they are folded into Horner's doublings on the host.  Everything here is plain
// sums, computed by folding one digit at a time, 8 (4, 2) points per thread per level:
//     out[o] = sum_{t<K} in[(outer*K + t)*inner + q],   o = outer*inner + q
// (inner = 1 folds a contiguous digit, inner > 1 folds a higher one).  Up to four independent jobs share a launch
// (blockIdx.y).  The first level is throughput-bound (one thread per output); the later, small levels are
// latency-bound and use four lanes per output (k_sum_groups_team below).
template <int N> struct sum_job_t {
  const ete_t<N>* in; ete_t<N>* out;
  uint32_t n_out;      // outputs per window (0 = no job)
  uint32_t K, inner;
  uint32_t in_per_window, out_per_window;
};
template <int N> struct sum_jobs_t { sum_job_t<N> j[4]; };
// PAIR: two adjacent lanes share an output -- each sums half of the K inputs, the even lane adds the odd lane's half (handed over
// with DPP moves) and stores: K/2 dependent additions instead of K - 1 for the same number of additions in all.  For levels with
// too few outputs to fill the machine (n <= 2^18: 73 728 outputs on 1024 SIMDs -- a lone wave issues a dependent chain at half
// rate): 60 -> 52 us at n = 2^16 and 2^18.
template <int N> __device__ __forceinline__ ete_t<N> pair_swap(const ete_t<N>& a) {        // the other lane of my pair (lanes 2i, 2i + 1)
  ete_t<N> r;
#pragma unroll
  for (int i = 0; i < N; i++) {
    r.x.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.x.v[i], 0xb1, 0xf, 0xf, true);
    r.y.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.y.v[i], 0xb1, 0xf, 0xf, true);
    r.z.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.z.v[i], 0xb1, 0xf, 0xf, true);
    r.t.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.t.v[i], 0xb1, 0xf, 0xf, true);
  }
  return r;
}
template <int N, bool PAIR>
__global__ void __launch_bounds__(256, N == 9 ? 2 : 1) k_sum_groups(sum_jobs_t<N> js, uint32_t nw) {
  const sum_job_t<N>& j = js.j[blockIdx.y];
  constexpr uint32_t TPO = PAIR ? 2u : 1u;                // threads per output
  const uint32_t total = j.n_out * nw * TPO, Kt = j.K / TPO;      // PAIR needs an even K (the host folds by 8, 4 or 2)
  for (uint32_t gt = blockIdx.x * 256u + threadIdx.x; gt < total; gt += gridDim.x * 256u) {      // pairs stay in adjacent lanes
    const uint32_t g = gt / TPO, half = gt % TPO;
    const uint32_t k = g / j.n_out, o = g - k * j.n_out;
    const uint32_t outer = o / j.inner, q = o - outer * j.inner;
    const ete_t<N>* src = j.in + (size_t)k * j.in_per_window + ((size_t)outer * j.K + (size_t)half * Kt) * j.inner + q;
    ete_t<N> acc = load_ete<N>(src);
    if (Kt > 1u) {
      ete_t<N> nxt = load_ete<N>(src + (size_t)j.inner);        // the next operand is in flight during the addition
      for (uint32_t t = 1; t < Kt; t++) {
        const ete_t<N> cur = nxt;
        if (t + 1 < Kt) nxt = load_ete<N>(src + (size_t)(t + 1) * j.inner);
        acc = ete_add<N>(acc, cur);
      }
    }
    if (PAIR) {
      const ete_t<N> other = pair_swap<N>(acc);
      acc = ete_add<N>(acc, other);                          // both lanes compute it (one instruction stream), the even one stores
      if (half != 0u) continue;
    }
    store_ete<N>(j.out + (size_t)k * j.out_per_window + o, acc);
  }
}

// ------------------------------------------------------------------------------------------------
// Team additions for the latency-bound tail of the reduction.  The last marginal-sum levels and the weighted sums
// are chains of dependent full additions executed by a handful of waves: one lane doing the 9 products of an
// addition in sequence takes ~5 us per link.  Here FOUR adjacent lanes share one point -- lane q of the quad holds
// coordinate q of (X, Y, T, Z) -- and an addition is three rounds of one product per lane:
//     round 1   lane0: A = (Y1-X1)(Y2-X2)   lane1: B = (Y1+X1)(Y2+X2)   lane2: T1*T2        lane3: Z1*Z2
//     round 2   lane2: C = 2d * T1T2 (the other lanes' product is discarded; lane3: D = 2 Z1Z2 by an addition)
//     round 3   lane0: X3 = EF              lane1: Y3 = HG              lane2: T3 = EH       lane3: Z3 = FG
// with E = B-A, H = B+A, F = D-C, G = D+C formed on every lane after a quad broadcast (DPP quad_perm, no LDS).
// 3 product latencies instead of 9, at 75 % lane efficiency -- the right trade when the machine is idle anyway.
template <int N> __device__ __forceinline__ fel<N> quad_bcast(const fel<N>& v, const int k) {     // value of lane k of my quad, k uniform constant 0..3
  fel<N> r;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int x = (int)v.v[i];
    int y;
    switch (k) {
      case 0: y = __builtin_amdgcn_mov_dpp(x, 0x00, 0xf, 0xf, true); break;
      case 1: y = __builtin_amdgcn_mov_dpp(x, 0x55, 0xf, 0xf, true); break;
      case 2: y = __builtin_amdgcn_mov_dpp(x, 0xaa, 0xf, 0xf, true); break;
      default: y = __builtin_amdgcn_mov_dpp(x, 0xff, 0xf, 0xf, true); break;
    }
    r.v[i] = (uint32_t)y;
  }
  return r;
}
template <int N> __device__ __forceinline__ fel<N> quad_swap1(const fel<N>& v) {                 // lanes 0<->1, 2<->3 of every quad
  fel<N> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.v[i], 0xb1, 0xf, 0xf, true);
  return r;
}
template <int N> __device__ __forceinline__ fel<N> fp_select(bool c, const fel<N>& a, const fel<N>& b) {
  fel<N> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
// per-lane choice among field elements, limb by limb under a lane mask (m = all ones: b, else a).  NOT written as `c ? b : a` on
// the structs: that is a choice between two OBJECTS, which the compiler resolved for one limb of ete_add_team's operands by
// storing both candidates to scratch memory and loading one back with a per-lane offset -- a store + load round trip through
// the vector memory path inside every team addition of the latency-bound kernels (12 bytes of scratch per lane in
// k_seg_combine_all, k_sum_groups_team and k_reduce_tail until round 4).
template <int N> __device__ __forceinline__ fel<N> fe_pick(uint32_t m, const fel<N>& b, const fel<N>& a) {
  fel<N> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = mask_select(m, b.v[i], a.v[i]);
  return r;
}
// q = lane & 3 selects the coordinate this lane holds: 0 X, 1 Y, 2 T, 3 Z.  All four lanes of a quad must call.
// Limb rule: N = 9 normalises the first-point difference of round 1 and F (E F, H G, E H, F G then hold one narrow operand or are
// "difference x sum"); N = 14 one operand of every product (lanes 0, 1: the first-point side of round 1; E, F, G).
template <int N> __device__ __forceinline__ fel<N> ete_add_team(const fel<N>& m1, const fel<N>& m2, const uint32_t q) {
  const fel<N> o1 = quad_swap1<N>(m1), o2 = quad_swap1<N>(m2);          // lane0 sees Y, lane1 sees X (lanes 2,3: unused)
  // round 1 operands
  const fel<N> u0 = fe_norm(fe_sub<2>(o1, m1)), v0 = fe_sub<2>(o2, m2); // lane0: Y1-X1, Y2-X2
  const fel<N> u1 = fe_norm_if_needed(fe_add(m1, o1)), v1 = fe_add(m2, o2);   // lane1: Y1+X1, Y2+X2
  const uint32_t q0 = q == 0u ? ~0u : 0u, q1 = q == 1u ? ~0u : 0u, q2 = q == 2u ? ~0u : 0u, q3 = q == 3u ? ~0u : 0u;     // lane masks
  const fel<N> u = fe_pick<N>(q0, u0, fe_pick<N>(q1, u1, m1));
  const fel<N> v = fe_pick<N>(q0, v0, fe_pick<N>(q1, v1, m2));
  const fel<N> s1 = fe_mul(u, v);                                       // A | B | T1T2 | Z1Z2
  // round 2
  const fel<N> s2 = fe_mul_k2d(s1);                                     // meaningful on lane2 only (N = 9: ~50 instructions instead of a product: a third of the chain's products gone)
  const fel<N> val = fe_pick<N>(q2, s2, fe_pick<N>(q3, fe_add(s1, s1), s1));      // A | B | C | D
  // round 3
  const fel<N> A = quad_bcast<N>(val, 0), B = quad_bcast<N>(val, 1), C = quad_bcast<N>(val, 2), D = quad_bcast<N>(val, 3);
  const fel<N> E = fe_norm_if_needed(fe_sub<2>(B, A)), H = fe_add(B, A), F = fe_norm(fe_sub<2>(D, C)), G = fe_norm_if_needed(fe_add(D, C));   // limb rule: see ete_add
  const fel<N> uu = fe_pick<N>(q0 | q2, E, fe_pick<N>(q1, H, F));
  const fel<N> vv = fe_pick<N>(q0, F, fe_pick<N>(q2, H, G));
  return fe_mul(uu, vv);                                                // X3 | Y3 | T3 | Z3
}
// word offset of coordinate q inside a stored point (memory order x | y | z | t)
template <int N> __device__ __forceinline__ uint32_t team_word(uint32_t q) { return (q == 0 ? 0u : q == 1 ? 1u : q == 2 ? 3u : 2u) * N; }
template <int N> __device__ __forceinline__ fel<N> load_coord(const uint32_t* p) { fel<N> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = p[i];
  return r; }
template <int N> __device__ __forceinline__ void store_coord(uint32_t* p, const fel<N>& a) {
#pragma unroll
  for (int i = 0; i < N; i++) p[i] = a.v[i];
}
template <int N> __device__ __forceinline__ fel<N> identity_coord(uint32_t q) { return (q == 1 || q == 3) ? fe_one<N>() : fe_zero<N>(); }
template <int N> __device__ __forceinline__ const uint32_t* words(const ete_t<N>* p) { return reinterpret_cast<const uint32_t*>(p); }
template <int N> __device__ __forceinline__ uint32_t* words(ete_t<N>* p) { return reinterpret_cast<uint32_t*>(p); }

// block-wide sum of cnt points src[0], src[stride], ... by 64 quads: strided serial team additions, then an LDS tree.
// Returns the sum in quad 0.  blockDim.x must be 256; lds holds 64 points.
template <int N> __device__ __forceinline__ fel<N> load_coord_agent(const uint32_t* p) { fel<N> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = ld_agent(p + i);
  return r; }
template <int N> __device__ __forceinline__ void store_coord_agent(uint32_t* p, const fel<N>& a) {
#pragma unroll
  for (int i = 0; i < N; i++) st_agent(p + i, a.v[i]);
}
// COHERENT: the inputs were written by other blocks of this launch (store_coord_agent): read them past the L2
template <int N, bool COHERENT = false> __device__ __forceinline__ fel<N> block_sum_points(const ete_t<N>* __restrict__ src, uint32_t stride, uint32_t cnt, uint32_t* lds) {
  constexpr uint32_t PW = geo<N>::PW;
  const uint32_t i = threadIdx.x >> 2, q = threadIdx.x & 3u, w = team_word<N>(q);
  fel<N> acc = identity_coord<N>(q);
  auto fetch = [&](uint32_t j) { return COHERENT ? load_coord_agent<N>(words<N>(src + (size_t)j * stride) + w) : load_coord<N>(words<N>(src + (size_t)j * stride) + w); };
  if (i < cnt) {                                  // quad-uniform trip count differs between quads: DPP stays inside a quad
    acc = fetch(i);
    fel<N> nxt = fetch(min(i + 64u, cnt - 1u));   // the next operand is in flight during the addition
    for (uint32_t j = i + 64u; j < cnt; j += 64u) {
      const fel<N> cur = nxt;
      if (j + 64u < cnt) nxt = fetch(j + 64u);
      acc = ete_add_team<N>(acc, cur, q);
    }
  }
  for (uint32_t s = 32; s > 0; s >>= 1) {
    if (s >= cnt) continue;                                 // uniform: no quad beyond s holds a value
    if (i >= s && i < 2 * s) store_coord<N>(lds + (size_t)i * PW + w, acc);
    __syncthreads();
    if (((threadIdx.x & ~63u) >> 2) < s) {                  // a wave all of whose quads are beyond s has nothing to add (wave-uniform)
      const bool act = i < s && i + s < cnt;
      const fel<N> other = act ? load_coord<N>(lds + (size_t)(i + s) * PW + w) : identity_coord<N>(q);
      const fel<N> sum = ete_add_team<N>(acc, other, q);
      acc = fp_select<N>(act, sum, acc);
    }
    __syncthreads();
  }
  return acc;
}
// sums the parts of split buckets.  Blocks [0, quad_blocks) sum the buckets cut into 2..16 parts, one quad (team addition)
// per entry of the split list; the remaining blocks sum the runs of TE_GIANT_RUN parts of GIANT buckets (skewed scalars, or a top
// window with one or two occupied buckets) into the first slot of each run, and the block that finishes a bucket's LAST run
// sums those slots into the bucket (until round 4 a second launch, k_seg_combine_large2, near-empty for well-spread digits:
// 5 us on every MSM's critical path).  The arrival counter of a bucket is its bucket_cursor word, dead since the placement:
// it holds bucket_start + bucket_count, the arrivals are counted on top.  One block per work item, grid-stride.
template <int N>
__global__ void __launch_bounds__(256) k_seg_combine_all(const uint32_t* __restrict__ split_list, const uint32_t* __restrict__ counts,
                                                         const uint32_t* __restrict__ chunk_list, const uint32_t* __restrict__ bucket_count,
                                                         const uint32_t* __restrict__ seg_base, ete_t<N>* __restrict__ seg_out, ete_t<N>* __restrict__ buckets,
                                                         uint32_t seg_len, uint32_t chunk_cap, uint32_t quad_blocks,
                                                         const uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ bucket_cursor) {
  __shared__ uint32_t lds[64 * geo<N>::PW];
  __shared__ uint32_t arrived;
  const uint32_t q = threadIdx.x & 3u, wq = team_word<N>(q);
  if (blockIdx.x < quad_blocks) {
    const uint32_t nsplit = counts[0];
    for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) >> 2; i < nsplit; i += (quad_blocks * 256u) >> 2) {
      const uint32_t g = split_list[i];
      const uint32_t ns = (bucket_count[g] + seg_len - 1u) / seg_len, s0 = seg_base[g];
      fel<N> acc = load_coord<N>(words<N>(seg_out + s0) + wq);
      fel<N> nxt = load_coord<N>(words<N>(seg_out + s0 + 1u) + wq);                  // ns >= 2: the next part is in flight during the addition
      for (uint32_t j = 1; j < ns; j++) {
        const fel<N> cur = nxt;
        if (j + 1u < ns) nxt = load_coord<N>(words<N>(seg_out + s0 + j + 1u) + wq);
        acc = ete_add_team<N>(acc, cur, q);
      }
      store_coord<N>(words<N>(buckets + g) + wq, acc);
    }
    return;
  }
  const uint32_t items = min(counts[2], chunk_cap), nb = gridDim.x - quad_blocks;
  constexpr uint32_t run = TE_GIANT_RUN;
  for (uint32_t it = blockIdx.x - quad_blocks; it < items; it += nb) {
    const uint32_t g = chunk_list[2 * it], part = chunk_list[2 * it + 1];
    const uint32_t ns = (bucket_count[g] + seg_len - 1u) / seg_len, cnt = min(run, ns - part);
    ete_t<N>* base = seg_out + seg_base[g] + part;
    const fel<N> r = block_sum_points<N>(base, 1u, cnt, lds);
    const uint32_t nchunks = (ns + run - 1u) / run;
    if (nchunks == 1u) {                                  // (cannot happen: a giant bucket has more than 16 parts; kept exact)
      if ((threadIdx.x >> 2) == 0) store_coord<N>(words<N>(buckets + g) + wq, r);
      __syncthreads();
      continue;
    }
    if ((threadIdx.x >> 2) == 0) store_coord_agent<N>(words<N>(base) + wq, r);      // past the L2: another XCD's block may add the runs up
    wait_own_accesses();                                   // ... and performed before the arrival is counted
    __syncthreads();
    if (threadIdx.x == 0) arrived = handoff_arrive(&bucket_cursor[g]) - (bucket_start[g] + bucket_count[g]);
    __syncthreads();
    if (arrived + 1u == nchunks) {                         // uniform: this block finished the bucket's last run
      const fel<N> t = block_sum_points<N, true>(seg_out + seg_base[g], run, nchunks, lds);
      if ((threadIdx.x >> 2) == 0) store_coord<N>(words<N>(buckets + g) + wq, t);
    }
    __syncthreads();
  }
}
// k_sum_groups with a quad per output (levels where the grid is too small to fill the machine): 4 threads per output.
// (Two quads per output -- each half of the chain, handed over with ds_bpermute: K/2 dependent team additions instead of K - 1 --
// was measured in round 4 and changes nothing: 17.8 against 18.0 us at n = 2^20, 16.4 against 16.3 at 2^16.)
template <int N>
__global__ void __launch_bounds__(256) k_sum_groups_team(sum_jobs_t<N> js, uint32_t nw) {
  const sum_job_t<N>& j = js.j[blockIdx.y];
  const uint32_t total = j.n_out * nw, q = threadIdx.x & 3u, wq = team_word<N>(q);
  for (uint32_t g = (blockIdx.x * 256u + threadIdx.x) >> 2; g < total; g += (gridDim.x * 256u) >> 2) {
    const uint32_t k = g / j.n_out, o = g - k * j.n_out;
    const uint32_t outer = o / j.inner, qq = o - outer * j.inner;
    const ete_t<N>* src = j.in + (size_t)k * j.in_per_window + (size_t)outer * j.K * j.inner + qq;
    // the next operand is in flight during the addition (loaded when needed, every step of the chain paid a memory latency)
    fel<N> acc = load_coord<N>(words<N>(src) + wq);
    fel<N> nxt = load_coord<N>(words<N>(src + (size_t)j.inner) + wq);                 // K >= 2
    for (uint32_t t = 1; t < j.K; t++) {
      const fel<N> cur = nxt;
      if (t + 1 < j.K) nxt = load_coord<N>(words<N>(src + (size_t)(t + 1) * j.inner) + wq);
      acc = ete_add_team<N>(acc, cur, q);
    }
    store_coord<N>(words<N>(j.out + (size_t)k * j.out_per_window + o) + wq, acc);
  }
}

// ------------------------------------------------------------------------------------------------
// K4 tail: everything after the wide fold levels in ONE launch, one block per (window, digit) -- team additions, points in
// LDS.  Replaces the last fold levels, the four second-phase chains and the weighted-sum kernel: six dependent,
// latency-bound launches of the first version.  Bucket index j = hi * L + lo with lo = (d1 d0) of w1 + w0 bits, hi = (d3 d2)
// of w3 + w2 bits.
//   in : xin[hi * rx + g], g < rx   partial row sums     X2[hi] = sum_g xin[hi * rx + g]       (H = 2^(w2+w3) values)
//        yin[h * L + lo], h < ry    partial column sums  Y2[lo] = sum_h yin[h * L + lo]        (L = 2^(w0+w1) values)
//   block (digit, window):  digit 3: M3[d3] = sum_d2 X2    digit 2: M2[d2] = sum_d3 X2
//                           digit 1: M1[d1] = sum_d0 Y2    digit 0: M0[d0] = sum_d1 Y2      (blocks of a window recompute X2 / Y2:
//                           a few hundred additions, cheaper than a launch boundary between them)
//   A    X2 or Y2 -> LDS: one quad per value, its rx (ry) <= 4 inputs loaded up front
//   B    the digit's marginal as a binary tree in place: item (o, t), t < m/2: in[o, t] += in[o, t + m/2]
//   C    first wave only: suffix scan S_v = sum_{u >= v} M[u] (T = S_0 of digit 0), W = sum_{v >= 1} S_v by a tree
//   out: slot 1 + digit (and slot 0 = T from the digit-0 block) of the row [T | W0 | W1 | W2 | W3] that the host tail folds
//        (te_host::horner_to_affine / te377_host::horner_to_affine).
// One level costs ~3 us (three dependent field products per team addition on a wave that owns its SIMD), a busy CU ~5 us:
// the blocks are kept small so that the ~15 levels of a window run on four nearly idle CUs.
template <int N> struct tail_params_t {
  const ete_t<N>* xin; const ete_t<N>* yin;
  uint32_t rx, ry, x_per_window, y_per_window;
  uint32_t w[4];
  ete_t<N>* rows; uint32_t row_stride;   // points
  uint32_t win_per_msm, msm_stride;      // batch: window k of the launch is window k % win_per_msm of MSM k / win_per_msm, whose rows start msm_stride points further
  // rows in HOST memory (te_msm_run / submit / collect: `rows` is the device-visible address of the work set's pinned block):
  // block (0, 0) also copies the MSM's flag words -- final-carry flag, counters, k_accumulate's clock words, all final before
  // this launch -- to the head of that block, so that no device-to-host copy (4 us and a launch gap) follows the kernel
  const uint32_t* flag_src; uint32_t* flag_dst; uint32_t flag_words;
};
template <int N> __device__ __forceinline__ uint32_t* lds_point(uint32_t* base, uint32_t idx) { return base + (size_t)idx * geo<N>::PW; }
// exchange through LDS among the lanes of ONE wave: the wave's earlier LDS writes are complete and visible before its later reads
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

template <int N>
__global__ void __launch_bounds__(1024) k_reduce_tail(tail_params_t<N> prm) {
  extern __shared__ uint32_t tl[];
  const uint32_t dgt = blockIdx.x, k = blockIdx.y, i = threadIdx.x >> 2, q = threadIdx.x & 3u, wq = team_word<N>(q), Q = blockDim.x >> 2;
  const uint32_t w0 = prm.w[0], w1 = prm.w[1], w2 = prm.w[2], w3 = prm.w[3];
  const bool isx = dgt >= 2u;
  if (prm.flag_dst && dgt == 0u && k == 0u)
    for (uint32_t j = threadIdx.x; j < prm.flag_words; j += blockDim.x) prm.flag_dst[j] = prm.flag_src[j];
  const uint32_t nv = isx ? 1u << (w2 + w3) : 1u << (w0 + w1);          // values of X2 (Y2)
  uint32_t* sv = tl;                                                     // X2 / Y2, then the tree in place
  uint32_t* sc = sv + (size_t)nv * geo<N>::PW;                           // 16 points of scratch for step C
  // ---- A
  {
    const uint32_t cnt = isx ? prm.rx : prm.ry, stride = isx ? 1u : nv;
    for (uint32_t o = i; o < nv; o += Q) {
      const ete_t<N>* src = isx ? prm.xin + (size_t)k * prm.x_per_window + (size_t)o * prm.rx : prm.yin + (size_t)k * prm.y_per_window + o;
      fel<N> in[4];
#pragma unroll
      for (int t = 0; t < 4; t++) in[t] = load_coord<N>(words<N>(src + (size_t)min((uint32_t)t, cnt - 1u) * stride) + wq);
      fel<N> acc = in[0];                                                                          // cnt is uniform
      if (cnt > 1u) acc = ete_add_team<N>(acc, in[1], q);
      if (cnt > 2u) acc = ete_add_team<N>(acc, in[2], q);
      if (cnt > 3u) acc = ete_add_team<N>(acc, in[3], q);
      store_coord<N>(lds_point<N>(sv, o) + wq, acc);
    }
  }
  __syncthreads();
  // ---- B: outputs nout, m inputs each; element (o, t) at o * ostr + t * istr
  const uint32_t wd = dgt == 0 ? w0 : dgt == 1 ? w1 : dgt == 2 ? w2 : w3;       // bits of this digit
  const uint32_t wo = dgt == 0 ? w1 : dgt == 1 ? w0 : dgt == 2 ? w3 : w2;       // bits of the digit summed away
  const bool low = dgt == 0 || dgt == 2;                                        // this digit is the low one of its pair
  const uint32_t nout = 1u << wd, ostr = low ? 1u : 1u << wo, istr = low ? 1u << wd : 1u;
  for (uint32_t m = 1u << wo; m > 1u; m >>= 1) {
    const uint32_t half = m >> 1, total = nout * half;
    for (uint32_t it0 = 0; it0 < total; it0 += Q) {                            // uniform trip count
      // A wave none of whose quads has an item skips the addition (quads never straddle waves, so the DPP exchanges inside it
      // stay whole): until round 4 all sixteen waves of the block executed every level under lane masks -- four waves per SIMD
      // issuing the ~1000 instructions of an addition for the one or two that had work, 6 us per level instead of 2-3.
      if (it0 + ((threadIdx.x & ~63u) >> 2) >= total) continue;               // wave-uniform
      const uint32_t it = it0 + i;
      const bool act = it < total;
      const uint32_t o = act ? it / half : 0u, t = act ? it - o * half : 0u;
      const uint32_t a_idx = o * ostr + t * istr, b_idx = a_idx + (act ? half * istr : 0u);
      const fel<N> a = load_coord<N>(lds_point<N>(sv, a_idx) + wq), b = load_coord<N>(lds_point<N>(sv, b_idx) + wq);
      const fel<N> sum = ete_add_team<N>(a, b, q);
      if (act) store_coord<N>(lds_point<N>(sv, a_idx) + wq, sum);
    }
    __syncthreads();
  }
  // ---- C: the first wave alone (16 quads, quad v holds M[v]); the other waves are done.  One wave: its LDS operations
  // complete in order, so a wave-level fence orders the exchanges below (no block barrier after other waves have left)
  if (threadIdx.x >= 64u) return;
  {
    const uint32_t v = i, Nv = nout;                                           // Nv <= 16
    fel<N> mine = v < Nv ? load_coord<N>(lds_point<N>(sv, v * ostr) + wq) : identity_coord<N>(q);
    uint32_t* slot = lds_point<N>(sc, v);
    for (uint32_t d = 1; d < 16u; d <<= 1) {           // inclusive suffix scan
      store_coord<N>(slot + wq, mine);
      wave_lds_sync();
      const bool act = v + d < Nv;
      const fel<N> other = act ? load_coord<N>(lds_point<N>(sc, v + d) + wq) : identity_coord<N>(q);
      const fel<N> sum = ete_add_team<N>(mine, other, q);
      mine = fp_select<N>(act, sum, mine);
      wave_lds_sync();
    }
    ete_t<N>* row = prm.rows + (size_t)(k / prm.win_per_msm) * prm.msm_stride + (size_t)(k % prm.win_per_msm) * prm.row_stride;
    if (v == 0) { if (dgt == 0) store_coord<N>(words<N>(row) + wq, mine); mine = identity_coord<N>(q); }
    for (uint32_t s = 8; s > 0; s >>= 1) {             // tree sum of S_1..S_{N-1} (slot 0 = identity)
      if (v >= s && v < 2 * s) store_coord<N>(slot + wq, mine);
      wave_lds_sync();
      const bool act = v < s && v + s < Nv;
      const fel<N> other = act ? load_coord<N>(lds_point<N>(sc, v + s) + wq) : identity_coord<N>(q);
      const fel<N> sum = ete_add_team<N>(mine, other, q);
      mine = fp_select<N>(act, sum, mine);
      wave_lds_sync();
    }
    if (v == 0) store_coord<N>(words<N>(row + 1 + dgt) + wq, mine);
  }
}

// ------------------------------------------------------------------------------------------------
// K1a for BLS12-377 G1: short-Weierstrass affine (x, y), 48-byte little-endian each -> projective twisted-Edwards record
// (curve.hpp, pnt_from_sw377), one lane per point; loads and stores are 16 bytes per lane.
__device__ __forceinline__ void prep_point377(uint32_t i, const uint4* __restrict__ pts, rec_slot<14>* __restrict__ recs) {
  uint4 u[6];
#pragma unroll
  for (int j = 0; j < 6; j++) u[j] = pts[6 * (size_t)i + j];
  const uint32_t xw[12] = {u[0].x, u[0].y, u[0].z, u[0].w, u[1].x, u[1].y, u[1].z, u[1].w, u[2].x, u[2].y, u[2].z, u[2].w};
  const uint32_t yw[12] = {u[3].x, u[3].y, u[3].z, u[3].w, u[4].x, u[4].y, u[4].z, u[4].w, u[5].x, u[5].y, u[5].z, u[5].w};
  store_pnt<14>(recs + i, pnt_from_sw377(te377::fq_from_words32(xw), te377::fq_from_words32(yw)));
}
__global__ void __launch_bounds__(256) k_prep_points377(batch_ptrs in, batch_slabs row_slab, rec_slot<14>* __restrict__ recs, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  prep_point377(i, in.p[blockIdx.y], recs + (size_t)row_slab.s[blockIdx.y] * n);
}
// Level 1 of the sort and the BLS12-377 record conversion in one launch (k_part_scatter_prep for the other curve): the scatter is
// bound by its LDS rounds and barriers, the conversion by its eight 14-limb products per point -- back to back they cost 46 + 130 us
// of one MSM's critical path.  1-D grid, the two kinds of blocks interleaved evenly; a conversion block converts 512 points (one
// per thread, no LDS).  Four waves per SIMD: the conversion needs ~100 VGPRs.
__global__ void __launch_bounds__(512, 4) k_part_scatter_prep377(scatter_args a, uint32_t scatter_blocks, batch_ptrs in, batch_slabs row_slab,
                                                                 rec_slot<14>* __restrict__ recs, uint32_t n, uint32_t prep_blocks_per_row, uint32_t prep_blocks) {
  __shared__ uint32_t lds[TE_SCATTER_LDS_WORDS];
  const uint64_t tot = (uint64_t)scatter_blocks + prep_blocks, b = blockIdx.x;
  const uint32_t s_before = (uint32_t)(b * scatter_blocks / tot), s_after = (uint32_t)((b + 1u) * scatter_blocks / tot);
  if (s_after > s_before) {                               // this block is scatter block number s_before
    part_scatter_block(s_before % a.g.CH, s_before / a.g.CH, lds, a);
  } else {
    const uint32_t pb = (uint32_t)b - s_before, row = pb / prep_blocks_per_row, blk = pb - row * prep_blocks_per_row;
    const uint32_t i = blk * 512u + threadIdx.x;
    if (i < n) prep_point377(i, in.p[row], recs + (size_t)row_slab.s[row] * n);
  }
}

// ------------------------------------------------------------------------------------------------
// te_msm_bind_points, BLS12-377: projective records (k_prep_points377) -> AFFINE records, once per bound point set.  Every
// thread takes TE_AFF_GROUP consecutive points: prefix products of their z (kept in the output slots, which are written last),
// ONE inversion of the group's product by Fermat's exponent q - 2 (a fixed square-and-multiply chain: 376 squarings and 188
// products, the same instruction stream in every lane), then z_j^-1 = inv * prefix_{j-1} backwards and three products per
// point.  About 0.6 k products per thread -- a fraction of one MSM's accumulation, paid once for every MSM over the set.
// A point whose z is 0 modulo q (y = 0 or the points of order 4: never in G1) keeps the factor 1 in the chain and gets an
// all-zero record, so that it cannot spoil its neighbours' inverses (its own bucket is undefined either way: include/te_msm.h).
#define TE_AFF_GROUP 8u
__device__ __forceinline__ bool fq_is_zero_mod_q(const fel<14>& a) {      // a: product output (class N, value < 2q): 0 or q
  const fel<14> t = fe_norm(fe_mul(a, fe_one<14>()));                     // a * R / R: below q + a little, limbs normalised
  const fel<14> qq = te377::fq_Q();
  uint32_t or0 = 0u, dq = 0u;
#pragma unroll
  for (int i = 0; i < 14; i++) { or0 |= t.v[i]; dq |= t.v[i] ^ qq.v[i]; }
  return or0 == 0u || dq == 0u;
}
__global__ void __launch_bounds__(256) k_affine377(const rec_slot<14>* __restrict__ proj, rec_aff377* __restrict__ out, uint32_t n) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x, lo = t * TE_AFF_GROUP;
  if (lo >= n) return;
  const uint32_t cnt = min(TE_AFF_GROUP, n - lo);
  const fel<14> one = fe_one<14>();
  // forward: prefix products (prefix_j = z_0 ... z_j with zeros skipped) into the first 14 words of output slot j
  fel<14> run = one;
  for (uint32_t j = 0; j < cnt; j++) {
    const pnt_t<14> r = load_pnt<14>(proj, lo + j);
    if (!fq_is_zero_mod_q(r.z)) run = fe_mul(run, r.z);
#pragma unroll
    for (int i = 0; i < 14; i++) out[lo + j].w[i] = run.v[i];
  }
  // inv = run^(q - 2): square-and-multiply from the top bit (bit 376 of q - 2 is set)
  // q - 2 in 32-bit words: q = 1 (mod 2^32), so the lowest word becomes ffffffff and the next one loses the borrow
  const uint32_t ex[12] = {0xffffffffu, te377::Q_W32[1] - 1u, te377::Q_W32[2], te377::Q_W32[3], te377::Q_W32[4], te377::Q_W32[5],
                           te377::Q_W32[6], te377::Q_W32[7], te377::Q_W32[8], te377::Q_W32[9], te377::Q_W32[10], te377::Q_W32[11]};
  fel<14> inv = run;
  for (int w = 11; w >= 0; w--) {
    const uint32_t word = ex[w];                          // (w is a loop counter the compiler unrolls over: no indexed register array)
    for (int b = (w == 11 ? 23 : 31); b >= 0; b--) {      // the top word holds bits 352 .. 376; bit 376 is `run` itself
      inv = fe_mul(inv, inv);
      if ((word >> b) & 1u) inv = fe_mul(inv, run);
    }
  }
  // backwards: z_j^-1 = inv * prefix_{j-1}; inv *= z_j
  for (uint32_t jj = cnt; jj-- > 0u;) {
    const pnt_t<14> r = load_pnt<14>(proj, lo + jj);
    pnt_aff377 a;
    if (fq_is_zero_mod_q(r.z)) {
      a.hm = a.hp = a.dt = fe_zero<14>();
    } else {
      fel<14> pre = one;
      if (jj > 0u) {
#pragma unroll
        for (int i = 0; i < 14; i++) pre.v[i] = out[lo + jj - 1u].w[i];
      }
      const fel<14> zi = fe_mul(inv, pre);
      inv = fe_mul(inv, r.z);
      const fel<14> l[3] = {r.hm, r.hp, r.dt}, rr[3] = {zi, zi, zi};
      fel<14> o[3];
      fe_mul_x<3>(l, rr, o);
      a.hm = o[0]; a.hp = o[1]; a.dt = o[2];
    }
    store_pnt_aff377(out + lo + jj, a);
  }
}

// ------------------------------------------------------------------------------------------------
// FIXED-BASE WINDOWS over a bound point set (te_msm_bind_points with option "bind_fixed_base" = c; round-5 verdict item 7).
// With the points resident, the multiples 2^(c w) P_i of every window w can be tabulated once.  Then every window's digit
// addresses the SAME bucket set: sum_i s_i P_i = sum_w sum_i d_{w,i} (2^(c w) P_i) = sum_b (b + 1) B_b with ONE set of 2^(c-1)
// buckets, filled by the W n table entries -- no per-window bucket sets, so c can grow: c = 20 has 13 windows and 2^19 buckets,
// 13 n + 2^20 additions where 16 windows of 16 bits need 16 n + 2^20 (n = 2^20: -17 %), at the price of a gather table of
// W x 128 MB that no longer fits the 256 MB Infinity Cache (profiles/r06_fixed_base_windows.txt).
// The engine's sort, accumulation and reduction run UNCHANGED on it: the bucket index b (c - 1 bits) is cut into
// (h_b : the top bits | lo : 15 bits) and the entries of equal h_b form a PSEUDO-WINDOW of 2^15 buckets -- a "row" of the digit
// buffer holding codes lo + 1 (0 = no entry; unsigned geometry, half = 0) with a parallel remap row (table index w n + i |
// sign << 31) that the level-1 scatter turns into the entry's index.  The top window of a canonical (253-bit) scalar has only
// 13 significant bits: all its n digits have h_b = 0; they go to one EXTRA row of weight offset 0 so that the rows stay
// balanced (12 n / RB entries in each regular row, about n in the extra one).  Weight of row r: h_b(r) 2^15 + (lo + 1); the
// host tail adds 2^15 sum_r h_b(r) T_r to the rows' own weighted sums (te_host::fixed_base_to_affine).
struct fb_digit_args {
  uint32_t half[10];        // sum_w 2^(c w + c - 1) over the W windows (as digits_params.half)
  uint32_t n, idx_base, n_table;     // scalars of this launch; table index of an entry = w * n_table + idx_base + i
  uint32_t W;               // windows of the decomposition
  uint32_t rows, rb;        // rows = rb + 1 pseudo-windows: rb regular ones (h_b), the extra one (index rb) for the top window's h_b = 0
  uint32_t cap;             // entries a row can hold (= row stride nst of the digit / remap buffers)
  uint32_t chunk_len, CH, P, logS;   // level-1 geometry of the sort (chunks per row, partitions per row)
  uint16_t* digits; uint32_t* remap; // [rows][cap]
  uint32_t* row_fill;       // [rows] entries reserved so far (zeroed per MSM)
  uint32_t* counts1;        // [rows][CH][P] level-1 histogram of the sort
  uint32_t* err;            // word 0: final carry; word 1: a row overflowed (the host falls back to the ordinary windows)
};
#define TE_FB_THREADS 256u
#define TE_FB_LOBITS 15u
// (Round 6 tried two refinements of this kernel and kept neither: staging the block's entries in LDS so that they leave as one contiguous
// run per row -- 25 KB more LDS per block, a third of the resident blocks: 122 instead of 106 us for 13.6 M entries -- and, on top of it,
// wave-aggregated ranks by ballots instead of one LDS atomic per entry: 140 us.  The kernel is bound by its occupancy and the latency of
// its scattered stores, not by the atomics.)
// one scalar per thread; LDS: cnt[rows] | base[rows] | hist[rows][2][P] (dynamic)
template <int C>
__global__ void __launch_bounds__(TE_FB_THREADS) k_fb_digits(const uint4* __restrict__ scalars, fb_digit_args a) {
  extern __shared__ uint32_t fl[];
  uint32_t* const cnt = fl; uint32_t* const base = fl + a.rows; uint32_t* const hist = fl + 2u * a.rows;
  const uint32_t t = threadIdx.x, i = blockIdx.x * TE_FB_THREADS + t, hn = a.rows * 2u * a.P;
  for (uint32_t j = t; j < 2u * a.rows + hn; j += TE_FB_THREADS) fl[j] = 0u;
  __syncthreads();
  constexpr int WMAX = (255 + C) / C + 1;
  const bool live = i < a.n;
  uint32_t s[10];
  {
    const size_t ii = live ? i : 0u;
    const uint4 a0 = scalars[2 * ii], a1 = scalars[2 * ii + 1];
    const uint32_t raw[10] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, 0u, 0u};
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 10; j++) { c += (uint64_t)raw[j] + a.half[j]; s[j] = (uint32_t)c; c >>= 32; }
  }
  // phase A: the row of every non-zero digit and its rank among the block's entries of that row
  uint32_t rank[WMAX]; bool bad = false;
#pragma unroll
  for (int w = 0; w < WMAX; w++) {
    rank[w] = 0xffffffffu;
    const int bit = w * C;
    if (bit >= 320) continue;
    const int word = bit >> 5, off = bit & 31;
    uint32_t v = s[word] >> off;
    if (off + C > 32 && word + 1 < 10) v |= s[word + 1] << (32 - off);
    v &= (1u << C) - 1u;
    if ((uint32_t)w >= a.W) { bad |= live && v != 0u; continue; }
    const int d = (int)v - (1 << (C - 1));
    if (!live || d == 0) continue;
    const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u, hb = b >> TE_FB_LOBITS;
    const uint32_t row = ((uint32_t)w + 1u == a.W && hb == 0u) ? a.rb : hb;
    rank[w] = atomicAdd(&cnt[row], 1u);
  }
  if (bad) atomicOr(a.err, 1u);
  __syncthreads();
  if (t < a.rows) base[t] = cnt[t] ? atomicAdd(&a.row_fill[t], cnt[t]) : 0u;        // one reservation per block and row
  __syncthreads();
  // phase B: codes and remap words to their rows; level-1 histogram of the block's entries (a block's range of a row touches two
  // chunks at most unless the digits are badly skewed: further chunks go straight to memory)
#pragma unroll
  for (int w = 0; w < WMAX; w++) {
    if (rank[w] == 0xffffffffu) continue;
    const int bit = w * C, word = bit >> 5, off = bit & 31;
    uint32_t v = s[word] >> off;
    if (off + C > 32 && word + 1 < 10) v |= s[word + 1] << (32 - off);
    v &= (1u << C) - 1u;
    const int d = (int)v - (1 << (C - 1));
    const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u, hb = b >> TE_FB_LOBITS, lo = b & ((1u << TE_FB_LOBITS) - 1u);
    const uint32_t row = ((uint32_t)w + 1u == a.W && hb == 0u) ? a.rb : hb;
    const uint32_t pos = base[row] + rank[w];
    if (pos >= a.cap) { atomicOr(a.err + 1, 1u); continue; }
    a.digits[(size_t)row * a.cap + pos] = (uint16_t)(lo + 1u);
    a.remap[(size_t)row * a.cap + pos] = ((uint32_t)w * a.n_table + a.idx_base + i) | (d < 0 ? 0x80000000u : 0u);
    const uint32_t ch = pos / a.chunk_len, slot = ch - base[row] / a.chunk_len, part = lo >> a.logS;
    if (slot < 2u) atomicAdd(&hist[(row * 2u + slot) * a.P + part], 1u);
    else atomicAdd(&a.counts1[((size_t)row * a.CH + ch) * a.P + part], 1u);
  }
  __syncthreads();
  for (uint32_t j = t; j < hn; j += TE_FB_THREADS) {
    const uint32_t v = hist[j];
    if (!v) continue;
    const uint32_t row = j / (2u * a.P), rem = j - row * 2u * a.P, slot = rem / a.P, part = rem - slot * a.P;
    const uint32_t ch = base[row] / a.chunk_len + slot;
    if (ch < a.CH) atomicAdd(&a.counts1[((size_t)row * a.CH + ch) * a.P + part], v);
  }
}

// ---- the table, once per bound set: records of window w = records of 2^(c w) P_i.  Window 0 is the ordinary conversion
// (k_prep_points); the extended points are carried from window to window by c doublings (the unified addition with itself) and
// every window's points go back to affine records with ONE inversion per TE_AFF_GROUP points (Montgomery's trick, Fermat chain).
// extended point from a record: x = hp - hm, y = hp + hm, z = 1, t = x y
__global__ void __launch_bounds__(256) k_fb_ext_from_recs(const pnt_slot* __restrict__ recs, ete* __restrict__ ext, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  store_ete<9>(ext + i, ete_from_pnt(load_pnt<9>(reinterpret_cast<const rec_slot<9>*>(recs), i)));      // (E 1, H 1, E H, 1): the point itself
}
__global__ void __launch_bounds__(256) k_fb_double(ete* __restrict__ ext, uint32_t n, uint32_t times) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  ete e = load_ete<9>(ext + i);
  for (uint32_t k = 0; k < times; k++) e = ete_add<9>(e, e);
  store_ete<9>(ext + i, e);
}
__device__ __forceinline__ bool fp_is_zero_mod_p(const fp& a) {           // a: product output (value < 2p): 0 or p
  const fp t = fp_norm(mont_mul(a, fp_R1()));
  uint32_t or0 = 0u, dp = 0u;
#pragma unroll
  for (int i = 0; i < NL; i++) { or0 |= t.v[i]; dp |= t.v[i] ^ p_limb(i); }
  return or0 == 0u || dp == 0u;
}
__global__ void __launch_bounds__(256) k_fb_records(const ete* __restrict__ ext, pnt_slot* __restrict__ out, uint32_t n) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x, lo = t * TE_AFF_GROUP;
  if (lo >= n) return;
  const uint32_t cnt = min(TE_AFF_GROUP, n - lo);
  const fp one = fp_R1();
  fp plain1 = fp_zero(); plain1.v[0] = 1u;
  const fp half_m = mont_mul(fp_R2_HALF(), plain1);                       // R / 2:  (a R)(R / 2) / R = (a / 2) R
  const fp negd_m = mont_mul(mont_mul(fp_NEG_D_R3(), plain1), plain1);    // -d R
  uint32_t* const scratch = reinterpret_cast<uint32_t*>(out);             // prefix products in the first 9 words of the (not yet written) slots
  fp run = one;
  for (uint32_t j = 0; j < cnt; j++) {
    const ete e = load_ete<9>(ext + lo + j);
    if (!fp_is_zero_mod_p(e.z)) run = mont_mul(run, e.z);
#pragma unroll
    for (int i = 0; i < NL; i++) scratch[(size_t)(lo + j) * 32u + i] = run.v[i];
  }
  // run^(p - 2): p = 1 (mod 2^32), so p - 2 ends in ... (P_W32[1] - 1) ffffffff; bit 252 of p - 2 is `run` itself
  const uint32_t ex[8] = {0xffffffffu, P_W32[1] - 1u, P_W32[2], P_W32[3], P_W32[4], P_W32[5], P_W32[6], P_W32[7]};
  fp inv = run;
  for (int w = 7; w >= 0; w--) {
    const uint32_t word = ex[w];
    for (int b = (w == 7 ? 27 : 31); b >= 0; b--) {
      inv = mont_mul(inv, inv);
      if ((word >> b) & 1u) inv = mont_mul(inv, run);
    }
  }
  for (uint32_t jj = cnt; jj-- > 0u;) {
    const ete e = load_ete<9>(ext + lo + jj);
    uint32_t w[32];
#pragma unroll
    for (int j = 0; j < 32; j++) w[j] = 0u;
    if (!fp_is_zero_mod_p(e.z)) {
      fp pre = one;
      if (jj > 0u) {
#pragma unroll
        for (int i = 0; i < NL; i++) pre.v[i] = scratch[(size_t)(lo + jj - 1u) * 32u + i];
      }
      const fp zi = mont_mul(inv, pre);
      inv = mont_mul(inv, e.z);
      const fp xy[2] = {e.x, e.y}, zz[2] = {zi, zi};
      fp o[2];
      mont_mul_x<2>(xy, zz, o);                                           // x, y (Montgomery form, class N)
      const fp l[3] = {fp_sub<2>(o[1], o[0]), fp_add(o[1], o[0]), o[0]}, rr[3] = {half_m, half_m, o[1]};
      fp q[3];
      mont_mul_x<3>(l, rr, q);                                            // (y - x) / 2, (y + x) / 2, x y
      const fp dt = mont_mul(q[2], negd_m);
#pragma unroll
      for (int j = 0; j < NL; j++) { w[j] = q[0].v[j]; w[NL + j] = q[1].v[j]; w[2 * NL + j] = dt.v[j]; }
    }
    uint4* dst = reinterpret_cast<uint4*>(out + lo + jj);
#pragma unroll
    for (int q4 = 0; q4 < 8; q4++) dst[q4] = make_uint4(w[4 * q4], w[4 * q4 + 1], w[4 * q4 + 2], w[4 * q4 + 3]);
  }
}

}  // namespace te
