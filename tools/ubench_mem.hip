// ubench_mem.hip -- why does a 1024-thread block with a big LDS allocation stream so slowly?
// Reads two arrays (keys u16 x N, idx u32 x N) partition-wise like k_bucket_sort's load phase.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void __launch_bounds__(1024) k_read(const uint4* __restrict__ k4, const uint4* __restrict__ i4, uint32_t* out, uint32_t groups_per_block, int chunks) {
  extern __shared__ uint32_t lds[];
  const uint32_t t = threadIdx.x;
  const uint4* kk = k4 + (size_t)blockIdx.x * groups_per_block;
  const uint4* ii = i4 + (size_t)blockIdx.x * groups_per_block * 2;
  uint32_t x = 0;
  for (int c = 0; c < chunks; c++) {
    const uint32_t gi = (uint32_t)c * 1024u + t;
    if (gi < groups_per_block) { const uint4 a = kk[gi], b = ii[2 * gi], d = ii[2 * gi + 1]; x ^= a.x ^ a.w ^ b.y ^ d.z; }
  }
  if (x == 0x1234567u) { out[0] = x; lds[t] = x; }
}
// variant B: exactly k_bucket_sort's load phase: 5 groups per thread issued up front at a clamped index
__global__ void __launch_bounds__(1024) k_read_b(const uint16_t* __restrict__ keys, const uint32_t* __restrict__ idx, uint32_t* out,
                                                 const uint32_t* __restrict__ starts, uint32_t nparts, uint32_t nst, int mode) {
  extern __shared__ uint32_t lds[];
  __shared__ uint32_t sm[17];
  const uint32_t p = blockIdx.x, k = blockIdx.y, t = threadIdx.x;
  const uint32_t start = starts[p], cnt = starts[p + 1] - starts[p];
  for (uint32_t j = t; j < 1024; j += 1024u) lds[j] = 0u;
  __syncthreads();
  const uint32_t start_al = start & ~7u, head = start - start_al, total = head + cnt;
  const uint4* k4 = reinterpret_cast<const uint4*>(keys + (size_t)k * nst + start_al);
  const uint4* i4 = reinterpret_cast<const uint4*>(idx + (size_t)k * nst + start_al);
  const uint32_t groups = (total + 7u) >> 3;
  uint4 a[5], b[5], d[5];
#pragma unroll
  for (int c = 0; c < 5; c++) {
    uint32_t gi = (uint32_t)c * 1024u + t;
    if (mode == 0) gi = min(gi, groups - 1u);           // clamp (as in the kernel)
    else gi = gi < groups ? gi : t;                      // redirect to distinct early groups instead of one address
    a[c] = k4[gi]; b[c] = i4[2 * gi]; d[c] = i4[2 * gi + 1];
  }
  uint32_t x = 0;
#pragma unroll
  for (int c = 0; c < 5; c++) x ^= a[c].x ^ b[c].y ^ d[c].w;
  if (x == 0x1234567u) { out[0] = x; sm[0] = x; }
}
__global__ void __launch_bounds__(256) k_fill(uint4* p, size_t n16, uint32_t v) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256ull) p[i] = make_uint4(v, v + 1, v + 2, (uint32_t)i);
}
// producer emulating k_part_scatter's stores: each block writes runs of `run` consecutive entries (2-byte keys,
// 4-byte indices) to pseudo-random run slots, one element per lane
__global__ void __launch_bounds__(512) k_fill_runs(uint16_t* keys, uint32_t* idx, uint32_t n, uint32_t run) {
  const uint32_t nruns = n / run;
  for (uint32_t r = blockIdx.x; r < nruns; r += gridDim.x) {
    const uint32_t slot = (uint32_t)(((uint64_t)r * 2654435761ull) % nruns);
    for (uint32_t e = threadIdx.x; e < run; e += 512u) { keys[(size_t)slot * run + e] = (uint16_t)(e + r); idx[(size_t)slot * run + e] = e * 3u + r; }
  }
}
int main() {
  const size_t N = 16ull << 20;                       // entries
  uint4 *k4, *i4; uint32_t* out;
  hipMalloc(&k4, N * 2); hipMalloc(&i4, N * 4); hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct cfg { int blocks; size_t lds; bool refill; const char* name; } cfgs[] = {
    {512, 0, false, "512 blocks, no LDS, data cold-ish"}, {512, 151552, false, "512 blocks, 148 KB LDS"},
    {512, 0, true, "512 blocks, no LDS, after fill"}, {512, 151552, true, "512 blocks, 148 KB LDS, after fill"},
    {2048, 0, true, "2048 blocks, no LDS, after fill"}, {2048, 49152, true, "2048 blocks, 48 KB LDS, after fill"},
    {8192, 0, true, "8192 blocks, no LDS, after fill"},
  };
  hipFuncSetAttribute((const void*)k_read, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  for (auto& c : cfgs) {
    const uint32_t gpb = (uint32_t)(N / 8 / c.blocks);
    const int chunks = (gpb + 1023) / 1024;
    float best = 1e9;
    for (int r = 0; r < 4; r++) {
      if (c.refill) { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, k4, N * 2 / 16, r); hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, i4, N * 4 / 16, r); }
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_read, dim3(c.blocks), dim3(1024), c.lds, 0, k4, i4, out, gpb, chunks);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-44s %8.1f us  %6.2f TB/s\n", c.name, best * 1e3, (double)N * 6 / (best * 1e-3) / 1e12);
  }
  {
    const uint32_t P = 32, W = 16, nst = 1u << 20;
    uint32_t hs[33]; for (uint32_t p = 0; p <= P; p++) hs[p] = p * (nst / P) + (p && p < P ? (p * 37u) % 301u : 0u);
    uint32_t* ds; hipMalloc(&ds, sizeof hs); hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_read_b, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    for (int mode = 0; mode < 2; mode++) for (size_t lds : {(size_t)4096, (size_t)151552}) {
      float best = 1e9;
      for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_read_b, dim3(P, W), dim3(1024), lds, 0, (const uint16_t*)k4, (const uint32_t*)i4, out, ds, P, nst, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("variant B mode %d (0 = clamp to last group) lds %6zu: %8.1f us\n", mode, lds, best * 1e3);
    }
  }
  {
    const uint32_t P = 32, W = 16, nst = 1u << 20;
    uint32_t hs[33]; for (uint32_t p = 0; p <= P; p++) hs[p] = p * (nst / P);
    uint32_t* ds; hipMalloc(&ds, sizeof hs); hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice);
    for (uint32_t run : {32u, 128u, 1024u}) {
      float best = 1e9, fill = 0;
      for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fill_runs, dim3(2048), dim3(512), 0, 0, (uint16_t*)k4, (uint32_t*)i4, (uint32_t)N, run);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&fill, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_read_b, dim3(P, W), dim3(1024), 151552, 0, (const uint16_t*)k4, (const uint32_t*)i4, out, ds, P, nst, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("after k_fill_runs(run=%4u: %6.1f us): read %8.1f us\n", run, fill * 1e3, best * 1e3);
    }
  }
  return 0;
}
