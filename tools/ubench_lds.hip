// ubench_lds.hip -- LDS atomic / scatter rates on gfx950: ds_add_u32, ds_add_rtn_u32, plain ds_write/ds_read
// with pseudo-random addresses over NB bins, 1024-thread blocks, 1 block per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 256
template <int MODE, int NB>
__global__ void __launch_bounds__(1024) k(uint32_t* out, uint32_t seed) {
  __shared__ uint32_t h[NB];
  for (int i = threadIdx.x; i < NB; i += 1024) h[i] = 0;
  __syncthreads();
  uint32_t x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x, acc = 0;
  for (int it = 0; it < ITERS; it++) {
    x = x * 1664525u + 1013904223u;
    const uint32_t b = (x >> 10) & (NB - 1);
    if (MODE == 0) atomicAdd(&h[b], 1u);
    else if (MODE == 1) acc += atomicAdd(&h[b], 1u);
    else if (MODE == 2) h[b] = x;
    else acc += h[b];
  }
  __syncthreads();
  if (acc == 0x12345u) out[0] = acc + h[threadIdx.x % NB];
  if (threadIdx.x == 0 && seed == 77) out[1] = h[3];
}
template <int MODE, int NB> void run(const char* name, uint32_t* d, int cus) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NB>), dim3(cus), dim3(1024), 0, 0, d, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<MODE, NB>), dim3(cus), dim3(1024), 0, 0, d, 1u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)ITERS * 16 * 5;     // wave-instructions per CU
  printf("%-28s NB=%6d  %8.1f ns per wave-instr per CU  (%.2f lanes/ns/CU)\n", name, NB, ms * 1e6 / winstr, 64.0 * winstr / (ms * 1e6));
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  uint32_t* d; hipMalloc(&d, 64);
  const int cus = prop.multiProcessorCount;
  run<0, 32>("ds_add_u32", d, cus); run<0, 1024>("ds_add_u32", d, cus); run<0, 8192>("ds_add_u32", d, cus);
  run<1, 32>("ds_add_rtn_u32", d, cus); run<1, 1024>("ds_add_rtn_u32", d, cus); run<1, 8192>("ds_add_rtn_u32", d, cus);
  run<2, 1024>("ds_write_b32 random", d, cus); run<2, 8192>("ds_write_b32 random", d, cus);
  run<3, 1024>("ds_read_b32 random", d, cus); run<3, 8192>("ds_read_b32 random", d, cus);
  return 0;
}
