// probe.hip.hpp -- the hardware-queue probe kernel of te_msm.hip (classify_streams_by_queue).  Kept apart from
// kernels.hip.hpp: it is not an MSM stage, and bench.py hashes the stage sources to decide whether a committed counter
// profile (profiles/pmc_traffic.json) still describes the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace te {

// the hardware-queue probe: one wave that waits `ticks` of the constant-rate wall clock (bounded: it always ends).
// Two of these on two streams take as long as one when the streams sit on different hardware queues, twice as long when
// the runtime put them on the same one.
__global__ void __launch_bounds__(64) k_spin(unsigned long long ticks, uint32_t* out) {
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  while ((unsigned long long)wall_clock64() - t0 < ticks) { }
  if (out && threadIdx.x == 0) out[0] = 1u;
}

}  // namespace te
