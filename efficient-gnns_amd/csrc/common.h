// Shared helpers for the gfx950 kernels of libegnn_hip.so.  CDNA4 only: wave64, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/egnn_hip.h"

#define EGNN_WAVE 64

#define EGNN_CHECK_ARG(cond) \
  do {                       \
    if (!(cond)) return EGNN_EINVAL; \
  } while (0)

static inline int egnn_launch_status() { return hipGetLastError() == hipSuccess ? EGNN_OK : EGNN_ELAUNCH; }

static inline bool egnn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ int egnn_lane() { return threadIdx.x & 63; }
// wave index inside the workgroup as a provably wave-uniform (SGPR) value
__device__ __forceinline__ int egnn_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ float egnn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float egnn_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
