// Pieces of the fused BatchNorm1d (+ ReLU + dropout) kernels shared by fused_bn.hip and the skinny GEMM that applies them in its
// operand load (gemm_skinny.hip): the parameter block, the counter-based dropout uniform and the per-element forward.
#pragma once
#include "common.h"

namespace egnn_bn {

struct BnParams {
  const float* x; int64_t ldx;
  int64_t n, C;
  const float* mean; const float* var; float eps;
  const float* gamma; const float* beta;
  int relu; float p; unsigned long long seed;
  const unsigned long long* seed_dev;   // nullable: added to `seed` (a per-step value kept on the device: hipGraph replays)
  const int64_t* pick;                  // nullable: the kernel works on the n rows x[pick[i]] (unique ids); y / dy rows are i
};
__device__ __forceinline__ int64_t bn_row(const BnParams& q, int64_t i) { return q.pick ? q.pick[i] : i; }

// counter-based uniform in [0,1) of (seed, element index): two rounds of a 32-bit multiply-xorshift mixer (constants of C. Wellons'
// "lowbias32"), the halves of the seed injected before each round.  32-bit integer VALU only: the mask is RECOMPUTED in every
// backward pass and inside GEMM epilogues, where a 64-bit splitmix (three 64 x 64 multiplies) cost more issue slots than the arithmetic
// around it.  Forward and backward only have to agree with each other (no other RNG is reproduced).
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long idx) {
  unsigned h = mix32((unsigned)idx ^ (unsigned)seed);
  h = mix32(h + (unsigned)(seed >> 32) + (unsigned)(idx >> 32) * 0x9E3779B9U);
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// per-element forward pieces shared by forward and backward
__device__ __forceinline__ void bn_elem(const BnParams& q, float x, float mean, float rstd, float g, float b, int64_t row,
                                        int64_t c, float& xhat, float& gate) {
  xhat = (x - mean) * rstd;
  const float pre = g * xhat + b;
  gate = (q.relu && !(pre > 0.f)) ? 0.f : 1.f;
  if (q.p > 0.f) {
    const float u = uniform01(q.seed + (q.seed_dev ? *q.seed_dev : 0ull), (unsigned long long)(row * q.C + c));
    gate = u >= q.p ? gate / (1.f - q.p) : 0.f;
  }
}

}  // namespace egnn_bn
