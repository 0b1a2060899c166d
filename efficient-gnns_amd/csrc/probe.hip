// Diagnostic entry point: the request-path ceiling of the neighbour aggregation.
//
// egnn_probe_gather_lines_f32 replays the GATHER STREAM of one egnn_spmm_csr_blk_f32 call and nothing else: for every stored
// entry e of the adjacency, the 128-byte column slice s of source row col[e] of X is read with the kernel's own binding
// (an 8-lane sub-group x 16 bytes per line; slice s of every row handled by workgroups with blockIdx % n_slices == s, i.e.
// on one XCD / one L2).  No row pointers, no values, no reduction tree, no Y: what is left is the cost of putting
// nnz * K * 4 bytes of randomly addressed 128-byte lines through TA -> TCP -> TCC (-> Infinity Cache / HBM).  bench.py times it
// next to the real kernel (roofline.gather_ceiling_GBs): the aggregation cannot gather faster than this on the same graph,
// whatever its schedule.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void probe_gather_kernel(const float* __restrict__ X, int64_t ldx, const int* __restrict__ col, int64_t nnz,
                                                           int n_slices, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63, sub = lane >> 3, li = lane & 7;
  const int slice = blockIdx.x % n_slices;
  const int64_t wg = blockIdx.x / n_slices, n_wg = (gridDim.x + n_slices - 1 - slice) / n_slices;   // workgroups of this slice
  const int64_t group = (wg * 4 + (threadIdx.x >> 6)) * 8 + sub, n_groups = n_wg * 32;                // 8-lane sub-groups of this slice
  const float* base = X + slice * 32 + li * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t e = group; e < nnz; e += n_groups * UNROLL) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int64_t j = e + u * n_groups;
      if (j >= nnz) j = e;
      v[u] = *reinterpret_cast<const f32x4*>(base + (int64_t)col[j] * ldx);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;   // never true for finite data: keeps the loads alive
}

}  // namespace

extern "C" int egnn_probe_gather_lines_f32(const float* X, int64_t ldx, int64_t n_src, int64_t K, const int32_t* col, int64_t nnz,
                                           int blocks_per_slice, int loads_in_flight, float* sink, void* stream) {
  EGNN_CHECK_ARG(X && col && sink && K > 0 && K % 32 == 0 && ldx >= K && ldx % 4 == 0 && n_src > 0 && nnz >= 0 && blocks_per_slice > 0);
  if (!egnn_aligned16(X)) return EGNN_EALIGN;
  if (nnz == 0) return EGNN_OK;
  const int n_slices = (int)(K / 32);
  const dim3 grid((unsigned)(n_slices * blocks_per_slice));
  if (loads_in_flight >= 16) hipLaunchKernelGGL(probe_gather_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, col, nnz, n_slices, sink);
  else if (loads_in_flight >= 8) hipLaunchKernelGGL(probe_gather_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, col, nnz, n_slices, sink);
  else hipLaunchKernelGGL(probe_gather_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, col, nnz, n_slices, sink);
  return egnn_launch_status();
}
