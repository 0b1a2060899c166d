// Integer graph-structure kernels (bit-exact vs the torch-sparse / PyG semantics of SURVEY.md 9.1-9.3).
//   ind2ptr         T.ToSparseTensor()            /root/reference/arxiv_pyg/gnn.py:237
//   gcn_norm        GCNConv first forward (cached) /root/reference/arxiv_pyg/gnn.py:28-35,47
// HBM-bound integer/byte work: coalesced wave-per-row sweeps, no GEMM reshaping.
#include "common.h"

namespace {

__global__ void rowptr_kernel(const int64_t* __restrict__ row, int64_t nnz, int64_t n_rows, int64_t* __restrict__ rowptr) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i > n_rows) return;
  // lower_bound(row, i): first position whose row id is >= i
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (row[mid] < i) lo = mid + 1; else hi = mid;
  }
  rowptr[i] = lo;
}

__global__ void narrow_kernel(const int64_t* __restrict__ src, int64_t n, int32_t* __restrict__ dst, int32_t* overflow) {
  bool bad = false;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = src[i];
    bad |= (v < 0 || v > 0x7fffffffLL);
    dst[i] = (int32_t)v;
  }
  if (bad && overflow) atomicOr(overflow, 1);
}

// wave per row: lt = #entries with col < row, nd = #entries with col == row
__device__ __forceinline__ void diag_stats(const int64_t* col, int64_t start, int64_t end, int64_t row, int lane,
                                           int& lt, int& nd) {
  int l = 0, d = 0;
  for (int64_t e = start + lane; e < end; e += 64) {
    const int64_t c = col[e];
    l += c < row;
    d += c == row;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    l += __shfl_xor(l, o);
    d += __shfl_xor(d, o);
  }
  lt = l;
  nd = d;
}

__global__ __launch_bounds__(256) void gcn_count_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                        int64_t n, int64_t* __restrict__ out_count) {
  const int lane = egnn_lane();
  const int64_t row = blockIdx.x * 4LL + egnn_wave_id();
  if (row >= n) return;
  const int64_t start = rowptr[row], end = rowptr[row + 1];
  int lt, nd;
  diag_stats(col, start, end, row, lane, lt, nd);
  if (lane == 0) out_count[row] = (end - start) - nd + 1;
}

__global__ __launch_bounds__(256) void gcn_fill_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                       int64_t n, const int64_t* __restrict__ rowptr_out,
                                                       int64_t* __restrict__ col_out, float* __restrict__ dinv) {
  const int lane = egnn_lane();
  const int64_t row = blockIdx.x * 4LL + egnn_wave_id();
  if (row >= n) return;
  const int64_t start = rowptr[row], end = rowptr[row + 1];
  const int64_t ostart = rowptr_out[row];
  int lt, nd;
  diag_stats(col, start, end, row, lane, lt, nd);
  for (int64_t e = start + lane; e < end; e += 64) {
    const int64_t c = col[e];
    if (c < row) col_out[ostart + (e - start)] = c;
    else if (c > row) col_out[ostart + (e - start) - nd + 1] = c;
  }
  if (lane == 0) {
    col_out[ostart + lt] = row;
    // deg = sum of ones over the row incl. the inserted diagonal (exact in fp32 below 2^24)
    const float deg = (float)((end - start) - nd + 1);
    float d = 1.0f / sqrtf(deg);
    if (isinf(d)) d = 0.f;
    dinv[row] = d;
  }
}

__global__ __launch_bounds__(256) void gcn_values_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                         int64_t n, const float* __restrict__ dinv, float* __restrict__ val) {
  const int lane = egnn_lane();
  const int64_t row = blockIdx.x * 4LL + egnn_wave_id();
  if (row >= n) return;
  const int64_t start = rowptr[row], end = rowptr[row + 1];
  const float dr = 1.0f * dinv[row];
  for (int64_t e = start + lane; e < end; e += 64) val[e] = dr * dinv[col[e]];
}

}  // namespace

extern "C" int egnn_rowptr_from_sorted_rows_i64(const int64_t* row, int64_t nnz, int64_t n_rows, int64_t* rowptr, void* stream) {
  EGNN_CHECK_ARG(nnz >= 0 && n_rows >= 0 && rowptr && (nnz == 0 || row));
  const int64_t blocks = (n_rows + 1 + 255) / 256;
  hipLaunchKernelGGL(rowptr_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, row, nnz, n_rows, rowptr);
  return egnn_launch_status();
}

extern "C" int egnn_narrow_i64_to_i32(const int64_t* src, int64_t n, int32_t* dst, int32_t* overflow, void* stream) {
  EGNN_CHECK_ARG(n >= 0);
  if (n == 0) return EGNN_OK;
  EGNN_CHECK_ARG(src && dst);
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(narrow_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, src, n, dst, overflow);
  return egnn_launch_status();
}

extern "C" int egnn_gcn_norm_count_i64(const int64_t* rowptr, const int64_t* col, int64_t n, int64_t* out_count, void* stream) {
  EGNN_CHECK_ARG(n >= 0);
  if (n == 0) return EGNN_OK;
  EGNN_CHECK_ARG(rowptr && out_count);
  hipLaunchKernelGGL(gcn_count_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rowptr, col, n, out_count);
  return egnn_launch_status();
}

extern "C" int egnn_gcn_norm_fill_i64(const int64_t* rowptr, const int64_t* col, int64_t n, const int64_t* rowptr_out,
                                      int64_t* col_out, float* dinv, void* stream) {
  EGNN_CHECK_ARG(n >= 0);
  if (n == 0) return EGNN_OK;
  EGNN_CHECK_ARG(rowptr && rowptr_out && col_out && dinv);
  hipLaunchKernelGGL(gcn_fill_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rowptr, col, n, rowptr_out, col_out, dinv);
  return egnn_launch_status();
}

extern "C" int egnn_gcn_norm_values_i64(const int64_t* rowptr_out, const int64_t* col_out, int64_t n, const float* dinv,
                                        float* val_out, void* stream) {
  EGNN_CHECK_ARG(n >= 0);
  if (n == 0) return EGNN_OK;
  EGNN_CHECK_ARG(rowptr_out && col_out && dinv && val_out);
  hipLaunchKernelGGL(gcn_values_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rowptr_out, col_out, n, dinv, val_out);
  return egnn_launch_status();
}
