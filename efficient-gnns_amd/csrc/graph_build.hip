// On-device graph construction (integer work, bit-exact vs the torch-sparse semantics of SURVEY.md 9.1-9.2):
//   egnn_csr_from_coo_i64   edge list -> CSR sorted by (row, col); optional symmetrise + de-duplicate
//                           (T.ToSparseTensor() and SparseTensor.to_symmetric(), /root/reference/arxiv_pyg/gnn.py:236-240)
//   egnn_csr_transpose_i64  CSR -> CSC with the csr2csc permutation torch-sparse caches for the backward
//                           (/root/reference/arxiv_pyg/gnn.py:192)
// Keys (row * N + col, < 2^63) are radix-sorted over only the bits they use; de-duplication is a device select; row
// pointers come from a binary search over the sorted keys, so no atomics and no dependence on launch order.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace {

constexpr size_t kAlign = 256;
inline size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

int bits_for(unsigned long long v) {  // number of bits needed to represent values in [0, v]
  int b = 1;
  while (b < 64 && (v >> b) != 0) ++b;
  return b;
}

__global__ void coo_keys_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t E, int64_t N, int symmetric,
                                int64_t* __restrict__ keys) {
  for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
    const int64_t r = row[e], c = col[e];
    keys[e] = r * N + c;
    if (symmetric) keys[E + e] = c * N + r;
  }
}

// rowptr[i] = number of keys < i * N (keys sorted ascending, n_keys read from the device when given), col[j] = key % N
__global__ void csr_from_keys_kernel(const int64_t* __restrict__ keys, int64_t n_keys_host, const int64_t* __restrict__ n_keys_dev,
                                     int64_t N, int64_t* __restrict__ rowptr, int64_t* __restrict__ col, int64_t* __restrict__ nnz_out) {
  const int64_t n = n_keys_dev ? n_keys_dev[0] : n_keys_host;
  const int64_t t0 = blockIdx.x * 256LL + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = t0; i <= N; i += stride) {
    const int64_t target = i * N;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    rowptr[i] = lo;
  }
  for (int64_t j = t0; j < n; j += stride) col[j] = keys[j] % N;
  if (t0 == 0 && nnz_out) nnz_out[0] = n;
}

__global__ void iota_kernel(int64_t* __restrict__ v, int64_t n) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v[i] = i;
}

// colptr[c] = number of sorted columns < c ; row_out[j] = row that owns entry perm[j] (upper bound in rowptr)
__global__ void transpose_finish_kernel(const int64_t* __restrict__ sorted_col, const int64_t* __restrict__ perm,
                                        const int64_t* __restrict__ rowptr, int64_t n_rows, int64_t n_cols, int64_t nnz,
                                        int64_t* __restrict__ colptr, int64_t* __restrict__ row_out) {
  const int64_t t0 = blockIdx.x * 256LL + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  for (int64_t c = t0; c <= n_cols; c += stride) {
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sorted_col[mid] < c) lo = mid + 1; else hi = mid;
    }
    colptr[c] = lo;
  }
  for (int64_t j = t0; j < nnz; j += stride) {
    const int64_t e = perm[j];
    int64_t lo = 0, hi = n_rows;  // last row r with rowptr[r] <= e
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (rowptr[mid] <= e) lo = mid; else hi = mid - 1;
    }
    row_out[j] = lo;
  }
}

unsigned grid_for(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

size_t sort_keys_temp(int64_t n, int end_bit) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (int)n, 0, end_bit);
  return bytes;
}
size_t sort_pairs_temp(int64_t n, int end_bit) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (const int64_t*)nullptr,
                                     (int64_t*)nullptr, (int)n, 0, end_bit);
  return bytes;
}
size_t unique_temp(int64_t n) {
  size_t bytes = 0;
  (void)hipcub::DeviceSelect::Unique(nullptr, bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr, (int)n);
  return bytes;
}

}  // namespace

extern "C" size_t egnn_csr_from_coo_ws_bytes(int64_t E, int64_t N, int symmetric) {
  if (E <= 0 || N <= 0) return kAlign;
  const int64_t M = symmetric ? 2 * E : E;
  const int end_bit = bits_for((unsigned long long)N * (unsigned long long)N);
  size_t t = sort_keys_temp(M, end_bit);
  if (symmetric) { const size_t u = unique_temp(M); t = u > t ? u : t; }
  return align_up((size_t)M * 8) * (symmetric ? 3 : 2) + align_up(t) + kAlign;
}

extern "C" int egnn_csr_from_coo_i64(const int64_t* row, const int64_t* col, int64_t E, int64_t N, int symmetric, int64_t* rowptr,
                                     int64_t* col_out, int64_t* nnz_out, void* ws, size_t ws_bytes, void* stream) {
  EGNN_CHECK_ARG(E >= 0 && N >= 0 && rowptr && nnz_out && (E == 0 || (row && col && col_out)));
  EGNN_CHECK_ARG(N < 3037000499LL);                    // row * N + col must fit in int64
  EGNN_CHECK_ARG((symmetric ? 2 * E : E) < 0x7fffffffLL);  // hipCUB item counts are int
  hipStream_t st = (hipStream_t)stream;
  if (ws_bytes < egnn_csr_from_coo_ws_bytes(E, N, symmetric) || !ws) return EGNN_EWORKSPACE;
  const int64_t M = symmetric ? 2 * E : E;
  char* p = (char*)ws;
  int64_t* keys = (int64_t*)p;  p += align_up((size_t)(M > 0 ? M : 1) * 8);
  int64_t* sorted = (int64_t*)p; p += align_up((size_t)(M > 0 ? M : 1) * 8);
  int64_t* uniq = nullptr;
  if (symmetric) { uniq = (int64_t*)p; p += align_up((size_t)(M > 0 ? M : 1) * 8); }
  void* temp = p;
  size_t temp_bytes = ws_bytes - (size_t)(p - (char*)ws);
  if (M > 0) {
    const int end_bit = bits_for((unsigned long long)N * (unsigned long long)N);
    hipLaunchKernelGGL(coo_keys_kernel, dim3(grid_for(E)), dim3(256), 0, st, row, col, E, N, symmetric, keys);
    size_t tb = temp_bytes;
    if (hipcub::DeviceRadixSort::SortKeys(temp, tb, keys, sorted, (int)M, 0, end_bit, st) != hipSuccess) return EGNN_ELAUNCH;
    if (symmetric) {
      tb = temp_bytes;
      if (hipcub::DeviceSelect::Unique(temp, tb, sorted, uniq, nnz_out, (int)M, st) != hipSuccess) return EGNN_ELAUNCH;
    }
  }
  hipLaunchKernelGGL(csr_from_keys_kernel, dim3(grid_for((M > N ? M : N) + 1)), dim3(256), 0, st, symmetric ? uniq : sorted, M,
                     (symmetric && M > 0) ? nnz_out : nullptr, N, rowptr, col_out, nnz_out);
  return egnn_launch_status();
}

extern "C" size_t egnn_csr_transpose_ws_bytes(int64_t nnz, int64_t n_cols) {
  if (nnz <= 0) return kAlign;
  const int end_bit = bits_for((unsigned long long)(n_cols > 0 ? n_cols : 1));
  return align_up((size_t)nnz * 8) * 2 + align_up(sort_pairs_temp(nnz, end_bit)) + kAlign;
}

extern "C" int egnn_csr_transpose_i64(const int64_t* rowptr, const int64_t* col, int64_t n_rows, int64_t n_cols, int64_t nnz,
                                      int64_t* colptr, int64_t* row_out, int64_t* perm, void* ws, size_t ws_bytes, void* stream) {
  EGNN_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && rowptr && colptr && (nnz == 0 || (col && row_out && perm)));
  EGNN_CHECK_ARG(nnz < 0x7fffffffLL);
  hipStream_t st = (hipStream_t)stream;
  if (ws_bytes < egnn_csr_transpose_ws_bytes(nnz, n_cols) || !ws) return EGNN_EWORKSPACE;
  char* p = (char*)ws;
  int64_t* iota = (int64_t*)p;       p += align_up((size_t)(nnz > 0 ? nnz : 1) * 8);
  int64_t* sorted_col = (int64_t*)p; p += align_up((size_t)(nnz > 0 ? nnz : 1) * 8);
  void* temp = p;
  size_t tb = ws_bytes - (size_t)(p - (char*)ws);
  if (nnz > 0) {
    const int end_bit = bits_for((unsigned long long)(n_cols > 0 ? n_cols : 1));
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(nnz)), dim3(256), 0, st, iota, nnz);
    // LSD radix sort is stable: entries of one column keep their CSR (= ascending row) order, as torch-sparse's csr2csc
    if (hipcub::DeviceRadixSort::SortPairs(temp, tb, col, sorted_col, iota, perm, (int)nnz, 0, end_bit, st) != hipSuccess) return EGNN_ELAUNCH;
  }
  hipLaunchKernelGGL(transpose_finish_kernel, dim3(grid_for((nnz > n_cols ? nnz : n_cols) + 1)), dim3(256), 0, st, sorted_col, perm, rowptr,
                     n_rows, n_cols, nnz, colptr, row_out);
  return egnn_launch_status();
}
