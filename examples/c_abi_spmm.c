/* A plain-C host calling libegnn_hip.so through include/egnn_hip.h -- no Python, no torch.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I include examples/c_abi_spmm.c \
 *       -L efficient-gnns_amd/lib -legnn_hip -L /opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/efficient-gnns_amd/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_spmm && /tmp/c_abi_spmm
 *
 * Builds a small CSR on the host, runs  Y = A X  (sum) and the GCN normalisation entry points on the device and checks
 * both against loops written here.  Exit code 0 = all good.  (Device memory through the HIP runtime C API.) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "egnn_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define EG(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, r_, egnn_error_string(r_)); return 3; } } while (0)

int main(void) {
  const int64_t n = 1000, K = 64;
  /* ring + a hub: row i lists {i-1, i, i+1} (mod n); row 0 additionally lists every 7th node */
  int64_t* rowptr = (int64_t*)malloc((n + 1) * sizeof(int64_t));
  int64_t cap = 3 * n + n / 7 + 8, nnz = 0;
  int64_t* col = (int64_t*)malloc(cap * sizeof(int64_t));
  float* val = (float*)malloc(cap * sizeof(float));
  for (int64_t i = 0; i < n; ++i) {
    rowptr[i] = nnz;
    int64_t c[3] = {(i + n - 1) % n, i, (i + 1) % n};
    /* keep the columns ascending inside the row */
    for (int a = 0; a < 3; ++a) for (int b = a + 1; b < 3; ++b) if (c[b] < c[a]) { int64_t t = c[a]; c[a] = c[b]; c[b] = t; }
    if (i == 0) {
      for (int64_t j = 0; j < n; ++j) if (j % 7 == 0 || j == 1 || j == n - 1) { col[nnz] = j; val[nnz] = 0.25f + (float)(j % 5); ++nnz; }
    } else {
      for (int a = 0; a < 3; ++a) { col[nnz] = c[a]; val[nnz] = 1.0f / (float)(1 + (i + a) % 4); ++nnz; }
    }
  }
  rowptr[n] = nnz;
  float* X = (float*)malloc(n * K * sizeof(float));
  for (int64_t i = 0; i < n * K; ++i) X[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;

  char info[256];
  egnn_build_info(info, sizeof info);
  printf("%s (abi %d)\n", info, egnn_abi_version());

  int64_t *d_rowptr, *d_col;
  float *d_val, *d_X, *d_Y;
  CK(hipMalloc((void**)&d_rowptr, (n + 1) * sizeof(int64_t)));
  CK(hipMalloc((void**)&d_col, nnz * sizeof(int64_t)));
  CK(hipMalloc((void**)&d_val, nnz * sizeof(float)));
  CK(hipMalloc((void**)&d_X, n * K * sizeof(float)));
  CK(hipMalloc((void**)&d_Y, n * K * sizeof(float)));
  CK(hipMemcpy(d_rowptr, rowptr, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_col, col, nnz * sizeof(int64_t), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, val, nnz * sizeof(float), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_X, X, n * K * sizeof(float), hipMemcpyHostToDevice));

  /* Y = A X, every row through the one-wavefront path (no schedule lists), default stream */
  EG(egnn_spmm_csr_f32(n, n, K, d_rowptr, d_col, 64, d_val, NULL, NULL, d_X, K, d_Y, K, EGNN_SUM, NULL, NULL, 0, NULL, 0, NULL, 0, NULL));
  CK(hipDeviceSynchronize());
  float* Y = (float*)malloc(n * K * sizeof(float));
  CK(hipMemcpy(Y, d_Y, n * K * sizeof(float), hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int64_t i = 0; i < n; ++i)
    for (int64_t k = 0; k < K; ++k) {
      double s = 0.0;
      for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) s += (double)val[e] * (double)X[col[e] * K + k];
      const double d = fabs(s - (double)Y[i * K + k]) / (1.0 + fabs(s));
      if (d > worst) worst = d;
    }
  printf("spmm: nnz=%lld  max rel err vs host loop = %.3g\n", (long long)nnz, worst);
  if (!(worst < 1e-5)) return 1;

  /* gcn_norm structure: count pass (rows get their diagonal), same answer as a host loop */
  int64_t* d_cnt;
  CK(hipMalloc((void**)&d_cnt, n * sizeof(int64_t)));
  EG(egnn_gcn_norm_count_i64(d_rowptr, d_col, n, d_cnt, NULL));
  CK(hipDeviceSynchronize());
  int64_t* cnt = (int64_t*)malloc(n * sizeof(int64_t));
  CK(hipMemcpy(cnt, d_cnt, n * sizeof(int64_t), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; ++i) {
    int64_t off = 0;
    for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) off += col[e] != i;
    if (cnt[i] != off + 1) { fprintf(stderr, "gcn_norm count mismatch at row %lld\n", (long long)i); return 1; }
  }
  printf("gcn_norm count: ok\n");
  /* argument errors come back as codes, nothing is launched */
  if (egnn_spmm_csr_f32(n, n, K, NULL, d_col, 64, NULL, NULL, NULL, d_X, K, d_Y, K, EGNN_SUM, NULL, NULL, 0, NULL, 0, NULL, 0, NULL) >= 0) return 1;
  printf("ok\n");
  return 0;
}
