// gemm_lab -- stand-alone driver for egnn_gemm_f32 (plain HIP host program over the C ABI, no Python): times the shapes
// of the GCN student's epoch and reports the error of every result against a float64 product on sampled outputs,
// normalised by sum_k |a_k b_k| (the natural scale of fp32 rounding).  Run once per pipeline:
//     EGNN_GEMM_PIPE=f32 gemm_lab     (f32-input MFMA)        gemm_lab     (bf16-pipe split products, the default)
//   gemm_lab [--iters N] [--only name]        one JSON line per shape
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/egnn_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

struct Shape {
  const char* name;
  int ta, tb;
  int64_t M, N, K;
  int split_k;
};

int main(int argc, char** argv) {
  int iters = 10;
  const char* only = nullptr;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i];
  }
  const int64_t n = 169343;
  const Shape shapes[] = {
      {"xW1", 0, 1, n, 256, 128, 1},      {"xW2", 0, 1, n, 256, 256, 1},     {"xW2_kn", 0, 0, n, 256, 256, 1},
      {"dX2", 0, 0, n, 256, 256, 1},      {"dW2", 1, 0, 256, 256, n, 64},    {"dW1", 1, 0, 128, 256, n, 64}, {"dW2_sk128", 1, 0, 256, 256, n, 128}, {"dW2_sk192", 1, 0, 256, 256, n, 192}, {"dWt_sk42", 1, 0, 256, 752, 90941, 42}, {"dWt_sk84", 1, 0, 256, 752, 90941, 84},
      {"proj_s", 0, 1, 16384, 128, 256, 1}, {"proj_t", 0, 1, 16384, 128, 768, 1}, {"sq4k", 0, 1, 4096, 4096, 4096, 1},
      {"small", 0, 1, 300, 200, 72, 1},   {"small_tn", 1, 0, 130, 257, 1000, 3}, {"small_nn", 0, 0, 129, 130, 50, 1},
      {"small_tt", 1, 1, 200, 140, 90, 1}, {"proj_t750", 0, 1, 90941, 256, 750, 1}, {"small_pb", 0, 1, 1000, 200, 72, 1}, {"small_pb_kn", 0, 0, 1203, 136, 200, 1},
      {"small_pb_t", 1, 0, 700, 130, 100, 1}, {"small_pb_sk", 0, 0, 900, 256, 1000, 3},
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::mt19937 rng(11);
  const char* pipe = getenv("EGNN_GEMM_PIPE") ? getenv("EGNN_GEMM_PIPE") : "split";
  for (const Shape& s : shapes) {
    if (only && !strstr(s.name, only)) continue;
    const int64_t a_rows = s.ta ? s.K : s.M, a_cols = s.ta ? s.M : s.K;
    const int64_t b_rows = s.tb ? s.N : s.K, b_cols = s.tb ? s.K : s.N;
    std::vector<float> ha((size_t)a_rows * a_cols), hb((size_t)b_rows * b_cols), hbias(s.N), hc((size_t)s.M * s.N);
    std::normal_distribution<float> nd(0.f, 1.f);
    // a wide dynamic range inside every dot product: exercises the low planes of the split
    for (auto& v : ha) v = nd(rng) * std::exp2f((float)(rng() % 12) - 6.f);
    for (auto& v : hb) v = nd(rng) * std::exp2f((float)(rng() % 12) - 6.f);
    for (auto& v : hbias) v = nd(rng);
    float *da, *db, *dbias, *dc, *dws = nullptr;
    CK(hipMalloc(&da, ha.size() * 4));
    CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMalloc(&dbias, hbias.size() * 4));
    CK(hipMalloc(&dc, hc.size() * 4));
    CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
    const size_t wsf = egnn_gemm_ws_floats(s.ta, s.tb, s.M, s.N, s.K, s.split_k);
    if (wsf) CK(hipMalloc(&dws, wsf * 4));
    const bool use_bias = s.split_k == 1;
    auto run = [&]() {
      return egnn_gemm_f32(s.ta, s.tb, s.M, s.N, s.K, 1.f, da, a_cols, db, b_cols, use_bias ? dbias : nullptr, dc, s.N, s.split_k, dws,
                           wsf * 4, st);
    };
    int rc = run();
    CK(hipStreamSynchronize(st));
    if (rc != 0) { printf("{\"shape\": \"%s\", \"rc\": %d}\n", s.name, rc); continue; }
    CK(hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost));
    // sampled float64 check
    double max_rel = 0, sum_rel = 0;
    const int samples = 4000;
    for (int t = 0; t < samples; ++t) {
      const int64_t i = t < 8 ? (t & 1 ? s.M - 1 : 0) : (int64_t)(rng() % s.M), j = t < 8 ? (t & 2 ? s.N - 1 : 0) : (int64_t)(rng() % s.N);
      double ref = use_bias ? hbias[j] : 0.0, scale = 0;
      for (int64_t k = 0; k < s.K; ++k) {
        const double a = s.ta ? ha[(size_t)k * a_cols + i] : ha[(size_t)i * a_cols + k];
        const double b = s.tb ? hb[(size_t)j * b_cols + k] : hb[(size_t)k * b_cols + j];
        ref += a * b;
        scale += std::fabs(a * b);
      }
      const double rel = std::fabs(hc[(size_t)i * s.N + j] - ref) / (scale + 1e-30);
      max_rel = std::max(max_rel, rel);
      sum_rel += rel;
    }
    for (int w = 0; w < 2; ++w) run();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it) run();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    printf("{\"pipe\": \"%s\", \"shape\": \"%s\", \"ta\": %d, \"tb\": %d, \"M\": %ld, \"N\": %ld, \"K\": %ld, \"split_k\": %d, \"us\": %.1f, "
           "\"tflops\": %.1f, \"max_err_over_sum_abs\": %.3g, \"mean_err_over_sum_abs\": %.3g}\n",
           pipe, s.name, s.ta, s.tb, (long)s.M, (long)s.N, (long)s.K, s.split_k, us, 2.0 * s.M * s.N * s.K / us * 1e-6, max_rel,
           sum_rel / samples);
    fflush(stdout);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dbias); (void)hipFree(dc);
    if (dws) (void)hipFree(dws);
  }
  return 0;
}
