// spmm_lab -- stand-alone driver for the aggregation entry points of libegnn_hip.so (plain HIP host program over the C
// ABI, no Python): times schedule / flag variants on the graph files of tools/lab/make_graphs.py, checks every variant
// against a host double-precision gather-sum, and serves as the (fast-starting) workload of the rocprofv3 PMC passes.
//
//   spmm_lab <graph.bin> [K=256] [--only <name-substring>] [--iters N] [--pmc]     (one JSON line per variant)
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

struct Graph {
  int64_t n = 0, nnz = 0, n_comm = 0;
  std::vector<int32_t> rowptr, col, comm_ptr;
  std::vector<float> val;
};

static Graph load(const char* path) {
  Graph g;
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  int64_t h[3];
  if (fread(h, 8, 3, f) != 3) exit(2);
  g.n = h[0]; g.nnz = h[1]; g.n_comm = h[2];
  g.rowptr.resize(g.n + 1); g.col.resize(g.nnz); g.val.resize(g.nnz);
  if (fread(g.rowptr.data(), 4, g.n + 1, f) != (size_t)g.n + 1) exit(2);
  if (fread(g.col.data(), 4, g.nnz, f) != (size_t)g.nnz) exit(2);
  if (fread(g.val.data(), 4, g.nnz, f) != (size_t)g.nnz) exit(2);
  if (g.n_comm > 0) {
    g.comm_ptr.resize(g.n_comm + 1);
    if (fread(g.comm_ptr.data(), 4, g.n_comm + 1, f) != (size_t)g.n_comm + 1) exit(2);
  }
  fclose(f);
  return g;
}

template <typename T>
static T* to_dev(const std::vector<T>& v) {
  T* d = nullptr;
  CK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

struct SegPlan {  // the segment schedule of SparseTensor._seg_plan: `all` covers every row, `hub` only the rows > seg_max
  std::vector<int64_t> seg_all, seg_hub, crow, cptr;
  int64_t slots = 0;
};

static SegPlan make_seg_plan(const Graph& g, int seg_max) {
  SegPlan p;
  std::vector<int64_t> direct;
  p.cptr.push_back(0);
  for (int64_t r = 0; r < g.n; ++r) {
    const int64_t s = g.rowptr[r], e = g.rowptr[r + 1], c = e - s;
    if (c <= seg_max) {
      direct.insert(direct.end(), {s, e, r});
    } else {
      const int64_t ns = (c + seg_max - 1) / seg_max;
      for (int64_t k = 0; k < ns; ++k) {
        const int64_t a = s + k * c / ns, b = s + (k + 1) * c / ns;
        p.seg_hub.insert(p.seg_hub.end(), {a, b, g.n + p.slots});
        ++p.slots;
      }
      p.crow.push_back(r);
      p.cptr.push_back(p.slots);
    }
  }
  p.seg_all = p.seg_hub;
  p.seg_all.insert(p.seg_all.end(), direct.begin(), direct.end());
  return p;
}

struct Variant {
  std::string name;
  int kind;  // 0: segment schedule (round-1 product path), 1: row blocks + hub segments
  int R;
  int flags;
  bool lds;
  bool stats;
  bool comm_blocks;  // variable blocks cut at community boundaries
};

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: spmm_lab graph.bin [K] [--only s] [--iters n] [--pmc]\n"); return 2; }
  const char* path = argv[1];
  int K = 256, iters = 10;
  const char* only = nullptr;
  bool pmc = false, exact = false;
  for (int i = 2; i < argc; ++i) {
    if (!strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i];
    else if (!strcmp(argv[i], "--exact") && i + 1 < argc) { only = argv[++i]; exact = true; }
    else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--pmc")) pmc = true;
    else K = atoi(argv[i]);
  }
  const int seg_max = 64;
  Graph g = load(path);
  const int64_t n = g.n;
  SegPlan sp = make_seg_plan(g, seg_max);
  std::vector<int64_t> rp64(g.rowptr.begin(), g.rowptr.end());
  (void)rp64;

  int32_t *d_rp = to_dev(g.rowptr), *d_col = to_dev(g.col);
  float* d_val = to_dev(g.val);
  int64_t *d_seg_all = to_dev(sp.seg_all), *d_crow = to_dev(sp.crow), *d_cptr = to_dev(sp.cptr);
  std::vector<int32_t> hseg32;
  for (size_t i = 0; i + 2 < sp.seg_hub.size(); i += 3) {
    hseg32.push_back((int32_t)sp.seg_hub[i]); hseg32.push_back((int32_t)sp.seg_hub[i + 1]);
    hseg32.push_back((int32_t)(sp.seg_hub[i + 2] - g.n)); hseg32.push_back(0);
  }
  int32_t* d_hseg = to_dev(hseg32);
  const int64_t n_hseg = (int64_t)hseg32.size() / 4;
  std::vector<float> hx((size_t)n * K);
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& v : hx) v = nd(rng);
  float* d_x = to_dev(hx);
  float *d_y = nullptr, *d_partial = nullptr, *d_stat = nullptr, *d_mean = nullptr, *d_var = nullptr, *d_shift = nullptr;
  CK(hipMalloc(&d_y, (size_t)n * K * 4));
  CK(hipMalloc(&d_partial, std::max<size_t>(sp.slots, 1) * K * 4));
  CK(hipMalloc(&d_stat, (size_t)(n / 32 + 2) * 16 * 2 * K * 4));
  float* d_fold = nullptr;
  CK(hipMalloc(&d_fold, egnn_bn_stats_merge_ws_floats(K) * 4));
  CK(hipMalloc(&d_mean, K * 4));
  CK(hipMalloc(&d_var, K * 4));
  std::vector<float> hshift(K, 0.05f);
  d_shift = to_dev(hshift);
  int32_t* d_win = nullptr;
  CK(hipMalloc(&d_win, (size_t)n * 2 * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));

  // host reference on a sample of rows (double accumulation): every hub row's first 40 plus 3000 others
  std::vector<int64_t> check_rows;
  for (size_t i = 0; i < sp.crow.size() && i < 40; ++i) check_rows.push_back(sp.crow[i]);
  std::uniform_int_distribution<int64_t> ur(0, n - 1);
  for (int i = 0; i < 3000; ++i) check_rows.push_back(ur(rng));
  check_rows.push_back(0);
  check_rows.push_back(n - 1);
  std::vector<double> ref(check_rows.size() * K, 0.0);
  double ref_max = 0;
  for (size_t i = 0; i < check_rows.size(); ++i) {
    const int64_t r = check_rows[i];
    for (int64_t e = g.rowptr[r]; e < g.rowptr[r + 1]; ++e)
      for (int k = 0; k < K; ++k) ref[i * K + k] += (double)g.val[e] * hx[(size_t)g.col[e] * K + k];
    for (int k = 0; k < K; ++k) ref_max = std::max(ref_max, std::fabs(ref[i * K + k]));
  }
  // full-matrix column statistics for the epilogue-stats variants
  const int64_t alg_bytes = egnn_spmm_algorithmic_bytes(n, n, K, g.nnz, 32, 1);
  if (K % 4 != 0) { fprintf(stderr, "K must be a multiple of 4\n"); return 2; }

  std::vector<Variant> vars;
  vars.push_back({"seg_r01", 0, 0, 0, false, false, false});
  for (int R : {32, 64, 128})
    for (int fl : {0, 4})
      vars.push_back({"blk_R" + std::to_string(R) + "_f" + std::to_string(fl), 1, R, fl, false, false, false});
  vars.push_back({"default", 1, 32, 4, false, false, false});   // what ops.spmm_raw launches (EGNN_SPMM_BLK_ROWS = 32, sc1 stores)
  vars.push_back({"blk_R32_f0_stats", 1, 32, 0, false, true, false});
  vars.push_back({"blk_R64_f0_stats", 1, 64, 0, false, true, false});
  vars.push_back({"blk_R128_f0_stats", 1, 128, 0, false, true, false});
  if (K % 32 == 0)
    for (int R : {128, 256, 512}) {
      vars.push_back({"lds_R" + std::to_string(R) + "_f0", 1, R, 0, true, false, false});
      if (g.n_comm > 0) vars.push_back({"ldsc_R" + std::to_string(R) + "_f0", 1, R, 0, true, false, true});
    }
  if (K % 32 == 0) vars.push_back({"lds_R512_f0_stats", 1, 512, 0, true, true, false});

  if (pmc) {   // calibration for FETCH_SIZE / WRITE_SIZE: two 256 MiB device-to-device copies (known bytes) in the same pass
    void *ca = nullptr, *cb = nullptr;
    CK(hipMalloc(&ca, 256u << 20));
    CK(hipMalloc(&cb, 256u << 20));
    CK(hipMemsetAsync(ca, 1, 256u << 20, st));
    for (int i = 0; i < 2; ++i) CK(hipMemcpyAsync(cb, ca, 256u << 20, hipMemcpyDeviceToDevice, st));
    CK(hipStreamSynchronize(st));
  }
  std::vector<float> hy((size_t)n * K);
  for (const Variant& v : vars) {
    if (only && (exact ? v.name != only : v.name.find(only) == std::string::npos)) continue;
    // variable blocks: whole communities merged up to R rows, larger ones split evenly
    std::vector<int32_t> blk;
    int32_t* d_blk = nullptr;
    int64_t n_blk = 0;
    if (v.comm_blocks) {
      blk.push_back(0);
      int32_t open = 0;  // start of the block being filled
      for (int64_t c = 0; c < g.n_comm; ++c) {
        const int32_t cs = g.comm_ptr[c], ce = g.comm_ptr[c + 1];
        if (ce - open <= v.R) continue;                 // still fits: keep merging
        if (cs > open) { blk.push_back(cs); open = cs; }  // close the block in front of this community
        if (ce - open > v.R) {                            // a community larger than a block: split evenly
          const int parts = (ce - cs + v.R - 1) / v.R;
          for (int q = 1; q <= parts; ++q) blk.push_back(cs + (int32_t)((int64_t)(ce - cs) * q / parts));
          open = ce;
        }
      }
      if (blk.back() != (int32_t)n) blk.push_back((int32_t)n);
      n_blk = (int64_t)blk.size() - 1;
      d_blk = to_dev(blk);
    }
    double inblk = 0;
    if (v.lds) {
      int rc = egnn_spmm_blk_window_i32(d_rp, d_col, n, v.R, d_blk, n_blk, d_win, st);
      if (rc) { printf("{\"name\":\"%s\",\"error\":\"window rc=%d\"}\n", v.name.c_str(), rc); continue; }
      std::vector<int32_t> hw((size_t)n * 2);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(hw.data(), d_win, hw.size() * 4, hipMemcpyDeviceToHost));
      int64_t in = 0;
      for (int64_t r = 0; r < n; ++r)
        if (g.rowptr[r + 1] - g.rowptr[r] <= seg_max) in += hw[2 * r + 1] - hw[2 * r];
      inblk = (double)in / (double)g.nnz;
    }
    const int64_t nb = v.comm_blocks ? n_blk : (n + v.R - 1) / std::max(v.R, 1);
    auto run = [&]() -> int {
      if (v.kind == 0)
        return egnn_spmm_csr_seg_f32(n, n, K, d_rp, d_col, 32, d_val, nullptr, nullptr, d_x, K, d_y, K, EGNN_SUM, d_seg_all,
                                     (int64_t)sp.seg_all.size() / 3, d_crow, d_cptr, (int64_t)sp.crow.size(), d_partial, sp.slots, st);
      int rc = egnn_spmm_csr_blk_f32(n, n, K, d_rp, d_col, d_val, nullptr, nullptr, d_x, K, d_y, K, EGNN_SUM, seg_max, v.R, d_blk, n_blk,
                                     v.lds ? d_win : nullptr, d_hseg, n_hseg, d_partial, nullptr, 0, v.stats ? d_stat : nullptr,
                                     v.stats ? d_shift : nullptr, v.flags, st);
      if (rc) return rc;
      const int64_t n_stat = v.stats ? egnn_spmm_blk_stat_rows(n, v.R, v.lds) : 0;
      if (!sp.crow.empty())   // the hub rows: fixed-order sum of the partial slots the block kernel's launch filled
        rc = egnn_spmm_combine_f32(n, K, d_rp, 32, nullptr, d_y, K, EGNN_SUM, d_crow, d_cptr, (int64_t)sp.crow.size(), d_partial, nullptr, 0,
                                   v.stats ? d_stat : nullptr, n_stat, v.stats ? d_shift : nullptr, 0, st);
      if (rc) return rc;
      if (v.stats)
        rc = egnn_bn_stats_merge_f32(d_stat, n_stat + (int64_t)sp.crow.size(), K, nullptr, 0, nullptr, 0, d_shift, n, d_mean, d_var, d_fold,
                                     egnn_bn_stats_merge_ws_floats(K), st);
      return rc;
    };
    CK(hipMemsetAsync(d_y, 0xFF, (size_t)n * K * 4, st));  // NaN pattern: an unwritten row cannot pass the check
    int rc = run();
    if (rc) { printf("{\"name\":\"%s\",\"error\":\"rc=%d\"}\n", v.name.c_str(), rc); continue; }
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hy.data(), d_y, hy.size() * 4, hipMemcpyDeviceToHost));
    double max_err = 0;
    bool finite = true;
    for (size_t i = 0; i < check_rows.size(); ++i)
      for (int k = 0; k < K; ++k) {
        const float got = hy[(size_t)check_rows[i] * K + k];
        if (!std::isfinite(got)) finite = false;
        max_err = std::max(max_err, std::fabs((double)got - ref[i * K + k]));
      }
    // every row written? (cheap: scan one column for the NaN pattern)
    int64_t unwritten = 0;
    for (int64_t r = 0; r < n; ++r)
      if (!std::isfinite(hy[(size_t)r * K]) || !std::isfinite(hy[(size_t)r * K + K - 1])) ++unwritten;
    double stat_err = -1;
    if (v.stats) {
      std::vector<float> hm(K), hv(K);
      CK(hipMemcpy(hm.data(), d_mean, K * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hv.data(), d_var, K * 4, hipMemcpyDeviceToHost));
      stat_err = 0;
      for (int k = 0; k < K; k += 37) {
        double s = 0, ss = 0;
        for (int64_t r = 0; r < n; ++r) { const double y = hy[(size_t)r * K + k]; s += y; ss += y * y; }
        const double m = s / n, var = ss / n - m * m;
        stat_err = std::max(stat_err, std::fabs(hm[k] - m) / (std::fabs(m) + 1e-3));
        stat_err = std::max(stat_err, std::fabs(hv[k] - var) / (var + 1e-12));
      }
    }
    const int reps = pmc ? 3 : iters;
    for (int w = 0; w < (pmc ? 0 : 2); ++w) run();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("{\"name\":\"%s\",\"K\":%d,\"us\":%.1f,\"alg_GBs\":%.1f,\"frac_of_8TBs\":%.4f,\"gather_GBs\":%.0f,\"max_err_over_max\":%.2e,"
           "\"finite\":%d,\"unwritten_rows\":%lld,\"in_block_frac\":%.3f,\"stat_rel_err\":%.2e,\"n_blk\":%lld}\n",
           v.name.c_str(), K, us, alg_bytes / us / 1e3, alg_bytes / us / 1e3 / 8000.0, (double)g.nnz * K * 4 / us / 1e3,
           max_err / ref_max, (int)finite, (long long)unwritten, inblk, stat_err, (long long)nb);
    fflush(stdout);
    if (d_blk) CK(hipFree(d_blk));
  }
  return 0;
}
