// gemm3_lab -- stand-alone A/B driver for the "DMA" pipeline of csrc/gemm3.h against the shipped egnn_gemm_f32 (C ABI of
// libegnn_hip.so), all variants interleaved in ONE process on the same data; every result is checked on sampled outputs
// against a float64 product (error unit = sum_k |a_k b_k|).  C[M,N] = A[M,K] * B[N,K]^T (both operands k-contiguous as
// given; F32M variants receive a transposed copy of A).  Whole tiles only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/lab/gemm3_lab.hip -o tools/lab/gemm3_lab \
//         -Lefficient-gnns_amd/lib -legnn_hip -Wl,-rpath,'$ORIGIN/../../efficient-gnns_amd/lib'
//   gemm3_lab [--iters N] [--rounds R] [--only substr]
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

#include "../../efficient-gnns_amd/csrc/gemm3.h"

using namespace egnn_gemm3;
using egnn_gemm::f32x16;

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

struct LabArgs {
  const void* A; int64_t la;      // matrix + leading dimension, or packed planes + number of k-steps
  const void* B; int64_t lb;
  float* C; int64_t ldc;          // [splits][M][ldc]
  int64_t M, N, K, k_per_split;
};

template <int AMODE, int BMODE, int TM, int TN, int BKT, int NB, int OCC, int PIPE, int ABL>
__global__ __launch_bounds__(256, OCC) void lab_kernel(const LabArgs g) {
  using T = Tile<AMODE, BMODE, TM, TN, BKT, NB>;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int64_t tiles_n = g.N / T::BN;
  int64_t tile = blockIdx.x;
  if (tiles_n > 1 && tiles_n <= 8) {   // consecutive column tiles of a row tile on one XCD (as gemm.hip)
    const int64_t tiles = gridDim.x, q = tiles >> 3, rem = tiles & 7, xcd = tile & 7, j = tile >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
  }
  const int64_t m0 = (tile / tiles_n) * T::BM, n0 = (tile % tiles_n) * T::BN;
  const int64_t kbeg = (int64_t)blockIdx.y * g.k_per_split;
  const int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  mainloop<AMODE, BMODE, TM, TN, BKT, NB, PIPE, ABL>(acc, g.A, g.la, m0, g.B, g.lb, n0, kbeg, kend, smem);
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  float* out = g.C + (int64_t)blockIdx.y * g.M * g.ldc;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int64_t c = n0 + wn * 32 * TN + tn * 32 + (lane & 31);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[row * g.ldc + c] = acc[tm][tn][r];
      }
  }
}

struct Shape { const char* name; int64_t M, N, K; int splits; };

// `only` = comma-separated substrings; true when `name` contains one of them (or no filter is given)
static bool selected(const char* name, const char* only) {
  if (!only) return true;
  std::string o(only);
  size_t pos = 0;
  while (pos <= o.size()) {
    const size_t c = o.find(',', pos);
    const std::string tok = o.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
    if (!tok.empty() && strstr(name, tok.c_str())) return true;
    if (c == std::string::npos) break;
    pos = c + 1;
  }
  return false;
}

struct Variant {
  const char* name;
  int amode, bmode, BM, BN, BKT;
  size_t shm;
  void (*launch)(const LabArgs&, dim3, hipStream_t);
  int abl;   // != 0: an ablation (results wrong on purpose; the timing shows what the removed part costs)
};

template <int AMODE, int BMODE, int TM, int TN, int BKT, int NB, int OCC, int PIPE, int ABL>
void launch_v(const LabArgs& g, dim3 grid, hipStream_t st) {
  using T = Tile<AMODE, BMODE, TM, TN, BKT, NB>;
  auto* fn = lab_kernel<AMODE, BMODE, TM, TN, BKT, NB, OCC, PIPE, ABL>;
  static bool once = false;
  if (!once) {
    CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    once = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), T::SMEM_BYTES, st, g);
}

#define VARIANT(name, AM, BMo, TM, TN, BKT, NB, OCC, PIPE) \
  Variant{name, AM, BMo, 64 * TM, 64 * TN, BKT, (size_t)Tile<AM, BMo, TM, TN, BKT, NB>::SMEM_BYTES, launch_v<AM, BMo, TM, TN, BKT, NB, OCC, PIPE, 0>, 0}
// scheduling-hint variants: results are checked like any variant (the template's ABL bits 32 / 64 / 128 only move instructions)
#define SCHED(name, AM, BMo, TM, TN, BKT, NB, OCC, PIPE, HINT) \
  Variant{name, AM, BMo, 64 * TM, 64 * TN, BKT, (size_t)Tile<AM, BMo, TM, TN, BKT, NB>::SMEM_BYTES, launch_v<AM, BMo, TM, TN, BKT, NB, OCC, PIPE, HINT>, 0}
#define ABLATION(name, AM, BMo, TM, TN, BKT, NB, OCC, PIPE, ABL) \
  Variant{name, AM, BMo, 64 * TM, 64 * TN, BKT, (size_t)Tile<AM, BMo, TM, TN, BKT, NB>::SMEM_BYTES, launch_v<AM, BMo, TM, TN, BKT, NB, OCC, PIPE, ABL>, ABL}

int main(int argc, char** argv) {
  int iters = 5, rounds = 3;
  const char* only = nullptr;
  const char* only_shape = nullptr;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--rounds") && i + 1 < argc) rounds = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i];
    else if (!strcmp(argv[i], "--shape") && i + 1 < argc) only_shape = argv[++i];
  }
  const Shape shapes[] = {
      {"nce_fwd_16k_16k_256", 16384, 16384, 256, 1},
      {"layer_169k_256_256", 169216, 256, 256, 1},
      {"layer128_169k_256_128", 169216, 256, 128, 1},
      {"nce_bwd_16k_256_16k_sk8", 16384, 256, 16384, 8},
      {"sq4k", 4096, 4096, 4096, 1},
  };
  const Variant variants[] = {
#ifdef LAB_SCHED_ONLY
      // (round 6) a small build: the shipped forms and their scheduling-hint twins only (-DLAB_SCHED_ONLY; the full list takes > 15 min to compile)
      VARIANT("f32k_pln_128x128_bk32_nb2_p", F32K, PLANES, 2, 2, 32, 2, 2, 1),
      VARIANT("pln_pln_128x128_bk16_nb3_p", PLANES, PLANES, 2, 2, 16, 3, 2, 1),
      VARIANT("f32m_pln_256x256_bk16_nb3_p", F32M, PLANES, 4, 4, 16, 3, 1, 1),
      VARIANT("pln_pln_256x256_bk16_nb3_p", PLANES, PLANES, 4, 4, 16, 3, 1, 1),
#ifdef LAB_LAYER
      // short-K layer shapes (K = 128 / 256): stage size, ring depth and row-tile height around the shipped form (first line)
      VARIANT("f32k_pln_128x128_bk16_nb3_p", F32K, PLANES, 2, 2, 16, 3, 2, 1),
      VARIANT("f32k_pln_128x128_bk16_nb4_p", F32K, PLANES, 2, 2, 16, 4, 2, 1),
      VARIANT("f32k_pln_128x128_bk32_nb2", F32K, PLANES, 2, 2, 32, 2, 2, 0),
      VARIANT("f32k_pln_256x128_bk16_nb3_p", F32K, PLANES, 4, 2, 16, 3, 1, 1),
      VARIANT("f32k_pln_128x256_bk16_nb3_p", F32K, PLANES, 2, 4, 16, 3, 1, 1),
#endif
#if defined(LAB_SCHED32)
      SCHED("sched32_f32k_pln_128x128_bk32_nb2_p", F32K, PLANES, 2, 2, 32, 2, 2, 1, 32),
#endif
#if defined(LAB_SCHED64)
      SCHED("sched64_f32k_pln_128x128_bk32_nb2_p", F32K, PLANES, 2, 2, 32, 2, 2, 1, 64),
#endif
#if defined(LAB_SCHED128)
      SCHED("sched128_f32k_pln_128x128_bk32_nb2_p", F32K, PLANES, 2, 2, 32, 2, 2, 1, 128),
#endif
#if defined(LAB_SCHED32)
      SCHED("sched32_pln_pln_128x128_bk16_nb3_p", PLANES, PLANES, 2, 2, 16, 3, 2, 1, 32),
#endif
#if defined(LAB_SCHED64)
      SCHED("sched64_pln_pln_128x128_bk16_nb3_p", PLANES, PLANES, 2, 2, 16, 3, 2, 1, 64),
#endif
#if defined(LAB_SCHED128)
      SCHED("sched128_pln_pln_128x128_bk16_nb3_p", PLANES, PLANES, 2, 2, 16, 3, 2, 1, 128),
#endif
#if defined(LAB_SCHED32) && defined(LAB_BIG)
      SCHED("sched32_f32m_pln_256x256_bk16_nb3_p", F32M, PLANES, 4, 4, 16, 3, 1, 1, 32),
#endif
#if defined(LAB_SCHED32) && defined(LAB_BIG)
      SCHED("sched32_pln_pln_256x256_bk16_nb3_p", PLANES, PLANES, 4, 4, 16, 3, 1, 1, 32),
#endif
#else
      // name: <A form>_<B form>_<block tile>_bk<k per stage>_nb<LDS stages>[_p = fragments one k-block ahead in registers]
      VARIANT("f32k_f32k_128x128_bk32_nb2", F32K, F32K, 2, 2, 32, 2, 2, 0),
      VARIANT("f32k_f32k_128x128_bk16_nb3", F32K, F32K, 2, 2, 16, 3, 2, 0),
      VARIANT("f32k_f32k_128x128_bk16_nb2_p", F32K, F32K, 2, 2, 16, 2, 2, 1),
      VARIANT("f32k_f32k_128x128_bk16_nb3_p", F32K, F32K, 2, 2, 16, 3, 2, 1),
      VARIANT("f32k_f32k_128x128_bk32_nb2_p", F32K, F32K, 2, 2, 32, 2, 2, 1),
      VARIANT("f32k_pln_128x128_bk32_nb2", F32K, PLANES, 2, 2, 32, 2, 2, 0),
      VARIANT("f32k_pln_128x128_bk16_nb3", F32K, PLANES, 2, 2, 16, 3, 2, 0),
      VARIANT("f32k_pln_128x128_bk16_nb3_p", F32K, PLANES, 2, 2, 16, 3, 2, 1),
      VARIANT("f32k_pln_128x128_bk32_nb2_p", F32K, PLANES, 2, 2, 32, 2, 2, 1),
      VARIANT("pln_pln_128x128_bk16_nb3", PLANES, PLANES, 2, 2, 16, 3, 2, 0),
      VARIANT("pln_pln_128x128_bk16_nb3_p", PLANES, PLANES, 2, 2, 16, 3, 2, 1),
      VARIANT("pln_pln_128x128_bk32_nb2_p", PLANES, PLANES, 2, 2, 32, 2, 1, 1),
      VARIANT("f32k_f32k_256x128_bk16_nb3_p", F32K, F32K, 4, 2, 16, 3, 1, 1),
      VARIANT("f32k_pln_256x128_bk16_nb2_p", F32K, PLANES, 4, 2, 16, 2, 1, 1),
      VARIANT("f32k_pln_256x128_bk16_nb3_p", F32K, PLANES, 4, 2, 16, 3, 1, 1),
      VARIANT("pln_pln_256x128_bk16_nb2_p", PLANES, PLANES, 4, 2, 16, 2, 1, 1),
      VARIANT("pln_pln_256x128_bk16_nb3_p", PLANES, PLANES, 4, 2, 16, 3, 1, 1),
      VARIANT("f32m_pln_128x128_bk16_nb3", F32M, PLANES, 2, 2, 16, 3, 2, 0),
      VARIANT("f32m_pln_128x128_bk16_nb3_p", F32M, PLANES, 2, 2, 16, 3, 2, 1),
      VARIANT("f32m_f32m_128x128_bk16_nb3_p", F32M, F32M, 2, 2, 16, 3, 2, 1),
      // 256 x 256 block tiles, four waves of 128 x 128 (256 accumulator registers, one wave per SIMD): half the L2 -> LDS bytes per FLOP
      VARIANT("pln_pln_256x256_bk16_nb3", PLANES, PLANES, 4, 4, 16, 3, 1, 0),
      VARIANT("pln_pln_256x256_bk16_nb3_p", PLANES, PLANES, 4, 4, 16, 3, 1, 1),
      VARIANT("f32k_pln_256x256_bk16_nb3", F32K, PLANES, 4, 4, 16, 3, 1, 0),
      VARIANT("f32k_pln_256x256_bk16_nb3_p", F32K, PLANES, 4, 4, 16, 3, 1, 1),
      VARIANT("f32k_f32k_256x256_bk16_nb3", F32K, F32K, 4, 4, 16, 3, 1, 0),
      VARIANT("f32k_f32k_256x256_bk16_nb4_p", F32K, F32K, 4, 4, 16, 4, 1, 1),
      VARIANT("f32k_f32k_256x256_bk32_nb2_p", F32K, F32K, 4, 4, 32, 2, 1, 1),
      VARIANT("f32m_pln_256x256_bk16_nb3_p", F32M, PLANES, 4, 4, 16, 3, 1, 1),
      VARIANT("f32m_f32m_256x256_bk16_nb3_p", F32M, F32M, 4, 4, 16, 3, 1, 1),
      VARIANT("f32k_f32k_256x128_bk16_nb3", F32K, F32K, 4, 2, 16, 3, 2, 0),
      // ablations of pln_pln_128x128_bk16_nb3_p (no split VALU in the loop at all): what bounds it?
      ABLATION("abl_pp_setprio", PLANES, PLANES, 2, 2, 16, 3, 2, 1, 16),
      ABLATION("abl_pp_nodma", PLANES, PLANES, 2, 2, 16, 3, 2, 1, 1 | 8),
      ABLATION("abl_pp_nodma_nobarrier", PLANES, PLANES, 2, 2, 16, 3, 2, 1, 1 | 8 | 2),
      ABLATION("abl_pp_nodma_nofrag", PLANES, PLANES, 2, 2, 16, 3, 2, 1, 1 | 8 | 4),
      ABLATION("abl_pp_mfma_only", PLANES, PLANES, 2, 2, 16, 3, 2, 1, 1 | 8 | 4 | 2),
      ABLATION("abl_pp_mfma_only_1wg", PLANES, PLANES, 2, 2, 16, 3, 1, 1, 1 | 8 | 4 | 2),
      ABLATION("abl_pp_dma_nofrag", PLANES, PLANES, 2, 2, 16, 3, 2, 1, 4),
      ABLATION("abl_fp_nodma", F32K, PLANES, 2, 2, 16, 3, 2, 1, 1 | 8),
      ABLATION("abl_fp_setprio", F32K, PLANES, 2, 2, 32, 2, 2, 1, 16),
      ABLATION("abl_pp256_mfma_only", PLANES, PLANES, 4, 2, 16, 3, 1, 1, 1 | 8 | 4 | 2),
#endif
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::mt19937 rng(11);
  for (const Shape& s : shapes) {
    if (!selected(s.name, only_shape)) continue;
    std::vector<float> ha((size_t)s.M * s.K), hb((size_t)s.N * s.K);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : ha) v = nd(rng) * std::exp2f((float)(rng() % 12) - 6.f);   // twelve binades inside every dot product
    for (auto& v : hb) v = nd(rng) * std::exp2f((float)(rng() % 12) - 6.f);
    std::vector<float> hat((size_t)s.M * s.K), hbt((size_t)s.N * s.K);     // [K, M] / [K, N] copies for the F32M variants
    for (int64_t i = 0; i < s.M; ++i)
      for (int64_t k = 0; k < s.K; ++k) hat[(size_t)k * s.M + i] = ha[(size_t)i * s.K + k];
    for (int64_t j = 0; j < s.N; ++j)
      for (int64_t k = 0; k < s.K; ++k) hbt[(size_t)k * s.N + j] = hb[(size_t)j * s.K + k];
    float *da, *db, *dat, *dbt, *dc, *dws = nullptr;
    char *pa = nullptr, *pb = nullptr;
    CK(hipMalloc(&da, ha.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMalloc(&dat, ha.size() * 4)); CK(hipMalloc(&dbt, hb.size() * 4));
    CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dat, hat.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbt, hbt.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dc, (size_t)s.splits * s.M * s.N * 4));
    const size_t pa_cap = planes_bytes(s.M, s.K, 256, 32) + (1 << 20), pb_cap = planes_bytes(s.N, s.K, 256, 32) + (1 << 20);
    CK(hipMalloc(&pa, pa_cap)); CK(hipMalloc(&pb, pb_cap));
    std::vector<float> hc((size_t)s.M * s.N);
    std::vector<std::pair<int64_t, int64_t>> samp;
    std::vector<double> ref, scale;
    for (int t = 0; t < 3000; ++t) {
      const int64_t i = t < 8 ? (t & 1 ? s.M - 1 : 0) : (int64_t)(rng() % s.M), j = t < 8 ? (t & 2 ? s.N - 1 : 0) : (int64_t)(rng() % s.N);
      double r = 0, sc = 0;
      for (int64_t k = 0; k < s.K; ++k) {
        const double a = ha[(size_t)i * s.K + k], b = hb[(size_t)j * s.K + k];
        r += a * b;
        sc += std::fabs(a * b);
      }
      samp.push_back({i, j}); ref.push_back(r); scale.push_back(sc);
    }
    auto check = [&](int splits, double& mean_rel, double& max_rel) {
      CK(hipStreamSynchronize(st));
      std::vector<float> part((size_t)s.M * s.N);
      std::vector<double> got(samp.size(), 0.0);
      for (int sp = 0; sp < splits; ++sp) {
        CK(hipMemcpy(part.data(), dc + (size_t)sp * s.M * s.N, part.size() * 4, hipMemcpyDeviceToHost));
        for (size_t q = 0; q < samp.size(); ++q) got[q] += part[(size_t)samp[q].first * s.N + samp[q].second];
      }
      mean_rel = max_rel = 0;
      for (size_t q = 0; q < samp.size(); ++q) {
        const double rel = std::fabs(got[q] - ref[q]) / (scale[q] + 1e-30);
        mean_rel += rel / samp.size();
        max_rel = std::max(max_rel, rel);
      }
    };
    // the shipped kernel
    const size_t wsf = egnn_gemm_ws_floats(0, 1, s.M, s.N, s.K, s.splits);
    if (wsf) CK(hipMalloc(&dws, wsf * 4));
    struct Row { std::string name; std::vector<double> us; double mean_rel, max_rel; size_t shm; };
    std::vector<Row> rows;
    auto time_it = [&](auto&& run) {
      for (int w = 0; w < 2; ++w) run();
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; ++it) run();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      return (double)ms * 1e3 / iters;
    };
    auto run_ref = [&]() {
      int rc = egnn_gemm_f32(0, 1, s.M, s.N, s.K, 1.f, da, s.K, db, s.K, nullptr, dc, s.N, s.splits, dws, wsf * 4, st);
      if (rc) { fprintf(stderr, "egnn_gemm_f32 rc=%d\n", rc); exit(3); }
    };
    if (selected("shipped", only)) {
      CK(hipMemsetAsync(dc, 0, (size_t)s.M * s.N * 4, st));
      run_ref();
      Row r{"shipped_egnn_gemm_f32", {}, 0, 0, 0};
      check(1, r.mean_rel, r.max_rel);
      rows.push_back(r);
    }
    // the variants: pack what they need, check once
    std::vector<const Variant*> act;
    std::vector<LabArgs> largs;
    std::vector<dim3> grids;
    for (const Variant& v : variants) {
      if (!selected(v.name, only)) continue;
      if (s.M % v.BM || s.N % v.BN || (s.K / s.splits) % v.BKT) continue;
      act.push_back(&v);
    }
    // planes are packed per (RB, BKT) on demand; variants with different packings run in separate groups to bound memory
    for (const Variant* v : act) {
      LabArgs g{};
      g.M = s.M; g.N = s.N; g.K = s.K; g.C = dc; g.ldc = s.N;
      g.k_per_split = s.K / s.splits;
      const int64_t nks = s.K / v->BKT;
      auto pack = [&](const float* X, int64_t rows_, int RB, char* dst) {
        const int64_t total = ((rows_ + RB - 1) / RB) * nks * RB * (v->BKT / 8);
        const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535);
        if (RB == 128 && v->BKT == 16) hipLaunchKernelGGL((pack_planes_kernel<128, 16>), dim3(blocks), dim3(256), 0, st, X, s.K, 1, rows_, s.K, nullptr, nullptr, nullptr, 0.f, dst);
        else if (RB == 128 && v->BKT == 32) hipLaunchKernelGGL((pack_planes_kernel<128, 32>), dim3(blocks), dim3(256), 0, st, X, s.K, 1, rows_, s.K, nullptr, nullptr, nullptr, 0.f, dst);
        else if (RB == 256 && v->BKT == 16) hipLaunchKernelGGL((pack_planes_kernel<256, 16>), dim3(blocks), dim3(256), 0, st, X, s.K, 1, rows_, s.K, nullptr, nullptr, nullptr, 0.f, dst);
        else if (RB == 256 && v->BKT == 32) hipLaunchKernelGGL((pack_planes_kernel<256, 32>), dim3(blocks), dim3(256), 0, st, X, s.K, 1, rows_, s.K, nullptr, nullptr, nullptr, 0.f, dst);
        else { fprintf(stderr, "no packer for RB=%d BK=%d\n", RB, v->BKT); exit(4); }
      };
      if (v->amode == PLANES) { pack(da, s.M, v->BM, pa); g.A = pa; g.la = nks; }
      else if (v->amode == F32M) { g.A = dat; g.la = s.M; }
      else { g.A = da; g.la = s.K; }
      if (v->bmode == PLANES) { pack(db, s.N, v->BN, pb); g.B = pb; g.lb = nks; }
      else if (v->bmode == F32M) { g.B = dbt; g.lb = s.N; }
      else { g.B = db; g.lb = s.K; }
      const dim3 grid((unsigned)((s.M / v->BM) * (s.N / v->BN)), (unsigned)s.splits);
      CK(hipMemsetAsync(dc, 0, (size_t)s.splits * s.M * s.N * 4, st));
      v->launch(g, grid, st);
      CK(hipStreamSynchronize(st));
      CK(hipGetLastError());
      Row r{v->name, {}, 0, 0, v->shm};
      if (!v->abl) check(s.splits, r.mean_rel, r.max_rel);
      else r.mean_rel = r.max_rel = -1;
      // interleaved timing rounds happen below; planes are re-packed per variant there as well (outside the timed region)
      rows.push_back(r);
      largs.push_back(g);
      grids.push_back(grid);
    }
    for (int rd = 0; rd < rounds; ++rd) {
      size_t ri = 0;
      if (selected("shipped", only)) rows[ri++].us.push_back(time_it(run_ref));
      for (size_t vi = 0; vi < act.size(); ++vi, ++ri) {
        const Variant* v = act[vi];
        // re-pack: variants with the same mode but another (RB, BK) share the plane buffers
        LabArgs g = largs[vi];
        const int64_t nks = s.K / v->BKT;
        auto pack = [&](const float* X, int64_t rows_, int RB, char* dst) {
          const int64_t total = ((rows_ + RB - 1) / RB) * nks * RB * (v->BKT / 8);
          const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535);
          if (RB == 128 && v->BKT == 16) hipLaunchKernelGGL((pack_planes_kernel<128, 16>), dim3(blocks), dim3(256), 0, st, X, s.K, 1, rows_, s.K, nullptr, nullptr, nullptr, 0.f, dst);
          else if (RB == 128 && v->BKT == 32) hipLaunchKernelGGL((pack_planes_kernel<128, 32>), dim3(blocks), dim3(256), 0, st, X, s.K, 1, rows_, s.K, nullptr, nullptr, nullptr, 0.f, dst);
          else if (RB == 256 && v->BKT == 16) hipLaunchKernelGGL((pack_planes_kernel<256, 16>), dim3(blocks), dim3(256), 0, st, X, s.K, 1, rows_, s.K, nullptr, nullptr, nullptr, 0.f, dst);
          else hipLaunchKernelGGL((pack_planes_kernel<256, 32>), dim3(blocks), dim3(256), 0, st, X, s.K, 1, rows_, s.K, nullptr, nullptr, nullptr, 0.f, dst);
        };
        if (v->amode == PLANES) pack(da, s.M, v->BM, pa);
        if (v->bmode == PLANES) pack(db, s.N, v->BN, pb);
        CK(hipStreamSynchronize(st));
        rows[ri].us.push_back(time_it([&]() { v->launch(g, grids[vi], st); }));
      }
    }
    for (const Row& r : rows) {
      std::vector<double> u = r.us;
      std::sort(u.begin(), u.end());
      const double med = u.empty() ? 0 : u[u.size() / 2], mn = u.empty() ? 0 : u[0];
      printf("{\"shape\": \"%s\", \"M\": %ld, \"N\": %ld, \"K\": %ld, \"splits\": %d, \"variant\": \"%s\", \"lds\": %zu, \"us_med\": %.1f, \"us_min\": %.1f, "
             "\"tf_med\": %.1f, \"frac_of_417\": %.3f, \"mean_err\": %.3g, \"max_err\": %.3g}\n",
             s.name, (long)s.M, (long)s.N, (long)s.K, s.splits, r.name.c_str(), r.shm, med, mn, med > 0 ? 2.0 * s.M * s.N * s.K / med * 1e-6 : 0.0,
             med > 0 ? 2.0 * s.M * s.N * s.K / med * 1e-6 / 416.7 : 0.0, r.mean_rel, r.max_rel);
    }
    fflush(stdout);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dat); (void)hipFree(dbt); (void)hipFree(dc); (void)hipFree(pa); (void)hipFree(pb);
    if (dws) { (void)hipFree(dws); dws = nullptr; }
  }
  return 0;
}
