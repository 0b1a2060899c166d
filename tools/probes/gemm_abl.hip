// Ablation probe for the fp32 MFMA GEMM mainloop (tools only; not part of libegnn_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm_abl tools/probes/gemm_abl.hip && /tmp/gemm_abl
// C[M,N] = A[M,K] * B[N,K]^T with the production Stager / load_frag; ABL bits switch parts of the loop off
// (results are then wrong on purpose; the timing shows what that part costs).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../efficient-gnns_amd/csrc/gemm_core.h"
using namespace egnn_gemm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int BM, int BN, int ABL>
__global__ __launch_bounds__(256) void abl_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                  int64_t M, int64_t N, int64_t K) {
  using TS = TileShape<BM, BN>;
  __shared__ __attribute__((aligned(16))) float smem[TS::SMEM_FLOATS];
  constexpr int A_BUF = BM * LDS_LD, B_BUF = BN * LDS_LD, B_OFF = 2 * BM * LDS_LD;
  const int64_t tiles_n = N / BN;
  const int64_t m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int64_t lm0 = (ABL & 16) ? 0 : m0, ln0 = (ABL & 16) ? 0 : n0;
  f32x16 acc[TS::TM][TS::TN];
  zero_acc(acc);
  IdentityXf id;
  Stager<BM, KMAJOR, true, IdentityXf> sa;
  Stager<BN, KMAJOR, true, IdentityXf> sb;
  const int nk = (int)(K / BK);
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  sa.load(A, K, lm0, M, 0, K, id);
  sb.load(B, K, ln0, N, 0, K, id);
  sa.store(smem); sb.store(smem + B_OFF);
  sa.store(smem + A_BUF); sb.store(smem + B_OFF + B_BUF);
  __syncthreads();
  f32x4v a[TS::TM], b[TS::TN];
  if (ABL & 4) {
    for (int tm = 0; tm < TS::TM; ++tm) a[tm] = load_frag<BM, KMAJOR>(smem, wm * TS::WM + tm * 32, 0, lane);
    for (int tn = 0; tn < TS::TN; ++tn) b[tn] = load_frag<BN, KMAJOR>(smem + B_OFF, wn * TS::WN + tn * 32, 0, lane);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (!(ABL & 1) && !(ABL & 512) && more) {
      sa.load(A, K, lm0, M, (int64_t)(kt + 1) * BK, K, id);
      sb.load(B, K, ln0, N, (int64_t)(kt + 1) * BK, K, id);
    }
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      if (!(ABL & 4)) {
#pragma unroll
        for (int tm = 0; tm < TS::TM; ++tm) a[tm] = load_frag<BM, KMAJOR>(smem + cur * A_BUF, wm * TS::WM + tm * 32, kb, lane);
#pragma unroll
        for (int tn = 0; tn < TS::TN; ++tn) b[tn] = load_frag<BN, KMAJOR>(smem + B_OFF + cur * B_BUF, wn * TS::WN + tn * 32, kb, lane);
      }
      if (ABL & 128) __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int tm = 0; tm < TS::TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TS::TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][m], b[tn][m], acc[tm][tn], 0, 0, 0);
        if ((ABL & 64) && kb == 0 && m == 1 && more) {   // after 8 of the 32 MFMAs
          if (ABL & 256) {
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) { asm volatile("" :: "v"(sa.v[i][j])); asm volatile("" :: "v"(sb.v[i][j])); }
          } else {
            sa.store(smem + (cur ^ 1) * A_BUF);
            sb.store(smem + B_OFF + (cur ^ 1) * B_BUF);
          }
        }
      }
      if (ABL & 128) __builtin_amdgcn_s_setprio(0);
      if ((ABL & 32) && kb == 0 && more) {
        sa.store(smem + (cur ^ 1) * A_BUF);
        sb.store(smem + B_OFF + (cur ^ 1) * B_BUF);
      }
    }
    if (!(ABL & 1) && !(ABL & 32) && !(ABL & 64) && more) {
      sa.store(smem + (cur ^ 1) * A_BUF);
      sb.store(smem + B_OFF + (cur ^ 1) * B_BUF);
    }
    if (!(ABL & 2)) __syncthreads();
  }
  if (ABL & 8) {
    float s = 0.f;
    for (int tm = 0; tm < TS::TM; ++tm) for (int tn = 0; tn < TS::TN; ++tn) for (int r = 0; r < 16; ++r) s += acc[tm][tn][r];
    if (s == 123.456f) C[0] = s;
    return;
  }
#pragma unroll
  for (int tn = 0; tn < TS::TN; ++tn) {
    const int64_t c = n0 + acc_col<BM, BN>(wn, tn, lane);
#pragma unroll
    for (int tm = 0; tm < TS::TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[tm][tn][r];
        if (ABL & 1024) v = expf(v * 0.01f - 13.f);
        if (ABL & 2048) v = __builtin_amdgcn_exp2f(v * 0.0144f - 18.7f);
        C[(m0 + acc_row<BM, BN>(wm, tm, r, lane)) * N + c] = v;
      }
  }
}

template <int BM, int BN, int ABL>
int run(const char* name, const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((M / BM) * (N / BN));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((abl_kernel<BM, BN, ABL>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e0));
  const int it = 5;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((abl_kernel<BM, BN, ABL>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("%-44s %4dx%-4d abl=%2d  %8.1f us  %6.1f TF/s\n", name, BM, BN, ABL, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
  return 0;
}

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 12288, N = argc > 2 ? atoll(argv[2]) : 8192, K = argc > 3 ? atoll(argv[3]) : 1024;
  printf("M=%lld N=%lld K=%lld\n", (long long)M, (long long)N, (long long)K);
  float *A, *B, *C;
  CK(hipMalloc(&A, M * K * 4)); CK(hipMalloc(&B, N * K * 4)); CK(hipMalloc(&C, M * N * 4));
  std::vector<float> h(M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 255) / 256.f - 0.5f;
  CK(hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), N * K * 4, hipMemcpyHostToDevice));
  run<128, 128, 0>("baseline", A, B, C, M, N, K);
  run<128, 128, 8>("no epilogue", A, B, C, M, N, K);
  run<128, 128, 9>("no epilogue, no global loads/LDS stores", A, B, C, M, N, K);
  run<128, 128, 11>("... and no barrier", A, B, C, M, N, K);
  run<128, 128, 15>("... and no LDS fragment reads (MFMA only)", A, B, C, M, N, K);
  run<128, 128, 24>("no epilogue, all WGs load the same tile", A, B, C, M, N, K);
  run<128, 128, 16>("epilogue, all WGs load the same tile", A, B, C, M, N, K);
  run<128, 128, 40>("no epilogue, LDS store after first MFMA half", A, B, C, M, N, K);
  run<128, 128, 32>("epilogue, LDS store after first MFMA half", A, B, C, M, N, K);
  run<128, 128, 64>("epilogue, LDS store after 8 MFMAs", A, B, C, M, N, K);
  run<128, 128, 64 + 1024>("store after 8, epilogue with expf", A, B, C, M, N, K);
  run<128, 128, 64 + 2048>("store after 8, epilogue with v_exp_f32", A, B, C, M, N, K);
  run<128, 128, 64 + 256>("epilogue, loads but no LDS stores", A, B, C, M, N, K);
  run<128, 128, 64 + 512>("epilogue, LDS stores but no loads", A, B, C, M, N, K);
  run<128, 128, 64 + 128>("epilogue, store after 8 + setprio", A, B, C, M, N, K);
  run<128, 128, 128>("epilogue, baseline + setprio", A, B, C, M, N, K);
  run<128, 64, 32>("epilogue, LDS store after first MFMA half", A, B, C, M, N, K);
  run<128, 128, 10>("no epilogue, no barrier (racy)", A, B, C, M, N, K);
  run<128, 64, 0>("baseline", A, B, C, M, N, K);
  run<128, 64, 15>("MFMA only", A, B, C, M, N, K);
  return 0;
}
