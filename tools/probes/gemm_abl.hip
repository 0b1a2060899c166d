// Ablation probe for the fp32 MFMA GEMM mainloop (tools only; not part of libegnn_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm_abl tools/probes/gemm_abl.hip && /tmp/gemm_abl
// C[M,N] = A[M,K] * B[N,K]^T with the production Stager / load_frag; ABL bits switch parts of the loop off
// (results are then wrong on purpose; the timing shows what that part costs).
#include <cstdio>
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
  f32x16 acc[TS::TM][TS::TN];
  zero_acc(acc);
  IdentityXf id;
  Stager<BM, KMAJOR, true, IdentityXf> sa;
  Stager<BN, KMAJOR, true, IdentityXf> sb;
  const int nk = (int)(K / BK);
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  sa.load(A, K, m0, M, 0, K, id);
  sb.load(B, K, n0, N, 0, K, id);
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
    if (!(ABL & 1) && more) {
      sa.load(A, K, m0, M, (int64_t)(kt + 1) * BK, K, id);
      sb.load(B, K, n0, N, (int64_t)(kt + 1) * BK, K, id);
    }
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      if (!(ABL & 4)) {
#pragma unroll
        for (int tm = 0; tm < TS::TM; ++tm) a[tm] = load_frag<BM, KMAJOR>(smem + cur * A_BUF, wm * TS::WM + tm * 32, kb, lane);
#pragma unroll
        for (int tn = 0; tn < TS::TN; ++tn) b[tn] = load_frag<BN, KMAJOR>(smem + B_OFF + cur * B_BUF, wn * TS::WN + tn * 32, kb, lane);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int tm = 0; tm < TS::TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TS::TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][m], b[tn][m], acc[tm][tn], 0, 0, 0);
    }
    if (!(ABL & 1) && more) {
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
      for (int r = 0; r < 16; ++r) C[(m0 + acc_row<BM, BN>(wm, tm, r, lane)) * N + c] = acc[tm][tn][r];
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

int main() {
  const int64_t M = 12288, N = 8192, K = 1024;
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
  run<128, 128, 12>("no epilogue, no frag reads (loads+barrier)", A, B, C, M, N, K);
  run<128, 128, 10>("no epilogue, no barrier (racy)", A, B, C, M, N, K);
  run<128, 64, 0>("baseline", A, B, C, M, N, K);
  run<128, 64, 15>("MFMA only", A, B, C, M, N, K);
  return 0;
}
