// G-CRD / InfoNCE loss (/root/reference/arxiv_pyg/criterion.py:139-145) on the f32-input MFMA.
//
//   Z = fhat that^T / tau  [S,S] ;  loss = mean_i ( logsumexp_j Z_ij - Z_ii )
//
// Forward: one workgroup owns 128 rows and walks a contiguous range of 128-column blocks; every lane
// keeps an ONLINE (max, sum-exp) pair for each of its 32 accumulator rows, so the soft-max statistics
// never leave registers until the block is done; lanes / waves / column splits are then merged in a
// fixed order.  Z is written once (fp32) for the backward: with the fp32 MFMA at 1/16 of the bf16 rate
// a stored Z (HBM has room for it and the traffic hides under the matrix pipe) is cheaper than the
// flash-style recompute (3 GEMM passes instead of 4).
// Backward: dfhat = c (P - I) that, dthat = c (P - I)^T fhat with P = exp(Z - lse): two GEMMs whose A
// operand is transformed from Z while it is staged into LDS.
#include <stdlib.h>

#include "gemm_core.h"

using namespace egnn_gemm;

namespace {

#ifndef EGNN_NCE_FWD_WAVES
#define EGNN_NCE_FWD_WAVES 1  // waves per SIMD the forward kernel is compiled for (2 spills 47 floats per lane and measures the same)
#endif
constexpr int kMaxSplit = 8;
constexpr int FB = 128;  // forward block tile (rows and columns)

__device__ __forceinline__ void lse_merge(float& m, float& s, float om, float os) {
  const float mn = fmaxf(m, om);
  if (mn == -INFINITY) return;  // both empty
  s = s * expf(m - mn) + os * expf(om - mn);  // exp(-inf - finite) == 0
  m = mn;
}

// FIXED: rows are unit vectors, so every logit is <= 1/tau and that bound serves as the soft-max shift: one exp per
// element and half the per-lane state of the online-max form (which costs a whole wave per SIMD in registers).
template <bool VEC4, bool FIXED>
__global__ __launch_bounds__(256, EGNN_NCE_FWD_WAVES) void nce_fwd_kernel(const float* __restrict__ fhat, int64_t ldf,
                                                      const float* __restrict__ that, int64_t ldt, int64_t Sr, int64_t Sc,
                                                      int64_t diag_off, int64_t P, float inv_tau,
                                                      float* __restrict__ Z, float* __restrict__ zdiag,
                                                      float* __restrict__ pm, float* __restrict__ ps, int nsplit,
                                                      int cb_per_split) {
  using TS = TileShape<FB, FB>;
  __shared__ __attribute__((aligned(16))) float smem[TS::SMEM_FLOATS];
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t i0 = (int64_t)blockIdx.x * FB;
  const int split = blockIdx.y;
  const int64_t ncb = (Sc + FB - 1) / FB;
  const int64_t cb0 = (int64_t)split * cb_per_split;
  int64_t cb1 = cb0 + cb_per_split;
  if (cb1 > ncb) cb1 = ncb;

  float rm[FIXED ? 1 : TS::TM][FIXED ? 1 : 16], rs[TS::TM][16];
  const float shift = inv_tau * 1.0001f;
#pragma unroll
  for (int tm = 0; tm < TS::TM; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (!FIXED) rm[tm][r] = -INFINITY;
      rs[tm][r] = 0.f;
    }

  IdentityXf id;
  for (int64_t cb = cb0; cb < cb1; ++cb) {
    const int64_t j0 = cb * FB;
    f32x16 acc[TS::TM][TS::TN];
    zero_acc(acc);
    mainloop<FB, FB, KMAJOR, KMAJOR, VEC4>(acc, fhat, ldf, i0, Sr, that, ldt, j0, Sc, 0, P, id, id, smem);
#pragma unroll
    for (int tm = 0; tm < TS::TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = i0 + acc_row<FB, FB>(wm, tm, r, lane);
        float zt[TS::TN];
        float mt = -INFINITY;
#pragma unroll
        for (int tn = 0; tn < TS::TN; ++tn) {
          const int64_t c = j0 + acc_col<FB, FB>(wn, tn, lane);
          const float z = acc[tm][tn][r] * inv_tau;
          const bool ok = row < Sr && c < Sc;
          if (ok) {
            if (Z) Z[row * Sc + c] = z;
            if (row + diag_off == c) zdiag[row] = z;
          }
          zt[tn] = ok ? z : -INFINITY;
          mt = fmaxf(mt, zt[tn]);
        }
        if constexpr (FIXED) {
#pragma unroll
          for (int tn = 0; tn < TS::TN; ++tn) rs[tm][r] += expf(zt[tn] - shift);  // exp(-inf) == 0 for masked columns
        } else if (mt > -INFINITY) {
          const float mn = fmaxf(rm[tm][r], mt);
          float s = rs[tm][r] * expf(rm[tm][r] - mn);
#pragma unroll
          for (int tn = 0; tn < TS::TN; ++tn) s += expf(zt[tn] - mn);
          rs[tm][r] = s;
          rm[tm][r] = mn;
        }
      }
    }
  }

  // merge the 32 lanes that share a row (same lane >> 5), then the two column-waves through LDS
  float* lm = smem;            // [2][FB]
  float* lsum = smem + 2 * FB; // [2][FB]
#pragma unroll
  for (int tm = 0; tm < TS::TM; ++tm) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float m, s = rs[tm][r];
      if constexpr (FIXED) {
        m = shift;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) s += __shfl_xor(s, o);
      } else {
        m = rm[tm][r];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          const float om = __shfl_xor(m, o);
          const float os = __shfl_xor(s, o);
          lse_merge(m, s, om, os);
        }
      }
      if ((lane & 31) == 0) {
        const int lr = acc_row<FB, FB>(wm, tm, r, lane);
        lm[wn * FB + lr] = m;
        lsum[wn * FB + lr] = s;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < FB) {
    const int64_t row = i0 + threadIdx.x;
    if (row < Sr) {
      float m = lm[threadIdx.x], s = lsum[threadIdx.x];
      lse_merge(m, s, lm[FB + threadIdx.x], lsum[FB + threadIdx.x]);
      pm[row * nsplit + split] = m;
      ps[row * nsplit + split] = s;
    }
  }
}

__global__ __launch_bounds__(1024) void nce_finalize_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                            const float* __restrict__ zdiag, int64_t S, int nsplit,
                                                            float inv_count, float* __restrict__ lse,
                                                            float* __restrict__ loss) {
  __shared__ float red[1024];
  float local = 0.f;
  for (int64_t row = threadIdx.x; row < S; row += 1024) {
    float m = -INFINITY, s = 0.f;
    for (int k = 0; k < nsplit; ++k) lse_merge(m, s, pm[row * nsplit + k], ps[row * nsplit + k]);
    const float l = m + logf(s);
    lse[row] = l;
    local += l - zdiag[row];
  }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] * inv_count;
}

// A-operand transform of the backward GEMMs: z -> exp(z - lse[i]) - [i == j]
struct NceGradXf {
  const float* lse;
  int lse_by_k;      // 0: i = gemm row (A = Z, k-major);  1: i = gemm k (A = Z^T, stored [k][m])
  int64_t diag_off;  // the positive of Z row i sits in column i + diag_off
  __device__ __forceinline__ float operator()(float v, int64_t r, int64_t k) const {
    const int64_t i = lse_by_k ? k : r;
    const int64_t j = lse_by_k ? r : k;
    return expf(v - lse[i]) - (i + diag_off == j ? 1.f : 0.f);
  }
};

// C[M, N=P] = scale * xf(Z or Z^T) [M,Kd] * B[Kd,P]   (B row-major, n contiguous; Z is [Sr,Sc], ld = Sc)
template <int BM, int AMAJ, bool VEC4>
__global__ __launch_bounds__(256) void nce_bwd_kernel(const float* __restrict__ Z, int64_t ldz, int64_t M, int64_t Kd,
                                                      int64_t diag_off, const float* __restrict__ lse,
                                                      const float* __restrict__ Bm, int64_t P, int64_t ldb,
                                                      float coef, const float* __restrict__ g, float* __restrict__ C,
                                                      int64_t ldc) {
  constexpr int BN = 128;
  using TS = TileShape<BM, BN>;
  __shared__ __attribute__((aligned(16))) float smem[TS::SMEM_FLOATS];
  const int64_t tiles_n = (P + BN - 1) / BN;
  const int64_t m0 = (blockIdx.x / tiles_n) * BM;
  const int64_t n0 = (blockIdx.x % tiles_n) * BN;
  f32x16 acc[TS::TM][TS::TN];
  zero_acc(acc);
  NceGradXf xf{lse, AMAJ == MNMAJOR, diag_off};
  IdentityXf id;
  mainloop<BM, BN, AMAJ, MNMAJOR, VEC4>(acc, Z, ldz, m0, M, Bm, ldb, n0, P, 0, Kd, xf, id, smem);
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const float scale = coef * (g ? g[0] : 1.f);
#pragma unroll
  for (int tn = 0; tn < TS::TN; ++tn) {
    const int64_t c = n0 + acc_col<BM, BN>(wn, tn, lane);
    if (c >= P) continue;
#pragma unroll
    for (int tm = 0; tm < TS::TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + acc_row<BM, BN>(wm, tm, r, lane);
        if (row < M) C[row * ldc + c] = scale * acc[tm][tn][r];
      }
  }
}

template <int AMAJ>
void launch_bwd(const float* Z, int64_t ldz, int64_t M, int64_t Kd, int64_t diag_off, const float* lse, const float* Bm,
                int64_t P, int64_t ldb, float coef, const float* g, float* C, int64_t ldc, bool vec4, hipStream_t st) {
  const int64_t tiles_n = (P + 127) / 128;
  const int64_t t128 = ((M + 127) / 128) * tiles_n;
  // 128-row tiles only when they still give every CU at least two workgroups (one wave per SIMD cannot hide its own
  // staging); below that the 64-row tile doubles the workgroup count.  EGNN_NCE_BM128_MIN_TILES overrides (tuning).
  static const int64_t min_tiles = getenv("EGNN_NCE_BM128_MIN_TILES") ? atoll(getenv("EGNN_NCE_BM128_MIN_TILES")) : 600;
  if (t128 >= min_tiles) {
    if (vec4) hipLaunchKernelGGL((nce_bwd_kernel<128, AMAJ, true>), dim3((unsigned)t128), dim3(256), 0, st, Z, ldz, M, Kd, diag_off, lse, Bm, P, ldb, coef, g, C, ldc);
    else hipLaunchKernelGGL((nce_bwd_kernel<128, AMAJ, false>), dim3((unsigned)t128), dim3(256), 0, st, Z, ldz, M, Kd, diag_off, lse, Bm, P, ldb, coef, g, C, ldc);
  } else {  // fewer than ~one block per CU: halve the row tile
    const int64_t t64 = ((M + 63) / 64) * tiles_n;
    if (vec4) hipLaunchKernelGGL((nce_bwd_kernel<64, AMAJ, true>), dim3((unsigned)t64), dim3(256), 0, st, Z, ldz, M, Kd, diag_off, lse, Bm, P, ldb, coef, g, C, ldc);
    else hipLaunchKernelGGL((nce_bwd_kernel<64, AMAJ, false>), dim3((unsigned)t64), dim3(256), 0, st, Z, ldz, M, Kd, diag_off, lse, Bm, P, ldb, coef, g, C, ldc);
  }
}

}  // namespace

extern "C" size_t egnn_nce_ws_floats(int64_t S) { return (size_t)S * (1 + 2 * kMaxSplit); }

extern "C" int egnn_nce_block_fwd_f32(const float* fhat, int64_t ld_f, const float* that, int64_t ld_t, int64_t Sr, int64_t Sc,
                                      int64_t diag_off, int64_t P, float tau, float inv_count, int unit_rows, float* Z,
                                      float* lse, float* loss, float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(Sr > 0 && Sc > 0 && P > 0 && ld_f >= P && ld_t >= P && tau > 0.f && fhat && that && lse && loss && ws);
  EGNN_CHECK_ARG(diag_off >= 0 && diag_off + Sr <= Sc);
  if (ws_floats < egnn_nce_ws_floats(Sr)) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rb = (Sr + FB - 1) / FB;
  const int64_t ncb = (Sc + FB - 1) / FB;
  int nsplit = (int)((512 + rb - 1) / rb);
  if (nsplit > kMaxSplit) nsplit = kMaxSplit;
  if (nsplit > ncb) nsplit = (int)ncb;
  if (nsplit < 1) nsplit = 1;
  const int cb_per_split = (int)((ncb + nsplit - 1) / nsplit);
  nsplit = (int)((ncb + cb_per_split - 1) / cb_per_split);  // no empty splits
  float* zdiag = ws;
  float* pm = ws + Sr;
  float* ps = pm + Sr * kMaxSplit;
  const bool vec4 = (ld_f % 4 == 0) && (ld_t % 4 == 0) && egnn_aligned16(fhat) && egnn_aligned16(that);
  dim3 grid((unsigned)rb, (unsigned)nsplit);
  // the bound 1/tau is a usable shift while exp(-2/tau) stays a normal float (tau >= 0.025 leaves ample room)
  const bool fixed = unit_rows && (2.f / tau) <= 80.f;
#define EGNN_NCE_FWD(V, F) hipLaunchKernelGGL((nce_fwd_kernel<V, F>), grid, dim3(256), 0, st, fhat, ld_f, that, ld_t, Sr, Sc, diag_off, P, 1.f / tau, Z, zdiag, pm, ps, nsplit, cb_per_split)
  if (vec4) { if (fixed) EGNN_NCE_FWD(true, true); else EGNN_NCE_FWD(true, false); }
  else { if (fixed) EGNN_NCE_FWD(false, true); else EGNN_NCE_FWD(false, false); }
#undef EGNN_NCE_FWD
  hipLaunchKernelGGL(nce_finalize_kernel, dim3(1), dim3(1024), 0, st, pm, ps, zdiag, Sr, nsplit, inv_count, lse, loss);
  return egnn_launch_status();
}

extern "C" int egnn_nce_block_bwd_f32(const float* fhat, int64_t ld_f, const float* that, int64_t ld_t, int64_t Sr, int64_t Sc,
                                      int64_t diag_off, int64_t P, float scale, const float* Z, const float* lse, const float* g,
                                      float* dfhat, int64_t ld_df, float* dthat, int64_t ld_dt, void* stream) {
  EGNN_CHECK_ARG(Sr > 0 && Sc > 0 && P > 0 && ld_f >= P && ld_t >= P && fhat && that && Z && lse);
  EGNN_CHECK_ARG((dfhat == nullptr || ld_df >= P) && (dthat == nullptr || ld_dt >= P));
  hipStream_t st = (hipStream_t)stream;
  const bool vec4 = (ld_f % 4 == 0) && (ld_t % 4 == 0) && (Sc % 4 == 0) && egnn_aligned16(fhat) && egnn_aligned16(that) && egnn_aligned16(Z);
  // dfhat [Sr,P] = scale g (P - I) that ;  dthat [Sc,P] = scale g (P - I)^T fhat
  if (dfhat) launch_bwd<KMAJOR>(Z, Sc, Sr, Sc, diag_off, lse, that, P, ld_t, scale, g, dfhat, ld_df, vec4, st);
  if (dthat) launch_bwd<MNMAJOR>(Z, Sc, Sc, Sr, diag_off, lse, fhat, P, ld_f, scale, g, dthat, ld_dt, vec4, st);
  return egnn_launch_status();
}

extern "C" int egnn_nce_fwd_f32(const float* fhat, const float* that, int64_t S, int64_t P, int64_t ld, float tau, int unit_rows,
                                float* Z, float* lse, float* loss, float* ws, size_t ws_floats, void* stream) {
  return egnn_nce_block_fwd_f32(fhat, ld, that, ld, S, S, 0, P, tau, S > 0 ? 1.f / (float)S : 0.f, unit_rows, Z, lse, loss, ws, ws_floats, stream);
}

extern "C" int egnn_nce_bwd_f32(const float* fhat, const float* that, int64_t S, int64_t P, int64_t ld, float tau,
                                const float* Z, const float* lse, const float* g, float* dfhat, float* dthat,
                                void* stream) {
  EGNN_CHECK_ARG(S > 0 && tau > 0.f);
  return egnn_nce_block_bwd_f32(fhat, ld, that, ld, S, S, 0, P, 1.f / ((float)S * tau), Z, lse, g, dfhat, ld, dthat, ld, stream);
}
