// "Skinny" fp32 GEMMs for gfx950: one matrix dimension is the class count (40 on ogbn-arxiv, <= 64), the long one is
// the node count.  These shapes -- the output layer x W3 (/root/reference/arxiv_pyg/gnn.py:52,84), its dX = dOut W3^T and
// dW3 = X^T dOut of loss.backward() (:192) -- are HBM-bound (one pass over a [N,256] matrix per call); the 128 x 128
// MFMA tile of gemm_core.h spends them in padding and epilogues (38 TFLOP/s, 90 / 219 us).  Here:
//   * v_mfma_f32_16x16x4_f32 tiles (exact fp32, 32-cycle issue), accumulators in registers, the long operand streamed from
//     HBM ONCE with 16-byte loads, the small one (W, <= 64 KB) staged in LDS or read through L1;
//   * "float4 along k / n" trick: a lane's 16-byte load holds 4 consecutive k (or n) values; MFMA step j consumes
//     component j of every lane, i.e. the k (n) index set {4q + j}: a permutation of the reduction (column) order that the
//     other operand mirrors -- sums are unchanged (fp32 re-association), loads and stores stay 16-byte wide;
//   * the row reduction of dW is split over workgroups; partial tiles are combined in a fixed order (deterministic).
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------------------------
// Y[M,N] = X[M,K] W (+ bias),  N <= 64, K % 16 == 0.   W is [K,N] (w_kmajor = 0) or [N,K] (w_kmajor = 1, nn.Linear).
// Workgroup = 4 waves x 64 rows.  sW[K][NP + 4]: the +4 makes the per-step fragment read (rows 4q + j, q = lane >> 4)
// conflict-free for ds_read_b32 (half-wave q = {0,1}: banks 0-15 / 16-31).
template <int NT>
__global__ __launch_bounds__(256, 2) void skinny_fwd_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W,
                                                         int64_t ldw, int w_kmajor, const float* __restrict__ bias,
                                                         float* __restrict__ Y, int64_t ldy, int64_t M, int N, int K, float alpha) {
  extern __shared__ float sW[];
  constexpr int NP = NT * 16, LDW = NP + 4;
  if (!w_kmajor && N % 4 == 0 && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(W) & 15u) == 0) {
    // [K,N] row-major: 16-byte loads, four in flight per thread (a scalar loop spends the workgroup's life in load latency)
    const int nq = N / 4, total = K * nq;
    for (int i0 = threadIdx.x; i0 < total; i0 += 1024) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256;
        v[u] = i < total ? *reinterpret_cast<const float4*>(W + (int64_t)(i / nq) * ldw + (i % nq) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256;
        if (i < total) *reinterpret_cast<float4*>(sW + (i / nq) * LDW + (i % nq) * 4) = v[u];
      }
    }
    for (int i = threadIdx.x; i < K * (NP - N); i += 256) sW[(i / (NP - N)) * LDW + N + i % (NP - N)] = 0.f;
  } else {
    for (int i = threadIdx.x; i < K * NP; i += 256) {
      const int k = w_kmajor ? i % K : i / NP, n = w_kmajor ? i / K : i % NP;   // k-major W: consecutive threads walk k (coalesced)
      float v = 0.f;
      if (n < N) v = w_kmajor ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n];
      sW[k * LDW + n] = v;
    }
  }
  __syncthreads();
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int r16 = lane & 15, q = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * 256 + wave * 64;
  if (m0 >= M) return;
  const float* xp[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int64_t r = m0 + t * 16 + r16;
    if (r >= M) r = M - 1;   // clamped rows are computed and dropped at the store
    xp[t] = X + r * ldx + 4 * q;
  }
  f4 acc[4][NT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = f4{0.f, 0.f, 0.f, 0.f};
  // X is streamed in pairs of k-steps (two 16-byte loads per row group in flight per wave and pair, the next pair issued
  // before the current one is consumed): with one step in flight per wave the kernel was latency-bound at 2.6 TB/s
  float4 a_cur[2][4], a_nxt[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    a_cur[0][t] = *reinterpret_cast<const float4*>(xp[t]);
    a_cur[1][t] = *reinterpret_cast<const float4*>(xp[t] + (K > 16 ? 16 : 0));
  }
  for (int k0 = 0; k0 < K; k0 += 32) {
    if (k0 + 32 < K) {
      const int k1 = k0 + 48 < K ? k0 + 48 : k0 + 32;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a_nxt[0][t] = *reinterpret_cast<const float4*>(xp[t] + k0 + 32);
        a_nxt[1][t] = *reinterpret_cast<const float4*>(xp[t] + k1);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (k0 + 16 * h < K) {   // block-uniform
        const float* wrow = sW + (k0 + 16 * h + 4 * q) * LDW + r16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // step j: k = k0 + 16 h + 4q + j on both operands
          float b[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) b[n] = wrow[j * LDW + n * 16];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float a = j == 0 ? a_cur[h][t].x : (j == 1 ? a_cur[h][t].y : (j == 2 ? a_cur[h][t].z : a_cur[h][t].w));
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = mfma16(a, b[n], acc[t][n]);
          }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < 4; ++t) a_cur[h][t] = a_nxt[h][t];
  }
  // C layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int c = n * 16 + r16;
    if (c >= N) continue;
    const float bv = bias ? bias[c] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + t * 16 + 4 * q + r;
        if (row < M) Y[row * ldy + c] = alpha * acc[t][n][r] + bv;
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Y[M,Nbig] = G[M,Ks] Wt,  Ks <= 64, Nbig % 64 == 0.   Wt given as B[k][n]: b_kmajor = 0 -> stored [Ks, Nbig];
// b_kmajor = 1 -> stored [Nbig, Ks].  Output-write-bound.  A wave owns ONE 64-column block for `rblocks` consecutive
// 64-row blocks: its B fragments (Ks/4 steps x 4 column tiles; tile j = the columns {4 * (lane & 15) + j}) are loaded once
// into registers and reused; per row block it streams G (4-byte loads, L1-resident after the first touch of a row) and
// stores float4s (the four tiles' accumulators of one row are four consecutive columns: 16-byte coalesced stores).
template <int KSTEPS>
__global__ __launch_bounds__(256) void skinny_dx_kernel(const float* __restrict__ G, int64_t ldg, const float* __restrict__ B,
                                                        int64_t ldb, int b_kmajor, float* __restrict__ Y, int64_t ldy, int64_t M,
                                                        int Nbig, int Ks, float alpha, int rblocks) {
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int r16 = lane & 15, q = lane >> 4;
  const int ncb = Nbig / 64;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;   // (row super-block, column block)
  const int64_t sb = item / ncb;
  const int cb = (int)(item % ncb);
  const int64_t mbeg = sb * 64 * rblocks;
  if (mbeg >= M) return;
  float4 bf[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int k = 4 * s + q;
    bf[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < Ks) {
      const int n = cb * 64 + 4 * r16;
      if (!b_kmajor) bf[s] = *reinterpret_cast<const float4*>(B + (int64_t)k * ldb + n);
      else bf[s] = make_float4(B[(int64_t)n * ldb + k], B[(int64_t)(n + 1) * ldb + k], B[(int64_t)(n + 2) * ldb + k], B[(int64_t)(n + 3) * ldb + k]);
    }
  }
  for (int rbi = 0; rbi < rblocks; ++rbi) {
    const int64_t m0 = mbeg + (int64_t)rbi * 64;
    if (m0 >= M) break;
    const float* gp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int64_t r = m0 + t * 16 + r16;
      if (r >= M) r = M - 1;
      gp[t] = G + r * ldg + q;
    }
    f4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[t][n] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {               // one MFMA step: k = 4s + q
      float a[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = (4 * s + q < Ks) ? gp[t][4 * s] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t][0] = mfma16(a[t], bf[s].x, acc[t][0]);
        acc[t][1] = mfma16(a[t], bf[s].y, acc[t][1]);
        acc[t][2] = mfma16(a[t], bf[s].z, acc[t][2]);
        acc[t][3] = mfma16(a[t], bf[s].w, acc[t][3]);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + t * 16 + 4 * q + r;
        if (row < M)
          *reinterpret_cast<float4*>(Y + row * ldy + cb * 64 + 4 * r16) =
              make_float4(alpha * acc[t][0][r], alpha * acc[t][1][r], alpha * acc[t][2][r], alpha * acc[t][3][r]);
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// T[s, b] = sum_r S[r, s] * Bg[r, b]:  S = the narrow matrix [R, Ns] (Ns <= 64), Bg = the wide one [R, Nb] (Nb % 64 == 0), R = the
// node count.  A wave owns a 64-column slice of Bg and a chunk of rows; per 4 rows: one 16-byte load of Bg per lane
// (columns 4 * (lane & 15) .. + 3 of row r0 + (lane >> 4)), NT 4-byte loads of S, 4 * NT MFMAs.  The 4 waves of a workgroup
// take 4 consecutive row chunks of the same slice and add their tiles through LDS in wave order; workgroup partials
// [n_wg][Ns][Nb] are then summed in index order by skinny_dw_reduce_kernel.
template <int NT>
__global__ __launch_bounds__(256) void skinny_dw_kernel(const float* __restrict__ S, int64_t lds_, const float* __restrict__ Bg,
                                                        int64_t ldb, int64_t R, int Ns, int Nb, int64_t rows_per_wave,
                                                        float* __restrict__ part) {
  __shared__ float4 red[3][NT * 4][64];             // waves 1..3: [m-tile * 4 + reg][lane]
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int r16 = lane & 15, q = lane >> 4;
  const int nsl = Nb / 64;
  const int64_t wg = blockIdx.x / nsl;
  const int sl = (int)(blockIdx.x % nsl);
  const int64_t rbeg = (wg * 4 + wave) * rows_per_wave;
  int64_t rend = rbeg + rows_per_wave;
  if (rend > R) rend = R;
  f4 acc[NT][4];
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[m][j] = f4{0.f, 0.f, 0.f, 0.f};
  const float* bcol = Bg + sl * 64 + 4 * r16;
  const float* scol = S + r16;
  float4 b_cur = make_float4(0.f, 0.f, 0.f, 0.f), b_nxt = b_cur;
  float a_cur[NT], a_nxt[NT];
  auto load = [&](int64_t r0, float4& b, float (&a)[NT]) {
    const int64_t r = r0 + q;
    const bool ok = r < rend;
    b = ok ? *reinterpret_cast<const float4*>(bcol + r * ldb) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < NT; ++m) a[m] = (ok && m * 16 + r16 < Ns) ? scol[r * lds_ + m * 16] : 0.f;
  };
  if (rbeg < rend) load(rbeg, b_cur, a_cur);
  for (int64_t r0 = rbeg; r0 < rend; r0 += 4) {
    if (r0 + 4 < rend) load(r0 + 4, b_nxt, a_nxt);
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      acc[m][0] = mfma16(a_cur[m], b_cur.x, acc[m][0]);
      acc[m][1] = mfma16(a_cur[m], b_cur.y, acc[m][1]);
      acc[m][2] = mfma16(a_cur[m], b_cur.z, acc[m][2]);
      acc[m][3] = mfma16(a_cur[m], b_cur.w, acc[m][3]);
    }
    b_cur = b_nxt;
#pragma unroll
    for (int m = 0; m < NT; ++m) a_cur[m] = a_nxt[m];
  }
  // tile (m, j), reg: row s = m*16 + 4q + reg of T, column 4*r16 + j of the slice -> the four j form a float4
  if (wave > 0) {
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][m * 4 + r][lane] = make_float4(acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]);
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float4 v = make_float4(acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]);
        for (int w = 0; w < 3; ++w) {
          const float4 o = red[w][m * 4 + r][lane];
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        const int s = m * 16 + 4 * q + r;
        if (s < Ns) *reinterpret_cast<float4*>(part + ((int64_t)wg * Ns + s) * Nb + sl * 64 + 4 * r16) = v;
      }
  }
}

// C[i] = alpha * sum_w part[w][..] in w order.  Element e = s * Nb + b of T goes to C[s * c_ld_s + b * c_ld_b].
__global__ __launch_bounds__(256) void skinny_dw_reduce_kernel(const float* __restrict__ part, int64_t n_wg, int Ns, int Nb, float alpha,
                                                               float* __restrict__ C, int64_t c_ld_s, int64_t c_ld_b) {
  __shared__ float sh[8][32];
  const int e_lane = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t total = (int64_t)Ns * Nb;
  const int64_t e = (int64_t)blockIdx.x * 32 + e_lane;
  float s = 0.f;
  if (e < total)
    for (int64_t w = g; w < n_wg; w += 8) s += part[w * total + e];
  sh[g][e_lane] = s;
  __syncthreads();
  if (g == 0 && e < total) {
    float t = 0.f;
    for (int k = 0; k < 8; ++k) t += sh[k][e_lane];
    C[(e / Nb) * c_ld_s + (e % Nb) * c_ld_b] = alpha * t;
  }
}

}  // namespace

// Dispatch helpers used by gemm.hip (declared there).  Return EGNN_OK when the shape was taken, 1 when it is not a skinny shape.
int egnn_skinny_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, int w_kmajor, const float* bias, float* Y, int64_t ldy,
                    int64_t M, int64_t N, int64_t K, float alpha, hipStream_t st) {
  if (N > 64 || N < 1 || K % 16 != 0 || K < 16 || K > 1024 || M < 4096 || ldx % 4 != 0 || !egnn_aligned16(X)) return 1;
  const int nt = (int)((N + 15) / 16);
  const size_t shm = (size_t)K * (nt * 16 + 4) * sizeof(float);
  if (shm > 160 * 1024 - 2048) return 1;
  const unsigned grid = (unsigned)((M + 255) / 256);
#define EGNN_SK_FWD(NT)                                                                                                           \
  do {                                                                                                                            \
    if (shm > 65536 && hipFuncSetAttribute((const void*)skinny_fwd_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) \
      return EGNN_ELAUNCH;                                                                                                        \
    hipLaunchKernelGGL(skinny_fwd_kernel<NT>, dim3(grid), dim3(256), shm, st, X, ldx, W, ldw, w_kmajor, bias, Y, ldy, M, (int)N, (int)K, alpha); \
  } while (0)
  switch (nt) {
    case 1: EGNN_SK_FWD(1); break;
    case 2: EGNN_SK_FWD(2); break;
    case 3: EGNN_SK_FWD(3); break;
    default: EGNN_SK_FWD(4); break;
  }
#undef EGNN_SK_FWD
  return egnn_launch_status();
}

int egnn_skinny_dx(const float* G, int64_t ldg, const float* B, int64_t ldb, int b_kmajor, float* Y, int64_t ldy, int64_t M, int64_t Nbig,
                   int64_t Ks, float alpha, hipStream_t st) {
  if (Ks > 64 || Ks < 1 || Nbig % 64 != 0 || Nbig < 64 || Nbig > 1024 || M < 4096 || ldy % 4 != 0 || !egnn_aligned16(Y)) return 1;
  if (!b_kmajor && (ldb % 4 != 0 || !egnn_aligned16(B))) return 1;
  const int rblocks = 4;                                        // 256 rows per wave: the B fragments are loaded once per wave
  const int64_t items = (M + 64 * rblocks - 1) / (64 * rblocks) * (Nbig / 64);
  const unsigned grid = (unsigned)((items + 3) / 4);
  const int ksteps = (int)((Ks + 3) / 4);
#define EGNN_SK_DX(KS) hipLaunchKernelGGL(skinny_dx_kernel<KS>, dim3(grid), dim3(256), 0, st, G, ldg, B, ldb, b_kmajor, Y, ldy, M, (int)Nbig, (int)Ks, alpha, rblocks)
  if (ksteps <= 4) EGNN_SK_DX(4);
  else if (ksteps <= 8) EGNN_SK_DX(8);
  else if (ksteps <= 10) EGNN_SK_DX(10);
  else if (ksteps <= 12) EGNN_SK_DX(12);
  else EGNN_SK_DX(16);
#undef EGNN_SK_DX
  return egnn_launch_status();
}

size_t egnn_skinny_dw_ws_floats(int64_t R, int64_t Ns, int64_t Nb) {
  const int64_t rows_per_wave = 256;
  const int64_t n_wg = (R + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
  return (size_t)(n_wg * Ns * Nb);
}

// C = alpha * S^T Bg (T[s,b]) written with strides (c_ld_s, c_ld_b): (ldc, 1) when the narrow matrix indexes the rows of C,
// (1, ldc) when it indexes the columns
int egnn_skinny_dw(const float* S, int64_t lds_, const float* Bg, int64_t ldb, int64_t R, int64_t Ns, int64_t Nb, float alpha, float* C,
                   int64_t c_ld_s, int64_t c_ld_b, float* ws, size_t ws_floats, hipStream_t st) {
  if (Ns > 64 || Ns < 1 || Nb % 64 != 0 || Nb < 64 || R < 4096 || ldb % 4 != 0 || !egnn_aligned16(Bg)) return 1;
  if (ws == nullptr || ws_floats < egnn_skinny_dw_ws_floats(R, Ns, Nb) || !egnn_aligned16(ws)) return 1;
  const int64_t rows_per_wave = 256;
  const int64_t n_wg = (R + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
  const int nt = (int)((Ns + 15) / 16);
  const unsigned grid = (unsigned)(n_wg * (Nb / 64));
  switch (nt) {
    case 1: hipLaunchKernelGGL(skinny_dw_kernel<1>, dim3(grid), dim3(256), 0, st, S, lds_, Bg, ldb, R, (int)Ns, (int)Nb, rows_per_wave, ws); break;
    case 2: hipLaunchKernelGGL(skinny_dw_kernel<2>, dim3(grid), dim3(256), 0, st, S, lds_, Bg, ldb, R, (int)Ns, (int)Nb, rows_per_wave, ws); break;
    case 3: hipLaunchKernelGGL(skinny_dw_kernel<3>, dim3(grid), dim3(256), 0, st, S, lds_, Bg, ldb, R, (int)Ns, (int)Nb, rows_per_wave, ws); break;
    default: hipLaunchKernelGGL(skinny_dw_kernel<4>, dim3(grid), dim3(256), 0, st, S, lds_, Bg, ldb, R, (int)Ns, (int)Nb, rows_per_wave, ws); break;
  }
  const int64_t total = Ns * Nb;
  hipLaunchKernelGGL(skinny_dw_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, ws, n_wg, (int)Ns, (int)Nb, alpha, C, c_ld_s,
                     c_ld_b);
  return egnn_launch_status();
}
