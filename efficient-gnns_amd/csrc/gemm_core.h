// fp32 MFMA GEMM building blocks for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, == an fmaf chain).
//
// Block tile BM x BN, 256 threads = 4 waves in a 2 x 2 arrangement, each wave owning
// (BM/2) x (BN/2) as TM x TN MFMA tiles of 32 x 32.  K is consumed in steps of BK = 16 through two
// LDS buffers; global loads for step k+1 are issued before the MFMAs of step k and written to the
// other LDS buffer a quarter of the way through them (one barrier per step).  An operand keeps its global orientation in LDS so
// that staging is always a conflict-free ds_write_b128: k-major operands live as [rows][BK+4] and are read
// with one ds_read_b128 per 4 MFMAs; row-major-in-k operands ([K, rows] in memory) live as [BK][rows+4] and
// are read with four conflict-free ds_read_b32.  Either way lane l holds k = {4h .. 4h+3}, h = l >> 5, and
// MFMA #m of a group consumes element m of both fragments -- a permutation of the k order that A and B
// share, so the sum is unchanged.
// At 64 cycles per MFMA per SIMD the LDS and the staging VALU have an order of magnitude of slack;
// the kernel is paced by the matrix pipe.
#pragma once
#include "common.h"

namespace egnn_gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef EGNN_BK
#define EGNN_BK 16
#endif
#ifndef EGNN_STORE_AFTER
#define EGNN_STORE_AFTER 1  // MFMA group (of 4) of the first k-block after which the prefetched tile is written to LDS
#endif
constexpr int BK = EGNN_BK;     // k-step per barrier (16 or 32)
constexpr int LDS_LD = BK + 4;  // floats per LDS row: (BK+4)*4 B keeps every row 16-byte aligned
constexpr int KQ = BK / 4;      // float4 per k-row segment

constexpr int KMAJOR = 0;   // operand stored [rows, K] (k contiguous)
constexpr int MNMAJOR = 1;  // operand stored [K, rows] (row index contiguous)

struct IdentityXf {
  __device__ __forceinline__ float operator()(float v, int64_t, int64_t) const { return v; }
};

// Stages one operand tile of R rows x BK through registers into LDS.  XF transforms in-range elements
// (value, global row, global k) on the way; out-of-range elements are exact zeros.
// GATHER: storage row i of the operand is row ridx[i] of the matrix at p (a row gather fused into the load: k-major
// operands gather their `rows`, [K,rows] operands gather along k).
template <int R, int MAJOR, bool VEC4, class XF, bool GATHER = false>
struct Stager {
  static constexpr int NV = R * BK / 1024;  // float4 per thread per k-step
  static constexpr int RPP = 256 / KQ;        // k-major: rows covered per pass of the 256 threads
  float v[NV][4];

  // FULL: the whole R x BK tile is in range (block-uniform) -> straight-line vector loads, no per-element guards
  template <bool FULL>
  __device__ __forceinline__ void load_impl(const float* __restrict__ p, int64_t ld, int64_t r0, int64_t rmax, int64_t k0,
                                            int64_t kmax, const XF& xf, const int64_t* __restrict__ ridx) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if constexpr (MAJOR == KMAJOR) {
        const int64_t r = r0 + t / KQ + RPP * i;
        const int64_t k = k0 + (t % KQ) * 4;
        int64_t rs = r;
        if constexpr (GATHER) rs = (FULL || r < rmax) ? ridx[r] : 0;
        const float* q = p + rs * ld + k;
        if constexpr (FULL) {
          float x0, x1, x2, x3;
          if constexpr (VEC4) { const float4 x = *reinterpret_cast<const float4*>(q); x0 = x.x; x1 = x.y; x2 = x.z; x3 = x.w; }
          else { x0 = q[0]; x1 = q[1]; x2 = q[2]; x3 = q[3]; }
          v[i][0] = xf(x0, r, k); v[i][1] = xf(x1, r, k + 1); v[i][2] = xf(x2, r, k + 2); v[i][3] = xf(x3, r, k + 3);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[i][j] = (r < rmax && k + j < kmax) ? xf(q[j], r, k + j) : 0.f;
        }
      } else {
        constexpr int TPR = R / 4;          // threads per k-row
        constexpr int KPP = 256 / TPR;      // k-rows per pass
        const int64_t k = k0 + t / TPR + KPP * i;
        const int64_t r = r0 + (t % TPR) * 4;
        int64_t ks = k;
        if constexpr (GATHER) ks = (FULL || k < kmax) ? ridx[k] : 0;
        const float* q = p + ks * ld + r;
        if constexpr (FULL) {
          float x0, x1, x2, x3;
          if constexpr (VEC4) { const float4 x = *reinterpret_cast<const float4*>(q); x0 = x.x; x1 = x.y; x2 = x.z; x3 = x.w; }
          else { x0 = q[0]; x1 = q[1]; x2 = q[2]; x3 = q[3]; }
          v[i][0] = xf(x0, r, k); v[i][1] = xf(x1, r + 1, k); v[i][2] = xf(x2, r + 2, k); v[i][3] = xf(x3, r + 3, k);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[i][j] = (k < kmax && r + j < rmax) ? xf(q[j], r + j, k) : 0.f;
        }
      }
    }
  }

  // FULLONLY: the caller guarantees that every tile is in range (the slow path is not even compiled: its loop-invariant
  // masks and addresses otherwise get hoisted into the caller's loops and cost ~100 registers)
  template <bool FULLONLY = false>
  __device__ __forceinline__ void load(const float* __restrict__ p, int64_t ld, int64_t r0, int64_t rmax, int64_t k0,
                                       int64_t kmax, const XF& xf, const int64_t* __restrict__ ridx = nullptr) {
    if constexpr (FULLONLY) {
      load_impl<true>(p, ld, r0, rmax, k0, kmax, xf, ridx);
    } else {
      const bool full = (r0 + R <= rmax) && (k0 + BK <= kmax);  // block-uniform
      if (full) load_impl<true>(p, ld, r0, rmax, k0, kmax, xf, ridx);
      else load_impl<false>(p, ld, r0, rmax, k0, kmax, xf, ridx);
    }
  }

  // LDS image: KMAJOR -> [R][LDS_LD] ; MNMAJOR -> [BK][R + 4]   (both fit in R * LDS_LD floats)
  __device__ __forceinline__ void store(float* lds) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const f32x4v x = {v[i][0], v[i][1], v[i][2], v[i][3]};
      if constexpr (MAJOR == KMAJOR) {
        *reinterpret_cast<f32x4v*>(lds + (t / KQ + RPP * i) * LDS_LD + (t % KQ) * 4) = x;
      } else {
        constexpr int TPR = R / 4;
        constexpr int KPP = 256 / TPR;
        *reinterpret_cast<f32x4v*>(lds + (t / TPR + KPP * i) * (R + 4) + (t % TPR) * 4) = x;
      }
    }
  }
};

// MFMA fragment of 32 rows starting at `row0` for k-block kb (8 k values): element m <-> k = kb*8 + 4*(lane>>5) + m
template <int R, int MAJOR>
__device__ __forceinline__ f32x4v load_frag(const float* lds, int row0, int kb, int lane) {
  if constexpr (MAJOR == KMAJOR) {
    return *reinterpret_cast<const f32x4v*>(lds + (row0 + (lane & 31)) * LDS_LD + kb * 8 + (lane >> 5) * 4);
  } else {
    const float* p = lds + (kb * 8 + (lane >> 5) * 4) * (R + 4) + row0 + (lane & 31);
    f32x4v f = {p[0], p[R + 4], p[2 * (R + 4)], p[3 * (R + 4)]};
    return f;
  }
}

template <int BM, int BN>
struct TileShape {
  static constexpr int WM = BM / 2, WN = BN / 2;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int SMEM_FLOATS = 2 * (BM + BN) * LDS_LD;
  static_assert(TM >= 1 && TN >= 1, "tile too small for the 2x2 wave arrangement");
};

// row / column of accumulator register `reg` of MFMA tile (tm, tn) inside the block tile
template <int BM, int BN>
__device__ __forceinline__ int acc_row(int wm, int tm, int reg, int lane) {
  return wm * TileShape<BM, BN>::WM + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
template <int BM, int BN>
__device__ __forceinline__ int acc_col(int wn, int tn, int lane) {
  return wn * TileShape<BM, BN>::WN + tn * 32 + (lane & 31);
}

// The software pipeline of one workgroup: operand tiles travel global -> registers (prefetch) -> LDS (commit, inside
// step) -> MFMA fragments.  mainloop() below drives it for one output tile; kernels that walk several tiles (G-CRD
// forward) drive it themselves so that the first k-step of the next tile is already in flight during an epilogue.
template <int BM, int BN, int AMAJ, int BMAJ, bool VEC4, bool FULLONLY, class XFA, class XFB, bool GA = false, bool GB = false>
struct Pipeline {
  using TS = TileShape<BM, BN>;
  // plain offset arithmetic on the __shared__ base keeps the LDS address space visible to the compiler
  // (an array of buffer pointers indexed by `cur` decays to flat loads, which also drain the global prefetch)
  static constexpr int A_BUF = BM * LDS_LD, B_BUF = BN * LDS_LD, B_OFF = 2 * BM * LDS_LD;
  Stager<BM, AMAJ, VEC4, XFA, GA> sa;
  Stager<BN, BMAJ, VEC4, XFB, GB> sb;
  const int64_t* arows = nullptr;  // GA / GB: storage-row indirection of A / B (see Stager)
  const int64_t* brows = nullptr;

  __device__ __forceinline__ void prefetch(const float* __restrict__ A, int64_t lda, int64_t m0, int64_t M,
                                           const float* __restrict__ B, int64_t ldb, int64_t n0, int64_t N, int64_t k0,
                                           int64_t kend, const XFA& xfa, const XFB& xfb) {
    sa.template load<FULLONLY>(A, lda, m0, M, k0, kend, xfa, arows);
    sb.template load<FULLONLY>(B, ldb, n0, N, k0, kend, xfb, brows);
  }
  __device__ __forceinline__ void commit(float* smem, int buf) const {
    sa.store(smem + buf * A_BUF);
    sb.store(smem + B_OFF + buf * B_BUF);
  }
  // one k-step out of LDS buffer `cur`; with `more`, the prefetched registers are committed to buffer cur^1 on the way
  template <int TM_, int TN_>
  __device__ __forceinline__ void step(f32x16 (&acc)[TM_][TN_], float* smem, int cur, bool more, int lane, int wm, int wn) const {
    static_assert(TM_ == TS::TM && TN_ == TS::TN, "accumulator shape does not match the block tile");
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      f32x4v a[TS::TM], b[TS::TN];
#pragma unroll
      for (int tm = 0; tm < TS::TM; ++tm) a[tm] = load_frag<BM, AMAJ>(smem + cur * A_BUF, wm * TS::WM + tm * 32, kb, lane);
#pragma unroll
      for (int tn = 0; tn < TS::TN; ++tn) b[tn] = load_frag<BN, BMAJ>(smem + B_OFF + cur * B_BUF, wn * TS::WN + tn * 32, kb, lane);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int tm = 0; tm < TS::TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TS::TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][m], b[tn][m], acc[tm][tn], 0, 0, 0);
        // The staged tile goes to the other LDS buffer EARLY in the step (after the first quarter of its MFMAs), not
        // after the last one: the ds_write latency and the barrier that follows are then covered by this wave's own
        // remaining MFMAs instead of draining the matrix pipe (measured with tools/probes/gemm_abl.hip: +16-20 %).
        // (when BOTH operands are [K,rows] the fragments cost 4x the LDS instructions and the early commit gets in
        // their way: dW = X^T dY measured 277 us early vs 247 us late, so that layout commits after the last MFMA)
        constexpr bool kLate = (AMAJ == MNMAJOR && BMAJ == MNMAJOR);
        if (more && (kLate ? (kb == BK / 8 - 1 && m == 3) : (kb == 0 && m == EGNN_STORE_AFTER))) commit(smem, cur ^ 1);
      }
    }
  }
};

// acc += A[m0:m0+BM, kbeg:kend] * B[kbeg:kend, n0:n0+BN]   (all 256 threads must call with equal bounds)
template <int BM, int BN, int AMAJ, int BMAJ, bool VEC4, bool FULLONLY = false, bool GA = false, bool GB = false, class XFA, class XFB,
          int TM_, int TN_>
__device__ __forceinline__ void mainloop(f32x16 (&acc)[TM_][TN_],
                                         const float* __restrict__ A, int64_t lda, int64_t m0, int64_t M,
                                         const float* __restrict__ B, int64_t ldb, int64_t n0, int64_t N,
                                         int64_t kbeg, int64_t kend, const XFA& xfa, const XFB& xfb, float* smem,
                                         const int64_t* arows = nullptr, const int64_t* brows = nullptr) {
  Pipeline<BM, BN, AMAJ, BMAJ, VEC4, FULLONLY, XFA, XFB, GA, GB> pipe;
  pipe.arows = arows;
  pipe.brows = brows;
  const int nk = (int)((kend - kbeg + BK - 1) / BK);
  if (nk <= 0) return;
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;

  pipe.prefetch(A, lda, m0, M, B, ldb, n0, N, kbeg, kend, xfa, xfb);
  pipe.commit(smem, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) pipe.prefetch(A, lda, m0, M, B, ldb, n0, N, kbeg + (int64_t)(kt + 1) * BK, kend, xfa, xfb);
    pipe.step(acc, smem, kt & 1, more, lane, wm, wn);
    __syncthreads();
  }
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
}

}  // namespace egnn_gemm
