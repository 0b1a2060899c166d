// G-CRD / InfoNCE loss (/root/reference/arxiv_pyg/criterion.py:139-145) on the f32-input MFMA.
//
//   Z = fhat that^T / tau  [S,S] ;  loss = mean_i ( logsumexp_j Z_ij - Z_ii )
//
// Forward: one workgroup owns 128 rows and walks a contiguous range of 128-column blocks; every lane
// keeps an ONLINE (max, sum-exp) pair for each of its 32 accumulator rows, so the soft-max statistics
// never leave registers until the block is done; lanes / waves / column splits are then merged in a
// fixed order.  The scores are written once (fp32) for the backward: with the fp32 MFMA at 1/16 of the bf16 rate
// a stored [S,S] matrix (HBM has room for it and the traffic hides under the matrix pipe) is cheaper than the
// flash-style recompute (3 GEMM passes instead of 4).
// Backward: dfhat = c (P - I) that, dthat = c (P - I)^T fhat with P = exp(Z - lse).
//   general form : Z is stored; two GEMMs whose A operand is transformed from Z while it is staged into LDS;
//   unit rows    : every logit is <= 1/tau, so the forward already evaluates E = exp(Z - shift) with the constant
//                  shift = 1.0001/tau for the row sums.  It stores E instead of Z, and with w_i = exp(shift - lse_i)
//                  = 1 / sum_j E_ij the backward needs no exp at all:
//                      dfhat_i = c (w_i (E that)_i - that_{i+off}),  dthat_j = c ((E^T (w o fhat))_j - fhat_{j-off})
//                  i.e. two plain GEMMs on E with a row scale on one side (one exp per 4 staged elements of fhat).
#include <stdlib.h>

#include "gemm3.h"

using namespace egnn_gemm;

namespace {

#ifndef EGNN_NCE_FWD_WAVES
#define EGNN_NCE_FWD_WAVES 2  // waves per SIMD the edge-free forward variant is compiled for
#endif
constexpr int kMaxSplit = 8;
constexpr int FB = 128;  // forward block tile (rows and columns)

// the constant soft-max shift of the unit-rows form; forward and backward must evaluate it identically
__device__ __forceinline__ float nce_shift(float inv_tau) { return inv_tau * 1.0001f; }
// the bound 1/tau is a usable shift while exp(-2/tau) stays a normal float (tau >= 0.025 leaves ample room)
inline bool nce_unit_form(float tau, int unit_rows) { return unit_rows && (2.f / tau) <= 80.f; }

// exp(x) for -100 < x < 80 (no overflow / denormal handling): the usual two-constant argument reduction in front of
// v_exp_f32, i.e. libm's expf without its range checks (8 instead of 17 VALU instructions; the epilogue's VALU work
// competes with other waves' MFMA issue).  Used where the unit-rows bound guarantees the range.
__device__ __forceinline__ float exp_bounded(float x) {
  const float kL2eHi = 1.44269502162933349609375f, kL2eLo = 1.925963033500011e-8f;
  const float n = __builtin_rintf(x * kL2eHi);
  float f = fmaf(x, kL2eHi, -n);
  f = fmaf(x, kL2eLo, f);
  return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

__device__ __forceinline__ void lse_merge(float& m, float& s, float om, float os) {
  const float mn = fmaxf(m, om);
  if (mn == -INFINITY) return;  // both empty
  s = s * expf(m - mn) + os * expf(om - mn);  // exp(-inf - finite) == 0
  m = mn;
}

// Epilogue of one interior 128 x 128 tile of the unit-rows form: E = exp(acc / tau - shift) is stored and added to the
// per-lane row sums.  Wave-uniform row bases + one 32-bit lane offset; nothing per-row is live outside.
template <class TS, int TM_, int TN_>
__device__ __forceinline__ void nce_tile_epilogue(const f32x16 (&acc)[TM_][TN_], float (&rs)[TM_][16], float* __restrict__ Z,
                                                  float* __restrict__ zdiag, int64_t ib, int64_t j0, int64_t sc,
                                                  int64_t diag_off, float inv_tau, float shift, int lane, int wm, int wn) {
  float* zb = Z + (ib + wm * TS::WM) * sc + (j0 + wn * TS::WN);
  const unsigned voff = (unsigned)(4 * (lane >> 5)) * (unsigned)sc + (unsigned)(lane & 31);
#pragma unroll
  for (int tm = 0; tm < TM_; ++tm) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* zr = zb + (int64_t)(tm * 32 + (r & 3) + 8 * (r >> 2)) * sc;
#pragma unroll
      for (int tn = 0; tn < TN_; ++tn) {
        const float e = exp_bounded(fmaf(acc[tm][tn][r], inv_tau, -shift));
        zr[voff + tn * 32] = e;
        rs[tm][r] += e;
      }
      __builtin_amdgcn_sched_barrier(0);  // one row at a time: keeps the 64 exp chains from being interleaved (registers)
    }
  }
  if (j0 < ib + diag_off + FB && ib + diag_off < j0 + FB) {  // the tile crosses the diagonal of positives
    // element (lr, lc) of the tile is a positive iff lr - lc == d; split into a lane part and a wave-uniform part
    // (32-bit: |d| < FB here) so that nothing per-row survives outside this rarely taken branch
    const int d = (int)(j0 - ib - diag_off);
    const int lv = 4 * (lane >> 5) - (lane & 31);
#pragma unroll
    for (int tm = 0; tm < TM_; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ur = wm * TS::WM + tm * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int tn = 0; tn < TN_; ++tn)
          if (lv == d - (ur - wn * TS::WN - tn * 32)) zdiag[ib + ur + 4 * (lane >> 5)] = acc[tm][tn][r] * inv_tau;
      }
  }
}

// FIXED: rows are unit vectors, so every logit is <= 1/tau and that bound serves as the soft-max shift: one exp per
// element and half the per-lane state of the online-max form (which costs a whole wave per SIMD in registers).
// ALIGNED: Sr and Sc are multiples of the tile, P of the k-step and the operands are float4-addressable (the sampled
// G-CRD problem: 16384 x 16384 x 256); no edge handling is compiled into that variant.
// SPLIT: products on the bf16 matrix pipe from the three-way split of the fp32 operands (gemm_split.h)
template <bool VEC4, bool FIXED, bool ALIGNED, bool SPLIT = false>
__global__ __launch_bounds__(256, ALIGNED ? EGNN_NCE_FWD_WAVES : 1) void nce_fwd_kernel(const float* __restrict__ fhat, int64_t ldf,
                                                      const float* __restrict__ that, int64_t ldt, int64_t Sr, int64_t Sc,
                                                      int64_t diag_off, int64_t P, float inv_tau,
                                                      float* __restrict__ Z, float* __restrict__ zdiag,
                                                      float* __restrict__ pm, float* __restrict__ ps, int nsplit,
                                                      int cb_per_split) {
  using TS = typename TileSel<SPLIT, FB, FB>::type;
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
  const float shift = nce_shift(inv_tau);
#pragma unroll
  for (int tm = 0; tm < TS::TM; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (!FIXED) rm[tm][r] = -INFINITY;
      rs[tm][r] = 0.f;
    }

  IdentityXf id;
  if constexpr (ALIGNED && SPLIT) {
    // the same continuous pipeline on the split ring: stage s <-> (column tile cb0 + s / nk, k-step s % nk); P % 64 == 0
    using PS = PipelineS<FB, FB, KMAJOR, KMAJOR, true, true, IdentityXf, IdentityXf>;
    static_assert(PS::UNROLL == PS::DEPTH && PS::DEPTH % 2 == 0, "ring depth must be even here");
    PS pipe(fhat, ldf, Sr, that, ldt, Sc, P, id, id);
    const int nk = (int)(P / BK);
    const int total = (int)(cb1 - cb0) * nk;
    f32x16 acc[TS::TM][TS::TN];
    zero_acc(acc);
    if (total > 0) {
      auto fetch = [&](auto slot, int st) {
        if (st >= total) st = total - 1;
        pipe.template prefetch<slot.value>(i0, (cb0 + st / nk) * FB, (int64_t)(st % nk) * BK);
      };
      static_for<PS::DEPTH>([&](auto d) { fetch(d, d.value); });
      pipe.template commit<0>(smem, 0);
      fetch(std::integral_constant<int, 0>{}, PS::DEPTH);
      __syncthreads();
      int base = 0;
      for (int64_t cb = cb0; cb < cb1; ++cb) {
        for (int kt0 = 0; kt0 < nk; kt0 += PS::UNROLL, base += PS::UNROLL) {
          static_for<PS::UNROLL>([&](auto u) {
            constexpr int slot = (u.value + 1) % PS::DEPTH;
            pipe.template step<slot>(acc, smem, u.value & 1, true, lane, wm, wn,
                                     [&]() { fetch(std::integral_constant<int, slot>{}, base + u.value + 1 + PS::DEPTH); });
            __syncthreads();
          });
        }
        int64_t sc = Sc, ib = i0;
        asm volatile("" : "+s"(sc), "+s"(ib));
        nce_tile_epilogue<TS>(acc, rs, Z, zdiag, ib, cb * FB, sc, diag_off, inv_tau, shift, lane, wm, wn);
        zero_acc(acc);
      }
    }
  } else if constexpr (ALIGNED) {
    // One continuous software pipeline over (column tile, k-step): the first k-step of the next tile is fetched and
    // committed to LDS during the last step of the current one, so an epilogue is followed by MFMAs at once.
    Pipeline<FB, FB, KMAJOR, KMAJOR, true, true, IdentityXf, IdentityXf> pipe;
    const int nk = (int)(P / BK);
    const int64_t total = (cb1 - cb0) * nk;
    f32x16 acc[TS::TM][TS::TN];
    zero_acc(acc);
    if (total > 0) {
      pipe.prefetch(fhat, ldf, i0, Sr, that, ldt, cb0 * FB, Sc, 0, P, id, id);
      pipe.commit(smem, 0);
    }
    __syncthreads();
    int kt = 0;
    int64_t cb = cb0;
    for (int64_t g = 0; g < total; ++g) {
      const bool more = g + 1 < total;
      int kn = kt + 1;
      int64_t cbn = cb;
      if (kn == nk) { kn = 0; ++cbn; }
      if (more) pipe.prefetch(fhat, ldf, i0, Sr, that, ldt, cbn * FB, Sc, (int64_t)kn * BK, P, id, id);
      pipe.step(acc, smem, (int)(g & 1), more, lane, wm, wn);
      __syncthreads();
      if (kn == 0) {
        // launder the loop-invariant scalars: see the general path below
        int64_t sc = Sc, ib = i0;
        asm volatile("" : "+s"(sc), "+s"(ib));
        nce_tile_epilogue<TS>(acc, rs, Z, zdiag, ib, cb * FB, sc, diag_off, inv_tau, shift, lane, wm, wn);
        zero_acc(acc);
      }
      kt = kn;
      cb = cbn;
    }
  } else
  for (int64_t cb = cb0; cb < cb1; ++cb) {
    const int64_t j0 = cb * FB;
    f32x16 acc[TS::TM][TS::TN];
    zero_acc(acc);
    mainloop_sel<SPLIT, FB, FB, KMAJOR, KMAJOR, VEC4>(acc, fhat, ldf, i0, Sr, that, ldt, j0, Sc, 0, P, id, id, smem);
    // The epilogue addresses are functions of loop-invariant quantities (i0, Sc); left alone, the compiler hoists one
    // 64-bit pointer per accumulator row out of the column-block loop (~150 registers, one wave per SIMD).  Laundering
    // the two scalars through an empty asm makes it recompute them per tile instead (a few dozen scalar ops).
    int64_t sc = Sc, ib = i0;
    asm volatile("" : "+s"(sc), "+s"(ib));
    const bool interior = (ib + FB <= Sr) && (j0 + FB <= sc);  // block-uniform
    if (FIXED && Z && interior && sc < (1LL << 28)) {
      if constexpr (FIXED) nce_tile_epilogue<TS>(acc, rs, Z, zdiag, ib, j0, sc, diag_off, inv_tau, shift, lane, wm, wn);
      continue;
    }
#pragma unroll
    for (int tm = 0; tm < TS::TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = ib + acc_row<FB, FB>(wm, tm, r, lane);
        float zt[TS::TN];
        float mt = -INFINITY;
#pragma unroll
        for (int tn = 0; tn < TS::TN; ++tn) {
          const int64_t c = j0 + acc_col<FB, FB>(wn, tn, lane);
          const float z = acc[tm][tn][r] * inv_tau;
          const bool ok = row < Sr && c < sc;
          if (ok) {
            if (Z) Z[row * sc + c] = FIXED ? exp_bounded(fmaf(acc[tm][tn][r], inv_tau, -shift)) : z;
            if (row + diag_off == c) zdiag[row] = z;
          }
          zt[tn] = ok ? z : -INFINITY;
          mt = fmaxf(mt, zt[tn]);
        }
        if constexpr (FIXED) {
#pragma unroll
          for (int tn = 0; tn < TS::TN; ++tn)  // masked columns contribute nothing
            if (zt[tn] > -INFINITY) rs[tm][r] += exp_bounded(fmaf(acc[tm][tn][r], inv_tau, -shift));
        } else if (mt > -INFINITY) {
          const float mn = fmaxf(rm[tm][r], mt);
          float s = rs[tm][r] * expf(rm[tm][r] - mn);
#pragma unroll
          for (int tn = 0; tn < TS::TN; ++tn) s += expf(zt[tn] - mn);
          rs[tm][r] = s;
          rm[tm][r] = mn;
        }
        __builtin_amdgcn_sched_barrier(0);
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

// lse_i = merge of the column-split partials; loss = inv_count * sum_i (lse_i - z_ii).  Two launches: rows in parallel with
// one partial per block, then a fixed-order sum of the block partials (the former single-block kernel took 43 us at S = 16384).
constexpr int kFinalBlocks = 1024;
__global__ __launch_bounds__(256) void nce_finalize_rows_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                                const float* __restrict__ zdiag, int64_t S, int nsplit,
                                                                float* __restrict__ lse, float* __restrict__ block_part) {
  __shared__ float red[256];
  float local = 0.f;
  for (int64_t row = blockIdx.x * 256LL + threadIdx.x; row < S; row += (int64_t)gridDim.x * 256) {
    float m = -INFINITY, s = 0.f;
    for (int k = 0; k < nsplit; ++k) lse_merge(m, s, pm[row * nsplit + k], ps[row * nsplit + k]);
    const float l = m + logf(s);
    lse[row] = l;
    local += l - zdiag[row];
  }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void nce_finalize_sum_kernel(const float* __restrict__ block_part, int nblocks, float inv_count,
                                                               float* __restrict__ loss) {
  __shared__ float red[256];
  float local = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) local += block_part[i];
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
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

// B-operand transform of the unit-rows backward: row k of fhat scaled by w_k = exp(shift - lse_k) = 1 / sum_j E_kj
struct RowWeightXf {
  const float* lse;
  float shift;
  __device__ __forceinline__ float operator()(float v, int64_t, int64_t k) const { return v * expf(shift - lse[k]); }
};

// C[M, N=P] = scale * (P - I or its transpose) [M,Kd] * B[Kd,P]   (B row-major, n contiguous; ZE is [Sr,Sc], ld = Sc)
//   EXPZ = false: ZE holds Z, the A operand is transformed (NceGradXf) on its way into LDS
//   EXPZ = true : ZE holds E = exp(Z - shift); A is staged as it is, the row weights and the "- I" term are applied
//                 to B (AMAJ == MNMAJOR, i.e. the teacher side) or in the epilogue
// epilogue shared by the GEMM kernel (one k range) and the split-K reduce: v = sum_k A_mk B_kc for element (row, c)
template <int AMAJ, bool EXPZ>
__device__ __forceinline__ float nce_bwd_finish(float v, int64_t row, int64_t c, int64_t Kd, int64_t diag_off,
                                                const float* __restrict__ lse, float shift, const float* __restrict__ Im,
                                                int64_t ldi, float scale) {
  if constexpr (EXPZ) {
    // Im = the matrix whose rows the "- I" term picks: that (student side, row i -> i + off) or fhat (row j -> j - off)
    if constexpr (AMAJ == KMAJOR) {
      v = v * expf(shift - lse[row]) - Im[(row + diag_off) * ldi + c];
    } else {
      const int64_t i = row - diag_off;
      if (i >= 0 && i < Kd) v -= Im[i * ldi + c];
    }
  }
  return scale * v;
}

template <int BM, int AMAJ, bool VEC4, bool EXPZ, bool BW = true, bool SPLIT = false>
__global__ __launch_bounds__(256, SPLIT ? 2 : 1) void nce_bwd_kernel(const float* __restrict__ Z, int64_t ldz, int64_t M, int64_t Kd,
                                                      int64_t diag_off, const float* __restrict__ lse, float shift,
                                                      const float* __restrict__ Bm, int64_t P, int64_t ldb,
                                                      const float* __restrict__ Im, int64_t ldi,
                                                      float coef, const float* __restrict__ g, float* __restrict__ C,
                                                      int64_t ldc, int64_t k_per_split, float* __restrict__ ws) {
  constexpr int BN = 128;
  using TS = typename TileSel<SPLIT, BM, BN>::type;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t tiles_n = (P + BN - 1) / BN;
  const int64_t tiles = (int64_t)gridDim.x;
  // workgroup b runs on XCD b % 8: keep the column tiles of one row tile on the same XCD and next to each other in
  // time, so that the second read of the (large) A tile hits that XCD's L2
  int64_t mt, nt;
  if (tiles_n > 1 && tiles % (8 * tiles_n) == 0) {
    nt = (blockIdx.x >> 3) % tiles_n;
    mt = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * tiles_n));
  } else {
    mt = blockIdx.x / tiles_n;
    nt = blockIdx.x % tiles_n;
  }
  const int64_t m0 = mt * BM, n0 = nt * BN;
  const int64_t kbeg = (int64_t)blockIdx.y * k_per_split;
  int64_t kend = kbeg + k_per_split;
  if (kend > Kd) kend = Kd;
  f32x16 acc[TS::TM][TS::TN];
  zero_acc(acc);
  IdentityXf id;
  // interior tiles with whole k-steps run the loop without edge handling (the split pipeline's ring needs it)
  const bool full = SPLIT && (kend - kbeg) % BK == 0;   // rows past the edge are clamped inside the unguarded loop
  auto run = [&](auto fo, const auto& xa, const auto& xb) {
    mainloop_sel<SPLIT, BM, BN, AMAJ, MNMAJOR, VEC4, fo.value>(acc, Z, ldz, m0, M, Bm, ldb, n0, P, kbeg, kend, xa, xb, smem);
  };
  auto both = [&](const auto& xa, const auto& xb) {
    if constexpr (SPLIT) {
      if (full) run(std::true_type{}, xa, xb);
      else run(std::false_type{}, xa, xb);
    } else {
      run(std::false_type{}, xa, xb);
    }
  };
  if constexpr (!EXPZ) {
    NceGradXf xf{lse, AMAJ == MNMAJOR, diag_off};
    both(xf, id);
  } else if constexpr (AMAJ == KMAJOR) {
    both(id, id);
  } else if constexpr (BW) {   // teacher side, rows of fhat weighted while they are staged
    RowWeightXf xw{lse, shift};
    both(id, xw);
  } else {                     // teacher side, Bm already holds w o fhat (nce_row_weight_kernel)
    both(id, id);
  }
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const bool partial = gridDim.y > 1;
  const float scale = coef * (g ? g[0] : 1.f);
  float* out = partial ? ws + (int64_t)blockIdx.y * M * P : C;
  const int64_t ldo = partial ? P : ldc;
#pragma unroll
  for (int tn = 0; tn < TS::TN; ++tn) {
    const int64_t c = n0 + acc_col<BM, BN>(wn, tn, lane);
    if (c >= P) continue;
#pragma unroll
    for (int tm = 0; tm < TS::TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + acc_row<BM, BN>(wm, tm, r, lane);
        if (row >= M) continue;
        const float v = acc[tm][tn][r];
        out[row * ldo + c] = partial ? v : nce_bwd_finish<AMAJ, EXPZ>(v, row, c, Kd, diag_off, lse, shift, Im, ldi, scale);
      }
  }
}

// fixed-order sum of the split-K partials + the epilogue
template <int AMAJ, bool EXPZ>
__global__ __launch_bounds__(256) void nce_bwd_reduce_kernel(const float* __restrict__ ws, int nsplit, int64_t M, int64_t P, int64_t Kd,
                                                             int64_t diag_off, const float* __restrict__ lse, float shift,
                                                             const float* __restrict__ Im, int64_t ldi, float coef,
                                                             const float* __restrict__ g, float* __restrict__ C, int64_t ldc) {
  const int64_t total = M * P;
  const float scale = coef * (g ? g[0] : 1.f);
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    float v = 0.f;
    for (int k = 0; k < nsplit; ++k) v += ws[(int64_t)k * total + t];
    const int64_t row = t / P, c = t % P;
    C[row * ldc + c] = nce_bwd_finish<AMAJ, EXPZ>(v, row, c, Kd, diag_off, lse, shift, Im, ldi, scale);
  }
}

// out[i,:] = exp(shift - lse[i]) * x[i,:]: the weighted student rows of the unit-rows backward, formed once instead of
// inside the GEMM's staging loop (an exp per four staged elements there competes with the MFMA issue)
__global__ __launch_bounds__(256) void nce_row_weight_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ lse,
                                                             float shift, int64_t n, int64_t P, float* __restrict__ out) {
  const int64_t total = n * P;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t i = t / P, c = t % P;
    out[t] = x[i * ldx + c] * expf(shift - lse[i]);
  }
}

// split-K factor of a backward GEMM with M output rows over a reduction of Kd: aim at >= 3 workgroups of 128 x 128 per CU
inline int nce_bwd_split(int64_t M, int64_t P, int64_t Kd) {
  const int64_t t128 = ((M + 127) / 128) * ((P + 127) / 128);
  int64_t n = (768 + t128 - 1) / t128;
  const int64_t ksteps = (Kd + BK - 1) / BK;
  if (n > ksteps / 8) n = ksteps / 8;  // at least 8 k-steps per split
  if (n > kMaxSplit) n = kMaxSplit;
  return n < 1 ? 1 : (int)n;
}

// ---- unit-rows backward on the DMA pipeline (gemm3.h) ----------------------------------------------------------------------
// Both backward products have E (fp32, [Sr, Sc], written by the forward) as their big operand and a [S, P] matrix of unit rows
// as the small one.  The small operand is cut ONCE per call into tile-packed bf16 planes (T^ as it is for the student side,
// the rows of F^ weighted by w_i for the teacher side) and streamed by lane-linear DMA; E goes global -> LDS by DMA as fp32 and
// is cut on the fragment side, as rows (student side: A = E, k contiguous) or down columns (teacher side: A = E^T, the [k][m]
// image read with ds_read_b32).  No staging registers, no ds_write pass, no split VALU for the small operand.  Partials of the
// split reduction go to the workspace; nce_bwd_reduce_kernel adds them in a fixed order and applies the epilogue.
//   student side: 128 x 128 tiles, k-steps of 32, two LDS stages (lab: 705 vs 795 us at S = 16384, P = 256)
//   teacher side: 256 x 256 tiles, four waves of 128 x 128, k-steps of 16, three LDS stages (667 vs 810 us)
template <int AMODE, int TM, int TN, int BKT, int NB, int OCC>
__global__ __launch_bounds__(256, OCC) void nce3_bwd_kernel(const float* __restrict__ E, int64_t lde, const char* __restrict__ planes, int64_t nks,
                                                            int64_t M, int64_t P, int64_t k_per_split, float* __restrict__ ws) {
  using T = egnn_gemm3::Tile<AMODE, egnn_gemm3::PLANES, TM, TN, BKT, NB>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t tiles_n = P / T::BN;
  int64_t tile = blockIdx.x;
  if (tiles_n > 1 && tiles_n <= 8) {   // the column tiles of a row tile follow each other on ONE XCD (they share the E tile)
    const int64_t tiles = gridDim.x, q = tiles >> 3, rem = tiles & 7, xcd = tile & 7, j = tile >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
  }
  const int64_t m0 = (tile / tiles_n) * T::BM, n0 = (tile % tiles_n) * T::BN;
  const int64_t kbeg = (int64_t)blockIdx.y * k_per_split, kend = kbeg + k_per_split;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // (scheduling hints per form, lab-measured on this very shape: the hand-placed order for the 128 x 128 student side -3.5 %,
  //  iglp_opt(0) for the 256 x 256 teacher side -5 %)
  constexpr int SCHED = AMODE == egnn_gemm3::F32M ? egnn_gemm3::SCHED_IGLP0 : egnn_gemm3::SCHED_HAND;
  egnn_gemm3::mainloop<AMODE, egnn_gemm3::PLANES, TM, TN, BKT, NB, 1, SCHED>(acc, E, lde, m0, planes, nks, n0, kbeg, kend, reinterpret_cast<char*>(smem));
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  float* out = ws + (int64_t)blockIdx.y * M * P;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int64_t c = n0 + wn * 32 * TN + tn * 32 + (lane & 31);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[row * P + c] = acc[tm][tn][r];
      }
  }
}

constexpr int kNce3SplitMax = 8;   // k ranges of the reduction (fixed-order reduce afterwards); the workspace is sized for the maximum
static inline int nce3_split(int64_t M) {
  // 4 k-ranges: 2 378 vs 2 433 us (8) vs 2 718 us (2) for forward + backward at S = 16 384, P = 256 (fewer partials to reduce, still one
  // workgroup per CU on the 256 x 256 side).  At S = 8 192 the 256 x 256 side has 32 tiles: 8 ranges fill the 256 CUs, 4 leave half of
  // them idle (735 vs 814 us).
  return (M / 256) * 4 >= 256 ? 4 : 8;
}

// floats of workspace the DMA backward needs for one side: partials + the packed planes of the small operand (+ slack for alignment)
inline size_t nce3_ws_floats(int64_t M, int64_t P, int64_t Kd) {
  return (size_t)kNce3SplitMax * M * P + (egnn_gemm3::planes_bytes(P, Kd, 256, 32) + 1024) / 4;
}

// true when this side of the backward can take the DMA pipeline (whole tiles, aligned E, unit-rows form with a workspace)
template <int AMAJ>
inline bool nce3_takes(int64_t M, int64_t Kd, int64_t P, int64_t ldz, const float* Z, bool expz, bool vec4, const float* ws) {
  constexpr int BM = AMAJ == KMAJOR ? 128 : 256, BN = AMAJ == KMAJOR ? 128 : 256, BKT = AMAJ == KMAJOR ? 32 : 16;
  return egnn_split_pipe() && expz && vec4 && ws && M % BM == 0 && P % BN == 0 && Kd % (kNce3SplitMax * BKT) == 0 && ldz % 4 == 0 &&
         egnn_aligned16(Z) && Kd / kNce3SplitMax >= 4 * BKT;
}

template <int AMAJ>
int launch_bwd3(const float* Z, int64_t ldz, int64_t M, int64_t Kd, int64_t diag_off, const float* lse, float shift, const float* Bm, int64_t P,
                int64_t ldb, const float* Im, int64_t ldi, float coef, const float* g, float* C, int64_t ldc, float* ws, hipStream_t st) {
  using namespace egnn_gemm3;
  float* partial = ws;
  const int kNce3Split = nce3_split(M);
  char* planes = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws + (size_t)kNce3SplitMax * M * P) + 1023) & ~(uintptr_t)1023);
  const int64_t k_per_split = Kd / kNce3Split;
  int rc;
  if constexpr (AMAJ == KMAJOR) {
    // student side: C = E T^ ; B(k = j, n = p) = T^[j, p]  ->  planes of T^ transposed (rows p, k = j)
    constexpr int TM = 2, TN = 2, BKT = 32, NB = 2;
    pack_planes<64 * TN, BKT>(Bm, ldb, 0, P, Kd, nullptr, nullptr, nullptr, 0.f, planes, st);
    using T = Tile<F32K, PLANES, TM, TN, BKT, NB>;
    const dim3 grid((unsigned)((M / T::BM) * (P / T::BN)), (unsigned)kNce3Split);
    rc = launch_dyn_lds<nce3_bwd_kernel<F32K, TM, TN, BKT, NB, 2>>(grid, dim3(256), (size_t)T::SMEM_BYTES, st, Z, ldz, (const char*)planes, Kd / BKT, M, P,
                                                                   k_per_split, partial);
  } else {
    // teacher side: C = E^T (w o F^) ; A(m = j, k = i) = E[i, j] ; B(k = i, n = p) = w_i F^[i, p]  ->  planes of (w o F^) transposed
    constexpr int TM = 4, TN = 4, BKT = 16, NB = 3;
    pack_planes<64 * TN, BKT>(Bm, ldb, 0, P, Kd, nullptr, nullptr, lse, shift, planes, st);
    using T = Tile<F32M, PLANES, TM, TN, BKT, NB>;
    const dim3 grid((unsigned)((M / T::BM) * (P / T::BN)), (unsigned)kNce3Split);
    rc = launch_dyn_lds<nce3_bwd_kernel<F32M, TM, TN, BKT, NB, 1>>(grid, dim3(256), (size_t)T::SMEM_BYTES, st, Z, ldz, (const char*)planes, Kd / BKT, M, P,
                                                                   k_per_split, partial);
  }
  if (rc != EGNN_OK) return rc;
  const int64_t rb = (M * P + 255) / 256;
  hipLaunchKernelGGL((nce_bwd_reduce_kernel<AMAJ, true>), dim3((unsigned)(rb < 4096 ? rb : 4096)), dim3(256), 0, st, partial, kNce3Split, M, P, Kd,
                     diag_off, lse, shift, Im, ldi, coef, g, C, ldc);
  return EGNN_OK;
}

template <int AMAJ>
int launch_bwd(const float* Z, int64_t ldz, int64_t M, int64_t Kd, int64_t diag_off, const float* lse, float shift, bool expz,
                const float* Bm, int64_t P, int64_t ldb, const float* Im, int64_t ldi, float coef, const float* g, float* C,
                int64_t ldc, bool vec4, float* ws, hipStream_t st) {
  if (nce3_takes<AMAJ>(M, Kd, P, ldz, Z, expz, vec4, ws)) {
    return launch_bwd3<AMAJ>(Z, ldz, M, Kd, diag_off, lse, shift, Bm, P, ldb, Im, ldi, coef, g, C, ldc, ws, st);   // a refused big-LDS opt-in / failed launch is the caller's error
  }
  const int64_t tiles_n = (P + 127) / 128;
  const int64_t t128 = ((M + 127) / 128) * tiles_n;
  // With a workspace the reduction is split so that every CU gets ~3 workgroups of 128 x 128 (fixed-order reduce
  // afterwards).  Without one: 128-row tiles only when they still give every CU at least two workgroups (one wave per
  // SIMD cannot hide its own staging), else the 64-row tile doubles the workgroup count.
  constexpr int64_t min_tiles = 600;
  const int nsplit = ws ? nce_bwd_split(M, P, Kd) : 1;
  const bool big = ws || t128 >= min_tiles;
  const int64_t ksteps = (Kd + BK - 1) / BK;
  const int64_t k_per_split = ((ksteps + nsplit - 1) / nsplit) * BK;
  const dim3 grid((unsigned)(big ? t128 : ((M + 63) / 64) * tiles_n), (unsigned)nsplit);
  const bool split = big && egnn_split_pipe();   // 128-row tiles: products on the bf16 pipe (gemm_split.h)
  constexpr size_t shm_f32 = (size_t)TileShape<128, 128>::SMEM_FLOATS * 4, shm_split = (size_t)TileShapeS<128, 128>::SMEM_BYTES;
  int rc = EGNN_OK;
#define EGNN_NCE_BWD_ARGS(B_, LDB_) Z, ldz, M, Kd, diag_off, lse, shift, B_, P, LDB_, Im, ldi, coef, g, C, ldc, k_per_split, ws
  if (AMAJ == MNMAJOR && ws && expz && vec4 && big) {   // teacher side of the unit-rows form: weight the Kd student rows once, then a plain GEMM
    float* wx = ws + (size_t)nsplit * M * P;
    const int64_t blocks = (Kd * P + 255) / 256;
    hipLaunchKernelGGL(nce_row_weight_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, Bm, ldb, lse, shift, Kd, P, wx);
    if (split) rc = launch_dyn_lds<nce_bwd_kernel<128, AMAJ, true, true, false, true>>(grid, dim3(256), shm_split, st, EGNN_NCE_BWD_ARGS(wx, P));
    else rc = launch_dyn_lds<nce_bwd_kernel<128, AMAJ, true, true, false, false>>(grid, dim3(256), shm_f32, st, EGNN_NCE_BWD_ARGS(wx, P));
    if (rc != EGNN_OK) return rc;
    if (nsplit > 1) {
      const int64_t rb = (M * P + 255) / 256;
      hipLaunchKernelGGL((nce_bwd_reduce_kernel<AMAJ, true>), dim3((unsigned)(rb < 4096 ? rb : 4096)), dim3(256), 0, st, ws, nsplit, M, P, Kd,
                         diag_off, lse, shift, Im, ldi, coef, g, C, ldc);
    }
    return EGNN_OK;
  }
#define EGNN_NCE_BWD(BM_, V, E, S_) rc_main = launch_dyn_lds<nce_bwd_kernel<BM_, AMAJ, V, E, true, S_>>(grid, dim3(256), S_ ? shm_split : (size_t)TileShape<BM_, 128>::SMEM_FLOATS * 4, st, EGNN_NCE_BWD_ARGS(Bm, ldb))
  int rc_main = EGNN_OK;
  if (split) {
    if (vec4) { if (expz) EGNN_NCE_BWD(128, true, true, true); else EGNN_NCE_BWD(128, true, false, true); }
    else { if (expz) EGNN_NCE_BWD(128, false, true, true); else EGNN_NCE_BWD(128, false, false, true); }
  } else if (big) {
    if (vec4) { if (expz) EGNN_NCE_BWD(128, true, true, false); else EGNN_NCE_BWD(128, true, false, false); }
    else { if (expz) EGNN_NCE_BWD(128, false, true, false); else EGNN_NCE_BWD(128, false, false, false); }
  } else {
    if (vec4) { if (expz) EGNN_NCE_BWD(64, true, true, false); else EGNN_NCE_BWD(64, true, false, false); }
    else { if (expz) EGNN_NCE_BWD(64, false, true, false); else EGNN_NCE_BWD(64, false, false, false); }
  }
#undef EGNN_NCE_BWD
#undef EGNN_NCE_BWD_ARGS
  if (rc_main != EGNN_OK) return rc_main;
  if (nsplit > 1) {
    const int64_t blocks = (M * P + 255) / 256;
    const dim3 rgrid((unsigned)(blocks < 4096 ? blocks : 4096));
    if (expz) hipLaunchKernelGGL((nce_bwd_reduce_kernel<AMAJ, true>), rgrid, dim3(256), 0, st, ws, nsplit, M, P, Kd, diag_off, lse, shift, Im, ldi, coef, g, C, ldc);
    else hipLaunchKernelGGL((nce_bwd_reduce_kernel<AMAJ, false>), rgrid, dim3(256), 0, st, ws, nsplit, M, P, Kd, diag_off, lse, shift, Im, ldi, coef, g, C, ldc);
  }
  return EGNN_OK;
}

}  // namespace

extern "C" size_t egnn_nce_ws_floats(int64_t S) { return (size_t)S * (1 + 2 * kMaxSplit) + kFinalBlocks; }

extern "C" size_t egnn_nce_bwd_ws_floats(int64_t Sr, int64_t Sc, int64_t P) {
  if (Sr <= 0 || Sc <= 0 || P <= 0) return 0;
  const size_t a = (size_t)nce_bwd_split(Sr, P, Sc) * Sr * P, b = (size_t)nce_bwd_split(Sc, P, Sr) * Sc * P;
  const size_t staged = (a > b ? a : b) + (size_t)Sr * P;   // split-K partials + the weighted student rows of the teacher-side GEMM
  const size_t a3 = nce3_ws_floats(Sr, P, Sc), b3 = nce3_ws_floats(Sc, P, Sr);   // the DMA pipeline's partials + packed planes
  const size_t dma = a3 > b3 ? a3 : b3;
  return staged > dma ? staged : dma;
}

extern "C" int egnn_nce_saves_exp(float tau, int unit_rows) { return tau > 0.f && nce_unit_form(tau, unit_rows) ? 1 : 0; }

extern "C" int egnn_nce_block_fwd_f32(const float* fhat, int64_t ld_f, const float* that, int64_t ld_t, int64_t Sr, int64_t Sc,
                                      int64_t diag_off, int64_t P, float tau, float inv_count, int unit_rows, float* Z,
                                      float* lse, float* loss, float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(Sr > 0 && Sc > 0 && P > 0 && ld_f >= P && ld_t >= P && tau > 0.f && fhat && that && lse && loss && ws);
  EGNN_CHECK_ARG(diag_off >= 0 && diag_off + Sr <= Sc);
  if (ws_floats < egnn_nce_ws_floats(Sr)) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rb = (Sr + FB - 1) / FB;
  const int64_t ncb = (Sc + FB - 1) / FB;
  int nsplit = (int)((256 * EGNN_NCE_FWD_WAVES + rb - 1) / rb);  // one resident workgroup per wave slot of every CU
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
  const bool fixed = nce_unit_form(tau, unit_rows);
  const bool aligned = Z && vec4 && fixed && Sr % FB == 0 && Sc % FB == 0 && P % BK == 0 && Sc < (1LL << 28);
  const bool split = egnn_split_pipe() && (!aligned || P % (BK * PipelineS<FB, FB, KMAJOR, KMAJOR, true, true, IdentityXf, IdentityXf>::UNROLL) == 0);
  int rc_fwd = EGNN_OK;
#define EGNN_NCE_FWD(V, F, A)                                                                                                              \
  do {                                                                                                                                     \
    if (split) rc_fwd = launch_dyn_lds<nce_fwd_kernel<V, F, A, true>>(grid, dim3(256), (size_t)TileShapeS<FB, FB>::SMEM_BYTES, st, fhat, ld_f, that, ld_t, Sr, \
                                                             Sc, diag_off, P, 1.f / tau, Z, zdiag, pm, ps, nsplit, cb_per_split);           \
    else rc_fwd = launch_dyn_lds<nce_fwd_kernel<V, F, A, false>>(grid, dim3(256), (size_t)TileShape<FB, FB>::SMEM_FLOATS * 4, st, fhat, ld_f, that, ld_t, Sr,  \
                                                        Sc, diag_off, P, 1.f / tau, Z, zdiag, pm, ps, nsplit, cb_per_split);                \
  } while (0)
  if (aligned) EGNN_NCE_FWD(true, true, true);
  else if (vec4) { if (fixed) EGNN_NCE_FWD(true, true, false); else EGNN_NCE_FWD(true, false, false); }
  else { if (fixed) EGNN_NCE_FWD(false, true, false); else EGNN_NCE_FWD(false, false, false); }
#undef EGNN_NCE_FWD
  if (rc_fwd != EGNN_OK) return rc_fwd;
  float* block_part = ps + Sr * kMaxSplit;
  const int fb = (int)((Sr + 255) / 256 < kFinalBlocks ? (Sr + 255) / 256 : kFinalBlocks);
  hipLaunchKernelGGL(nce_finalize_rows_kernel, dim3(fb), dim3(256), 0, st, pm, ps, zdiag, Sr, nsplit, lse, block_part);
  hipLaunchKernelGGL(nce_finalize_sum_kernel, dim3(1), dim3(256), 0, st, block_part, fb, inv_count, loss);
  return egnn_launch_status();
}

extern "C" int egnn_nce_block_bwd_f32(const float* fhat, int64_t ld_f, const float* that, int64_t ld_t, int64_t Sr, int64_t Sc,
                                      int64_t diag_off, int64_t P, float tau, float scale, int unit_rows, const float* Z,
                                      const float* lse, const float* g, float* dfhat, int64_t ld_df, float* dthat, int64_t ld_dt,
                                      float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(Sr > 0 && Sc > 0 && P > 0 && ld_f >= P && ld_t >= P && tau > 0.f && fhat && that && Z && lse);
  if (ws && ws_floats < egnn_nce_bwd_ws_floats(Sr, Sc, P)) return EGNN_EWORKSPACE;
  EGNN_CHECK_ARG(diag_off >= 0 && diag_off + Sr <= Sc);
  EGNN_CHECK_ARG((dfhat == nullptr || ld_df >= P) && (dthat == nullptr || ld_dt >= P));
  hipStream_t st = (hipStream_t)stream;
  const bool vec4 = (ld_f % 4 == 0) && (ld_t % 4 == 0) && (Sc % 4 == 0) && egnn_aligned16(fhat) && egnn_aligned16(that) && egnn_aligned16(Z);
  const bool expz = nce_unit_form(tau, unit_rows);   // what the forward stored in Z
  const float shift = (1.f / tau) * 1.0001f;         // == nce_shift(inv_tau) of the forward
  // dfhat [Sr,P] = scale g (P - I) that ;  dthat [Sc,P] = scale g (P - I)^T fhat
  int rc = EGNN_OK;
  if (dfhat) rc = launch_bwd<KMAJOR>(Z, Sc, Sr, Sc, diag_off, lse, shift, expz, that, P, ld_t, that, ld_t, scale, g, dfhat, ld_df, vec4, ws, st);
  if (rc == EGNN_OK && dthat) rc = launch_bwd<MNMAJOR>(Z, Sc, Sc, Sr, diag_off, lse, shift, expz, fhat, P, ld_f, fhat, ld_f, scale, g, dthat, ld_dt, vec4, ws, st);
  if (rc != EGNN_OK) return rc;
  return egnn_launch_status();
}

extern "C" int egnn_nce_fwd_f32(const float* fhat, const float* that, int64_t S, int64_t P, int64_t ld, float tau, int unit_rows,
                                float* Z, float* lse, float* loss, float* ws, size_t ws_floats, void* stream) {
  return egnn_nce_block_fwd_f32(fhat, ld, that, ld, S, S, 0, P, tau, S > 0 ? 1.f / (float)S : 0.f, unit_rows, Z, lse, loss, ws, ws_floats, stream);
}

extern "C" int egnn_nce_bwd_f32(const float* fhat, const float* that, int64_t S, int64_t P, int64_t ld, float tau, int unit_rows,
                                const float* Z, const float* lse, const float* g, float* dfhat, float* dthat,
                                float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(S > 0 && tau > 0.f);
  return egnn_nce_block_bwd_f32(fhat, ld, that, ld, S, S, 0, P, tau, 1.f / ((float)S * tau), unit_rows, Z, lse, g, dfhat, ld, dthat,
                                ld, ws, ws_floats, stream);
}
