// fp32 GEMM products on the bf16 matrix pipe of gfx950 ("split" pipeline, drop-in for egnn_gemm::Pipeline).
//
// The f32-input MFMA runs at 1/16 of the bf16 MFMA rate on CDNA4 (157 vs 2500 TFLOP/s).  Every fp32 operand element
// x is therefore cut, while its tile is staged into LDS, into three bf16 terms
//       x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)        (round to nearest)
// The two subtractions are exact in fp32, so |x - (x0 + x1 + x2)| <= 2^-27 |x|: the three terms carry the whole 24-bit
// significand.  A product a * b is then accumulated as the six partial products a_i * b_j with i + j <= 2 -- each EXACT
// in the fp32 accumulator (8-bit x 8-bit significands) -- on v_mfma_f32_32x32x16_bf16; the three dropped products are
// bounded by 2^-26 |a b|, a quarter of the rounding of ONE fp32 multiply-add.  6 MFMAs of 16 k-values at 32 cycles
// replace 8 MFMAs of 2 k-values at 64 cycles: 2.67 x the matrix rate of the fp32 pipe for the same fp32 answer
// (tests/test_gpu_parity.py compares both pipelines with a float64 product: the split one is at least as close).
// Non-finite inputs (inf) turn into NaN (inf - inf in the split) -- the fp32 pipe would give inf or NaN.
//
// Tile / LDS: block tile BM x BN (multiples of 128), 4 waves 2 x 2, k-step 16 = one MFMA; an operand tile lives in LDS as
// three planes [rows][48 B] (8 + 8 bf16 of one row = 32 B, padded to a 3 x 16-byte stride: both the ds_write_b128 of the
// staging pass and the ds_read_b128 of the fragments are bank-conflict-free), two buffers, one barrier per step.
// Staging threads own one row x 8 consecutive k (k-major operand: two 16-byte loads; row-major-in-k operand: eight
// 4-byte loads, lanes along the rows), i.e. exactly the 16-byte fragment of one lane.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "gemm_core.h"

namespace egnn_gemm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static_assert(BK == 16, "the split pipeline consumes one 32x32x16 MFMA per k-step");
constexpr int S_ROW = 48;   // bytes per LDS row of one plane

__device__ __forceinline__ unsigned pack_bf16(float x, float y) {   // v_cvt_pk_bf16_f32: x -> low half, y -> high half
  const f32x2v v = {x, y};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// eight consecutive-k values of one row -> their three bf16 planes (16 bytes each)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x = v[2 * q], y = v[2 * q + 1];
    const unsigned a = pack_bf16(x, y);
    const float rx = x - __uint_as_float(a << 16), ry = y - __uint_as_float(a & 0xffff0000u);
    const unsigned b = pack_bf16(rx, ry);
    const float sx = rx - __uint_as_float(b << 16), sy = ry - __uint_as_float(b & 0xffff0000u);
    p0[q] = a;
    p1[q] = b;
    p2[q] = pack_bf16(sx, sy);
  }
}

// Stages one operand tile of R rows x 16 k: global (fp32) -> registers -> three bf16 planes in LDS.
// NTHR = threads of the workgroup (256: 4 waves; 512: the 8-wave 256 x 128 tile).  An operand with fewer than NTHR
// (row, chunk) items is staged by the first waves only (wave-uniform).
template <int R, int MAJOR, bool VEC4, class XF, bool GATHER = false, int NTHR = 256>
struct StagerS {
  static_assert(R % 128 == 0, "split tiles are multiples of 128 rows");
  static constexpr int ITEMS = 2 * R;                                  // (row, 8-k chunk) items per k-step
  static constexpr int NC = ITEMS >= NTHR ? ITEMS / NTHR : 1;          // items per (active) thread
  static constexpr bool PARTIAL = ITEMS < NTHR;                        // only threads < ITEMS take part
  float v[NC][8];

  // item i of thread t -> (row inside the tile, chunk)
  static __device__ __forceinline__ int item_row(int t, int i) {
    if constexpr (MAJOR == KMAJOR) return (t >> 1) + (NTHR / 2) * i;
    else return (t + NTHR * i) % R;
  }
  static __device__ __forceinline__ int item_chunk(int t, int i) {
    if constexpr (MAJOR == KMAJOR) return t & 1;
    else return ((t + NTHR * i) / R) & 1;
  }
  static __device__ __forceinline__ bool active(int t) { return !PARTIAL || t < ITEMS; }

  template <bool FULL>
  __device__ __forceinline__ void load_impl(const float* __restrict__ p, int64_t ld, int64_t r0, int64_t rmax, int64_t k0,
                                            int64_t kmax, const XF& xf, const int64_t* __restrict__ ridx) {
    const int t = threadIdx.x;
    if (!active(t)) return;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      if constexpr (MAJOR == KMAJOR) {
        int64_t r = r0 + item_row(t, i);
        if (FULL && r >= rmax) r = rmax - 1;   // FULL = whole k-steps, no per-element guards: rows past the edge re-read the last row (never stored)
        const int64_t k = k0 + item_chunk(t, i) * 8;
        int64_t rs = r;
        if constexpr (GATHER) rs = (FULL || r < rmax) ? ridx[r] : 0;
        const float* q = p + rs * ld + k;
        if constexpr (FULL) {
          float x[8];
          if constexpr (VEC4) {
            const float4 lo = *reinterpret_cast<const float4*>(q), hi = *reinterpret_cast<const float4*>(q + 4);
            x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w; x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = q[j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][j] = xf(x[j], r, k + j);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][j] = (r < rmax && k + j < kmax) ? xf(q[j], r, k + j) : 0.f;
        }
      } else {
        int64_t r = r0 + item_row(t, i);
        if (FULL && r >= rmax) r = rmax - 1;
        const int64_t k = k0 + item_chunk(t, i) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          int64_t ks = k + j;
          if constexpr (GATHER) ks = (FULL || k + j < kmax) ? ridx[k + j] : 0;
          if constexpr (FULL) v[i][j] = xf(p[ks * ld + r], r, k + j);
          else v[i][j] = (k + j < kmax && r < rmax) ? xf(p[ks * ld + r], r, k + j) : 0.f;
        }
      }
    }
  }

  template <bool FULLONLY = false>
  __device__ __forceinline__ void load(const float* __restrict__ p, int64_t ld, int64_t r0, int64_t rmax, int64_t k0,
                                       int64_t kmax, const XF& xf, const int64_t* __restrict__ ridx = nullptr) {
    if constexpr (FULLONLY) {
      load_impl<true>(p, ld, r0, rmax, k0, kmax, xf, ridx);
    } else {
      const bool full = k0 + BK <= kmax;  // block-uniform (rows past the edge are clamped, see load_impl)
      if (full) load_impl<true>(p, ld, r0, rmax, k0, kmax, xf, ridx);
      else load_impl<false>(p, ld, r0, rmax, k0, kmax, xf, ridx);
    }
  }

  // LDS image: [3][R][S_ROW bytes]; chunk g of a row sits at byte 16 g
  __device__ __forceinline__ void store(char* lds) const {
    const int t = threadIdx.x;
    if (!active(t)) return;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int row = item_row(t, i);
      const int g = item_chunk(t, i);
      u32x4 p0, p1, p2;
      split8(v[i], p0, p1, p2);
      char* d = lds + row * S_ROW + g * 16;
      *reinterpret_cast<u32x4*>(d) = p0;
      *reinterpret_cast<u32x4*>(d + R * S_ROW) = p1;
      *reinterpret_cast<u32x4*>(d + 2 * R * S_ROW) = p2;
    }
  }
};

template <int BM, int BN, int WAVES_M = 2>
struct TileShapeS {
  static constexpr int WM = BM / WAVES_M, WN = BN / 2;   // waves WAVES_M x 2
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int A_BUF = 3 * BM * S_ROW, B_BUF = 3 * BN * S_ROW;   // bytes per buffer
  static constexpr int SMEM_BYTES = 2 * (A_BUF + B_BUF);
  static constexpr int SMEM_FLOATS = SMEM_BYTES / 4;
};

#ifndef EGNN_SPLIT_DEPTH
#define EGNN_SPLIT_DEPTH 4   // k-steps of operand tiles in flight (register ring)
#endif
#ifndef EGNN_SPLIT_COMMIT_AFTER
#define EGNN_SPLIT_COMMIT_AFTER 1   // group (of TM x TN MFMAs, 6 per step) after which the next tile is cut and written to LDS
#endif

// The software pipeline of one workgroup.  A k-step on the bf16 pipe is 2.7 x shorter than on the fp32 pipe while a
// global load takes as long as before, so operand tiles travel through a RING of D register stages: while step kt
// runs out of LDS buffer kt & 1, stage kt + 1 is cut into planes and written to the other buffer (its loads were issued
// D steps earlier) and the loads of stage kt + 1 + D are issued into the registers it frees.  Slots are compile-time
// (the drivers unroll by lcm(D, 2)), so the ring lives in registers and the compiler's vmcnt counts stay exact.
template <int BM, int BN, int AMAJ, int BMAJ, bool VEC4, bool FULLONLY, class XFA, class XFB, bool GA = false, bool GB = false,
          int D = EGNN_SPLIT_DEPTH, int NTHR = 256>
struct PipelineS {
  using TS = TileShapeS<BM, BN, NTHR / 128>;
  static constexpr int DEPTH = D, UNROLL = (D % 2 == 0) ? D : 2 * D;
  static constexpr int B_OFF = 2 * TS::A_BUF;
  StagerS<BM, AMAJ, VEC4, XFA, GA, NTHR> sa[D];
  StagerS<BN, BMAJ, VEC4, XFB, GB, NTHR> sb[D];
  const float* __restrict__ A;
  const float* __restrict__ B;
  int64_t lda, ldb, M, N, kend;
  XFA xfa;
  XFB xfb;
  const int64_t* arows = nullptr;
  const int64_t* brows = nullptr;

  __device__ __forceinline__ PipelineS(const float* A_, int64_t lda_, int64_t M_, const float* B_, int64_t ldb_, int64_t N_, int64_t kend_,
                                       const XFA& xa, const XFB& xb)
      : A(A_), B(B_), lda(lda_), ldb(ldb_), M(M_), N(N_), kend(kend_), xfa(xa), xfb(xb) {}

  template <int SLOT>
  __device__ __forceinline__ void prefetch(int64_t m0, int64_t n0, int64_t k0) {
    sa[SLOT].template load<FULLONLY>(A, lda, m0, M, k0, kend, xfa, arows);
    sb[SLOT].template load<FULLONLY>(B, ldb, n0, N, k0, kend, xfb, brows);
  }
  template <int SLOT>
  __device__ __forceinline__ void commit(float* smem, int buf) const {
    char* s = reinterpret_cast<char*>(smem);
    sa[SLOT].store(s + buf * TS::A_BUF);
    sb[SLOT].store(s + B_OFF + buf * TS::B_BUF);
  }
  // one k-step out of LDS buffer `cur`; with `more`, ring slot SLOT is committed to buffer cur ^ 1 on the way and
  // refill() (the loads of the stage that takes the slot over) is called right behind it
  template <int SLOT, int TM_, int TN_, class F>
  __device__ __forceinline__ void step(f32x16 (&acc)[TM_][TN_], float* smem, int cur, bool more, int lane, int wm, int wn, F&& refill) {
    static_assert(TM_ == TS::TM && TN_ == TS::TN, "accumulator shape does not match the block tile");
    const char* s = reinterpret_cast<const char*>(smem);
    const char* sA = s + cur * TS::A_BUF + (wm * TS::WM + (lane & 31)) * S_ROW + (lane >> 5) * 16;
    const char* sB = s + B_OFF + cur * TS::B_BUF + (wn * TS::WN + (lane & 31)) * S_ROW + (lane >> 5) * 16;
    u32x4 a[TS::TM][3], b[TS::TN][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int tm = 0; tm < TS::TM; ++tm) a[tm][p] = *reinterpret_cast<const u32x4*>(sA + p * BM * S_ROW + tm * 32 * S_ROW);
#pragma unroll
      for (int tn = 0; tn < TS::TN; ++tn) b[tn][p] = *reinterpret_cast<const u32x4*>(sB + p * BN * S_ROW + tn * 32 * S_ROW);
    }
    // the six kept partial products, smallest first; consecutive MFMAs go to different accumulators
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
#pragma unroll
      for (int tm = 0; tm < TS::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TS::TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[tm][PA[t]]),
                                                                __builtin_bit_cast(bf16x8, b[tn][PB[t]]), acc[tm][tn], 0, 0, 0);
      if (t == EGNN_SPLIT_COMMIT_AFTER && more) {
        commit<SLOT>(smem, cur ^ 1);
        refill();
      }
    }
    // (round 6: __builtin_amdgcn_iglp_opt(0) here was measured on one box, builds interleaved A B A B: 6.611 / 6.589 vs 6.642 / 6.583 ms
    //  per replayed epoch -- no effect on this register-staged loop; not kept.  The DMA pipeline's loops do profit: gemm3.h SCHED_*.)
  }
};

template <int U, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < U) {
    f(std::integral_constant<int, I>{});
    static_for<U, I + 1>(f);
  }
}

// acc += A[m0:m0+BM, kbeg:kend] * B[kbeg:kend, n0:n0+BN] on the split pipeline (all 256 threads call with equal bounds)
template <int BM, int BN, int AMAJ, int BMAJ, bool VEC4, bool FULLONLY = false, bool GA = false, bool GB = false, int NTHR = 256,
          class XFA, class XFB, int TM_, int TN_>
__device__ __forceinline__ void mainloop_split(f32x16 (&acc)[TM_][TN_], const float* __restrict__ A, int64_t lda, int64_t m0, int64_t M,
                                               const float* __restrict__ B, int64_t ldb, int64_t n0, int64_t N, int64_t kbeg,
                                               int64_t kend, const XFA& xfa, const XFB& xfb, float* smem,
                                               const int64_t* arows = nullptr, const int64_t* brows = nullptr) {
  using P = PipelineS<BM, BN, AMAJ, BMAJ, VEC4, FULLONLY, XFA, XFB, GA, GB, EGNN_SPLIT_DEPTH, NTHR>;
  P pipe(A, lda, M, B, ldb, N, kend, xfa, xfb);
  pipe.arows = arows;
  pipe.brows = brows;
  const int nk = (int)((kend - kbeg + BK - 1) / BK);
  if (nk <= 0) return;
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  // Stage s covers k = kbeg + 16 s.  Stages past the end are clamped to the last one and never committed: every step then
  // issues the same loads, which keeps the compiler's vmcnt bookkeeping exact (a conditional refill makes it fall back
  // to vmcnt(0), i.e. to a ring of depth one); the re-loaded lines hit L1 / L2.
  auto k_of = [&](int stage) { return kbeg + (int64_t)(stage < nk ? stage : nk - 1) * BK; };
  static_for<P::DEPTH>([&](auto d) { pipe.template prefetch<d.value>(m0, n0, k_of(d.value)); });
  pipe.template commit<0>(smem, 0);
  pipe.template prefetch<0>(m0, n0, k_of(P::DEPTH));
  __syncthreads();
  int kt0 = 0;
  if constexpr (FULLONLY) {
    // whole groups of UNROLL steps as ONE basic block (commit and refill unconditional; what the last step commits is
    // never read): with a branch between a load and its use the compiler sinks the load next to the use, which turns
    // the ring back into a prefetch distance of one step
    for (; kt0 + P::UNROLL <= nk; kt0 += P::UNROLL) {
      static_for<P::UNROLL>([&](auto u) {
        constexpr int slot = (u.value + 1) % P::DEPTH;
        pipe.template step<slot>(acc, smem, u.value & 1, true, lane, wm, wn,
                                 [&]() { pipe.template prefetch<slot>(m0, n0, k_of(kt0 + u.value + 1 + P::DEPTH)); });
        __syncthreads();
      });
    }
  }
  for (; kt0 < nk; kt0 += P::UNROLL) {
    static_for<P::UNROLL>([&](auto u) {
      const int kt = kt0 + u.value;
      if (kt < nk) {   // block-uniform
        constexpr int slot = (u.value + 1) % P::DEPTH;
        pipe.template step<slot>(acc, smem, u.value & 1, kt + 1 < nk, lane, wm, wn,
                                 [&]() { pipe.template prefetch<slot>(m0, n0, k_of(kt + 1 + P::DEPTH)); });
        __syncthreads();
      }
    });
  }
}

// EGNN_GEMM_PIPE=f32 keeps every product on the f32-input MFMA (the A/B switch of the two pipelines)
static inline bool egnn_split_pipe() {
  static const bool on = !(getenv("EGNN_GEMM_PIPE") && getenv("EGNN_GEMM_PIPE")[0] == 'f');
  return on;
}

// The opt-in for more than 64 KB of dynamic LDS is a per-device attribute of the kernel: set once per (kernel, device) --
// a process that drives several GPUs reaches every one of them (0 = not yet set, 1 = set, -1 = refused).
template <auto Kernel>
static inline bool allow_big_lds(int bytes) {
  constexpr int kMaxDev = 64;
  static signed char state[kMaxDev] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return false;
  if (dev >= kMaxDev) return hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (state[dev] == 0)
    state[dev] = hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 1 : -1;
  return state[dev] == 1;
}

// launch with dynamic LDS (the split image of a 128 x 128 tile is 72 KB: above the 64 KB that need no opt-in)
template <auto Kernel, class... Args>
static inline int launch_dyn_lds(dim3 grid, dim3 block, size_t shm, hipStream_t st, Args... args) {
  if (shm > 65536 && !allow_big_lds<Kernel>(160 * 1024)) return EGNN_ELAUNCH;
  hipLaunchKernelGGL(Kernel, grid, block, shm, st, args...);
  return EGNN_OK;
}

template <bool SPLIT, int BM, int BN>
struct TileSel { using type = TileShape<BM, BN>; };
template <int BM, int BN>
struct TileSel<true, BM, BN> { using type = TileShapeS<BM, BN>; };

// compile-time choice of the pipeline: SPLIT = products on the bf16 pipe (this file), else the f32-input MFMA
template <bool SPLIT, int BM, int BN, int AMAJ, int BMAJ, bool VEC4, bool FULLONLY = false, bool GA = false, bool GB = false, class XFA,
          class XFB, int TM_, int TN_>
__device__ __forceinline__ void mainloop_sel(f32x16 (&acc)[TM_][TN_], const float* __restrict__ A, int64_t lda, int64_t m0, int64_t M,
                                             const float* __restrict__ B, int64_t ldb, int64_t n0, int64_t N, int64_t kbeg,
                                             int64_t kend, const XFA& xfa, const XFB& xfb, float* smem,
                                             const int64_t* arows = nullptr, const int64_t* brows = nullptr) {
  if constexpr (SPLIT) mainloop_split<BM, BN, AMAJ, BMAJ, VEC4, FULLONLY, GA, GB>(acc, A, lda, m0, M, B, ldb, n0, N, kbeg, kend, xfa, xfb, smem, arows, brows);
  else mainloop<BM, BN, AMAJ, BMAJ, VEC4, FULLONLY, GA, GB>(acc, A, lda, m0, M, B, ldb, n0, N, kbeg, kend, xfa, xfb, smem, arows, brows);
}

}  // namespace egnn_gemm
