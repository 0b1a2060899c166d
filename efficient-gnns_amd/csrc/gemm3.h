// fp32 GEMM products on the bf16 matrix pipe, third form ("DMA" pipeline): operand tiles go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass), and the three-way bf16 split of gemm_split.h
//       x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)
// happens on the FRAGMENT side, on the values a wave is about to feed to its MFMAs -- or not in the loop at all, for
// an operand that was cut once into bf16 planes ahead of the call ("planes" operand: layer weights, the unit rows of
// the G-CRD loss).  Same six exact partial products per fp32 product, same accumulation: the arithmetic of
// gemm_split.h, only the data path differs (the staging of both operands through VGPRs, not the
// matrix pipe, bounded the round-2 loop).
//
// Operand forms (of a [rows, K] operand, rows = M for A, N for B):
//   F32K    fp32, k contiguous ([rows, ld]).  LDS stage image: [rows][BK floats], the 16-byte chunks of a row XOR-
//           swizzled (on the SOURCE address of the DMA, the LDS side of an LDS-DMA is lane-linear) so that the
//           fragment ds_read_b128 of 32 consecutive rows are bank-conflict-free.
//   F32M    fp32, row index contiguous ([K, ld]: the operand of a transposed product, E^T G or X^T dY).  LDS stage
//           image: [BK][rows] as in memory; a lane's fragment (8 consecutive k of one row) is 8 ds_read_b32 down a
//           column (conflict-free: 32 lanes = 32 consecutive dwords).
//   PLANES  cut ahead of the call by egnn_gemm3::pack_planes into bf16 planes, TILE-PACKED: unit (row block of RB rows,
//           k-step of BK, plane p) = RB x BK bf16, rows of BK * 2 bytes with the same kind of chunk swizzle baked in;
//           the three planes of a (row block, k-step) are contiguous, so a stage is ONE contiguous global range, copied
//           by lane-linear DMA pieces of 1 KB.  No VALU and no VGPR traffic in the loop for this operand.
//
// Pipeline: NB LDS stages of BK k-values; per step ONE barrier: [wait own DMA of stage s] [barrier] [issue DMA of stage
// s + NB - 1 into the buffer everybody has just finished reading] [MFMA work of stage s].  Block tile (64 TM) x (64 TN),
// four waves 2 x 2, a wave owns TM x TN MFMA tiles of 32 x 32 (v_mfma_f32_32x32x16_bf16, 6 per product tile and k-block).
#pragma once
#include "gemm_split.h"

namespace egnn_gemm3 {

using egnn_gemm::bf16x8;
using egnn_gemm::f32x16;
using egnn_gemm::split8;
using egnn_gemm::u32x4;

constexpr int F32K = 0, PLANES = 1, F32M = 2;
// scheduling hints of the steady-state k-block (the ABL template parameter of `mainloop`; results unchanged).  Measured per form in
// tools/lab/gemm3_lab (-DLAB_SCHED_ONLY; profiles/r06_gemm3_sched_lab.txt): 2-8 % on the shipped forms; iglp_opt(1) does not finish
// compiling on these basic blocks (> 5 min per kernel, tens of GB).
#ifdef EGNN_GEMM3_NO_SCHED          // (A/B builds: tools/r06/ab.sh)
constexpr int SCHED_IGLP0 = 0, SCHED_HAND = 0;
#else
constexpr int SCHED_IGLP0 = 32;    // __builtin_amdgcn_iglp_opt(0)
constexpr int SCHED_HAND = 128;    // sched_group_barrier: one MFMA, then a share of the next k-block's LDS reads and cutting VALU
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- geometry of one operand's LDS stage ---------------------------------------------------------------------------
template <int MODE, int R, int BKT>
struct Stage {
  static constexpr int BYTES = MODE == PLANES ? 3 * R * BKT * 2 : R * BKT * 4;
  static constexpr int PIECES = BYTES / 1024;               // 1 KB = one wave-instruction of global_load_lds_dwordx4
  static_assert(BYTES % 4096 == 0, "a stage is a whole number of DMA pieces per wave");
  // F32K: chunks (16 B) per row and the swizzle of row r
  static constexpr int CH = BKT / 4;                         // 4 (BK 16) or 8 (BK 32)
  static constexpr int RPB = 16 / CH;                        // rows per 256-byte bank row
  static __device__ __forceinline__ int swz(int r) { return (r / RPB) % CH; }
  // PLANES: chunks (16 B = 8 bf16) per row of one plane
  static constexpr int CHP = BKT / 8;                        // 2 (BK 16) or 4 (BK 32)
  static constexpr int RPBP = 16 / CHP;
  static __device__ __forceinline__ int swzp(int r) { return (r / RPBP) % CHP; }
};

__device__ __forceinline__ void dma16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Issues this wave's share of the DMA of one operand stage.
//   F32K / F32M: p = fp32 matrix, ld = leading dimension, r0 = first row of the block tile, k0 = first k of the stage
//   PLANES:      p = packed planes, nks = k-steps of the whole operand, r0 / k0 as above (multiples of R / BKT)
// rmax (F32K): rows at or past it re-read row rmax - 1 (the last row tile of a ragged operand; such rows are never stored);
// rmax (F32M): the same for the k-rows of the transposed operand (a ragged reduction whose other operand is zero-padded planes)
template <int MODE, int R, int BKT>
__device__ __forceinline__ void issue_stage(const void* __restrict__ p, int64_t ld_or_nks, int64_t r0, int64_t k0, char* lds, int wave, int lane,
                                            int64_t rmax = INT64_MAX) {
  using S = Stage<MODE, R, BKT>;
#pragma unroll
  for (int i = 0; i < S::PIECES / 4; ++i) {
    const int q = wave + 4 * i;                              // piece
    char* dst = lds + q * 1024;
    if constexpr (MODE == PLANES) {
      const char* base = (const char*)p + ((r0 / R) * ld_or_nks + k0 / BKT) * (int64_t)S::BYTES;
      dma16(base + q * 1024 + lane * 16, dst);
    } else if constexpr (MODE == F32K) {
      constexpr int RPP = 64 / S::CH;                        // rows per piece
      const int row = q * RPP + lane / S::CH, s = lane % S::CH;
      const int c = s ^ S::swz(row);
      int64_t rr = r0 + row;
      if (rr >= rmax) rr = rmax - 1;
      dma16((const float*)p + rr * ld_or_nks + k0 + 4 * c, dst);
    } else {                                                 // F32M: image [BKT][R] floats, as in memory
      constexpr int CPR = R / 4;                             // 16-byte chunks per k-row
      const int cidx = q * 64 + lane;
      const int kk = cidx / CPR, c = cidx % CPR;
      int64_t kr = k0 + kk;
      if (kr >= rmax) kr = rmax - 1;                         // rmax = k limit here: k-rows past it re-read the last one (their partner planes are zero)
      dma16((const float*)p + kr * ld_or_nks + r0 + 4 * c, dst);
    }
  }
}

// the three bf16 planes of the fragment (8 consecutive k of row `row`, k-block kb, half h) of one operand stage
template <int MODE, int R, int BKT>
__device__ __forceinline__ void load_frag(const char* lds, int row, int kb, int h, u32x4& p0, u32x4& p1, u32x4& p2) {
  using S = Stage<MODE, R, BKT>;
  if constexpr (MODE == PLANES) {
    const int c = (kb * 2 + h) ^ S::swzp(row);
    const char* a = lds + row * (BKT * 2) + c * 16;
    p0 = *reinterpret_cast<const u32x4*>(a);
    p1 = *reinterpret_cast<const u32x4*>(a + R * BKT * 2);
    p2 = *reinterpret_cast<const u32x4*>(a + 2 * R * BKT * 2);
  } else {
    float v[8];
    if constexpr (MODE == F32K) {
      const int c0 = kb * 4 + 2 * h, x = S::swz(row);
      const char* a = lds + row * (BKT * 4);
      const f32x4 lo = *reinterpret_cast<const f32x4*>(a + ((c0 ^ x) * 16));
      const f32x4 hi = *reinterpret_cast<const f32x4*>(a + (((c0 + 1) ^ x) * 16));
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    } else {
      const float* a = reinterpret_cast<const float*>(lds) + (kb * 16 + 8 * h) * R + row;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = a[j * R];
    }
    split8(v, p0, p1, p2);
  }
}

template <int AMODE, int BMODE, int TM, int TN, int BKT, int NB>
struct Tile {
  static constexpr int BM = 64 * TM, BN = 64 * TN;
  using SA = Stage<AMODE, BM, BKT>;
  using SB = Stage<BMODE, BN, BKT>;
  static constexpr int STAGE_BYTES = SA::BYTES + SB::BYTES;
  static constexpr int SMEM_BYTES = NB * STAGE_BYTES;
  static constexpr int DMA_PER_WAVE = (SA::PIECES + SB::PIECES) / 4;   // wave-instructions per stage and wave
};

// one k-block of MFMA work on fragments held in registers: the six kept products, smallest first
template <int TM, int TN>
__device__ __forceinline__ void mfma_block(f32x16 (&acc)[TM][TN], const u32x4 (&a)[TM][3], const u32x4 (&b)[TN][3]) {
  constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[tm][PA[t]]), __builtin_bit_cast(bf16x8, b[tn][PB[t]]),
                                                              acc[tm][tn], 0, 0, 0);
}

// acc += A[m0 : m0 + BM, kbeg : kend] * B[n0 : n0 + BN, kbeg : kend]^T ; kend - kbeg a positive multiple of BKT; whole tiles only
// (the callers route ragged shapes to gemm_split.h).  pa / pb: matrix or packed planes; la / lb: leading dimension or the
// operand's number of k-steps (PLANES).
// PIPE = 0: fragments are read (and cut) at the head of the k-block that consumes them.
// PIPE = 1: one k-block of fragments lives in REGISTERS ahead of the MFMAs: while the 6 TM TN MFMAs of k-block f run out of
//           registers, the fragments of k-block f + 1 are read from LDS and cut -- the split VALU and the LDS latency sit in
//           the shadow of the matrix pipe inside ONE wave (an in-order wave cannot overlap them otherwise: ds_read -> cut ->
//           MFMA is a dependency chain).  The stage read ahead is one beyond the stage computed, so the DMA ring is NB
//           stages deep on top of the register stage.
// ABL (lab only, tools/lab/gemm3_lab.hip; results are then wrong on purpose): 1 = no DMA, 2 = no barriers, 4 = fragments read
// once and reused, 8 = no vmcnt waits, 16 = s_setprio(1) around every MFMA block.
// Scheduling hints (results stay right; round 6): 32 = __builtin_amdgcn_iglp_opt(0) per k-block, 64 = iglp_opt(1),
// 128 = a hand-placed order of the k-block -- one MFMA, then a share of the next k-block's LDS reads and cutting VALU
// (sched_group_barrier), instead of what the compiler's list scheduler picks.
template <int AMODE, int BMODE, int TM, int TN, int BKT, int NB, int PIPE = 0, int ABL = 0>
__device__ __forceinline__ void mainloop(f32x16 (&acc)[TM][TN], const void* __restrict__ pa, int64_t la, int64_t m0, const void* __restrict__ pb,
                                         int64_t lb, int64_t n0, int64_t kbeg, int64_t kend, char* smem, int64_t a_rows = INT64_MAX) {
  using T = Tile<AMODE, BMODE, TM, TN, BKT, NB>;
  constexpr int G = T::DMA_PER_WAVE, KB = BKT / 16;
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (int)((kend - kbeg) / BKT);
  auto issue = [&](int stage) {
    if constexpr (ABL & 1) return;
    char* s = smem + (stage % NB) * T::STAGE_BYTES;
    issue_stage<AMODE, T::BM, BKT>(pa, la, m0, kbeg + (int64_t)stage * BKT, s, wave, lane, a_rows);
    issue_stage<BMODE, T::BN, BKT>(pb, lb, n0, kbeg + (int64_t)stage * BKT, s + T::SA::BYTES, wave, lane);
  };
  bool frags_done = false;
  auto frags = [&](int stage, int kb, u32x4 (&a)[TM][3], u32x4 (&b)[TN][3]) {
    if constexpr (ABL & 4) {
      if (frags_done) {   // keep the registers live and opaque, read nothing
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(a[tm][p]));
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(b[tn][p]));
        return;
      }
      frags_done = stage > 0 || PIPE == 0 || kb > 0;   // PIPE: both register sets get one real read
    }
    const char* sa = smem + (stage % NB) * T::STAGE_BYTES;
    const char* sb = sa + T::SA::BYTES;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) load_frag<AMODE, T::BM, BKT>(sa, wm * 32 * TM + tm * 32 + (lane & 31), kb, lane >> 5, a[tm][0], a[tm][1], a[tm][2]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) load_frag<BMODE, T::BN, BKT>(sb, wn * 32 * TN + tn * 32 + (lane & 31), kb, lane >> 5, b[tn][0], b[tn][1], b[tn][2]);
  };
  if constexpr (PIPE == 0) {
#pragma unroll
    for (int d = 0; d < NB - 1; ++d)
      if (d < nk) issue(d);
    for (int s = 0; s < nk; ++s) {
      // my DMA pieces of stage s have landed (the younger stages may still be in flight), then everybody's
      if constexpr (NB == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (s + NB - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail: fewer stages in flight than the steady state
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my LDS reads of stage s - 1 are complete before its buffer is re-filled
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + NB - 1 < nk) issue(s + NB - 1);                    // into the buffer stage s - 1 was read from
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        u32x4 a[TM][3], b[TN][3];
        frags(s, kb, a, b);
        mfma_block<TM, TN>(acc, a, b);
      }
    }
  } else {
    // DMA ring: stages s + 1 .. s + NB are in LDS / in flight while stage s is consumed out of registers
    u32x4 a0[TM][3], b0[TN][3], a1[TM][3], b1[TN][3];
#pragma unroll
    for (int d = 0; d < NB; ++d)
      if (d < nk) issue(d);
    if (nk >= NB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    frags(0, 0, a0, b0);
    // One k-block: MFMAs on (ac, bc) while (an, bn) are filled with the next k-block's fragments.  STEADY (compile time): stage
    // s + NB exists, so every wait / issue / read below is unconditional and the whole k-block is ONE basic block -- only then
    // does the scheduler weave the cutting VALU and the LDS reads between the MFMAs (with a branch in between they end up in
    // a block of their own, in front of 6 TM TN back-to-back MFMAs).
    auto block = [&](auto steady_c, int s, int kb, u32x4 (&ac)[TM][3], u32x4 (&bc)[TN][3], u32x4 (&an)[TM][3], u32x4 (&bn)[TN][3]) {
      constexpr bool STEADY = decltype(steady_c)::value;
      if (kb == KB - 1) {                                        // the next fragments come from stage s + 1
        if constexpr (!(ABL & 8)) {
          if constexpr (STEADY) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * G) : "memory");
          } else {
            if (s + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
        if constexpr (!(ABL & 2)) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // my LDS reads of stage s are complete before its buffer is re-filled
          __builtin_amdgcn_s_barrier();                          // stage s + 1 landed for everybody; everybody has read stage s
        }
        asm volatile("" ::: "memory");
        if constexpr (STEADY) {
          issue(s + NB);                                         // into the buffer of stage s
          frags(s + 1, 0, an, bn);
        } else {
          if (s + NB < nk) issue(s + NB);
          if (s + 1 < nk) frags(s + 1, 0, an, bn);
        }
      } else {
        frags(s, kb + 1, an, bn);
      }
      if constexpr (ABL & 16) __builtin_amdgcn_s_setprio(1);
      mfma_block<TM, TN>(acc, ac, bc);
      if constexpr (ABL & 16) __builtin_amdgcn_s_setprio(0);
      if constexpr (ABL & 32) __builtin_amdgcn_iglp_opt(0);
      if constexpr (ABL & 64) __builtin_amdgcn_iglp_opt(1);
      if constexpr (ABL & 128) {
        // per k-block: 6 TM TN MFMAs; fragments of the next k-block: (TM + TN) x {PLANES: 3 ds_read_b128; F32K: 2 ds_read_b128 + ~44
        // VALU; F32M: 8 ds_read_b32 + ~44 VALU}
        constexpr int NM = 6 * TM * TN;
        constexpr int DSA = AMODE == PLANES ? 3 : (AMODE == F32K ? 2 : 8), DSB = BMODE == PLANES ? 3 : (BMODE == F32K ? 2 : 8);
        constexpr int NDS = TM * DSA + TN * DSB;
        constexpr int NVA = (AMODE == PLANES ? 0 : 44 * TM) + (BMODE == PLANES ? 0 : 44 * TN);
        constexpr int DS_PER = (NDS + NM - 1) / NM, VA_PER = (NVA + NM - 1) / NM;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // one MFMA
          if (i * DS_PER < NDS) __builtin_amdgcn_sched_group_barrier(0x100, DS_PER, 0);   // LDS reads of the next fragments, front-loaded
          if (VA_PER > 0) __builtin_amdgcn_sched_group_barrier(0x002, VA_PER, 0);   // their cutting VALU, spread over the MFMAs
        }
      }
    };
    // k-blocks alternate between the two register sets; a group of GS stages is an even number of k-blocks, so that both
    // sets are compile-time names inside the (unrolled) group and every group starts on set 0
    constexpr int GS = KB % 2 == 0 ? 1 : 2;
    auto group = [&](auto steady_c, int s) {
      if constexpr (KB == 1) {
        block(steady_c, s, 0, a0, b0, a1, b1);
        block(steady_c, s + 1, 0, a1, b1, a0, b0);
      } else {
#pragma unroll
        for (int kb = 0; kb < KB; kb += 2) {
          block(steady_c, s, kb, a0, b0, a1, b1);
          block(steady_c, s, kb + 1, a1, b1, a0, b0);
        }
      }
    };
    int s = 0;
    for (; s + GS - 1 + NB < nk; s += GS) group(std::true_type{}, s);
    for (; s + GS <= nk; s += GS) group(std::false_type{}, s);
    if (s < nk) block(std::false_type{}, s, 0, a0, b0, a1, b1);   // KB == 1 and an odd number of stages left
  }
}

// ---- cutting an operand into tile-packed planes --------------------------------------------------------------------
__host__ __device__ inline size_t planes_bytes(int64_t rows, int64_t K, int RB, int BKT) {
  return (size_t)((rows + RB - 1) / RB) * (size_t)((K + BKT - 1) / BKT) * 3 * RB * BKT * 2;
}

// X(r, k): k_major = 1 -> X[r * ld + k], else X[k * ld + r].  One thread per 16-byte output chunk (8 k-values of one row);
// rows / k past the end are zero planes.  Optional factors applied before the cut: scale[row] (per operand row) and
// exp(kshift - klse[k]) (per k: the row weights w_i of the G-CRD backward, criterion.py:139-145 via nce.hip).  ridx: row gather;
// kidx: gather along k (k-value j of the operand is storage index kidx[j]: the train rows of a [N, features] matrix as the
// reduction dimension of dW = dY^T X[train_idx]).
template <int RB, int BKT>
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ X, int64_t ld, int k_major, int64_t rows, int64_t K,
                                                          const float* __restrict__ scale, const int64_t* __restrict__ ridx,
                                                          const float* __restrict__ klse, float kshift, char* __restrict__ out,
                                                          const int64_t* __restrict__ kidx = nullptr, int64_t K_pad = 0) {
  using S = Stage<PLANES, RB, BKT>;
  // K_pad > K: zero planes up to K_pad (whole k-ranges of a split reduction)
  const int64_t nks = ((K_pad > K ? K_pad : K) + BKT - 1) / BKT, nrb = (rows + RB - 1) / RB;
  const int64_t total = nrb * nks * RB * S::CHP;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    // consecutive threads walk the contiguous direction of X: chunks of one row (k-major) or rows of one chunk
    const int c = k_major ? (int)(t % S::CHP) : (int)((t / RB) % S::CHP);
    const int r = k_major ? (int)((t / S::CHP) % RB) : (int)(t % RB);
    const int64_t unit = t / (S::CHP * RB), ks = unit % nks, rb = unit / nks;
    const int64_t row = rb * RB + r, k0 = ks * BKT + c * 8;
    float v[8];
    const int64_t rs = (row < rows && ridx) ? ridx[row] : row;
    const float sc = (scale && row < rows) ? scale[row] : 1.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t k = k0 + j;
      float x = 0.f;
      if (row < rows && k < K) {
        const int64_t ks = kidx ? kidx[k] : k;
        x = sc * (k_major ? X[rs * ld + ks] : X[ks * ld + rs]);
        if (klse) x *= expf(kshift - klse[k]);
      }
      v[j] = x;
    }
    u32x4 p0, p1, p2;
    split8(v, p0, p1, p2);
    char* o = out + unit * (int64_t)S::BYTES + r * (BKT * 2) + ((c ^ S::swzp(r)) * 16);
    *reinterpret_cast<u32x4*>(o) = p0;
    *reinterpret_cast<u32x4*>(o + RB * BKT * 2) = p1;
    *reinterpret_cast<u32x4*>(o + 2 * RB * BKT * 2) = p2;
  }
}

template <int RB, int BKT>
static inline void pack_planes(const float* X, int64_t ld, int k_major, int64_t rows, int64_t K, const float* scale, const int64_t* ridx,
                               const float* klse, float kshift, char* out, hipStream_t st, const int64_t* kidx = nullptr) {
  const int64_t total = ((rows + RB - 1) / RB) * ((K + BKT - 1) / BKT) * RB * (BKT / 8);
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL((pack_planes_kernel<RB, BKT>), dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, X, ld, k_major, rows, K,
                     scale, ridx, klse, kshift, out, kidx);
}

}  // namespace egnn_gemm3
