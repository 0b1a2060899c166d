// Row-block neighbour aggregation for gfx950 (egnn_spmm_csr_blk_f32): Y[i,:] = (sum_e val[e] * X[col[e],:]) * inv_i + bias
//
// Second-generation schedule of the aggregation behind torch_sparse::spmm (/root/reference/arxiv_pyg/gnn.py:47,52,79,84,
// /root/reference/mag_pyg/gnn.py:162), HBM/L2-bound, no MFMA.  What differs from the per-range kernels of spmm.hip:
//   * ONE launch walks a list of work items per 128-byte column slice (slice <-> XCD binding as before): first the
//     hub-row segments (rows longer than seg_max entries, cut into <= seg_max-entry ranges that write partial slots,
//     one range per 8-lane sub-group -- the longest items are dispatched first), then blocks of consecutive rows whose
//     sub-groups walk the block's rows round-robin.  spmm_combine_kernel (spmm.hip) finishes the hub rows;
//   * 32-bit everything on the gather path: int32 indices, byte offsets premultiplied once per entry, X addressed
//     through a buffer descriptor (buffer_load_dwordx4, 32-bit voffset).  Padded lanes point their offset OUT OF RANGE
//     of the descriptor: the hardware returns zeros without touching memory, which replaces the per-element selects
//     of the first kernel (0 * 0 added to the accumulator is exact); lane-group broadcasts are ds_swizzle
//     bit-mask patterns (no address register);
//   * any K % 4 == 0: ceil(K / 32) slices, lanes past K are masked through the same out-of-range offset;
//   * optional LDS staging of the block's OWN source rows (`win` != NULL, a graph in a locality order): 16-wave
//     workgroups, the block's X rows are streamed into LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip,
//     asynchronous) WHILE the sub-groups gather the rows' out-of-block entries from L2 / fabric; after one barrier the
//     in-block entries -- a contiguous sub-range of every row, columns being sorted -- are read from LDS (ds_read_b128);
//   * optional BatchNorm statistics in the epilogue (`stat_part` != NULL, /root/reference/arxiv_pyg/gnn.py:47-48): every
//     wave leaves the shifted column sums of the rows it stored (registers -> a swizzle tree over its sub-groups; no
//     barrier, no second pass over Y), folded in a fixed order by egnn_bn_stats_merge_f32 -- the separate full pass over
//     Y disappears.  (Measured alternatives: a workgroup-level reduction costs a barrier per 32 rows, +25 % kernel time;
//     re-reading the block from L2 likewise.)
// Accumulation order is fixed by the schedule => run-to-run bit-stable.
#include "common.h"

namespace {

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct BlkArgs {
  int64_t n_rows, K;
  const int32_t* rowptr;
  const int32_t* col;
  const float* val;
  const float* src_scale;
  const float* bias;
  const float* X;
  int64_t ldx;
  float* Y;
  int64_t ldy;
  int mean;
  int seg_max;
  int rows_per_blk;
  const int32_t* blk_ptr;  // variable row blocks (nullable: fixed rows_per_blk)
  int64_t n_blk;
  const int32_t* win;      // [n_rows][2]: entries [win[2r], win[2r+1]) of row r have their source inside the row's block
  const int32_t* hseg;     // [n_hseg][4]: (first entry, end entry, partial slot, 0) of the hub-row segments
  int64_t n_hseg;
  float* P;                // [slots][K] partial sums of the hub segments
  const float* addend;     // nullable [n_rows][ld_add]: added to every stored row (accumulating aggregation)
  int64_t ld_add;
  float* stat_part;        // [n_blk * waves per workgroup][2][K]
  const float* stat_shift; // [K] nullable
  int NS, map_mode;
  uint32_t x_bytes;
  int flags;
};

constexpr int kFlagStoreNt = 2;     // Y rows: non-temporal stores
constexpr int kFlagStoreSc1 = 4;    // Y rows: write-through (sc1) stores, line dropped from L2
constexpr int kFlagRelu = 8;        // Y rows: max(y, 0) on the way out (eval-mode BatchNorm folded into weights / bias + ReLU); not with stat_part
constexpr uint32_t kOob = 0x80000000u;   // >= num_records of any descriptor built here (X is < 2^31 bytes): reads return 0

// broadcast lane (8*g + J) of every 8-lane group g to the lanes of that group: ds_swizzle in bit-mask mode
// (lane' = (lane & 0x18) | J inside each half-wave), no address register
template <int J>
__device__ __forceinline__ uint32_t bcast8(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, /*and*/ 0x18 | (/*or*/ J << 5));
}
template <int J>
__device__ __forceinline__ float bcast8f(float v) { return __uint_as_float(bcast8<J>(__float_as_uint(v))); }

__device__ __forceinline__ float4 buf_load4(rsrc_t rsrc, uint32_t off) {
  const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
  return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}

template <int J>
__device__ __forceinline__ void gather_one(rsrc_t rsrc, uint32_t off_l, float v_l, int n, uint32_t lane_off, float4& x, float& v) {
  const uint32_t o = bcast8<J>(off_l);
  v = bcast8f<J>(v_l);
#ifdef EGNN_SPMM_PRED
  // (lab switch, round 6) padded slots issue NO request: the load is predicated on the slot being a real entry (29 % of the slots of the
  // headline graph's short rows are padding at 8 entries per chunk)
  x = make_float4(0.f, 0.f, 0.f, 0.f);
  if (J < n) x = buf_load4(rsrc, o + lane_off);
#else
  x = buf_load4(rsrc, J < n ? o + lane_off : kOob);   // lane_off is kOob itself for lanes past K
#endif
}

__device__ __forceinline__ void fma4(float v, const float4& x, float (&acc)[4]) {
  acc[0] = fmaf(v, x.x, acc[0]);
  acc[1] = fmaf(v, x.y, acc[1]);
  acc[2] = fmaf(v, x.z, acc[2]);
  acc[3] = fmaf(v, x.w, acc[3]);
}

// one chunk of <= 8 entries of a sub-group's row: off_l / v_l hold entry `li` of the chunk (0 weight past n)
__device__ __forceinline__ void gather_chunk(rsrc_t rsrc, uint32_t off_l, float v_l, int n, uint32_t lane_off, float (&acc)[4]) {
  float4 x0, x1, x2, x3, x4, x5, x6, x7;
  float v0, v1, v2, v3, v4, v5, v6, v7;
  gather_one<0>(rsrc, off_l, v_l, n, lane_off, x0, v0);
  gather_one<1>(rsrc, off_l, v_l, n, lane_off, x1, v1);
  gather_one<2>(rsrc, off_l, v_l, n, lane_off, x2, v2);
  gather_one<3>(rsrc, off_l, v_l, n, lane_off, x3, v3);
  gather_one<4>(rsrc, off_l, v_l, n, lane_off, x4, v4);
  gather_one<5>(rsrc, off_l, v_l, n, lane_off, x5, v5);
  gather_one<6>(rsrc, off_l, v_l, n, lane_off, x6, v6);
  gather_one<7>(rsrc, off_l, v_l, n, lane_off, x7, v7);
  fma4(v0, x0, acc); fma4(v1, x1, acc); fma4(v2, x2, acc); fma4(v3, x3, acc);
  fma4(v4, x4, acc); fma4(v5, x5, acc); fma4(v6, x6, acc); fma4(v7, x7, acc);
}

// entries [start, end) of one row (or hub segment) gathered from L2 / fabric into acc
__device__ __forceinline__ void gather_range(const BlkArgs& a, rsrc_t rsrc, int start, int end, int li, uint32_t lane_off,
                                             uint32_t row_bytes, float (&acc)[4]) {
  for (int e = start; e < end; e += 8) {
    const int n = end - e < 8 ? end - e : 8;
    uint32_t off_l = 0;
    float v_l = 0.f;
    if (li < n) {
      const int c = a.col[e + li];
      off_l = (uint32_t)c * row_bytes;
      v_l = a.val ? a.val[e + li] : 1.f;
      if (a.src_scale) v_l *= a.src_scale[c];
    }
    gather_chunk(rsrc, off_l, v_l, n, lane_off, acc);
  }
}

template <int J>
__device__ __forceinline__ void lds_one(const float4* sX, uint32_t idx_l, float v_l, int li, float4& x, float& v) {
  const uint32_t r = bcast8<J>(idx_l);
  v = bcast8f<J>(v_l);
  x = sX[r * 8 + li];
}

__device__ __forceinline__ void lds_chunk(const float4* sX, uint32_t idx_l, float v_l, int li, float (&acc)[4]) {
  float4 x0, x1, x2, x3, x4, x5, x6, x7;
  float v0, v1, v2, v3, v4, v5, v6, v7;
  lds_one<0>(sX, idx_l, v_l, li, x0, v0);
  lds_one<1>(sX, idx_l, v_l, li, x1, v1);
  lds_one<2>(sX, idx_l, v_l, li, x2, v2);
  lds_one<3>(sX, idx_l, v_l, li, x3, v3);
  lds_one<4>(sX, idx_l, v_l, li, x4, v4);
  lds_one<5>(sX, idx_l, v_l, li, x5, v5);
  lds_one<6>(sX, idx_l, v_l, li, x6, v6);
  lds_one<7>(sX, idx_l, v_l, li, x7, v7);
  fma4(v0, x0, acc); fma4(v1, x1, acc); fma4(v2, x2, acc); fma4(v3, x3, acc);
  fma4(v4, x4, acc); fma4(v5, x5, acc); fma4(v6, x6, acc); fma4(v7, x7, acc);
}

__device__ __forceinline__ void store_row4(float* p, const float4& v, int flags) {
  v4f w = {v.x, v.y, v.z, v.w};
  if (flags & kFlagRelu) w = v4f{fmaxf(w.x, 0.f), fmaxf(w.y, 0.f), fmaxf(w.z, 0.f), fmaxf(w.w, 0.f)};
  if (flags & kFlagStoreSc1) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");   // s_nop: store-data hazard the assembler does not see
  } else if (flags & kFlagStoreNt) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(w.x, w.y, w.z, w.w);
  }
}

// WAVES x 64 threads; LDS: 16-wave workgroups with the block's X rows staged by LDS-DMA; ROWS = rows per sub-group the
// LDS variant keeps in registers between its two phases (rows_per_blk <= ROWS * WAVES * 8)
template <bool LDS, bool STATS, int WAVES, int ROWS>
__global__ __launch_bounds__(WAVES * 64) void spmm_blk_kernel(const BlkArgs a) {
  extern __shared__ float4 sX[];  // LDS variant: rows_per_blk source rows (this slice) + one all-zero row
  constexpr int NSUB = WAVES * 8;
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int li = lane & 7;
  const int subw = wave * 8 + (lane >> 3);
  const unsigned b = blockIdx.x;
  const int n_hgrp = (int)((a.n_hseg + NSUB - 1) / NSUB);
  const int n_items = n_hgrp + (int)a.n_blk;
  int slice, item;
  if (a.map_mode == 1) {  // NS in {1,2,4,8}: XCD x = b % 8 owns slice x % NS
    const int x = (int)(b & 7);
    const int r = 8 / a.NS;
    slice = x % a.NS;
    item = (int)(b >> 3) * r + x / a.NS;
  } else if (a.map_mode == 2) {  // NS multiple of 8: XCD x owns slices x, x+8, ...
    const int x = (int)(b & 7);
    const int q = (int)(b >> 3);
    const int per = a.NS >> 3;
    slice = x + 8 * (q % per);
    item = q / per;
  } else {
    slice = (int)(b % (unsigned)a.NS);
    item = (int)(b / (unsigned)a.NS);
  }
  if (item >= n_items) return;
  const int col0 = slice * 32 + li * 4;
  const bool colok = col0 < (int)a.K;
  const uint32_t lane_off = colok ? (uint32_t)li * 16u : kOob;
  const uint32_t row_bytes = (uint32_t)a.ldx * 4u;
  // descriptor of this slice's columns of X: base = X + slice*32 floats, offsets = source row * row pitch (+ lane*16)
  const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X + slice * 32), 0,
                                                        (int)(a.x_bytes - (uint32_t)slice * 128u), 0x00020000);
  if (item < n_hgrp) {  // ---- hub segments: one per sub-group, raw partial sums ------------------------------------
    const int64_t s = (int64_t)item * NSUB + subw;
    if (s < a.n_hseg) {
      const int start = a.hseg[4 * s], end = a.hseg[4 * s + 1], slot = a.hseg[4 * s + 2];
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      gather_range(a, rsrc, start, end, li, lane_off, row_bytes, acc);
      if (colok) *reinterpret_cast<float4*>(a.P + (int64_t)slot * a.K + col0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    return;
  }
  const int blk = item - n_hgrp;
  const int row0 = a.blk_ptr ? a.blk_ptr[blk] : blk * a.rows_per_blk;
  int row1 = a.blk_ptr ? a.blk_ptr[blk + 1] : row0 + a.rows_per_blk;
  if (row1 > (int)a.n_rows) row1 = (int)a.n_rows;
  const int nrows = row1 - row0;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};   // STATS: sum (y - shift), sum (y - shift)^2 of my rows
  auto tally = [&](const float4& y) {
    float4 sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.stat_shift) sh = *reinterpret_cast<const float4*>(a.stat_shift + col0);
    const float d0 = y.x - sh.x, d1 = y.y - sh.y, d2 = y.z - sh.z, d3 = y.w - sh.w;
    st1[0] += d0; st1[1] += d1; st1[2] += d2; st1[3] += d3;
    st2[0] = fmaf(d0, d0, st2[0]); st2[1] = fmaf(d1, d1, st2[1]); st2[2] = fmaf(d2, d2, st2[2]); st2[3] = fmaf(d3, d3, st2[3]);
  };

  if constexpr (LDS) {
    // phase 0: start streaming the block's X rows into LDS (8 rows = 1 KiB per wave instruction, lane-contiguous)
    for (int rb = wave * 8; rb < nrows; rb += NSUB) {
      const int r = rb + (lane >> 3);
      if (r < nrows && colok)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.X + (int64_t)(row0 + r) * a.ldx + col0),
                                         (__attribute__((address_space(3))) void*)(sX + rb * 8), 16, 0, 0);
    }
    if (threadIdx.x < 8) sX[a.rows_per_blk * 8 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    // phase 1: out-of-block entries of my rows from L2 / fabric: [start, wa) then [wb, end)
    float acc[ROWS][4];
    int cnt_[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.f;
      cnt_[j] = -1;
      const int r = j * NSUB + subw;
      if (r < nrows) {
        const int row = row0 + r;
        const int start = a.rowptr[row], end = a.rowptr[row + 1];
        if (end - start <= a.seg_max) {
          cnt_[j] = end - start;
          const int wa = a.win[2 * row], wb = a.win[2 * row + 1];
          gather_range(a, rsrc, start, wa, li, lane_off, row_bytes, acc[j]);
          gather_range(a, rsrc, wb, end, li, lane_off, row_bytes, acc[j]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my LDS-DMA pieces have landed
    __syncthreads();
    // phase 2: in-block entries from LDS, then the store
    if (a.bias && colok) bias = *reinterpret_cast<const float4*>(a.bias + col0);
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      if (cnt_[j] < 0) continue;
      const int row = row0 + j * NSUB + subw;
      const int wa = a.win[2 * row], wb = a.win[2 * row + 1];
      for (int e = wa; e < wb; e += 8) {
        const int n = wb - e < 8 ? wb - e : 8;
        uint32_t idx_l = (uint32_t)a.rows_per_blk;  // zero row
        float v_l = 0.f;
        if (li < n) {
          const int c = a.col[e + li];
          idx_l = (uint32_t)(c - row0);
          v_l = a.val ? a.val[e + li] : 1.f;
          if (a.src_scale) v_l *= a.src_scale[c];
        }
        lds_chunk(sX, idx_l, v_l, li, acc[j]);
      }
      if (colok) {
        const float inv = a.mean ? 1.f / (float)(cnt_[j] > 0 ? cnt_[j] : 1) : 1.f;
        float4 y = make_float4(acc[j][0] * inv + bias.x, acc[j][1] * inv + bias.y, acc[j][2] * inv + bias.z, acc[j][3] * inv + bias.w);
        if (a.addend) { const float4 ad = *reinterpret_cast<const float4*>(a.addend + (int64_t)row * a.ld_add + col0); y.x += ad.x; y.y += ad.y; y.z += ad.z; y.w += ad.w; }
        store_row4(a.Y + (int64_t)row * a.ldy + col0, y, a.flags);
        if constexpr (STATS) tally(y);
      }
    }
  } else if constexpr (ROWS == 1) {
    // rows_per_blk == 32: every sub-group owns exactly one row -- no loop, nothing but the row's accumulators is live
    // during the gathers (the statistics are formed from y after the store)
    const int r = subw;
    if (r < nrows) {
      const int row = row0 + r;
      const int start = a.rowptr[row], end = a.rowptr[row + 1];
      const int cnt = end - start;
      if (cnt <= a.seg_max) {  // else hub row: written by the combine kernel
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        gather_range(a, rsrc, start, end, li, lane_off, row_bytes, acc);
        if (colok) {
          if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + col0);
          const float inv = a.mean ? 1.f / (float)(cnt > 0 ? cnt : 1) : 1.f;
          float4 y = make_float4(acc[0] * inv + bias.x, acc[1] * inv + bias.y, acc[2] * inv + bias.z, acc[3] * inv + bias.w);
          if (a.addend) { const float4 ad = *reinterpret_cast<const float4*>(a.addend + (int64_t)row * a.ld_add + col0); y.x += ad.x; y.y += ad.y; y.z += ad.z; y.w += ad.w; }
          store_row4(a.Y + (int64_t)row * a.ldy + col0, y, a.flags);
          if constexpr (STATS) tally(y);
        }
      }
    }
  } else {
    for (int rb = 0; rb < nrows; rb += NSUB) {
      const int r = rb + subw;
      if (r >= nrows) continue;
      const int row = row0 + r;
      const int start = a.rowptr[row], end = a.rowptr[row + 1];
      const int cnt = end - start;
      if (cnt > a.seg_max) continue;  // hub row: written by the combine kernel
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      gather_range(a, rsrc, start, end, li, lane_off, row_bytes, acc);
      if (colok) {
        if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + col0);   // L1-resident; not held across the gathers
        const float inv = a.mean ? 1.f / (float)(cnt > 0 ? cnt : 1) : 1.f;
        float4 y = make_float4(acc[0] * inv + bias.x, acc[1] * inv + bias.y, acc[2] * inv + bias.z, acc[3] * inv + bias.w);
        if (a.addend) { const float4 ad = *reinterpret_cast<const float4*>(a.addend + (int64_t)row * a.ld_add + col0); y.x += ad.x; y.y += ad.y; y.z += ad.z; y.w += ad.w; }
        store_row4(a.Y + (int64_t)row * a.ldy + col0, y, a.flags);
        if constexpr (STATS) tally(y);
      }
    }
  }

  if constexpr (STATS) {
    // per-WAVE partials, no barrier and no second pass: the 8 sub-groups of a wave hold the same 4 columns in lanes of
    // equal `li`; they are added by xor-swizzles over lane bits 3..5 (a fixed tree) and lanes 0..7 write the wave's
    // [2][32 columns of this slice] shifted sums.  egnn_bn_stats_merge_f32 folds the partials in index order.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        st1[q] += __shfl_xor(st1[q], o);
        st2[q] += __shfl_xor(st2[q], o);
      }
    }
    if (lane < 8 && colok) {
      float* sp = a.stat_part + ((int64_t)(blk * WAVES + wave) * 2) * a.K + col0;
      *reinterpret_cast<float4*>(sp) = make_float4(st1[0], st1[1], st1[2], st1[3]);
      *reinterpret_cast<float4*>(sp + a.K) = make_float4(st2[0], st2[1], st2[2], st2[3]);
    }
  }
}

// per-row in-block entry ranges: win[2r] = first entry of row r with col >= blk_start(r), win[2r+1] = first with col >= blk_end(r)
__global__ __launch_bounds__(256) void spmm_blk_window_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                             int64_t n_rows, int rows_per_blk, const int32_t* __restrict__ blk_ptr,
                                                             int64_t n_blk, int32_t* __restrict__ win) {
  const int64_t row = blockIdx.x * 256LL + threadIdx.x;
  if (row >= n_rows) return;
  int64_t lo_id, hi_id;
  if (blk_ptr) {
    int64_t lo = 0, hi = n_blk;  // last block whose start <= row
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (blk_ptr[mid] <= row) lo = mid; else hi = mid;
    }
    lo_id = blk_ptr[lo];
    hi_id = blk_ptr[lo + 1];
  } else {
    lo_id = row / rows_per_blk * rows_per_blk;
    hi_id = lo_id + rows_per_blk;
  }
  const int s = rowptr[row], e = rowptr[row + 1];
  auto lower = [&](int64_t key) {
    int l = s, h = e;
    while (l < h) {
      const int m = (l + h) >> 1;
      if (col[m] < key) l = m + 1; else h = m;
    }
    return l;
  };
  win[2 * row] = lower(lo_id);
  win[2 * row + 1] = lower(hi_id);
}

// Statistics merge, stage 1: the [n_blk][2][C] block partials are folded into kStatSplits partial rows.  Workgroup
// (column group of 32, split s): thread t adds column (t & 31) over blocks first + (t >> 5), + 8, ... of its split (128-byte
// coalesced reads), the 8 row groups are added in index order.  Fixed order => deterministic.
constexpr int kStatSplits = 512;
__global__ __launch_bounds__(256) void bn_stats_fold_kernel(const float* __restrict__ part, int64_t n_blk, int64_t C,
                                                            float* __restrict__ fold) {
  __shared__ float sh[2][8][32];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t col = (int64_t)blockIdx.x * 32 + c;
  const int s = blockIdx.y;
  const int64_t per = (n_blk + kStatSplits - 1) / kStatSplits;
  const int64_t b0 = s * per, b1 = b0 + per < n_blk ? b0 + per : n_blk;
  float sa = 0.f, sb = 0.f;
  if (col < C) {
    int64_t i = b0 + g;
    for (; i + 24 < b1; i += 32) {   // four independent loads per sum in flight
      const float a0 = part[(i * 2) * C + col], a1 = part[((i + 8) * 2) * C + col], a2 = part[((i + 16) * 2) * C + col],
                  a3 = part[((i + 24) * 2) * C + col];
      const float c0 = part[(i * 2 + 1) * C + col], c1 = part[((i + 8) * 2 + 1) * C + col], c2 = part[((i + 16) * 2 + 1) * C + col],
                  c3 = part[((i + 24) * 2 + 1) * C + col];
      sa += a0; sa += a1; sa += a2; sa += a3;
      sb += c0; sb += c1; sb += c2; sb += c3;
    }
    for (; i < b1; i += 8) {
      sa += part[(i * 2) * C + col];
      sb += part[(i * 2 + 1) * C + col];
    }
  }
  sh[0][g][c] = sa;
  sh[1][g][c] = sb;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5;
    float t = 0.f;
    for (int q = 0; q < 8; ++q) t += sh[which][q][c];
    if (col < C) fold[((int64_t)s * 2 + which) * C + col] = t;
  }
}

// stage 2: mean / biased variance from the folded partials plus the rows the block kernel skipped (hub rows, read back
// from Y).  8 columns per workgroup x 32 groups; fixed order.
__global__ __launch_bounds__(256) void bn_stats_merge_kernel(const float* __restrict__ part, int64_t n_part, int64_t C,
                                                             const float* __restrict__ Y, int64_t ldy, const int64_t* __restrict__ extra_rows,
                                                             int64_t n_extra, const float* __restrict__ shift, int64_t n_total,
                                                             float* __restrict__ mean, float* __restrict__ var) {
  __shared__ float sh[2][32][8];
  const int colo = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int64_t c = (int64_t)blockIdx.x * 8 + colo;
  float sa = 0.f, sb = 0.f;
  if (c < C) {
    const float sft = shift ? shift[c] : 0.f;
    for (int64_t i = grp; i < n_part; i += 32) {
      sa += part[(i * 2) * C + c];
      sb += part[(i * 2 + 1) * C + c];
    }
    for (int64_t i = grp; i < n_extra; i += 32) {
      const float d = Y[extra_rows[i] * ldy + c] - sft;
      sa += d;
      sb = fmaf(d, d, sb);
    }
  }
  sh[0][grp][colo] = sa;
  sh[1][grp][colo] = sb;
  __syncthreads();
  if (grp == 0 && c < C) {
    float a = 0.f, b = 0.f;
    for (int g = 0; g < 32; ++g) { a += sh[0][g][colo]; b += sh[1][g][colo]; }
    const float inv = 1.f / (float)n_total;
    const float m1 = a * inv;
    mean[c] = (shift ? shift[c] : 0.f) + m1;
    var[c] = fmaxf(b * inv - m1 * m1, 0.f);
  }
}

template <bool LDS, bool STATS, int WAVES, int ROWS>
int launch_blk(const BlkArgs& a, unsigned grid, size_t shm, hipStream_t st) {
  if (shm > 65536 && hipFuncSetAttribute((const void*)spmm_blk_kernel<LDS, STATS, WAVES, ROWS>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
    return EGNN_ELAUNCH;
  hipLaunchKernelGGL((spmm_blk_kernel<LDS, STATS, WAVES, ROWS>), dim3(grid), dim3(WAVES * 64), shm, st, a);
  return egnn_launch_status();
}

}  // namespace

extern "C" int64_t egnn_spmm_blk_stat_rows(int64_t n_rows, int rows_per_blk, int lds) {
  if (n_rows <= 0 || rows_per_blk <= 0) return 0;
  return (n_rows + rows_per_blk - 1) / rows_per_blk * (lds ? 16 : 4);   // one partial row per wave of every row block
}

extern "C" int egnn_spmm_blk_window_i32(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int rows_per_blk,
                                        const int32_t* blk_ptr, int64_t n_blk, int32_t* win, void* stream) {
  EGNN_CHECK_ARG(n_rows >= 0 && rows_per_blk > 0 && (blk_ptr == nullptr || n_blk > 0));
  if (n_rows == 0) return EGNN_OK;
  EGNN_CHECK_ARG(rowptr && win);
  hipLaunchKernelGGL(spmm_blk_window_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col,
                     n_rows, rows_per_blk, blk_ptr, n_blk, win);
  return egnn_launch_status();
}

extern "C" int egnn_spmm_csr_blk_f32(int64_t n_rows, int64_t n_src, int64_t K, const int32_t* rowptr, const int32_t* col,
                                     const float* val, const float* src_scale, const float* bias, const float* X, int64_t ldx,
                                     float* Y, int64_t ldy, int reduce, int seg_max, int rows_per_blk, const int32_t* blk_ptr,
                                     int64_t n_blk, const int32_t* win, const int32_t* hub_seg, int64_t n_hub_seg, float* partial,
                                     const float* addend, int64_t ld_addend, float* stat_part, const float* stat_shift, int flags,
                                     void* stream) {
  EGNN_CHECK_ARG(n_rows >= 0 && n_src >= 0 && K >= 0 && ldx >= K && ldy >= K);
  EGNN_CHECK_ARG(reduce == EGNN_SUM || reduce == EGNN_MEAN);
  EGNN_CHECK_ARG(rows_per_blk > 0 && rows_per_blk % 32 == 0 && seg_max > 0 && n_hub_seg >= 0);
  if (n_rows == 0 || K == 0) return EGNN_OK;
  EGNN_CHECK_ARG(rowptr && col && X && Y && (n_hub_seg == 0 || (hub_seg && partial)));
  if (K % 4 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || !egnn_aligned16(X) || !egnn_aligned16(Y) || (bias && !egnn_aligned16(bias)) ||
      (partial && !egnn_aligned16(partial)) || (addend && (!egnn_aligned16(addend) || ld_addend % 4 != 0 || ld_addend < K)))
    return EGNN_EALIGN;
  const uint64_t xb = (uint64_t)n_src * (uint64_t)ldx * 4ull;
  if (xb > 0x7FFFFFFFull) return EGNN_EALIGN;  // 32-bit descriptor offsets: the host uses the 64-bit kernels of spmm.hip
  if (blk_ptr == nullptr) n_blk = (n_rows + rows_per_blk - 1) / rows_per_blk;
  EGNN_CHECK_ARG(n_blk > 0);
  const bool lds = win != nullptr;
  EGNN_CHECK_ARG(!lds || (n_src == n_rows && rows_per_blk <= 512 && rows_per_blk % 128 == 0));
  BlkArgs a{n_rows, K, rowptr, col, val, src_scale, bias, X, ldx, Y, ldy, reduce == EGNN_MEAN, seg_max, rows_per_blk, blk_ptr, n_blk,
            win, hub_seg, n_hub_seg, partial, addend, ld_addend, stat_part, stat_shift, 0, 0, (uint32_t)xb, flags};
  a.NS = (int)((K + 31) / 32);
  a.map_mode = (a.NS <= 8 && 8 % a.NS == 0) ? 1 : (a.NS % 8 == 0 ? 2 : 0);
  const int nsub = lds ? 128 : 32;
  const int64_t n_items = (n_hub_seg + nsub - 1) / nsub + n_blk;
  int64_t grid;
  if (a.map_mode == 1) {
    const int r = 8 / a.NS;
    grid = (n_items + r - 1) / r * 8;
  } else {
    grid = n_items * a.NS;
  }
  if (grid > 0x7fffffffLL) return EGNN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool stats = stat_part != nullptr;
  if (!lds) {   // ROWS = 1: the one-row-per-sub-group form (rows_per_blk == 32, the host default); ROWS = 0: the row loop
    if (rows_per_blk == 32)
      return stats ? launch_blk<false, true, 4, 1>(a, (unsigned)grid, 0, st) : launch_blk<false, false, 4, 1>(a, (unsigned)grid, 0, st);
    return stats ? launch_blk<false, true, 4, 0>(a, (unsigned)grid, 0, st) : launch_blk<false, false, 4, 0>(a, (unsigned)grid, 0, st);
  }
  const size_t shm = (size_t)(rows_per_blk + 1) * 128;
  return stats ? launch_blk<true, true, 16, 4>(a, (unsigned)grid, shm, st) : launch_blk<true, false, 16, 4>(a, (unsigned)grid, shm, st);
}

extern "C" size_t egnn_bn_stats_merge_ws_floats(int64_t C) { return (size_t)kStatSplits * 2 * (size_t)C; }

extern "C" int egnn_bn_stats_merge_f32(const float* part, int64_t n_blk, int64_t C, const float* Y, int64_t ldy,
                                       const int64_t* extra_rows, int64_t n_extra, const float* shift, int64_t n_total, float* mean,
                                       float* var, float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(n_blk >= 0 && C > 0 && n_extra >= 0 && n_total > 0 && mean && var);
  EGNN_CHECK_ARG((n_blk == 0 || part) && (n_extra == 0 || (Y && extra_rows && ldy >= C)));
  hipStream_t st = (hipStream_t)stream;
  int64_t n_part = n_blk;
  if (n_blk > 2 * kStatSplits) {   // many blocks: fold them into kStatSplits partial rows first (two short launches)
    if (ws == nullptr || ws_floats < egnn_bn_stats_merge_ws_floats(C)) return EGNN_EWORKSPACE;
    hipLaunchKernelGGL(bn_stats_fold_kernel, dim3((unsigned)((C + 31) / 32), kStatSplits), dim3(256), 0, st, part, n_blk, C, ws);
    part = ws;
    n_part = kStatSplits;
  }
  hipLaunchKernelGGL(bn_stats_merge_kernel, dim3((unsigned)((C + 7) / 8)), dim3(256), 0, st, part, n_part, C, Y, ldy, extra_rows,
                     n_extra, shift, n_total, mean, var);
  return egnn_launch_status();
}
