// Row-block neighbour aggregation for gfx950 (egnn_spmm_csr_blk_f32): Y[i,:] = (sum_e val[e] * X[col[e],:]) * inv_i + bias
//
// Second-generation schedule of the aggregation behind torch_sparse::spmm (/root/reference/arxiv_pyg/gnn.py:47,52,79,84,
// /root/reference/mag_pyg/gnn.py:162), HBM/L2-bound, no MFMA.  What differs from the per-range kernel of spmm.hip:
//   * a WORKGROUP owns a block of consecutive rows for one 128-byte column slice (slice <-> XCD binding as before): the
//     32 8-lane sub-groups of its four waves walk the block's rows round-robin, so one launch of the kernel body serves
//     R/32 rows per sub-group and the rowptr -> col -> X latency chain of one row overlaps the gathers of the others;
//   * 32-bit everything on the gather path: int32 indices, byte offsets premultiplied once per entry, X addressed
//     through a buffer descriptor (buffer_load_dwordx4 with a 32-bit voffset).  Padded lanes point their offset OUT OF
//     RANGE of the descriptor: the hardware returns zeros without touching memory, which replaces the per-element
//     selects of the first kernel (0 * 0 added to the accumulator is exact);
//   * index / value streams can be loaded non-temporally and Y stored write-through (`flags`): neither is re-read by
//     this kernel, so they need not compete with the gathered X lines for the 4 MiB L2 of the XCD;
//   * optional LDS staging of the block's OWN source rows (`win` != NULL: the diagonal block of a locality-ordered
//     graph): the rows' in-block entries -- a contiguous sub-range of every row, columns being sorted -- read X from LDS
//     (ds_read_b128, 4x the L1/L2 gather rate), the remaining entries gather from L2 / fabric as before;
//   * optional BatchNorm statistics in the store epilogue (`stat_part` != NULL, /root/reference/arxiv_pyg/gnn.py:47-48):
//     per-workgroup shifted column sums of the rows it wrote, merged in a fixed order by egnn_bn_stats_merge_f32.
// Rows longer than `seg_max` entries (the hubs) are skipped here: the segment schedule (egnn_spmm_csr_seg_f32 on the
// hub ranges only) writes them.  Accumulation order is fixed => run-to-run bit-stable.
#include "common.h"

namespace {

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
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
  float* stat_part;        // [n_blk][2][K]
  const float* stat_shift; // [K] nullable
  int NS, map_mode;
  uint32_t x_bytes;
  int flags;
};

constexpr int kFlagNtIndex = 1;     // index / value streams: non-temporal loads
constexpr int kFlagStoreNt = 2;     // Y rows: non-temporal stores
constexpr int kFlagStoreSc1 = 4;    // Y rows: write-through (sc1) stores, line dropped from L2
constexpr int kFlagPipe = 8;        // software-pipelined row loop: next row's rowptr + first index chunk fetched under the current gathers
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
  x = buf_load4(rsrc, J < n ? o + lane_off : kOob);
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

template <bool NT>
__device__ __forceinline__ int ld_idx(const int32_t* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ float ld_val(const float* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_y(float* p, const float4& v, int flags) {
  const v4f w = {v.x, v.y, v.z, v.w};
  if (flags & kFlagStoreSc1) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");   // s_nop: store-data hazard the assembler does not see
  } else if (flags & kFlagStoreNt) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}

template <bool LDS, bool NT, bool STATS, int OCC, bool PIPE>
__global__ __launch_bounds__(256, OCC) void spmm_blk_kernel(const BlkArgs a) {
  extern __shared__ float4 sX[];  // LDS variant: rows_per_blk source rows (this slice) + one all-zero row
  __shared__ float s_stat[4][2][32];
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int li = lane & 7;
  const int sub32 = wave * 8 + (lane >> 3);
  const unsigned b = blockIdx.x;
  int slice, blk;
  if (a.map_mode == 1) {  // NS in {1,2,4,8}: XCD x = b % 8 owns slice x % NS
    const int x = (int)(b & 7);
    const int r = 8 / a.NS;
    slice = x % a.NS;
    blk = (int)(b >> 3) * r + x / a.NS;
  } else if (a.map_mode == 2) {  // NS multiple of 8: XCD x owns slices x, x+8, ...
    const int x = (int)(b & 7);
    const int q = (int)(b >> 3);
    const int per = a.NS >> 3;
    slice = x + 8 * (q % per);
    blk = q / per;
  } else {
    slice = (int)(b % (unsigned)a.NS);
    blk = (int)(b / (unsigned)a.NS);
  }
  if (blk >= (int)a.n_blk) return;
  const int row0 = a.blk_ptr ? a.blk_ptr[blk] : blk * a.rows_per_blk;
  int row1 = a.blk_ptr ? a.blk_ptr[blk + 1] : row0 + a.rows_per_blk;
  if (row1 > (int)a.n_rows) row1 = (int)a.n_rows;
  const int nrows = row1 - row0;
  const int col0 = slice * 32 + li * 4;
  const uint32_t lane_off = (uint32_t)li * 16u;
  const uint32_t row_bytes = (uint32_t)a.ldx * 4u;
  // descriptor of this slice's columns of X: base = X + slice*32 floats, offsets = source row * row pitch (+ lane*16)
  const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X + slice * 32), 0,
                                                        (int)(a.x_bytes - (uint32_t)slice * 128u), 0x00020000);
  if constexpr (LDS) {
    for (int i = threadIdx.x; i < nrows * 8; i += 256) {
      const int r = i >> 3, l = i & 7;
      sX[i] = *reinterpret_cast<const float4*>(a.X + (int64_t)(row0 + r) * a.ldx + slice * 32 + l * 4);
    }
    if (threadIdx.x < 8) sX[a.rows_per_blk * 8 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr bool pipe = PIPE && !LDS;
  int pstart = 0, pend = 0, pc = 0;
  float pv = 0.f;
  if constexpr (pipe) {   // prologue: bounds and first index chunk of this sub-group's first row
    if (sub32 < nrows) {
      pstart = a.rowptr[row0 + sub32];
      pend = a.rowptr[row0 + sub32 + 1];
    }
    if (li < pend - pstart) {
      pc = ld_idx<NT>(a.col + pstart + li);
      pv = a.val ? ld_val<NT>(a.val + pstart + li) : 1.f;
    }
  }

  for (int rb = 0; rb < nrows; rb += 32) {
    const int r = rb + sub32;
    const bool inblk = r < nrows;
    const int row = row0 + r;
    int start = 0, end = 0;
    if constexpr (pipe) {
      start = pstart;
      end = pend;
    } else if (inblk) {
      start = a.rowptr[row];
      end = a.rowptr[row + 1];
    }
    const int cnt = end - start;
    const bool live = inblk && cnt <= a.seg_max;  // hub rows: the segment path writes them
    if (!live) end = start;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LDS) {
      int wa = start, wb = start;
      if (live) {
        wa = a.win[2 * row];
        wb = a.win[2 * row + 1];
      }
      // in-block entries [wa, wb): sources staged in LDS
      for (int e = wa; e < wb; e += 8) {
        const int n = wb - e < 8 ? wb - e : 8;
        uint32_t idx_l = (uint32_t)a.rows_per_blk;  // zero row
        float v_l = 0.f;
        if (li < n) {
          const int c = ld_idx<NT>(a.col + e + li);
          idx_l = (uint32_t)(c - row0);
          v_l = a.val ? ld_val<NT>(a.val + e + li) : 1.f;
          if (a.src_scale) v_l *= a.src_scale[c];
        }
        lds_chunk(sX, idx_l, v_l, li, acc);
      }
      // the rest: [start, wa) then [wb, end), gathered from L2 / fabric
      const int nlo = wa - start;
      const int nout = nlo + (end - wb);
      for (int p = 0; p < nout; p += 8) {
        const int n = nout - p < 8 ? nout - p : 8;
        uint32_t off_l = 0;
        float v_l = 0.f;
        if (li < n) {
          const int q = p + li;
          const int e = q < nlo ? start + q : wb + (q - nlo);
          const int c = ld_idx<NT>(a.col + e);
          off_l = (uint32_t)c * row_bytes;
          v_l = a.val ? ld_val<NT>(a.val + e) : 1.f;
          if (a.src_scale) v_l *= a.src_scale[c];
        }
        gather_chunk(rsrc, off_l, v_l, n, lane_off, acc);
      }
    } else if constexpr (pipe) {
      // chunk 0 of this row was fetched during the previous row (pc / pv); fetch the next row's bounds now
      const int rn = r + 32;
      int nstart = 0, nend = 0;
      if (rn < nrows) {
        nstart = a.rowptr[row + 32];
        nend = a.rowptr[row + 33];
      }
      int cc = pc;
      float cv = pv;
      for (int e = start; e < end; e += 8) {
        const int n = end - e < 8 ? end - e : 8;
        const int n2 = end - e - 8;   // entries of the following chunk of this row
        int c2 = 0;
        float v2 = 0.f;
        if (li < n2) {
          c2 = ld_idx<NT>(a.col + e + 8 + li);
          v2 = a.val ? ld_val<NT>(a.val + e + 8 + li) : 1.f;
        }
        uint32_t off_l = 0;
        float v_l = 0.f;
        if (li < n) {
          off_l = (uint32_t)cc * row_bytes;
          v_l = cv;
          if (a.src_scale) v_l *= a.src_scale[cc];
        }
        gather_chunk(rsrc, off_l, v_l, n, lane_off, acc);
        cc = c2;
        cv = v2;
      }
      // first chunk of the next row (hub rows are skipped there, but their first chunk is harmless to fetch)
      pc = 0;
      pv = 0.f;
      if (li < nend - nstart) {
        pc = ld_idx<NT>(a.col + nstart + li);
        pv = a.val ? ld_val<NT>(a.val + nstart + li) : 1.f;
      }
      pstart = nstart;
      pend = nend;
    } else {
      for (int e = start; e < end; e += 8) {
        const int n = end - e < 8 ? end - e : 8;
        uint32_t off_l = 0;
        float v_l = 0.f;
        if (li < n) {
          const int c = ld_idx<NT>(a.col + e + li);
          off_l = (uint32_t)c * row_bytes;
          v_l = a.val ? ld_val<NT>(a.val + e + li) : 1.f;
          if (a.src_scale) v_l *= a.src_scale[c];
        }
        gather_chunk(rsrc, off_l, v_l, n, lane_off, acc);
      }
    }
    if (live) {
      const float inv = a.mean ? 1.f / (float)(cnt > 0 ? cnt : 1) : 1.f;
      float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + col0);   // L1-resident; not held across the gathers
      const float4 y = make_float4(acc[0] * inv + bias.x, acc[1] * inv + bias.y, acc[2] * inv + bias.z, acc[3] * inv + bias.w);
      store_y(a.Y + (int64_t)row * a.ldy + col0, y, a.flags);
      if constexpr (STATS) {
        float4 shift = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.stat_shift) shift = *reinterpret_cast<const float4*>(a.stat_shift + col0);
        const float d0 = y.x - shift.x, d1 = y.y - shift.y, d2 = y.z - shift.z, d3 = y.w - shift.w;
        s1[0] += d0; s1[1] += d1; s1[2] += d2; s1[3] += d3;
        s2[0] = fmaf(d0, d0, s2[0]); s2[1] = fmaf(d1, d1, s2[1]); s2[2] = fmaf(d2, d2, s2[2]); s2[3] = fmaf(d3, d3, s2[3]);
      }
    }
  }
  if constexpr (STATS) {
    // lanes with the same li hold the same 4 columns: add the 8 sub-groups of the wave (xor 8, 16, 32), then the 4 waves
    // in wave order through LDS -- a fixed order
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        s1[q] += __shfl_xor(s1[q], o);
        s2[q] += __shfl_xor(s2[q], o);
      }
    }
    if (lane < 8) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s_stat[wave][0][lane * 4 + q] = s1[q];
        s_stat[wave][1][lane * 4 + q] = s2[q];
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
      const float t = ((s_stat[0][which][c] + s_stat[1][which][c]) + s_stat[2][which][c]) + s_stat[3][which][c];
      a.stat_part[((int64_t)blk * 2 + which) * a.K + slice * 32 + c] = t;
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

// mean / biased variance of the rows of Y from the per-block shifted sums of spmm_blk_kernel plus the rows the block
// kernel skipped (hub rows, read back from Y): fixed order => deterministic.  One thread per column, 8 columns per block
// x 32 partial groups like fused_bn's merge.
__global__ __launch_bounds__(256) void bn_stats_merge_kernel(const float* __restrict__ part, int64_t n_blk, int64_t C,
                                                             const float* __restrict__ Y, int64_t ldy, const int64_t* __restrict__ extra_rows,
                                                             int64_t n_extra, const float* __restrict__ shift, int64_t n_total,
                                                             float* __restrict__ mean, float* __restrict__ var) {
  __shared__ float sh[2][32][8];
  const int colo = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int64_t c = (int64_t)blockIdx.x * 8 + colo;
  float sa = 0.f, sb = 0.f;
  if (c < C) {
    const float sft = shift ? shift[c] : 0.f;
    for (int64_t i = grp; i < n_blk; i += 32) {
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

}  // namespace

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
                                     int64_t n_blk, const int32_t* win, float* stat_part, const float* stat_shift, int flags,
                                     void* stream) {
  EGNN_CHECK_ARG(n_rows >= 0 && n_src >= 0 && K >= 0 && ldx >= K && ldy >= K);
  EGNN_CHECK_ARG(reduce == EGNN_SUM || reduce == EGNN_MEAN);
  EGNN_CHECK_ARG(rows_per_blk > 0 && rows_per_blk % 32 == 0 && seg_max > 0);
  if (n_rows == 0 || K == 0) return EGNN_OK;
  EGNN_CHECK_ARG(rowptr && col && X && Y);
  if (K % 32 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || !egnn_aligned16(X) || !egnn_aligned16(Y) || (bias && !egnn_aligned16(bias)) ||
      (stat_shift && !egnn_aligned16(stat_shift)))
    return EGNN_EALIGN;
  const uint64_t xb = (uint64_t)n_src * (uint64_t)ldx * 4ull;
  if (xb > 0x7FFFFFFFull) return EGNN_EALIGN;  // 32-bit descriptor offsets: the host uses the 64-bit kernels of spmm.hip
  if (blk_ptr == nullptr) n_blk = (n_rows + rows_per_blk - 1) / rows_per_blk;
  EGNN_CHECK_ARG(n_blk > 0);
  const bool lds = win != nullptr;
  EGNN_CHECK_ARG(!lds || (n_src == n_rows && rows_per_blk <= 1024));
  BlkArgs a{n_rows, K, rowptr, col, val, src_scale, bias, X, ldx, Y, ldy, reduce == EGNN_MEAN, seg_max, rows_per_blk, blk_ptr, n_blk,
            win, stat_part, stat_shift, 0, 0, (uint32_t)xb, flags};
  a.NS = (int)(K / 32);
  a.map_mode = (a.NS <= 8 && 8 % a.NS == 0) ? 1 : (a.NS % 8 == 0 ? 2 : 0);
  int64_t grid;
  if (a.map_mode == 1) {
    const int r = 8 / a.NS;
    grid = (n_blk + r - 1) / r * 8;
  } else {
    grid = n_blk * a.NS;
  }
  if (grid > 0x7fffffffLL) return EGNN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool nt = (flags & kFlagNtIndex) != 0;
  const bool stats = stat_part != nullptr;
  const int occ = (flags >> 8) & 0xF;   // experiment knob: waves per SIMD the register allocator is held to (0 = default 8)
  const bool pipe = !lds && (flags & kFlagPipe) != 0;
  const size_t shm = lds ? (size_t)(rows_per_blk + 1) * 128 : 0;
  const dim3 g((unsigned)grid), t(256);
#define EGNN_BLK_LAUNCH(L, N, S, O, P)                                                                                  \
  do {                                                                                                                  \
    if (shm > 65536 &&                                                                                                  \
        hipFuncSetAttribute((const void*)spmm_blk_kernel<L, N, S, O, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) \
      return EGNN_ELAUNCH;                                                                                              \
    hipLaunchKernelGGL((spmm_blk_kernel<L, N, S, O, P>), g, t, shm, st, a);                                             \
  } while (0)
#define EGNN_BLK_PIPE(L, N, S, O)                     \
  do {                                                \
    if (pipe) EGNN_BLK_LAUNCH(L, N, S, O, true);      \
    else EGNN_BLK_LAUNCH(L, N, S, O, false);          \
  } while (0)
#define EGNN_BLK_OCC(L, N, S)                    \
  do {                                           \
    if (occ == 6) EGNN_BLK_PIPE(L, N, S, 6);     \
    else if (occ == 4) EGNN_BLK_PIPE(L, N, S, 4); \
    else EGNN_BLK_PIPE(L, N, S, 8);              \
  } while (0)
#define EGNN_BLK_STATS(L, N)                \
  do {                                      \
    if (stats) EGNN_BLK_OCC(L, N, true);    \
    else EGNN_BLK_OCC(L, N, false);         \
  } while (0)
  if (lds) {
    if (nt) EGNN_BLK_STATS(true, true);
    else EGNN_BLK_STATS(true, false);
  } else {
    if (nt) EGNN_BLK_STATS(false, true);
    else EGNN_BLK_STATS(false, false);
  }
#undef EGNN_BLK_STATS
#undef EGNN_BLK_OCC
#undef EGNN_BLK_PIPE
#undef EGNN_BLK_LAUNCH
  return egnn_launch_status();
}

extern "C" int egnn_bn_stats_merge_f32(const float* part, int64_t n_blk, int64_t C, const float* Y, int64_t ldy,
                                       const int64_t* extra_rows, int64_t n_extra, const float* shift, int64_t n_total, float* mean,
                                       float* var, void* stream) {
  EGNN_CHECK_ARG(n_blk >= 0 && C > 0 && n_extra >= 0 && n_total > 0 && mean && var);
  EGNN_CHECK_ARG((n_blk == 0 || part) && (n_extra == 0 || (Y && extra_rows && ldy >= C)));
  hipLaunchKernelGGL(bn_stats_merge_kernel, dim3((unsigned)((C + 7) / 8)), dim3(256), 0, (hipStream_t)stream, part, n_blk, C, Y, ldy,
                     extra_rows, n_extra, shift, n_total, mean, var);
  return egnn_launch_status();
}
