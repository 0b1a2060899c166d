// CSR neighbour aggregation for gfx950:  Y[i,:] = REDUCE_e val[e] * src_scale[col[e]] * X[col[e],:]
//
// Replaces torch_sparse::spmm (forward and, on the transposed CSR, backward) reached from
// GCNConv / SAGEConv (/root/reference/arxiv_pyg/gnn.py:47,52,79,84,192) and
// adj_t.matmul(reduce='mean') (/root/reference/mag_pyg/gnn.py:162).
//
// Design (HBM/L2-bound gather; no MFMA):
//   * the feature dimension is cut into NS column slices of G lanes x VEC floats (128 B for the
//     K%32==0 cases) and slice s is pinned to XCD (blockIdx % 8): every XCD's private 4 MiB L2 then
//     caches ONE 128-byte line per source node instead of the whole K*4-byte row, i.e. 8x more
//     distinct neighbours stay L2-resident (observed dispatch: block b -> XCD b % 8; speed only);
//   * one wave64 per (row, slice): the wave reads up to 64 (col,val) pairs with ONE coalesced load,
//     then broadcasts them lane-to-lane (ds_bpermute, no LDS storage) while 64/G neighbours are
//     gathered per load instruction and 4 such loads are kept in flight;
//   * rows longer than `long_threshold` are reduced by a whole 1024-thread workgroup whose 16 waves
//     each own a contiguous chunk and are combined through LDS in a fixed order (bit-stable);
//   * fp32 accumulation order is fixed (lane-strided, then an xor tree), no atomics.
#include "common.h"

namespace {

template <typename IdxT>
struct SpmmArgs {
  int64_t n_rows, K;
  const IdxT* rowptr;
  const IdxT* col;
  const float* val;
  const float* src_scale;
  const float* bias;  // optional [K], added to every written row (GCNConv's `out += bias`)
  const float* X;
  int64_t ldx;
  float* Y;
  int64_t ldy;
  int mean;
  int64_t* argmax;
  const int64_t* rows;  // row list walked by the launch (nullptr: rows 0..n_list-1)
  int64_t n_list;
  const int64_t* seg;   // segment schedule (short-row kernel only): [n_list,3] = (first entry, end entry, destination)
  float* P;             // partial sums of the rows that were cut into several segments: [slots, K], ld = K
  const float* addend;      // combine kernel only: nullable [n_rows][ld_add], added to every combined row
  int64_t ld_add;
  float* stat_part;         // combine kernel only: [*, 2, K] rows stat_base + i receive (y - shift), (y - shift)^2 of combined row i
  const float* stat_shift;  // [K] nullable
  int64_t stat_base;
  int relu;                 // combine kernel only: max(y, 0) on the way out (eval-mode BatchNorm folded into the weights + ReLU)
  int logG;      // lanes per neighbour = 1 << logG
  int NS;        // number of column slices
  int map_mode;  // 0: slice = b % NS ; 1: NS divides 8 ; 2: NS multiple of 8
};

constexpr int kUnroll = 4;

// Accumulate entries [start, end) of one row into acc (and arg for max).  All 64 lanes take part in
// the broadcasts; lanes whose column is out of range (colok == false) only skip the loads.
template <typename IdxT, int VEC, bool IS_MAX>
__device__ __forceinline__ void accumulate_range(const SpmmArgs<IdxT>& a, int64_t start, int64_t end, int64_t col0,
                                                 bool colok, int lane, int sub, float (&acc)[VEC],
                                                 int64_t (&arg)[VEC]) {
  const int npw = 64 >> a.logG;
  for (int64_t base = start; base < end; base += 64) {
    const int64_t rem = end - base;
    const int n = rem < 64 ? (int)rem : 64;
    long long c_l = 0;
    float v_l = 1.f;
    if (lane < n) {
      c_l = (long long)a.col[base + lane];
      if (a.val) v_l = a.val[base + lane];
      if (a.src_scale) v_l *= a.src_scale[c_l];
    }
    for (int j0 = 0; j0 < n; j0 += npw * kUnroll) {
      float xv[kUnroll][VEC];
      float vv[kUnroll];
      bool ok[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        ok[u] = false;
        vv[u] = 0.f;
#pragma unroll
        for (int q = 0; q < VEC; ++q) xv[u][q] = 0.f;
        if (j0 + u * npw < n) {  // wave-uniform
          const int j = j0 + u * npw + sub;
          ok[u] = j < n;
          const int jj = ok[u] ? j : n - 1;  // clamp: same cache line as a live lane, never out of bounds
          const long long c = __shfl(c_l, jj);
          vv[u] = __shfl(v_l, jj);
          if (colok) {
            const float* xp = a.X + c * a.ldx + col0;
            if constexpr (VEC == 4) {
              const float4 t = *reinterpret_cast<const float4*>(xp);
              xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][3] = t.w;
            } else {
              xv[u][0] = *xp;
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        if (j0 + u * npw < n) {
          if constexpr (IS_MAX) {
            const int64_t e = base + j0 + u * npw + sub;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
              const float cand = vv[u] * xv[u][q];
              // entries arrive in increasing e per lane, so strict > keeps the first maximal entry
              const bool take = ok[u] && (arg[q] < 0 || cand > acc[q]);
              acc[q] = take ? cand : acc[q];
              arg[q] = take ? e : arg[q];
            }
          } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = ok[u] ? fmaf(vv[u], xv[u][q], acc[q]) : acc[q];
          }
        }
      }
    }
  }
}

// (value, first-entry) max-merge used by the cross-lane / cross-wave combines
__device__ __forceinline__ void max_merge(float& v, int64_t& e, float ov, int64_t oe) {
  const bool take = (oe >= 0) && (e < 0 || ov > v || (ov == v && oe < e));
  v = take ? ov : v;
  e = take ? oe : e;
}

template <int VEC, bool IS_MAX>
__device__ __forceinline__ void cross_sub_reduce(int logG, float (&acc)[VEC], int64_t (&arg)[VEC]) {
  for (int off = 1 << logG; off < 64; off <<= 1) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      const float ov = __shfl_xor(acc[q], off);
      if constexpr (IS_MAX) {
        const long long oe = __shfl_xor((long long)arg[q], off);
        max_merge(acc[q], arg[q], ov, (int64_t)oe);
      } else {
        acc[q] += ov;
      }
    }
  }
}

template <typename IdxT, int VEC, bool IS_MAX>
__device__ __forceinline__ void store_row(const SpmmArgs<IdxT>& a, int64_t row, int64_t col0, float inv,
                                          const float (&acc)[VEC], const int64_t (&arg)[VEC]) {
  float* yp = a.Y + row * a.ldy + col0;
  if constexpr (IS_MAX) {
    int64_t* ap = a.argmax + row * a.K + col0;
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      yp[q] = arg[q] < 0 ? 0.f : acc[q];
      ap[q] = arg[q];
    }
  } else if constexpr (VEC == 4) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) b = *reinterpret_cast<const float4*>(a.bias + col0);
    *reinterpret_cast<float4*>(yp) = make_float4(acc[0] * inv + b.x, acc[1] * inv + b.y, acc[2] * inv + b.z, acc[3] * inv + b.w);
  } else {
    yp[0] = acc[0] * inv + (a.bias ? a.bias[col0] : 0.f);
  }
}

// ---- wave-per-(row, slice) kernel ---------------------------------------------------------------
template <typename IdxT, int VEC, bool IS_MAX>
__global__ __launch_bounds__(256) void spmm_rows_kernel(const SpmmArgs<IdxT> a) {
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int64_t b = blockIdx.x;
  int64_t slice, rg;
  if (a.map_mode == 1) {  // NS in {1,2,4,8}: XCD x = b % 8 owns slice x % NS
    const int x = (int)(b & 7);
    const int r = 8 / a.NS;
    slice = x % a.NS;
    rg = (b >> 3) * r + x / a.NS;
  } else if (a.map_mode == 2) {  // NS multiple of 8: XCD x owns slices x, x+8, ...
    const int x = (int)(b & 7);
    const int64_t q = b >> 3;
    const int per = a.NS >> 3;
    slice = x + 8 * (q % per);
    rg = q / per;
  } else {
    slice = b % a.NS;
    rg = b / a.NS;
  }
  const int64_t ridx = rg * 4 + wave;
  if (ridx >= a.n_list) return;
  const int64_t row = a.rows ? a.rows[ridx] : ridx;
  const int64_t start = (int64_t)a.rowptr[row];
  const int64_t end = (int64_t)a.rowptr[row + 1];

  const int G = 1 << a.logG;
  const int sub = lane >> a.logG;
  const int li = lane & (G - 1);
  const int64_t col0 = (slice * G + li) * VEC;
  const bool colok = col0 < a.K;

  float acc[VEC];
  int64_t arg[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) { acc[q] = 0.f; arg[q] = -1; }
  accumulate_range<IdxT, VEC, IS_MAX>(a, start, end, col0, colok, lane, sub, acc, arg);
  cross_sub_reduce<VEC, IS_MAX>(a.logG, acc, arg);
  if (sub == 0 && colok) {
    const int64_t cnt = end - start;
    const float inv = a.mean ? 1.f / (float)(cnt > 0 ? cnt : 1) : 1.f;
    store_row<IdxT, VEC, IS_MAX>(a, row, col0, inv, acc, arg);
  }
}

// ---- workgroup-per-(long row, slice) kernel -------------------------------------------------------
constexpr int kLongWaves = 16;

template <typename IdxT, int VEC, bool IS_MAX>
__global__ __launch_bounds__(kLongWaves * 64) void spmm_long_rows_kernel(const SpmmArgs<IdxT> a) {
  __shared__ float s_val[kLongWaves][64 * VEC];
  __shared__ long long s_arg[IS_MAX ? kLongWaves : 1][IS_MAX ? 64 * VEC : 1];
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int64_t b = blockIdx.x;
  const int64_t slice = b % a.NS;
  const int64_t row = a.rows[b / a.NS];
  const int64_t start = (int64_t)a.rowptr[row];
  const int64_t end = (int64_t)a.rowptr[row + 1];
  // contiguous chunk per wave, multiple of 64 entries so index loads stay aligned/coalesced
  int64_t chunk = (end - start + kLongWaves - 1) / kLongWaves;
  chunk = (chunk + 63) / 64 * 64;
  int64_t ws = start + wave * chunk;
  int64_t we = ws + chunk;
  if (ws > end) ws = end;
  if (we > end) we = end;

  const int G = 1 << a.logG;
  const int sub = lane >> a.logG;
  const int li = lane & (G - 1);
  const int64_t col0 = (slice * G + li) * VEC;
  const bool colok = col0 < a.K;

  float acc[VEC];
  int64_t arg[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) { acc[q] = 0.f; arg[q] = -1; }
  accumulate_range<IdxT, VEC, IS_MAX>(a, ws, we, col0, colok, lane, sub, acc, arg);
  cross_sub_reduce<VEC, IS_MAX>(a.logG, acc, arg);
  if (sub == 0) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      s_val[wave][li * VEC + q] = acc[q];
      if constexpr (IS_MAX) s_arg[wave][li * VEC + q] = arg[q];
    }
  }
  __syncthreads();
  if (wave == 0 && sub == 0 && colok) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      float v = s_val[0][li * VEC + q];
      int64_t e = IS_MAX ? (int64_t)s_arg[0][li * VEC + q] : -1;
      for (int w = 1; w < kLongWaves; ++w) {  // fixed order
        if constexpr (IS_MAX) max_merge(v, e, s_val[w][li * VEC + q], (int64_t)s_arg[w][li * VEC + q]);
        else v += s_val[w][li * VEC + q];
      }
      acc[q] = v;
      arg[q] = e;
    }
    const int64_t cnt = end - start;
    const float inv = a.mean ? 1.f / (float)(cnt > 0 ? cnt : 1) : 1.f;
    store_row<IdxT, VEC, IS_MAX>(a, row, col0, inv, acc, arg);
  }
}


// ---- sub-group-per-(short row, slice) kernel ------------------------------------------------------
// The power-law bulk (degree <= short_max): a wave walks 64/G rows at once, one G-lane sub-group per row,
// so the rowptr -> col -> X dependency chain is paid once per 64/G rows and G gathers per row are in flight
// without any cross-lane reduction.  The caller orders `rows` so that a wave's rows have similar lengths.
// Persistent grid: each wave strides over row groups of its slice (slice <-> XCD binding as above).
template <typename IdxT, int LOGG>
__global__ __launch_bounds__(256) void spmm_short_rows_kernel(const SpmmArgs<IdxT> a) {
  constexpr int G = 1 << LOGG;
  constexpr int NPW = 64 >> LOGG;
  constexpr int UN = G < 8 ? G : 8;
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int sub = lane >> LOGG;
  const int li = lane & (G - 1);
  const int lane0 = lane & ~(G - 1);
  const int64_t n_groups = (a.n_list + NPW - 1) / NPW;
  const int64_t b = blockIdx.x, nb = gridDim.x;
  // enumerate this wave's (slice, group) items
  int64_t slice0, slice_step, n_slices_mine, g0, gstep;
  if (a.map_mode == 1) {
    const int x = (int)(b & 7);
    const int r = 8 / a.NS;
    slice0 = x % a.NS; slice_step = 1; n_slices_mine = 1;
    g0 = ((b >> 3) * r + x / a.NS) * 4 + wave;
    gstep = (nb >> 3) * r * 4;
  } else if (a.map_mode == 2) {
    const int x = (int)(b & 7);
    slice0 = x; slice_step = 8; n_slices_mine = a.NS >> 3;
    g0 = (b >> 3) * 4 + wave;
    gstep = (nb >> 3) * 4;
  } else {
    slice0 = 0; slice_step = 1; n_slices_mine = a.NS;
    g0 = b * 4 + wave;
    gstep = nb * 4;
  }
  for (int64_t si = 0; si < n_slices_mine; ++si) {
    const int64_t slice = slice0 + si * slice_step;
    const int64_t col0 = (slice * G + li) * 4;
    const bool colok = col0 < a.K;
    for (int64_t grp = g0; grp < n_groups; grp += gstep) {
      const int64_t sg = grp * NPW + sub;
      const bool live = sg < a.n_list;
      int64_t row = 0, start = 0, end = 0;
      if (live) {
        if (a.seg) {  // explicit entry range; destination < n_rows: that row of Y, else a slot of the partial buffer
          start = a.seg[3 * sg];
          end = a.seg[3 * sg + 1];
          row = a.seg[3 * sg + 2];
        } else {
          row = a.rows ? a.rows[sg] : sg;
          start = (int64_t)a.rowptr[row];
          end = (int64_t)a.rowptr[row + 1];
        }
      }
      float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
      for (int64_t e = start; e < end; e += G) {
        const int64_t rem = end - e;
        const int n = rem < G ? (int)rem : G;
        const int64_t ee = e + (li < n ? li : 0);  // lanes past the row end re-read its first entry (weight 0)
        const long long c_l = (long long)a.col[ee];
        float v_l = 0.f;
        if (li < n) {
          v_l = a.val ? a.val[ee] : 1.f;
          if (a.src_scale) v_l *= a.src_scale[c_l];
        }
#pragma unroll
        for (int j0 = 0; j0 < G; j0 += UN) {
          float4 x[UN];
          float v[UN];
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const long long c = __shfl(c_l, lane0 + j0 + u);
            v[u] = __shfl(v_l, lane0 + j0 + u);
            x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (colok) x[u] = *reinterpret_cast<const float4*>(a.X + c * a.ldx + col0);
          }
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const bool ok = j0 + u < n;  // select, not multiply: a padded lane must not turn inf/nan into nan
            acc0 = ok ? fmaf(v[u], x[u].x, acc0) : acc0;
            acc1 = ok ? fmaf(v[u], x[u].y, acc1) : acc1;
            acc2 = ok ? fmaf(v[u], x[u].z, acc2) : acc2;
            acc3 = ok ? fmaf(v[u], x[u].w, acc3) : acc3;
          }
        }
      }
      if (live && colok) {
        if (row >= a.n_rows) {  // one segment of a long row: raw partial sum, finished by spmm_combine_kernel
          *reinterpret_cast<float4*>(a.P + (row - a.n_rows) * a.K + col0) = make_float4(acc0, acc1, acc2, acc3);
        } else {
          const int64_t cnt = end - start;
          const float inv = a.mean ? 1.f / (float)(cnt > 0 ? cnt : 1) : 1.f;
          float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a.bias) b = *reinterpret_cast<const float4*>(a.bias + col0);
          *reinterpret_cast<float4*>(a.Y + row * a.ldy + col0) =
              make_float4(acc0 * inv + b.x, acc1 * inv + b.y, acc2 * inv + b.z, acc3 * inv + b.w);
        }
      }
    }
  }
}

// Y[row] = (sum of the row's segment partials) * inv + bias, for the rows cut into several segments.  One workgroup per
// row: 256 / kvp slot lanes per float4 column (kvp = columns rounded up to a power of two) each add every (256/kvp)-th
// slot with four loads in flight, then the lanes of a column are added in lane order -- a fixed order, so bit-stable;
// a hub of 12 505 entries has 196 slots, which one thread per column would walk as 196 dependent-latency steps.
template <typename IdxT>
__global__ __launch_bounds__(256) void spmm_combine_kernel(const SpmmArgs<IdxT> a, const int64_t* __restrict__ crow,
                                                           const int64_t* __restrict__ cptr, int kvp_log) {
  __shared__ float4 red[256];
  const int64_t kv = a.K >> 2;
  const int kvp = 1 << kvp_log;
  const int nsl = 256 >> kvp_log;
  const int t = threadIdx.x;
  const int sl = t >> kvp_log;
  const int64_t row = crow[blockIdx.x];
  const int64_t p0 = cptr[blockIdx.x], p1 = cptr[blockIdx.x + 1];
  for (int64_t cb = 0; cb < kv; cb += kvp) {   // one pass unless K > 1024
    const int64_t cv = cb + (t & (kvp - 1));
    const bool ok = cv < kv;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      const float* base = a.P + cv * 4;
      int64_t p = p0 + sl;
      for (; p + 3 * nsl < p1; p += 4 * nsl) {
        const float4 v0 = *reinterpret_cast<const float4*>(base + p * a.K);
        const float4 v1 = *reinterpret_cast<const float4*>(base + (p + nsl) * a.K);
        const float4 v2 = *reinterpret_cast<const float4*>(base + (p + 2 * nsl) * a.K);
        const float4 v3 = *reinterpret_cast<const float4*>(base + (p + 3 * nsl) * a.K);
        s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
        s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
        s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
        s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
      }
      for (; p < p1; p += nsl) {
        const float4 v = *reinterpret_cast<const float4*>(base + p * a.K);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    red[t] = s;
    __syncthreads();
    if (sl == 0 && ok) {
      for (int l = 1; l < nsl; ++l) {
        const float4 v = red[(l << kvp_log) + t];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int64_t cnt = (int64_t)a.rowptr[row + 1] - (int64_t)a.rowptr[row];
      const float inv = a.mean ? 1.f / (float)(cnt > 0 ? cnt : 1) : 1.f;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias) b = *reinterpret_cast<const float4*>(a.bias + cv * 4);
      float4 y = make_float4(s.x * inv + b.x, s.y * inv + b.y, s.z * inv + b.z, s.w * inv + b.w);
      if (a.addend) { const float4 ad = *reinterpret_cast<const float4*>(a.addend + row * a.ld_add + cv * 4); y.x += ad.x; y.y += ad.y; y.z += ad.z; y.w += ad.w; }
      if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
      *reinterpret_cast<float4*>(a.Y + row * a.ldy + cv * 4) = y;
      if (a.stat_part) {   // BatchNorm statistics of the aggregation epilogue: a hub row is one partial row of its own
        float4 sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.stat_shift) sh = *reinterpret_cast<const float4*>(a.stat_shift + cv * 4);
        const float4 d = make_float4(y.x - sh.x, y.y - sh.y, y.z - sh.z, y.w - sh.w);
        float* sp = a.stat_part + ((a.stat_base + blockIdx.x) * 2) * a.K + cv * 4;
        *reinterpret_cast<float4*>(sp) = d;
        *reinterpret_cast<float4*>(sp + a.K) = make_float4(d.x * d.x, d.y * d.y, d.z * d.z, d.w * d.w);
      }
    }
    __syncthreads();
  }
}

int ilog2_ceil(int64_t v) {
  int l = 0;
  while ((1LL << l) < v) ++l;
  return l;
}

struct RowPlan {
  const int64_t* short_rows; int64_t n_short;
  const int64_t* mid_rows; int64_t n_mid;
  const int64_t* long_rows; int64_t n_long;
};

template <typename IdxT, int VEC, bool IS_MAX>
void launch_wave_per_row(SpmmArgs<IdxT> a, const int64_t* rows, int64_t n_list, hipStream_t st) {
  if (n_list <= 0) return;
  a.rows = rows;
  a.n_list = n_list;
  const int64_t rgroups = (n_list + 3) / 4;
  int64_t grid;
  if (a.map_mode == 1) {
    const int r = 8 / a.NS;
    grid = ((rgroups + r - 1) / r) * 8;
  } else {
    grid = rgroups * a.NS;
  }
  hipLaunchKernelGGL((spmm_rows_kernel<IdxT, VEC, IS_MAX>), dim3((unsigned)grid), dim3(256), 0, st, a);
}

template <typename IdxT, int VEC, bool IS_MAX>
int launch(SpmmArgs<IdxT> a, const RowPlan& plan, hipStream_t st) {
  // slice geometry: 128-byte slices (G = 8 float4 lanes) whenever K allows, else one slice per <=64 lanes
  const int64_t kv = (a.K + VEC - 1) / VEC;  // columns in VEC units
  if (VEC == 4 && kv % 8 == 0) {
    a.logG = 3;
    a.NS = (int)(kv / 8);
  } else {
    a.logG = ilog2_ceil(kv < 64 ? kv : 64);
    a.NS = (int)((kv + (1 << a.logG) - 1) >> a.logG);
  }
  a.map_mode = (a.NS <= 8 && 8 % a.NS == 0) ? 1 : (a.NS % 8 == 0 ? 2 : 0);
  if ((a.n_rows + 3) / 4 * (int64_t)a.NS > 0x7fffffffLL) return EGNN_EINVAL;
  const bool planned = plan.short_rows || plan.mid_rows || plan.long_rows;
  if (!planned) {
    launch_wave_per_row<IdxT, VEC, IS_MAX>(a, nullptr, a.n_rows, st);
    return egnn_launch_status();
  }
  // short rows: sub-group-per-row kernel where it exists (float4 path, G in {8,16}, sum/mean)
  bool short_done = false;
  if constexpr (VEC == 4 && !IS_MAX) {
    if (plan.n_short > 0 && (a.logG == 3 || a.logG == 4)) {
      SpmmArgs<IdxT> s = a;
      s.rows = plan.short_rows;
      s.n_list = plan.n_short;
      const int npw = 64 >> a.logG;
      const int64_t n_groups = (plan.n_short + npw - 1) / npw;
      // waves that share a slice; 8 resident blocks per CU x 32 CUs per XCD is the ceiling worth launching
      const int teams = a.map_mode == 1 ? a.NS : (a.map_mode == 2 ? 8 : 1);
      int64_t blocks = (n_groups + 3) / 4 * teams;
      blocks = (blocks + 7) / 8 * 8;
      // one row group per wave: a static stride over length-sorted chunks would give some waves only the
      // heavy groups; let the hardware dispatcher balance instead (the in-kernel loop then runs once)
      if (blocks > 0x7ffffff8LL) blocks = 0x7ffffff8LL;
      if (a.logG == 3) hipLaunchKernelGGL((spmm_short_rows_kernel<IdxT, 3>), dim3((unsigned)blocks), dim3(256), 0, st, s);
      else hipLaunchKernelGGL((spmm_short_rows_kernel<IdxT, 4>), dim3((unsigned)blocks), dim3(256), 0, st, s);
      short_done = true;
    }
  }
  if (!short_done) launch_wave_per_row<IdxT, VEC, IS_MAX>(a, plan.short_rows, plan.n_short, st);
  launch_wave_per_row<IdxT, VEC, IS_MAX>(a, plan.mid_rows, plan.n_mid, st);
  if (plan.n_long > 0) {
    SpmmArgs<IdxT> l = a;
    l.rows = plan.long_rows;
    l.n_list = plan.n_long;
    const int64_t g2 = plan.n_long * a.NS;
    if (g2 > 0x7fffffffLL) return EGNN_EINVAL;
    hipLaunchKernelGGL((spmm_long_rows_kernel<IdxT, VEC, IS_MAX>), dim3((unsigned)g2), dim3(kLongWaves * 64), 0, st, l);
  }
  return egnn_launch_status();
}

template <typename IdxT>
int dispatch(SpmmArgs<IdxT> a, int reduce, const RowPlan& plan, hipStream_t st) {
  const bool vec4 = (a.K % 4 == 0) && (a.ldx % 4 == 0) && (a.ldy % 4 == 0) && egnn_aligned16(a.X) && egnn_aligned16(a.Y);
  if (reduce == EGNN_MAX) return vec4 ? launch<IdxT, 4, true>(a, plan, st) : launch<IdxT, 1, true>(a, plan, st);
  return vec4 ? launch<IdxT, 4, false>(a, plan, st) : launch<IdxT, 1, false>(a, plan, st);
}

// segment schedule: every row is one or more entry ranges of at most ~64 entries, all of them walked by the
// sub-group-per-row kernel; rows with several ranges are finished by spmm_combine_kernel
template <typename IdxT>
int launch_segments(SpmmArgs<IdxT> a, const int64_t* seg, int64_t n_seg, const int64_t* crow, const int64_t* cptr, int64_t n_comb,
                    float* partial, hipStream_t st) {
  const int64_t kv = a.K / 4;
  if (kv % 8 == 0) {
    a.logG = 3;
    a.NS = (int)(kv / 8);
  } else {
    a.logG = ilog2_ceil(kv < 64 ? kv : 64);
    a.NS = (int)((kv + (1 << a.logG) - 1) >> a.logG);
  }
  if (n_seg > 0 && a.logG != 3 && a.logG != 4) return EGNN_EALIGN;   // the sub-group kernel's forms; the combine step takes any K % 4 == 0
  a.map_mode = (a.NS <= 8 && 8 % a.NS == 0) ? 1 : (a.NS % 8 == 0 ? 2 : 0);
  a.seg = seg;
  a.P = partial;
  a.rows = nullptr;
  a.n_list = n_seg;
  const int npw = 64 >> a.logG;
  const int64_t n_groups = (n_seg + npw - 1) / npw;
  const int teams = a.map_mode == 1 ? a.NS : (a.map_mode == 2 ? 8 : 1);
  int64_t blocks = (n_groups + 3) / 4 * teams;
  blocks = (blocks + 7) / 8 * 8;
  if (blocks > 0x7ffffff8LL) return EGNN_EINVAL;
  if (n_seg > 0) {
    if (a.logG == 3) hipLaunchKernelGGL((spmm_short_rows_kernel<IdxT, 3>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((spmm_short_rows_kernel<IdxT, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  }
  if (n_comb > 0) {
    if (n_comb > 0x7fffffffLL) return EGNN_EINVAL;
    int kvp_log = ilog2_ceil(kv < 256 ? kv : 256);
    hipLaunchKernelGGL((spmm_combine_kernel<IdxT>), dim3((unsigned)n_comb), dim3(256), 0, st, a, crow, cptr, kvp_log);
  }
  return egnn_launch_status();
}

template <typename IdxT>
__global__ void spmm_max_bwd_kernel(int64_t total, int64_t K, const IdxT* col, const float* val, const int64_t* argmax,
                                    const float* dY, int64_t ldy, float* dX, int64_t ldx) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / K, k = t % K;
    const int64_t e = argmax[t];
    if (e < 0) continue;
    float g = dY[i * ldy + k];
    if (val) g *= val[e];
    atomicAdd(dX + (int64_t)col[e] * ldx + k, g);
  }
}

}  // namespace

extern "C" int egnn_spmm_csr_f32(int64_t n_rows, int64_t n_src, int64_t K, const void* rowptr, const void* col,
                                 int index_bits, const float* val, const float* src_scale, const float* bias, const float* X,
                                 int64_t ldx, float* Y, int64_t ldy, int reduce, int64_t* argmax, const int64_t* short_rows,
                                 int64_t n_short, const int64_t* mid_rows, int64_t n_mid, const int64_t* long_rows,
                                 int64_t n_long, void* stream) {
  EGNN_CHECK_ARG(n_rows >= 0 && n_src >= 0 && K >= 0 && ldx >= K && ldy >= K);
  EGNN_CHECK_ARG(index_bits == 32 || index_bits == 64);
  EGNN_CHECK_ARG(reduce == EGNN_SUM || reduce == EGNN_MEAN || reduce == EGNN_MAX);
  if (n_rows == 0 || K == 0) return EGNN_OK;
  EGNN_CHECK_ARG(rowptr && col && X && Y);
  EGNN_CHECK_ARG(reduce != EGNN_MAX || argmax != nullptr);
  EGNN_CHECK_ARG(bias == nullptr || (reduce != EGNN_MAX && (K % 4 != 0 || egnn_aligned16(bias))));
  EGNN_CHECK_ARG(n_short >= 0 && n_mid >= 0 && n_long >= 0);
  EGNN_CHECK_ARG((n_short == 0 || short_rows) && (n_mid == 0 || mid_rows) && (n_long == 0 || long_rows));
  const bool planned = short_rows || mid_rows || long_rows;
  EGNN_CHECK_ARG(!planned || n_short + n_mid + n_long <= n_rows);
  const RowPlan plan{short_rows, n_short, mid_rows, n_mid, long_rows, n_long};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (index_bits == 32) {
    SpmmArgs<int32_t> a{n_rows, K, (const int32_t*)rowptr, (const int32_t*)col, val, src_scale, bias, X, ldx, Y, ldy,
                        reduce == EGNN_MEAN, argmax, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0};
    return dispatch(a, reduce, plan, st);
  }
  SpmmArgs<int64_t> a{n_rows, K, (const int64_t*)rowptr, (const int64_t*)col, val, src_scale, bias, X, ldx, Y, ldy,
                      reduce == EGNN_MEAN, argmax, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0};
  return dispatch(a, reduce, plan, st);
}

extern "C" int egnn_spmm_csr_seg_f32(int64_t n_rows, int64_t n_src, int64_t K, const void* rowptr, const void* col, int index_bits,
                                     const float* val, const float* src_scale, const float* bias, const float* X, int64_t ldx,
                                     float* Y, int64_t ldy, int reduce, const int64_t* seg, int64_t n_seg, const int64_t* comb_rows,
                                     const int64_t* comb_ptr, int64_t n_comb, float* partial, int64_t partial_slots, void* stream) {
  EGNN_CHECK_ARG(n_rows >= 0 && n_src >= 0 && K >= 0 && ldx >= K && ldy >= K);
  EGNN_CHECK_ARG(index_bits == 32 || index_bits == 64);
  EGNN_CHECK_ARG(reduce == EGNN_SUM || reduce == EGNN_MEAN);
  if (n_rows == 0 || K == 0) return EGNN_OK;
  EGNN_CHECK_ARG(rowptr && col && X && Y && n_seg >= 0 && n_comb >= 0 && partial_slots >= 0);
  EGNN_CHECK_ARG((n_seg == 0 || seg) && (n_comb == 0 || (comb_rows && comb_ptr && partial)));
  if (K % 4 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || !egnn_aligned16(X) || !egnn_aligned16(Y) || (bias && !egnn_aligned16(bias)) ||
      (partial && !egnn_aligned16(partial)))
    return EGNN_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (index_bits == 32) {
    SpmmArgs<int32_t> a{n_rows, K, (const int32_t*)rowptr, (const int32_t*)col, val, src_scale, bias, X, ldx, Y, ldy,
                        reduce == EGNN_MEAN, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0};
    return launch_segments(a, seg, n_seg, comb_rows, comb_ptr, n_comb, partial, st);
  }
  SpmmArgs<int64_t> a{n_rows, K, (const int64_t*)rowptr, (const int64_t*)col, val, src_scale, bias, X, ldx, Y, ldy,
                      reduce == EGNN_MEAN, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0};
  return launch_segments(a, seg, n_seg, comb_rows, comb_ptr, n_comb, partial, st);
}

// The combine step on its own (the hub rows of the row-block schedule, egnn_spmm_csr_blk_f32): Y[r] = (sum of the row's
// partial slots, in slot order) * inv + bias; optionally one statistics partial row per combined row.
extern "C" int egnn_spmm_combine_f32(int64_t n_rows, int64_t K, const void* rowptr, int index_bits, const float* bias, float* Y,
                                     int64_t ldy, int reduce, const int64_t* comb_rows, const int64_t* comb_ptr, int64_t n_comb,
                                     const float* partial, const float* addend, int64_t ld_addend, float* stat_part, int64_t stat_base,
                                     const float* stat_shift, int flags, void* stream) {
  EGNN_CHECK_ARG(n_rows >= 0 && K >= 0 && ldy >= K && n_comb >= 0 && stat_base >= 0);
  EGNN_CHECK_ARG(index_bits == 32 || index_bits == 64);
  EGNN_CHECK_ARG(reduce == EGNN_SUM || reduce == EGNN_MEAN);
  if (n_comb == 0 || K == 0) return EGNN_OK;
  EGNN_CHECK_ARG(rowptr && Y && comb_rows && comb_ptr && partial && n_comb <= 0x7fffffffLL);
  if (K % 4 != 0 || ldy % 4 != 0 || !egnn_aligned16(Y) || !egnn_aligned16(partial) || (bias && !egnn_aligned16(bias)) ||
      (stat_part && !egnn_aligned16(stat_part)) || (stat_shift && !egnn_aligned16(stat_shift)) ||
      (addend && (!egnn_aligned16(addend) || ld_addend % 4 != 0 || ld_addend < K)))
    return EGNN_EALIGN;
  const int64_t kv = K / 4;
  const int kvp_log = ilog2_ceil(kv < 256 ? kv : 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (index_bits == 32) {
    SpmmArgs<int32_t> a{n_rows, K, (const int32_t*)rowptr, nullptr, nullptr, nullptr, bias, nullptr, 0, Y, ldy, reduce == EGNN_MEAN, nullptr,
                        nullptr, 0, nullptr, const_cast<float*>(partial), addend, ld_addend, stat_part, stat_shift, stat_base, (flags & 8) ? 1 : 0, 0, 0, 0};
    hipLaunchKernelGGL((spmm_combine_kernel<int32_t>), dim3((unsigned)n_comb), dim3(256), 0, st, a, comb_rows, comb_ptr, kvp_log);
  } else {
    SpmmArgs<int64_t> a{n_rows, K, (const int64_t*)rowptr, nullptr, nullptr, nullptr, bias, nullptr, 0, Y, ldy, reduce == EGNN_MEAN, nullptr,
                        nullptr, 0, nullptr, const_cast<float*>(partial), addend, ld_addend, stat_part, stat_shift, stat_base, (flags & 8) ? 1 : 0, 0, 0, 0};
    hipLaunchKernelGGL((spmm_combine_kernel<int64_t>), dim3((unsigned)n_comb), dim3(256), 0, st, a, comb_rows, comb_ptr, kvp_log);
  }
  return egnn_launch_status();
}

extern "C" int egnn_spmm_csr_max_bwd_f32(int64_t n_rows, int64_t K, const void* col, int index_bits, const float* val,
                                         const int64_t* argmax, const float* dY, int64_t ldy, float* dX, int64_t ldx,
                                         void* stream) {
  EGNN_CHECK_ARG(n_rows >= 0 && K >= 0 && (index_bits == 32 || index_bits == 64));
  if (n_rows == 0 || K == 0) return EGNN_OK;
  EGNN_CHECK_ARG(col && argmax && dY && dX);
  const int64_t total = n_rows * K;
  const int64_t blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (index_bits == 32)
    hipLaunchKernelGGL(spmm_max_bwd_kernel<int32_t>, dim3(grid), dim3(256), 0, st, total, K, (const int32_t*)col, val, argmax, dY, ldy, dX, ldx);
  else
    hipLaunchKernelGGL(spmm_max_bwd_kernel<int64_t>, dim3(grid), dim3(256), 0, st, total, K, (const int64_t*)col, val, argmax, dY, ldy, dX, ldx);
  return egnn_launch_status();
}

extern "C" int64_t egnn_spmm_algorithmic_bytes(int64_t n_rows, int64_t n_src, int64_t K, int64_t nnz, int index_bits,
                                               int has_val) {
  const int64_t ib = index_bits / 8;
  return 4 * n_src * K + 4 * n_rows * K + nnz * (ib + (has_val ? 4 : 0)) + (n_rows + 1) * ib;
}
