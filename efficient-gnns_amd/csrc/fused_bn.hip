// Fused BatchNorm1d (+ ReLU + dropout) over node rows for gfx950 -- SURVEY.md 8(f) rank 1 (K8).
// Replaces the ATen BatchNorm / threshold / fused_dropout kernel chain between every pair of convs
// (/root/reference/arxiv_pyg/gnn.py:48-50,80-82) and inside the projection heads (:296-306):
//   forward : column statistics (1 read of x) + one elementwise pass  y = drop(relu(g * xhat + b))
//   backward: column reductions (sum d, sum d*xhat) + one elementwise pass; xhat, the ReLU sign and the dropout
//             mask are RECOMPUTED from x and the 64-bit seed, so nothing but x is kept for the backward.
// HBM-bound: every wave streams whole rows (64 lanes x float4 = 256 columns) with coalesced 1 KiB accesses;
// column partials are merged in a fixed order (deterministic).
#include "bn_common.h"

namespace {

constexpr int kMaxChunks = 4;     // columns are processed in chunks of 256 (64 lanes x float4): C <= 1024
constexpr int kStatBlocks = 512;   // partial-sum rows (2 wave-sets per CU keep the stream saturated)

using namespace egnn_bn;

// ---- column statistics: shifted sums  s1 = sum (x - x0), s2 = sum (x - x0)^2 ------------------------------
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t n, int64_t C,
                                                               float* __restrict__ part) {
  __shared__ float sh[4][2][kMaxChunks * 256];
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int chunks = (int)((C + 255) / 256);
  float s1[kMaxChunks][4], s2[kMaxChunks][4], x0[kMaxChunks][4];
#pragma unroll
  for (int j = 0; j < kMaxChunks; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1[j][q] = 0.f; s2[j][q] = 0.f; x0[j][q] = 0.f; }
#pragma unroll
  for (int j = 0; j < kMaxChunks; ++j) {
    const int64_t c = j * 256 + lane * 4;
    if (j < chunks && c < C) {
      const float4 v = *reinterpret_cast<const float4*>(x + c);  // row 0 = shift
      x0[j][0] = v.x; x0[j][1] = v.y; x0[j][2] = v.z; x0[j][3] = v.w;
    }
  }
  auto tally = [&](int j, const float4& v) {
    const float d0 = v.x - x0[j][0], d1 = v.y - x0[j][1], d2 = v.z - x0[j][2], d3 = v.w - x0[j][3];
    s1[j][0] += d0; s1[j][1] += d1; s1[j][2] += d2; s1[j][3] += d3;
    s2[j][0] = fmaf(d0, d0, s2[j][0]); s2[j][1] = fmaf(d1, d1, s2[j][1]);
    s2[j][2] = fmaf(d2, d2, s2[j][2]); s2[j][3] = fmaf(d3, d3, s2[j][3]);
  };
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = blockIdx.x * 4LL + wave;
  if (chunks == 1) {   // C <= 256: four rows per iteration, four independent loads in flight per lane (same row order in the sums)
    const int64_t c = lane * 4;
    if (c < C) {
      for (; row + 3 * stride < n; row += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (row + u * stride) * ld + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) tally(0, v[u]);
      }
    }
  }
  for (; row < n; row += stride) {
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
      const int64_t c = j * 256 + lane * 4;
      if (j < chunks && c < C) tally(j, *reinterpret_cast<const float4*>(x + row * ld + c));
    }
  }
#pragma unroll
  for (int j = 0; j < kMaxChunks; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sh[wave][0][j * 256 + lane * 4 + q] = s1[j][q];
      sh[wave][1][j * 256 + lane * 4 + q] = s2[j][q];
    }
  __syncthreads();
  for (int64_t c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < 4; ++w) { a += sh[w][0][c]; b += sh[w][1][c]; }
    part[((int64_t)blockIdx.x * 2) * C + c] = a;
    part[((int64_t)blockIdx.x * 2 + 1) * C + c] = b;
  }
}

// merge the per-block partials: a block owns kMergeCols columns, 256 / kMergeCols thread groups stride over the partial
// rows (16 dependent-latency steps for 512 partials), fixed-order LDS combine.  (A single thread per column walking all
// partials took ~250 us for 1024 partials; 32 columns x 8 groups 22 us; this shape ~8 us.)
constexpr int kMergeCols = 8;
constexpr int kMergeGroups = 256 / kMergeCols;
__device__ __forceinline__ void merge_partials(const float* __restrict__ part, int nblocks, int64_t C, float& a, float& b, bool& owner,
                                               int64_t& c) {
  __shared__ float sh[2][kMergeGroups][kMergeCols];
  const int col = threadIdx.x % kMergeCols, grp = threadIdx.x / kMergeCols;
  c = (int64_t)blockIdx.x * kMergeCols + col;
  float sa = 0.f, sb = 0.f;
  if (c < C)
    for (int i = grp; i < nblocks; i += kMergeGroups) { sa += part[((int64_t)i * 2) * C + c]; sb += part[((int64_t)i * 2 + 1) * C + c]; }
  sh[0][grp][col] = sa;
  sh[1][grp][col] = sb;
  __syncthreads();
  owner = grp == 0 && c < C;
  a = 0.f; b = 0.f;
  if (owner)
    for (int g = 0; g < kMergeGroups; ++g) { a += sh[0][g][col]; b += sh[1][g][col]; }
}

__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ part, int nblocks, const float* __restrict__ x,
                                                             int64_t n, int64_t C, float* __restrict__ mean, float* __restrict__ var) {
  float a, b; bool owner; int64_t c;
  merge_partials(part, nblocks, C, a, b, owner, c);
  if (!owner) return;
  const float inv = 1.f / (float)n;
  const float m1 = a * inv;
  mean[c] = x[c] + m1;
  var[c] = fmaxf(b * inv - m1 * m1, 0.f);
}

// ---- forward apply ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const BnParams q, float* __restrict__ y, int64_t ldy) {
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int chunks = (int)((q.C + 255) / 256);
  for (int j = 0; j < chunks; ++j) {
    const int64_t c = j * 256 + lane * 4;
    if (c >= q.C) continue;
    float mean[4], rstd[4], g[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mean[k] = q.mean[c + k];
      rstd[k] = rsqrtf(q.var[c + k] + q.eps);
      g[k] = q.gamma ? q.gamma[c + k] : 1.f;
      b[k] = q.beta ? q.beta[c + k] : 0.f;
    }
    for (int64_t row = blockIdx.x * 4LL + wave; row < q.n; row += (int64_t)gridDim.x * 4) {
      const int64_t src = bn_row(q, row);
      const float4 v = *reinterpret_cast<const float4*>(q.x + src * q.ldx + c);
      const float xv[4] = {v.x, v.y, v.z, v.w};
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float xhat, gate;
        bn_elem(q, xv[k], mean[k], rstd[k], g[k], b[k], src, c + k, xhat, gate);
        o[k] = (g[k] * xhat + b[k]) * gate;
      }
      *reinterpret_cast<float4*>(y + row * ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// ---- backward: column reductions of d and d * xhat ----------------------------------------------------------
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const BnParams q, const float* __restrict__ dy, int64_t ldd,
                                                                float* __restrict__ part) {
  __shared__ float sh[4][2][kMaxChunks * 256];
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int chunks = (int)((q.C + 255) / 256);
  for (int j = 0; j < chunks; ++j) {
    const int64_t c = j * 256 + lane * 4;
    float sd[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < q.C) {
      float mean[4], rstd[4], g[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mean[k] = q.mean[c + k];
        rstd[k] = rsqrtf(q.var[c + k] + q.eps);
        g[k] = q.gamma ? q.gamma[c + k] : 1.f;
        b[k] = q.beta ? q.beta[c + k] : 0.f;
      }
      // four rows per iteration: eight independent 16-byte loads in flight per lane (one row at a time left the pass at
      // 3.5 TB/s; the sums keep their row order)
      const int64_t stride = (int64_t)gridDim.x * 4;
      int64_t row = blockIdx.x * 4LL + wave;
      for (; row + 3 * stride < q.n; row += 4 * stride) {
        float4 v[4], gd[4];
        int64_t src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          src[u] = bn_row(q, row + u * stride);
          v[u] = *reinterpret_cast<const float4*>(q.x + src[u] * q.ldx + c);
          gd[u] = *reinterpret_cast<const float4*>(dy + (row + u * stride) * ldd + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float xv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          const float gv[4] = {gd[u].x, gd[u].y, gd[u].z, gd[u].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float xhat, gate;
            bn_elem(q, xv[k], mean[k], rstd[k], g[k], b[k], src[u], c + k, xhat, gate);
            const float d = gv[k] * gate;
            sd[k] += d;
            sx[k] = fmaf(d, xhat, sx[k]);
          }
        }
      }
      for (; row < q.n; row += stride) {
        const int64_t src = bn_row(q, row);
        const float4 v = *reinterpret_cast<const float4*>(q.x + src * q.ldx + c);
        const float4 gd = *reinterpret_cast<const float4*>(dy + row * ldd + c);
        const float xv[4] = {v.x, v.y, v.z, v.w};
        const float gv[4] = {gd.x, gd.y, gd.z, gd.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float xhat, gate;
          bn_elem(q, xv[k], mean[k], rstd[k], g[k], b[k], src, c + k, xhat, gate);
          const float d = gv[k] * gate;
          sd[k] += d;
          sx[k] = fmaf(d, xhat, sx[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sh[wave][0][j * 256 + lane * 4 + k] = sd[k];
      sh[wave][1][j * 256 + lane * 4 + k] = sx[k];
    }
  }
  __syncthreads();
  for (int64_t c = threadIdx.x; c < q.C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < 4; ++w) { a += sh[w][0][c]; b += sh[w][1][c]; }
    part[((int64_t)blockIdx.x * 2) * q.C + c] = a;
    part[((int64_t)blockIdx.x * 2 + 1) * q.C + c] = b;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ part, int nblocks, int64_t C,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma) {
  float a, b; bool owner; int64_t c;
  merge_partials(part, nblocks, C, a, b, owner, c);
  if (!owner) return;
  dbeta[c] = a;
  dgamma[c] = b;
}

// dx = gamma * rstd * (d - (sum d + xhat * sum d xhat) / n_stat)    (train);   dx = gamma * rstd * d   (eval)
// COLSUM: the block also leaves the column sums of the dx rows it wrote in part[blockIdx.x][c] (blocks 2i / 2i + 1 are the
// two slots of pair i in merge_partials' layout; the grid is even): the gradient of a bias added in front of the BatchNorm (conv / Linear bias,
// gnn.py:47-48,296-306) is the column sum of dx -- formed here, the separate pass over dx (gy.sum(0)) disappears.
// MODE 0: as above.  The two halves of the backward of a BatchNorm whose OUTPUT was only formed for the rows `pick` (dy has one row
// per picked row, every other output row has no gradient): MODE 1 walks ALL rows with d = 0 (dx = -gamma rstd (sum d + xhat sum d xhat) / n),
// MODE 2 walks the picked rows and adds gamma rstd d to their dx rows (unique ids: no two waves touch the same row).
// MODE 3: dy already holds d = g * gate (egnn_skinny_dx_bn_bwd_f32 wrote it, possibly into dx itself: every element is read and then
// written by the same lane): no mask is recomputed.
template <bool COLSUM, int MODE = 0>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const BnParams q, const float* __restrict__ dy, int64_t ldd,
                                                               const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                               float inv_n_stat, float* __restrict__ dx, int64_t ldx_out,
                                                               float* __restrict__ part) {
  __shared__ float sh[COLSUM ? 4 : 1][COLSUM ? kMaxChunks * 256 : 1];
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int chunks = (int)((q.C + 255) / 256);
  for (int j = 0; j < chunks; ++j) {
    const int64_t c = j * 256 + lane * 4;
    float so[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (COLSUM) {
      if (c >= q.C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sh[wave][j * 256 + lane * 4 + k] = 0.f;
        continue;
      }
    } else if (c >= q.C) continue;
    float mean[4], rstd[4], g[4], b[4], sb[4], sg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mean[k] = q.mean[c + k];
      rstd[k] = rsqrtf(q.var[c + k] + q.eps);
      g[k] = q.gamma ? q.gamma[c + k] : 1.f;
      b[k] = q.beta ? q.beta[c + k] : 0.f;
      sb[k] = dbeta[c + k] * inv_n_stat;
      sg[k] = dgamma[c + k] * inv_n_stat;
    }
    for (int64_t row = blockIdx.x * 4LL + wave; row < q.n; row += (int64_t)gridDim.x * 4) {
      const int64_t src = MODE == 2 ? q.pick[row] : row;
      const float4 v = *reinterpret_cast<const float4*>(q.x + src * q.ldx + c);
      const float xv[4] = {v.x, v.y, v.z, v.w};
      float o[4];
      if constexpr (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xhat = (xv[k] - mean[k]) * rstd[k];
          o[k] = g[k] * rstd[k] * (0.f - sb[k] - xhat * sg[k]);
          if constexpr (COLSUM) so[k] += o[k];
        }
      } else {
        const float4 gd = *reinterpret_cast<const float4*>(dy + row * ldd + c);
        const float gv[4] = {gd.x, gd.y, gd.z, gd.w};
        float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (MODE == 2) prev = *reinterpret_cast<const float4*>(dx + src * ldx_out + c);
        const float pv[4] = {prev.x, prev.y, prev.z, prev.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float xhat, gate;
          if constexpr (MODE == 3) { xhat = (xv[k] - mean[k]) * rstd[k]; gate = 1.f; }   // dy is d = g * gate already
          else bn_elem(q, xv[k], mean[k], rstd[k], g[k], b[k], src, c + k, xhat, gate);
          const float d = gv[k] * gate;
          if constexpr (MODE == 2) o[k] = pv[k] + g[k] * rstd[k] * d;
          else o[k] = g[k] * rstd[k] * (d - sb[k] - xhat * sg[k]);
          if constexpr (COLSUM) so[k] += o[k];
        }
      }
      *reinterpret_cast<float4*>(dx + src * ldx_out + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if constexpr (COLSUM) {
#pragma unroll
      for (int k = 0; k < 4; ++k) sh[wave][j * 256 + lane * 4 + k] = so[k];
    }
  }
  if constexpr (COLSUM) {
    __syncthreads();
    for (int64_t c = threadIdx.x; c < q.C; c += 256) part[(int64_t)blockIdx.x * q.C + c] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
  }
}

// gamma != NULL (picked-rows backward): the MODE 2 half of dx adds gamma rstd d to the picked rows, whose column sum is gamma rstd dbeta
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nblocks, int64_t C, float* __restrict__ out,
                                                           const float* __restrict__ gamma, const float* __restrict__ var, float eps,
                                                           const float* __restrict__ dbeta) {
  float a, b; bool owner; int64_t c;
  merge_partials(part, nblocks, C, a, b, owner, c);   // nblocks = PAIRS of partial rows
  if (!owner) return;
  float r = a + b;
  if (dbeta) r += (gamma ? gamma[c] : 1.f) * rsqrtf(var[c] + eps) * dbeta[c];
  out[c] = r;
}

// nn.BatchNorm1d's state update of a training step in ONE launch (the ATen chain is five: add_, mul_, add_, mul_, add_):
//   num_batches_tracked += 1;  running = (1 - m) running + m stat, with the unbiased variance n / (n - 1) var;
//   momentum < 0: cumulative average, m = 1 / num_batches_tracked (momentum=None)
__global__ __launch_bounds__(256) void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ var, int64_t C,
                                                                float unbias, float momentum, float* __restrict__ rmean,
                                                                float* __restrict__ rvar, long long* __restrict__ tracked) {
  const long long t = tracked ? tracked[0] + 1 : 1;
  const float m = momentum < 0.f ? 1.f / (float)t : momentum;
  for (int64_t c = threadIdx.x; c < C; c += 256) {
    rmean[c] = (1.f - m) * rmean[c] + m * mean[c];
    rvar[c] = (1.f - m) * rvar[c] + (m * unbias) * var[c];
  }
  __syncthreads();
  if (threadIdx.x == 0 && tracked) tracked[0] = t;
}

// the same update with the row count read from the device (all-rank total of a node-range sharded run: no host read)
__global__ __launch_bounds__(256) void bn_running_update_dev_kernel(const float* __restrict__ mean, const float* __restrict__ var, int64_t C,
                                                                    const float* __restrict__ total, float momentum, float* __restrict__ rmean,
                                                                    float* __restrict__ rvar, long long* __restrict__ tracked) {
  const long long t = tracked ? tracked[0] + 1 : 1;
  const float m = momentum < 0.f ? 1.f / (float)t : momentum;
  const float n = total[0];
  const float unbias = n > 1.f ? n / (n - 1.f) : 1.f;
  for (int64_t c = threadIdx.x; c < C; c += 256) {
    rmean[c] = (1.f - m) * rmean[c] + m * mean[c];
    rvar[c] = (1.f - m) * rvar[c] + (m * unbias) * var[c];
  }
  __syncthreads();
  if (threadIdx.x == 0 && tracked) tracked[0] = t;
}

bool shape_ok(const void* p, int64_t ld, int64_t C) { return C > 0 && C % 4 == 0 && C <= kMaxChunks * 256 && ld % 4 == 0 && egnn_aligned16(p); }

int row_blocks(int64_t n) {
  const int64_t want = (n + 3) / 4;
  return (int)(want < 2048 ? (want < 1 ? 1 : want) : 2048);
}

}  // namespace

namespace {
// merge per-shard (mean | var | n) triples in shard order (Chan et al. pairwise update): one thread per column
__global__ void bn_merge_shards_kernel(const float* __restrict__ st, int world, int64_t C, float* __restrict__ mean,
                                       float* __restrict__ var, float* __restrict__ total) {
  const int64_t c = blockIdx.x * 256LL + threadIdx.x;
  if (c >= C) return;
  const int64_t ld = 2 * C + 1;
  float n = 0.f, m = 0.f, m2 = 0.f;   // running count, mean, sum of squared deviations
  for (int w = 0; w < world; ++w) {
    const float nb = st[w * ld + 2 * C];
    if (nb <= 0.f) continue;
    const float mb = st[w * ld + c], vb = st[w * ld + C + c];
    const float nt = n + nb, d = mb - m;
    m2 = m2 + vb * nb + d * d * (n * nb / nt);
    m = m + d * (nb / nt);
    n = nt;
  }
  mean[c] = m;
  var[c] = n > 0.f ? m2 / n : 0.f;
  if (c == 0) total[0] = n;
}
}  // namespace

extern "C" int egnn_bn_merge_shards_f32(const float* stats, int world, int64_t C, float* mean, float* var, float* total,
                                        void* stream) {
  EGNN_CHECK_ARG(world > 0 && C > 0 && stats && mean && var && total);
  hipLaunchKernelGGL(bn_merge_shards_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats, world, C, mean,
                     var, total);
  return egnn_launch_status();
}

extern "C" size_t egnn_bn_ws_floats(int64_t C) { return (size_t)kStatBlocks * 2 * (size_t)C; }

extern "C" int egnn_bn_stats_f32(const float* x, int64_t ld, int64_t n, int64_t C, float* mean, float* var, float* ws,
                                 size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(n > 0 && x && mean && var && ws && ld >= C);
  if (!shape_ok(x, ld, C)) return EGNN_EALIGN;
  if (ws_floats < egnn_bn_ws_floats(C)) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = (n + 3) / 4;
  const int nb = (int)(want < kStatBlocks ? want : kStatBlocks);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb), dim3(256), 0, st, x, ld, n, C, ws);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((unsigned)((C + kMergeCols - 1) / kMergeCols)), dim3(256), 0, st, ws, nb, x, n, C, mean, var);
  return egnn_launch_status();
}

extern "C" int egnn_bn_act_fwd_f32(const float* x, int64_t ld, int64_t n, int64_t C, const float* mean, const float* var, float eps,
                                   const float* gamma, const float* beta, int relu, float p, uint64_t seed,
                                   const uint64_t* seed_dev, float* y, int64_t ldy, void* stream) {
  EGNN_CHECK_ARG(n > 0 && x && mean && var && y && ld >= C && ldy >= C && p >= 0.f && p < 1.f);
  if (!shape_ok(x, ld, C) || !shape_ok(y, ldy, C)) return EGNN_EALIGN;
  const BnParams q{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, nullptr};
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(row_blocks(n)), dim3(256), 0, (hipStream_t)stream, q, y, ldy);
  return egnn_launch_status();
}

extern "C" int egnn_bn_act_bwd_reduce_f32(const float* x, int64_t ld, const float* dy, int64_t ld_dy, int64_t n, int64_t C,
                                          const float* mean, const float* var, float eps, const float* gamma, const float* beta,
                                          int relu, float p, uint64_t seed, const uint64_t* seed_dev, float* dgamma, float* dbeta,
                                          float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(n > 0 && x && dy && mean && var && dgamma && dbeta && ws && ld >= C && ld_dy >= C);
  if (!shape_ok(x, ld, C) || !shape_ok(dy, ld_dy, C)) return EGNN_EALIGN;
  if (ws_floats < egnn_bn_ws_floats(C)) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const BnParams q{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, nullptr};
  const int64_t want = (n + 3) / 4;
  const int nb = (int)(want < kStatBlocks ? want : kStatBlocks);
  hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(nb), dim3(256), 0, st, q, dy, ld_dy, ws);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((unsigned)((C + kMergeCols - 1) / kMergeCols)), dim3(256), 0, st, ws, nb, C, dbeta, dgamma);
  return egnn_launch_status();
}

extern "C" int egnn_bn_act_bwd_apply_f32(const float* x, int64_t ld, const float* dy, int64_t ld_dy, int64_t n, int64_t C,
                                         const float* mean, const float* var, float eps, const float* gamma, const float* beta,
                                         int relu, float p, uint64_t seed, const uint64_t* seed_dev, const float* sum_dbeta,
                                         const float* sum_dgamma, float inv_count, float* dx, int64_t ld_dx, void* stream) {
  EGNN_CHECK_ARG(n > 0 && x && dy && mean && var && sum_dbeta && sum_dgamma && dx && ld >= C && ld_dy >= C && ld_dx >= C);
  if (!shape_ok(x, ld, C) || !shape_ok(dy, ld_dy, C) || !shape_ok(dx, ld_dx, C)) return EGNN_EALIGN;
  const BnParams q{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, nullptr};
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel<false>, dim3(row_blocks(n)), dim3(256), 0, (hipStream_t)stream, q, dy, ld_dy, sum_dbeta,
                     sum_dgamma, inv_count, dx, ld_dx, nullptr);
  return egnn_launch_status();
}

// the apply half with the column sums of dx formed on the way (the bias gradient of the layer in front), for callers that put an
// all-rank reduction of the sums between the halves (SyncBN): ws of egnn_bn_ws_floats(C) floats
extern "C" int egnn_bn_act_bwd_apply_colsum_f32(const float* x, int64_t ld, const float* dy, int64_t ld_dy, int64_t n, int64_t C,
                                                const float* mean, const float* var, float eps, const float* gamma, const float* beta,
                                                int relu, float p, uint64_t seed, const uint64_t* seed_dev, const float* sum_dbeta,
                                                const float* sum_dgamma, float inv_count, float* dx, int64_t ld_dx, float* dx_colsum,
                                                float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(n > 0 && x && dy && mean && var && sum_dbeta && sum_dgamma && dx && dx_colsum && ws && ld >= C && ld_dy >= C && ld_dx >= C);
  if (!shape_ok(x, ld, C) || !shape_ok(dy, ld_dy, C) || !shape_ok(dx, ld_dx, C)) return EGNN_EALIGN;
  if (ws_floats < egnn_bn_ws_floats(C)) return EGNN_EWORKSPACE;
  const BnParams q{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, nullptr};
  int nb = row_blocks(n);
  if (nb > 2 * kStatBlocks) nb = 2 * kStatBlocks;
  nb = (nb + 1) & ~1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel<true>, dim3(nb), dim3(256), 0, st, q, dy, ld_dy, sum_dbeta, sum_dgamma, inv_count, dx, ld_dx, ws);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((C + kMergeCols - 1) / kMergeCols)), dim3(256), 0, st, ws, nb / 2, C, dx_colsum,
                     (const float*)nullptr, (const float*)nullptr, 0.f, (const float*)nullptr);
  return egnn_launch_status();
}

extern "C" int egnn_bn_act_bwd_colsum_f32(const float* x, int64_t ld, const float* dy, int64_t ld_dy, int64_t n, int64_t C,
                                          const float* mean, const float* var, float eps, const float* gamma, const float* beta, int relu,
                                          float p, uint64_t seed, const uint64_t* seed_dev, int batch_stats, float* dgamma, float* dbeta,
                                          float* dx, int64_t ld_dx, float* dx_colsum, float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(dx && ld_dx >= C);
  const int rc = egnn_bn_act_bwd_reduce_f32(x, ld, dy, ld_dy, n, C, mean, var, eps, gamma, beta, relu, p, seed, seed_dev, dgamma, dbeta,
                                            ws, ws_floats, stream);
  if (rc != EGNN_OK) return rc;
  const float inv_count = batch_stats ? 1.f / (float)n : 0.f;
  if (!dx_colsum)
    return egnn_bn_act_bwd_apply_f32(x, ld, dy, ld_dy, n, C, mean, var, eps, gamma, beta, relu, p, seed, seed_dev, dbeta, dgamma, inv_count, dx,
                                     ld_dx, stream);
  EGNN_CHECK_ARG(n > 0 && x && dy && mean && var && ld >= C && ld_dy >= C);
  if (!shape_ok(dx, ld_dx, C)) return EGNN_EALIGN;
  const BnParams q{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, nullptr};
  int nb = row_blocks(n);
  if (nb > 2 * kStatBlocks) nb = 2 * kStatBlocks;   // the workspace holds 2 kStatBlocks partial rows (the reduce half is done with it: same stream)
  nb = (nb + 1) & ~1;                               // whole pairs; a block past the rows writes zeros
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel<true>, dim3(nb), dim3(256), 0, st, q, dy, ld_dy, dbeta, dgamma, inv_count, dx, ld_dx, ws);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((C + kMergeCols - 1) / kMergeCols)), dim3(256), 0, st, ws, nb / 2, C, dx_colsum,
                     (const float*)nullptr, (const float*)nullptr, 0.f, (const float*)nullptr);
  return egnn_launch_status();
}

// ---- the same fused BatchNorm + activation whose OUTPUT is only needed on the rows `pick` (unique ids) ---------------------------
// The projection heads of the sampled criteria (gnn.py:296-306 feeding criterion.py:62-65,134-137): the statistics span all n rows of
// x, the criterion reads max_samples of the output rows.  Forward: only those rows are formed.  Backward: the reductions run over
// the picked rows (every other row has d = 0), dx still has all n rows (the mean / variance terms reach every row).
extern "C" int egnn_bn_act_rows_fwd_f32(const float* x, int64_t ld, int64_t n, int64_t C, const int64_t* pick, int64_t n_pick,
                                        const float* mean, const float* var, float eps, const float* gamma, const float* beta, int relu,
                                        float p, uint64_t seed, const uint64_t* seed_dev, float* y, int64_t ldy, void* stream) {
  EGNN_CHECK_ARG(n > 0 && n_pick > 0 && n_pick <= n && pick && x && mean && var && y && ld >= C && ldy >= C && p >= 0.f && p < 1.f);
  if (!shape_ok(x, ld, C) || !shape_ok(y, ldy, C)) return EGNN_EALIGN;
  const BnParams q{x, ld, n_pick, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, pick};
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(row_blocks(n_pick)), dim3(256), 0, (hipStream_t)stream, q, y, ldy);
  return egnn_launch_status();
}

// The backward of the picked-rows form in its two halves (a node-range shard all-reduces [sum d | sum d xhat] between them: SyncBN).
extern "C" int egnn_bn_act_rows_bwd_reduce_f32(const float* x, int64_t ld, int64_t n, int64_t C, const int64_t* pick, int64_t n_pick,
                                               const float* dy, int64_t ld_dy, const float* mean, const float* var, float eps,
                                               const float* gamma, const float* beta, int relu, float p, uint64_t seed,
                                               const uint64_t* seed_dev, float* dgamma, float* dbeta, float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(n > 0 && n_pick > 0 && n_pick <= n && pick && x && dy && mean && var && dgamma && dbeta && ws && ld >= C && ld_dy >= C);
  if (!shape_ok(x, ld, C) || !shape_ok(dy, ld_dy, C)) return EGNN_EALIGN;
  if (ws_floats < egnn_bn_ws_floats(C)) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const BnParams qp{x, ld, n_pick, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, pick};
  // sum d, sum d xhat over the picked rows (every other output row has no gradient)
  const int64_t want = (n_pick + 3) / 4;
  const int nbr = (int)(want < kStatBlocks ? want : kStatBlocks);
  hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(nbr), dim3(256), 0, st, qp, dy, ld_dy, ws);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((unsigned)((C + kMergeCols - 1) / kMergeCols)), dim3(256), 0, st, ws, nbr, C, dbeta, dgamma);
  return egnn_launch_status();
}

// sum_dbeta / sum_dgamma: the sums the mean / variance terms use (this tensor's own, or the all-rank ones), inv_count = 1 / rows they span
// (0: running statistics, no such terms).  local_dbeta (needed with dx_colsum): sum d over THIS tensor's picked rows -- the picked rows'
// own term gamma rstd d sums to gamma rstd local_dbeta per column.  n_pick == 0 (a shard without sampled rows): every row still
// receives the mean / variance terms.
extern "C" int egnn_bn_act_rows_bwd_apply_f32(const float* x, int64_t ld, int64_t n, int64_t C, const int64_t* pick, int64_t n_pick,
                                              const float* dy, int64_t ld_dy, const float* mean, const float* var, float eps,
                                              const float* gamma, const float* beta, int relu, float p, uint64_t seed,
                                              const uint64_t* seed_dev, const float* sum_dbeta, const float* sum_dgamma, float inv_count,
                                              const float* local_dbeta, float* dx, int64_t ld_dx, float* dx_colsum, float* ws,
                                              size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(n > 0 && n_pick >= 0 && n_pick <= n && x && mean && var && sum_dbeta && sum_dgamma && dx && ld >= C && ld_dx >= C);
  EGNN_CHECK_ARG(n_pick == 0 || (pick && dy && ld_dy >= C));
  EGNN_CHECK_ARG(dx_colsum == nullptr || n_pick == 0 || local_dbeta);
  if (!shape_ok(x, ld, C) || !shape_ok(dx, ld_dx, C) || (n_pick > 0 && !shape_ok(dy, ld_dy, C))) return EGNN_EALIGN;
  if (dx_colsum && (ws == nullptr || ws_floats < egnn_bn_ws_floats(C))) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const BnParams qp{x, ld, n_pick, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, pick};
  const BnParams qa{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, nullptr};
  // every row: the mean / variance terms;  then the picked rows: + gamma rstd d
  int nb = row_blocks(n);
  if (dx_colsum) {
    if (nb > 2 * kStatBlocks) nb = 2 * kStatBlocks;
    nb = (nb + 1) & ~1;
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<true, 1>), dim3(nb), dim3(256), 0, st, qa, (const float*)nullptr, (int64_t)0, sum_dbeta, sum_dgamma,
                       inv_count, dx, ld_dx, ws);
  } else {
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<false, 1>), dim3(nb), dim3(256), 0, st, qa, (const float*)nullptr, (int64_t)0, sum_dbeta, sum_dgamma,
                       inv_count, dx, ld_dx, (float*)nullptr);
  }
  if (n_pick > 0)
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<false, 2>), dim3(row_blocks(n_pick)), dim3(256), 0, st, qp, dy, ld_dy, sum_dbeta, sum_dgamma, inv_count,
                       dx, ld_dx, (float*)nullptr);
  if (dx_colsum)
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((C + kMergeCols - 1) / kMergeCols)), dim3(256), 0, st, ws, nb / 2, C, dx_colsum, gamma,
                       var, eps, n_pick > 0 ? local_dbeta : (const float*)nullptr);
  return egnn_launch_status();
}

extern "C" int egnn_bn_act_rows_bwd_f32(const float* x, int64_t ld, int64_t n, int64_t C, const int64_t* pick, int64_t n_pick,
                                        const float* dy, int64_t ld_dy, const float* mean, const float* var, float eps, const float* gamma,
                                        const float* beta, int relu, float p, uint64_t seed, const uint64_t* seed_dev, int batch_stats,
                                        float* dgamma, float* dbeta, float* dx, int64_t ld_dx, float* dx_colsum, float* ws, size_t ws_floats,
                                        void* stream) {
  EGNN_CHECK_ARG(n > 0 && n_pick > 0 && dx && ld_dx >= C);
  if (!shape_ok(dx, ld_dx, C)) return EGNN_EALIGN;
  const int rc = egnn_bn_act_rows_bwd_reduce_f32(x, ld, n, C, pick, n_pick, dy, ld_dy, mean, var, eps, gamma, beta, relu, p, seed, seed_dev, dgamma,
                                                 dbeta, ws, ws_floats, stream);
  if (rc != EGNN_OK) return rc;
  return egnn_bn_act_rows_bwd_apply_f32(x, ld, n, C, pick, n_pick, dy, ld_dy, mean, var, eps, gamma, beta, relu, p, seed, seed_dev, dbeta, dgamma,
                                        batch_stats ? 1.f / (float)n : 0.f, dbeta, dx, ld_dx, dx_colsum, ws, ws_floats, stream);
}

extern "C" int egnn_bn_act_bwd_f32(const float* x, int64_t ld, const float* dy, int64_t ld_dy, int64_t n, int64_t C,
                                   const float* mean, const float* var, float eps, const float* gamma, const float* beta, int relu,
                                   float p, uint64_t seed, const uint64_t* seed_dev, int batch_stats, float* dgamma, float* dbeta,
                                   float* dx, int64_t ld_dx, float* ws, size_t ws_floats, void* stream) {
  return egnn_bn_act_bwd_colsum_f32(x, ld, dy, ld_dy, n, C, mean, var, eps, gamma, beta, relu, p, seed, seed_dev, batch_stats, dgamma, dbeta, dx,
                                    ld_dx, nullptr, ws, ws_floats, stream);
}

extern "C" int egnn_bn_running_update_f32(const float* mean, const float* var, int64_t C, int64_t n, float momentum, float* running_mean,
                                          float* running_var, int64_t* num_batches_tracked, void* stream) {
  EGNN_CHECK_ARG(C > 0 && n > 0 && mean && var && running_mean && running_var);
  const float unbias = n > 1 ? (float)n / (float)(n - 1) : 1.f;
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean, var, C, unbias, momentum, running_mean,
                     running_var, (long long*)num_batches_tracked);
  return egnn_launch_status();
}

extern "C" int egnn_bn_running_update_dev_f32(const float* mean, const float* var, int64_t C, const float* total_rows, float momentum,
                                              float* running_mean, float* running_var, int64_t* num_batches_tracked, void* stream) {
  EGNN_CHECK_ARG(C > 0 && mean && var && total_rows && running_mean && running_var);
  hipLaunchKernelGGL(bn_running_update_dev_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean, var, C, total_rows, momentum, running_mean,
                     running_var, (long long*)num_batches_tracked);
  return egnn_launch_status();
}

namespace {
// ---- dX of a narrow Linear fused with the backward of the BatchNorm + activation in front of it -----------------------------
// The last hidden layer of the student (gnn.py:47-52): h = drop(relu(bn(y))), out = A^ (h W3) with W3 [C, Ks] (Ks = classes <= 64).
// loss.backward() forms dh = G W3^T (G = d(h W3) [M, Ks]), adds the projection head's rows (gnn.py:150), and runs the BatchNorm backward:
// three passes over [M, C] tensors (write dh; read-modify-write rows; read dh + y for the column sums) before the apply pass.  Here
// the MFMA tile of dh never leaves registers: rows of the head's input gradient (add_rows[add_inv[row]]) and a dense addend are added,
// the gate is recomputed from y, d = dh * gate is stored ONCE and the column sums  sum d, sum d xhat  are formed on the way;
// the apply pass then reads d (MODE 3 above).
// Two kernels: tail_bwd_tile_kernel below (C == 256, the students' hidden width: 133 us at N = 169 343 against 97 + 51 + 79 us of
// skinny_dx + rows_add + the reduce pass) and this one for any C % 64 == 0, which keeps the epilogue in the MFMA layout of
// skinny_dx_kernel (gemm_skinny.hip): a wave owns 64 columns for `rblocks` consecutive row blocks; its column sums go to partial row
// `row super-block` (one writer per entry).
typedef float f4 __attribute__((ext_vector_type(4)));
// RT = 16-row MFMA tiles per row block (4: 64 rows, 2: 32 rows -- half the accumulators); OCC = workgroups per CU the registers are capped for;
// ADD = a dense addend is given (h had another dense consumer: not the case in the reference's models)
template <int KSTEPS, int RT, int OCC, bool ADD>
__global__ __launch_bounds__(256, OCC) void skinny_dx_bn_kernel(const float* __restrict__ G, int64_t ldg, const float* __restrict__ B, int64_t ldb,
                                                                int b_kmajor, int Ks, float alpha, int rblocks, const float* __restrict__ addend,
                                                                int64_t ld_addend, const float* __restrict__ add_rows, int64_t ld_add_rows,
                                                                const int32_t* __restrict__ add_inv, const BnParams q, float* __restrict__ D,
                                                                int64_t ldd, float* __restrict__ part) {
  constexpr int RB = 16 * RT;                            // rows per block iteration
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int r16 = lane & 15, qq = lane >> 4;
  const int64_t M = q.n;
  const int Nbig = (int)q.C;
  const int ncb = Nbig / 64;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;   // (row super-block, column block)
  const int64_t sb = item / ncb;
  const int cb = (int)(item % ncb);
  const int64_t mbeg = sb * RB * rblocks;
  if (mbeg >= M) return;
  const int c0 = cb * 64 + 4 * r16;
  float4 bf[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int k = 4 * s + qq;
    bf[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < Ks) {
      if (!b_kmajor) bf[s] = *reinterpret_cast<const float4*>(B + (int64_t)k * ldb + c0);
      else bf[s] = make_float4(B[(int64_t)c0 * ldb + k], B[(int64_t)(c0 + 1) * ldb + k], B[(int64_t)(c0 + 2) * ldb + k], B[(int64_t)(c0 + 3) * ldb + k]);
    }
  }
  float mean[4], rstd[4], gm[4], bt[4], sd[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mean[k] = q.mean[c0 + k];
    rstd[k] = rsqrtf(q.var[c0 + k] + q.eps);
    gm[k] = q.gamma ? q.gamma[c0 + k] : 1.f;
    bt[k] = q.beta ? q.beta[c0 + k] : 0.f;
  }
  for (int rbi = 0; rbi < rblocks; ++rbi) {
    const int64_t m0 = mbeg + (int64_t)rbi * RB;
    if (m0 >= M) break;
    const float* gp[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      int64_t r = m0 + t * 16 + r16;
      if (r >= M) r = M - 1;
      gp[t] = G + r * ldg + qq;
    }
    f4 acc[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[t][n] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {               // one MFMA step: k = 4s + qq
      float a[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) a[t] = (4 * s + qq < Ks) ? gp[t][4 * s] : 0.f;
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bf[s].x, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bf[s].y, acc[t][1], 0, 0, 0);
        acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bf[s].z, acc[t][2], 0, 0, 0);
        acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bf[s].w, acc[t][3], 0, 0, 0);
      }
    }
    // epilogue in groups of four rows (fixed t): every load of a group is issued before the first use (index -> head row -> y), rows past
    // the end are clamped for the loads and dropped at the store / sums, so that nothing in the group is control-dependent
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      // 32-bit element offsets off the (scalar) base pointers: the host entry checks that every tensor is below 2^31 elements
      unsigned rowc[4];
      int slot[4];
      float4 yv[4], ev[4], av[ADD ? 4 : 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + t * 16 + 4 * qq + r;
        rowc[r] = (unsigned)(row < M ? row : M - 1);
        slot[r] = add_inv ? add_inv[rowc[r]] : -1;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        yv[r] = *reinterpret_cast<const float4*>(q.x + (rowc[r] * (unsigned)q.ldx + (unsigned)c0));
        if constexpr (ADD) av[r] = *reinterpret_cast<const float4*>(addend + (rowc[r] * (unsigned)ld_addend + (unsigned)c0));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ev[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (add_inv) {
          const float4 e = *reinterpret_cast<const float4*>(add_rows + ((unsigned)(slot[r] < 0 ? 0 : slot[r]) * (unsigned)ld_add_rows + (unsigned)c0));
          const float m = slot[r] < 0 ? 0.f : 1.f;
          ev[r] = make_float4(e.x * m, e.y * m, e.z * m, e.w * m);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool live = m0 + t * 16 + 4 * qq + r < M;
        float dh[4] = {alpha * acc[t][0][r] + ev[r].x, alpha * acc[t][1][r] + ev[r].y, alpha * acc[t][2][r] + ev[r].z, alpha * acc[t][3][r] + ev[r].w};
        if constexpr (ADD) { dh[0] += av[r].x; dh[1] += av[r].y; dh[2] += av[r].z; dh[3] += av[r].w; }
        const float xv[4] = {yv[r].x, yv[r].y, yv[r].z, yv[r].w};
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float xhat, gate;
          bn_elem(q, xv[k], mean[k], rstd[k], gm[k], bt[k], (int64_t)rowc[r], c0 + k, xhat, gate);
          d[k] = live ? dh[k] * gate : 0.f;
          sd[k] += d[k];
          sx[k] = fmaf(d[k], xhat, sx[k]);
        }
        if (live) *reinterpret_cast<float4*>(D + (rowc[r] * (unsigned)ldd + (unsigned)c0)) = make_float4(d[0], d[1], d[2], d[3]);
        __builtin_amdgcn_sched_barrier(0);   // one row at a time: the mask chains of sixteen elements interleaved cost a register each
      }
    }
  }
  // the four lane groups (qq) of a column quartet hold disjoint rows: fixed-order butterfly, lane group 0 writes
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sd[k] += __shfl_xor(sd[k], 16); sd[k] += __shfl_xor(sd[k], 32);
    sx[k] += __shfl_xor(sx[k], 16); sx[k] += __shfl_xor(sx[k], 32);
  }
  if (qq == 0) {
    *reinterpret_cast<float4*>(part + (sb * 2) * Nbig + c0) = make_float4(sd[0], sd[1], sd[2], sd[3]);
    *reinterpret_cast<float4*>(part + (sb * 2 + 1) * Nbig + c0) = make_float4(sx[0], sx[1], sx[2], sx[3]);
  }
}
// The same operation with the MFMA tile handed to a ROW-STREAMING epilogue through LDS ("tile" form; C == 256).  A workgroup walks
// 32-row blocks: (1) the G rows of the block are staged in LDS; (2) wave w forms the dh tile of columns [64 w, 64 w + 64) with
// v_mfma_f32_16x16x4_f32 and writes it to an LDS tile [32][256]; (3) the four waves stream the rows like bn_act_bwd_reduce_kernel does
// (a lane owns 4 fixed columns, a wave reads / writes whole 1 KB rows: y, the head's row, d), four rows in flight per wave.  The column sums
// stay in registers over the whole walk; one partial row per workgroup.  The MFMA-layout kernel above touches memory in 256-byte
// pieces of four rows per instruction and holds its loads behind the tile loop: 215-250 us at N = 169 343 (360 before its loads were
// grouped); this form is built like the streaming kernels and takes 133 us.
template <int KSTEPS, bool ADD>
__global__ __launch_bounds__(256, 3) void tail_bwd_tile_kernel(const float* __restrict__ G, int64_t ldg, const float* __restrict__ B, int64_t ldb,
                                                               int b_kmajor, int Ks, float alpha, const float* __restrict__ addend,
                                                               int64_t ld_addend, const float* __restrict__ add_rows, int64_t ld_add_rows,
                                                               const int32_t* __restrict__ add_inv, const BnParams q, float* __restrict__ D,
                                                               int64_t ldd, float* __restrict__ part) {
  constexpr int RB = 32, KP = KSTEPS * 4, LDG = KP + 1, LDT = 256 + 4;
  __shared__ float sG[RB * LDG];
  __shared__ __attribute__((aligned(16))) float sT[RB * LDT];
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int r16 = lane & 15, qq = lane >> 4;
  const int64_t M = q.n;
  // phase-2 ownership: columns 4 lane .. 4 lane + 3 of whole rows
  const int c2 = 4 * lane;
  float mean[4], rstd[4], gm[4], bt[4], sd[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mean[k] = q.mean[c2 + k];
    rstd[k] = rsqrtf(q.var[c2 + k] + q.eps);
    gm[k] = q.gamma ? q.gamma[c2 + k] : 1.f;
    bt[k] = q.beta ? q.beta[c2 + k] : 0.f;
  }
  // phase-1 ownership: MFMA tiles of columns 64 wave + 4 r16 + j
  const int c1 = wave * 64 + 4 * r16;
  float4 bf[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int k = 4 * s + qq;
    bf[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < Ks) {
      if (!b_kmajor) bf[s] = *reinterpret_cast<const float4*>(B + (int64_t)k * ldb + c1);
      else bf[s] = make_float4(B[(int64_t)c1 * ldb + k], B[(int64_t)(c1 + 1) * ldb + k], B[(int64_t)(c1 + 2) * ldb + k], B[(int64_t)(c1 + 3) * ldb + k]);
    }
  }
  const int64_t nblk = (M + RB - 1) / RB;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t m0 = blk * RB;
    // (1) G rows of the block -> LDS (zero past Ks / past M)
    for (int i = threadIdx.x; i < RB * KP; i += 256) {
      const int r = i / KP, k = i % KP;
      const int64_t row = m0 + r;
      sG[r * LDG + k] = (row < M && k < Ks) ? G[row * ldg + k] : 0.f;
    }
    __syncthreads();
    // (2) dh tile of this wave's 64 columns
    f4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[t][n] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const float a0 = sG[r16 * LDG + 4 * s + qq], a1 = sG[(16 + r16) * LDG + 4 * s + qq];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bf[s].x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bf[s].y, acc[0][1], 0, 0, 0);
      acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bf[s].z, acc[0][2], 0, 0, 0);
      acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bf[s].w, acc[0][3], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bf[s].x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bf[s].y, acc[1][1], 0, 0, 0);
      acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bf[s].z, acc[1][2], 0, 0, 0);
      acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bf[s].w, acc[1][3], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(sT + (16 * t + 4 * qq + r) * LDT + c1) =
            make_float4(alpha * acc[t][0][r], alpha * acc[t][1][r], alpha * acc[t][2][r], alpha * acc[t][3][r]);
    __syncthreads();
    // (3) stream the rows: wave w takes rows w, w + 4, ... of the block, four at a time (32-bit element offsets off scalar bases:
    //     the host entry keeps every tensor below 2^31 elements)
#pragma unroll
    for (int g4 = 0; g4 < RB / 16; ++g4) {
      unsigned rowc[4];
      int slot[4];
      float4 yv[4], ev[4], av[ADD ? 4 : 1], tv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t row = m0 + wave + 4 * (4 * g4 + u);
        rowc[u] = (unsigned)(row < M ? row : M - 1);
        slot[u] = add_inv ? add_inv[rowc[u]] : -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        yv[u] = *reinterpret_cast<const float4*>(q.x + (rowc[u] * (unsigned)q.ldx + (unsigned)c2));
        if constexpr (ADD) av[u] = *reinterpret_cast<const float4*>(addend + (rowc[u] * (unsigned)ld_addend + (unsigned)c2));
        tv[u] = *reinterpret_cast<const float4*>(sT + (wave + 4 * (4 * g4 + u)) * LDT + c2);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ev[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot[u] >= 0) ev[u] = *reinterpret_cast<const float4*>(add_rows + ((unsigned)slot[u] * (unsigned)ld_add_rows + (unsigned)c2));   // wave-uniform
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool live = m0 + wave + 4 * (4 * g4 + u) < M;
        float dh[4] = {tv[u].x + ev[u].x, tv[u].y + ev[u].y, tv[u].z + ev[u].z, tv[u].w + ev[u].w};
        if constexpr (ADD) { dh[0] += av[u].x; dh[1] += av[u].y; dh[2] += av[u].z; dh[3] += av[u].w; }
        const float xv[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float xhat, gate;
          bn_elem(q, xv[k], mean[k], rstd[k], gm[k], bt[k], (int64_t)rowc[u], c2 + k, xhat, gate);
          d[k] = live ? dh[k] * gate : 0.f;
          sd[k] += d[k];
          sx[k] = fmaf(d[k], xhat, sx[k]);
        }
        if (live) *reinterpret_cast<float4*>(D + (rowc[u] * (unsigned)ldd + (unsigned)c2)) = make_float4(d[0], d[1], d[2], d[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // sT / sG are rewritten by the next block
  }
  // column sums: the four waves hold disjoint rows of the same columns; fixed wave order through LDS (sT is free now)
  float* red = sT;   // [4][2][256]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[(wave * 2) * 256 + c2 + k] = sd[k];
    red[(wave * 2 + 1) * 256 + c2 + k] = sx[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int which = i / 256, c = i % 256;
    const float v = ((red[(0 * 2 + which) * 256 + c] + red[(1 * 2 + which) * 256 + c]) + red[(2 * 2 + which) * 256 + c]) + red[(3 * 2 + which) * 256 + c];
    part[((int64_t)blockIdx.x * 2 + which) * 256 + c] = v;
  }
}
constexpr int kTailTileBlocks = 768;   // workgroups of the tile form = partial rows of its column sums (3 per CU)

constexpr int kDxBnRows = 256;      // rows per wave of skinny_dx_bn_kernel (one partial row of the column sums per 256 rows)
inline bool tail_tile_form(int64_t C, int ksteps) {
  return C == 256 && ksteps <= 10;   // (other widths take the MFMA-layout kernel: tests cover C = 64 / 128)
}
}  // namespace

namespace {
// Forward of the pair in the same "tile" form (C == 256, Ks <= 64): a workgroup walks RB-row blocks; (A) its waves stream rows of x like
// bn_act_fwd_kernel (1 KB per row and wave), store h to memory AND to an LDS tile; (B) wave w multiplies the k-range [64 w, 64 w + 64) of
// the tile with its slice of W (v_mfma_f32_16x16x4_f32, W fragments in registers for the whole walk); (C) the four k-range partials are
// added in wave order and xw rows are stored.  h is bit-identical to bn_act_fwd_kernel; x is read once, h is never re-read.
// N = 169 343, Ks = 40: 115 us against 77 + 67 us of bn_act_fwd_kernel + skinny_fwd_kernel (the same transform inside skinny_fwd_kernel's
// operand load, on its MFMA-shaped 64-byte pieces: 163 us).
// BN = false: the plain narrow product XW = alpha x W + bias of a 256-wide x in the same form (rows streamed into the LDS tile as they
// are; q carries only x / ldx / n): the output conv in test() and SAGEConv's two narrow Linears.
template <int NT, bool BN = true>
__global__ __launch_bounds__(256, 4) void tail_fwd_tile_kernel(const BnParams q, const float* __restrict__ W, int64_t ldw, int w_kmajor,
                                                               int Ks, float* __restrict__ H, int64_t ldh, float* __restrict__ XW,
                                                               int64_t ldxw, const float* __restrict__ bias = nullptr, float alpha = 1.f) {
  constexpr int RB = 16, LDH = 256 + 4, NP = NT * 16, LDP = NP + 1, RTL = RB / 16;   // 32-row blocks measured slower (2 workgroups per CU)
  __shared__ __attribute__((aligned(16))) float sH[RB * LDH];
  __shared__ float sP[4 * RB * LDP];
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int r16 = lane & 15, qq = lane >> 4;
  const int64_t M = q.n;
  const int c2 = 4 * lane;
  float mean[4], rstd[4], gm[4], bt[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mean[k] = BN ? q.mean[c2 + k] : 0.f;
    rstd[k] = BN ? rsqrtf(q.var[c2 + k] + q.eps) : 1.f;
    gm[k] = (BN && q.gamma) ? q.gamma[c2 + k] : 1.f;
    bt[k] = (BN && q.beta) ? q.beta[c2 + k] : 0.f;
  }
  // W fragments of this wave's k-range: step s <-> k = 64 wave + 4 s + qq, tile j <-> n = 16 j + r16
  float bfr[16][NT];
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int k = 64 * wave + 4 * s + qq, n = 16 * j + r16;
      bfr[s][j] = n < Ks ? (w_kmajor ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n]) : 0.f;
    }
  const int64_t nblk = (M + RB - 1) / RB;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t m0 = blk * RB;
    // (A) rows wave, wave + 4, ... of the block, four in flight
#pragma unroll
    for (int g4 = 0; g4 < RB / 16; ++g4) {
      unsigned rowc[4];
      float4 yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t row = m0 + wave + 4 * (4 * g4 + u);
        rowc[u] = (unsigned)(row < M ? row : M - 1);
        yv[u] = *reinterpret_cast<const float4*>(q.x + (rowc[u] * (unsigned)q.ldx + (unsigned)c2));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int lr = wave + 4 * (4 * g4 + u);
        if constexpr (BN) {
          const float xv[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
          float o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float xhat, gate;
            bn_elem(q, xv[k], mean[k], rstd[k], gm[k], bt[k], (int64_t)rowc[u], c2 + k, xhat, gate);
            o[k] = (gm[k] * xhat + bt[k]) * gate;
          }
          const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
          if (m0 + lr < M) *reinterpret_cast<float4*>(H + (rowc[u] * (unsigned)ldh + (unsigned)c2)) = ov;
          *reinterpret_cast<float4*>(sH + lr * LDH + c2) = ov;
          __builtin_amdgcn_sched_barrier(0);
        } else {
          *reinterpret_cast<float4*>(sH + lr * LDH + c2) = yv[u];
        }
      }
    }
    __syncthreads();
    // (B) partial products of this wave's 64 k-values
    f4 acc[RTL][NT];
#pragma unroll
    for (int t = 0; t < RTL; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[t][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int t = 0; t < RTL; ++t) {
        const float a = sH[(16 * t + r16) * LDH + 64 * wave + 4 * s + qq];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfr[s][j], acc[t][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < RTL; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sP[(wave * RB + 16 * t + 4 * qq + r) * LDP + 16 * j + r16] = acc[t][j][r];
    __syncthreads();
    // (C) the four k-ranges in wave order
    for (int i = threadIdx.x; i < RB * Ks; i += 256) {
      const int r = i / Ks, c = i % Ks;
      if (m0 + r < M) {
        const float v = ((sP[(0 * RB + r) * LDP + c] + sP[(1 * RB + r) * LDP + c]) + sP[(2 * RB + r) * LDP + c]) + sP[(3 * RB + r) * LDP + c];
        XW[(m0 + r) * ldxw + c] = BN ? v : alpha * v + (bias ? bias[c] : 0.f);
      }
    }
  }
}
}  // namespace

// Y = alpha X W + bias for a 256-wide X and a narrow W in the tile form (see tail_fwd_tile_kernel<NT, false>); 1 = shape not taken.
// Called by gemm.hip in front of egnn_skinny_fwd (gemm_skinny.hip).
int egnn_skinny_fwd_tile(const float* X, int64_t ldx, const float* W, int64_t ldw, int w_kmajor, const float* bias, float* Y, int64_t ldy, int64_t M,
                         int64_t N, int64_t K, float alpha, hipStream_t st) {
  if (K != 256 || N < 1 || N > 64 || M < 4096 || ldx % 4 != 0 || !egnn_aligned16(X) || M * ldx >= (1LL << 31) - 64) return 1;
  const BnParams q{X, ldx, M, K, nullptr, nullptr, 0.f, nullptr, nullptr, 0, 0.f, 0ull, nullptr, nullptr};
  const int nt = (int)((N + 15) / 16);
  const int64_t nblk = (M + 15) / 16;
  const unsigned grid = (unsigned)(nblk < 1024 ? nblk : 1024);
  switch (nt) {
    case 1: hipLaunchKernelGGL((tail_fwd_tile_kernel<1, false>), dim3(grid), dim3(256), 0, st, q, W, ldw, w_kmajor, (int)N, (float*)nullptr, (int64_t)0, Y, ldy, bias, alpha); break;
    case 2: hipLaunchKernelGGL((tail_fwd_tile_kernel<2, false>), dim3(grid), dim3(256), 0, st, q, W, ldw, w_kmajor, (int)N, (float*)nullptr, (int64_t)0, Y, ldy, bias, alpha); break;
    case 3: hipLaunchKernelGGL((tail_fwd_tile_kernel<3, false>), dim3(grid), dim3(256), 0, st, q, W, ldw, w_kmajor, (int)N, (float*)nullptr, (int64_t)0, Y, ldy, bias, alpha); break;
    default: hipLaunchKernelGGL((tail_fwd_tile_kernel<4, false>), dim3(grid), dim3(256), 0, st, q, W, ldw, w_kmajor, (int)N, (float*)nullptr, (int64_t)0, Y, ldy, bias, alpha); break;
  }
  return egnn_launch_status();
}

extern "C" int egnn_bn_act_linear_fwd_f32(const float* x, int64_t ld, int64_t n, int64_t C, const float* mean, const float* var, float eps,
                                          const float* gamma, const float* beta, int relu, float p, uint64_t seed, const uint64_t* seed_dev,
                                          const float* W, int64_t ldw, int w_kmajor, int64_t Ks, float* h, int64_t ldh, float* xw,
                                          int64_t ld_xw, void* stream) {
  EGNN_CHECK_ARG(n > 0 && x && mean && var && W && h && xw && ld >= C && ldh >= C && ld_xw >= Ks && p >= 0.f && p < 1.f);
  if (!shape_ok(x, ld, C) || !shape_ok(h, ldh, C)) return EGNN_EALIGN;
  const BnParams q{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, nullptr};
  if (C != 256 || Ks < 1 || Ks > 64 || n * ld >= (1LL << 31) - 64 || n * ldh >= (1LL << 31) - 64) return EGNN_EALIGN;
  const int nt = (int)((Ks + 15) / 16);
  const int64_t nblk = (n + 15) / 16;
  const unsigned grid = (unsigned)(nblk < 1024 ? nblk : 1024);   // 4 workgroups per CU (29 KB of LDS, <= 128 VGPRs each)
  hipStream_t st = (hipStream_t)stream;
  switch (nt) {
    case 1: hipLaunchKernelGGL((tail_fwd_tile_kernel<1, true>), dim3(grid), dim3(256), 0, st, q, W, ldw, w_kmajor, (int)Ks, h, ldh, xw, ld_xw, (const float*)nullptr, 1.f); break;
    case 2: hipLaunchKernelGGL((tail_fwd_tile_kernel<2, true>), dim3(grid), dim3(256), 0, st, q, W, ldw, w_kmajor, (int)Ks, h, ldh, xw, ld_xw, (const float*)nullptr, 1.f); break;
    case 3: hipLaunchKernelGGL((tail_fwd_tile_kernel<3, true>), dim3(grid), dim3(256), 0, st, q, W, ldw, w_kmajor, (int)Ks, h, ldh, xw, ld_xw, (const float*)nullptr, 1.f); break;
    default: hipLaunchKernelGGL((tail_fwd_tile_kernel<4, true>), dim3(grid), dim3(256), 0, st, q, W, ldw, w_kmajor, (int)Ks, h, ldh, xw, ld_xw, (const float*)nullptr, 1.f); break;
  }
  return egnn_launch_status();
}

extern "C" size_t egnn_skinny_dx_bn_ws_floats(int64_t M, int64_t C) {
  size_t sbs = (size_t)((M + kDxBnRows - 1) / kDxBnRows);
  if (sbs < (size_t)kTailTileBlocks) sbs = kTailTileBlocks;
  const size_t a = sbs * 2 * (size_t)C, b = egnn_bn_ws_floats(C);
  return a > b ? a : b;
}

// The tail backward in its two halves, so that a node-range shard can put the all-rank sum of (sum d, sum d xhat) between them
// (SyncBN, dist.py): `reduce` leaves d = dh * gate in dx and this shard's column sums in dbeta / dgamma; `apply` turns the stored d
// into dx with whatever sums / 1/count it is handed.  egnn_skinny_dx_bn_bwd_f32 = reduce + apply with the local sums and 1/M.
extern "C" int egnn_skinny_dx_bn_bwd_reduce_f32(const float* G, int64_t ldg, const float* W, int64_t ldw, int w_kmajor, int64_t M, int64_t C,
                                                int64_t Ks, float alpha, const float* addend, int64_t ld_addend, const float* add_rows,
                                                int64_t ld_add_rows, const int32_t* add_inv, const float* x, int64_t ldx, const float* mean,
                                                const float* var, float eps, const float* gamma, const float* beta, int relu, float p,
                                                uint64_t seed, const uint64_t* seed_dev, float* dgamma, float* dbeta, float* dx,
                                                int64_t ld_dx, float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(M > 0 && G && W && x && mean && var && dgamma && dbeta && dx && ws && ldg >= Ks && ldx >= C && ld_dx >= C);
  EGNN_CHECK_ARG((add_inv == nullptr) == (add_rows == nullptr));
  if (Ks > 64 || Ks < 1 || C % 64 != 0 || C < 64 || C > 1024) return EGNN_EALIGN;
  if (!shape_ok(x, ldx, C) || !shape_ok(dx, ld_dx, C)) return EGNN_EALIGN;
  if (addend && !shape_ok(addend, ld_addend, C)) return EGNN_EALIGN;
  if (add_rows && !shape_ok(add_rows, ld_add_rows, C)) return EGNN_EALIGN;
  if (w_kmajor && (ldw % 4 != 0 || !egnn_aligned16(W))) return EGNN_EALIGN;
  if (ws_floats < egnn_skinny_dx_bn_ws_floats(M, C) || !egnn_aligned16(ws)) return EGNN_EWORKSPACE;
  const int b_kmajor = w_kmajor ? 0 : 1;   // the kernels index W^T as B[k = class][n = column]: "k-major" there = W stored [C, Ks]
  const int64_t lim = (1LL << 31) - 64;   // the kernel addresses with 32-bit element offsets
  if (M * ldx >= lim || M * ld_dx >= lim || (addend && M * ld_addend >= lim) || (add_rows && M * ld_add_rows >= lim)) return EGNN_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  const BnParams q{x, ldx, M, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, nullptr};
  int64_t sbs = (M + kDxBnRows - 1) / kDxBnRows;
  const int ksteps = (int)((Ks + 3) / 4);
  if (tail_tile_form(C, ksteps)) {
    // tile form: MFMA tile -> LDS -> row-streaming epilogue
    const int64_t nblk = (M + 31) / 32;
    sbs = nblk < kTailTileBlocks ? nblk : kTailTileBlocks;
#define EGNN_TAIL_TILE(KS, ADD)                                                                                                              \
  hipLaunchKernelGGL((tail_bwd_tile_kernel<KS, ADD>), dim3((unsigned)sbs), dim3(256), 0, st, G, ldg, W, ldw, b_kmajor, (int)Ks, alpha, addend,   \
                     ld_addend, add_rows, ld_add_rows, add_inv, q, dx, ld_dx, ws)
    if (ksteps <= 4) { if (addend) EGNN_TAIL_TILE(4, true); else EGNN_TAIL_TILE(4, false); }
    else { if (addend) EGNN_TAIL_TILE(10, true); else EGNN_TAIL_TILE(10, false); }
#undef EGNN_TAIL_TILE
  } else {
    // other widths: the MFMA-layout kernel, 16-row tiles, three workgroups per CU (the best of the shapes measured: 42 / 22 / 23 / 12 / 13)
    const int64_t items = sbs * (C / 64);
    const unsigned grid = (unsigned)((items + 3) / 4);
#define EGNN_DX_BN(KS, ADD)                                                                                                                  \
  hipLaunchKernelGGL((skinny_dx_bn_kernel<KS, 1, 3, ADD>), dim3(grid), dim3(256), 0, st, G, ldg, W, ldw, b_kmajor, (int)Ks, alpha,              \
                     kDxBnRows / 16, addend, ld_addend, add_rows, ld_add_rows, add_inv, q, dx, ld_dx, ws)
    if (ksteps <= 4) { if (addend) EGNN_DX_BN(4, true); else EGNN_DX_BN(4, false); }
    else if (ksteps <= 10) { if (addend) EGNN_DX_BN(10, true); else EGNN_DX_BN(10, false); }
    else { if (addend) EGNN_DX_BN(16, true); else EGNN_DX_BN(16, false); }
#undef EGNN_DX_BN
  }
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((unsigned)((C + kMergeCols - 1) / kMergeCols)), dim3(256), 0, st, ws, (int)sbs, C, dbeta, dgamma);
  return egnn_launch_status();
}

// dx <- gamma rstd (d - (sum_dbeta + xhat sum_dgamma) * inv_count), in place over the d that the reduce half left in dx
extern "C" int egnn_bn_bwd_apply_stored_f32(const float* x, int64_t ldx, int64_t M, int64_t C, const float* mean, const float* var, float eps,
                                            const float* gamma, const float* beta, int relu, float p, uint64_t seed, const uint64_t* seed_dev,
                                            const float* sum_dbeta, const float* sum_dgamma, float inv_count, float* dx, int64_t ld_dx,
                                            float* dx_colsum, float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(M > 0 && x && mean && var && sum_dbeta && sum_dgamma && dx && ldx >= C && ld_dx >= C);
  if (!shape_ok(x, ldx, C) || !shape_ok(dx, ld_dx, C)) return EGNN_EALIGN;
  if (dx_colsum && (!ws || ws_floats < egnn_bn_ws_floats(C))) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const BnParams q{x, ldx, M, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev, nullptr};
  int nb = row_blocks(M);
  if (dx_colsum) {
    if (nb > 2 * kStatBlocks) nb = 2 * kStatBlocks;
    nb = (nb + 1) & ~1;
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<true, 3>), dim3(nb), dim3(256), 0, st, q, (const float*)dx, ld_dx, sum_dbeta, sum_dgamma, inv_count,
                       dx, ld_dx, ws);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((C + kMergeCols - 1) / kMergeCols)), dim3(256), 0, st, ws, nb / 2, C, dx_colsum,
                       (const float*)nullptr, (const float*)nullptr, 0.f, (const float*)nullptr);
  } else {
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<false, 3>), dim3(nb), dim3(256), 0, st, q, (const float*)dx, ld_dx, sum_dbeta, sum_dgamma, inv_count,
                       dx, ld_dx, (float*)nullptr);
  }
  return egnn_launch_status();
}

extern "C" int egnn_skinny_dx_bn_bwd_f32(const float* G, int64_t ldg, const float* W, int64_t ldw, int w_kmajor, int64_t M, int64_t C,
                                         int64_t Ks, float alpha, const float* addend, int64_t ld_addend, const float* add_rows,
                                         int64_t ld_add_rows, const int32_t* add_inv, const float* x, int64_t ldx, const float* mean,
                                         const float* var, float eps, const float* gamma, const float* beta, int relu, float p,
                                         uint64_t seed, const uint64_t* seed_dev, int batch_stats, float* dgamma, float* dbeta, float* dx,
                                         int64_t ld_dx, float* dx_colsum, float* ws, size_t ws_floats, void* stream) {
  const int rc = egnn_skinny_dx_bn_bwd_reduce_f32(G, ldg, W, ldw, w_kmajor, M, C, Ks, alpha, addend, ld_addend, add_rows, ld_add_rows, add_inv, x,
                                                  ldx, mean, var, eps, gamma, beta, relu, p, seed, seed_dev, dgamma, dbeta, dx, ld_dx, ws,
                                                  ws_floats, stream);
  if (rc != EGNN_OK) return rc;
  return egnn_bn_bwd_apply_stored_f32(x, ldx, M, C, mean, var, eps, gamma, beta, relu, p, seed, seed_dev, dbeta, dgamma,
                                      batch_stats ? 1.f / (float)M : 0.f, dx, ld_dx, dx_colsum, ws, ws_floats, stream);
}

namespace {
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ W, int64_t ldw, int64_t rows, int64_t C,
                                                      const float* __restrict__ bias, const float* __restrict__ mean,
                                                      const float* __restrict__ var, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, float* __restrict__ Wo, int64_t ldo,
                                                      float* __restrict__ bo) {
  const int64_t total = rows * C;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / C, c = t % C;
    const float s = (gamma ? gamma[c] : 1.f) * rsqrtf(var[c] + eps);
    Wo[r * ldo + c] = W[r * ldw + c] * s;
    if (r == 0) bo[c] = ((bias ? bias[c] : 0.f) - mean[c]) * s + (beta ? beta[c] : 0.f);
  }
}
}  // namespace

extern "C" int egnn_bn_fold_f32(const float* W, int64_t ldw, int64_t rows, int64_t C, const float* bias, const float* mean, const float* var,
                                const float* gamma, const float* beta, float eps, float* W_out, int64_t ld_out, float* bias_out,
                                void* stream) {
  EGNN_CHECK_ARG(rows > 0 && C > 0 && W && mean && var && W_out && bias_out && ldw >= C && ld_out >= C);
  const int64_t blocks = (rows * C + 255) / 256;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream, W, ldw, rows, C, bias,
                     mean, var, gamma, beta, eps, W_out, ld_out, bias_out);
  return egnn_launch_status();
}
