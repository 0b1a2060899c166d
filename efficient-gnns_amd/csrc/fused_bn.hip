// Fused BatchNorm1d (+ ReLU + dropout) over node rows for gfx950 -- SURVEY.md 8(f) rank 1 (K8).
// Replaces the ATen BatchNorm / threshold / fused_dropout kernel chain between every pair of convs
// (/root/reference/arxiv_pyg/gnn.py:48-50,80-82) and inside the projection heads (:296-306):
//   forward : column statistics (1 read of x) + one elementwise pass  y = drop(relu(g * xhat + b))
//   backward: column reductions (sum d, sum d*xhat) + one elementwise pass; xhat, the ReLU sign and the dropout
//             mask are RECOMPUTED from x and the 64-bit seed, so nothing but x is kept for the backward.
// HBM-bound: every wave streams whole rows (64 lanes x float4 = 256 columns) with coalesced 1 KiB accesses;
// column partials are merged in a fixed order (deterministic).
#include "common.h"

namespace {

constexpr int kMaxChunks = 4;     // columns are processed in chunks of 256 (64 lanes x float4): C <= 1024
constexpr int kStatBlocks = 512;   // partial-sum rows (2 wave-sets per CU keep the stream saturated)

struct BnParams {
  const float* x; int64_t ldx;
  int64_t n, C;
  const float* mean; const float* var; float eps;
  const float* gamma; const float* beta;
  int relu; float p; unsigned long long seed;
  const unsigned long long* seed_dev;   // nullable: added to `seed` (a per-step value kept on the device: hipGraph replays)
};

// counter-based uniform in [0,1): splitmix64 of (seed + element index)
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);
}

// per-element forward pieces shared by forward and backward
__device__ __forceinline__ void bn_elem(const BnParams& q, float x, float mean, float rstd, float g, float b, int64_t row,
                                        int64_t c, float& xhat, float& gate) {
  xhat = (x - mean) * rstd;
  const float pre = g * xhat + b;
  gate = (q.relu && !(pre > 0.f)) ? 0.f : 1.f;
  if (q.p > 0.f) {
    const float u = uniform01(q.seed + (q.seed_dev ? *q.seed_dev : 0ull), (unsigned long long)(row * q.C + c));
    gate = u >= q.p ? gate / (1.f - q.p) : 0.f;
  }
}

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
      const float4 v = *reinterpret_cast<const float4*>(q.x + row * q.ldx + c);
      const float xv[4] = {v.x, v.y, v.z, v.w};
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float xhat, gate;
        bn_elem(q, xv[k], mean[k], rstd[k], g[k], b[k], row, c + k, xhat, gate);
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
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u] = *reinterpret_cast<const float4*>(q.x + (row + u * stride) * q.ldx + c);
          gd[u] = *reinterpret_cast<const float4*>(dy + (row + u * stride) * ldd + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float xv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          const float gv[4] = {gd[u].x, gd[u].y, gd[u].z, gd[u].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float xhat, gate;
            bn_elem(q, xv[k], mean[k], rstd[k], g[k], b[k], row + u * stride, c + k, xhat, gate);
            const float d = gv[k] * gate;
            sd[k] += d;
            sx[k] = fmaf(d, xhat, sx[k]);
          }
        }
      }
      for (; row < q.n; row += stride) {
        const float4 v = *reinterpret_cast<const float4*>(q.x + row * q.ldx + c);
        const float4 gd = *reinterpret_cast<const float4*>(dy + row * ldd + c);
        const float xv[4] = {v.x, v.y, v.z, v.w};
        const float gv[4] = {gd.x, gd.y, gd.z, gd.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float xhat, gate;
          bn_elem(q, xv[k], mean[k], rstd[k], g[k], b[k], row, c + k, xhat, gate);
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
template <bool COLSUM>
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
      const float4 v = *reinterpret_cast<const float4*>(q.x + row * q.ldx + c);
      const float4 gd = *reinterpret_cast<const float4*>(dy + row * ldd + c);
      const float xv[4] = {v.x, v.y, v.z, v.w};
      const float gv[4] = {gd.x, gd.y, gd.z, gd.w};
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float xhat, gate;
        bn_elem(q, xv[k], mean[k], rstd[k], g[k], b[k], row, c + k, xhat, gate);
        const float d = gv[k] * gate;
        o[k] = g[k] * rstd[k] * (d - sb[k] - xhat * sg[k]);
        if constexpr (COLSUM) so[k] += o[k];
      }
      *reinterpret_cast<float4*>(dx + row * ldx_out + c) = make_float4(o[0], o[1], o[2], o[3]);
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

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nblocks, int64_t C, float* __restrict__ out) {
  float a, b; bool owner; int64_t c;
  merge_partials(part, nblocks, C, a, b, owner, c);   // nblocks = PAIRS of partial rows
  if (owner) out[c] = a + b;
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
  const BnParams q{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev};
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
  const BnParams q{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev};
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
  const BnParams q{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev};
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel<false>, dim3(row_blocks(n)), dim3(256), 0, (hipStream_t)stream, q, dy, ld_dy, sum_dbeta,
                     sum_dgamma, inv_count, dx, ld_dx, nullptr);
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
  const BnParams q{x, ld, n, C, mean, var, eps, gamma, beta, relu, p, (unsigned long long)seed, (const unsigned long long*)seed_dev};
  int nb = row_blocks(n);
  if (nb > 2 * kStatBlocks) nb = 2 * kStatBlocks;   // the workspace holds 2 kStatBlocks partial rows (the reduce half is done with it: same stream)
  nb = (nb + 1) & ~1;                               // whole pairs; a block past the rows writes zeros
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel<true>, dim3(nb), dim3(256), 0, st, q, dy, ld_dy, dbeta, dgamma, inv_count, dx, ld_dx, ws);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((C + kMergeCols - 1) / kMergeCols)), dim3(256), 0, st, ws, nb / 2, C, dx_colsum);
  return egnn_launch_status();
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
