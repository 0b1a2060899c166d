// LSP building blocks (/root/reference/arxiv_pyg/criterion.py:95-126) for gfx950: per-edge similarity of
// gathered feature rows and the segment softmax of torch_geometric.utils.softmax (criterion.py:5,103-113).
//
// The reference materialises feat[src], feat[dst] ([E,D] twice per feature matrix) and uses atomic
// scatter_max / scatter_add for the softmax (K5/K6 in SURVEY.md 2.2).  Here an edge's two rows are read
// straight into registers by one wavefront (never written back), and segments are CSR rows, so the softmax
// needs no atomics and has a fixed summation order.  HBM/L2-bound gather work: no MFMA.
#include "common.h"

namespace {

enum { K_COSINE = 0, K_POLY = 1, K_L2 = 2, K_RBF = 3 };
constexpr float kCosEps = 1e-8f;

// one wavefront per edge: aux = (dot, |a|^2, |b|^2) for cosine/poly, (d2, 0, 0) for l2/rbf
template <bool VEC4>
__global__ __launch_bounds__(256) void edge_sim_kernel(const float* __restrict__ F, int64_t ld, int64_t D,
                                                       const int64_t* __restrict__ ia, const int64_t* __restrict__ ib,
                                                       int64_t E, int kernel, float* __restrict__ sim,
                                                       float* __restrict__ aux) {
  const int lane = egnn_lane();
  for (int64_t e = blockIdx.x * 4LL + egnn_wave_id(); e < E; e += (int64_t)gridDim.x * 4) {
    const float* a = F + ia[e] * ld;
    const float* b = F + ib[e] * ld;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if constexpr (VEC4) {
      // 16-byte aligned rows (ld % 4 == 0): whole float4 chunks first, then the D % 4 trailing columns one lane each (the teacher's
      // 750 columns behind a 752 pitch: the scalar walk of the whole row was 511 us of the LSP step)
      const int64_t Dv = D & ~(int64_t)3;
      if (lane < D - Dv) {
        const float x = a[Dv + lane], y = b[Dv + lane];
        if (kernel <= K_POLY) { s0 = x * y; s1 = x * x; s2 = y * y; }
        else { const float t = x - y; s0 = t * t; }
      }
      for (int64_t d = lane * 4; d < Dv; d += 256) {
        const float4 x = *reinterpret_cast<const float4*>(a + d);
        const float4 y = *reinterpret_cast<const float4*>(b + d);
        if (kernel <= K_POLY) {
          s0 += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
          s1 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
          s2 += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
        } else {
          const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
          s0 += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
      }
    } else {
      for (int64_t d = lane; d < D; d += 64) {
        const float x = a[d], y = b[d];
        if (kernel <= K_POLY) { s0 = fmaf(x, y, s0); s1 = fmaf(x, x, s1); s2 = fmaf(y, y, s2); }
        else { const float t = x - y; s0 = fmaf(t, t, s0); }
      }
    }
    s0 = egnn_wave_sum(s0);
    if (kernel <= K_POLY) { s1 = egnn_wave_sum(s1); s2 = egnn_wave_sum(s2); }
    if (lane == 0) {
      float v;
      if (kernel <= K_POLY) {
        // torch 1.7 F.cosine_similarity: dot / sqrt(max(|a|^2 |b|^2, eps^2))   
        const float c = s0 / sqrtf(fmaxf(s1 * s2, kCosEps * kCosEps));
        v = kernel == K_COSINE ? c : c * c;
      } else {
        v = kernel == K_L2 ? sqrtf(s0) : expf(-0.5f * s0);
      }
      sim[e] = v;
      aux[3 * e] = s0; aux[3 * e + 1] = s1; aux[3 * e + 2] = s2;
    }
  }
}

// d(sim)/d(a) = alpha * b + beta_a * a ; d(sim)/d(b) = alpha * a + beta_b * b   (times the upstream g[e])
__global__ __launch_bounds__(256) void edge_coef_kernel(const float* __restrict__ g, const float* __restrict__ sim,
                                                        const float* __restrict__ aux, int64_t E, int kernel,
                                                        float* __restrict__ alpha, float* __restrict__ beta_a,
                                                        float* __restrict__ beta_b) {
  for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
    const float ge = g[e];
    float al, ba, bb;
    if (kernel <= K_POLY) {
      const float dot = aux[3 * e], na2 = aux[3 * e + 1], nb2 = aux[3 * e + 2];
      const float den = sqrtf(fmaxf(na2 * nb2, kCosEps * kCosEps));
      const float c = dot / den;
      const float gg = kernel == K_COSINE ? ge : ge * 2.f * c;
      const bool clamped = na2 * nb2 <= kCosEps * kCosEps;
      al = gg / den;
      ba = clamped ? 0.f : -gg * c / na2;
      bb = clamped ? 0.f : -gg * c / nb2;
    } else if (kernel == K_L2) {
      const float d = sim[e];
      const float q = d > 0.f ? ge / d : 0.f;
      al = -q; ba = q; bb = q;
    } else {
      const float q = ge * sim[e];
      al = q; ba = -q; bb = -q;
    }
    alpha[e] = al; beta_a[e] = ba; beta_b[e] = bb;
  }
}

__global__ __launch_bounds__(256) void seg_softmax_fwd_kernel(const int64_t* __restrict__ ptr, const float* __restrict__ x,
                                                              int64_t n_seg, float* __restrict__ p) {
  const int lane = egnn_lane();
  for (int64_t s = blockIdx.x * 4LL + egnn_wave_id(); s < n_seg; s += (int64_t)gridDim.x * 4) {
    const int64_t b = ptr[s], e = ptr[s + 1];
    if (b == e) continue;
    float m = -INFINITY;
    for (int64_t i = b + lane; i < e; i += 64) m = fmaxf(m, x[i]);
    m = egnn_wave_max(m);
    float z = 0.f;
    for (int64_t i = b + lane; i < e; i += 64) z += expf(x[i] - m);
    z = egnn_wave_sum(z);
    const float den = z + 1e-16f;  // PyG: out / (sum + 1e-16) -- a DIVISION per entry as in the reference (one rounding; e * (1 / den) has two,
                                   // which shows in the KL of two nearly uniform distributions, a difference of O(1) sums)
    for (int64_t i = b + lane; i < e; i += 64) p[i] = expf(x[i] - m) / den;
  }
}

// gx = p * (gp - sum_seg p * gp)
__global__ __launch_bounds__(256) void seg_softmax_bwd_kernel(const int64_t* __restrict__ ptr, const float* __restrict__ p,
                                                              const float* __restrict__ gp, int64_t n_seg,
                                                              float* __restrict__ gx) {
  const int lane = egnn_lane();
  for (int64_t s = blockIdx.x * 4LL + egnn_wave_id(); s < n_seg; s += (int64_t)gridDim.x * 4) {
    const int64_t b = ptr[s], e = ptr[s + 1];
    float d = 0.f;
    for (int64_t i = b + lane; i < e; i += 64) d = fmaf(p[i], gp[i], d);
    d = egnn_wave_sum(d);
    for (int64_t i = b + lane; i < e; i += 64) gx[i] = p[i] * (gp[i] - d);
  }
}

__global__ __launch_bounds__(256) void seg_sum_kernel(const int64_t* __restrict__ ptr, const float* __restrict__ x,
                                                      int64_t n_seg, float* __restrict__ out) {
  // four segments per wave (16-lane groups); a segment longer than 256 entries is walked by the whole wave afterwards
  const int lane = egnn_lane();
  const int grp = lane >> 4, sl = lane & 15;
  for (int64_t s0 = (blockIdx.x * 4LL + egnn_wave_id()) * 4; s0 < n_seg; s0 += (int64_t)gridDim.x * 16) {
    const int64_t s = s0 + grp;
    const int64_t b = s < n_seg ? ptr[s] : 0, e = s < n_seg ? ptr[s + 1] : 0;
    const bool small = e - b <= 256;
    float d = 0.f;
    if (small) for (int64_t i = b + sl; i < e; i += 16) d += x[i];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if (small && sl == 0 && s < n_seg) out[s] = d;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t bg = __shfl(b, k * 16), eg = __shfl(e, k * 16);
      if (eg - bg > 256) {
        float w = 0.f;
        for (int64_t i = bg + lane; i < eg; i += 64) w += x[i];
        w = egnn_wave_sum(w);
        if (lane == 0) out[s0 + k] = w;
      }
    }
  }
}

// ---- LSP criterion tail (criterion.py:103-122) as ONE pass per direction ----------------------------------------------------
// loss = mean_e term(p_s[e], p_t[e]) with p_* = segment softmax of the similarity vectors (PyG softmax: / (sum + 1e-16)):
//   KLD: F.kl_div(log p_s, p_t, reduction='mean')  ->  term = p_t (log p_t - log p_s), 0 where p_t == 0
//   MSE: F.mse_loss(p_s, p_t)                      ->  term = (p_s - p_t)^2
// Both softmaxes, the element-wise term and its sum are formed by the wave that owns the segment; the sum over segments is a fixed
// grid-stride assignment + a fixed-order finalize (deterministic, and no torch reduction -- whose multi-block form zeroes its
// semaphores with a memset node that was seen NOT to take effect in hipGraph replays on this stack, profiles/r04_lsp_trace.txt).
constexpr int kLspBlocks = 1024;

// Segments are short on average (7.5 edges per train node) with a few hubs: a wave takes FOUR consecutive segments, one per 16-lane
// group (strided walk, 16-lane butterflies); a segment longer than kSubMax is then walked by the whole wave.  Fixed assignment and
// fixed reduction order: deterministic.
constexpr int kSubMax = 256;

__device__ __forceinline__ float sub16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float sub16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// (max, sum exp + 1e-16) of x[b, e) walked by lanes first, first + step, ... and reduced over the walking group
template <bool WAVE>
__device__ __forceinline__ void seg_softmax_stats(const float* __restrict__ x, int64_t b, int64_t e, int first, int step, float& m, float& den) {
  m = -INFINITY;
  for (int64_t i = b + first; i < e; i += step) m = fmaxf(m, x[i]);
  m = WAVE ? egnn_wave_max(m) : sub16_max(m);
  float z = 0.f;
  for (int64_t i = b + first; i < e; i += step) z += expf(x[i] - m);
  den = (WAVE ? egnn_wave_sum(z) : sub16_sum(z)) + 1e-16f;
}

template <bool WAVE>
__device__ __forceinline__ float lsp_fwd_segment(const float* __restrict__ xs, const float* __restrict__ xt, int64_t b, int64_t e, int first,
                                                 int step, int mse, float* __restrict__ ps, float* __restrict__ pt) {
  float ms, ds, mt, dt, acc = 0.f;
  seg_softmax_stats<WAVE>(xs, b, e, first, step, ms, ds);
  seg_softmax_stats<WAVE>(xt, b, e, first, step, mt, dt);
  for (int64_t i = b + first; i < e; i += step) {
    const float a = expf(xs[i] - ms) / ds;   // a DIVISION per entry as in the reference (see seg_softmax_fwd_kernel)
    const float t = expf(xt[i] - mt) / dt;
    ps[i] = a;
    pt[i] = t;
    if (mse) { const float d = a - t; acc = fmaf(d, d, acc); }
    else acc += t > 0.f ? t * (logf(t) - logf(a)) : 0.f;
  }
  return acc;
}

__global__ __launch_bounds__(256) void lsp_loss_fwd_kernel(const int64_t* __restrict__ ptr, const float* __restrict__ xs,
                                                           const float* __restrict__ xt, int64_t n_seg, int mse,
                                                           float* __restrict__ ps, float* __restrict__ pt,
                                                           float* __restrict__ partials) {
  __shared__ float s_part[4];
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int grp = lane >> 4, sl = lane & 15;
  float acc = 0.f;
  for (int64_t s0 = (blockIdx.x * 4LL + wave) * 4; s0 < n_seg; s0 += (int64_t)gridDim.x * 16) {
    const int64_t s = s0 + grp;
    const int64_t b = s < n_seg ? ptr[s] : 0, e = s < n_seg ? ptr[s + 1] : 0;
    if (e - b <= kSubMax) acc += lsp_fwd_segment<false>(xs, xt, b, e, sl, 16, mse, ps, pt);   // (an empty range walks nothing)
#pragma unroll
    for (int g = 0; g < 4; ++g) {   // the hubs among the four: wave-uniform bounds, the whole wave walks
      const int64_t bg = __shfl(b, g * 16), eg = __shfl(e, g * 16);
      if (eg - bg > kSubMax) acc += lsp_fwd_segment<true>(xs, xt, bg, eg, lane, 64, mse, ps, pt);
    }
  }
  acc = egnn_wave_sum(acc);
  if (lane == 0) s_part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

__global__ __launch_bounds__(256) void lsp_loss_final_kernel(const float* __restrict__ partials, int nblocks, float inv_count,
                                                             float* __restrict__ loss) {
  __shared__ float s[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) a += partials[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = s[0] * inv_count;
}

// d loss / d sim_s (and, when asked for, d loss / d sim_t) through the two softmaxes: gx = p * (gp - sum_seg p * gp) with
//   KLD: gp_s = -g p_t / (E p_s)          gp_t = g (log p_t - log p_s + 1) / E   (0 where p_t == 0)
//   MSE: gp_s = 2 g (p_s - p_t) / E       gp_t = -gp_s
template <bool WAVE>
__device__ __forceinline__ void lsp_bwd_segment(const float* __restrict__ ps, const float* __restrict__ pt, int64_t b, int64_t e, int first,
                                                int step, int mse, float gs, float* __restrict__ gxs, float* __restrict__ gxt) {
  float d_s = 0.f, d_t = 0.f;   // sum_seg p * gp for the student / teacher side
  for (int64_t i = b + first; i < e; i += step) {
    const float a = ps[i], t = pt[i];
    if (mse) {
      const float gp = 2.f * gs * (a - t);
      d_s = fmaf(a, gp, d_s);
      d_t = fmaf(t, -gp, d_t);
    } else {
      d_s -= gs * t;                                       // p_s * (-g p_t / (E p_s))
      if (gxt != nullptr && t > 0.f) d_t = fmaf(t, gs * (logf(t) - logf(a) + 1.f), d_t);
    }
  }
  d_s = WAVE ? egnn_wave_sum(d_s) : sub16_sum(d_s);
  if (gxt != nullptr) d_t = WAVE ? egnn_wave_sum(d_t) : sub16_sum(d_t);
  for (int64_t i = b + first; i < e; i += step) {
    const float a = ps[i], t = pt[i];
    if (mse) {
      const float gp = 2.f * gs * (a - t);
      gxs[i] = a * (gp - d_s);
      if (gxt != nullptr) gxt[i] = t * (-gp - d_t);
    } else {
      gxs[i] = -gs * t - a * d_s;
      if (gxt != nullptr) gxt[i] = t > 0.f ? t * (gs * (logf(t) - logf(a) + 1.f) - d_t) : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void lsp_loss_bwd_kernel(const int64_t* __restrict__ ptr, const float* __restrict__ ps,
                                                           const float* __restrict__ pt, int64_t n_seg, int mse,
                                                           const float* __restrict__ g, float inv_count,
                                                           float* __restrict__ gxs, float* __restrict__ gxt) {
  const int lane = egnn_lane();
  const int grp = lane >> 4, sl = lane & 15;
  const float gs = g[0] * inv_count;
  for (int64_t s0 = (blockIdx.x * 4LL + egnn_wave_id()) * 4; s0 < n_seg; s0 += (int64_t)gridDim.x * 16) {
    const int64_t s = s0 + grp;
    const int64_t b = s < n_seg ? ptr[s] : 0, e = s < n_seg ? ptr[s + 1] : 0;
    if (e - b <= kSubMax) lsp_bwd_segment<false>(ps, pt, b, e, sl, 16, mse, gs, gxs, gxt);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t bg = __shfl(b, k * 16), eg = __shfl(e, k * 16);
      if (eg - bg > kSubMax) lsp_bwd_segment<true>(ps, pt, bg, eg, lane, 64, mse, gs, gxs, gxt);
    }
  }
}

// Column sums of a [n, C] matrix: fixed row stripes per block, fixed-order finalize (bias gradients; no torch reduction, see above).
// HBM-bound streaming read: 256 threads = RPB row lanes x CW column lanes (CW = the power of two >= C, <= 256; wider matrices walk
// column chunks), four independent row streams per thread, one partial row per block; the finalize gives one wave per column.
constexpr int kColsumBlocks = 1024;

template <int CW>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t n, int64_t C,
                                                             float* __restrict__ partials) {
  constexpr int RPB = 256 / CW;
  const int tc = threadIdx.x % CW, tr = threadIdx.x / CW;
  __shared__ float s[256];
  const int64_t step = (int64_t)gridDim.x * RPB;
  for (int64_t c0 = 0; c0 < C; c0 += CW) {
    const int64_t c = c0 + tc;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
      const float* col = x + c;
      int64_t r = (int64_t)blockIdx.x * RPB + tr;
      for (; r + 3 * step < n; r += 4 * step) {
        a0 += col[r * ld];
        a1 += col[(r + step) * ld];
        a2 += col[(r + 2 * step) * ld];
        a3 += col[(r + 3 * step) * ld];
      }
      for (; r < n; r += step) a0 += col[r * ld];
    }
    if constexpr (RPB > 1) {
      s[threadIdx.x] = (a0 + a1) + (a2 + a3);
      __syncthreads();
      if (tr == 0 && c < C) {
        float t = s[tc];
#pragma unroll
        for (int k = 1; k < RPB; ++k) t += s[k * CW + tc];
        partials[(int64_t)blockIdx.x * C + c] = t;
      }
      __syncthreads();
    } else {
      if (c < C) partials[(int64_t)blockIdx.x * C + c] = (a0 + a1) + (a2 + a3);
    }
  }
}

__global__ __launch_bounds__(256) void colsum_wave_final_kernel(const float* __restrict__ partials, int nblocks, int64_t C,
                                                                float* __restrict__ out) {
  const int lane = egnn_lane();
  const int64_t c = (int64_t)blockIdx.x * 4 + egnn_wave_id();
  if (c >= C) return;
  float t = 0.f;
  for (int b = lane; b < nblocks; b += 64) t += partials[(int64_t)b * C + c];
  t = egnn_wave_sum(t);
  if (lane == 0) out[c] = t;
}

inline unsigned wave_grid(int64_t n) {
  const int64_t b = (n + 3) / 4;
  return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// GAT attention coefficients of one target row per wavefront (PyG <=1.7 GATConv.message + utils.softmax):
//   s_e,h = leaky_relu(alpha_src[col[e],h] + alpha_dst[i,h]);  att[h,e] = exp(s - max_e s) / (sum_e exp(s - max) + 1e-16)
// The [E,H] score tensor, its gathers (u_add_v SDDMM) and the scatter-softmax never exist in memory; fixed order.
__global__ __launch_bounds__(256) void gat_attention_fwd_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                                const float* __restrict__ asrc, const float* __restrict__ adst,
                                                                int64_t n_rows, int64_t nnz, int H, float slope,
                                                                float* __restrict__ att) {
  const int lane = egnn_lane();
  const int64_t row = (int64_t)blockIdx.x * 4 + egnn_wave_id();
  if (row >= n_rows) return;
  const int64_t start = rowptr[row], end = rowptr[row + 1];
  for (int h = 0; h < H; ++h) {
    const float ad = adst[row * H + h];
    float m = -INFINITY;
    for (int64_t e = start + lane; e < end; e += 64) {
      float s = asrc[col[e] * H + h] + ad;
      s = s > 0.f ? s : s * slope;
      m = fmaxf(m, s);
    }
    m = egnn_wave_max(m);
    float z = 0.f;
    for (int64_t e = start + lane; e < end; e += 64) {
      float s = asrc[col[e] * H + h] + ad;
      s = s > 0.f ? s : s * slope;
      z += expf(s - m);
    }
    z = egnn_wave_sum(z) + 1e-16f;
    for (int64_t e = start + lane; e < end; e += 64) {
      float s = asrc[col[e] * H + h] + ad;
      s = s > 0.f ? s : s * slope;
      att[(int64_t)h * nnz + e] = expf(s - m) / z;
    }
  }
}

}  // namespace

extern "C" int egnn_gat_attention_fwd_f32(const int64_t* rowptr, const int64_t* col, const float* alpha_src, const float* alpha_dst,
                                          int64_t n_rows, int64_t nnz, int H, float negative_slope, float* att, void* stream) {
  EGNN_CHECK_ARG(n_rows >= 0 && nnz >= 0 && H > 0 && H <= 64);
  if (n_rows == 0 || nnz == 0) return EGNN_OK;
  EGNN_CHECK_ARG(rowptr && col && alpha_src && alpha_dst && att);
  const int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 0x7fffffffLL) return EGNN_EINVAL;
  hipLaunchKernelGGL(gat_attention_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, alpha_src,
                     alpha_dst, n_rows, nnz, H, negative_slope, att);
  return egnn_launch_status();
}

extern "C" int egnn_edge_sim_f32(const float* F, int64_t ld, int64_t D, const int64_t* idx_a, const int64_t* idx_b, int64_t E,
                                 int kernel, float* sim, float* aux3, void* stream) {
  EGNN_CHECK_ARG(E >= 0 && D > 0 && ld >= D && kernel >= 0 && kernel <= 3);
  if (E == 0) return EGNN_OK;
  EGNN_CHECK_ARG(F && idx_a && idx_b && sim && aux3);
  const bool vec4 = (ld % 4 == 0) && egnn_aligned16(F) && D >= 4;
  hipStream_t st = (hipStream_t)stream;
  if (vec4) hipLaunchKernelGGL(edge_sim_kernel<true>, dim3(wave_grid(E)), dim3(256), 0, st, F, ld, D, idx_a, idx_b, E, kernel, sim, aux3);
  else hipLaunchKernelGGL(edge_sim_kernel<false>, dim3(wave_grid(E)), dim3(256), 0, st, F, ld, D, idx_a, idx_b, E, kernel, sim, aux3);
  return egnn_launch_status();
}

extern "C" int egnn_edge_sim_coef_f32(const float* g, const float* sim, const float* aux3, int64_t E, int kernel,
                                      float* alpha, float* beta_a, float* beta_b, void* stream) {
  EGNN_CHECK_ARG(E >= 0 && kernel >= 0 && kernel <= 3);
  if (E == 0) return EGNN_OK;
  EGNN_CHECK_ARG(g && sim && aux3 && alpha && beta_a && beta_b);
  const int64_t blocks = (E + 255) / 256;
  hipLaunchKernelGGL(edge_coef_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, g, sim,
                     aux3, E, kernel, alpha, beta_a, beta_b);
  return egnn_launch_status();
}

extern "C" int egnn_segment_softmax_fwd_f32(const int64_t* seg_ptr, const float* x, int64_t n_seg, float* p, void* stream) {
  EGNN_CHECK_ARG(n_seg >= 0);
  if (n_seg == 0) return EGNN_OK;
  EGNN_CHECK_ARG(seg_ptr && x && p);
  hipLaunchKernelGGL(seg_softmax_fwd_kernel, dim3(wave_grid(n_seg)), dim3(256), 0, (hipStream_t)stream, seg_ptr, x, n_seg, p);
  return egnn_launch_status();
}

extern "C" int egnn_segment_softmax_bwd_f32(const int64_t* seg_ptr, const float* p, const float* gp, int64_t n_seg, float* gx,
                                            void* stream) {
  EGNN_CHECK_ARG(n_seg >= 0);
  if (n_seg == 0) return EGNN_OK;
  EGNN_CHECK_ARG(seg_ptr && p && gp && gx);
  hipLaunchKernelGGL(seg_softmax_bwd_kernel, dim3(wave_grid(n_seg)), dim3(256), 0, (hipStream_t)stream, seg_ptr, p, gp, n_seg, gx);
  return egnn_launch_status();
}

extern "C" int egnn_segment_sum_f32(const int64_t* seg_ptr, const float* x, int64_t n_seg, float* out, void* stream) {
  EGNN_CHECK_ARG(n_seg >= 0);
  if (n_seg == 0) return EGNN_OK;
  EGNN_CHECK_ARG(seg_ptr && x && out);
  hipLaunchKernelGGL(seg_sum_kernel, dim3(wave_grid((n_seg + 3) / 4)), dim3(256), 0, (hipStream_t)stream, seg_ptr, x, n_seg, out);
  return egnn_launch_status();
}

extern "C" size_t egnn_lsp_loss_ws_floats(void) { return (size_t)kLspBlocks; }

extern "C" int egnn_lsp_loss_fwd_f32(const int64_t* seg_ptr, const float* sim_s, const float* sim_t, int64_t n_seg, int64_t E,
                                     int criterion, float* p_s, float* p_t, float* loss, float* ws, void* stream) {
  EGNN_CHECK_ARG(n_seg >= 0 && E > 0 && (criterion == 0 || criterion == 1));
  EGNN_CHECK_ARG(seg_ptr && sim_s && sim_t && p_s && p_t && loss && ws && n_seg > 0);
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = (n_seg + 15) / 16;   // 16 segments per workgroup pass
  const int nb = (int)(want < kLspBlocks ? want : kLspBlocks);
  hipLaunchKernelGGL(lsp_loss_fwd_kernel, dim3(nb), dim3(256), 0, st, seg_ptr, sim_s, sim_t, n_seg, criterion, p_s, p_t, ws);
  hipLaunchKernelGGL(lsp_loss_final_kernel, dim3(1), dim3(256), 0, st, ws, nb, 1.f / (float)E, loss);
  return egnn_launch_status();
}

extern "C" int egnn_lsp_loss_bwd_f32(const int64_t* seg_ptr, const float* p_s, const float* p_t, int64_t n_seg, int64_t E,
                                     int criterion, const float* g, float* gsim_s, float* gsim_t, void* stream) {
  EGNN_CHECK_ARG(n_seg > 0 && E > 0 && (criterion == 0 || criterion == 1));
  EGNN_CHECK_ARG(seg_ptr && p_s && p_t && g && gsim_s);
  hipLaunchKernelGGL(lsp_loss_bwd_kernel, dim3(wave_grid((n_seg + 3) / 4)), dim3(256), 0, (hipStream_t)stream, seg_ptr, p_s, p_t, n_seg,
                     criterion, g, 1.f / (float)E, gsim_s, gsim_t);
  return egnn_launch_status();
}

extern "C" size_t egnn_colsum_ws_floats(int64_t C) { return (size_t)kColsumBlocks * (size_t)(C > 0 ? C : 0); }

extern "C" int egnn_colsum_f32(const float* x, int64_t ld, int64_t n, int64_t C, float* out, float* ws, void* stream) {
  EGNN_CHECK_ARG(n > 0 && C > 0 && ld >= C && x && out && ws);
  hipStream_t st = (hipStream_t)stream;
  const int cw = C > 128 ? 256 : (C > 64 ? 128 : (C > 32 ? 64 : 32));
  const int rpb = 256 / cw;
  const int64_t want = (n + 4 * rpb - 1) / (4 * rpb);   // >= 4 rows per row lane before another block is worth its partial row
  const int nb = (int)(want < 1 ? 1 : (want < kColsumBlocks ? want : kColsumBlocks));
  switch (cw) {
    case 256: hipLaunchKernelGGL(colsum_partial_kernel<256>, dim3(nb), dim3(256), 0, st, x, ld, n, C, ws); break;
    case 128: hipLaunchKernelGGL(colsum_partial_kernel<128>, dim3(nb), dim3(256), 0, st, x, ld, n, C, ws); break;
    case 64: hipLaunchKernelGGL(colsum_partial_kernel<64>, dim3(nb), dim3(256), 0, st, x, ld, n, C, ws); break;
    default: hipLaunchKernelGGL(colsum_partial_kernel<32>, dim3(nb), dim3(256), 0, st, x, ld, n, C, ws); break;
  }
  hipLaunchKernelGGL(colsum_wave_final_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, st, ws, nb, C, out);
  return egnn_launch_status();
}
