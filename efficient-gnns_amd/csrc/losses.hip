// Row-wise fused losses for gfx950 (HBM/latency-bound; wave64 per row, fixed-order two-stage reductions).
//   cross entropy + logit KD     /root/reference/arxiv_pyg/criterion.py:8-21 (and the CE first line of
//                                every criterion)
//   gather + L2 row normalise    F.normalize after `feat[sampled_inds]`, criterion.py:64-72,136-140
#include "common.h"

namespace {

constexpr int kRowsPerBlock = 4;     // one wave per row
constexpr int kMaxBlocks = 1024;     // partial sums per launch (fixed => deterministic finalize)

// A row is walked by a sub-group of SUB lanes (64: one row per wave; 16: four rows per wave -- class counts <= 64, where a
// whole wave per 40-column row left the kernels latency-bound: 37 + 32 us for 90 941 x 40).
template <int SUB>
__device__ __forceinline__ float sub_max(float v) {
#pragma unroll
  for (int o = SUB / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
template <int SUB>
__device__ __forceinline__ float sub_sum(float v) {
#pragma unroll
  for (int o = SUB / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// log-sum-exp of (x * s) over a row distributed lane-strided over the sub-group; every lane of the sub-group gets it
template <int SUB>
__device__ __forceinline__ float row_lse(const float* p, int64_t C, float s, int sl) {
  float m = -INFINITY;
  for (int64_t c = sl; c < C; c += SUB) m = fmaxf(m, p[c] * s);
  m = sub_max<SUB>(m);
  float z = 0.f;
  for (int64_t c = sl; c < C; c += SUB) z += expf(p[c] * s - m);
  z = sub_sum<SUB>(z);
  return m + logf(z);
}

// Row i of the loss reads logits / teacher row r = rows ? rows[i] : i and label labels[r]: with a row list the
// `out[train_idx]`, `y[train_idx]`, `teacher_logits[train_idx]` gathers of train() (gnn.py:107-116) happen in the operand
// loads.  A label outside [0, C) is IGNORED (F.cross_entropy's ignore_index = -100 semantics: no loss, no gradient, not
// counted in the mean) instead of being used as an address.
template <int SUB>
__global__ __launch_bounds__(256) void ce_kd_fwd_kernel(const float* __restrict__ logits, int64_t ldl,
                                                        const float* __restrict__ teacher, int64_t ldt,
                                                        const int64_t* __restrict__ labels, const int64_t* __restrict__ rows,
                                                        int64_t n, int64_t C, float T, float* __restrict__ partials) {
  constexpr int RPW = 64 / SUB;   // rows per wave
  __shared__ float s_ce[kRowsPerBlock], s_kd[kRowsPerBlock], s_nv[kRowsPerBlock];
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int sl = lane % SUB, sg = lane / SUB;
  const float invT = 1.f / T;
  float ce = 0.f, kd = 0.f, nv = 0.f;  // lane 0 of each sub-group carries its running sums
  for (int64_t i0 = (blockIdx.x * (int64_t)kRowsPerBlock + wave) * RPW; i0 < n; i0 += (int64_t)gridDim.x * kRowsPerBlock * RPW) {
    const int64_t i = i0 + sg;
    const bool live = i < n;
    const int64_t row = live ? (rows ? rows[i] : i) : 0;
    const float* lp = logits + row * ldl;
    const float lse1 = row_lse<SUB>(lp, live ? C : 0, 1.f, sl);
    const int64_t y = live ? labels[row] : -1;
    const bool valid = y >= 0 && y < C;
    if (sl == 0 && valid) { ce += lse1 - lp[y]; nv += 1.f; }
    if (teacher != nullptr) {
      const float* tp = teacher + row * ldt;
      const float lseq = row_lse<SUB>(lp, live ? C : 0, invT, sl);
      const float lsep = row_lse<SUB>(tp, live ? C : 0, invT, sl);
      float acc = 0.f;
      for (int64_t c = sl; c < (live ? C : 0); c += SUB) {
        const float logp = tp[c] * invT - lsep;
        const float logq = lp[c] * invT - lseq;
        const float p = expf(logp);
        acc += p > 0.f ? p * (logp - logq) : 0.f;  // F.kl_div: 0 where target == 0
      }
      acc = sub_sum<SUB>(acc);
      if (sl == 0) kd += acc;
    }
  }
  if constexpr (RPW > 1) {   // the sub-group leaders of the wave, fixed order
    ce = egnn_wave_sum(sl == 0 ? ce : 0.f);
    kd = egnn_wave_sum(sl == 0 ? kd : 0.f);
    nv = egnn_wave_sum(sl == 0 ? nv : 0.f);
  }
  if (lane == 0) { s_ce[wave] = ce; s_kd[wave] = kd; s_nv[wave] = nv; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f, c = 0.f;
    for (int w = 0; w < kRowsPerBlock; ++w) { a += s_ce[w]; b += s_kd[w]; c += s_nv[w]; }
    partials[blockIdx.x] = a;
    partials[kMaxBlocks + blockIdx.x] = b;
    partials[2 * kMaxBlocks + blockIdx.x] = c;
  }
}

__global__ __launch_bounds__(256) void ce_kd_finalize_kernel(const float* __restrict__ partials, int nblocks, int64_t n,
                                                             int64_t C, int has_teacher, float* __restrict__ out3) {
  __shared__ float s[3][256];
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) { a += partials[i]; b += partials[kMaxBlocks + i]; c += partials[2 * kMaxBlocks + i]; }
  s[0][threadIdx.x] = a; s[1][threadIdx.x] = b; s[2][threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s[0][threadIdx.x] += s[0][threadIdx.x + o]; s[1][threadIdx.x] += s[1][threadIdx.x + o]; s[2][threadIdx.x] += s[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out3[0] = s[0][0] / s[2][0];   // mean over the rows whose label is valid (all of them in the reference's use); 0/0 = nan like torch
    out3[1] = has_teacher ? s[1][0] / ((float)n * (float)C) : 0.f;
    out3[2] = s[2][0];
  }
}

template <int SUB>
__global__ __launch_bounds__(256) void ce_kd_bwd_kernel(const float* __restrict__ logits, int64_t ldl,
                                                        const float* __restrict__ teacher, int64_t ldt,
                                                        const int64_t* __restrict__ labels, const int64_t* __restrict__ rows,
                                                        int64_t n, int64_t C, float T, const float* __restrict__ out3,
                                                        const float* __restrict__ g_cls, const float* __restrict__ g_kd,
                                                        float* __restrict__ dl, int64_t ldd) {
  constexpr int RPW = 64 / SUB;
  const int lane = egnn_lane();
  const int wave = egnn_wave_id();
  const int sl = lane % SUB, sg = lane / SUB;
  const float invT = 1.f / T;
  const float gc = g_cls ? g_cls[0] / out3[2] : 0.f;
  const float gk = (g_kd && teacher) ? g_kd[0] * invT / ((float)n * (float)C) : 0.f;
  for (int64_t i0 = (blockIdx.x * (int64_t)kRowsPerBlock + wave) * RPW; i0 < n; i0 += (int64_t)gridDim.x * kRowsPerBlock * RPW) {
    const int64_t i = i0 + sg;
    const bool live = i < n;
    const int64_t row = live ? (rows ? rows[i] : i) : 0;
    const int64_t Cr = live ? C : 0;
    const float* lp = logits + row * ldl;
    const float lse1 = row_lse<SUB>(lp, Cr, 1.f, sl);
    const int64_t y = live ? labels[row] : -1;
    const float gcr = (y >= 0 && y < C) ? gc : 0.f;
    float lseq = 0.f, lsep = 0.f;
    const float* tp = nullptr;
    if (gk != 0.f) {
      tp = teacher + row * ldt;
      lseq = row_lse<SUB>(lp, Cr, invT, sl);
      lsep = row_lse<SUB>(tp, Cr, invT, sl);
    }
    for (int64_t c = sl; c < Cr; c += SUB) {
      float g = gcr * (expf(lp[c] - lse1) - (c == y ? 1.f : 0.f));
      if (gk != 0.f) g += gk * (expf(lp[c] * invT - lseq) - expf(tp[c] * invT - lsep));
      dl[row * ldd + c] = g;
    }
  }
}

// ---- gather + L2 normalise ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_normalize_kernel(const float* __restrict__ x, int64_t ldx,
                                                               const int64_t* __restrict__ idx, int64_t n, int64_t D,
                                                               float eps, float* __restrict__ out, int64_t ldo,
                                                               float* __restrict__ inv_norm) {
  const int lane = egnn_lane();
  const int64_t row = blockIdx.x * 4LL + egnn_wave_id();
  if (row >= n) return;
  const float* xp = x + (idx ? idx[row] : row) * ldx;
  float ss = 0.f;
  for (int64_t d = lane; d < D; d += 64) { const float v = xp[d]; ss = fmaf(v, v, ss); }
  ss = egnn_wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), eps);
  for (int64_t d = lane; d < D; d += 64) out[row * ldo + d] = xp[d] * inv;
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
}

__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float* __restrict__ xhat, int64_t ldh,
                                                            const float* __restrict__ dout, int64_t ldd,
                                                            const float* __restrict__ inv_norm,
                                                            const int64_t* __restrict__ idx, int64_t n, int64_t D, float eps,
                                                            float* __restrict__ dx, int64_t ldx, int accumulate) {
  const int lane = egnn_lane();
  const int64_t row = blockIdx.x * 4LL + egnn_wave_id();
  if (row >= n) return;
  const float* hp = xhat + row * ldh;
  const float* gp = dout + row * ldd;
  const float inv = inv_norm[row];
  float dot = 0.f;
  for (int64_t d = lane; d < D; d += 64) dot = fmaf(gp[d], hp[d], dot);
  dot = egnn_wave_sum(dot);
  // y = x / max(||x||, eps): when the clamp is active (||x|| < eps) the denominator is constant
  const bool clamped = inv >= 1.f / eps;
  if (clamped) dot = 0.f;
  float* xp = dx + (idx ? idx[row] : row) * ldx;
  for (int64_t d = lane; d < D; d += 64) {
    const float g = inv * (gp[d] - hp[d] * dot);
    xp[d] = accumulate ? xp[d] + g : g;
  }
}

}  // namespace

extern "C" size_t egnn_ce_kd_ws_floats(int64_t) { return 3 * (size_t)kMaxBlocks; }

extern "C" int egnn_ce_kd_fwd_f32(const float* logits, int64_t ld_logits, const float* teacher, int64_t ld_teacher,
                                  const int64_t* labels, const int64_t* rows, int64_t n, int64_t C, float T, float* out3,
                                  float* partials, void* stream) {
  EGNN_CHECK_ARG(n > 0 && C > 0 && logits && labels && out3 && partials && ld_logits >= C && T > 0.f);
  EGNN_CHECK_ARG(teacher == nullptr || ld_teacher >= C);
  const int rpb = kRowsPerBlock * (C <= 64 ? 4 : 1);   // rows per block-iteration
  const int64_t want = (n + rpb - 1) / rpb;
  const int nblocks = (int)(want < kMaxBlocks ? want : kMaxBlocks);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 64) hipLaunchKernelGGL(ce_kd_fwd_kernel<16>, dim3(nblocks), dim3(256), 0, st, logits, ld_logits, teacher, ld_teacher, labels, rows, n, C, T, partials);
  else hipLaunchKernelGGL(ce_kd_fwd_kernel<64>, dim3(nblocks), dim3(256), 0, st, logits, ld_logits, teacher, ld_teacher, labels, rows, n, C, T, partials);
  hipLaunchKernelGGL(ce_kd_finalize_kernel, dim3(1), dim3(256), 0, st, partials, nblocks, n, C, teacher != nullptr, out3);
  return egnn_launch_status();
}

namespace {
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ p, int64_t ld, int64_t n, int64_t C) {
  const int64_t total = n * C;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) p[(t / C) * ld + (t % C)] = 0.f;
}
}  // namespace

extern "C" int egnn_ce_kd_bwd_f32(const float* logits, int64_t ld_logits, const float* teacher, int64_t ld_teacher,
                                  const int64_t* labels, const int64_t* rows, int64_t n_total_rows, int64_t n, int64_t C, float T,
                                  const float* out3, const float* g_cls, const float* g_kd, float* dlogits, int64_t ld_dlogits,
                                  void* stream) {
  EGNN_CHECK_ARG(n > 0 && C > 0 && logits && labels && dlogits && out3 && ld_logits >= C && ld_dlogits >= C && T > 0.f);
  EGNN_CHECK_ARG(rows == nullptr || n_total_rows >= n);
  hipStream_t st = (hipStream_t)stream;
  if (rows) {   // rows outside the list receive no gradient: clear the whole [n_total_rows, C] block first (a kernel, not a memset
                // node: the step is captured into hipGraphs, and a plain launch keeps one kind of node in them)
    const int64_t total = n_total_rows * C;
    const int64_t zb = (total + 1023) / 1024;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)(zb < 8192 ? zb : 8192)), dim3(256), 0, st, dlogits, ld_dlogits, n_total_rows, C);
  }
  const int rpb = kRowsPerBlock * (C <= 64 ? 4 : 1);
  const int64_t want = (n + rpb - 1) / rpb;
  const int nblocks = (int)(want < 4096 ? want : 4096);
  if (C <= 64) hipLaunchKernelGGL(ce_kd_bwd_kernel<16>, dim3(nblocks), dim3(256), 0, st, logits, ld_logits, teacher, ld_teacher,
                                  labels, rows, n, C, T, out3, g_cls, g_kd, dlogits, ld_dlogits);
  else hipLaunchKernelGGL(ce_kd_bwd_kernel<64>, dim3(nblocks), dim3(256), 0, st, logits, ld_logits, teacher, ld_teacher,
                          labels, rows, n, C, T, out3, g_cls, g_kd, dlogits, ld_dlogits);
  return egnn_launch_status();
}

extern "C" int egnn_gather_normalize_rows_f32(const float* x, int64_t ldx, const int64_t* idx, int64_t n, int64_t D, float eps,
                                              float* out, int64_t ldo, float* inv_norm, void* stream) {
  EGNN_CHECK_ARG(n >= 0 && D > 0 && ldx >= D && ldo >= D);
  if (n == 0) return EGNN_OK;
  EGNN_CHECK_ARG(x && out);
  hipLaunchKernelGGL(gather_normalize_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, idx, n, D, eps, out, ldo, inv_norm);
  return egnn_launch_status();
}

extern "C" int egnn_normalize_rows_bwd_f32(const float* xhat, int64_t ldh, const float* dout, int64_t ldd, const float* inv_norm,
                                           const int64_t* idx, int64_t n, int64_t D, float eps, float* dx, int64_t ldx,
                                           int accumulate, void* stream) {
  EGNN_CHECK_ARG(n >= 0 && D > 0 && ldh >= D && ldd >= D && ldx >= D);
  if (n == 0) return EGNN_OK;
  EGNN_CHECK_ARG(xhat && dout && inv_norm && dx);
  hipLaunchKernelGGL(normalize_bwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, xhat, ldh, dout, ldd,
                     inv_norm, idx, n, D, eps, dx, ldx, accumulate);
  return egnn_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// test() of the reference (gnn.py:198-218): y_pred = out.argmax(-1); Evaluator accuracy per split = hits / size.
// One pass: 16 lanes per row (first maximal column, torch.argmax's rule), per-split hit / row counts per block, then a
// fixed-order (integer, exact) sum of the block partials.  Replaces argmax + eq + cast + 3 x (index, mean) + cat.
namespace {
constexpr int kAccBlocks = 1024;

__global__ __launch_bounds__(256) void split_accuracy_kernel(const float* __restrict__ logits, int64_t ld, int64_t n, int64_t C,
                                                             const int64_t* __restrict__ y, const signed char* __restrict__ split_id,
                                                             int* __restrict__ part) {
  __shared__ int sh[4][6];
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int sub = lane >> 4, l16 = lane & 15;
  int hits[3] = {0, 0, 0}, cnt[3] = {0, 0, 0};
  for (int64_t r0 = (int64_t)blockIdx.x * 16; r0 < n; r0 += (int64_t)gridDim.x * 16) {
    const int64_t row = r0 + wave * 4 + sub;
    const bool live = row < n;
    float best = -INFINITY;
    int64_t arg = 0x7fffffffffffLL;
    if (live) {
      const float* lp = logits + row * ld;
      for (int64_t c = l16; c < C; c += 16) {
        const float v = lp[c];
        if (v > best || (v != v && best == best)) { best = v; arg = c; }   // NaN counts as the maximum, like torch
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {   // 16-lane butterfly: larger value wins, ties go to the smaller column
      const float ov = __shfl_xor(best, o);
      const int64_t oa = __shfl_xor(arg, o);
      const bool take = (ov > best) || (ov != ov && best == best) || (ov == best && oa < arg) || (ov != ov && best != best && oa < arg);
      if (take) { best = ov; arg = oa; }
    }
    if (live && l16 == 0) {
      const int s = split_id[row];
      if (s >= 0 && s < 3) {
        cnt[s] += 1;
        hits[s] += (arg == y[row]) ? 1 : 0;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { hits[k] += __shfl_xor(hits[k], o); cnt[k] += __shfl_xor(cnt[k], o); }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { sh[wave][k] = hits[k]; sh[wave][3 + k] = cnt[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void split_accuracy_final_kernel(const int* __restrict__ part, int nblocks, double* __restrict__ acc3) {
  const int lane = threadIdx.x;
  long long v[6] = {0, 0, 0, 0, 0, 0};
  for (int b = lane; b < nblocks; b += 64)
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] += part[b * 6 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
  if (lane < 3) acc3[lane] = v[3 + lane] > 0 ? (double)v[lane] / (double)v[3 + lane] : 0.0 / 0.0;   // empty split: nan, like mean of nothing
}

// the raw counts instead of the ratios: out6 = (hits of train / valid / test, sizes of train / valid / test) -- what a node-range shard
// contributes to the all-rank accuracies
__global__ __launch_bounds__(64) void split_counts_final_kernel(const int* __restrict__ part, int nblocks, double* __restrict__ out6) {
  const int lane = threadIdx.x;
  long long v[6] = {0, 0, 0, 0, 0, 0};
  for (int b = lane; b < nblocks; b += 64)
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] += part[b * 6 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 6; ++k) out6[k] = (double)v[k];
}

// dst[idx[r], :] += src[r, :] for UNIQUE ids (a row-compact gradient joining a dense one, ops._GradTap): one wave per row
__global__ __launch_bounds__(256) void rows_add_kernel(float* __restrict__ dst, int64_t ldd, const int64_t* __restrict__ idx,
                                                       const float* __restrict__ src, int64_t lds_, int64_t n, int64_t C, int vec4) {
  const int lane = egnn_lane(), wave = egnn_wave_id();
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
    float* d = dst + idx[r] * ldd;
    const float* s = src + r * lds_;
    if (vec4) {
      for (int64_t c = lane * 4; c < C; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(s + c);
        float4 b = *reinterpret_cast<float4*>(d + c);
        b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
        *reinterpret_cast<float4*>(d + c) = b;
      }
    } else {
      for (int64_t c = lane; c < C; c += 64) d[c] += s[c];
    }
  }
}
}  // namespace

extern "C" size_t egnn_split_accuracy_ws_ints(void) { return (size_t)kAccBlocks * 6; }

extern "C" int egnn_split_accuracy_f32(const float* logits, int64_t ld, int64_t n, int64_t C, const int64_t* y, const int8_t* split_id,
                                       double* acc3, int32_t* ws, size_t ws_ints, void* stream) {
  EGNN_CHECK_ARG(n > 0 && C > 0 && ld >= C && logits && y && split_id && acc3 && ws);
  if (ws_ints < egnn_split_accuracy_ws_ints()) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = (n + 15) / 16;
  const int nb = (int)(want < kAccBlocks ? want : kAccBlocks);
  hipLaunchKernelGGL(split_accuracy_kernel, dim3(nb), dim3(256), 0, st, logits, ld, n, C, y, (const signed char*)split_id, (int*)ws);
  hipLaunchKernelGGL(split_accuracy_final_kernel, dim3(1), dim3(64), 0, st, (const int*)ws, nb, acc3);
  return egnn_launch_status();
}

extern "C" int egnn_split_counts_f32(const float* logits, int64_t ld, int64_t n, int64_t C, const int64_t* y, const int8_t* split_id,
                                     double* out6, int32_t* ws, size_t ws_ints, void* stream) {
  EGNN_CHECK_ARG(n > 0 && C > 0 && ld >= C && logits && y && split_id && out6 && ws);
  if (ws_ints < egnn_split_accuracy_ws_ints()) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = (n + 15) / 16;
  const int nb = (int)(want < kAccBlocks ? want : kAccBlocks);
  hipLaunchKernelGGL(split_accuracy_kernel, dim3(nb), dim3(256), 0, st, logits, ld, n, C, y, (const signed char*)split_id, (int*)ws);
  hipLaunchKernelGGL(split_counts_final_kernel, dim3(1), dim3(64), 0, st, (const int*)ws, nb, out6);
  return egnn_launch_status();
}

extern "C" int egnn_rows_add_f32(float* dst, int64_t ld_dst, const int64_t* idx, const float* src, int64_t ld_src, int64_t n, int64_t C,
                                 void* stream) {
  EGNN_CHECK_ARG(n >= 0 && C > 0 && ld_dst >= C && ld_src >= C);
  if (n == 0) return EGNN_OK;
  EGNN_CHECK_ARG(dst && idx && src);
  const int vec4 = (C % 4 == 0 && ld_dst % 4 == 0 && ld_src % 4 == 0 && egnn_aligned16(dst) && egnn_aligned16(src)) ? 1 : 0;
  const int64_t want = (n + 3) / 4;
  hipLaunchKernelGGL(rows_add_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, dst, ld_dst, idx, src,
                     ld_src, n, C, vec4);
  return egnn_launch_status();
}
