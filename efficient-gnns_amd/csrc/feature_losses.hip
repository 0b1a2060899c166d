// FitNet and attention-transfer losses as row kernels (wave64 per row, fixed-order reductions; HBM / latency-bound).
//   fitnet : /root/reference/arxiv_pyg/criterion.py:24-36   loss = mse(F.normalize(f), F.normalize(t))      (mean over n * D)
//   at     : /root/reference/arxiv_pyg/criterion.py:39-54   e_s[i] = sum_d f_id^2, e_t likewise;
//                                                           loss = mse(F.normalize(e_s), F.normalize(e_t))  (norm ACROSS nodes)
// F.normalize(x) = x / max(||x||_2, eps), eps = 1e-12: where the clamp is active the denominator is a constant.
#include "common.h"

namespace {

constexpr int kFlBlocks = 1024;   // partial sums per launch (fixed => deterministic finalize)

__device__ __forceinline__ float block4_sum(float v, float* sh) {   // sum of the four waves' lane-0 values; result on every thread
  const int wave = egnn_wave_id();
  if (egnn_lane() == 0) sh[wave] = v;
  __syncthreads();
  const float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}

// ---- FitNet --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fitnet_fwd_kernel(const float* __restrict__ f, int64_t ldf, const float* __restrict__ t, int64_t ldt,
                                                         int64_t n, int64_t D, float eps, float* __restrict__ partials) {
  __shared__ float sh[4];
  const int lane = egnn_lane(), wave = egnn_wave_id();
  float acc = 0.f;
  for (int64_t row = blockIdx.x * 4LL + wave; row < n; row += (int64_t)gridDim.x * 4) {
    const float* fp = f + row * ldf;
    const float* tp = t + row * ldt;
    float ff = 0.f, tt = 0.f;
    for (int64_t d = lane; d < D; d += 64) { ff = fmaf(fp[d], fp[d], ff); tt = fmaf(tp[d], tp[d], tt); }
    ff = egnn_wave_sum(ff); tt = egnn_wave_sum(tt);
    const float inf_ = 1.f / fmaxf(sqrtf(ff), eps), int_ = 1.f / fmaxf(sqrtf(tt), eps);
    float s = 0.f;
    for (int64_t d = lane; d < D; d += 64) { const float u = fp[d] * inf_ - tp[d] * int_; s = fmaf(u, u, s); }
    acc += egnn_wave_sum(s);
  }
  const float tot = block4_sum(acc, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// df = c / |f| (d - f^ (f^ . d)),  dt = -c / |t| (d - t^ (t^ . d)),  d = f^ - t^,  c = 2 g / (n D); the projection term is
// dropped where the norm clamp is active
__global__ __launch_bounds__(256) void fitnet_bwd_kernel(const float* __restrict__ f, int64_t ldf, const float* __restrict__ t, int64_t ldt,
                                                         int64_t n, int64_t D, float eps, const float* __restrict__ g, float* __restrict__ df,
                                                         int64_t lddf, float* __restrict__ dt, int64_t lddt) {
  const int lane = egnn_lane();
  const int64_t row = blockIdx.x * 4LL + egnn_wave_id();
  if (row >= n) return;
  const float* fp = f + row * ldf;
  const float* tp = t + row * ldt;
  float ff = 0.f, tt = 0.f;
  for (int64_t d = lane; d < D; d += 64) { ff = fmaf(fp[d], fp[d], ff); tt = fmaf(tp[d], tp[d], tt); }
  ff = egnn_wave_sum(ff); tt = egnn_wave_sum(tt);
  const float nf = sqrtf(ff), nt = sqrtf(tt);
  const float inf_ = 1.f / fmaxf(nf, eps), int_ = 1.f / fmaxf(nt, eps);
  float fd = 0.f, td = 0.f;   // f^ . d and t^ . d
  for (int64_t d = lane; d < D; d += 64) {
    const float a = fp[d] * inf_, b = tp[d] * int_, u = a - b;
    fd = fmaf(a, u, fd); td = fmaf(b, u, td);
  }
  fd = egnn_wave_sum(fd); td = egnn_wave_sum(td);
  if (nf <= eps) fd = 0.f;
  if (nt <= eps) td = 0.f;
  const float c = 2.f * g[0] / ((float)n * (float)D);
  for (int64_t d = lane; d < D; d += 64) {
    const float a = fp[d] * inf_, b = tp[d] * int_, u = a - b;
    if (df) df[row * lddf + d] = c * inf_ * (u - a * fd);
    if (dt) dt[row * lddt + d] = -c * int_ * (u - b * td);
  }
}

__global__ __launch_bounds__(256) void fl_sum_kernel(const float* __restrict__ partials, int nb, float scale, float* __restrict__ out) {
  __shared__ float red[256];
  float v = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) v += partials[i];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// ---- attention transfer ------------------------------------------------------------------------------------------
// pass 1: row energies of both sides + per-block partials of sum e^2
__global__ __launch_bounds__(256) void at_energy_kernel(const float* __restrict__ f, int64_t ldf, int64_t Df, const float* __restrict__ t, int64_t ldt,
                                                        int64_t Dt, int64_t n, float* __restrict__ es, float* __restrict__ et,
                                                        float* __restrict__ partials) {
  __shared__ float sh[4];
  const int lane = egnn_lane(), wave = egnn_wave_id();
  float a2 = 0.f, b2 = 0.f;
  for (int64_t row = blockIdx.x * 4LL + wave; row < n; row += (int64_t)gridDim.x * 4) {
    const float* fp = f + row * ldf;
    const float* tp = t + row * ldt;
    float a = 0.f, b = 0.f;
    for (int64_t d = lane; d < Df; d += 64) a = fmaf(fp[d], fp[d], a);
    for (int64_t d = lane; d < Dt; d += 64) b = fmaf(tp[d], tp[d], b);
    a = egnn_wave_sum(a); b = egnn_wave_sum(b);
    if (lane == 0) { es[row] = a; et[row] = b; }
    a2 = fmaf(a, a, a2); b2 = fmaf(b, b, b2);
  }
  const float ta = block4_sum(a2, sh), tb = block4_sum(b2, sh);
  if (threadIdx.x == 0) { partials[blockIdx.x] = ta; partials[kFlBlocks + blockIdx.x] = tb; }
}

// pass 2 (one block): norms; then sum d^2, sum d e_s, sum d e_t with d = e_s / ns - e_t / nt.  scal = {ns, nt, sum d e_s, sum d e_t, raw |e_s|, raw |e_t|}
__global__ __launch_bounds__(1024) void at_finish_kernel(const float* __restrict__ es, const float* __restrict__ et, int64_t n, int nb,
                                                         const float* __restrict__ partials, float eps, float* __restrict__ scal,
                                                         float* __restrict__ loss) {
  __shared__ float red[3][1024];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < nb; i += 1024) { a += partials[i]; b += partials[kFlBlocks + i]; }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  const float raw_s = sqrtf(red[0][0]), raw_t = sqrtf(red[1][0]);
  const float ns = fmaxf(raw_s, eps), nt = fmaxf(raw_t, eps);
  __syncthreads();
  float d2 = 0.f, ds = 0.f, dt_ = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float d = es[i] / ns - et[i] / nt;
    d2 = fmaf(d, d, d2); ds = fmaf(d, es[i], ds); dt_ = fmaf(d, et[i], dt_);
  }
  red[0][threadIdx.x] = d2; red[1][threadIdx.x] = ds; red[2][threadIdx.x] = dt_;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = red[0][0] / (float)n;
    scal[0] = ns; scal[1] = nt; scal[2] = red[1][0]; scal[3] = red[2][0]; scal[4] = raw_s; scal[5] = raw_t;
  }
}

// dL/de_s[i] = 2/n (d_i / ns - [|e_s| > eps] e_s[i] (sum_j d_j e_s[j]) / ns^3), dL/de_t[i] = -2/n (d_i / nt - ...);  df_i = g dL/de_s[i] 2 f_i
__global__ __launch_bounds__(256) void at_bwd_kernel(const float* __restrict__ f, int64_t ldf, int64_t Df, const float* __restrict__ t, int64_t ldt,
                                                     int64_t Dt, int64_t n, const float* __restrict__ es, const float* __restrict__ et,
                                                     const float* __restrict__ scal, float eps, const float* __restrict__ g,
                                                     float* __restrict__ df, int64_t lddf, float* __restrict__ dt, int64_t lddt) {
  const int lane = egnn_lane();
  const int64_t row = blockIdx.x * 4LL + egnn_wave_id();
  if (row >= n) return;
  const float ns = scal[0], nt = scal[1];
  const float d = es[row] / ns - et[row] / nt;
  const float ps = scal[4] > eps ? es[row] * scal[2] / (ns * ns * ns) : 0.f;
  const float pt = scal[5] > eps ? et[row] * scal[3] / (nt * nt * nt) : 0.f;
  const float c = 2.f * g[0] / (float)n;
  const float ges = c * (d / ns - ps), get = -c * (d / nt - pt);
  if (df) for (int64_t k = lane; k < Df; k += 64) df[row * lddf + k] = 2.f * ges * f[row * ldf + k];
  if (dt) for (int64_t k = lane; k < Dt; k += 64) dt[row * lddt + k] = 2.f * get * t[row * ldt + k];
}

}  // namespace

extern "C" size_t egnn_feature_loss_ws_floats(int64_t n) { return 2 * (size_t)kFlBlocks + 2 * (size_t)(n > 0 ? n : 0) + 8; }

extern "C" int egnn_fitnet_fwd_f32(const float* f, int64_t ldf, const float* t, int64_t ldt, int64_t n, int64_t D, float eps, float* loss,
                                   float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(n > 0 && D > 0 && f && t && loss && ws && ldf >= D && ldt >= D && eps > 0.f);
  if (ws_floats < egnn_feature_loss_ws_floats(n)) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = (n + 3) / 4;
  const int nb = (int)(want < kFlBlocks ? want : kFlBlocks);
  hipLaunchKernelGGL(fitnet_fwd_kernel, dim3(nb), dim3(256), 0, st, f, ldf, t, ldt, n, D, eps, ws);
  hipLaunchKernelGGL(fl_sum_kernel, dim3(1), dim3(256), 0, st, ws, nb, 1.f / ((float)n * (float)D), loss);
  return egnn_launch_status();
}

extern "C" int egnn_fitnet_bwd_f32(const float* f, int64_t ldf, const float* t, int64_t ldt, int64_t n, int64_t D, float eps, const float* g,
                                   float* df, int64_t lddf, float* dt, int64_t lddt, void* stream) {
  EGNN_CHECK_ARG(n > 0 && D > 0 && f && t && g && ldf >= D && ldt >= D && (df == nullptr || lddf >= D) && (dt == nullptr || lddt >= D));
  hipLaunchKernelGGL(fitnet_bwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, f, ldf, t, ldt, n, D, eps, g, df, lddf,
                     dt, lddt);
  return egnn_launch_status();
}

// ws layout: [2 kFlBlocks partials][e_s n][e_t n][scal 8]; e_s / e_t / scal are what the backward reads (keep ws alive until then)
extern "C" int egnn_at_fwd_f32(const float* f, int64_t ldf, int64_t Df, const float* t, int64_t ldt, int64_t Dt, int64_t n, float eps,
                               float* loss, float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(n > 0 && Df > 0 && Dt > 0 && f && t && loss && ws && ldf >= Df && ldt >= Dt && eps > 0.f);
  if (ws_floats < egnn_feature_loss_ws_floats(n)) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* es = ws + 2 * kFlBlocks;
  float* et = es + n;
  float* scal = et + n;
  const int64_t want = (n + 3) / 4;
  const int nb = (int)(want < kFlBlocks ? want : kFlBlocks);
  hipLaunchKernelGGL(at_energy_kernel, dim3(nb), dim3(256), 0, st, f, ldf, Df, t, ldt, Dt, n, es, et, ws);
  hipLaunchKernelGGL(at_finish_kernel, dim3(1), dim3(1024), 0, st, es, et, n, nb, ws, eps, scal, loss);
  return egnn_launch_status();
}

extern "C" int egnn_at_bwd_f32(const float* f, int64_t ldf, int64_t Df, const float* t, int64_t ldt, int64_t Dt, int64_t n, float eps,
                               const float* ws, const float* g, float* df, int64_t lddf, float* dt, int64_t lddt, void* stream) {
  EGNN_CHECK_ARG(n > 0 && Df > 0 && Dt > 0 && f && t && ws && g && (df == nullptr || lddf >= Df) && (dt == nullptr || lddt >= Dt));
  const float* es = ws + 2 * kFlBlocks;
  const float* et = es + n;
  const float* scal = et + n;
  hipLaunchKernelGGL(at_bwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, f, ldf, Df, t, ldt, Dt, n, es, et, scal, eps, g,
                     df, lddf, dt, lddt);
  return egnn_launch_status();
}
