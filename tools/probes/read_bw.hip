// Probe: read bandwidth vs footprint (L2 / Infinity Cache / HBM) and random 128-byte line gathers.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probes/read_bw.hip -o /tmp/read_bw && /tmp/read_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

__global__ __launch_bounds__(256) void stream_read(const float4* __restrict__ p, size_t n4, int reps, float* out) {
  float4 acc = make_float4(0, 0, 0, 0);
  for (int r = 0; r < reps; ++r)
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
      const float4 v = p[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

// each 8-lane group gathers one random 128-byte line (float4 per lane); `per_xcd_slices`: if 1, block b only
// touches the 128-byte slice (b % 8) of each 1 KiB row (the SpMM slice<->XCD binding), else the full row set.
__global__ __launch_bounds__(256) void gather_lines(const float4* __restrict__ X, const int* __restrict__ idx, size_t n_idx,
                                                    int row_f4, int bind, int reps, float* out) {
  float4 acc = make_float4(0, 0, 0, 0);
  const int lane = threadIdx.x & 63, sub = lane >> 3, li = lane & 7;
  const size_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6;
  const size_t nwaves = (size_t)gridDim.x * 4;
  const int slice = bind ? (blockIdx.x & 7) : 0;
  for (int r = 0; r < reps; ++r)
    for (size_t i = wave * 8 + sub; i < n_idx; i += nwaves * 8 * 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        size_t j = i + (size_t)u * nwaves * 8;
        if (j >= n_idx) j = i;
        const int c = idx[j];
        const int s = bind ? slice : (int)((j + r) & 7);
        v[u] = X[(size_t)c * row_f4 + s * 8 + li];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

// hot/cold mix: idx < 0 marks a "cold" row (bit 31 set) that is gathered with a non-temporal load when use_nt != 0
__global__ __launch_bounds__(256) void gather_hotcold(const float4* __restrict__ X, const int* __restrict__ idx, size_t n_idx,
                                                      int row_f4, int use_nt, int reps, float* out) {
  float4 acc = make_float4(0, 0, 0, 0);
  const int lane = threadIdx.x & 63, sub = lane >> 3, li = lane & 7;
  const size_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6;
  const size_t nwaves = (size_t)gridDim.x * 4;
  const int slice = blockIdx.x & 7;
  for (int r = 0; r < reps; ++r)
    for (size_t i = wave * 8 + sub; i < n_idx; i += nwaves * 8 * 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        size_t j = i + (size_t)u * nwaves * 8;
        if (j >= n_idx) j = i;
        const int ce = idx[j];
        const int c = ce & 0x7fffffff;
        const float4* p = X + (size_t)c * row_f4 + slice * 8 + li;
        typedef float f4 __attribute__((ext_vector_type(4)));
        if (use_nt && ce < 0) { const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)); v[u] = make_float4(t.x, t.y, t.z, t.w); }
        else v[u] = *p;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float* out; hipMalloc(&out, 4);
  size_t maxb = 2ull << 30;
  float4* buf; hipMalloc(&buf, maxb); hipMemset(buf, 0, maxb);
  printf("== streaming read bandwidth vs footprint\n");
  for (size_t mb : {4, 16, 24, 48, 96, 160, 224, 512, 2048}) {
    size_t n4 = mb * (1ull << 20) / 16;
    int reps = (int)(8192 / mb); if (reps < 2) reps = 2;
    stream_read<<<2048, 256>>>(buf, n4, 2, out);
    hipEventRecord(a);
    stream_read<<<2048, 256>>>(buf, n4, reps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    printf("footprint %5zu MB: %8.1f GB/s\n", mb, (double)n4 * 16 * reps / (time_ms(a, b) * 1e-3) / 1e9);
  }
  printf("== random 128B-line gathers from [rows x 1KiB] (K=256 fp32), uniform random row ids\n");
  const size_t n_idx = 20u << 20;
  std::vector<int> h(n_idx);
  int* idx; hipMalloc(&idx, n_idx * 4);
  for (int rows : {4096, 32768, 169343, 1000000}) {
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n_idx; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (uint64_t)rows); }
    hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice);
    for (int bind : {0, 1}) {
      gather_lines<<<2048, 256>>>((const float4*)buf, idx, n_idx, 64, bind, 1, out);
      hipEventRecord(a);
      gather_lines<<<2048, 256>>>((const float4*)buf, idx, n_idx, 64, bind, 2, out);
      hipEventRecord(b); hipEventSynchronize(b);
      printf("rows %8d (%6.1f MB) bind=%d: %8.1f GB/s of 128B lines\n", rows, rows * 1024.0 / 1e6, bind,
             (double)n_idx * 128 * 2 / (time_ms(a, b) * 1e-3) / 1e9);
    }
  }
  printf("== hot/cold gathers, 169343 rows x 1KiB, slice<->XCD bound: hot set H rows gets 50%% of the accesses\n");
  for (int H : {8192, 16384, 32768}) {
    uint64_t s = 1234567ull;
    for (size_t i = 0; i < n_idx; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const bool hot = (s >> 40) & 1;
      const uint64_t r = s % (uint64_t)(hot ? H : 169343);
      h[i] = (int)r | (hot ? 0 : 0x80000000);
    }
    hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice);
    for (int nt : {0, 1}) {
      gather_hotcold<<<2048, 256>>>((const float4*)buf, idx, n_idx, 64, nt, 1, out);
      hipEventRecord(a);
      gather_hotcold<<<2048, 256>>>((const float4*)buf, idx, n_idx, 64, nt, 2, out);
      hipEventRecord(b); hipEventSynchronize(b);
      printf("hot %6d nt=%d: %8.1f GB/s of 128B lines\n", H, nt, (double)n_idx * 128 * 2 / (time_ms(a, b) * 1e-3) / 1e9);
    }
  }
  return 0;
}
