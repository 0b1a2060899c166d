// Dense fp32 GEMM entry point (K4): C = alpha * op(A) op(B) (+ bias), on the f32-input MFMA.
// Replaces cuBLAS SGEMM under `x @ W` (GCNConv), nn.Linear (SAGEConv, projection heads) and their
// backward GEMMs (/root/reference/arxiv_pyg/gnn.py:47,79,296-306,192).
#include <cstdlib>

#include "gemm3.h"

using namespace egnn_gemm;

// gemm_skinny.hip: one dimension <= 64 (the class count), the other the node count -- HBM-bound shapes.  Return 1 = not taken.
int egnn_skinny_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, int w_kmajor, const float* bias, float* Y, int64_t ldy,
                    int64_t M, int64_t N, int64_t K, float alpha, hipStream_t st);
int egnn_skinny_dx(const float* G, int64_t ldg, const float* B, int64_t ldb, int b_kmajor, float* Y, int64_t ldy, int64_t M, int64_t Nbig,
                   int64_t Ks, float alpha, hipStream_t st);
size_t egnn_skinny_dw_ws_floats(int64_t R, int64_t Ns, int64_t Nb);
// fused_bn.hip: the narrow forward of a 256-wide X with its rows streamed through an LDS tile (1 = shape not taken)
int egnn_skinny_fwd_tile(const float* X, int64_t ldx, const float* W, int64_t ldw, int w_kmajor, const float* bias, float* Y, int64_t ldy, int64_t M,
                         int64_t N, int64_t K, float alpha, hipStream_t st);
int egnn_skinny_dw(const float* S, int64_t lds_, const float* Bg, int64_t ldb, int64_t R, int64_t Ns, int64_t Nb, float alpha, float* C,
                   int64_t c_ld_s, int64_t c_ld_b, float* ws, size_t ws_floats, hipStream_t st);

namespace {

// which skinny form (if any) a plain GEMM call takes: 1 fwd, 2 dx, 3 dw with the narrow matrix = A, 4 dw with the narrow matrix = B
int skinny_kind(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, bool has_bias) {
  if (!trans_a && N <= 64 && M >= 4096 && K % 16 == 0 && K >= 16 && K <= 1024) return 1;
  if (!trans_a && K <= 64 && N % 64 == 0 && N >= 64 && N <= 1024 && M >= 4096 && !has_bias) return 2;
  if (trans_a && !trans_b && K >= 4096 && !has_bias) {
    if (M <= 64 && N % 64 == 0 && N >= 64) return 3;
    if (N <= 64 && M % 64 == 0 && M >= 64) return 4;
  }
  return 0;
}

bool gemm_split_pipe() { return egnn_split_pipe(); }


struct GemmArgs {
  int64_t M, N, K;
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  const float* bias;
  float* C; int64_t ldc;
  float alpha;
  const float* alpha_dev;  // optional device scalar multiplied into alpha
  int split_k;
  int64_t k_per_split;     // multiple of BK
  float* ws;               // [split_k, M, N] when split_k > 1
  const int64_t* rows;     // GATHER 1: storage rows of A ([M,K]) ; GATHER 2: storage rows of B ([K,N]); else unused
  int wide_store;          // output rows 16-byte aligned: epilogue through LDS with 16-byte stores
  int split_pipe;          // 1: products on the bf16 pipe (three-way operand split, gemm_split.h); 0: f32-input MFMA
  const u32x4* planes;     // B cut into tile-packed bf16 planes for the DMA form (gemm3.h) or null
  int relu;                // C = max(alpha A B + bias, 0)
  const float* addend;     // optional [M,N] matrix added in the store: C = alpha A B + bias + addend (never together with relu)
  int64_t ld_add;
};

// epilogue of one output tile (or of one split-K partial)
template <int BM, int BN, int WAVES_M = 2, int TM_, int TN_>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[TM_][TN_], const GemmArgs& g, int64_t m0, int64_t n0, int split,
                                           int lane, int wm, int wn) {
  constexpr int WM = BM / WAVES_M, WN = BN / 2;
  const bool partial = g.split_k > 1;
  const float alpha = partial ? 1.f : g.alpha * (g.alpha_dev ? g.alpha_dev[0] : 1.f);
  float* out = partial ? g.ws + (int64_t)split * g.M * g.N : g.C;
  const int64_t ldo = partial ? g.N : g.ldc;
#pragma unroll
  for (int tn = 0; tn < TN_; ++tn) {
    const int64_t c = n0 + wn * WN + tn * 32 + (lane & 31);
    if (c >= g.N) continue;
    const float bv = (!partial && g.bias) ? g.bias[c] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM_; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M) {
          float v = alpha * acc[tm][tn][r] + bv;
          if (g.addend && !partial) v += g.addend[row * g.ld_add + c];
          out[row * ldo + c] = (g.relu && !partial) ? fmaxf(v, 0.f) : v;
        }
      }
    }
  }
}

// The same epilogue through LDS: a wave's accumulators (a column per lane, 4-byte stores of 128 bytes per row) are turned
// into rows in the operand buffers, which are free once the main loop has finished, and leave as 16-byte stores -- 256 B
// contiguous per 16 lanes, 16 store instructions per wave instead of 64.  (The narrow stores were ~14 % of a K = 256 GEMM:
// profiles/r01_gemm_mainloop_ablation.txt, "no epilogue".)  Needs 16-byte aligned output rows; else store_tile.
template <int BM, int BN, int WAVES_M = 2, int TM_, int TN_>
__device__ __forceinline__ void store_tile_wide(const f32x16 (&acc)[TM_][TN_], const GemmArgs& g, int64_t m0, int64_t n0, int split,
                                                int lane, int wm, int wn, float* smem) {
  constexpr int WM = BM / WAVES_M, WN = BN / 2, LD = WN + 4, F4 = WN / 4, RPI = 64 / F4;   // float4 per row, rows per store instruction
  // the smaller of the two LDS images a kernel with this tile can have (f32 pipeline 4 waves; the 8-wave form is split-only)
  static_assert(2 * WAVES_M * 32 * LD <= (WAVES_M == 2 ? 2 * (BM + BN) * LDS_LD : 2 * 3 * (BM + BN) * S_ROW / 4), "epilogue staging must fit the operand buffers");
  const bool partial = g.split_k > 1;
  const float alpha = partial ? 1.f : g.alpha * (g.alpha_dev ? g.alpha_dev[0] : 1.f);
  float* out = partial ? g.ws + (int64_t)split * g.M * g.N : g.C;
  const int64_t ldo = partial ? g.N : g.ldc;
  float* sm = smem + (wm * 2 + wn) * (32 * LD);
  const bool cols_full = n0 + BN <= g.N;
  __syncthreads();   // every wave is done with the operand tiles
#pragma unroll
  for (int tm = 0; tm < TM_; ++tm) {
#pragma unroll
    for (int tn = 0; tn < TN_; ++tn) {
      const int64_t c = n0 + wn * WN + tn * 32 + (lane & 31);
      const float bv = (!partial && g.bias && c < g.N) ? g.bias[c] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = alpha * acc[tm][tn][r] + bv;
        sm[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LD + tn * 32 + (lane & 31)] = (g.relu && !partial) ? fmaxf(v, 0.f) : v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the wave's LDS stores are ordered before its loads (other lanes' data)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int rin = it * RPI + lane / F4, c4 = (lane % F4) * 4;
      const int64_t row = m0 + wm * WM + tm * 32 + rin;
      const int64_t col = n0 + wn * WN + c4;
      float4 v = *reinterpret_cast<const float4*>(sm + rin * LD + c4);
      float* o = out + row * ldo + col;
      if (g.addend && !partial && row < g.M) {   // the wide form is only taken with a 16-byte addressable addend (gemm_impl)
        const float* ad = g.addend + row * g.ld_add + col;
        if (cols_full) {
          const float4 a4 = *reinterpret_cast<const float4*>(ad);
          v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
        } else {
          if (col < g.N) v.x += ad[0];
          if (col + 1 < g.N) v.y += ad[1];
          if (col + 2 < g.N) v.z += ad[2];
          if (col + 3 < g.N) v.w += ad[3];
        }
      }
      if (cols_full) {               // block-uniform: the whole tile's columns are in range -> one 16-byte store per lane
        if (row < g.M) *reinterpret_cast<float4*>(o) = v;
      } else if (row < g.M) {
        if (col < g.N) o[0] = v.x;
        if (col + 1 < g.N) o[1] = v.y;
        if (col + 2 < g.N) o[2] = v.z;
        if (col + 3 < g.N) o[3] = v.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // ... and the loads before the next half's stores
    __builtin_amdgcn_wave_barrier();
  }
}

// SPLIT: products on the bf16 matrix pipe from a three-way split of the fp32 operands (gemm_split.h); its LDS image
// (72 KB for 128 x 128) is dynamic shared memory.
template <int BM, int BN, int AMAJ, int BMAJ, bool VEC4, int GATHER = 0, bool SPLIT = false>
__global__ __launch_bounds__(256, SPLIT ? 2 : 1) void gemm_kernel(const GemmArgs g) {
  constexpr int WAVES_M = 2;
  using TS = typename TileSel<SPLIT, BM, BN>::type;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t tiles_n = (g.N + BN - 1) / BN;
  // workgroup b runs on XCD b % 8: give every XCD a contiguous run of tiles, so that the column tiles of one row tile
  // (which read the same rows of A) follow each other on ONE L2 instead of being dealt out to eight (bijective for any
  // tile count: the first `rem` XCDs take one tile more)
  int64_t tile = blockIdx.x;
  if (tiles_n > 1 && tiles_n <= 8) {   // (wide outputs keep the round-robin order: an XCD then shares B tiles as well; 4096^3: 853 vs 896 us)
    const int64_t tiles = gridDim.x, q = tiles >> 3, rem = tiles & 7, xcd = tile & 7, j = tile >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
  }
  const int64_t m0 = (tile / tiles_n) * BM;
  const int64_t n0 = (tile % tiles_n) * BN;
  const int split = blockIdx.y;
  const int64_t kbeg = split * g.k_per_split;
  int64_t kend = kbeg + g.k_per_split;
  if (kend > g.K) kend = g.K;

  f32x16 acc[TS::TM][TS::TN];
  zero_acc(acc);
  IdentityXf id;
  if constexpr (SPLIT) {
    // the whole k-steps run a loop without per-element guards (rows past the edge of the last tile are clamped to the
    // last row and never stored); a ragged end of the reduction (K % 16) is one more, guarded, step
    const int64_t kfull = kbeg + ((kend - kbeg) / BK) * BK;
    mainloop_split<BM, BN, AMAJ, BMAJ, VEC4, true, GATHER == 1, GATHER == 2>(acc, g.A, g.lda, m0, g.M, g.B, g.ldb, n0, g.N, kbeg, kfull, id, id, smem,
                                                                               g.rows, g.rows);
    if (kfull < kend) {
      __syncthreads();
      mainloop_split<BM, BN, AMAJ, BMAJ, VEC4, false, GATHER == 1, GATHER == 2>(acc, g.A, g.lda, m0, g.M, g.B, g.ldb, n0, g.N, kfull, kend, id, id,
                                                                                  smem, g.rows, g.rows);
    }
  } else {
    mainloop<BM, BN, AMAJ, BMAJ, VEC4, false, GATHER == 1, GATHER == 2>(acc, g.A, g.lda, m0, g.M, g.B, g.ldb, n0, g.N, kbeg, kend, id, id,
                                                                          smem, g.rows, g.rows);
  }
  const int wave = egnn_wave_id();
  if (g.wide_store) store_tile_wide<BM, BN, WAVES_M>(acc, g, m0, n0, split, egnn_lane(), wave >> 1, wave & 1, smem);
  else store_tile<BM, BN, WAVES_M>(acc, g, m0, n0, split, egnn_lane(), wave >> 1, wave & 1);
}

// ---- DMA form (gemm3.h): a node-count-tall A [M, K] (k contiguous) against a SMALL B (layer weights) -------------------------
// x W of GCNConv / nn.Linear and dX = dY W^T (arxiv_pyg/gnn.py:47,52,79,84,192): B is cut once per call into tile-packed bf16
// planes (a few hundred KB), A goes global -> LDS by LDS-DMA as fp32 and is cut on the fragment side.  128 x 128 tiles, k-steps of
// 32, two LDS stages, fragments one k-block ahead in registers (lab: 137-143 vs 155-158 us on 169 216 x 256 x 256).
constexpr int DMA_BKT = 32;
using DmaTile = egnn_gemm3::Tile<egnn_gemm3::F32K, egnn_gemm3::PLANES, 2, 2, DMA_BKT, 2>;

__global__ __launch_bounds__(256, 2) void gemm_dma_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t tiles_n = g.N / 128;
  int64_t tile = blockIdx.x;
  if (tiles_n > 1 && tiles_n <= 8) {   // the column tiles of a row tile next to each other on one XCD (see gemm_kernel)
    const int64_t tiles = gridDim.x, q = tiles >> 3, rem = tiles & 7, xcd = tile & 7, j = tile >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
  }
  const int64_t m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
  f32x16 acc[2][2];
  zero_acc(acc);
  // (no scheduling hint here: iglp_opt(0) measured 137 vs 148 us in the lab on this shape but 146-149 vs 145 us inside the epoch, A/B on one box)
  egnn_gemm3::mainloop<egnn_gemm3::F32K, egnn_gemm3::PLANES, 2, 2, DMA_BKT, 2, 1>(acc, g.A, g.lda, m0, g.planes, g.K / DMA_BKT, n0, 0, g.K,
                                                                                 reinterpret_cast<char*>(smem), g.M);
  const int wave = egnn_wave_id();
  if (g.wide_store) store_tile_wide<128, 128, 2>(acc, g, m0, n0, 0, egnn_lane(), wave >> 1, wave & 1, smem);
  else store_tile<128, 128, 2>(acc, g, m0, n0, 0, egnn_lane(), wave >> 1, wave & 1);
}

// shapes the DMA form takes: tall A with k contiguous, whole 128-column tiles of a small B, whole k-steps of 32
bool dma_form(int trans_a, int64_t M, int64_t N, int64_t K, int split_k, bool gathers) {
  return gemm_split_pipe() && !trans_a && !gathers && split_k <= 1 && M >= 4096 && N % 128 == 0 && N >= 128 && N <= 1024 &&
         K % DMA_BKT == 0 && K >= 2 * DMA_BKT && K <= 4096;
}
inline size_t dma_ws_bytes(int64_t N, int64_t K) { return egnn_gemm3::planes_bytes(N, K, 128, DMA_BKT) + 1024; }

// ---- transposed product against a CONSTANT operand cut once into planes (gemm3.h, F32M x PLANES) ---------------------------------
// dW = dY^T X[idx] of the teacher projection head (arxiv_pyg/gnn.py:296-306 via autograd): X -- the teacher's [N, 750] features -- never
// changes, so its gathered rows are cut ONCE per (X, idx) into tile-packed bf16 planes with the reduction index (the train position)
// as k: the per-step product then reads dY down LDS columns (F32M: the only form a [K, M] operand can take) against planes that need
// neither a gather nor a cut in the loop -- the lab's fastest form (f32m_pln 256 x 256: 49-54 % of the six-product peak against 33 %
// for the register-staged gather-fused kernel on this shape).  256 x 256 tiles, k-steps of 16, three LDS stages, split over k.
constexpr int TNP_BKT = 16, TNP_RB = 256;
using TnpTile = egnn_gemm3::Tile<egnn_gemm3::F32M, egnn_gemm3::PLANES, 4, 4, TNP_BKT, 3>;

__global__ __launch_bounds__(256, 1) void gemm_tn_planes_kernel(const float* __restrict__ A, int64_t lda, const char* __restrict__ planes, int64_t nks,
                                                                int64_t M, int64_t Np, int64_t K, int64_t k_per_split, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t tiles_n = Np / TnpTile::BN;
  int64_t tile = blockIdx.x;
  if (tiles_n > 1 && tiles_n <= 8) {   // the column tiles of a row tile follow each other on ONE XCD (they share the dY tile)
    const int64_t tiles = gridDim.x, q = tiles >> 3, rem = tiles & 7, xcd = tile & 7, j = tile >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
  }
  const int64_t m0 = (tile / tiles_n) * TnpTile::BM, n0 = (tile % tiles_n) * TnpTile::BN;
  const int64_t kbeg = (int64_t)blockIdx.y * k_per_split, kend = kbeg + k_per_split;   // whole k-steps; past K: zero planes x the last dY row
  f32x16 acc[4][4];
  zero_acc(acc);
  egnn_gemm3::mainloop<egnn_gemm3::F32M, egnn_gemm3::PLANES, 4, 4, TNP_BKT, 3, 1, egnn_gemm3::SCHED_IGLP0>(acc, A, lda, m0, planes, nks, n0, kbeg, kend,
                                                                                    reinterpret_cast<char*>(smem), K);
  const int lane = egnn_lane(), wave = egnn_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  float* out = ws + (int64_t)blockIdx.y * M * Np;
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    const int64_t c = n0 + wn * 128 + tn * 32 + (lane & 31);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[row * Np + c] = acc[tm][tn][r];
      }
  }
}

// C[m, n] = alpha * sum over the k-ranges (fixed order: four interleaved lanes, combined in lane order -- see splitk_reduce_kernel) of
// ws[s][m][n], n < N (the planes' padded columns are dropped)
__global__ __launch_bounds__(256) void gemm_tn_planes_reduce_kernel(const float* __restrict__ ws, int splits, int64_t M, int64_t Np, int64_t N,
                                                                   float alpha, float* __restrict__ C, int64_t ldc) {
  __shared__ float sh[4][64];
  const int64_t total = M * N;
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
    const int64_t t = base + e;
    const int64_t row = t / N, c = t % N;
    float s = 0.f;
    if (t < total) {
      const float* p = ws + row * Np + c;
#pragma unroll 8
      for (int k = q; k < splits; k += 4) s += p[(int64_t)k * M * Np];
    }
    sh[q][e] = s;
    __syncthreads();
    if (q == 0 && t < total) C[row * ldc + c] = alpha * (((sh[0][e] + sh[1][e]) + sh[2][e]) + sh[3][e]);
    __syncthreads();
  }
}

// ---- forward of the same Linear: y = X[idx] W^T + b with the CONSTANT gathered rows cut once into planes (PLANES x PLANES) -------------
// A = the planes of X[idx] (row blocks of 128 gathered rows x k-steps of 16: the gather is baked in at pack time), B = W cut per call;
// both stages are lane-linear LDS-DMA copies, no VALU on either operand in the loop.  128 x 128 tiles, three LDS stages, two
// workgroups per CU.
constexpr int PP_BKT = 16, PP_RB = 128;
using PpTile = egnn_gemm3::Tile<egnn_gemm3::PLANES, egnn_gemm3::PLANES, 2, 2, PP_BKT, 3>;

__global__ __launch_bounds__(256, 2) void gemm_pp_kernel(const GemmArgs g, const char* __restrict__ planes_a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t tiles_n = g.N / 128;
  int64_t tile = blockIdx.x;
  if (tiles_n > 1 && tiles_n <= 8) {   // the column tiles of a row tile next to each other on one XCD (see gemm_kernel)
    const int64_t tiles = gridDim.x, q = tiles >> 3, rem = tiles & 7, xcd = tile & 7, j = tile >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + j;
  }
  const int64_t m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
  const int64_t nks = (g.K + PP_BKT - 1) / PP_BKT;
  f32x16 acc[2][2];
  zero_acc(acc);
  egnn_gemm3::mainloop<egnn_gemm3::PLANES, egnn_gemm3::PLANES, 2, 2, PP_BKT, 3, 1, egnn_gemm3::SCHED_HAND>(acc, planes_a, nks, m0, g.planes, nks, n0, 0, nks * PP_BKT,
                                                                                     reinterpret_cast<char*>(smem));
  const int wave = egnn_wave_id();
  if (g.wide_store) store_tile_wide<128, 128, 2>(acc, g, m0, n0, 0, egnn_lane(), wave >> 1, wave & 1, smem);
  else store_tile<128, 128, 2>(acc, g, m0, n0, 0, egnn_lane(), wave >> 1, wave & 1);
}

// k-ranges: fill the 256 CUs (one 256 x 256 workgroup each) with whole k-steps; <= 128 ranges
inline void tnp_split(int64_t Np, int64_t K, int& splits, int64_t& k_per_split) {
  const int64_t tiles = Np / TnpTile::BN;       // per 256-row tile of the output (the usual case: M = 256, one row tile)
  int64_t s = 256 / (tiles > 0 ? tiles : 1);
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  const int64_t ksteps = (K + TNP_BKT - 1) / TNP_BKT;
  if (s > ksteps / 8) s = ksteps / 8 > 0 ? ksteps / 8 : 1;       // at least 8 k-steps per range (the pipeline's fill)
  k_per_split = ((ksteps + s - 1) / s) * TNP_BKT;
  splits = (int)((K + k_per_split - 1) / k_per_split);
}

// Fixed-order reduction of the split-K partials: 64 output elements per workgroup x 4 split lanes (lane q adds partials q, q + 4, ...
// with eight loads in flight), the four lane sums are combined in lane order.  (One thread per element walking all partials in turn
// was latency-bound: 31 us for 128 partials of 256 x 256 -- 1 TB/s.)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs g) {
  __shared__ float sh[4][64];
  const int64_t total = g.M * g.N;
  const float alpha = g.alpha * (g.alpha_dev ? g.alpha_dev[0] : 1.f);
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
    const int64_t t = base + e;
    float s = 0.f;
    if (t < total) {
      const float* p = g.ws + t;
#pragma unroll 8
      for (int k = q; k < g.split_k; k += 4) s += p[(int64_t)k * total];
    }
    sh[q][e] = s;
    __syncthreads();
    if (q == 0 && t < total) {
      const float sum = ((sh[0][e] + sh[1][e]) + sh[2][e]) + sh[3][e];
      const int64_t row = t / g.N, c = t % g.N;
      const float v = alpha * sum + (g.bias ? g.bias[c] : 0.f) + (g.addend ? g.addend[row * g.ld_add + c] : 0.f);
      g.C[row * g.ldc + c] = g.relu ? fmaxf(v, 0.f) : v;
    }
    __syncthreads();
  }
}

template <int BM, int BN, int AMAJ, int BMAJ, bool VEC4, int GATHER, bool SPLIT>
int launch_one(const GemmArgs& g, dim3 grid, hipStream_t st) {
  using TS = typename TileSel<SPLIT, BM, BN>::type;
  constexpr size_t shm = (size_t)TS::SMEM_FLOATS * sizeof(float);
  return launch_dyn_lds<gemm_kernel<BM, BN, AMAJ, BMAJ, VEC4, GATHER, SPLIT>>(grid, dim3(256), shm, st, g);
}

template <int BM, int BN, int AMAJ, int BMAJ, int GATHER = 0>
int launch_tile(const GemmArgs& g, bool vec4, hipStream_t st) {
  const int64_t tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  if (tiles > 0x7fffffffLL) return EGNN_EINVAL;
  dim3 grid((unsigned)tiles, (unsigned)g.split_k);
  if constexpr (BM % 128 == 0 && BN % 128 == 0) {
    if (g.split_pipe) {
      return vec4 ? launch_one<BM, BN, AMAJ, BMAJ, true, GATHER, true>(g, grid, st)
                  : launch_one<BM, BN, AMAJ, BMAJ, false, GATHER, true>(g, grid, st);
    }
  }
  return vec4 ? launch_one<BM, BN, AMAJ, BMAJ, true, GATHER, false>(g, grid, st)
              : launch_one<BM, BN, AMAJ, BMAJ, false, GATHER, false>(g, grid, st);
}

template <int AMAJ, int BMAJ, int GATHER = 0>
int launch_major(const GemmArgs& g, bool vec4, hipStream_t st) {
  // narrow outputs (N <= 64, e.g. the 40-class layer) take the 128x64 tile
  if (g.N <= 64) return launch_tile<128, 64, AMAJ, BMAJ, GATHER>(g, vec4, st);
  return launch_tile<128, 128, AMAJ, BMAJ, GATHER>(g, vec4, st);
}

}  // namespace

static int gemm_impl(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                     const int64_t* a_rows, const float* B, int64_t ldb, const int64_t* b_rows, const float* bias, float* C,
                     int64_t ldc, int split_k, float* ws, size_t ws_bytes, void* stream, int flags = 0, const float* addend = nullptr,
                     int64_t ld_add = 0) {
  EGNN_CHECK_ARG(M >= 0 && N >= 0 && K >= 0);
  if (M == 0 || N == 0) return EGNN_OK;
  EGNN_CHECK_ARG(A && B && C && ldc >= N);
  EGNN_CHECK_ARG(addend == nullptr || (ld_add >= N && !(flags & 1)));
  EGNN_CHECK_ARG(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N));
  // fused row gathers: rows of an [M,K]-stored A (the operand of x[idx] @ W^T) or of a [K,N]-stored B (dW = dY^T x[idx])
  EGNN_CHECK_ARG(!(a_rows && b_rows) && !(a_rows && trans_a) && !(b_rows && trans_b));
  hipStream_t st0 = (hipStream_t)stream;
  if (!a_rows && !b_rows && flags == 0 && !addend) {   // class-count-wide shapes: dedicated HBM-bound kernels (gemm_skinny.hip); rc 1 = shape not taken
    const int kind = skinny_kind(trans_a, trans_b, M, N, K, bias != nullptr);
    int rc = 1;
    if (kind == 1) {
      rc = egnn_skinny_fwd_tile(A, lda, B, ldb, trans_b, bias, C, ldc, M, N, K, alpha, st0);
      if (rc == 1) rc = egnn_skinny_fwd(A, lda, B, ldb, trans_b, bias, C, ldc, M, N, K, alpha, st0);
    }
    else if (kind == 2) rc = egnn_skinny_dx(A, lda, B, ldb, trans_b, C, ldc, M, N, K, alpha, st0);
    else if (kind == 3) rc = egnn_skinny_dw(A, lda, B, ldb, K, M, N, alpha, C, ldc, 1, ws, ws_bytes / sizeof(float), st0);
    else if (kind == 4) rc = egnn_skinny_dw(B, ldb, A, lda, K, N, M, alpha, C, 1, ldc, ws, ws_bytes / sizeof(float), st0);
    if (rc != 1) return rc;
  }
  if (split_k < 1) split_k = 1;
  int64_t ksteps = (K + BK - 1) / BK;
  if (split_k > ksteps) split_k = (int)(ksteps > 0 ? ksteps : 1);
  if (split_k > 1) {
    if (!ws || ws_bytes < (size_t)split_k * M * N * sizeof(float)) return EGNN_EWORKSPACE;
  }
  const bool wide = split_k > 1 ? (N % 4 == 0 && egnn_aligned16(ws))
                                : (ldc % 4 == 0 && egnn_aligned16(C) && (!addend || (ld_add % 4 == 0 && egnn_aligned16(addend))));
  GemmArgs g{M, N, K, A, lda, B, ldb, bias, C, ldc, alpha, nullptr, split_k,
             ((ksteps + split_k - 1) / split_k) * BK, ws, a_rows ? a_rows : b_rows, wide ? 1 : 0,
             gemm_split_pipe() ? 1 : 0,
             nullptr, (flags & 1) ? 1 : 0, addend, ld_add};
  hipStream_t st = (hipStream_t)stream;
  const bool vec4 = (lda % 4 == 0) && (ldb % 4 == 0) && egnn_aligned16(A) && egnn_aligned16(B);
  const int amaj = trans_a ? MNMAJOR : KMAJOR;   // A stored [K,M] when transposed
  const int bmaj = trans_b ? KMAJOR : MNMAJOR;   // B stored [N,K] when transposed, else [K,N]
  int rc;
  if (dma_form(trans_a, M, N, K, split_k, a_rows || b_rows) && ws && ws_bytes >= dma_ws_bytes(N, K) && lda % 4 == 0 && egnn_aligned16(A)) {
    char* planes = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 1023) & ~(uintptr_t)1023);
    // B(k, n): stored [N, K] when trans_b (k contiguous), else [K, N]
    egnn_gemm3::pack_planes<128, DMA_BKT>(B, ldb, trans_b ? 1 : 0, N, K, nullptr, nullptr, nullptr, 0.f, planes, st);
    g.planes = reinterpret_cast<const u32x4*>(planes);
    const int64_t tiles = ((M + 127) / 128) * (N / 128);
    if (tiles > 0x7fffffffLL) return EGNN_EINVAL;
    rc = launch_dyn_lds<gemm_dma_kernel>(dim3((unsigned)tiles), dim3(256), (size_t)DmaTile::SMEM_BYTES, st, g);
    if (rc != EGNN_OK) return rc;
    return egnn_launch_status();
  }
  if (a_rows) {
    rc = bmaj == KMAJOR ? launch_major<KMAJOR, KMAJOR, 1>(g, vec4, st) : launch_major<KMAJOR, MNMAJOR, 1>(g, vec4, st);
  } else if (b_rows) {
    rc = amaj == KMAJOR ? launch_major<KMAJOR, MNMAJOR, 2>(g, vec4, st) : launch_major<MNMAJOR, MNMAJOR, 2>(g, vec4, st);
  } else if (amaj == KMAJOR && bmaj == KMAJOR) rc = launch_major<KMAJOR, KMAJOR>(g, vec4, st);
  else if (amaj == KMAJOR) rc = launch_major<KMAJOR, MNMAJOR>(g, vec4, st);
  else if (bmaj == KMAJOR) rc = launch_major<MNMAJOR, KMAJOR>(g, vec4, st);
  else rc = launch_major<MNMAJOR, MNMAJOR>(g, vec4, st);
  if (rc != EGNN_OK) return rc;
  if (split_k > 1) {
    const int64_t blocks = (M * N + 63) / 64;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, g);
  }
  return egnn_launch_status();
}

extern "C" size_t egnn_gemm_ws_floats(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, int split_k) {
  size_t need = split_k > 1 ? (size_t)split_k * (size_t)M * (size_t)N : 0;
  const int kind = skinny_kind(trans_a, trans_b, M, N, K, false);
  if (kind == 0 && dma_form(trans_a, M, N, K, split_k, false)) {   // tile-packed planes of B for the DMA form (gemm3.h)
    const size_t dma = dma_ws_bytes(N, K) / sizeof(float) + 1;
    if (dma > need) need = dma;
  }
  size_t sk = 0;
  if (kind == 3) sk = egnn_skinny_dw_ws_floats(K, M, N);
  else if (kind == 4) sk = egnn_skinny_dw_ws_floats(K, N, M);
  return need > sk ? need : sk;
}

extern "C" int egnn_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                             int64_t lda, const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc,
                             int split_k, float* ws, size_t ws_bytes, void* stream) {
  return gemm_impl(trans_a, trans_b, M, N, K, alpha, A, lda, nullptr, B, ldb, nullptr, bias, C, ldc, split_k, ws, ws_bytes, stream);
}

extern "C" int egnn_gemm_ex_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                                const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc, int split_k, float* ws,
                                size_t ws_bytes, int flags, void* stream) {
  return gemm_impl(trans_a, trans_b, M, N, K, alpha, A, lda, nullptr, B, ldb, nullptr, bias, C, ldc, split_k, ws, ws_bytes, stream, flags);
}

extern "C" int egnn_gemm_add_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                                 const float* B, int64_t ldb, const float* bias, const float* addend, int64_t ld_addend, float* C,
                                 int64_t ldc, int split_k, float* ws, size_t ws_bytes, void* stream) {
  return gemm_impl(trans_a, trans_b, M, N, K, alpha, A, lda, nullptr, B, ldb, nullptr, bias, C, ldc, split_k, ws, ws_bytes, stream, 0, addend,
                   ld_addend);
}

extern "C" int egnn_gemm_rows_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                                  int64_t lda, const int64_t* a_rows, const float* B, int64_t ldb, const int64_t* b_rows,
                                  const float* bias, float* C, int64_t ldc, int split_k, float* ws, size_t ws_bytes,
                                  void* stream) {
  return gemm_impl(trans_a, trans_b, M, N, K, alpha, A, lda, a_rows, B, ldb, b_rows, bias, C, ldc, split_k, ws, ws_bytes, stream);
}

// ---- C = alpha A^T B for a CONSTANT, row-gathered B cut once into planes (see gemm_tn_planes_kernel) -----------------------------------
extern "C" size_t egnn_gemm_tn_planes_bytes(int64_t N, int64_t K) {
  const int64_t Np = (N + TNP_RB - 1) / TNP_RB * TNP_RB;
  int splits; int64_t kps;
  tnp_split(Np, K, splits, kps);
  return egnn_gemm3::planes_bytes(Np, (int64_t)splits * kps, TNP_RB, TNP_BKT) + 1024;
}

extern "C" int egnn_gemm_tn_planes_pack_f32(const float* B, int64_t ldb, const int64_t* b_rows, int64_t N, int64_t K, void* planes,
                                            size_t planes_bytes, void* stream) {
  EGNN_CHECK_ARG(B && planes && N > 0 && K > 0 && ldb >= N);
  if (planes_bytes < egnn_gemm_tn_planes_bytes(N, K) || (reinterpret_cast<uintptr_t>(planes) & 1023u)) return EGNN_EWORKSPACE;
  const int64_t Np = (N + TNP_RB - 1) / TNP_RB * TNP_RB;
  int splits; int64_t kps;
  tnp_split(Np, K, splits, kps);
  // operand rows = the N columns of B (zero planes past N), k = the K (gathered) rows of B (zero planes past K, up to whole k-ranges)
  egnn_gemm3::pack_planes_kernel<TNP_RB, TNP_BKT><<<dim3(16384), dim3(256), 0, (hipStream_t)stream>>>(
      B, ldb, 0, N, K, nullptr, nullptr, nullptr, 0.f, (char*)planes, b_rows, (int64_t)splits * kps);
  return egnn_launch_status();
}

extern "C" size_t egnn_gemm_tn_planes_ws_floats(int64_t M, int64_t N, int64_t K) {
  const int64_t Np = (N + TNP_RB - 1) / TNP_RB * TNP_RB;
  int splits; int64_t kps;
  tnp_split(Np, K, splits, kps);       // (the k-ranges do not depend on M: the planes were padded with the same)
  return (size_t)splits * (size_t)M * (size_t)Np;
}

extern "C" int egnn_gemm_tn_planes_f32(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const void* planes,
                                       float* C, int64_t ldc, float* ws, size_t ws_floats, void* stream) {
  EGNN_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && planes && C && lda >= M && ldc >= N);
  if (M % TnpTile::BM != 0 || lda % 4 != 0 || !egnn_aligned16(A) || (reinterpret_cast<uintptr_t>(planes) & 1023u) || !gemm_split_pipe())
    return EGNN_EALIGN;
  if (ws == nullptr || ws_floats < egnn_gemm_tn_planes_ws_floats(M, N, K)) return EGNN_EWORKSPACE;
  const int64_t Np = (N + TNP_RB - 1) / TNP_RB * TNP_RB;
  int splits; int64_t kps;
  tnp_split(Np, K, splits, kps);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((M / TnpTile::BM) * (Np / TnpTile::BN)), (unsigned)splits);
  const int rc = launch_dyn_lds<gemm_tn_planes_kernel>(grid, dim3(256), (size_t)TnpTile::SMEM_BYTES, st, A, lda, (const char*)planes,
                                                       (int64_t)splits * kps / TNP_BKT, M, Np, K, kps, ws);
  if (rc != EGNN_OK) return rc;
  const int64_t blocks = (M * N + 63) / 64;
  hipLaunchKernelGGL(gemm_tn_planes_reduce_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, ws, splits, M, Np, N, alpha, C,
                     ldc);
  return egnn_launch_status();
}

// ---- C = alpha A[a_rows] B^T + bias for a CONSTANT, row-gathered A cut once into planes (see gemm_pp_kernel) ----------------------------
extern "C" size_t egnn_gemm_rows_planes_bytes(int64_t M, int64_t K) { return egnn_gemm3::planes_bytes(M, K, PP_RB, PP_BKT) + 1024; }

extern "C" int egnn_gemm_rows_planes_pack_f32(const float* A, int64_t lda, const int64_t* a_rows, int64_t M, int64_t K, void* planes,
                                              size_t planes_bytes, void* stream) {
  EGNN_CHECK_ARG(A && planes && M > 0 && K > 0 && lda >= K);
  if (planes_bytes < egnn_gemm_rows_planes_bytes(M, K) || (reinterpret_cast<uintptr_t>(planes) & 1023u)) return EGNN_EWORKSPACE;
  egnn_gemm3::pack_planes<PP_RB, PP_BKT>(A, lda, 1, M, K, nullptr, a_rows, nullptr, 0.f, (char*)planes, (hipStream_t)stream);
  return egnn_launch_status();
}

extern "C" size_t egnn_gemm_rows_planes_ws_bytes(int64_t N, int64_t K) { return egnn_gemm3::planes_bytes(N, K, PP_RB, PP_BKT) + 2048; }

extern "C" int egnn_gemm_rows_planes_f32(int64_t M, int64_t N, int64_t K, float alpha, const void* planes_a, const float* B, int64_t ldb,
                                         const float* bias, float* C, int64_t ldc, void* ws, size_t ws_bytes, void* stream) {
  EGNN_CHECK_ARG(M > 0 && N > 0 && K > 0 && planes_a && B && C && ldb >= K && ldc >= N);
  if (N % 128 != 0 || (reinterpret_cast<uintptr_t>(planes_a) & 1023u) || !gemm_split_pipe()) return EGNN_EALIGN;
  if (ws == nullptr || ws_bytes < egnn_gemm_rows_planes_ws_bytes(N, K)) return EGNN_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* planes_b = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 1023) & ~(uintptr_t)1023);
  egnn_gemm3::pack_planes<PP_RB, PP_BKT>(B, ldb, 1, N, K, nullptr, nullptr, nullptr, 0.f, planes_b, st);   // B stored [N, K]: k contiguous
  const bool wide = ldc % 4 == 0 && egnn_aligned16(C);
  GemmArgs g{M, N, K, nullptr, 0, nullptr, 0, bias, C, ldc, alpha, nullptr, 1, 0, nullptr, nullptr, wide ? 1 : 0, 1,
             reinterpret_cast<const u32x4*>(planes_b), 0, nullptr, 0};
  const int64_t tiles = ((M + 127) / 128) * (N / 128);
  if (tiles > 0x7fffffffLL) return EGNN_EINVAL;
  const int rc = launch_dyn_lds<gemm_pp_kernel>(dim3((unsigned)tiles), dim3(256), (size_t)PpTile::SMEM_BYTES, st, g, (const char*)planes_a);
  if (rc != EGNN_OK) return rc;
  return egnn_launch_status();
}
