/*
 * egnn_hip.h -- C ABI of libegnn_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * GNN-distillation hot path of chaitjo/efficient-gnns.
 *
 * The reference has no FFI of its own: its hot path calls the PyG / torch-sparse / torch-scatter
 * Python operator API, whose native kernels live in un-vendored wheels.  Each entry
 * point below therefore cites the reference call site whose third-party kernel it replaces.  The
 * Python host side (efficient-gnns_amd/) binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator); no entry
 *     point allocates, frees or synchronises, and none keeps state between calls: all are re-entrant.
 *     Two process-wide, write-once values exist: the GEMM pipeline choice read from the environment on
 *     first use (EGNN_GEMM_PIPE=f32 pins every product to the f32-input MFMA; default: the bf16-split
 *     pipeline, csrc/gemm_split.h) and the per-device "large dynamic LDS" opt-in of the kernels that need
 *     more than 64 KB (hipFuncSetAttribute, once per kernel and device).  No other environment variable
 *     is read by the library;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - dense matrices are row-major fp32 with an explicit leading dimension in ELEMENTS;
 *   - sparse structure is CSR; index arrays are int32 or int64, selected by `index_bits` (32|64).
 *     The exported structure is always int64 (torch-sparse convention); int32 is an internal
 *     narrowing the host performs when nnz and N fit;
 *   - return value: 0 on success, a negative EGNN_E* code otherwise (nothing enqueued on error,
 *     except EGNN_ELAUNCH which reports hipGetLastError() after a launch).
 */
#ifndef EGNN_HIP_H
#define EGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGNN_OK 0
#define EGNN_EINVAL (-1)   /* bad argument (null pointer, negative size, unsupported enum) */
#define EGNN_ELAUNCH (-2)  /* the HIP runtime reported a launch error */
#define EGNN_EWORKSPACE (-3) /* caller-provided workspace too small */
#define EGNN_EALIGN (-4)   /* pointer / leading dimension not aligned as the entry point requires */

#define EGNN_ABI_VERSION 6
int egnn_abi_version(void);
const char* egnn_error_string(int code);
/* Number of distinct kernels-families compiled in; used by the loader's self check. */
int egnn_build_info(char* buf, size_t buf_bytes);

/* ------------------------------------------------------------------------------------------------
 * Neighbour aggregation: Y[i,:] = REDUCE_{e in row i} val[e] * src_scale[col[e]] * X[col[e],:]  (+ bias)
 *
 * Replaces torch_sparse::spmm (K1/K2 in SURVEY.md 2.2), reached from
 *   GCNConv.forward      /root/reference/arxiv_pyg/gnn.py:47,52   (reduce=sum, val = gcn_norm)
 *   SAGEConv.forward     /root/reference/arxiv_pyg/gnn.py:79,84   (reduce=mean, val = NULL)
 *   adj_t.matmul(mean)   /root/reference/mag_pyg/gnn.py:162
 *   loss.backward()      /root/reference/arxiv_pyg/gnn.py:192     (same kernel on the transposed CSR)
 *
 * reduce: EGNN_SUM, EGNN_MEAN (sum / max(stored entries of the row, 1)), EGNN_MAX (argmax = index of
 * the FIRST maximal stored entry in CSR order, -1 and 0.0 for an empty row).
 * val        [nnz]    nullable (= 1.0)
 * src_scale  [n_src]  nullable; per-source-row factor, used for the backward of mean
 *                     (dX = A^T (dY / cnt)) without materialising a per-entry value array
 * bias       [K]      nullable; added to every written row (GCNConv `out += bias`, gnn.py:47); not with EGNN_MAX
 * argmax     [n_rows, K] int64, required for EGNN_MAX, ignored otherwise
 * Row schedule (built once per sparsity structure by the caller; integer preprocessing):
 *   short_rows [n_short]  row ids whose entry count is small; a wavefront walks 64/G of them at once (one
 *                         G-lane sub-group per row, G = lanes per 128-byte column slice), so list neighbours
 *                         with similar lengths next to each other
 *   mid_rows   [n_mid]    row ids reduced by one wavefront each
 *   long_rows  [n_long]   row ids reduced by a whole 16-wave workgroup with a fixed-order LDS combine
 * A call writes exactly the rows it lists (each at most once); with all three lists NULL every row of
 * [0, n_rows) takes the one-wavefront path.  The host normally covers all rows with one call, or with two calls
 * on two streams (short rows | mid + long rows) so the few heavy rows overlap the bulk.  The accumulation order
 * is fixed by the schedule, so results are run-to-run bit-stable for every degree distribution.
 * ---------------------------------------------------------------------------------------------- */
#define EGNN_SUM 0
#define EGNN_MEAN 1
#define EGNN_MAX 2

int egnn_spmm_csr_f32(int64_t n_rows, int64_t n_src, int64_t K,
                      const void* rowptr, const void* col, int index_bits,
                      const float* val, const float* src_scale, const float* bias,
                      const float* X, int64_t ldx, float* Y, int64_t ldy,
                      int reduce, int64_t* argmax,
                      const int64_t* short_rows, int64_t n_short, const int64_t* mid_rows, int64_t n_mid,
                      const int64_t* long_rows, int64_t n_long, void* stream);

/* The same aggregation (EGNN_SUM / EGNN_MEAN) under a SEGMENT schedule: every row is cut into entry ranges of at most
 * ~64 entries and ALL ranges go through the sub-group-per-row kernel, so the few hub rows of a power-law graph (8 % of the
 * entries in 138 rows on the arxiv-shaped graph) get the same parallelism as the bulk instead of one workgroup each.
 *   seg       [n_seg,3] int64: (first entry, end entry, destination); destination < n_rows: that row of Y is written
 *             directly (the row's only range); destination >= n_rows: slot (destination - n_rows) of `partial`
 *   comb_rows [n_comb], comb_ptr [n_comb+1]: row r = comb_rows[i] is the sum of partial slots comb_ptr[i] .. comb_ptr[i+1]-1,
 *             added in slot order (fixed => bit-stable), then scaled (mean) and biased like a direct row
 *   partial   [partial_slots, K] fp32 workspace (16-byte aligned)
 * Requires K % 4 == 0 and 16-byte aligned X / Y / bias (else EGNN_EALIGN: use egnn_spmm_csr_f32).  Every row of
 * [0, n_rows) must be covered exactly once by the schedule (rows with no entries as an empty direct range). */
int egnn_spmm_csr_seg_f32(int64_t n_rows, int64_t n_src, int64_t K,
                          const void* rowptr, const void* col, int index_bits,
                          const float* val, const float* src_scale, const float* bias,
                          const float* X, int64_t ldx, float* Y, int64_t ldy, int reduce,
                          const int64_t* seg, int64_t n_seg, const int64_t* comb_rows, const int64_t* comb_ptr, int64_t n_comb,
                          float* partial, int64_t partial_slots, void* stream);

/* The same aggregation (EGNN_SUM / EGNN_MEAN) under the ROW-BLOCK schedule (csrc/spmm_blk.hip), the default of the host
 * layer: ONE launch walks, per 128-byte column slice, first the hub segments and then blocks of `rows_per_blk`
 * consecutive rows (or rows [blk_ptr[b], blk_ptr[b+1]) when blk_ptr != NULL; every block at most rows_per_blk rows);
 * int32 indices; any K % 4 == 0; X addressed through a 32-bit buffer descriptor (n_src * ldx * 4 < 2^31 bytes, else
 * EGNN_EALIGN).
 *   hub_seg    [n_hub_seg,4] int32 (first entry, end entry, partial slot, 0): the entry ranges (<= seg_max entries each) of
 *              the rows with MORE than seg_max entries; their sums go to partial[slot,:] ([slots,K] fp32, 16-byte aligned)
 *              and the caller finishes those rows with egnn_spmm_combine_f32.
 *              Rows with at most seg_max entries are written here, completely.
 *   win        nullable [n_rows,2] int32 from egnn_spmm_blk_window_i32: entries [win[2r], win[2r+1]) of row r have their
 *              source inside row r's own block.  When given (square adjacency, X rows in node order, rows_per_blk a multiple
 *              of 128, <= 512) the block's X rows are streamed into LDS (LDS-DMA) while the out-of-block entries are
 *              gathered, and the in-block entries read LDS instead of L2 (graphs in a locality order).
 *   addend     nullable [n_rows, ld_addend] fp32: added to every row this call stores (Y = A X + addend): the sharded run
 *              aggregates its local and its halo columns in two calls, the second one accumulating onto the first
 *   stat_part  nullable [egnn_spmm_blk_stat_rows(n_rows, rows_per_blk, win != NULL), 2, K] fp32: per WAVE of every row block
 *              sum_r (y_r - shift) and sum_r (y_r - shift)^2 over the rows that wave stored -- BatchNorm statistics in the
 *              aggregation epilogue (gnn.py:47-48), finished by egnn_bn_stats_merge_f32 (n_blk = that row count);
 *              stat_shift: nullable [K] (any vector near the column means, e.g. BatchNorm's running_mean; exactness does
 *              not depend on it, only the conditioning of the variance)
 *   flags      bit 1 (2): non-temporal Y stores; bit 2 (4): write-through (sc1) Y stores (tuning knobs, results are identical);
 *              bit 3 (8): Y = max(Y, 0) on the way out -- ReLU fused into the store (an eval-mode BatchNorm folded into the
 *              weights / bias by egnn_bn_fold_f32 + ReLU, gnn.py:47-49 under model.eval()); not together with stat_part */
int egnn_spmm_csr_blk_f32(int64_t n_rows, int64_t n_src, int64_t K,
                          const int32_t* rowptr, const int32_t* col, const float* val, const float* src_scale, const float* bias,
                          const float* X, int64_t ldx, float* Y, int64_t ldy, int reduce,
                          int seg_max, int rows_per_blk, const int32_t* blk_ptr, int64_t n_blk, const int32_t* win,
                          const int32_t* hub_seg, int64_t n_hub_seg, float* partial,
                          const float* addend, int64_t ld_addend,
                          float* stat_part, const float* stat_shift, int flags, void* stream);
int64_t egnn_spmm_blk_stat_rows(int64_t n_rows, int rows_per_blk, int lds);
int egnn_spmm_blk_window_i32(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int rows_per_blk,
                             const int32_t* blk_ptr, int64_t n_blk, int32_t* win, void* stream);
/* mean[c], var[c] (biased) of n_total rows of Y: the block partials of egnn_spmm_csr_blk_f32 plus the rows listed in
 * extra_rows (the hub rows, whose Y rows the combine step wrote), read from Y.  Fixed summation order (deterministic).
 * ws: egnn_bn_stats_merge_ws_floats(C) floats (needed when n_blk > 1024: the partials are folded in two short launches). */
size_t egnn_bn_stats_merge_ws_floats(int64_t C);
int egnn_bn_stats_merge_f32(const float* stat_part, int64_t n_blk, int64_t C, const float* Y, int64_t ldy,
                            const int64_t* extra_rows, int64_t n_extra, const float* stat_shift, int64_t n_total,
                            float* mean, float* var, float* ws, size_t ws_floats, void* stream);

/* The combine step on its own -- the hub rows of egnn_spmm_csr_blk_f32: Y[r] = (sum of partial slots comb_ptr[i] .. comb_ptr[i+1]-1
 * of row r = comb_rows[i], added in slot order) * (1 / rowcount for EGNN_MEAN) + bias (+ addend[r], nullable).  K % 4 == 0,
 * 16-byte aligned rows.
 * stat_part != NULL: row (stat_base + i) of the [*, 2, K] statistics partials receives (y - shift) and (y - shift)^2 of
 * combined row i (one partial row per hub row, folded by egnn_bn_stats_merge_f32 together with the block kernel's).
 * flags: bit 3 (8) = ReLU on the way out, as in egnn_spmm_csr_blk_f32. */
int egnn_spmm_combine_f32(int64_t n_rows, int64_t K, const void* rowptr, int index_bits, const float* bias, float* Y, int64_t ldy,
                          int reduce, const int64_t* comb_rows, const int64_t* comb_ptr, int64_t n_comb, const float* partial,
                          const float* addend, int64_t ld_addend, float* stat_part, int64_t stat_base, const float* stat_shift,
                          int flags, void* stream);

/* Backward of EGNN_MAX: dX[col[argmax[i,k]], k] += val * dY[i,k].  dX must be zero-filled by the
 * caller.  Uses float atomics (the only entry point that does); max-aggregation is never exercised
 * by the reference and is provided because north_star names it. */
int egnn_spmm_csr_max_bwd_f32(int64_t n_rows, int64_t K, const void* col, int index_bits, const float* val,
                              const int64_t* argmax, const float* dY, int64_t ldy, float* dX, int64_t ldx,
                              void* stream);

/* Algorithmic (compulsory) HBM bytes of one egnn_spmm_csr_f32 call, SURVEY.md 8(d):
 * 4*n_src*K (read X) + 4*n_rows*K (write Y) + nnz*(index bytes + value bytes) + rowptr. */
int64_t egnn_spmm_algorithmic_bytes(int64_t n_rows, int64_t n_src, int64_t K, int64_t nnz, int index_bits,
                                    int has_val);

/* ------------------------------------------------------------------------------------------------
 * Graph structure (integer work is bit-exact vs torch-sparse semantics)
 * ---------------------------------------------------------------------------------------------- */
/* Edge list -> CSR sorted by (row, col), on the device (hipCUB radix sort over the bits the keys use; bit-exact).
 *   symmetric == 0: T.ToSparseTensor() (/root/reference/arxiv_pyg/gnn.py:237): duplicates are KEPT;
 *                   pass row = edge targets, col = edge sources; col_out holds E entries.
 *   symmetric != 0: SparseTensor.to_symmetric() (gnn.py:240): union of (r,c) and (c,r), duplicates merged;
 *                   col_out needs room for 2*E entries.
 * rowptr [N+1], nnz_out [1] (device scalar: the entry count, = E when symmetric == 0).  ws: egnn_csr_from_coo_ws_bytes.
 * Requires N < 3 037 000 499 (row * N + col in int64) and fewer than 2^31 keys. */
size_t egnn_csr_from_coo_ws_bytes(int64_t E, int64_t N, int symmetric);
int egnn_csr_from_coo_i64(const int64_t* row, const int64_t* col, int64_t E, int64_t N, int symmetric,
                          int64_t* rowptr, int64_t* col_out, int64_t* nnz_out, void* ws, size_t ws_bytes, void* stream);

/* CSR -> CSC: colptr [n_cols+1], row_out [nnz] (the rows of each column, ascending) and perm [nnz] = the csr2csc
 * permutation torch-sparse caches for the backward SpMM (stable sort of the entries by column; SURVEY 9.6,
 * /root/reference/arxiv_pyg/gnn.py:192): value_csc = value[perm].  ws: egnn_csr_transpose_ws_bytes(nnz, n_cols). */
size_t egnn_csr_transpose_ws_bytes(int64_t nnz, int64_t n_cols);
int egnn_csr_transpose_i64(const int64_t* rowptr, const int64_t* col, int64_t n_rows, int64_t n_cols, int64_t nnz,
                           int64_t* colptr, int64_t* row_out, int64_t* perm, void* ws, size_t ws_bytes, void* stream);

/* rowptr[i] = #entries with row < i, for i in [0, n_rows]; `row` sorted ascending.
 * Replaces torch_sparse ind2ptr inside T.ToSparseTensor()  /root/reference/arxiv_pyg/gnn.py:237. */
int egnn_rowptr_from_sorted_rows_i64(const int64_t* row, int64_t nnz, int64_t n_rows, int64_t* rowptr, void* stream);

/* int64 -> int32 narrowing of an index array; *overflow (device int32, caller-zeroed) is set to 1
 * if any element does not fit. */
int egnn_narrow_i64_to_i32(const int64_t* src, int64_t n, int32_t* dst, int32_t* overflow, void* stream);

/* gcn_norm on a value-less, row-sorted CSR (GCNConv first forward, /root/reference/arxiv_pyg/gnn.py:28-35;
 * SURVEY 9.3): A^ = D^-1/2 (A + I) D^-1/2 with fill_diag(1) replacing any stored diagonal.
 *  step 1: out_count[i] = #off-diagonal entries of row i + 1                 (caller scans -> rowptr_out)
 *  step 2: writes col_out (diagonal inserted at its sorted slot) and dinv[i] = deg^-1/2 (inf -> 0)
 *  step 3: val_out[e] = (1 * dinv[row(e)]) * dinv[col_out[e]]
 */
int egnn_gcn_norm_count_i64(const int64_t* rowptr, const int64_t* col, int64_t n, int64_t* out_count, void* stream);
int egnn_gcn_norm_fill_i64(const int64_t* rowptr, const int64_t* col, int64_t n, const int64_t* rowptr_out,
                           int64_t* col_out, float* dinv, void* stream);
int egnn_gcn_norm_values_i64(const int64_t* rowptr_out, const int64_t* col_out, int64_t n, const float* dinv,
                             float* val_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense fp32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32 == fmaf chain).
 * C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] (+ bias[N]) ; op = identity or transpose.
 *   trans_a = 0: A is [M,K] row-major (lda >= K);  trans_a = 1: A is stored [K,M] (lda >= M)
 *   trans_b = 0: B is [K,N] row-major (ldb >= N);  trans_b = 1: B is stored [N,K] (ldb >= K)
 * Replaces cuBLAS SGEMM under torch.matmul / nn.Linear (K4): GCNConv `x @ W`
 * (/root/reference/arxiv_pyg/gnn.py:47), SAGEConv lin_l/lin_r (:79), projection heads (:296-306) and
 * their backward (dX = dY W^T, dW = X^T dY).
 * split_k > 1 writes split_k partial products into `ws` ([split_k, M, N] floats) and reduces them in
 * a fixed order (deterministic); ws may be NULL when split_k <= 1.
 * ---------------------------------------------------------------------------------------------- */
/* Workspace floats a call with these arguments needs (split-K partials; the row-chunk partials of the class-count-wide dW
 * form of csrc/gemm_skinny.hip).  0 = none. */
size_t egnn_gemm_ws_floats(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, int split_k);
int egnn_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha,
                  const float* A, int64_t lda, const float* B, int64_t ldb,
                  const float* bias, float* C, int64_t ldc,
                  int split_k, float* ws, size_t ws_bytes, void* stream);

/* egnn_gemm_f32 with an epilogue option: flags bit 0 (1) = C = max(alpha op(A) op(B) + bias, 0) -- the ReLU that follows an
 * eval-mode BatchNorm whose scale / shift were folded into B and bias (egnn_bn_fold_f32). */
int egnn_gemm_ex_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                     const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc, int split_k, float* ws,
                     size_t ws_bytes, int flags, void* stream);
/* egnn_gemm_f32 with an accumulating store: C = alpha op(A) op(B) + bias + addend (addend [M,N], ld_addend >= N, nullable; may be C
 * itself only if nothing else reads it).  SAGEConv's lin_l(agg) + lin_r(x) (/root/reference/arxiv_pyg/gnn.py:79-84 via PyG SAGEConv.forward,
 * SURVEY 9.5) and the two-path input gradient of its backward become one store each instead of a GEMM + an element-wise pass. */
int egnn_gemm_add_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                      const float* B, int64_t ldb, const float* bias, const float* addend, int64_t ld_addend, float* C, int64_t ldc,
                      int split_k, float* ws, size_t ws_bytes, void* stream);
/* The same GEMM with a row gather fused into one operand's load, for the `feat[train_idx]` gathers in front of the
 * projection heads (/root/reference/arxiv_pyg/gnn.py:150-156: `out_feat[train_idx]`, `teacher_out_feat[train_idx]`,
 * the latter 273 MB read + written per step in the reference):
 *   a_rows [M] (requires trans_a == 0): row m of op(A) is row a_rows[m] of the matrix at A        (y = x[idx] W^T)
 *   b_rows [K] (requires trans_b == 0): row k of B      is row b_rows[k] of the matrix at B        (dW = dY^T x[idx])
 * At most one of them; both NULL = egnn_gemm_f32.  Ids must be valid row numbers (not checked on the device). */
int egnn_gemm_rows_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha,
                       const float* A, int64_t lda, const int64_t* a_rows, const float* B, int64_t ldb, const int64_t* b_rows,
                       const float* bias, float* C, int64_t ldc,
                       int split_k, float* ws, size_t ws_bytes, void* stream);

/* C [M, N] = alpha A^T B[b_rows] for a CONSTANT B (dW = dY^T x[idx] of a Linear over a constant input: the teacher projection head
 * of /root/reference/arxiv_pyg/gnn.py:296-306, whose input -- the teacher's [N, 750] features -- never changes): B's gathered rows
 * are cut ONCE into tile-packed bf16 planes with the gathered row index as the reduction dimension (egnn_gemm_tn_planes_pack_f32;
 * egnn_gemm_tn_planes_bytes of 1 KB-aligned device memory, 6 bytes per element); every later product reads A [K, M] (lda >= M,
 * 16-byte aligned rows, M % 256 == 0) down LDS columns against those planes -- no gather and no operand cut in the loop.  Same
 * six-product fp32 arithmetic as egnn_gemm_f32 on the bf16 pipe; fixed-order split over k (workspace egnn_gemm_tn_planes_ws_floats).
 * EGNN_EALIGN: shape / alignment not taken, or EGNN_GEMM_PIPE=f32 (callers then use egnn_gemm_rows_f32). */
size_t egnn_gemm_tn_planes_bytes(int64_t N, int64_t K);
int egnn_gemm_tn_planes_pack_f32(const float* B, int64_t ldb, const int64_t* b_rows, int64_t N, int64_t K, void* planes,
                                 size_t planes_bytes, void* stream);
size_t egnn_gemm_tn_planes_ws_floats(int64_t M, int64_t N, int64_t K);
int egnn_gemm_tn_planes_f32(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const void* planes, float* C,
                            int64_t ldc, float* ws, size_t ws_floats, void* stream);

/* The forward of the same Linear, C [M, N] = alpha A[a_rows] B^T + bias (B stored [N, K] as nn.Linear keeps its weight; N % 128 == 0):
 * the CONSTANT A's gathered rows are cut once into planes (row blocks of 128 gathered rows x k-steps of 16: the gather is baked in),
 * B is cut per call into the workspace (egnn_gemm_rows_planes_ws_bytes); both operand tiles then travel global -> LDS as lane-linear
 * DMA copies with no VALU work in the loop.  Same arithmetic and return codes as above. */
size_t egnn_gemm_rows_planes_bytes(int64_t M, int64_t K);
int egnn_gemm_rows_planes_pack_f32(const float* A, int64_t lda, const int64_t* a_rows, int64_t M, int64_t K, void* planes,
                                   size_t planes_bytes, void* stream);
size_t egnn_gemm_rows_planes_ws_bytes(int64_t N, int64_t K);
int egnn_gemm_rows_planes_f32(int64_t M, int64_t N, int64_t K, float alpha, const void* planes_a, const float* B, int64_t ldb,
                              const float* bias, float* C, int64_t ldc, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused losses (K5-K7).  Every loss entry point comes as fwd (scalars out) + bwd (input grads out);
 * upstream gradients are passed as DEVICE scalars so that no host sync is needed.
 * ---------------------------------------------------------------------------------------------- */

/* Row-wise cross entropy + logit-KD, /root/reference/arxiv_pyg/criterion.py:8-21 (and the
 * `loss_cls = F.cross_entropy(logits, labels)` first line of every criterion, :26,41,60,98,132).
 * Row i of the loss reads row r = rows ? rows[i] : i of logits / teacher and labels[r]: with a row list the
 * `out[train_idx]`, `y[train_idx]`, `teacher_logits[train_idx]` gathers of train() (gnn.py:107-116) are fused into the
 * operand loads (logits / teacher / labels are then the FULL [N,*] arrays).  A label outside [0, C) is ignored the
 * way F.cross_entropy ignores ignore_index (no loss, no gradient, not counted in the mean); it is never an address.
 *   out3[0] = mean over valid-label rows of CE(logits_r, labels_r)
 *   out3[1] = sum_{i,c} p_t (log p_t - log q) / (n*C)   (F.kl_div default reduction='mean'), q = softmax(logits/T),
 *             p_t = softmax(teacher/T); 0 when teacher == NULL
 *   out3[2] = number of valid-label rows (the backward's CE denominator)
 * partials: workspace of egnn_ce_kd_ws_floats(n) floats. */
size_t egnn_ce_kd_ws_floats(int64_t n);
int egnn_ce_kd_fwd_f32(const float* logits, int64_t ld_logits, const float* teacher, int64_t ld_teacher,
                       const int64_t* labels, const int64_t* rows, int64_t n, int64_t C, float T,
                       float* out3, float* partials, void* stream);
/* dlogits[r,c] = g_cls * (softmax(logits_r)_c - [c == label_r]) / out3[2]  +  g_kd * (q_rc - p_t,rc) / (T * n * C)
 * g_cls / g_kd: device scalars (nullable = 0); out3: the forward's output.  With a row list, dlogits is the full
 * [n_total_rows, C] block: it is cleared first and only the listed rows are written (the scatter of the
 * `out[train_idx]` backward). */
int egnn_ce_kd_bwd_f32(const float* logits, int64_t ld_logits, const float* teacher, int64_t ld_teacher,
                       const int64_t* labels, const int64_t* rows, int64_t n_total_rows, int64_t n, int64_t C, float T,
                       const float* out3, const float* g_cls, const float* g_kd, float* dlogits, int64_t ld_dlogits,
                       void* stream);

/* Gather + L2 row normalisation:  out[i,:] = x[idx[i],:] / max(||x[idx[i],:]||_2, eps), inv_norm[i] = 1/max(..).
 * idx nullable (identity).  F.normalize in criterion.py:29-30,71-72,139-140 fused with the
 * `feat[sampled_inds]` gather (:64-65,136-137). */
int egnn_gather_normalize_rows_f32(const float* x, int64_t ldx, const int64_t* idx, int64_t n, int64_t D, float eps,
                                   float* out, int64_t ldo, float* inv_norm, void* stream);
/* Backward of the above: dx[idx[i],:] (+)= inv_norm[i] * (dout[i,:] - xhat[i,:] * <dout[i,:], xhat[i,:]>)
 * (zero where the eps clamp was active).  `accumulate` != 0 adds into dx (rows of idx are unique). */
int egnn_normalize_rows_bwd_f32(const float* xhat, int64_t ldh, const float* dout, int64_t ldd, const float* inv_norm,
                                const int64_t* idx, int64_t n, int64_t D, float eps,
                                float* dx, int64_t ldx, int accumulate, void* stream);

/* G-CRD / InfoNCE, /root/reference/arxiv_pyg/criterion.py:139-145:
 *   Z = fhat that^T / tau  ([S,S]);  loss = mean_i( logsumexp_j Z_ij - Z_ii ).
 * fwd: Z tiles are produced on the fp32 MFMA, the row log-sum-exp is accumulated online per lane and
 *      merged in a fixed order; the scores are saved in `Z` ([S,S], ld = S) for the backward when Z != NULL.
 *      lse [S] and the scalar loss are outputs.  ws: egnn_nce_ws_floats(S) floats.
 *      unit_rows != 0 asserts ||fhat_i|| = ||that_j|| = 1 (what criterion.py:139-140 guarantees): logits are then
 *      bounded by 1/tau and that bound replaces the running row maximum (one exp per element, half the state);
 *      with unit_rows == 0 (or tau < 0.025) the general online-max form runs.
 *      What `Z` holds: the logits Z_ij in the general form; E_ij = exp(Z_ij - 1.0001/tau) in the unit-rows form
 *      (egnn_nce_saves_exp(tau, unit_rows) tells which) -- the forward needs E for the row sums anyway, and with
 *      w_i = exp(1.0001/tau - lse_i) the backward becomes two exp-free GEMMs on E.
 * bwd: dfhat = g/(S tau) (P - I) that ; dthat = g/(S tau) (P - I)^T fhat, P = exp(Z - lse) = diag(w) E; g device
 *      scalar.  Requires the `Z` written by fwd and the SAME tau / unit_rows.
 *      ws (nullable): egnn_nce_bwd_ws_floats(S, S, P) floats.  With it the two GEMMs (reduction length S, only
 *      S*P outputs) split their reduction over enough workgroups to fill the chip and sum the partials in a fixed
 *      order; without it they fall back to smaller row tiles. */
size_t egnn_nce_ws_floats(int64_t S);
size_t egnn_nce_bwd_ws_floats(int64_t Sr, int64_t Sc, int64_t P);
int egnn_nce_saves_exp(float tau, int unit_rows);
int egnn_nce_fwd_f32(const float* fhat, const float* that, int64_t S, int64_t P, int64_t ld, float tau, int unit_rows,
                     float* Z, float* lse, float* loss, float* ws, size_t ws_floats, void* stream);
int egnn_nce_bwd_f32(const float* fhat, const float* that, int64_t S, int64_t P, int64_t ld, float tau, int unit_rows,
                     const float* Z, const float* lse, const float* g,
                     float* dfhat, float* dthat, float* ws, size_t ws_floats, void* stream);

/* Row-block form of the two entry points above, for node-range sharding: this rank owns Sr of the
 * S_total sampled rows, `that` holds ALL Sc = S_total teacher rows (all-gathered over RCCL), and the positive of
 * local row i is column i + diag_off.  loss = inv_count * sum_i (lse_i - Z_i,i+off)  (pass 1/S_total and all-reduce);
 * dfhat [Sr,P] is complete, dthat [Sc,P] is this rank's contribution (all-reduce it); scale = 1 / (S_total * tau).
 * Z is [Sr,Sc] (ld = Sc); fwd ws: egnn_nce_ws_floats(Sr); bwd ws (nullable): egnn_nce_bwd_ws_floats(Sr, Sc, P). */
int egnn_nce_block_fwd_f32(const float* fhat, int64_t ld_f, const float* that, int64_t ld_t, int64_t Sr, int64_t Sc,
                           int64_t diag_off, int64_t P, float tau, float inv_count, int unit_rows, float* Z, float* lse,
                           float* loss, float* ws, size_t ws_floats, void* stream);
int egnn_nce_block_bwd_f32(const float* fhat, int64_t ld_f, const float* that, int64_t ld_t, int64_t Sr, int64_t Sc,
                           int64_t diag_off, int64_t P, float tau, float scale, int unit_rows, const float* Z,
                           const float* lse, const float* g, float* dfhat, int64_t ld_df, float* dthat, int64_t ld_dt,
                           float* ws, size_t ws_floats, void* stream);

/* GSP all-pairs similarity loss, /root/reference/arxiv_pyg/criterion.py:69-88:
 *   loss = mean_ij (k(xs_i,xs_j) - k(xt_i,xt_j))^2 ; kernel: EGNN_K_COSINE / _POLY (rows must be unit vectors,
 *   criterion.py:71-72,76-77) or EGNN_K_L2 / _RBF (raw rows; ||a-b||^2 via ||a||^2+||b||^2-2<a,b>, diagonal exact 0;
 *   the reference's [S,S,D] difference tensor is never formed).
 * Writes Ws, Wt ([S,S], ld = S) such that, for an upstream gradient g,
 *   dxs = g * (Ws xs - rowsum(Ws) o xs),  dxt = g * (Wt xt - rowsum(Wt) o xt)   (rowsum term for L2/RBF only),
 * i.e. one egnn_gemm_f32 + egnn_rowsum_f32 + egnn_scale_rowcorr_f32 per side.  ws: egnn_gsp_ws_floats(S). */
#define EGNN_K_COSINE 0
#define EGNN_K_POLY 1
#define EGNN_K_L2 2
#define EGNN_K_RBF 3
size_t egnn_gsp_ws_floats(int64_t S);
int egnn_gsp_fwd_f32(const float* xs, int64_t ld_s, int64_t Ps, const float* xt, int64_t ld_t, int64_t Pt, int64_t S,
                     int kernel, float* Ws, float* Wt, float* loss, float* ws, size_t ws_floats, void* stream);
/* out[i] = sum_j W[i,j] (fixed order) */
int egnn_rowsum_f32(const float* W, int64_t ld, int64_t n, int64_t m, float* out, void* stream);
/* out[i,:] = g * (WX[i,:] - r[i] * X[i,:]);  r nullable (no correction), g nullable device scalar (= 1) */
int egnn_scale_rowcorr_f32(const float* WX, int64_t ldw, const float* X, int64_t ldx, const float* r, const float* g,
                           int64_t n, int64_t P, float* out, int64_t ldo, void* stream);

/* PPI multi-label logit KD, /root/reference/ppi_pyg/criterion.py:11-13: out2[0] = mean BCEWithLogits(logits, labels),
 * out2[1] = mean BCEWithLogits(logits, sigmoid(teacher)) over all `total` = n*C elements (contiguous arrays). */
size_t egnn_bce_pair_ws_floats(void);
int egnn_bce_pair_fwd_f32(const float* logits, const float* labels, const float* teacher, int64_t total, float* out2,
                          float* ws, void* stream);
int egnn_bce_pair_bwd_f32(const float* logits, const float* labels, const float* teacher, int64_t total,
                          const float* g_cls, const float* g_kd, float* dlogits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LSP building blocks, /root/reference/arxiv_pyg/criterion.py:95-126 (K5/K6 in SURVEY.md 2.2)
 * ---------------------------------------------------------------------------------------------- */
/* Per-edge similarity of rows a = F[idx_a[e]], b = F[idx_b[e]] (never materialising feat[src] / feat[dst]):
 *   EGNN_K_COSINE  <a,b> / sqrt(max(|a|^2 |b|^2, 1e-16))   (F.cosine_similarity, criterion.py:103)
 *   EGNN_K_POLY    cosine^2 (:106)      EGNN_K_L2  ||a-b||_2 (:109)      EGNN_K_RBF  exp(-||a-b||^2 / 2) (:112)
 * aux3 [E,3] keeps (dot,|a|^2,|b|^2) or (||a-b||^2,0,0) for the backward. */
int egnn_edge_sim_f32(const float* F, int64_t ld, int64_t D, const int64_t* idx_a, const int64_t* idx_b, int64_t E,
                      int kernel, float* sim, float* aux3, void* stream);
/* Backward coefficients: with g = dL/dsim,  dL/dF[a] += alpha*b + beta_a*a  and  dL/dF[b] += alpha*a + beta_b*b per edge;
 * the caller turns them into two egnn_spmm_csr_f32 calls (values = alpha) plus a per-row scale (segment sums of beta). */
int egnn_edge_sim_coef_f32(const float* g, const float* sim, const float* aux3, int64_t E, int kernel,
                           float* alpha, float* beta_a, float* beta_b, void* stream);
/* torch_geometric.utils.softmax over CSR segments: p = exp(x - max_seg) / (sum_seg + 1e-16); x, p in segment
 * order (seg_ptr [n_seg+1]).  bwd: gx = p * (gp - sum_seg p*gp).  No atomics, fixed order. */
int egnn_segment_softmax_fwd_f32(const int64_t* seg_ptr, const float* x, int64_t n_seg, float* p, void* stream);
int egnn_segment_softmax_bwd_f32(const int64_t* seg_ptr, const float* p, const float* gp, int64_t n_seg, float* gx, void* stream);
int egnn_segment_sum_f32(const int64_t* seg_ptr, const float* x, int64_t n_seg, float* out, void* stream);

/* Tail of lpw_criterion (/root/reference/arxiv_pyg/criterion.py:103-122) in one pass per direction: both segment softmaxes of
 * the student / teacher similarity vectors (same PyG form as above), the element-wise criterion and its mean over the E edges:
 *   criterion 0 (kld): F.kl_div(log p_s, p_t, reduction='mean') = mean_e p_t (log p_t - log p_s), 0 where p_t == 0   (:118)
 *   criterion 1 (mse): F.mse_loss(p_s, p_t)                                                                          (:120)
 * p_s, p_t [E] are written for the backward; loss [1]; ws: egnn_lsp_loss_ws_floats() floats.  Fixed summation order.
 * bwd: g [1] = dL/dloss on the device; gsim_s [E] = dL/dsim_s; gsim_t [E] nullable (= dL/dsim_t when the teacher side needs one). */
size_t egnn_lsp_loss_ws_floats(void);
int egnn_lsp_loss_fwd_f32(const int64_t* seg_ptr, const float* sim_s, const float* sim_t, int64_t n_seg, int64_t E,
                          int criterion, float* p_s, float* p_t, float* loss, float* ws, void* stream);
int egnn_lsp_loss_bwd_f32(const int64_t* seg_ptr, const float* p_s, const float* p_t, int64_t n_seg, int64_t E,
                          int criterion, const float* g, float* gsim_s, float* gsim_t, void* stream);

/* out[c] = sum_r x[r, c] of a row-major [n, C] matrix (leading dimension ld): the bias gradients of GCNConv / nn.Linear
 * (arxiv_pyg/gnn.py:28-33 modules' `bias`), fixed row stripes + fixed-order finalize.  ws: egnn_colsum_ws_floats(C) floats. */
size_t egnn_colsum_ws_floats(int64_t C);
int egnn_colsum_f32(const float* x, int64_t ld, int64_t n, int64_t C, float* out, float* ws, void* stream);

/* GAT attention coefficients (the teacher that the PPI / MAG train loops run inside the student step,
 * /root/reference/ppi_pyg/gnn.py:86-117,208-209; PyG <=1.7 GATConv.message + utils.softmax):
 *   s[e,h]   = leaky_relu(alpha_src[col[e],h] + alpha_dst[row(e),h], negative_slope)          (u_add_v SDDMM)
 *   att[h,e] = exp(s - max over the row's entries) / (sum exp(..) + 1e-16)                     (edge softmax per target)
 * rowptr / col: CSR by target, int64; alpha_src [n_src,H], alpha_dst [n_rows,H] row-major; att is HEAD-major [H,nnz]
 * so that head h's values are a contiguous per-entry array for egnn_spmm_csr_*_f32 (u_mul_e_sum).  Forward only
 * (the teacher is frozen inside the student step; teacher training is out of scope). */
int egnn_gat_attention_fwd_f32(const int64_t* rowptr, const int64_t* col, const float* alpha_src, const float* alpha_dst,
                               int64_t n_rows, int64_t nnz, int H, float negative_slope, float* att, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused BatchNorm1d (+ ReLU + dropout) over node rows -- SURVEY.md 8(f) rank 1; replaces the ATen BatchNorm /
 * threshold / fused_dropout chain at /root/reference/arxiv_pyg/gnn.py:48-50,80-82,296-306.
 * Shapes: x [n,C] fp32, C % 4 == 0, C <= 1024, ld % 4 == 0, 16-byte aligned (else EGNN_EALIGN: the host uses
 * the torch operators for such shapes).
 *   egnn_bn_stats_f32     mean[c], biased var[c] over the n rows (training statistics)
 *   egnn_bn_act_fwd_f32   y = drop_p(relu?(gamma * (x - mean) * rsqrt(var + eps) + beta)); the dropout mask is a
 *                         counter-based hash of (seed + *seed_dev, row*C + c): keep if u >= p, kept values scaled 1/(1-p);
 *                         seed_dev: nullable device scalar added to `seed` -- a per-step value that lives on the device,
 *                         so that a captured hipGraph of the step draws a fresh mask on every replay
 *   egnn_bn_act_bwd_f32   recomputes xhat / ReLU sign / mask from x and seed; dgamma, dbeta [C]; dx [n,C];
 *                         batch_stats != 0: mean/var are this batch's statistics (training backward, the
 *                         -(sum d + xhat sum d xhat)/n terms apply); 0: running statistics (eval-mode graph)
 *   egnn_bn_act_bwd_reduce_f32 / _apply_f32   the two halves of the backward, for batch statistics that span several
 *                         GPUs (node-range shards): reduce writes this shard's dbeta = sum d and
 *                         dgamma = sum d*xhat; the caller all-reduces them over RCCL and passes the totals plus
 *                         inv_count = 1 / (rows of ALL shards) to apply (inv_count = 0: running statistics)
 * ws: egnn_bn_ws_floats(C) floats. */
size_t egnn_bn_ws_floats(int64_t C);
int egnn_bn_stats_f32(const float* x, int64_t ld, int64_t n, int64_t C, float* mean, float* var, float* ws, size_t ws_floats,
                      void* stream);
int egnn_bn_act_fwd_f32(const float* x, int64_t ld, int64_t n, int64_t C, const float* mean, const float* var, float eps,
                        const float* gamma, const float* beta, int relu, float p, uint64_t seed, const uint64_t* seed_dev,
                        float* y, int64_t ldy, void* stream);
int egnn_bn_act_bwd_f32(const float* x, int64_t ld, const float* dy, int64_t ld_dy, int64_t n, int64_t C, const float* mean,
                        const float* var, float eps, const float* gamma, const float* beta, int relu, float p, uint64_t seed,
                        const uint64_t* seed_dev, int batch_stats, float* dgamma, float* dbeta, float* dx, int64_t ld_dx, float* ws,
                        size_t ws_floats, void* stream);

/* Merge the per-shard statistics of a sharded batch: stats [world, 2C+1] = (mean[C] | biased var[C] | rows) per shard
 * (all-gathered over RCCL), combined in shard order with the pairwise update of Chan et al. (identical on every rank);
 * shards with 0 rows are skipped.  Outputs mean [C], biased var [C], total [1] (rows of all shards). */
int egnn_bn_merge_shards_f32(const float* stats, int world, int64_t C, float* mean, float* var, float* total, void* stream);
int egnn_bn_act_bwd_reduce_f32(const float* x, int64_t ld, const float* dy, int64_t ld_dy, int64_t n, int64_t C,
                               const float* mean, const float* var, float eps, const float* gamma, const float* beta, int relu,
                               float p, uint64_t seed, const uint64_t* seed_dev, float* dgamma, float* dbeta, float* ws,
                               size_t ws_floats, void* stream);
int egnn_bn_act_bwd_apply_f32(const float* x, int64_t ld, const float* dy, int64_t ld_dy, int64_t n, int64_t C,
                              const float* mean, const float* var, float eps, const float* gamma, const float* beta, int relu,
                              float p, uint64_t seed, const uint64_t* seed_dev, const float* sum_dbeta, const float* sum_dgamma,
                              float inv_count, float* dx, int64_t ld_dx, void* stream);
/* The apply half that also returns the column sums of dx (dx_colsum [C]: the bias gradient of the layer in front of the BatchNorm);
 * ws: egnn_bn_ws_floats(C) floats. */
int egnn_bn_act_bwd_apply_colsum_f32(const float* x, int64_t ld, const float* dy, int64_t ld_dy, int64_t n, int64_t C,
                                     const float* mean, const float* var, float eps, const float* gamma, const float* beta, int relu,
                                     float p, uint64_t seed, const uint64_t* seed_dev, const float* sum_dbeta, const float* sum_dgamma,
                                     float inv_count, float* dx, int64_t ld_dx, float* dx_colsum, float* ws, size_t ws_floats, void* stream);

/* egnn_bn_act_bwd_f32 that also leaves dx_colsum[c] = sum over rows of dx[:,c] (nullable): the gradient of a bias added in
 * front of the BatchNorm (GCNConv / nn.Linear bias, /root/reference/arxiv_pyg/gnn.py:47-48,296-306), formed while dx is
 * written instead of by a second pass over it (ATen: gy.sum(0)).  Same workspace. */
int egnn_bn_act_bwd_colsum_f32(const float* x, int64_t ld, const float* dy, int64_t ld_dy, int64_t n, int64_t C, const float* mean,
                               const float* var, float eps, const float* gamma, const float* beta, int relu, float p, uint64_t seed,
                               const uint64_t* seed_dev, int batch_stats, float* dgamma, float* dbeta, float* dx, int64_t ld_dx,
                               float* dx_colsum, float* ws, size_t ws_floats, void* stream);

/* The fused BatchNorm + activation when only the OUTPUT ROWS `pick` (n_pick unique ids into the n rows of x) are read afterwards: the
 * projection heads feeding the sampled criteria (/root/reference/arxiv_pyg/gnn.py:296-306 -> criterion.py:62-65,134-137: the
 * statistics span all train rows, the criterion keeps max_samples of the output rows).
 *   fwd: y [n_pick, C], y[i] = act(bn(x[pick[i]]))  (mean / var: the statistics of ALL n rows, egnn_bn_stats_f32)
 *   bwd: dy [n_pick, C] -> dgamma, dbeta (sums over the picked rows: every other output row has no gradient), dx [n, C] (all rows:
 *        the mean / variance terms reach every row), dx_colsum nullable as in egnn_bn_act_bwd_colsum_f32.  Same arithmetic as
 *        scattering dy into a zero [n, C] gradient and calling egnn_bn_act_bwd_colsum_f32, without the zero rows. */
int egnn_bn_act_rows_fwd_f32(const float* x, int64_t ld, int64_t n, int64_t C, const int64_t* pick, int64_t n_pick, const float* mean,
                             const float* var, float eps, const float* gamma, const float* beta, int relu, float p, uint64_t seed,
                             const uint64_t* seed_dev, float* y, int64_t ldy, void* stream);
int egnn_bn_act_rows_bwd_f32(const float* x, int64_t ld, int64_t n, int64_t C, const int64_t* pick, int64_t n_pick, const float* dy,
                             int64_t ld_dy, const float* mean, const float* var, float eps, const float* gamma, const float* beta,
                             int relu, float p, uint64_t seed, const uint64_t* seed_dev, int batch_stats, float* dgamma, float* dbeta,
                             float* dx, int64_t ld_dx, float* dx_colsum, float* ws, size_t ws_floats, void* stream);

/* The two halves of egnn_bn_act_rows_bwd_f32 (which is their composition): on node-range shards the projection heads' BatchNorm has
 * all-rank statistics (/root/reference/arxiv_pyg/gnn.py:296-306 on shards: dist.SyncBatchNorm1d), so [sum d | sum d xhat] is all-reduced
 * between them.
 *   reduce: dgamma = sum d xhat, dbeta = sum d over the picked rows of THIS tensor (same workspace as egnn_bn_ws_floats).
 *   apply:  dx [n, C] from the sums the mean / variance terms use (sum_dbeta / sum_dgamma with inv_count = 1 / rows they span; the
 *           all-rank sums already divided by the row total go with inv_count = 1), n_pick >= 0; dx_colsum nullable -- it needs
 *           local_dbeta, this tensor's own sum d, because the picked rows' own term is local. */
int egnn_bn_act_rows_bwd_reduce_f32(const float* x, int64_t ld, int64_t n, int64_t C, const int64_t* pick, int64_t n_pick, const float* dy,
                                    int64_t ld_dy, const float* mean, const float* var, float eps, const float* gamma, const float* beta,
                                    int relu, float p, uint64_t seed, const uint64_t* seed_dev, float* dgamma, float* dbeta, float* ws,
                                    size_t ws_floats, void* stream);
int egnn_bn_act_rows_bwd_apply_f32(const float* x, int64_t ld, int64_t n, int64_t C, const int64_t* pick, int64_t n_pick, const float* dy,
                                   int64_t ld_dy, const float* mean, const float* var, float eps, const float* gamma, const float* beta,
                                   int relu, float p, uint64_t seed, const uint64_t* seed_dev, const float* sum_dbeta,
                                   const float* sum_dgamma, float inv_count, const float* local_dbeta, float* dx, int64_t ld_dx,
                                   float* dx_colsum, float* ws, size_t ws_floats, void* stream);

/* Backward of  h = act(bn(x)) [M, C]  followed by the narrow Linear  h W  (W [C, Ks] for w_kmajor = 0, [Ks, C] rows for 1 -- the same
 * flag as egnn_bn_act_linear_fwd_f32; Ks <= 64, C % 64 == 0)
 * in one pass over the [M, C] tensors -- the last hidden layer of the students (/root/reference/arxiv_pyg/gnn.py:47-52 under loss.backward()):
 *   dh = alpha G W^T (+ addend, nullable dense [M, C]) (+ add_rows[add_inv[row]] where add_inv[row] >= 0: the row-compact input gradient of
 *   the projection head, gnn.py:150; add_inv int32 [M], -1 = no row), d = dh * gate(x) never stored as dh; dgamma / dbeta / dx / dx_colsum as
 *   egnn_bn_act_bwd_colsum_f32 would return them for dy = dh.  ws: egnn_skinny_dx_bn_ws_floats(M, C) floats. */
/* Forward of the same pair in one pass over x: h = act(bn(x)) [n, C] is stored AND multiplied by the narrow W ([C, Ks] for w_kmajor = 0 /
 * [Ks, C] rows for 1) while its 16-byte pieces are in registers: xw = h W [n, Ks].  h is bit-identical to egnn_bn_act_fwd_f32.
 * EGNN_EALIGN when the shape is not taken (Ks > 64, C % 16 != 0, C > 1024): the caller then makes the two calls. */
int egnn_bn_act_linear_fwd_f32(const float* x, int64_t ld, int64_t n, int64_t C, const float* mean, const float* var, float eps,
                               const float* gamma, const float* beta, int relu, float p, uint64_t seed, const uint64_t* seed_dev,
                               const float* W, int64_t ldw, int w_kmajor, int64_t Ks, float* h, int64_t ldh, float* xw, int64_t ld_xw,
                               void* stream);
size_t egnn_skinny_dx_bn_ws_floats(int64_t M, int64_t C);
int egnn_skinny_dx_bn_bwd_f32(const float* G, int64_t ldg, const float* W, int64_t ldw, int w_kmajor, int64_t M, int64_t C, int64_t Ks,
                              float alpha, const float* addend, int64_t ld_addend, const float* add_rows, int64_t ld_add_rows,
                              const int32_t* add_inv, const float* x, int64_t ldx, const float* mean, const float* var, float eps,
                              const float* gamma, const float* beta, int relu, float p, uint64_t seed, const uint64_t* seed_dev,
                              int batch_stats, float* dgamma, float* dbeta, float* dx, int64_t ld_dx, float* dx_colsum, float* ws,
                              size_t ws_floats, void* stream);
/* The same backward in its two halves, for BatchNorm statistics that span several node-range shards (SyncBN semantics):
 * `reduce` leaves d = dh * gate in dx and THIS shard's (sum d, sum d xhat) in dbeta / dgamma; the caller all-reduces them; `apply` then
 * turns the stored d into dx = gamma rstd (d - (sum_dbeta + xhat sum_dgamma) inv_count) in place (dx_colsum nullable: column sums of dx,
 * ws of egnn_bn_ws_floats(C) floats then).  egnn_skinny_dx_bn_bwd_f32 = reduce + apply with the local sums and inv_count = 1 / M. */
int egnn_skinny_dx_bn_bwd_reduce_f32(const float* G, int64_t ldg, const float* W, int64_t ldw, int w_kmajor, int64_t M, int64_t C,
                                     int64_t Ks, float alpha, const float* addend, int64_t ld_addend, const float* add_rows,
                                     int64_t ld_add_rows, const int32_t* add_inv, const float* x, int64_t ldx, const float* mean,
                                     const float* var, float eps, const float* gamma, const float* beta, int relu, float p, uint64_t seed,
                                     const uint64_t* seed_dev, float* dgamma, float* dbeta, float* dx, int64_t ld_dx, float* ws,
                                     size_t ws_floats, void* stream);
int egnn_bn_bwd_apply_stored_f32(const float* x, int64_t ldx, int64_t M, int64_t C, const float* mean, const float* var, float eps,
                                 const float* gamma, const float* beta, int relu, float p, uint64_t seed, const uint64_t* seed_dev,
                                 const float* sum_dbeta, const float* sum_dgamma, float inv_count, float* dx, int64_t ld_dx,
                                 float* dx_colsum, float* ws, size_t ws_floats, void* stream);

/* nn.BatchNorm1d's training-step state update in one launch (torch/nn/modules/batchnorm.py: num_batches_tracked += 1,
 * running = (1 - m) running + m stat with the unbiased variance n/(n-1) var):  mean / var [C] = this batch's statistics,
 * momentum < 0 = cumulative average (momentum=None); num_batches_tracked: nullable device int64. */
int egnn_bn_running_update_f32(const float* mean, const float* var, int64_t C, int64_t n, float momentum, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, void* stream);
/* The same update with the row count read from device memory (total_rows [1], float): the all-rank row total of a node-range
 * sharded run (SyncBN) never visits the host. */
int egnn_bn_running_update_dev_f32(const float* mean, const float* var, int64_t C, const float* total_rows, float momentum,
                                   float* running_mean, float* running_var, int64_t* num_batches_tracked, void* stream);

/* Eval-mode BatchNorm1d folded into the layer in front of it (model.eval(): y = (x W + b - running_mean) * gamma /
 * sqrt(running_var + eps) + beta, gnn.py:47-49,198-201):  W_out[i,c] = W[i,c] * s_c,  bias_out[c] = (bias[c] - mean[c]) * s_c
 * + beta[c],  s_c = gamma[c] / sqrt(var[c] + eps).  W [rows, C] row-major (GCNConv layout [in, out]); bias / gamma / beta
 * nullable (0 / 1 / 0).  The aggregation is linear, so the fold also holds with A^ between W and the BatchNorm. */
int egnn_bn_fold_f32(const float* W, int64_t ldw, int64_t rows, int64_t C, const float* bias, const float* mean, const float* var,
                     const float* gamma, const float* beta, float eps, float* W_out, int64_t ld_out, float* bias_out, void* stream);

/* test() of /root/reference/arxiv_pyg/gnn.py:198-218 in one pass: acc3[k] = |{i : split_id[i] == k, argmax_c logits[i,c] == y[i]}|
 * / |{i : split_id[i] == k}| for the splits k = 0, 1, 2 (train / valid / test; any other id = not evaluated), as the ogb
 * Evaluator forms it (integer hit count over split size, in double).  argmax = first maximal column (torch.argmax).
 * ws: egnn_split_accuracy_ws_ints() int32. */
size_t egnn_split_accuracy_ws_ints(void);
int egnn_split_accuracy_f32(const float* logits, int64_t ld, int64_t n, int64_t C, const int64_t* y, const int8_t* split_id,
                            double* acc3, int32_t* ws, size_t ws_ints, void* stream);
/* The same pass returning the raw counts: out6 = (hits_train, hits_valid, hits_test, n_train, n_valid, n_test) as doubles -- what
 * one node-range shard contributes to the all-rank accuracies of test() (summed over ranks by an all-reduce). */
int egnn_split_counts_f32(const float* logits, int64_t ld, int64_t n, int64_t C, const int64_t* y, const int8_t* split_id,
                          double* out6, int32_t* ws, size_t ws_ints, void* stream);

/* dst[idx[r], :] += src[r, :], r < n, for UNIQUE ids: the row-compact gradient of x[idx] joins the dense gradient of x
 * (the backward of gnn.py:150 `feat[train_idx]` next to the conv's gradient) without a zero-filled [N,C] temporary. */
int egnn_rows_add_f32(float* dst, int64_t ld_dst, const int64_t* idx, const float* src, int64_t ld_src, int64_t n, int64_t C,
                      void* stream);

/* FitNet (/root/reference/arxiv_pyg/criterion.py:24-36, ppi_pyg/criterion.py:21-33): loss[0] = mean over n * D of
 * (F.normalize(f) - F.normalize(t))^2, F.normalize(x) = x / max(||x||_2, eps); one wave per row, fixed-order sums.
 * bwd: df / dt (nullable) = g[0] * dloss/df, dloss/dt.  ws: egnn_feature_loss_ws_floats(n) floats. */
size_t egnn_feature_loss_ws_floats(int64_t n);
int egnn_fitnet_fwd_f32(const float* f, int64_t ldf, const float* t, int64_t ldt, int64_t n, int64_t D, float eps, float* loss,
                        float* ws, size_t ws_floats, void* stream);
int egnn_fitnet_bwd_f32(const float* f, int64_t ldf, const float* t, int64_t ldt, int64_t n, int64_t D, float eps, const float* g,
                        float* df, int64_t lddf, float* dt, int64_t lddt, void* stream);

/* Attention transfer (criterion.py:39-54): e_s[i] = sum_d f[i,d]^2 over Df columns, e_t likewise over Dt; both length-n vectors are
 * L2-normalised ACROSS the n nodes (the reference's F.normalize on a 1-D tensor) and loss[0] = mean_i (e_s[i]/|e_s| - e_t[i]/|e_t|)^2.
 * The forward leaves e_s, e_t and its scalars in ws (same size as above); the backward reads them: keep ws until then. */
int egnn_at_fwd_f32(const float* f, int64_t ldf, int64_t Df, const float* t, int64_t ldt, int64_t Dt, int64_t n, float eps, float* loss,
                    float* ws, size_t ws_floats, void* stream);
int egnn_at_bwd_f32(const float* f, int64_t ldf, int64_t Df, const float* t, int64_t ldt, int64_t Dt, int64_t n, float eps, const float* ws,
                    const float* g, float* df, int64_t lddf, float* dt, int64_t lddt, void* stream);

/* DIAGNOSTIC (measurement only; bench.py's roofline.gather_ceiling_GBs): replays the gather stream of one aggregation call and
 * nothing else -- for every stored entry e, the 128-byte slice s of row col[e] of X [n_src, K] is read by an 8-lane sub-group,
 * slice s by the workgroups with blockIdx % (K / 32) == s (the aggregation kernel's slice <-> XCD binding,
 * efficient-gnns_amd/csrc/spmm_blk.hip).  nnz * K * 4 bytes of lines per call; `sink` [1] is never written for finite data.
 * K % 32 == 0, X 16-byte aligned, int32 column ids; loads_in_flight = independent 16-byte gathers per lane (4, 8 or 16).
 * No counterpart in the reference. */
int egnn_probe_gather_lines_f32(const float* X, int64_t ldx, int64_t n_src, int64_t K, const int32_t* col, int64_t nnz,
                                int blocks_per_slice, int loads_in_flight, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EGNN_HIP_H */
