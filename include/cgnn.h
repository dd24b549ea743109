/* cgnn.h -- C ABI of libcgnn_hip.so: the MI355X (gfx950) kernels behind the
 * batched message-passing path of connectome-gnn-suite.
 *
 * The reference has no FFI seam on this path (it is pure PyTorch); each entry
 * point below names the reference lines whose arithmetic it replaces.  The reference-side
 * binding (ctypes) is shown in INTEGRATION.md and implemented in connectome_gnn_amd/_lib.py.
 *
 * Conventions
 *   - extern "C"; every function returns int: CGNN_OK or a negative CGNN_E* code.  Nothing
 *     throws, aborts, allocates, frees or synchronises.
 *   - Pointers are DEVICE pointers unless the name ends in _host.  Sizes are element counts.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work is
 *     enqueued on it; outputs are valid once the stream reaches that point.
 *   - Re-entrant and thread-safe for distinct buffers/streams; no global mutable state.
 *   - Floating point is fp32 storage / fp32 accumulate unless a name says otherwise
 *     (BatchNorm statistics are accumulated in fp64).  Indices inside the library are int32;
 *     the int64 COO of the reference (graph.py:104-105) is read only by cgnn_csr_build.
 *
 * Layout vocabulary (graph.py:101-140): a ConnectomeBatch packs B graphs block-diagonally:
 *   Nn nodes, Ee directed edges, node n of graph g has global id ptr[g] + n;
 *   edge_index is [2, Ee] int64 row-major (row 0 = src, row 1 = dst).
 */
#ifndef CGNN_H
#define CGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGNN_OK            0
#define CGNN_EINVAL       (-1)  /* bad argument (null pointer, negative size, unsupported width) */
#define CGNN_ELAUNCH      (-2)  /* hipGetLastError() != hipSuccess after a launch */
#define CGNN_EUNSUPPORTED (-3)  /* shape outside what this build of the kernels covers */

#define CGNN_ABI_VERSION 2   /* 2: every scratch / partial-sum buffer the library WRITES is passed with its
                              * byte count (`<name>_bytes` right after the pointer); a buffer shorter than what
                              * the path about to run writes makes the call return CGNN_EINVAL before any launch
                              * (a NULL optional buffer ignores its count) */

/* Library/ABI version and the gfx target the kernels were compiled for ("gfx950"). */
int cgnn_abi_version(void);
const char* cgnn_build_target(void);

/* ---------------------------------------------------------------------------------------
 * Structure: destination- and source-sorted CSR of the batch COO.
 * Replaces the implicit index handling of scatter_add_/index in models.py:40-54,103-113,
 * 146-149 (forward = rows by dst; autograd's backward of x[src] = rows by src).
 * Slots inside a row keep COO order (stable), so per-row sums run in the reference's order
 * and duplicate edges survive.
 * ------------------------------------------------------------------------------------- */

/* Bytes of scratch cgnn_csr_build needs for (num_nodes, num_edges). */
int64_t cgnn_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges);

/* edge_index : int64 [2, Ee]         (graph.py:152 offsets already applied)
 * node_graph : int64 [Nn] or NULL    (ConnectomeBatch.batch; enables the cross-graph check)
 * rowptr_dst : int32 [Nn+1] out      in-edge row pointer       rowptr_src: same for out-edges
 * eid_dst    : int32 [Ee]   out      COO edge id of each slot  eid_src
 * col_dst    : int32 [Ee]   out      src node of each slot     col_src  : dst node of each slot
 * flags      : int32 [4]    out      [0] edges with an endpoint outside [0,Nn) (dropped),
 *                                    [1] edges whose endpoints lie in different graphs,
 *                                    [2] max in-degree, [3] max out-degree
 * workspace  : cgnn_csr_workspace_bytes() bytes, 16-byte aligned */
int cgnn_csr_build(const int64_t* edge_index, const int64_t* node_graph,
                   int64_t num_nodes, int64_t num_edges,
                   int32_t* rowptr_dst, int32_t* eid_dst, int32_t* col_dst,
                   int32_t* rowptr_src, int32_t* eid_src, int32_t* col_src,
                   int32_t* flags, void* workspace, int64_t workspace_bytes, void* stream);

/* Same outputs as cgnn_csr_build for batches whose COO is grouped by graph: the edges of graph g
 * are exactly the run [eptr[g], eptr[g+1]) and both endpoints lie in [gptr[g], gptr[g+1]) -- what
 * collate_graphs (graph.py:143-167) produces.  One workgroup builds a whole graph in LDS (single
 * pass over the COO, no global atomics): ~10x faster than the generic build at 4096 x 360-ROI.
 * gptr/eptr int32 [B+1]; max_nodes <= 1024, max_edges < 65535 and LDS = 16*(max_nodes+1) +
 * 8*max_edges <= 128 KB, else CGNN_EUNSUPPORTED.  flags as cgnn_csr_build; if flags[0] or flags[1] is non-zero the input
 * was not grouped/block-diagonal and the outputs are incomplete: rebuild with cgnn_csr_build. */
int cgnn_csr_build_grouped(const int64_t* edge_index, const int32_t* gptr, const int32_t* eptr,
                           int32_t num_graphs, int64_t num_nodes, int64_t num_edges,
                           int32_t max_nodes, int32_t max_edges,
                           int32_t* rowptr_dst, int32_t* eid_dst, int32_t* col_dst,
                           int32_t* rowptr_src, int32_t* eid_src, int32_t* col_src,
                           int32_t* flags, void* stream);

/* GCN symmetric normalisation, models.py:94-108: self-loop weight 1 appended last,
 * deg[i] = sum_{e: src=i} w_e + 1 (SOURCE side), dis = (deg + 1e-8)^-1/2,
 * c_e = dis[src]*w_e*dis[dst].  Layer independent -> once per batch.
 * dis [Nn], selfc [Nn] = dis^2, coef_dst [Ee] (dst-CSR slot order), coef_src [Ee]. */
int cgnn_gcn_norm(const int64_t* edge_index, const float* edge_weight,
                  int64_t num_nodes, int64_t num_edges,
                  const int32_t* rowptr_dst, const int32_t* eid_dst,
                  const int32_t* rowptr_src, const int32_t* eid_src,
                  float* dis, float* selfc, float* coef_dst, float* coef_src, void* stream);

/* GraphSAGE weighted-mean normalisation, models.py:146-149:
 * den[d] = sum_{e: dst=d} w_e + 1e-8; w_dst [Ee] = w permuted to dst-CSR slots;
 * coef_src_bwd [Ee] = w_e / den[dst_e] in src-CSR slots (backward of the mean; may be NULL when
 * the caller's backward divides by den itself, as cgnn_aggregate_tiled_f32 does). */
int cgnn_sage_norm(const int64_t* edge_index, const float* edge_weight,
                   int64_t num_nodes, int64_t num_edges,
                   const int32_t* rowptr_dst, const int32_t* eid_dst,
                   const int32_t* rowptr_src, const int32_t* eid_src,
                   float* den, float* w_dst, float* coef_src_bwd, void* stream);

/* ---------------------------------------------------------------------------------------
 * Edge-weighted segment reduction (the "scatter"), models.py:50-54,112-114,146-149.
 *   Y[r, :] = ( sum_{s in row r} coef[s] * X[col[s], :] ) / rowdiv[r]
 *             + selfc[r] * X[r, :] + bias[:]
 * rowdiv, selfc, bias may be NULL (treated as 1, 0, 0).  No atomics: one wave owns a row,
 * slots are summed in order.  The transposed pass (autograd of x[src]) is the same call on
 * the src-sorted CSR.  ldx/ldy are row strides in elements (>= F).
 * ------------------------------------------------------------------------------------- */
int cgnn_aggregate_f32(const int32_t* rowptr, const int32_t* col, const float* coef,
                       const float* selfc, const float* rowdiv, const float* bias,
                       const float* X, int64_t ldx, float* Y, int64_t ldy,
                       int64_t num_rows, int32_t F, void* stream);
/* Y[r, :] += the same sum (F = 64, 128 or 256; else CGNN_EUNSUPPORTED): the edges outside the dense
 * fragments that cgnn_band_aggregate_f32 has just written to Y (below). */
int cgnn_aggregate_acc_f32(const int32_t* rowptr, const int32_t* col, const float* coef,
                           const float* selfc, const float* rowdiv, const float* bias,
                           const float* X, int64_t ldx, float* Y, int64_t ldy,
                           int64_t num_rows, int32_t F, void* stream);

/* ---------------------------------------------------------------------------------------
 * Feature projection on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32),
 * models.py:111 (GCN, no bias) and :151-152 (SAGE: Linear([X || agg]) + bias, ReLU).
 *   fwd        : Y[M,N]  = act( X1[M,K1] W[:, 0:K1]^T + X2[M,K2] W[:, K1:K1+K2]^T + bias )
 *   bwd_input  : dX[M,K] = dY[M,N] W[:, k0:k0+K]        (ldw = K1+K2)
 *   bwd_weight : dW[N, k0:k0+K] = dY^T X                 (partials in `slab`, then reduced
 *                                                          in fp64, deterministic)
 * W is the nn.Linear weight [N, K1+K2] row-major.  X2/K2 = NULL/0 for one panel.
 * relu != 0 applies max(.,0) in the epilogue.
 * ------------------------------------------------------------------------------------- */
int cgnn_linear_fwd_f32(const float* X1, int64_t ldx1, int32_t K1,
                        const float* X2, int64_t ldx2, int32_t K2,
                        const float* W, const float* bias, int32_t relu,
                        float* Y, int64_t ldy, int64_t M, int32_t N, void* stream);

/* cgnn_linear_fwd_f32 that also leaves the BatchNorm statistics of its output behind:
 * stat_slab fp64 [cgnn_fused_grid()][2N] = per-workgroup (sum Y | sum Y^2), the layout
 * cgnn_bn_act_finalize reads -- saves the separate pass of cgnn_bn_act_fwd_stats over Y.
 * Only the tall weight-stationary shapes (M >= 4096, N in {64,128}, K % 32 == 0, K*N <= 32768,
 * 16-byte aligned rows) are covered; others return CGNN_EUNSUPPORTED and launch nothing. */
int cgnn_linear_fwd_stats_f32(const float* X1, int64_t ldx1, int32_t K1,
                              const float* X2, int64_t ldx2, int32_t K2,
                              const float* W, const float* bias, int32_t relu,
                              float* Y, int64_t ldy, int64_t M, int32_t N,
                              double* stat_slab, int64_t stat_slab_bytes, void* stream);

int cgnn_linear_bwd_input_f32(const float* dY, int64_t lddy, const float* W, int32_t ldw,
                              int32_t k0, float* dX, int64_t lddx,
                              int64_t M, int32_t N, int32_t K, void* stream);

/* Bytes of `slab` scratch for cgnn_linear_bwd_weight_f32(M, N, K) / for cgnn_linear_bwd_weight2_f32 with
 * panels K1, K2 (the larger of the joint one-pass form and the per-panel forms it may fall back to). */
int64_t cgnn_linear_bwd_weight_workspace_bytes(int64_t M, int32_t N, int32_t K);
int64_t cgnn_linear_bwd_weight2_workspace_bytes(int64_t M, int32_t N, int32_t K1, int32_t K2);

int cgnn_linear_bwd_weight_f32(const float* dY, int64_t lddy, const float* X, int64_t ldx,
                               float* dW, int32_t ldw, int32_t k0,
                               int64_t M, int32_t N, int32_t K, void* slab, int64_t slab_bytes, void* stream);

/* Both K-panels of dW = dY^T [X1 | X2] in one pass over dY (SAGELayer's Linear(2*in, out)).
 * slab: cgnn_linear_bwd_weight2_workspace_bytes(M, N, K1, K2) bytes. */
int cgnn_linear_bwd_weight2_f32(const float* dY, int64_t lddy, const float* X1, int64_t ldx1,
                                int32_t K1, const float* X2, int64_t ldx2, int32_t K2, float* dW,
                                int32_t ldw, int64_t M, int32_t N, void* slab, int64_t slab_bytes, void* stream);

/* fp16-STORAGE forms of the projection for large dense parcellations (BASELINE config 5; the
 * reference, models.py:111 and its autograd backward, has no fp16 path -- results are the fp32
 * oracle's to fp16 resolution).  Activations X / Y / dY / dX are IEEE half, the weight and its
 * gradient stay fp32 (converted to half once per workgroup while it is laid out as MFMA operands
 * in LDS), accumulation is fp32 on v_mfma_f32_32x32x16_f16.
 *   fwd       : Y[M,N] = X[M,K] W^T + bias   W fp32 [N, Kw] row-major with row stride ldw; columns
 *               k >= Kw of X meet zeros (layer 0's 5 input features ride in a 64-column panel)
 *   bwd_input : dX[M,K] = dY[M,N] W           W fp32 [N, K], row stride ldw
 *   bwd_weight: dW[n*ldw + k] = sum_m dY[m,n] X[m,k] for k < Kw   (fp32 partials per run of rows in
 *               `slab`, cgnn_linear_bwd_weight_f16_workspace_bytes(M,N,K) bytes, folded in fixed
 *               order with fp64 accumulation)
 * Shapes: N, K in {64,128,256} for fwd / bwd_input (K any multiple of 32 <= 256 on the reduction
 * side), N and K in {64,128,256} for bwd_weight; rows 16-byte aligned (ld % 8 == 0).
 * Anything else returns CGNN_EUNSUPPORTED and launches nothing. */
int cgnn_linear_fwd_f16(const void* X, int64_t ldx, int32_t K, const float* W, int32_t ldw, int32_t Kw,
                        const float* bias, void* Y, int64_t ldy, int64_t M, int32_t N, void* stream);
int cgnn_linear_bwd_input_f16(const void* dY, int64_t lddy, const float* W, int32_t ldw, void* dX,
                              int64_t lddx, int64_t M, int32_t N, int32_t K, void* stream);
/* cgnn_linear_fwd_f16 with the BatchNorm statistics of its output left behind by the epilogue
 * (per-workgroup fp64 partials, stat_slab [cgnn_fused_grid()][2 * N]: sum | sum of squares of the
 * half-rounded output columns, N = 128 or 256; combined by cgnn_bn_act_finalize with rows =
 * cgnn_fused_grid()): the projection in front of a BatchNorm (models.py:111,208) needs no statistics pass. */
int cgnn_linear_fwd_stats_f16(const void* X, int64_t ldx, int32_t K, const float* W, int32_t ldw, int32_t Kw,
                              const float* bias, void* Y, int64_t ldy, int64_t M, int32_t N, double* stat_slab, int64_t stat_slab_bytes,
                              void* stream);
int64_t cgnn_linear_bwd_weight_f16_workspace_bytes(int64_t M, int32_t N, int32_t K);
int cgnn_linear_bwd_weight_f16(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW,
                               int32_t ldw, int32_t Kw, int64_t M, int32_t N, int32_t K, void* slab, int64_t slab_bytes,
                               void* stream);
/* Y[M, Fp] (half) = [X[M, F] (fp32) | zeros]: the input features of a batch as a half panel. */
int cgnn_pad_cast_f16(const float* X, int64_t ldx, int32_t F, void* Y, int32_t Fp, int64_t M, void* stream);

/* Column sums (bias gradients, models.py:81,114): out[j] = sum_r A[r, j]; fp64 combine.
 * slab: cgnn_colsum_workspace_bytes(M, N) bytes. */
int64_t cgnn_colsum_workspace_bytes(int64_t M, int32_t N);
int cgnn_colsum_f32(const float* A, int64_t lda, float* out, int64_t M, int32_t N,
                    void* slab, int64_t slab_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Per-graph mean-pool readout, models.py:40-47,57-59: P[g,:] = sum_{n in g} X[n,:] / (n_g+1e-8)
 * using the contiguous node ranges gptr (int32 [B+1], = ConnectomeBatch.ptr).
 * ------------------------------------------------------------------------------------- */
int cgnn_pool_mean_fwd_f32(const float* X, int64_t ldx, const int32_t* gptr, float* P,
                           int32_t num_graphs, int32_t F, void* stream);
int cgnn_pool_mean_bwd_f32(const float* dP, const int32_t* gptr, float* dX, int64_t lddx,
                           int32_t num_graphs, int32_t F, void* stream);


/* =======================================================================================
 * FUSED PER-TILE GCN PATH (hidden = 64, fp32): the MI355X-first form of models.py:84-114 +
 * 203-216 and of its autograd backward.
 *
 * A *tile* is a run of consecutive whole graphs with at most CGNN_FUSED_MAX_ROWS nodes; its
 * [rows x 64] fp32 feature tile (<= 96 KB) lives in the CU's 160 KB LDS for the whole layer.
 * One persistent workgroup (8 waves) per CU walks the tiles.  Per layer, HBM sees exactly:
 * read the previous layer's pre-BatchNorm output, read the CSR once, write this layer's
 * pre-BatchNorm output.  BatchNorm-apply + ReLU + dropout are applied while the tile is being
 * staged, the projection runs on the matrix cores out of LDS, BatchNorm statistics are
 * accumulated (fp64) in the epilogue.  The batch-wide BatchNorm reduction is the only global
 * barrier, hence one launch per layer.
 *
 *   forward  l=0 : T = X0 W0^T (MFMA) -> LDS ; Y0 = A_hat T + b                (fwd_first)
 *   forward  l>0 : X = drop(relu(a*Yprev+b)) -> LDS ; Y = (A_hat X) W^T + b    (fwd)
 *   backward l   : dY = BN'(dZ) -> LDS ; dT = A_hat^T dY ; dW += dT^T X ;
 *                  dZprev = (dT W) * drop' * relu'                              (bwd, bwd_first)
 *
 * `bn` blocks are float[4*64]: a = gamma*invstd | b = beta - mean*a | mean | invstd.
 * `bwc` blocks are float[2*64]: c1 = sum(dZ)/N | c2 = sum(dZ*xhat)/N.
 * Dropout keep-bits are produced by a counter-based hash of (seed, element) when a layer's
 * output is first consumed and stored as one byte per (node, 4-column chunk) in `mask`
 * ([Nn*16] bytes); mask == NULL or p == 0 means no dropout.
 * Per-workgroup partial sums go to `*_slab` arrays with cgnn_fused_grid() rows and are
 * combined by cgnn_slab_reduce_* / cgnn_bn_reduce in a fixed order (deterministic, no atomics).
 * ===================================================================================== */
#define CGNN_FUSED_HIDDEN   64
#define CGNN_FUSED_MAX_ROWS 384
#define CGNN_FUSED_MAX_F0   16

/* Blocked-ELL metadata of one edge ordering (rows = destinations for forward, rows = sources
 * for the transposed passes).  Rows are grouped in 16-row blocks aligned to the start of each
 * tile; block b stores width_b = max degree in the block + 1 steps of 16 entries each
 * (entry (s, i) at blk_off[b] + 16*s + i), every entry 8 bytes:
 *     uint32 lds_offset = 256 * (neighbour's row inside the tile)   float weight
 * A row's real edges come first in COO order, then its self-loop (weight 1), then zero-weight
 * padding, so the kernels' inner loop is branch-free and sums in the reference's order.
 * Built once per batch (the weights are the raw edge weights; the symmetric normalisation is
 * applied as  dis[d] * sum_e w_e * (dis[s_e] * x[s_e])  with `dis` recomputed every step). */

/* Pass 1: widths + exclusive scan.  tile_ptr/tile_blk int32 [T+1] (tile_blk = prefix of
 * ceil(rows/16), NB = num_blocks = tile_blk[T]), rowptr int32 [Nn+1] of the ordering.
 * blk_off int32 [NB+1] out (in entries); scratch: int32 [NB/2048 + 8]. */
int cgnn_bell_plan(const int32_t* tile_ptr, const int32_t* tile_blk, int32_t num_tiles,
                   int32_t num_blocks, const int32_t* rowptr, int32_t* blk_off, int32_t* scratch, int64_t scratch_bytes,
                   void* stream);
/* Pass 2: fill entries (8 bytes each, blk_off[NB] of them).  self_weight: weight of the appended
 * self-loop entry -- 1 for GCN (models.py:97-100), 0 for GraphSAGE (no self-loop, models.py:146). */
int cgnn_bell_fill(const int32_t* tile_ptr, const int32_t* tile_blk, int32_t num_tiles,
                   const int32_t* rowptr, const int32_t* col, const int32_t* eid,
                   const float* edge_weight, float self_weight, const int32_t* blk_off,
                   void* entries, void* stream);

/* out[i] = src[idx[i]] (edge weights permuted to CSR slot order, once per batch). */
int cgnn_gather_f32(const float* src, const int32_t* idx, int64_t n, float* out, void* stream);

/* Several row gathers by one id list in ONE launch: dst[j][i] = src[j][ids[i]] for rows of
 * row_bytes[j] bytes (a multiple of 4; pointers 4-byte aligned).  The on-device form of the
 * reference's collate for device-resident regular datasets (graph.py:143-167 stacks node features and
 * labels per graph) extended to the per-subject structure arrays: node features, labels, blocked-ELL
 * block offsets and `dis` of a batch leave in one kernel.
 * ids_offset (nullable, device): the id list starts at ids + *ids_offset -- a captured step reads the
 * batch's position in the epoch's permutation from a device cursor, so replays need no host copy. */
#define CGNN_GATHER_MAX_JOBS 8
typedef struct cgnn_gather_jobs {
  int32_t n;
  const void* src[CGNN_GATHER_MAX_JOBS];
  void* dst[CGNN_GATHER_MAX_JOBS];
  int64_t row_bytes[CGNN_GATHER_MAX_JOBS];
} cgnn_gather_jobs;
int cgnn_gather_rows(const cgnn_gather_jobs* jobs, const int64_t* ids, int32_t num_ids,
                     const int64_t* ids_offset, void* stream);
/* End-of-step bookkeeping of an epoch replayed from one captured step (reference train.py:52-54 keeps
 * the running loss on the host): *tally += *loss * weight (tally nullable), *cursor += step. */
int cgnn_epoch_advance(int64_t* cursor, int64_t step, const float* loss, float weight, float* tally,
                       void* stream);

/* GCN degree normalisation, models.py:97-105, every step: dis[i] = (sum of row i of w_src
 * (COO order) + 1 + 1e-8)^-1/2 with w_src the edge weights in src-CSR slot order. */
int cgnn_gcn_dis(const float* w_src, const int32_t* rowptr_src, int64_t num_nodes, float* dis,
                 void* stream);

/* HOST-side parameter block (plain device pointers) describing the tiling of one batch. */
typedef struct cgnn_tiles {
  int64_t num_nodes;
  int32_t num_tiles;
  int32_t max_tile_rows;        /* <= CGNN_FUSED_MAX_ROWS */
  const int32_t* tile_ptr;      /* [num_tiles+1] node offsets; tiles are unions of whole graphs */
  const int32_t* tile_blk;      /* [num_tiles+1] first 16-row block of each tile */
  const int32_t* blk_off_dst; const void* ent_dst;   /* blocked-ELL, rows = destinations */
  const int32_t* blk_off_src; const void* ent_src;   /* blocked-ELL, rows = sources */
  const float* dis;             /* [num_nodes] */
} cgnn_tiles;

/* Layer 0's output in factored form.  For F0 <= 8 input features the fused path never writes
 * Y0 [Nn,64] to HBM: consumers rebuild a row from the narrow aggregate P0 = A_hat X0 (32 bytes per
 * node) as  y[c] = b0[c] + sum_{k<F0} P0[row][k] * W0[c][k]  (k ascending, fused multiply-adds --
 * the same expression in every kernel, so all of them see the same bits).  Saves one 256-byte
 * write and three 256-byte reads per node and step for 5 FMAs per rebuilt element. */
typedef struct cgnn_l0src {
  const float* P0;              /* [num_nodes, 8], columns >= F0 zero */
  const float* W0;              /* [64, F0] */
  const float* b0;              /* [64] */
  int32_t F0;                   /* 1..8 */
} cgnn_l0src;

/* Number of persistent workgroups every fused kernel launches (= rows of every slab). */
int cgnn_fused_grid(void);
/* TEST HOOK: make every persistent kernel launch `workgroups` workgroups instead of one per CU
 * (0 restores the device's CU count), so that small parity batches put several tiles / row blocks
 * on one workgroup -- the regime the benchmarks run in.  Process-wide; slabs sized with the old
 * value must not be reused across a change.  Not for production callers. */
int cgnn_set_fused_grid(int32_t workgroups);

/* Tiled edge-weighted aggregation for wide features (F % 64 == 0), the LDS-staged form of
 * cgnn_aggregate_f32 (models.py:112-114, :146-149 and their autograd transposes):
 *     Y[r, :] = post(r) * sum_{e in row r} w_e * pre(c_e) * X[c_e, :]  (+ bias) (+ Yadd[r, :])
 * over the blocked-ELL of `t` (rows = destinations, or sources when CGNN_AGG_TRANSPOSED), one
 * persistent workgroup per (tile, 64-column slice): the slice of the tile is staged in LDS once and
 * every neighbour row is read from there.  pre/post: float [Nn] or NULL; with CGNN_AGG_PRE_DIV /
 * CGNN_AGG_POST_DIV the row is divided by the vector instead of multiplied (SAGE's
 * sum / (wsum + 1e-8)).  Yadd (nullable, may alias Y) is added row-wise: the SAGE backward's
 * dX = dPre W1 + A^T(dPre W2) in one pass.  Whether the ELL carries a self-loop is decided when
 * it is filled (cgnn_bell_fill self_weight).  t->dis is not read. */
#define CGNN_AGG_TRANSPOSED 1
#define CGNN_AGG_PRE_DIV 2
#define CGNN_AGG_POST_DIV 4
int cgnn_aggregate_tiled_f32(const cgnn_tiles* t, int32_t flags, const float* X, int64_t ldx,
                             int32_t F, const float* pre, const float* post, const float* bias,
                             const float* Yadd, int64_t ldadd, float* Y, int64_t ldy, void* stream);
/* The same with BatchNorm(+ReLU)+dropout of the input applied while staging: X = drop(act(a Z + b)),
 * coef = [a | b | ..] of cgnn_bn_act_finalize over the F columns; X is also written to Xout and its
 * keep bytes to mask_out ([num_nodes][F/4], may be NULL) -- cgnn_bn_act_fwd_apply +
 * cgnn_aggregate_tiled_f32 in one launch, bit for bit. */
int cgnn_aggregate_tiled_bn_f32(const cgnn_tiles* t, int32_t flags, const float* Z, int64_t ldz,
                                int32_t F, const float* pre, const float* post, const float* bias,
                                float* Y, int64_t ldy, const float* coef, int32_t relu,
                                float p_drop, uint64_t seed, const uint32_t* seed_dev,
                                uint8_t* mask_out, float* Xout, int64_t ldxo, void* stream);

/* fp16-storage / fp32-accumulate form of cgnn_aggregate_tiled_f32 for large dense parcellations
 * (BASELINE config 5: 1000-ROI graphs, 10 % density, hidden 256): X, Y are IEEE half [Nn, F],
 * tiles of up to CGNN_H16_MAX_ROWS rows (a whole 1000-ROI graph x 64 columns = 128 KB of LDS),
 * entries streamed 16 steps at a time (any degree).  pre/post/bias stay fp32.  The reference has
 * no fp16 path; results are the fp32 oracle's to fp16 resolution (tests use 2e-3 of the scale). */
#define CGNN_H16_MAX_ROWS 1024
int cgnn_aggregate_tiled_f16(const cgnn_tiles* t, int32_t flags, const void* X, int64_t ldx,
                             int32_t F, const float* pre, const float* post, const float* bias,
                             void* Y, int64_t ldy, void* stream);

/* Dense form of the same aggregation for large dense parcellations, on the fp16 matrix cores
 * (v_mfma_f32_32x32x16_f16, fp32 accumulate): Y_g = M_g X_g per graph.
 * cgnn_dense_adj_f16 builds M [B][P][P] half once per batch from a CSR ordering: M_g[r][c] = sum of
 * coef over the slots (row r, column c) (+ selfc[r] on the diagonal, nullable); rows/columns past a
 * graph's size are zero.  The element order inside M is the kernel's own (MFMA-fragment-major):
 * M is only ever passed back to cgnn_dense_aggregate_f16.  Pass the dst-sorted CSR + coef_dst for the forward operator, the
 * src-sorted CSR + coef_src for its transpose.  P: common pitch, multiple of 64, <= 1024, >= the
 * largest graph.  X, Y half [Nn, F], F % 64 == 0; bias fp32 or NULL. */
int cgnn_dense_adj_f16(const int32_t* rowptr, const int32_t* col, const float* coef,
                       const float* selfc, const int32_t* gptr, int32_t num_graphs, int32_t P,
                       void* M, void* stream);
int cgnn_dense_aggregate_f16(const void* M, int32_t P, const int32_t* gptr, int32_t num_graphs,
                             const void* X, int64_t ldx, int32_t F, const float* bias, void* Y,
                             int64_t ldy, double* stat_slab, int64_t stat_slab_bytes, void* stream);

/* stat_slab (both aggregates; NULL = none): [cgnn_fused_grid()][2 * F] fp64 per-workgroup column sums
 * and sums of squares of the half-rounded result (BatchNorm statistics in the epilogue; finalise with
 * cgnn_bn_act_finalize(slab, cgnn_fused_grid(), F, ...)). */
/* The same operator stored per MFMA A fragment (32 rows x 16 sources) in the form that fits it:
 * fragments with more than 64 non-zeros in a DENSE list (1 KB each, operand-major, `dfrag`), those
 * with 1..64 in a SPARSE list (one chunk of 64 (slot | half value << 16) words each, padding slot
 * 0xFFFF, four chunks interleaved per lane, `sent`), empty ones nowhere; `dstep` / `sstep` hold the
 * k-step of every item, `doff` / `soff` [num_graphs * P/32 + 1] the item ranges of the (graph, row
 * block)s (sparse ranges are multiples of four chunks).  Building (P as above):
 *   1. cgnn_dense_pack_count: counts[num_graphs * P/32 * P/16] <- non-zeros per fragment;
 *   2. the caller classifies, forms doff / soff and fpos[fragment] (list position; | 0x80000000 for
 *      the dense list; 0xFFFFFFFF = empty), allocates dfrag / sent (sent pre-filled with 0xFFFF
 *      words, sstep with 0) and calls cgnn_dense_pack_fill.
 * cgnn_dense_aggregate_c16 computes what cgnn_dense_aggregate_f16 computes with the same MFMAs,
 * accumulated dense-list-first (equal up to the order of the fp32 accumulation).  On small-world
 * graphs at 10 % density the operator shrinks ~3x and the dense form is HBM-bound on it. */
int cgnn_dense_pack_count(const int32_t* rowptr, const int32_t* col, const float* coef,
                          const float* selfc, const int32_t* gptr, int32_t num_graphs, int32_t P,
                          uint32_t* counts, void* stream);
int cgnn_dense_pack_fill(const int32_t* rowptr, const int32_t* col, const float* coef,
                         const float* selfc, const int32_t* gptr, int32_t num_graphs, int32_t P,
                         const uint32_t* fpos, void* dfrag, int32_t* dstep, uint32_t* sent,
                         int32_t* sstep, void* stream);
int cgnn_dense_aggregate_c16(const void* dfrag, const int32_t* dstep, const uint32_t* doff,
                             const uint32_t* sent, const int32_t* sstep, const uint32_t* soff,
                             int32_t P, const int32_t* gptr, int32_t num_graphs, const void* X,
                             int64_t ldx, int32_t F, const float* bias, void* Y, int64_t ldy,
                             double* stat_slab, int64_t stat_slab_bytes, void* stream);
/* The backward product dT = A_hat^T dY of a GCN layer with dY never materialised: the slice handed to
 * the matrix cores is formed while it is staged,
 *     dY = a * (dX' * f - c1 - xhat * c2),   f = relu' * keep / (1-p),   xhat = (Yl - mean) * invstd
 * (the arithmetic of cgnn_bn_act_bwd_apply_f16, models.py:208-210 through autograd), from the gradient
 * of the layer's activation -- dX' [M, F] half, or for the last layer the readout gradient dP [B, F]
 * fp32 (row gradient dP[graph] / (n_g + 1e-8)); exactly one of the two -- the layer's pre-BatchNorm
 * output Yl, keep bytes, coefficient block `coef` and the backward coefficients `bwc`.  dY feeds
 * nothing else but the bias gradient: its column sums are left per graph in cs_slab [B][F] fp64
 * (combine with cgnn_slab_reduce_f64(cs_slab, B, F, db)). */
int cgnn_dense_aggregate_c16_bnbwd(const void* dfrag, const int32_t* dstep, const uint32_t* doff,
                                   const uint32_t* sent, const int32_t* sstep, const uint32_t* soff,
                                   int32_t P, const int32_t* gptr, int32_t num_graphs, const void* dX,
                                   int64_t lddx, const float* dP, const void* Yl, int64_t ldyl,
                                   const uint8_t* mask, const float* coef, const float* bwc, int32_t relu,
                                   float p_drop, int32_t F, void* dT, int64_t lddt, double* cs_slab, int64_t cs_slab_bytes,
                                   void* stream);

/* The DENSE FRAGMENTS of an aggregation operator applied in fp32 on the bf16 matrix pipe with exactly
 * split operands (band_aggregate.hip) -- for graphs of more than 384 nodes in the reference's own
 * arithmetic (models.py:112-114 / :146-149 and their autograd transposes at 1000 ROI), where the gather
 * kernel cgnn_aggregate_f32 is bound by reading ~100 neighbour rows per output row out of L2.
 *   cgnn_band_pack_f32: from one CSR ordering (as cgnn_dense_pack_count/fill: P = dense pitch, fpos
 *     [num_graphs * P/32 * P/16] = position of a fragment in the dense list or 0xFFFFFFFF) the listed
 *     fragments as MFMA A operands cut into three bf16 pieces: bfrag [items][3][64 lanes][8 bf16]
 *     (3 KB per fragment), bstep [items] = k-step.  Duplicate edges add up in fp32.  Static per batch.
 *   cgnn_band_aggregate_f32: Y[r,:] = (sum over the listed fragments of row block r/32) (/ rowdiv[r]) for
 *     every row (0 where the row block lists nothing), + Yadd[r,:] when Yadd is not NULL (GraphSAGE's
 *     dX = dX1 + A^T(dA / den), models.py:146-152 backward; Yadd may not alias Y); boff [num_graphs * P/32 + 1] = item ranges of the
 *     (graph, row block)s; F % 32 == 0, ldx / ldy even, X / Y 8-byte aligned, Y != X.  The caller then
 *     runs cgnn_aggregate_acc_f32 on the CSR of the edges OUTSIDE the listed fragments (it also carries
 *     the self-loop term, the row division and the bias) on top: together the full operator, each matrix
 *     product exact to 2^-24 relative, the sums in another order than the reference's. */
int cgnn_band_pack_f32(const int32_t* rowptr, const int32_t* col, const float* coef, const int32_t* gptr,
                       int32_t num_graphs, int32_t P, const uint32_t* fpos, void* bfrag, int32_t* bstep,
                       void* stream);
int cgnn_band_aggregate_f32(const void* bfrag, const int32_t* bstep, const int32_t* boff, int32_t P,
                            const int32_t* gptr, int32_t num_graphs, const float* X, int64_t ldx, int32_t F,
                            const float* rowdiv, const float* Yadd, int64_t ldadd, float* Y, int64_t ldy,
                            void* stream);


/* BatchNorm finalisation in the PRODUCER's tail (round 4; csrc/bn_tail.h): instead of writing its
 * per-workgroup partial sums to a slab for cgnn_bn_stats_finalize_rng / cgnn_bn_bwd_stats_finalize to fold
 * in a launch of their own, a tile kernel adds them to an accumulator (fixed-point 64-bit atomic adds:
 * order-independent, bit-identical reruns) and the workgroup that arrives last writes the layer's
 * coefficient block.  `acc`: CGNN_BN_ACC_BYTES bytes of device memory, ZERO before the first launch that
 * uses it; every launch leaves it zero again.  One accumulator may serve launches on ONE stream.
 *   mode 0 (statistics of a forward layer): the arithmetic of cgnn_bn_stats_finalize_rng -- count rows,
 *     gamma/beta, running statistics and num_batches_tracked updated in place, bn_out float[4*64], rng_state /
 *     rng_n as there (the step's dropout words, refreshed by the tail; NULL / 0 = none);
 *   mode 1 (sums of a backward layer): the arithmetic of cgnn_bn_bwd_stats_finalize -- dgamma, dbeta
 *     float[64], bwc float[2*64], zero_coef as there.
 * Passed as the last argument of cgnn_gcn_l0_fwd (mode 0; factored layer 0: the centred form's mean offset
 * is the kernel's own), cgnn_gcn_fused_fwd (mode 0) and cgnn_gcn_fused_bwd (mode 1, the sums of the layer
 * BELOW); NULL = the slab protocol.  With a tail the corresponding slab argument may be NULL. */
#define CGNN_BN_ACC_BYTES 16448
typedef struct cgnn_bn_tail {
  void* acc;
  double count;
  int32_t mode, zero_coef;
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;
  float momentum, eps;
  int64_t* num_batches_tracked;
  float* bn_out;
  uint32_t* rng_state; int32_t rng_n; int32_t reserved;
  float* dgamma; float* dbeta; float* bwc;
} cgnn_bn_tail;

/* Layer 0 forward.  X0 [Nn,F0] (F0 <= 16), W0 [64,F0], bias [64] -> Y [Nn,64];
 * stat_slab [grid][128] fp64 (sum(y) | sum(y^2)) or NULL (eval). */
int cgnn_gcn_fused_fwd_first(const cgnn_tiles* t, const float* X0, int32_t F0, const float* W0,
                             const float* bias, float* Y, double* stat_slab, int64_t stat_slab_bytes, void* stream);

/* Layer l>0 forward.  Yprev [Nn,64] + bn_prev -> X on the fly; W [64,64]; mask_out nullable.
 * Yprev == NULL: the previous layer is layer 0 in factored form, rows rebuilt from *l0. */
/* seed_dev (nullable): device word XOR-ed into the dropout key at kernel start.  Under HIP-graph
 * replay the by-value `seed` is frozen in the graph; a captured cgnn_rng_advance on the same
 * stream then makes every replay draw fresh masks. */
int cgnn_rng_advance(uint32_t* state, int32_t n, void* stream);
int cgnn_gcn_fused_fwd(const cgnn_tiles* t, const float* Yprev, const cgnn_l0src* l0,
                       const float* bn_prev, float p_drop, uint64_t seed, const uint32_t* seed_dev,
                       uint8_t* mask_out, const float* W, const float* bias, float* Y,
                       double* stat_slab, int64_t stat_slab_bytes, const cgnn_bn_tail* tail, void* stream);

/* slab [rows][width] fp64 -> sums [width] fp64 (fixed-order tree). */
int cgnn_bn_reduce(const double* slab, int32_t rows, int32_t width, double* sums, void* stream);

/* BatchNorm1d forward coefficients (models.py:191-193 semantics: biased batch variance,
 * eps, momentum; running_var gets the unbiased variance).  training != 0: from sums
 * (sum(y)|sum(y^2), [128]) over `count` rows, and running stats are updated in place;
 * training == 0: from running stats.  bn_out float[4*64]. */
/* count_dev (nullable): device double that overrides `count` -- the all-reduced row count of a
 * multi-rank batch stays on the device, no host read-back between reduce and finalise. */
int cgnn_bn_finalize(const double* sums, double count, const double* count_dev, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, int32_t training, float* bn_out, const float* mean_offset, void* stream);

/* Readout: P[g,:] = mean over nodes of drop(relu(a*Y+b)) (models.py:209-211,57-59).
 * F1, F2 (both or neither; [num_graphs,64]): per graph and column the sums over the graph's rows
 * of f = relu'(z)*keep/(1-p) and of f*xhat -- what cgnn_gcn_fused_pool_bwd_sums needs. */
int cgnn_gcn_fused_pool_fwd(const float* Y, const float* bn, float p_drop, uint64_t seed,
                            const uint32_t* seed_dev, uint8_t* mask_out, const int32_t* gptr,
                            int32_t num_graphs, float* P, float* F1, float* F2, void* stream);

/* BatchNorm-backward sums of the LAST layer without re-reading Y: the readout's gradient is
 * dP[g]/(n_g+1e-8) for every row of graph g, so  sum dZ = sum_g dP[g]/n_g * F1[g]  and
 * sum dZ*xhat = sum_g dP[g]/n_g * F2[g].  s_slab [cgnn_fused_grid()][128] fp64 partials. */
int cgnn_gcn_fused_pool_bwd_sums(const float* dP, const float* F1, const float* F2,
                                 const int32_t* gptr, int32_t num_graphs, double* s_slab, int64_t s_slab_bytes,
                                 void* stream);
/* (cgnn_gcn_fused_pool_bwd_sums + cgnn_bn_bwd_stats_finalize) in one launch: dgamma/dbeta [64] and
 * the c1|c2 block bwc [128] of the last layer, for per-rank BatchNorm statistics (no exchange
 * between the sums and the coefficients).  count = rows of the batch. */
int cgnn_gcn_fused_pool_bwd_finalize(const float* dP, const float* F1, const float* F2,
                                     const int32_t* gptr, int32_t num_graphs, double count,
                                     int32_t zero_coef, float* dgamma, float* dbeta, float* bwc,
                                     void* stream);

/* Backward of the readout through dropout/ReLU: dZ = dP[g]/(n_g+1e-8) * drop' * relu';
 * also the BatchNorm-backward sums of the last layer: s_slab [grid][128] = sum dZ | sum dZ*xhat.
 * dZ may be NULL (sums only) when the last layer's backward rebuilds dZ itself (dP != NULL there). */
int cgnn_gcn_fused_pool_bwd(const float* dP, const float* Y, const float* bn, float p_drop,
                            const uint8_t* mask, const int32_t* gptr, int32_t num_graphs,
                            float* dZ, double* s_slab, int64_t s_slab_bytes, void* stream);

/* BatchNorm backward coefficients: dgamma = sum dZ*xhat, dbeta = sum dZ, bwc = [c1|c2]. */
int cgnn_bn_bwd_finalize(const double* sums, double count, const double* count_dev,
                         int32_t zero_coef, float* dgamma, float* dbeta, float* bwc, void* stream);

/* Layer l>0 backward.  Inputs: dZ (grad wrt BN output of layer l), Y (its pre-BN output), bn,
 * bwc; previous layer's Yprev/bn_prev/mask_prev (to rebuild X_l and apply relu'/drop').
 * Outputs: dZprev [Nn,64]; s_slab_prev [grid][128]; dW_slab [grid][64*64]; db_slab [grid][64]. */
/* Last layer only: pass dP != NULL (then dZ is ignored and may be NULL) together with
 * node_graph int32 [Nn], gptr int32 [B+1] and mask_cur (this layer's keep bits): the incoming
 * gradient is rebuilt per row as dP[g]/(n_g+1e-8) * relu' * dropout'.  Otherwise dP = NULL.
 * Yprev == NULL: the previous layer is layer 0 in factored form, rows rebuilt from *l0. */
int cgnn_gcn_fused_bwd(const cgnn_tiles* t, const float* dZ, const float* Y, const float* bn,
                       const float* bwc, const float* Yprev, const cgnn_l0src* l0,
                       const float* bn_prev, float p_drop, const uint8_t* mask_prev, const float* W,
                       float* dZprev, double* s_slab_prev, int64_t s_slab_prev_bytes, float* dW_slab, int64_t dW_slab_bytes, double* db_slab, int64_t db_slab_bytes,
                       const float* dP, const int32_t* node_graph, const int32_t* gptr,
                       const uint8_t* mask_cur, const cgnn_bn_tail* tail, void* stream);

/* Layer 0 backward: dW0 = dT^T X0 only.  dW_slab [grid][64*16] (columns >= F0 are zero). */
int cgnn_gcn_fused_bwd_first(const cgnn_tiles* t, const float* dZ, const float* Y,
                             const float* bn, const float* bwc, const float* X0, int32_t F0,
                             float* dW_slab, int64_t dW_slab_bytes, double* db_slab, int64_t db_slab_bytes, float p_drop, const float* dP,
                             const int32_t* node_graph, const int32_t* gptr,
                             const uint8_t* mask_cur, void* stream);

/* Layer 0, narrow form (F0 <= 8): Y0 = (A_hat X0) W0^T + b with P0 = A_hat X0 kept for backward
 * ([Nn,8] fp32, columns >= F0 zero); dW0 = dY0^T P0, db0 = sum dY0 need no aggregation.
 * Slabs have cgnn_l0_grid(num_nodes) rows: stat_slab [..][128] fp64, dW_slab [..][64*8] f32,
 * db_slab [..][64] fp64.  (cgnn_gcn_fused_fwd_first/bwd_first remain for 8 < F0 <= 16 and for
 * one-layer models.) */
int cgnn_l0_grid(int64_t num_nodes);
/* Y (forward) may be NULL: only P0 and the statistics are produced and every consumer rebuilds
 * Y0's rows from a cgnn_l0src.  Backward: Y == NULL -> rebuilt from P0 with l0->W0/b0/F0.
 *
 * Centred form (Y == NULL, F0 <= 7; round 3).  With `center` (float[8], device): c = center[0..F0)
 * near the column means of X0 and rbar = center[7] near the mean of r = A_hat 1 (the normalised
 * operator's row sums) -- cgnn_gcn_l0_center computes both from the batch's first non-empty tile;
 * any finite values give the same result up to rounding --
 *     A_hat X0 = A_hat (X0 - 1 c^T) + r c^T
 *     P0' = [A_hat (X0 - 1 c^T) | r - rbar | 0..]   ([Nn,8]; column F0 = the aggregated ones column,
 *                                                     summed in fp64)
 *     W'  = [W0 | W0 c]                              (`w_eff`, float[64][F0 + 1], written by the launch)
 *     Y0  = P0 W0^T + b = P0' W'^T + mean_offset,    mean_offset = b + rbar W0 c  (float[64], written)
 * The layer is handed on WITHOUT its constant term: consumers take cgnn_l0src{P0', w_eff, zeros, F0 + 1},
 * the statistics in stat_slab are those of y_c = Y0 - mean_offset, and the BatchNorm finalisation
 * (cgnn_bn_stats_finalize_rng / cgnn_bn_finalize with the same mean_offset) describes y_c -- BatchNorm
 * is invariant under a per-channel shift; only the running mean sees the constant.  The SAME function
 * of the parameters, evaluated at the scale of the features' spread: with node features far from
 * zero (un-normalised strength / degree columns) the raw form puts rounding of the size
 * 2^-24 * mean into every aggregated row and every rebuilt y, and loses log2((mean/sigma)^2) bits of
 * the BatchNorm variance.  cgnn_gcn_l0_bwd (same `center`, l0->F0 = F0 + 1) returns
 * dW0[:, k] = dW'[:, k] + c[k] (dW'[:, F0] + rbar db0) in the first F0 of the 8 slab columns. */
int cgnn_gcn_l0_center(const cgnn_tiles* t, const float* X0, int32_t F0, float* center, void* stream);
int cgnn_gcn_l0_fwd(const cgnn_tiles* t, const float* X0, int32_t F0, const float* W0,
                    const float* bias, float* P0, float* Y, double* stat_slab, int64_t stat_slab_bytes, const float* center,
                    float* w_eff, float* mean_offset, const cgnn_bn_tail* tail, void* stream);
int cgnn_gcn_l0_bwd(const float* dZ, const float* Y, const cgnn_l0src* l0, const float* bn,
                    const float* bwc, const float* P0, int64_t num_nodes, float* dW_slab, int64_t dW_slab_bytes,
                    double* db_slab, int64_t db_slab_bytes, const float* center, void* stream);

/* fp16-storage forms of the BatchNorm(+ReLU)+dropout kernels (cgnn_bn_act_*): the [M,N] activation
 * arrays (Y, X, dX, dY) are IEEE half, the arithmetic is fp32, the statistics fp64, coefficient
 * blocks / masks / slabs / pooled rows exactly as in the fp32 forms.  Used by the fp16-storage
 * GCN encoder for large dense parcellations (BASELINE config 5); the reference has no fp16 path
 * (models.py is fp32-only), results are the fp32 oracle's to fp16 resolution. */
int cgnn_bn_act_fwd_stats_f16(const void* Y, int64_t M, int32_t N, double* slab, int64_t slab_bytes, void* stream);
int cgnn_bn_act_fwd_apply_f16(const void* Y, const float* coef, int32_t relu, float p_drop, uint64_t seed,
                              const uint32_t* seed_dev, uint8_t* mask_out, void* X, int64_t M,
                              int32_t N, void* stream);
int cgnn_bn_act_pool_fwd_f16(const void* Y, const float* coef, int32_t relu, float p_drop, uint64_t seed,
                             const uint32_t* seed_dev, uint8_t* mask_out, const int32_t* gptr,
                             int32_t num_graphs, float* P, int32_t N, float* Fsum, void* stream);
int cgnn_bn_act_bwd_stats_f16(const void* dX, const void* Y, const uint8_t* mask, const float* coef,
                              int32_t relu, float p_drop, int64_t M, int32_t N, double* slab, int64_t slab_bytes,
                              const float* dP, const int32_t* node_graph, const int32_t* gptr,
                              void* stream);
int cgnn_bn_act_bwd_apply_f16(const void* dX, const void* Y, const uint8_t* mask, const float* coef,
                              const float* bwc, int32_t relu, float p_drop, int32_t relu_in,
                              double* colsum_slab, int64_t colsum_slab_bytes, void* dY, int64_t M, int32_t N, const float* dP,
                              const int32_t* node_graph, const int32_t* gptr, void* stream);

/* ---- optimizer: torch.optim.Adam's update (L2 weight decay, no amsgrad) for up to
 * CGNN_ADAM_MAX_JOBS fp32 tensors in ONE launch (reference: torch.optim.Adam in demo.py:105-134).
 * `step` is the shared fp32 step counter on the device (t = *step + 1 is used); with advance != 0
 * the same launch stores *step + 1 once every workgroup has read it (`arrivals`: a zeroed uint32
 * the launch leaves zero again).  Graph-capturable: nothing comes from the host but lr/betas. */
#define CGNN_ADAM_MAX_JOBS 32
typedef struct cgnn_adam_jobs {
  int32_t n;
  int64_t numel[CGNN_ADAM_MAX_JOBS];
  float* param[CGNN_ADAM_MAX_JOBS];
  const float* grad[CGNN_ADAM_MAX_JOBS];
  float* exp_avg[CGNN_ADAM_MAX_JOBS];
  float* exp_avg_sq[CGNN_ADAM_MAX_JOBS];
} cgnn_adam_jobs;
int cgnn_adam_step(const cgnn_adam_jobs* jobs, float* step, uint32_t* arrivals, int32_t advance,
                   double lr, double beta1, double beta2, double eps, double weight_decay, void* stream);

/* Single-launch forms of (cgnn_bn_reduce + cgnn_bn_finalize [+ num_batches_tracked += 1]),
 * (cgnn_bn_reduce + cgnn_bn_bwd_finalize) and (cgnn_slab_reduce_f32 + cgnn_slab_reduce_f64):
 * used when no cross-rank exchange sits between the reduction and the finalisation.
 * zero_coef != 0 writes c1 = c2 = 0 (eval-mode BatchNorm backward is a fixed affine map). */
int cgnn_bn_stats_finalize(const double* slab, int32_t rows, double count, const float* gamma,
                           const float* beta, float* running_mean, float* running_var,
                           float momentum, float eps, int64_t* num_batches_tracked, float* bn_out,
                           void* stream);
/* the same, also refreshing the rng_n (<= 64) device dropout words of a graph-captured step
 * (cgnn_rng_advance's arithmetic) in the same launch; rng_state NULL / rng_n 0: no refresh.
 * mean_offset (float[64], nullable; also cgnn_bn_finalize): the slab holds the statistics of
 * y - mean_offset[c] (the centred factored layer 0 is handed on without its constant term, which
 * BatchNorm's output does not depend on): bn_out describes that shifted variable, the module's
 * running_mean is updated with (resp. in eval mode read as) the mean of y itself. */
int cgnn_bn_stats_finalize_rng(const double* slab, int32_t rows, double count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, int64_t* num_batches_tracked, float* bn_out,
                               uint32_t* rng_state, int32_t rng_n, const float* mean_offset, void* stream);
int cgnn_bn_bwd_stats_finalize(const double* slab, int32_t rows, double count, int32_t zero_coef,
                               float* dgamma, float* dbeta, float* bwc, void* stream);
int cgnn_dw_db_reduce(const float* dw_slab, const double* db_slab, int32_t rows, int32_t out_cols,
                      int32_t take_cols, float* dW, int32_t ld_dw, float* db, void* stream);
/* The same for up to CGNN_DW_MAX_JOBS layers in one launch (dW[i] dense [64][take_cols[i]]): the
 * reductions do not feed the backward chain, so all of a model's layers wait for its end. */
#define CGNN_DW_MAX_JOBS 8
typedef struct cgnn_dw_jobs {
  int32_t n;
  const float* dw_slab[CGNN_DW_MAX_JOBS];
  const double* db_slab[CGNN_DW_MAX_JOBS];
  int32_t rows[CGNN_DW_MAX_JOBS], out_cols[CGNN_DW_MAX_JOBS], take_cols[CGNN_DW_MAX_JOBS];
  float* dW[CGNN_DW_MAX_JOBS];
  float* db[CGNN_DW_MAX_JOBS];
} cgnn_dw_jobs;
int cgnn_dw_db_reduce_multi(const cgnn_dw_jobs* jobs, void* stream);

/* Fixed-order combination of per-workgroup partials (fp64 accumulate):
 * f32 slab [rows][width] -> out[r*ld_out + c] for width = out_rows*out_cols (take the first
 * `take_cols` of every `out_cols` columns); f64 slab [rows][width] -> f32 out [width]. */
int cgnn_slab_reduce_f32(const float* slab, int32_t rows, int32_t out_rows, int32_t out_cols,
                         int32_t take_cols, float* out, int32_t ld_out, void* stream);
int cgnn_slab_reduce_f64(const double* slab, int32_t rows, int32_t width, float* out,
                         void* stream);
/* f32 slab [rows][width] -> out[0..split) and out_tail[0..width-split): the same fold with its result in two
 * places -- the classifier's parameter gradients straight into a caller-owned gradient buffer (data-parallel
 * training: the flat all-reduce buffer) and the loss column of cgnn_head_loss_f32's slab next to it. */
int cgnn_slab_reduce_f32_split(const float* slab, int32_t rows, int32_t width, int32_t split, float* out,
                               float* out_tail, void* stream);
/* up to CGNN_REDUCE_MAX_JOBS of the f64 form in ONE launch (the bias gradients of every layer of a
 * backward pass: their per-block column sums are final long before the pass ends) */
#define CGNN_REDUCE_MAX_JOBS 8
typedef struct cgnn_reduce_jobs {
  int32_t n;
  const double* slab[CGNN_REDUCE_MAX_JOBS];
  int32_t rows[CGNN_REDUCE_MAX_JOBS];
  int32_t width[CGNN_REDUCE_MAX_JOBS];
  float* out[CGNN_REDUCE_MAX_JOBS];
} cgnn_reduce_jobs;
int cgnn_slab_reduce_f64_multi(const cgnn_reduce_jobs* jobs, void* stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm1d (+ReLU) + dropout for the layered path (any power-of-two width 4..1024),
 * models.py:208-210 (GCN: BN, ReLU, dropout) and :260-261 (SAGE: BN, dropout):
 *   forward : cgnn_bn_act_fwd_stats -> cgnn_bn_act_finalize -> cgnn_bn_act_fwd_apply
 *             X' = drop(act(a*Y + b)); coef float[4N] = a | b | mean | invstd
 *   backward: cgnn_bn_act_bwd_stats -> cgnn_bn_act_bwd_finalize -> cgnn_bn_act_bwd_apply
 *             dY = a*(dX'*drop'*act' - c1 - xhat*c2); bwc float[2N] = c1 | c2
 * slabs: fp64 [cgnn_bn_act_slab_rows(M)][2N]; mask: one byte (4 keep bits) per 4-column chunk,
 * [M*N/4] bytes, may be NULL when p_drop == 0.  Semantics of nn.BatchNorm1d as in
 * cgnn_bn_finalize (training: batch stats + running update; eval: running stats).
 * ------------------------------------------------------------------------------------- */
int cgnn_bn_act_width_ok(int32_t N);
int64_t cgnn_bn_act_slab_rows(int64_t M);
int cgnn_bn_act_fwd_stats(const float* Y, int64_t M, int32_t N, double* slab, int64_t slab_bytes, void* stream);
/* count_dev (nullable): the row count read from device memory instead of `count` -- under
 * full-batch BatchNorm across ranks the caller all-reduces [sum | sumsq | rows] and passes the
 * reduced block as a 1-row slab with count_dev = &block[2N]; nothing returns to the host. */
int cgnn_bn_act_finalize(const double* slab, int32_t rows, int32_t N, double count,
                         const double* count_dev, int32_t training,
                         const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps,
                         int64_t* num_batches_tracked, float* coef, void* stream);
int cgnn_bn_act_fwd_apply(const float* Y, const float* coef, int32_t relu, float p_drop,
                          uint64_t seed, const uint32_t* seed_dev, uint8_t* mask_out, float* X,
                          int64_t M, int32_t N, void* stream);
/* Readout fused with the last layer's BatchNorm(+act)+dropout (models.py:211 / :262):
 * P[g,:] = sum_{rows of g} drop(act(a*Y+b)) / (n_g + 1e-8); X' is never materialised.  gptr int32
 * [B+1].  Its backward is the dP form of the two kernels below: dP != NULL replaces dX by
 * dP[node_graph[r],:] / (n_g + 1e-8), rebuilt per row (dX may then be NULL). */
int cgnn_bn_act_pool_fwd(const float* Y, const float* coef, int32_t relu, float p_drop, uint64_t seed,
                         const uint32_t* seed_dev, uint8_t* mask_out, const int32_t* gptr,
                         int32_t num_graphs, float* P, int32_t N, float* Fsum, void* stream);
/* Fsum (nullable, float [2][B][N]): the pooled pass also leaves the per-graph factor sums
 * F1[g][c] = sum_rows f, F2[g][c] = sum_rows f * xhat (f = act' * keep / (1-p)); the readout's
 * gradient is constant per graph, so this layer's BatchNorm-backward sums follow from them without
 * a pass over Y:  cgnn_bn_act_pool_bwd_finalize == cgnn_bn_act_bwd_stats(dP form) + cgnn_bn_act_bwd_finalize. */
int cgnn_bn_act_pool_bwd_finalize(const float* dP, const float* Fsum, const int32_t* gptr, int32_t num_graphs,
                                  int32_t N, double count, int32_t zero_coef, float* dgamma, float* dbeta,
                                  float* bwc, void* stream);
int cgnn_bn_act_bwd_stats(const float* dX, const float* Y, const uint8_t* mask, const float* coef,
                          int32_t relu, float p_drop, int64_t M, int32_t N, double* slab, int64_t slab_bytes,
                          const float* dP, const int32_t* node_graph, const int32_t* gptr,
                          void* stream);
int cgnn_bn_act_bwd_finalize(const double* slab, int32_t rows, int32_t N, double count,
                             const double* count_dev, int32_t zero_coef, float* dgamma,
                             float* dbeta, float* bwc, void* stream);
/* relu_in != 0: Y is itself the output of a ReLU (SAGELayer, models.py:152): dY is additionally
 * masked by Y > 0, i.e. it is the gradient of the layer's pre-activation.  colsum_slab (nullable):
 * fp64 [cgnn_bn_act_apply_blocks(M, N)][N] per-block column sums of dY (the bias gradient),
 * combined with cgnn_slab_reduce_f64. */
int64_t cgnn_bn_act_apply_blocks(int64_t M, int32_t N);
int cgnn_bn_act_bwd_apply(const float* dX, const float* Y, const uint8_t* mask, const float* coef,
                          const float* bwc, int32_t relu, float p_drop, int32_t relu_in,
                          double* colsum_slab, int64_t colsum_slab_bytes, float* dY, int64_t M, int32_t N, const float* dP,
                          const int32_t* node_graph, const int32_t* gptr, void* stream);

/* ---------------------------------------------------------------------------------------
 * Graph-level classifier head, models.py:196-201 + :213-216:
 *   logits = Linear2(dropout(relu(Linear1(P)))),  P [B,H] (one row per graph), W1 [H2,H], W2 [C,H2].
 * One kernel each way instead of six small library GEMM/elementwise launches.
 * forward also writes H1 [B,H2] (the activations after ReLU and dropout) and fac [B,H2] (relu' *
 * dropout' factor).  backward: dP [B,H] and a slab [cgnn_head_grid(B,H2)][WD] of per-workgroup
 * partial parameter gradients, row layout dW1 [H2*H] | db1 [H2] | dW2 [C*H2] | db2 [C]
 * (WD = their sum), combined with cgnn_slab_reduce_f32(slab, rows, 1, WD, WD, out, WD).
 * cgnn_head_supported: H <= 128, H2 <= 64 and a divisor of 256, C <= 16.
 * ------------------------------------------------------------------------------------- */
int cgnn_head_supported(int32_t H, int32_t H2, int32_t C);
int cgnn_head_grid(int32_t B, int32_t H, int32_t H2, int32_t C);   /* rows of cgnn_head_bwd_f32's slab */
int cgnn_head_fwd_f32(const float* P, int32_t B, int32_t H, int32_t H2, int32_t C, const float* W1,
                      const float* b1, const float* W2, const float* b2, float p_drop,
                      uint64_t seed, const uint32_t* seed_dev, float* H1, float* fac, float* logits,
                      void* stream);
int cgnn_head_bwd_f32(const float* dlogits, const float* P, const float* H1, const float* fac,
                      int32_t B, int32_t H, int32_t H2, int32_t C, const float* W1, const float* W2,
                      float* dP, float* slab, int64_t slab_bytes, void* stream);

/* Mean cross-entropy of the step (torch.nn.CrossEntropyLoss defaults; reference train.py:39,49):
 * loss[0] = mean_i(logsumexp(logits[i,:]) - logits[i, labels[i]]); dlogits [B,C] = its gradient
 * for a unit upstream gradient, (softmax - onehot) / V.  labels int64 [B].  Rows labelled -100
 * (torch's default ignore_index) contribute nothing, get a zero gradient and are left out of
 * the row count V; any other label outside [0, C) -- where torch raises -- turns the loss and
 * that row's gradient into NaN (a kernel cannot raise; the NaN surfaces at the next read-back). */
int cgnn_cross_entropy_f32(const float* logits, const int64_t* labels, int32_t B, int32_t C,
                           float* loss, float* dlogits, void* stream);

/* The classifier's whole training step in one launch -- cgnn_head_fwd_f32, cgnn_cross_entropy_f32 and
 * cgnn_head_bwd_f32 (models.py:196-201,213-216 and train.py:46-50 with the loss's unit upstream gradient),
 * the same arithmetic row by row -- for the register-tiled head shapes (C = 2, H = 2 * H2 in {32, 64, 128,
 * 256}; else CGNN_EUNSUPPORTED).  Outputs: H1, fac, logits as the forward; dP [B,H]; slab
 * [cgnn_head_loss_grid(B,H,H2,C)][WD + 1]: the backward's row layout plus one column holding the workgroup's share
 * of the loss (sum of its rows' losses / V), so cgnn_slab_reduce_f32(slab, rows, 1, WD + 1, WD + 1, out,
 * WD + 1) yields the parameter gradients and out[WD] = the mean loss.  V is counted from `labels` by every
 * workgroup (no cross-workgroup wait); label rules as cgnn_cross_entropy_f32. */
int cgnn_head_loss_grid(int32_t B, int32_t H, int32_t H2, int32_t C);   /* rows of cgnn_head_loss_f32's slab */
int cgnn_head_loss_f32(const float* P, int32_t B, int32_t H, int32_t H2, int32_t C, const float* W1,
                       const float* b1, const float* W2, const float* b2, const int64_t* labels,
                       float p_drop, uint64_t seed, const uint32_t* seed_dev, float* H1, float* fac,
                       float* logits, float* dP, float* slab, int64_t slab_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CGNN_H */
