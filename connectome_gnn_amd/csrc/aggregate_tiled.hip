// aggregate_tiled.hip -- LDS-staged edge-weighted aggregation for wide features (gfx950, fp32).
//
//   Y[r,:] = post(r) * sum_{e in row r} w_e * pre(c_e) * X[c_e,:]   (+ bias) (+ Yadd[r,:])
//
// Replaces gather -> mul -> scatter_add_ of models.py:112-114 (GCN, pre = post = D^-1/2 with the
// self-loop carried by the ELL) and :146-149 (SAGE, post = 1/(wsum + 1e-8), no self-loop), and
// their autograd transposes on the source-sorted ELL.
//
// cgnn_aggregate_f32 (aggregate.hip) gathers every neighbour row from L2/HBM -- 14x the bytes of
// the feature matrix at the 360-ROI density -- and is latency-bound at ~1.7 TB/s of useful
// traffic.  Graphs of a connectome batch are small (<= 384 nodes), so here one persistent
// workgroup per CU stages a [tile rows x 64 columns] slice in LDS (96 KB) with coalesced 16-byte
// loads -- read once from HBM -- and every neighbour row comes out of LDS (agg_block.h: metadata in
// registers, DPP broadcast, ds_read_b128).  The next slice's rows are requested into registers
// while the current one is aggregated.
#include "agg_block.h"
#include "drop_ew.h"

namespace {

#define TA_LD ld4
#define TA_ST st4

constexpr int TA_MAXR = CGNN_FUSED_MAX_ROWS;   // 384
#define CGNN_TA_NW 12
#define CGNN_TA_G 1
constexpr int TA_NW = CGNN_TA_NW;
constexpr int TA_THR = TA_NW * 64;
constexpr int TA_RPP = TA_THR / 16;            // rows per staging pass (16 lanes per row)
constexpr int TA_PF = (TA_MAXR + TA_RPP - 1) / TA_RPP;   // rows per thread in the staging pass
constexpr int TA_NR = (TA_MAXR / 16 + TA_NW - 1) / TA_NW; // 16-row blocks per wave

// Optional prologue: X is the PRE-BatchNorm array Z of the previous layer and the kernel applies
// X' = drop(act(a Z + b)) while staging -- the arithmetic, keep bits and mask bytes of
// k_bn_act_apply<false> (elementwise.hip) -- and also writes X' (the projection reads it too): the
// stand-alone apply pass, one read of the array and one launch, disappear.
struct AggPre {
  const float* coef;        // [a | b | ...] of cgnn_bn_act_finalize, or NULL: no prologue
  int relu, use_drop, N;    // N = row width of Z (columns of the whole array)
  DropCfg drop;
  uint8_t* mask_out;        // [M][N/4] or NULL
  float* Xout;              // [M][ldxo]
  int64_t ldxo;
};

__global__ void __launch_bounds__(TA_THR) k_agg_tiled(
    cgnn_tiles t, int flags, const float* __restrict__ X, int64_t ldx, int nslices,
    const float* __restrict__ pre, const float* __restrict__ post, const float* __restrict__ bias,
    const float* Yadd, int64_t ldadd, float* Y, int64_t ldy, AggPre pr) {
  if (pr.coef && pr.drop.dev_key) pr.drop.key1 ^= pr.drop.dev_key[0];
  __shared__ __attribute__((aligned(16))) float tile[TA_MAXR * 64];
  __shared__ float postl[TA_MAXR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
  const bool transposed = flags & CGNN_AGG_TRANSPOSED, pre_div = flags & CGNN_AGG_PRE_DIV;
  const bool post_div = flags & CGNN_AGG_POST_DIV, accumulate = Yadd != nullptr;
  const uint4* ent = static_cast<const uint4*>(transposed ? t.ent_src : t.ent_dst);
  const int32_t* blk_off = transposed ? t.blk_off_src : t.blk_off_dst;
  const int units = t.num_tiles * nslices;

  // staged rows of the next unit: row (tid>>4) + 32u, columns 64*slice + 4j..+3
  float4 pfx[TA_PF];
  float pfs[TA_PF], pfp[TA_PF];
  auto request = [&](int u) {
    const int tid = u / nslices, slice = u - tid * nslices;
    const int nb = t.tile_ptr[tid], nn = t.tile_ptr[tid + 1] - nb;
#pragma unroll
    for (int k = 0; k < TA_PF; ++k) {
      const int row = (threadIdx.x >> 4) + TA_RPP * k;
      pfx[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      pfs[k] = 1.f;
      pfp[k] = 1.f;
      if (row < nn) {
        pfx[k] = TA_LD(X + (int64_t)(nb + row) * ldx + 64 * slice + 4 * j);
        if (pre) pfs[k] = pre[nb + row];
        if (post) pfp[k] = post[nb + row];
      }
    }
  };
  if ((int)blockIdx.x < units) request(blockIdx.x);

  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int tid = u / nslices, slice = u - tid * nslices;
    const int base = t.tile_ptr[tid];
    const int n = t.tile_ptr[tid + 1] - base;
    const int nblk = (n + 15) >> 4;
    const int gb0 = t.tile_blk[tid];
    int boff[TA_NR + 1] = {}, bwid[TA_NR + 1] = {};
#pragma unroll
    for (int k = 0; k < TA_NR; ++k) {
      const int bb = cgnn_uniform(wave) + TA_NW * k;
      if (bb < nblk) {
        boff[k] = blk_off[gb0 + bb];
        bwid[k] = (blk_off[gb0 + bb + 1] - boff[k]) >> 4;
      }
    }
    int off0 = boff[0], width = bwid[0], bk = 0;
    MetaRegs m;
    if (wave < nblk) m = meta_issue<true>(ent + (off0 >> 1), width, q, j);

    // ---- stage the slice (rows pre-scaled)
    float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pr.coef) {
      ca = ld4(pr.coef + 64 * slice + 4 * j);
      cb = ld4(pr.coef + pr.N + 64 * slice + 4 * j);
    }
#pragma unroll
    for (int k = 0; k < TA_PF; ++k) {
      const int row = (threadIdx.x >> 4) + TA_RPP * k;
      if (row < nblk * 16) {
        float4 x = pfx[k];
        if (pr.coef && row < n) {
          const int64_t i = (int64_t)(base + row) * (pr.N >> 2) + 16 * slice + j;     // chunk index
          const float zx = fmaf(ca.x, x.x, cb.x), zy = fmaf(ca.y, x.y, cb.y);
          const float zz = fmaf(ca.z, x.z, cb.z), zw = fmaf(ca.w, x.w, cb.w);
          uint32_t kb = 0xFu;
          if (pr.use_drop) {
            kb = drop_bits(pr.drop, (uint32_t)i);
            if (pr.mask_out) pr.mask_out[i] = (uint8_t)kb;
          }
          const float fx = ((!pr.relu || zx > 0.f) && (kb & 1u)) ? pr.drop.scale : 0.f;
          const float fy = ((!pr.relu || zy > 0.f) && (kb & 2u)) ? pr.drop.scale : 0.f;
          const float fz = ((!pr.relu || zz > 0.f) && (kb & 4u)) ? pr.drop.scale : 0.f;
          const float fw = ((!pr.relu || zw > 0.f) && (kb & 8u)) ? pr.drop.scale : 0.f;
          x = make_float4(zx * fx, zy * fy, zz * fz, zw * fw);
          st4(pr.Xout + (int64_t)(base + row) * pr.ldxo + 64 * slice + 4 * j, x);
        }
        const float s = pre_div ? 1.0f / pfs[k] : pfs[k];
        st4(tile + row * 64 + 4 * j, pre_div ? make_float4(x.x / pfs[k], x.y / pfs[k],
                                                          x.z / pfs[k], x.w / pfs[k])
                                             : scale4(x, s));
        if (j == 0) postl[row] = pfp[k];
      }
    }
    __syncthreads();
    if (u + (int)gridDim.x < units) request(u + gridDim.x);

    // ---- aggregate 16-row blocks out of LDS
    const float4 b4 = bias ? ld4(bias + 64 * slice + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = wave; b < nblk; b += TA_NW) {
      ++bk;
      int off1 = 0, width1 = 0;
#pragma unroll
      for (int k = 1; k < TA_NR; ++k)
        if (bk == k) { off1 = boff[k]; width1 = bwid[k]; }
      float4 yo[4];
      if (accumulate) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = 16 * b + 4 * q + it;
          yo[it] = row < n ? TA_LD(Yadd + (int64_t)(base + row) * ldadd + 64 * slice + 4 * j)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      float4 ag[4];
      agg_block<CGNN_TA_G, true>(tile, m, ent + (off0 >> 1), width, q, j, ag);
      if (b + TA_NW < nblk) m = meta_issue<true>(ent + (off1 >> 1), width1, q, j);
      off0 = off1; width = width1;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = 16 * b + 4 * q + it;
        if (row < n) {
          const float p = postl[row];
          float4 v = post_div ? make_float4(ag[it].x / p, ag[it].y / p, ag[it].z / p, ag[it].w / p)
                              : scale4(ag[it], p);
          v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          if (accumulate) { v.x += yo[it].x; v.y += yo[it].y; v.z += yo[it].z; v.w += yo[it].w; }
          TA_ST(Y + (int64_t)(base + row) * ldy + 64 * slice + 4 * j, v);
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int cgnn_aggregate_tiled_f32(const cgnn_tiles* t, int32_t flags, const float* X,
                                        int64_t ldx, int32_t F, const float* pre,
                                        const float* post, const float* bias, const float* Yadd,
                                        int64_t ldadd, float* Y, int64_t ldy, void* stream) {
  if (!t || t->num_nodes < 0 || t->num_tiles < 0 || F <= 0 || ldx < F || ldy < F) return CGNN_EINVAL;
  if (Yadd && ldadd < F) return CGNN_EINVAL;
  if (F % 64 || ldx % 4 || ldy % 4 || (Yadd && ldadd % 4) || t->max_tile_rows > TA_MAXR)
    return CGNN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) |
       reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(Yadd)) & 15)
    return CGNN_EUNSUPPORTED;
  if (t->num_nodes == 0 || t->num_tiles == 0) return CGNN_OK;
  const bool tr = flags & CGNN_AGG_TRANSPOSED;
  if (!X || !Y || !t->tile_ptr || !t->tile_blk || !(tr ? t->ent_src : t->ent_dst) ||
      !(tr ? t->blk_off_src : t->blk_off_dst))
    return CGNN_EINVAL;
  k_agg_tiled<<<cgnn_fused_grid(), TA_THR, 0, cgnn_stream(stream)>>>(*t, flags, X, ldx, F / 64, pre,
                                                                     post, bias, Yadd, ldadd, Y, ldy, AggPre{});
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

// cgnn_aggregate_tiled_f32 on X' = drop(act(a Z + b)) with X' formed while staging (and written to
// Xout, and its keep bytes to mask_out): cgnn_bn_act_fwd_apply + cgnn_aggregate_tiled_f32 in one
// launch, bit for bit (same arithmetic, same keep bits).  coef = the [a | b | mean | invstd] block
// of cgnn_bn_act_finalize over N = F columns.
extern "C" int cgnn_aggregate_tiled_bn_f32(const cgnn_tiles* t, int32_t flags, const float* Z,
                                           int64_t ldz, int32_t F, const float* pre,
                                           const float* post, const float* bias, float* Y,
                                           int64_t ldy, const float* coef, int32_t relu,
                                           float p_drop, uint64_t seed, const uint32_t* seed_dev,
                                           uint8_t* mask_out, float* Xout, int64_t ldxo,
                                           void* stream) {
  if (!t || t->num_nodes < 0 || t->num_tiles < 0 || F <= 0 || ldz < F || ldy < F || ldxo < F) return CGNN_EINVAL;
  if (p_drop < 0.f || p_drop >= 1.f || !coef || !Xout) return CGNN_EINVAL;
  if (F % 64 || ldz % 4 || ldy % 4 || ldxo % 4 || t->max_tile_rows > TA_MAXR) return CGNN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(bias) |
       reinterpret_cast<uintptr_t>(Xout) | reinterpret_cast<uintptr_t>(coef)) & 15)
    return CGNN_EUNSUPPORTED;
  if (t->num_nodes == 0 || t->num_tiles == 0) return CGNN_OK;
  const bool tr = flags & CGNN_AGG_TRANSPOSED;
  if (!Z || !Y || !t->tile_ptr || !t->tile_blk || !(tr ? t->ent_src : t->ent_dst) ||
      !(tr ? t->blk_off_src : t->blk_off_dst))
    return CGNN_EINVAL;
  AggPre pr;
  pr.coef = coef;
  pr.relu = relu;
  pr.N = F;
  pr.drop = make_drop(p_drop, seed, &pr.use_drop);
  pr.drop.dev_key = seed_dev;
  pr.mask_out = mask_out;
  pr.Xout = Xout;
  pr.ldxo = ldxo;
  k_agg_tiled<<<cgnn_fused_grid(), TA_THR, 0, cgnn_stream(stream)>>>(*t, flags, Z, ldz, F / 64, pre, post, bias,
                                                                     nullptr, 0, Y, ldy, pr);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}
