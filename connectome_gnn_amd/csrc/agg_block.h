// agg_block.h -- LDS-tile neighbour aggregation shared by the fused GCN kernels
// (fused_gcn.hip) and the tiled aggregate (aggregate_tiled.hip).  gfx950, wave64.
//
// A tile = a run of whole graphs, [rows][64] fp32 in LDS.  Each wave aggregates 16-row blocks:
// lane (q, j) owns rows 4q..4q+3 of the block and columns 4j..4j+3.  The block's blocked-ELL
// entries (cgnn_bell_fill: byte offset of the neighbour's row in the tile, raw edge weight,
// zero-weight padding up to the block's width) are held in registers and broadcast inside the
// 16-lane group with DPP row_newbcast; neighbour rows are ds_read_b128 out of the tile.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float4 scale4(const float4& v, float s) {
  return make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
}

// ------------------------------------------------------------------------------------------
// Aggregate one 16-row block out of the LDS tile.  Lane (q, j): rows 4q..4q+3 of the block,
// columns 4j..4j+3.  `mp` = the block's blocked-ELL entries (8 uint4 = 16 entries per step);
// the lane group reads its 4 entries of a step as two 16-byte loads (group-uniform address)
// and is always one step ahead of the LDS reads.  Branch-free: padding entries have weight 0.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void fma4(float4& acc, uint32_t wbits, const float4& v) {
  const float w = __uint_as_float(wbits);
  acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y);
  acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}

__device__ __forceinline__ float4 ldsrow(const char* tb, uint32_t off) {
  return *reinterpret_cast<const float4*>(tb + off);
}

// Per-wave metadata pipeline.  Lane (q, j) keeps, in registers, the 4 entries (32 bytes) its
// row group needs at step j (batch 0) and step 16+j (batch 1): one coalesced 2 KB wave load
// per 16 steps, issued one block ahead (in flight during the previous block's MFMAs).  Inside
// the loop an entry is broadcast to the 16 lanes of the group with DPP row_newbcast, so the
// only LDS traffic is the neighbour rows themselves and every row-read address is available
// without a memory round trip.  Steps >= 32 (very high degree) come from global memory.
struct MetaRegs { uint4 a0, a1, b0, b1; };
// blocked-ELL entries: default cache policy.  They are read once per LAUNCH but by several launches
// of a step (212 MB, within reach of the 256 MB MALL): non-temporal loads cost the forward kernel 9 us
#define CGNN_META_LD(p) (*(p))

// AHEAD2: also prefetch batch 1 a block ahead (8 more live VGPRs); otherwise batch 1 is
// requested at the start of its own block's aggregation and lands while batch 0 is processed.
template <bool AHEAD2>
__device__ __forceinline__ MetaRegs meta_issue(const uint4* __restrict__ mp, int width, int q, int j) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  MetaRegs m{z, z, z, z};
  if (j < width) { m.a0 = CGNN_META_LD(mp + 8 * j + 2 * q); m.a1 = CGNN_META_LD(mp + 8 * j + 2 * q + 1); }
  if (AHEAD2 && 16 + j < width) {
    m.b0 = CGNN_META_LD(mp + 8 * (16 + j) + 2 * q);
    m.b1 = CGNN_META_LD(mp + 8 * (16 + j) + 2 * q + 1);
  }
  return m;
}

#define CGNN_BC(v, S) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x150 + (S), 0xf, 0xf, true))
// One step, unconditional: entries past the block's width are (offset 0, weight 0).
#define CGNN_AGG_STEP(M0, M1, S)                                                            \
  {                                                                                         \
    const uint32_t o0 = CGNN_BC(M0.x, S), w0 = CGNN_BC(M0.y, S), o1 = CGNN_BC(M0.z, S),     \
                   w1 = CGNN_BC(M0.w, S), o2 = CGNN_BC(M1.x, S), w2 = CGNN_BC(M1.y, S),     \
                   o3 = CGNN_BC(M1.z, S), w3 = CGNN_BC(M1.w, S);                            \
    fma4(acc[0], w0, ldsrow(tb, o0)); fma4(acc[1], w1, ldsrow(tb, o1));                     \
    fma4(acc[2], w2, ldsrow(tb, o2)); fma4(acc[3], w3, ldsrow(tb, o3));                     \
  }
// G steps = 4G independent 16-byte LDS reads in flight (16G VGPRs), then 16G FMAs; the
// sched_barrier stops the scheduler from hoisting every read of the block (256 VGPRs -> spills).
#define CGNN_AGG_2(M0, M1, S)                                                               \
  CGNN_AGG_STEP(M0, M1, S) CGNN_AGG_STEP(M0, M1, (S) + 1) __builtin_amdgcn_sched_barrier(0);
#define CGNN_AGG_4W(M0, M1, S)                                                              \
  CGNN_AGG_STEP(M0, M1, S) CGNN_AGG_STEP(M0, M1, (S) + 1) CGNN_AGG_STEP(M0, M1, (S) + 2)    \
  CGNN_AGG_STEP(M0, M1, (S) + 3) __builtin_amdgcn_sched_barrier(0);
#define CGNN_AGG_4(M0, M1, S)                                                               \
  if (G >= 4) { CGNN_AGG_4W(M0, M1, S) } else { CGNN_AGG_2(M0, M1, S) CGNN_AGG_2(M0, M1, (S) + 2) }

// `width` must be wave-uniform.  Steps run in straight-line groups (no per-step branch: a
// branch per step would fence the scheduler and expose the LDS latency of every step).
// G = steps whose row reads may be in flight together (register budget of the caller).
// scalar-branched forms: one basic block per step (or per pair of steps; the second step of a
// pair may be padding, which is harmless: weight 0, row 0)
#define CGNN_AGG_STEP_IF(M0, M1, S, W) if ((S) < (W)) CGNN_AGG_STEP(M0, M1, S)
#define CGNN_AGG_PAIR_IF(M0, M1, S, W) \
  if ((S) < (W)) { CGNN_AGG_STEP(M0, M1, S) CGNN_AGG_STEP(M0, M1, (S) + 1) }
#define CGNN_AGG_16_IF(M0, M1, W)                                                             \
  CGNN_AGG_STEP_IF(M0, M1, 0, W) CGNN_AGG_STEP_IF(M0, M1, 1, W) CGNN_AGG_STEP_IF(M0, M1, 2, W)     \
  CGNN_AGG_STEP_IF(M0, M1, 3, W) CGNN_AGG_STEP_IF(M0, M1, 4, W) CGNN_AGG_STEP_IF(M0, M1, 5, W)     \
  CGNN_AGG_STEP_IF(M0, M1, 6, W) CGNN_AGG_STEP_IF(M0, M1, 7, W) CGNN_AGG_STEP_IF(M0, M1, 8, W)     \
  CGNN_AGG_STEP_IF(M0, M1, 9, W) CGNN_AGG_STEP_IF(M0, M1, 10, W) CGNN_AGG_STEP_IF(M0, M1, 11, W)   \
  CGNN_AGG_STEP_IF(M0, M1, 12, W) CGNN_AGG_STEP_IF(M0, M1, 13, W) CGNN_AGG_STEP_IF(M0, M1, 14, W)  \
  CGNN_AGG_STEP_IF(M0, M1, 15, W)
#define CGNN_AGG_16_PAIRS(M0, M1, W)                                                          \
  CGNN_AGG_PAIR_IF(M0, M1, 0, W) CGNN_AGG_PAIR_IF(M0, M1, 2, W) CGNN_AGG_PAIR_IF(M0, M1, 4, W)     \
  CGNN_AGG_PAIR_IF(M0, M1, 6, W) CGNN_AGG_PAIR_IF(M0, M1, 8, W) CGNN_AGG_PAIR_IF(M0, M1, 10, W)    \
  CGNN_AGG_PAIR_IF(M0, M1, 12, W) CGNN_AGG_PAIR_IF(M0, M1, 14, W)

template <int G, bool AHEAD2>
__device__ __forceinline__ void agg_block(const float* __restrict__ tile, MetaRegs m,
                                          const uint4* __restrict__ mp, int width, int q, int j,
                                          float4 (&acc)[4]) {
  const char* tb = reinterpret_cast<const char*>(tile) + 16 * j;
  if (!AHEAD2 && 16 + j < width) {
    m.b0 = CGNN_META_LD(mp + 8 * (16 + j) + 2 * q);
    m.b1 = CGNN_META_LD(mp + 8 * (16 + j) + 2 * q + 1);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (G == 1) {
    // lowest register pressure: one scalar-branched basic block per step (the generic backward
    // kernel holds 64 dW accumulators and cannot afford more rows in flight)
    CGNN_AGG_16_IF(m.a0, m.a1, width)
    if (width > 16) {
      const int w1 = width - 16;
      CGNN_AGG_16_IF(m.b0, m.b1, w1)
    }
  } else if (G == 2) {
    // scalar-branched pairs: 8 row reads in flight per basic block
    CGNN_AGG_16_PAIRS(m.a0, m.a1, width)
    if (width > 16) {
      const int w1 = width - 16;
      CGNN_AGG_16_PAIRS(m.b0, m.b1, w1)
    }
  } else {
    if (width >= 13) {
      CGNN_AGG_4(m.a0, m.a1, 0) CGNN_AGG_4(m.a0, m.a1, 4) CGNN_AGG_4(m.a0, m.a1, 8) CGNN_AGG_4(m.a0, m.a1, 12)
    } else {
      CGNN_AGG_4(m.a0, m.a1, 0)
      if (width > 4) { CGNN_AGG_4(m.a0, m.a1, 4) }
      if (width > 8) { CGNN_AGG_4(m.a0, m.a1, 8) }
    }
    if (width > 16) {
      CGNN_AGG_4(m.b0, m.b1, 0)
      if (width > 20) { CGNN_AGG_4(m.b0, m.b1, 4) }
      if (width > 24) { CGNN_AGG_4(m.b0, m.b1, 8) CGNN_AGG_4(m.b0, m.b1, 12) }
    }
  }
  for (int s = 32; s < width; ++s) {              // overflow steps: from global
    const uint4 e0 = mp[8 * s + 2 * q], e1 = mp[8 * s + 2 * q + 1];
    fma4(acc[0], e0.y, ldsrow(tb, e0.x)); fma4(acc[1], e0.w, ldsrow(tb, e0.z));
    fma4(acc[2], e1.y, ldsrow(tb, e1.x)); fma4(acc[3], e1.w, ldsrow(tb, e1.z));
  }
}


}  // namespace
