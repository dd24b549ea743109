// fused_gcn.hip -- per-tile fused GCN layer kernels for gfx950 (hidden = 64, fp32).
//
// Reference arithmetic replaced: GCNLayer.forward (models.py:84-114), the BatchNorm1d / ReLU /
// dropout chain and mean-pool of GCNConnectome.encode (models.py:203-211), and autograd's
// backward of all of it.  See include/cgnn.h ("FUSED PER-TILE GCN PATH") for the contract.
//
// Execution shape (one persistent workgroup = 8 waves per CU, one CU = one tile at a time):
//
//   LDS  tile [<=384 rows][64] fp32  (96 KB)  the layer input (fwd) / dY (bwd) of the tile
//        stg  [8 waves][16][68] fp32 (34 KB)  per-wave 16-row block handed to the matrix core
//        Wl   [64][64] fp32          (16 KB)  projection weight (backward only)
//
//   phase A  all 512 threads stream the tile from HBM with 16-byte loads, apply the fused
//            elementwise prologue (BatchNorm-apply+ReLU+dropout, or BatchNorm-backward), and
//            write it to LDS.
//   phase B  each wave owns 16-row blocks.  Four rows at a time (16 lanes x float4 per row) it
//            walks the rows' CSR slots -- 16 slots of (col, coef) are fetched with one coalesced
//            load and broadcast inside the 16-lane row with DPP row_newbcast -- and accumulates
//            neighbour rows straight out of the LDS tile with ds_read_b128.  The finished block
//            goes through `stg` to v_mfma_f32_16x16x4_f32 (exact fp32) and the epilogue
//            (bias / BatchNorm statistics / ReLU' * dropout') runs on the accumulators.
//
// The MFMA reduction index is permuted freely (lane group kk supplies k = 16*kk + s) so that
// every operand fragment is a run of 16-byte LDS/register accesses; output tile tj holds the
// columns {4*c + tj}, so each lane ends up with 4 consecutive columns of 4 rows = float4 stores.
//
// No atomics anywhere: per-workgroup partial sums go to slabs reduced in a fixed order.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HID = CGNN_FUSED_HIDDEN;      // 64
constexpr int NWAVE = 8;
constexpr int NTHR = NWAVE * 64;            // 512
constexpr int SLD = 68;                     // staging row stride (floats)
constexpr int STG_FLOATS = 16 * SLD;        // per wave

struct DropCfg {
  uint32_t thr16;     // keep iff 16-bit hash >= thr16  (thr16 = round(p * 65536))
  float scale;        // 1 / (1 - p)
  uint32_t key0, key1;
};

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// 4 keep-bits for the float4 at (global row, chunk); bit i <-> column 4*chunk + i.
__device__ __forceinline__ uint32_t drop_bits(const DropCfg& d, uint32_t row, uint32_t chunk) {
  const uint32_t e = (row * 16u + chunk) * 2u;
  const uint32_t h0 = mix32(mix32(e ^ d.key0) + d.key1);
  const uint32_t h1 = mix32(mix32((e + 1u) ^ d.key0) + d.key1);
  uint32_t b = 0;
  b |= ((h0 & 0xFFFFu) >= d.thr16) ? 1u : 0u;
  b |= ((h0 >> 16) >= d.thr16) ? 2u : 0u;
  b |= ((h1 & 0xFFFFu) >= d.thr16) ? 4u : 0u;
  b |= ((h1 >> 16) >= d.thr16) ? 8u : 0u;
  return b;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

// x = drop(relu(a*y + b)); returns x, and the combined (z>0 & keep) factor per component in f.
__device__ __forceinline__ float4 act4(const float4& y, const float4& a, const float4& b,
                                        uint32_t keep, float scale, float4& f) {
  float4 z, x;
  z.x = fmaf(a.x, y.x, b.x); z.y = fmaf(a.y, y.y, b.y);
  z.z = fmaf(a.z, y.z, b.z); z.w = fmaf(a.w, y.w, b.w);
  f.x = (z.x > 0.f && (keep & 1u)) ? scale : 0.f;
  f.y = (z.y > 0.f && (keep & 2u)) ? scale : 0.f;
  f.z = (z.z > 0.f && (keep & 4u)) ? scale : 0.f;
  f.w = (z.w > 0.f && (keep & 8u)) ? scale : 0.f;
  x.x = z.x * f.x; x.y = z.y * f.y; x.z = z.z * f.z; x.w = z.w * f.w;
  return x;
}

// ------------------------------------------------------------------------------------------
// 4 rows per wave (16 lanes x float4 each): acc = sum_slots coef * tile[col - base] (+ self).
// `row` is this lane-group's local row, `valid` whether it exists.
// ------------------------------------------------------------------------------------------
#define CGNN_AGG_STEP(S)                                                                        \
  {                                                                                             \
    const int c_ = __builtin_amdgcn_update_dpp(0, mycol, 0x150 + (S), 0xf, 0xf, false);         \
    const float w_ = __int_as_float(                                                            \
        __builtin_amdgcn_update_dpp(0, __float_as_int(myw), 0x150 + (S), 0xf, 0xf, false));     \
    if ((S) < cnt) {                                                                            \
      const float4 v_ = ld4(tile + c_ * HID + 4 * j);                                           \
      acc.x = fmaf(w_, v_.x, acc.x); acc.y = fmaf(w_, v_.y, acc.y);                             \
      acc.z = fmaf(w_, v_.z, acc.z); acc.w = fmaf(w_, v_.w, acc.w);                             \
    }                                                                                           \
  }

__device__ __forceinline__ float4 agg_rows4(const float* __restrict__ tile, int base, int row,
                                            bool valid, int j, const int32_t* __restrict__ rowptr,
                                            const int32_t* __restrict__ col,
                                            const float* __restrict__ coef,
                                            const float* __restrict__ selfc) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int beg = 0, end = 0;
  if (valid) {
    beg = rowptr[base + row];
    end = rowptr[base + row + 1];
  }
  for (int s0 = beg; __any(s0 < end); s0 += 16) {
    const int cnt = end - s0;                 // may be <= 0 for finished groups
    int mycol = 0;
    float myw = 0.f;
    if (j < cnt) {
      mycol = col[s0 + j] - base;
      myw = coef[s0 + j];
    }
    CGNN_AGG_STEP(0) CGNN_AGG_STEP(1) CGNN_AGG_STEP(2) CGNN_AGG_STEP(3)
    CGNN_AGG_STEP(4) CGNN_AGG_STEP(5) CGNN_AGG_STEP(6) CGNN_AGG_STEP(7)
    CGNN_AGG_STEP(8) CGNN_AGG_STEP(9) CGNN_AGG_STEP(10) CGNN_AGG_STEP(11)
    CGNN_AGG_STEP(12) CGNN_AGG_STEP(13) CGNN_AGG_STEP(14) CGNN_AGG_STEP(15)
  }
  if (valid && selfc) {                       // the appended self-loop, last (models.py:98-100)
    const float sc = selfc[base + row];
    const float4 v = ld4(tile + row * HID + 4 * j);
    acc.x = fmaf(sc, v.x, acc.x); acc.y = fmaf(sc, v.y, acc.y);
    acc.z = fmaf(sc, v.z, acc.z); acc.w = fmaf(sc, v.w, acc.w);
  }
  return acc;
}

// Reduce per-lane fp64 column partials (lane (q,j): columns 4j..4j+3) over the workgroup and
// write slab_row[0..63] (= s1) and slab_row[64..127] (= s2).  `red` >= 8*128 doubles of LDS.
__device__ __forceinline__ void reduce_stats(double (&s1)[4], double (&s2)[4], double* red,
                                             double* __restrict__ slab_row) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s1[i] += __shfl_xor(s1[i], 16, 64); s1[i] += __shfl_xor(s1[i], 32, 64);
    s2[i] += __shfl_xor(s2[i], 16, 64); s2[i] += __shfl_xor(s2[i], 32, 64);
  }
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[wave * 128 + 4 * j + i] = s1[i];
      red[wave * 128 + 64 + 4 * j + i] = s2[i];
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) t += red[w * 128 + threadIdx.x];
    slab_row[threadIdx.x] = t;
  }
  __syncthreads();
}

// ==========================================================================================
// forward
// ==========================================================================================
template <int MAXR, bool FIRST>
__global__ void __launch_bounds__(NTHR) k_gcn_fwd(
    cgnn_tiles t, const float* __restrict__ Xin, int F0, const float* __restrict__ bn_prev,
    DropCfg drop, int use_drop, uint8_t* __restrict__ mask_out, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ Y, double* __restrict__ stat_slab) {
  __shared__ __attribute__((aligned(16))) float tile[MAXR * HID];
  __shared__ __attribute__((aligned(16))) float stg_all[NWAVE * STG_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
  float* stg = stg_all + wave * STG_FLOATS;

  // B-operand fragments of the projection, resident in registers for the whole kernel.
  //   generic: B[k][col] = W[col][k], lane (kk=q, jj=j), tile tj <-> col 4j+tj, k = 16q + s
  //   first  : same with k = 4s + q < F0 (F0 <= 16 -> 4 k-steps)
  float wreg[4][16];
  if (FIRST) {
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 4 * s + q;
        wreg[tj][s] = k < F0 ? W[(4 * j + tj) * F0 + k] : 0.f;
      }
  } else {
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = ld4(W + (4 * j + tj) * HID + 16 * q + 4 * u);
        wreg[tj][4 * u + 0] = v.x; wreg[tj][4 * u + 1] = v.y;
        wreg[tj][4 * u + 2] = v.z; wreg[tj][4 * u + 3] = v.w;
      }
  }
  const float4 bias4 = ld4(bias + 4 * j);
  float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
  if (!FIRST) {
    pa = ld4(bn_prev + 4 * j);
    pb = ld4(bn_prev + HID + 4 * j);
  }
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};

  for (int tid = blockIdx.x; tid < t.num_tiles; tid += gridDim.x) {
    const int base = t.tile_ptr[tid];
    const int n = t.tile_ptr[tid + 1] - base;
    const int nblk = (n + 15) >> 4;

    // ---------------------------------------------------------------- phase A: fill the tile
    if (FIRST) {
      // T = X0 W0^T on the matrix core, written straight into the tile.
      for (int b = wave; b < nblk; b += NWAVE) {
        const int arow = 16 * b + j;                    // A operand: row i = j, k-slot kk = q
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int k = 4 * s + q;
          float av = 0.f;
          if (4 * s < F0) {
            if (arow < n && k < F0) av = Xin[(int64_t)(base + arow) * F0 + k];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
              acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wreg[tj][s], acc[tj], 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          st4(tile + (16 * b + 4 * q + r) * HID + 4 * j,
              make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]));
      }
    } else {
      for (int idx = threadIdx.x; idx < nblk * 256; idx += NTHR) {
        const int row = idx >> 4;                        // chunk == j (NTHR % 16 == 0)
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < n) {
          const float4 y = ld4(Xin + (int64_t)(base + row) * HID + 4 * j);
          uint32_t keep = 0xFu;
          if (use_drop) {
            keep = drop_bits(drop, (uint32_t)(base + row), (uint32_t)j);
            if (mask_out) mask_out[(int64_t)(base + row) * 16 + j] = (uint8_t)keep;
          }
          float4 f;
          x = act4(y, pa, pb, keep, drop.scale, f);
        }
        st4(tile + row * HID + 4 * j, x);
      }
    }
    __syncthreads();

    // ------------------------------------------------- phase B: aggregate (+ project) blocks
    for (int b = wave; b < nblk; b += NWAVE) {
      f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      if (FIRST) {
        // tile already holds T: Y = A_hat T + b, block rows come out in registers directly.
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = 16 * b + 4 * it + q;
          const float4 a = agg_rows4(tile, base, row, row < n, j, t.rowptr_dst, t.col_dst,
                                     t.coef_dst, t.selfc);
          if (row < n) {
            const float4 y = make_float4(a.x + bias4.x, a.y + bias4.y, a.z + bias4.z, a.w + bias4.w);
            st4(Y + (int64_t)(base + row) * HID + 4 * j, y);
            s1[0] += y.x; s1[1] += y.y; s1[2] += y.z; s1[3] += y.w;
            s2[0] += (double)y.x * y.x; s2[1] += (double)y.y * y.y;
            s2[2] += (double)y.z * y.z; s2[3] += (double)y.w * y.w;
          }
        }
        continue;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = 16 * b + 4 * it + q;
        const float4 a = agg_rows4(tile, base, row, row < n, j, t.rowptr_dst, t.col_dst,
                                   t.coef_dst, t.selfc);
        st4(stg + (4 * it + q) * SLD + 4 * j, a);
      }
      __builtin_amdgcn_wave_barrier();
      // A fragments: lane (i=j, kk=q) holds P[row j][k = 16q + s], s = 0..15
      float af[16];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = ld4(stg + j * SLD + 16 * q + 4 * u);
        af[4 * u + 0] = v.x; af[4 * u + 1] = v.y; af[4 * u + 2] = v.z; af[4 * u + 3] = v.w;
      }
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], wreg[tj][s], acc[tj], 0, 0, 0);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * b + 4 * q + r;
        if (row < n) {
          const float4 y = make_float4(acc[0][r] + bias4.x, acc[1][r] + bias4.y,
                                       acc[2][r] + bias4.z, acc[3][r] + bias4.w);
          st4(Y + (int64_t)(base + row) * HID + 4 * j, y);
          s1[0] += y.x; s1[1] += y.y; s1[2] += y.z; s1[3] += y.w;
          s2[0] += (double)y.x * y.x; s2[1] += (double)y.y * y.y;
          s2[2] += (double)y.z * y.z; s2[3] += (double)y.w * y.w;
        }
      }
    }
    __syncthreads();
  }
  if (stat_slab)
    reduce_stats(s1, s2, reinterpret_cast<double*>(tile), stat_slab + (int64_t)blockIdx.x * 128);
}

// ==========================================================================================
// backward
// ==========================================================================================
template <int MAXR, bool FIRST>
__global__ void __launch_bounds__(NTHR) k_gcn_bwd(
    cgnn_tiles t, const float* __restrict__ dZ, const float* __restrict__ Y,
    const float* __restrict__ bn, const float* __restrict__ bwc,
    const float* __restrict__ Xprev /* Yprev [Nn,64] or X0 [Nn,F0] */, int F0,
    const float* __restrict__ bn_prev, DropCfg drop, int use_drop,
    const uint8_t* __restrict__ mask_prev, const float* __restrict__ W,
    float* __restrict__ dZprev, double* __restrict__ s_slab, float* __restrict__ dW_slab,
    double* __restrict__ db_slab) {
  __shared__ __attribute__((aligned(16))) float tile[MAXR * HID];
  __shared__ __attribute__((aligned(16))) float stg_all[NWAVE * STG_FLOATS];
  __shared__ __attribute__((aligned(16))) float Wl[FIRST ? 4 : HID * HID];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
  float* stg = stg_all + wave * STG_FLOATS;

  if (!FIRST) {
    for (int i = threadIdx.x; i < HID * HID / 4; i += NTHR) st4(Wl + 4 * i, ld4(W + 4 * i));
  }
  // phase-A constants for this thread's 4 columns
  const float4 ca = ld4(bn + 4 * j), cmean = ld4(bn + 2 * HID + 4 * j), cis = ld4(bn + 3 * HID + 4 * j);
  const float4 c1 = ld4(bwc + 4 * j), c2 = ld4(bwc + HID + 4 * j);
  float4 pa = make_float4(0, 0, 0, 0), pb = pa, pmean = pa, pis = pa;
  if (!FIRST) {
    pa = ld4(bn_prev + 4 * j); pb = ld4(bn_prev + HID + 4 * j);
    pmean = ld4(bn_prev + 2 * HID + 4 * j); pis = ld4(bn_prev + 3 * HID + 4 * j);
  }
  // dW accumulators: FIRST: dw[ti][0] only (16 input columns); else dw[ti][tj].
  f32x4 dw[4][FIRST ? 1 : 4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b2 = 0; b2 < (FIRST ? 1 : 4); ++b2) dw[a][b2] = f32x4{0, 0, 0, 0};
  double db[4] = {0, 0, 0, 0};
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};

  for (int tid = blockIdx.x; tid < t.num_tiles; tid += gridDim.x) {
    const int base = t.tile_ptr[tid];
    const int n = t.tile_ptr[tid + 1] - base;
    const int nblk = (n + 15) >> 4;

    // ------------------------------------------- phase A: dY = BatchNorm'(dZ) into the tile
    for (int idx = threadIdx.x; idx < nblk * 256; idx += NTHR) {
      const int row = idx >> 4;
      float4 dy = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < n) {
        const float4 dz = ld4(dZ + (int64_t)(base + row) * HID + 4 * j);
        const float4 y = ld4(Y + (int64_t)(base + row) * HID + 4 * j);
        dy.x = ca.x * (dz.x - c1.x - (y.x - cmean.x) * cis.x * c2.x);
        dy.y = ca.y * (dz.y - c1.y - (y.y - cmean.y) * cis.y * c2.y);
        dy.z = ca.z * (dz.z - c1.z - (y.z - cmean.z) * cis.z * c2.z);
        dy.w = ca.w * (dz.w - c1.w - (y.w - cmean.w) * cis.w * c2.w);
        db[0] += dy.x; db[1] += dy.y; db[2] += dy.z; db[3] += dy.w;
      }
      st4(tile + row * HID + 4 * j, dy);
    }
    __syncthreads();

    // --------------------- phase B: dT = A_hat^T dY per block; dW += dT^T X; dZprev = ...
    for (int b = wave; b < nblk; b += NWAVE) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = 16 * b + 4 * it + q;
        const float4 a = agg_rows4(tile, base, row, row < n, j, t.rowptr_src, t.col_src,
                                   t.coef_src, t.selfc);
        st4(stg + (4 * it + q) * SLD + 4 * j, a);
      }
      __builtin_amdgcn_wave_barrier();

      if (FIRST) {
        // B operand: X0[row 4q+s][col j] (zero beyond F0); dW0[o][jcol] tile ti: o = 16ti+4q+r
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int row = 16 * b + 4 * q + s;
          float xv = 0.f;
          if (row < n && j < F0) xv = Xprev[(int64_t)(base + row) * F0 + j];
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) {
            const float av = stg[(4 * q + s) * SLD + 16 * ti + j];
            dw[ti][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xv, dw[ti][0], 0, 0, 0);
          }
        }
        __builtin_amdgcn_wave_barrier();
        continue;
      }

      // previous layer's block: rows 4q+r, columns 4j..4j+3 -> X (B operand of dW) and the
      // relu'/dropout' factor + xhat for the epilogue.
      float4 xb[4], fac[4], xh[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * b + 4 * q + r;
        xb[r] = fac[r] = xh[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < n) {
          const float4 y = ld4(Xprev + (int64_t)(base + row) * HID + 4 * j);
          uint32_t keep = 0xFu;
          if (use_drop) keep = mask_prev[(int64_t)(base + row) * 16 + j];
          xb[r] = act4(y, pa, pb, keep, drop.scale, fac[r]);
          xh[r] = make_float4((y.x - pmean.x) * pis.x, (y.y - pmean.y) * pis.y,
                              (y.z - pmean.z) * pis.z, (y.w - pmean.w) * pis.w);
        }
      }
      // dW[o][col] += sum_m dT[m][o] X[m][col]; k <-> m = 4q + s
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float bx[4] = {xb[s].x, xb[s].y, xb[s].z, xb[s].w};
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
          const float av = stg[(4 * q + s) * SLD + 16 * ti + j];
#pragma unroll
          for (int tj = 0; tj < 4; ++tj)
            dw[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bx[tj], dw[ti][tj], 0, 0, 0);
        }
      }
      // dX[row][col] = sum_o dT[row][o] W[o][col]; k <-> o = 16q + s
      f32x4 dx[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      float af[16];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = ld4(stg + j * SLD + 16 * q + 4 * u);
        af[4 * u + 0] = v.x; af[4 * u + 1] = v.y; af[4 * u + 2] = v.z; af[4 * u + 3] = v.w;
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float4 w4 = ld4(Wl + (16 * q + s) * HID + 4 * j);
        dx[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], w4.x, dx[0], 0, 0, 0);
        dx[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], w4.y, dx[1], 0, 0, 0);
        dx[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], w4.z, dx[2], 0, 0, 0);
        dx[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], w4.w, dx[3], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * b + 4 * q + r;
        if (row < n) {
          const float4 dzp = make_float4(dx[0][r] * fac[r].x, dx[1][r] * fac[r].y,
                                         dx[2][r] * fac[r].z, dx[3][r] * fac[r].w);
          st4(dZprev + (int64_t)(base + row) * HID + 4 * j, dzp);
          s1[0] += dzp.x; s1[1] += dzp.y; s1[2] += dzp.z; s1[3] += dzp.w;
          s2[0] += (double)dzp.x * xh[r].x; s2[1] += (double)dzp.y * xh[r].y;
          s2[2] += (double)dzp.z * xh[r].z; s2[3] += (double)dzp.w * xh[r].w;
        }
      }
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- workgroup reductions
  // db: 32 threads share a chunk j (threadIdx % 16); reduce through the tile memory.
  {
    double* red = reinterpret_cast<double*>(tile);       // [32][64]
    __syncthreads();
    const int g = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) red[g * 64 + 4 * j + i] = db[i];
    __syncthreads();
    if (threadIdx.x < 64) {
      double s = 0.0;
      for (int g2 = 0; g2 < NTHR / 16; ++g2) s += red[g2 * 64 + threadIdx.x];
      db_slab[(int64_t)blockIdx.x * 64 + threadIdx.x] = s;
    }
    __syncthreads();
  }
  if (!FIRST) reduce_stats(s1, s2, reinterpret_cast<double*>(tile), s_slab + (int64_t)blockIdx.x * 128);
  // dW: tree over the 8 waves through LDS (fixed order), wave 0 writes the partial.
  {
    constexpr int NTJ = FIRST ? 1 : 4;
    constexpr int PER = 64 * 16 * NTJ;                    // floats per wave partial
    float* red = tile;                                    // up to 4 * 4096 floats = 64 KB
    __syncthreads();
    for (int half = NWAVE / 2; half >= 1; half >>= 1) {
      if (wave >= half && wave < 2 * half) {
        float* dst = red + (wave - half) * PER;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int tj = 0; tj < NTJ; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((ti * NTJ + tj) * 4 + r) * 64 + lane] = dw[ti][tj][r];
      }
      __syncthreads();
      if (wave < half) {
        const float* src = red + wave * PER;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int tj = 0; tj < NTJ; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) dw[ti][tj][r] += src[((ti * NTJ + tj) * 4 + r) * 64 + lane];
      }
      __syncthreads();
    }
    if (wave == 0) {
      // element (ti, tj, r) of lane (q, j): o = 16ti + 4q + r ; col = FIRST ? j : 4j + tj
      constexpr int NC = FIRST ? 16 : HID;
      float* out = dW_slab + (int64_t)blockIdx.x * 64 * NC;
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * ti + 4 * q + r;
          if (FIRST) out[o * NC + j] = dw[ti][0][r];
          else st4(out + o * NC + 4 * j, make_float4(dw[ti][0][r], dw[ti][NTJ > 1 ? 1 : 0][r],
                                                     dw[ti][NTJ > 2 ? 2 : 0][r], dw[ti][NTJ > 3 ? 3 : 0][r]));
        }
    }
  }
}

// ==========================================================================================
// readout (mean-pool) with the last layer's BatchNorm+ReLU+dropout fused in, and its backward
// ==========================================================================================
constexpr int PTHR = 256;   // 16 row-lanes x 16 chunks

__global__ void __launch_bounds__(PTHR) k_pool_fwd(const float* __restrict__ Y,
                                                   const float* __restrict__ bn, DropCfg drop,
                                                   int use_drop, uint8_t* __restrict__ mask_out,
                                                   const int32_t* __restrict__ gptr, int B,
                                                   float* __restrict__ P) {
  __shared__ float red[16 * HID];
  const int j = threadIdx.x & 15, rr = threadIdx.x >> 4;
  const float4 a = ld4(bn + 4 * j), b = ld4(bn + HID + 4 * j);
  for (int g = blockIdx.x; g < B; g += gridDim.x) {
    const int rbeg = gptr[g], rend = gptr[g + 1];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int row = rbeg + rr; row < rend; row += 16) {
      const float4 y = ld4(Y + (int64_t)row * HID + 4 * j);
      uint32_t keep = 0xFu;
      if (use_drop) {
        keep = drop_bits(drop, (uint32_t)row, (uint32_t)j);
        if (mask_out) mask_out[(int64_t)row * 16 + j] = (uint8_t)keep;
      }
      float4 f;
      const float4 x = act4(y, a, b, keep, drop.scale, f);
      s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
    }
    st4(red + rr * HID + 4 * j, s);
    __syncthreads();
    if (threadIdx.x < HID) {
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += red[k * HID + threadIdx.x];
      P[(int64_t)g * HID + threadIdx.x] = tot / ((float)(rend - rbeg) + 1e-8f);
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(PTHR) k_pool_bwd(const float* __restrict__ dP,
                                                   const float* __restrict__ Y,
                                                   const float* __restrict__ bn, DropCfg drop,
                                                   int use_drop, const uint8_t* __restrict__ mask,
                                                   const int32_t* __restrict__ gptr, int B,
                                                   float* __restrict__ dZ,
                                                   double* __restrict__ s_slab) {
  __shared__ double red[16 * 128];
  const int j = threadIdx.x & 15, rr = threadIdx.x >> 4;
  const float4 a = ld4(bn + 4 * j), b = ld4(bn + HID + 4 * j);
  const float4 mean = ld4(bn + 2 * HID + 4 * j), is = ld4(bn + 3 * HID + 4 * j);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int g = blockIdx.x; g < B; g += gridDim.x) {
    const int rbeg = gptr[g], rend = gptr[g + 1];
    const float inv = 1.0f / ((float)(rend - rbeg) + 1e-8f);
    float4 gp = ld4(dP + (int64_t)g * HID + 4 * j);
    gp.x *= inv; gp.y *= inv; gp.z *= inv; gp.w *= inv;
    for (int row = rbeg + rr; row < rend; row += 16) {
      const float4 y = ld4(Y + (int64_t)row * HID + 4 * j);
      uint32_t keep = 0xFu;
      if (use_drop) keep = mask[(int64_t)row * 16 + j];
      float4 f;
      act4(y, a, b, keep, drop.scale, f);
      const float4 dz = make_float4(gp.x * f.x, gp.y * f.y, gp.z * f.z, gp.w * f.w);
      st4(dZ + (int64_t)row * HID + 4 * j, dz);
      s1[0] += dz.x; s1[1] += dz.y; s1[2] += dz.z; s1[3] += dz.w;
      s2[0] += (double)dz.x * ((y.x - mean.x) * is.x); s2[1] += (double)dz.y * ((y.y - mean.y) * is.y);
      s2[2] += (double)dz.z * ((y.z - mean.z) * is.z); s2[3] += (double)dz.w * ((y.w - mean.w) * is.w);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    red[rr * 128 + 4 * j + i] = s1[i];
    red[rr * 128 + 64 + 4 * j + i] = s2[i];
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[k * 128 + threadIdx.x];
    s_slab[(int64_t)blockIdx.x * 128 + threadIdx.x] = tot;
  }
}

// ==========================================================================================
// small reductions / BatchNorm coefficient kernels
// ==========================================================================================
template <typename T>
__global__ void __launch_bounds__(256) k_slab_reduce(const T* __restrict__ slab, int rows,
                                                     int width, double* __restrict__ out_d,
                                                     float* __restrict__ out_f, int out_cols,
                                                     int take_cols, int ld_out) {
  // one block per 4 output elements: 64 threads (one wave) per element, fixed-order tree
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  if (e < width)
    for (int r = lane; r < rows; r += 64) s += (double)slab[(int64_t)r * width + e];
  s = cgnn_wave_sum(s);
  if (e < width && lane == 0) {
    if (out_d) out_d[e] = s;
    if (out_f) {
      const int rr = e / out_cols, cc = e % out_cols;
      if (cc < take_cols) out_f[(int64_t)rr * ld_out + cc] = (float)s;
    }
  }
}

__global__ void k_bn_finalize(const double* __restrict__ sums, double count,
                              const float* __restrict__ gamma, const float* __restrict__ beta,
                              float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                              float eps, int training, float* __restrict__ bn_out) {
  const int c = threadIdx.x;
  if (c >= HID) return;
  float mean, var;
  if (training) {
    const double m = sums[c] / count;
    double v = sums[HID + c] / count - m * m;            // biased batch variance
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
    rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unbiased;
  } else {
    mean = rmean[c];
    var = rvar[c];
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  const float a = gamma[c] * invstd;
  bn_out[c] = a;
  bn_out[HID + c] = beta[c] - mean * a;
  bn_out[2 * HID + c] = mean;
  bn_out[3 * HID + c] = invstd;
}

__global__ void k_bn_bwd_finalize(const double* __restrict__ sums, double count,
                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                  float* __restrict__ bwc) {
  const int c = threadIdx.x;
  if (c >= HID) return;
  dbeta[c] = (float)sums[c];
  dgamma[c] = (float)sums[HID + c];
  bwc[c] = (float)(sums[c] / count);
  bwc[HID + c] = (float)(sums[HID + c] / count);
}

int g_grid_cache = 0;

int fused_grid() {
  if (g_grid_cache == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      return 256;
    g_grid_cache = cus;
  }
  return g_grid_cache;
}

DropCfg make_drop(float p, uint64_t seed, int* use_drop) {
  DropCfg d;
  *use_drop = (p > 0.f) ? 1 : 0;
  double thr = (double)p * 65536.0 + 0.5;
  if (thr > 65535.0) thr = 65535.0;
  d.thr16 = (uint32_t)thr;
  d.scale = p > 0.f ? (float)(1.0 / (1.0 - (double)d.thr16 / 65536.0)) : 1.0f;
  d.key0 = (uint32_t)(seed & 0xFFFFFFFFu) * 0x9E3779B9u + 0x85EBCA6Bu;
  d.key1 = (uint32_t)(seed >> 32) ^ 0xC2B2AE35u;
  return d;
}

bool tiles_ok(const cgnn_tiles* t) {
  return t && t->num_tiles >= 0 && t->num_nodes >= 0 && t->max_tile_rows <= CGNN_FUSED_MAX_ROWS &&
         (t->num_tiles == 0 || (t->tile_ptr && t->rowptr_dst && t->rowptr_src && t->selfc));
}

}  // namespace

extern "C" {

int cgnn_fused_grid(void) { return fused_grid(); }

int cgnn_gcn_fused_fwd_first(const cgnn_tiles* t, const float* X0, int32_t F0, const float* W0,
                             const float* bias, float* Y, double* stat_slab, void* stream) {
  if (!tiles_ok(t) || F0 <= 0 || F0 > CGNN_FUSED_MAX_F0) return t && t->max_tile_rows > CGNN_FUSED_MAX_ROWS ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (!X0 || !W0 || !bias || !Y) return CGNN_EINVAL;
  DropCfg d{};
  k_gcn_fwd<CGNN_FUSED_MAX_ROWS, true><<<fused_grid(), NTHR, 0, cgnn_stream(stream)>>>(
      *t, X0, F0, nullptr, d, 0, nullptr, W0, bias, Y, stat_slab);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_fwd(const cgnn_tiles* t, const float* Yprev, const float* bn_prev, float p_drop,
                       uint64_t seed, uint8_t* mask_out, const float* W, const float* bias,
                       float* Y, double* stat_slab, void* stream) {
  if (!tiles_ok(t)) return t && t->max_tile_rows > CGNN_FUSED_MAX_ROWS ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (!Yprev || !bn_prev || !W || !bias || !Y || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  int use_drop;
  DropCfg d = make_drop(p_drop, seed, &use_drop);
  k_gcn_fwd<CGNN_FUSED_MAX_ROWS, false><<<fused_grid(), NTHR, 0, cgnn_stream(stream)>>>(
      *t, Yprev, 0, bn_prev, d, use_drop, mask_out, W, bias, Y, stat_slab);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_bn_reduce(const double* slab, int32_t rows, int32_t width, double* sums, void* stream) {
  if (!slab || !sums || rows <= 0 || width <= 0) return CGNN_EINVAL;
  k_slab_reduce<double><<<(width + 3) / 4, 256, 0, cgnn_stream(stream)>>>(slab, rows, width, sums,
                                                                        nullptr, 1, 1, 1);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_bn_finalize(const double* sums, double count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps,
                     int32_t training, float* bn_out, void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || !bn_out) return CGNN_EINVAL;
  if (training && (!sums || count <= 0.0)) return CGNN_EINVAL;
  k_bn_finalize<<<1, 64, 0, cgnn_stream(stream)>>>(sums, count, gamma, beta, running_mean,
                                                   running_var, momentum, eps, training, bn_out);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_pool_fwd(const float* Y, const float* bn, float p_drop, uint64_t seed,
                            uint8_t* mask_out, const int32_t* gptr, int32_t num_graphs, float* P,
                            void* stream) {
  if (num_graphs < 0 || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (num_graphs == 0) return CGNN_OK;
  if (!Y || !bn || !gptr || !P) return CGNN_EINVAL;
  int use_drop;
  DropCfg d = make_drop(p_drop, seed, &use_drop);
  const int grid = num_graphs < 8 * fused_grid() ? num_graphs : 8 * fused_grid();
  k_pool_fwd<<<grid, PTHR, 0, cgnn_stream(stream)>>>(Y, bn, d, use_drop, mask_out, gptr, num_graphs, P);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_pool_bwd(const float* dP, const float* Y, const float* bn, float p_drop,
                            const uint8_t* mask, const int32_t* gptr, int32_t num_graphs,
                            float* dZ, double* s_slab, void* stream) {
  if (num_graphs < 0 || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (!dP || !Y || !bn || !gptr || !dZ || !s_slab) return CGNN_EINVAL;
  if (p_drop > 0.f && !mask) return CGNN_EINVAL;
  int use_drop;
  DropCfg d = make_drop(p_drop, 0, &use_drop);
  // exactly cgnn_fused_grid() workgroups so that the slab has the documented row count
  k_pool_bwd<<<fused_grid(), PTHR, 0, cgnn_stream(stream)>>>(dP, Y, bn, d, use_drop, mask, gptr,
                                                             num_graphs, dZ, s_slab);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_bn_bwd_finalize(const double* sums, double count, float* dgamma, float* dbeta, float* bwc,
                         void* stream) {
  if (!sums || !dgamma || !dbeta || !bwc || count <= 0.0) return CGNN_EINVAL;
  k_bn_bwd_finalize<<<1, 64, 0, cgnn_stream(stream)>>>(sums, count, dgamma, dbeta, bwc);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_bwd(const cgnn_tiles* t, const float* dZ, const float* Y, const float* bn,
                       const float* bwc, const float* Yprev, const float* bn_prev, float p_drop,
                       const uint8_t* mask_prev, const float* W, float* dZprev,
                       double* s_slab_prev, float* dW_slab, double* db_slab, void* stream) {
  if (!tiles_ok(t)) return t && t->max_tile_rows > CGNN_FUSED_MAX_ROWS ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (!dZ || !Y || !bn || !bwc || !Yprev || !bn_prev || !W || !dZprev || !s_slab_prev || !dW_slab ||
      !db_slab || p_drop < 0.f || p_drop >= 1.f)
    return CGNN_EINVAL;
  if (p_drop > 0.f && !mask_prev) return CGNN_EINVAL;
  int use_drop;
  DropCfg d = make_drop(p_drop, 0, &use_drop);
  k_gcn_bwd<CGNN_FUSED_MAX_ROWS, false><<<fused_grid(), NTHR, 0, cgnn_stream(stream)>>>(
      *t, dZ, Y, bn, bwc, Yprev, 0, bn_prev, d, use_drop, mask_prev, W, dZprev, s_slab_prev,
      dW_slab, db_slab);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_bwd_first(const cgnn_tiles* t, const float* dZ, const float* Y, const float* bn,
                             const float* bwc, const float* X0, int32_t F0, float* dW_slab,
                             double* db_slab, void* stream) {
  if (!tiles_ok(t) || F0 <= 0 || F0 > CGNN_FUSED_MAX_F0) return t && t->max_tile_rows > CGNN_FUSED_MAX_ROWS ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (!dZ || !Y || !bn || !bwc || !X0 || !dW_slab || !db_slab) return CGNN_EINVAL;
  DropCfg d{};
  k_gcn_bwd<CGNN_FUSED_MAX_ROWS, true><<<fused_grid(), NTHR, 0, cgnn_stream(stream)>>>(
      *t, dZ, Y, bn, bwc, X0, F0, nullptr, d, 0, nullptr, nullptr, nullptr, nullptr, dW_slab,
      db_slab);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_slab_reduce_f32(const float* slab, int32_t rows, int32_t out_rows, int32_t out_cols,
                         int32_t take_cols, float* out, int32_t ld_out, void* stream) {
  if (!slab || !out || rows <= 0 || out_rows <= 0 || out_cols <= 0 || take_cols <= 0 ||
      take_cols > out_cols || ld_out < take_cols)
    return CGNN_EINVAL;
  const int width = out_rows * out_cols;
  k_slab_reduce<float><<<(width + 3) / 4, 256, 0, cgnn_stream(stream)>>>(
      slab, rows, width, nullptr, out, out_cols, take_cols, ld_out);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_slab_reduce_f64(const double* slab, int32_t rows, int32_t width, float* out, void* stream) {
  if (!slab || !out || rows <= 0 || width <= 0) return CGNN_EINVAL;
  k_slab_reduce<double><<<(width + 3) / 4, 256, 0, cgnn_stream(stream)>>>(
      slab, rows, width, nullptr, out, width, width, width);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
