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
//   phase B  each wave owns 16-row blocks.  Lane group q (16 lanes x float4 = one 64-wide row)
//            owns rows 4q..4q+3 of the block.  The block's blocked-ELL entries (byte offset of
//            the neighbour's row in the tile, raw edge weight; self-loop last; zero-weight
//            padding up to the block's width) are fetched one block ahead with one coalesced
//            32-byte load per lane, kept in registers and broadcast inside the 16-lane group
//            with DPP row_newbcast; neighbour rows come straight out of the LDS tile with
//            ds_read_b128.  The finished block goes through `stg` to v_mfma_f32_16x16x4_f32
//            (exact fp32) and the epilogue (bias / BatchNorm statistics / ReLU' * dropout')
//            runs on the accumulators.  The symmetric normalisation is applied as
//            dis[d] * sum_e w_e * (dis[s_e] * x[s_e]): rows are scaled by dis when staged and
//            again when they leave the aggregation, so the metadata is static per batch.
//
// The MFMA reduction index is permuted freely (lane group kk supplies k = 16*kk + s) so that
// every operand fragment is a run of 16-byte LDS/register accesses; output tile tj holds the
// columns {4*c + tj}, so each lane ends up with 4 consecutive columns of 4 rows = float4 stores.
//
// No atomics anywhere: per-workgroup partial sums go to slabs reduced in a fixed order.
#include "common.h"
#include "agg_block.h"
#include "split_bf16.h"
#include "bn_tail.h"
#include "l0src.h"

// streamed operands / results of the tile kernels (each read or written once per launch)
// (measured: the backward kernels gain ~8 us each; the forward kernel's output is the next
// kernel's input and is better left to the default policy, its input likewise)
#define PF_LD ld4
#define PF_ST st4
#define BW_LD ldnt4
#define BW_ST stnt4

// Diagnostic build only (-DCGNN_STAMPS, tools/stamp_probe.py): per-phase s_memtime shares.
#ifdef CGNN_STAMPS
__device__ unsigned long long g_stamps[1024 * 8 * 8];   // [wg][wave][slot]
#define CGNN_STAMP_DECL unsigned long long stamp_t_ = 0; (void)stamp_t_;
#define CGNN_STAMP_BEGIN() stamp_t_ = __builtin_amdgcn_s_memtime();
#define CGNN_STAMP(slot)                                                               \
  {                                                                                    \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
    if ((threadIdx.x & 63) == 0)                                                       \
      g_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (slot)] += now_ - stamp_t_; \
    stamp_t_ = now_;                                                                   \
  }
#else
#define CGNN_STAMP_DECL
#define CGNN_STAMP_BEGIN()
#define CGNN_STAMP(slot)
#endif

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
  const uint32_t* dev_key;   // optional device word XOR-ed into key1 (graph replay: a captured
                             // kernel advances it, so replays draw fresh masks)
};

__device__ __forceinline__ DropCfg drop_resolve(DropCfg d) {
  if (d.dev_key) d.key1 ^= d.dev_key[0];
  return d;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// 4 keep-bits for the float4 at (global row, chunk); bit i <-> column 4*chunk + i.  One keyed
// counter hash gives the first 32 random bits, a second round of the same mixer the other 32
// (chained, not a second counter: half the multiplies of two independent hashes).
__device__ __forceinline__ uint32_t drop_bits(const DropCfg& d, uint32_t row, uint32_t chunk) {
  const uint32_t e = row * 16u + chunk;
  const uint32_t h0 = mix32((e ^ d.key0) + d.key1);
  const uint32_t h1 = mix32(h0 + 0x9E3779B9u);
  uint32_t b = 0;
  b |= ((h0 & 0xFFFFu) >= d.thr16) ? 1u : 0u;
  b |= ((h0 >> 16) >= d.thr16) ? 2u : 0u;
  b |= ((h1 & 0xFFFFu) >= d.thr16) ? 4u : 0u;
  b |= ((h1 >> 16) >= d.thr16) ? 8u : 0u;
  return b;
}


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


// Reduce per-lane fp64 column partials (lane (q,j): columns 4j..4j+3) over the workgroup and
// write slab_row[0..63] (= s1) and slab_row[64..127] (= s2).  `red` >= 8*128 doubles of LDS.
template <typename T, int NW = 8>
__device__ __forceinline__ void reduce_stats(T (&s1)[4], T (&s2)[4], double* red,
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
      red[wave * 128 + 4 * j + i] = (double)s1[i];
      red[wave * 128 + 64 + 4 * j + i] = (double)s2[i];
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w * 128 + threadIdx.x];
    slab_row[threadIdx.x] = t;
  }
  __syncthreads();
}

// rows-in-flight policy of the aggregation (see agg_block): tunables for A/B runs
#define CGNN_FWD_G 4
#define CGNN_BWD_G 2
constexpr int FWD_G = CGNN_FWD_G;
constexpr int BWD_G = CGNN_BWD_G;

// ==========================================================================================
// forward
// ==========================================================================================
template <int MAXR, bool FIRST>
__global__ void __launch_bounds__(NTHR) k_gcn_fwd(
    cgnn_tiles t, const float* __restrict__ Xin, int F0, const float* __restrict__ bn_prev,
    DropCfg drop_in, int use_drop, uint8_t* __restrict__ mask_out, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ Y, double* __restrict__ stat_slab) {
  const DropCfg drop = drop_resolve(drop_in);
  __shared__ __attribute__((aligned(16))) float tile[MAXR * HID];
  __shared__ __attribute__((aligned(16))) float stg_all[NWAVE * STG_FLOATS];
  __shared__ float disl[MAXR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
  float* stg = stg_all + wave * STG_FLOATS;
  const uint4* ent = static_cast<const uint4*>(t.ent_dst);

  // B operand of the projection: B[k][col] = W[col][k], lane (kk=q, jj=j), tile tj <-> col 4j+tj.
  //   first  : k = 4s + q < F0 (F0 <= 16 -> 4 k-steps), 16 registers
  //   generic: k = 16q + s; W^T lives in LDS (one conflict-free ds_read_b128 per k-step) so that
  //            the registers can hold the NEXT tile's rows while this tile is being processed
  __shared__ __attribute__((aligned(16))) float Wt[FIRST ? 4 : HID * HID];
  float wreg[4][4];
  if (FIRST) {
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 4 * s + q;
        wreg[tj][s] = k < F0 ? W[(4 * j + tj) * F0 + k] : 0.f;
      }
  } else {
    for (int i = threadIdx.x; i < HID * HID; i += NTHR) Wt[(i & 63) * HID + (i >> 6)] = W[i];
  }
  const float4 bias4 = ld4(bias + 4 * j);
  float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
  if (!FIRST) {
    pa = ld4(bn_prev + 4 * j);
    pb = ld4(bn_prev + HID + 4 * j);
  }
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  CGNN_STAMP_DECL
  // cross-tile prefetch registers: row (tid>>4) + 32u, columns 4j..4j+3 of the next tile
  constexpr int PF = FIRST ? 1 : MAXR / 32;
  float4 pfy[PF];
  float pfd[PF];
  if (!FIRST && (int)blockIdx.x < t.num_tiles) {
    const int nb2 = t.tile_ptr[blockIdx.x], nn2 = t.tile_ptr[blockIdx.x + 1] - nb2;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int row = (threadIdx.x >> 4) + 32 * u;
      pfy[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      pfd[u] = 0.f;
      if (row < nn2) {
        pfy[u] = ld4(Xin + (int64_t)(nb2 + row) * HID + 4 * j);
        pfd[u] = t.dis[nb2 + row];
      }
    }
  }

  for (int tid = blockIdx.x; tid < t.num_tiles; tid += gridDim.x) {
    CGNN_STAMP_BEGIN()
    const int base = t.tile_ptr[tid];
    const int n = t.tile_ptr[tid + 1] - base;
    const int nblk = (n + 15) >> 4;
    const int gb0 = t.tile_blk[tid];
    // metadata of this wave's first block: in flight during phase A
    // entry offsets / widths of all of this wave's blocks (<= 3): scalar loads, once per tile
    int boff[3] = {0, 0, 0}, bwid[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int bb = cgnn_uniform(wave) + NWAVE * k;
      if (bb < nblk) {
        boff[k] = t.blk_off_dst[gb0 + bb];
        bwid[k] = (t.blk_off_dst[gb0 + bb + 1] - boff[k]) >> 4;
      }
    }
    int off0 = boff[0], width = bwid[0], bk = 0;
    MetaRegs pre;
    if (wave < nblk) pre = meta_issue<true>(ent + (off0 >> 1), width, q, j);

    // ---------------------------------------------------------------- phase A: fill the tile
    // (rows are pre-scaled by dis[row]: A_hat X = dis * (A_w + I)(dis * X))
    if (FIRST) {
      // T = X0 W0^T on the matrix core, written straight into the tile (all loads first).
      for (int r = threadIdx.x; r < nblk * 16; r += NTHR) disl[r] = r < n ? t.dis[base + r] : 0.f;
      for (int b = wave; b < nblk; b += NWAVE) {
        const int arow = 16 * b + j;                    // A operand: row i = j, k-slot kk = q
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        float av[4], dv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int k = 4 * s + q;
          av[s] = (arow < n && k < F0) ? Xin[(int64_t)(base + arow) * F0 + k] : 0.f;
          const int row = 16 * b + 4 * q + s;
          dv[s] = row < n ? t.dis[base + row] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (4 * s < F0) {
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
              acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], wreg[tj][s], acc[tj], 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          st4(tile + (16 * b + 4 * q + r) * HID + 4 * j,
              make_float4(acc[0][r] * dv[r], acc[1][r] * dv[r], acc[2][r] * dv[r], acc[3][r] * dv[r]));
      }
    } else {
      // the tile's rows were requested a whole phase B ago (cross-tile register prefetch)
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int row = (threadIdx.x >> 4) + 32 * u;
        if (row < nblk * 16) {
          float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
          if (row < n) {
            uint32_t keep = 0xFu;
            if (use_drop) {
              keep = drop_bits(drop, (uint32_t)(base + row), (uint32_t)j);
              if (mask_out) mask_out[(int64_t)(base + row) * 16 + j] = (uint8_t)keep;
            }
            float4 f;
            x = scale4(act4(pfy[u], pa, pb, keep, drop.scale, f), pfd[u]);
          }
          st4(tile + row * HID + 4 * j, x);
          if (j == 0) disl[row] = pfd[u];
        }
      }
    }
    __syncthreads();
    if (!FIRST) {
      // request the NEXT tile's rows now; they land while this tile is aggregated/projected
      const int nxt = tid + gridDim.x;
      if (nxt < t.num_tiles) {
        const int nb2 = t.tile_ptr[nxt], nn2 = t.tile_ptr[nxt + 1] - nb2;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int row = (threadIdx.x >> 4) + 32 * u;
          pfy[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          pfd[u] = 0.f;
          if (row < nn2) {
            pfy[u] = ld4(Xin + (int64_t)(nb2 + row) * HID + 4 * j);
            pfd[u] = t.dis[nb2 + row];
          }
        }
      }
    }
    CGNN_STAMP(1)      // phase A + barrier

    // ------------------------------------------------- phase B: aggregate (+ project) blocks
    for (int b = wave; b < nblk; b += NWAVE) {
      f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      ++bk;
      const int off1 = bk == 1 ? boff[1] : boff[2], width1 = bk == 1 ? bwid[1] : bwid[2];
      CGNN_STAMP(2)
      float4 ag[4];
      agg_block<FWD_G, true>(tile, pre, ent + (off0 >> 1), width, q, j, ag);
      CGNN_STAMP(3)    // aggregation
      if (b + NWAVE < nblk) pre = meta_issue<true>(ent + (off1 >> 1), width1, q, j);
      off0 = off1; width = width1;
      if (FIRST) {
        // tile already holds dis*T: Y = dis * (A_w + I)(dis*T) + b, straight from registers.
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = 16 * b + 4 * q + it;
          if (row < n) {
            const float4 a = scale4(ag[it], disl[row]);
            const float4 y = make_float4(a.x + bias4.x, a.y + bias4.y, a.z + bias4.z, a.w + bias4.w);
            st4(Y + (int64_t)(base + row) * HID + 4 * j, y);
            s1[0] += y.x; s1[1] += y.y; s1[2] += y.z; s1[3] += y.w;
            s2[0] += (double)y.x * y.x; s2[1] += (double)y.y * y.y;
            s2[2] += (double)y.z * y.z; s2[3] += (double)y.w * y.w;
          }
        }
        CGNN_STAMP(4)  // epilogue (first layer)
        continue;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it)
        st4(stg + (4 * q + it) * SLD + 4 * j, scale4(ag[it], disl[16 * b + 4 * q + it]));
      __builtin_amdgcn_wave_barrier();
      // A fragments: lane (i=j, kk=q) holds P[row j][k = 16q + s], s = 0..15
      float af[16];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = ld4(stg + j * SLD + 16 * q + 4 * u);
        af[4 * u + 0] = v.x; af[4 * u + 1] = v.y; af[4 * u + 2] = v.z; af[4 * u + 3] = v.w;
      }
      CGNN_STAMP(4)    // staging write + A-fragment read (+ next metadata issue)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float4 w4 = ld4(Wt + (16 * q + s) * HID + 4 * j);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], w4.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], w4.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], w4.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], w4.w, acc[3], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
#ifdef CGNN_STAMPS
      asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]));
#endif
      CGNN_STAMP(5)    // 64 MFMAs
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
      CGNN_STAMP(7)    // epilogue: bias, store, fp64 statistics
    }
    __syncthreads();
    CGNN_STAMP(6)      // end-of-tile barrier wait
  }
  if (stat_slab)
    reduce_stats(s1, s2, reinterpret_cast<double*>(tile), stat_slab + (int64_t)blockIdx.x * 128);
}

// ------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix pipe, exactly split ("bf16x3").
// An fp32 value is cut into three bf16 pieces by TRUNCATION: h = top 8 significant bits,
// m = the next 8, l = the last 8, so x == h + m + l exactly and every piece is exactly a bf16.
// A product x*w is then the six partial products of total order <= 2
//     h*h' + (h*m' + m*h') + (h*l' + m*m' + l*h')
// each exact in fp32 (8 x 8 significant bits) and accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16; the three dropped terms are <= 2^-24 of the product, i.e. at the
// rounding level of an fp32 multiply-add.  Six bf16 MFMAs of K = 32 replace sixteen fp32 MFMAs
// of K = 4 for the same 16x16x64 product: 6 x 2 x 16 = 192 instead of 16 x 32 = 512 matrix-pipe
// cycles per output tile (the fp32 pipe is 1/16 of the bf16 rate on gfx950).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma_bf16(const uint4& a, const uint4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                 c, 0, 0, 0);
}
// acc += A * B for split operands, smallest terms first
__device__ __forceinline__ f32x4 mfma_split(const Split8& a, const uint4& bh, const uint4& bm,
                                            const uint4& bl, f32x4 acc) {
  acc = mfma_bf16(a.l, bh, acc);
  acc = mfma_bf16(a.h, bl, acc);
  acc = mfma_bf16(a.m, bm, acc);
  acc = mfma_bf16(a.m, bh, acc);
  acc = mfma_bf16(a.h, bm, acc);
  acc = mfma_bf16(a.h, bh, acc);
  return acc;
}

// Split weight panel in LDS for  out[row][4j+tj] = sum_k x[row][k] * B[k][4j+tj]  (16x16x32 tiles):
// fragment (term, mstep, tj) of lane (q, j) holds k = 16*(2*mstep + (e>>2)) + 4q + (e&3), e = 0..7
// -- the k order in which the A side keeps a row (two float4 chunks per MFMA).  Layout
// wsp[((term*2 + mstep)*4 + tj)*64 + lane], 16 bytes each: every B read is one conflict-free
// ds_read_b128.  24 KB.
constexpr int WSP_FRAGS = 3 * 2 * 4 * 64;
template <bool TRANSPOSED>     // false: B[k][n] = W[n][k] (X W^T);  true: B[k][n] = W[k][n] (dT W)
__device__ __forceinline__ void stage_split_weight(uint4* wsp, const float* __restrict__ W, int nthreads) {
  for (int idx = threadIdx.x; idx < 2 * 4 * 64; idx += nthreads) {
    const int ln = idx & 63, tj = (idx >> 6) & 3, ms = idx >> 8;
    const int qq = ln >> 4, jj = ln & 15;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * (2 * ms + (e >> 2)) + 4 * qq + (e & 3), n = 4 * jj + tj;
      v[e] = TRANSPOSED ? W[k * HID + n] : W[n * HID + k];
    }
    const Split8 s = split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
    wsp[((0 * 2 + ms) * 4 + tj) * 64 + ln] = s.h;
    wsp[((1 * 2 + ms) * 4 + tj) * 64 + ln] = s.m;
    wsp[((2 * 2 + ms) * 4 + tj) * 64 + ln] = s.l;
  }
}

// ==========================================================================================
// forward, projection first (layers l > 0)
// ==========================================================================================
// Y = A_hat (X W^T) + b  -- the reference's own order (models.py:111-114) -- with
// X = drop(relu(BatchNorm(Y_prev))) rebuilt on the fly.  Twelve waves (3 per SIMD) share one tile:
//
//   phase A  each wave takes 16-row blocks of Y_prev straight into MATRIX-CORE operand layout
//            (lane (q, j): row j of the block, columns 16c + 4q .. +3 for c = 0..3, i.e. the
//            reduction index is permuted so that every operand is a 16-byte access; the rows
//            were requested a whole phase B earlier), applies BatchNorm + ReLU + dropout and the
//            dis[row] pre-scaling in registers, multiplies by W^T (LDS) with 64
//            v_mfma_f32_16x16x4_f32 and stores dis * T into the LDS tile.  No staging buffer,
//            no wave barriers: the only LDS writes are the T rows themselves.
//   phase B  blocked-ELL aggregation out of the tile (agg_block.h); the finished rows leave
//            straight from the aggregation registers: * dis[row], + bias, 16-byte stores, fp64
//            BatchNorm statistics.
//
// Compared with aggregate-then-project this drops the per-block transposition through LDS and
// the register footprint of the projection from phase B, which is what lets a third wave per
// SIMD fit (168 VGPRs): the phases are bound by LDS/issue latency, not by any pipe's peak rate.
constexpr int PF_NW = 12;
constexpr int PF_NTHR = PF_NW * 64;          // 768
#define CGNN_PF_G 2

// FROM_P0: the previous layer is layer 0 in factored form (cgnn_l0src): a row of Y0 is rebuilt
// from its 32-byte narrow aggregate instead of being read (256 bytes) from HBM.
template <int MAXR, bool FROM_P0>
__global__ void __launch_bounds__(PF_NTHR) k_gcn_fwd_pf(
    cgnn_tiles t, const float* __restrict__ Xin, cgnn_l0src l0, const float* __restrict__ bn_prev, DropCfg drop_in,
    int use_drop, uint8_t* __restrict__ mask_out, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ Y, double* __restrict__ stat_slab, cgnn_bn_tail tail) {
  const DropCfg drop = drop_resolve(drop_in);
  __shared__ __attribute__((aligned(16))) float tile[MAXR * HID];
  __shared__ uint4 wsp[WSP_FRAGS];                               // split W^T panel, 24 KB
  __shared__ __attribute__((aligned(16))) float bnab[2 * HID];
  __shared__ float disl[MAXR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
  const uint4* ent = static_cast<const uint4*>(t.ent_dst);
  constexpr int BPW = (MAXR / 16 + PF_NW - 1) / PF_NW;        // blocks per wave and tile (2)

  // FROM_P0: rows of Y0 = P0 W0^T + b0 are rebuilt on the fp32 matrix pipe (l0src.h): D[col][row]
  // puts columns 16c + 4q .. +3 of row j in lane (q, j), the layout phase A works in
  __shared__ __attribute__((aligned(16))) float b0l[FROM_P0 ? HID : 4];
  stage_split_weight<false>(wsp, W, PF_NTHR);
  for (int i = threadIdx.x; i < 2 * HID; i += PF_NTHR) bnab[i] = bn_prev[i];
  L0W w0;
  if (FROM_P0) {
    w0 = l0w_cols(l0, lane);
    for (int i = threadIdx.x; i < HID; i += PF_NTHR) b0l[i] = l0.b0[i];
  }
  const float4 bias4 = ld4(bias + 4 * j);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};

  // operand rows of the NEXT tile: block u of this wave, row j, columns 16c + 4q .. +3
  // (FROM_P0: elements q and 4 + q of the row's narrow aggregate, in px[u][0].x / .y)
  constexpr int NPX = FROM_P0 ? 1 : 4;
  float4 px[BPW][NPX];
  float pd[BPW];
  // block u of tile `tid`: issued right after the previous tile's block u has been consumed, so
  // the loads are in flight through the rest of phase A AND all of phase B (HBM never idles)
  auto request_block = [&](int tid, int u) {
    const int nb2 = t.tile_ptr[tid], nn2 = t.tile_ptr[tid + 1] - nb2;
    const int row = 16 * (wave + PF_NW * u) + j;
    pd[u] = 0.f;
#pragma unroll
    for (int c = 0; c < NPX; ++c) px[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nn2) {
      if (FROM_P0) {
        const float* src = l0.P0 + (int64_t)(nb2 + row) * L0_FP + q;
        px[u][0].x = src[0];
        px[u][0].y = src[4];
      } else {
        const float* src = Xin + (int64_t)(nb2 + row) * HID + 4 * q;
#pragma unroll
        for (int c = 0; c < NPX; ++c) px[u][c] = PF_LD(src + 16 * c);
      }
      pd[u] = t.dis[nb2 + row];
    }
  };
  auto request = [&](int tid) {
#pragma unroll
    for (int u = 0; u < BPW; ++u) request_block(tid, u);
  };
  if ((int)blockIdx.x < t.num_tiles) request(blockIdx.x);
  __syncthreads();                       // Wt / bnab visible

  for (int tid = blockIdx.x; tid < t.num_tiles; tid += gridDim.x) {
    const int base = t.tile_ptr[tid];
    const int n = t.tile_ptr[tid + 1] - base;
    const int nblk = (n + 15) >> 4;
    const int gb0 = t.tile_blk[tid];
    int boff[BPW], bwid[BPW];
#pragma unroll
    for (int k = 0; k < BPW; ++k) {
      boff[k] = bwid[k] = 0;
      const int bb = cgnn_uniform(wave) + PF_NW * k;
      if (bb < nblk) {
        boff[k] = t.blk_off_dst[gb0 + bb];
        bwid[k] = (t.blk_off_dst[gb0 + bb + 1] - boff[k]) >> 4;
      }
    }
    MetaRegs pre;
    if (wave < nblk) pre = meta_issue<true>(ent + (boff[0] >> 1), bwid[0], q, j);

    // ------------------------------------------------ phase A: transform + project into the tile
    const int nxt = tid + gridDim.x;
#pragma unroll
    for (int u = 0; u < BPW; ++u) {
      const int b = wave + PF_NW * u;
      if (b >= nblk) break;
      const int row = 16 * b + j;
      const bool live = row < n;                 // dead rows: px = 0 and pd = 0 -> x = 0
      f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        float4 xc[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int c = 2 * ms + h2;
          uint32_t keep = 0xFu;
          if (use_drop) {
            keep = drop_bits(drop, (uint32_t)(base + row), (uint32_t)(4 * c + q));
            if (mask_out && live) mask_out[(int64_t)(base + row) * 16 + 4 * c + q] = (uint8_t)keep;
          }
          float4 f;
          float4 yraw = px[u][c < NPX ? c : 0];
          if (FROM_P0) {
            const float4 bq = ld4(b0l + 16 * c + 4 * q);
            const l0_f32x4 y0 = l0_mfma(w0.a[c], w0.b[c], px[u][0].x, px[u][0].y, l0_f32x4{bq.x, bq.y, bq.z, bq.w});
            yraw = make_float4(y0[0], y0[1], y0[2], y0[3]);
          }
          xc[h2] = scale4(act4(yraw, ld4(bnab + 16 * c + 4 * q), ld4(bnab + HID + 16 * c + 4 * q),
                               keep, drop.scale, f), pd[u]);
        }
        // k-step ms of the 16x16x32 product: this lane's 8 reduction indices are the columns
        // 16*(2ms) + 4q .. +3 and 16*(2ms+1) + 4q .. +3 (the order of stage_split_weight)
        const Split8 a = split8(xc[0], xc[1]);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          const uint4 bh = wsp[((0 * 2 + ms) * 4 + tj) * 64 + lane];
          const uint4 bm = wsp[((1 * 2 + ms) * 4 + tj) * 64 + lane];
          const uint4 bl = wsp[((2 * 2 + ms) * 4 + tj) * 64 + lane];
          acc[tj] = mfma_split(a, bh, bm, bl, acc[tj]);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the LDS reads of the next k-step from being
                                                // hoisted above this one's (register pressure)
      }
      // accumulator tile tj holds columns 4j + tj of rows 4q + r: one float4 per row
#pragma unroll
      for (int r = 0; r < 4; ++r)
        st4(tile + (16 * b + 4 * q + r) * HID + 4 * j, make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]));
      if (q == 0) disl[row] = pd[u];
      if (nxt < t.num_tiles) request_block(nxt, u);
    }
    if (nxt < t.num_tiles) {
      // blocks this wave did not have in THIS tile (ragged tiles) but may have in the next one
#pragma unroll
      for (int u = 0; u < BPW; ++u)
        if (wave + PF_NW * u >= nblk) request_block(nxt, u);
    }
    __syncthreads();

    // ------------------------------------------------ phase B: aggregate, bias, store, statistics
#pragma unroll
    for (int u = 0; u < BPW; ++u) {
      const int b = wave + PF_NW * u;
      if (b >= nblk) break;
      float4 ag[4];
      agg_block<CGNN_PF_G, true>(tile, pre, ent + (boff[u] >> 1), bwid[u], q, j, ag);
      if (u + 1 < BPW && b + PF_NW < nblk) pre = meta_issue<true>(ent + (boff[u + 1 < BPW ? u + 1 : u] >> 1), bwid[u + 1 < BPW ? u + 1 : u], q, j);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = 16 * b + 4 * q + it;
        if (row < n) {
          const float4 a = scale4(ag[it], disl[row]);
          const float4 y = make_float4(a.x + bias4.x, a.y + bias4.y, a.z + bias4.z, a.w + bias4.w);
          PF_ST(Y + (int64_t)(base + row) * HID + 4 * j, y);
          s1[0] += y.x; s1[1] += y.y; s1[2] += y.z; s1[3] += y.w;
          s2[0] += (double)y.x * y.x; s2[1] += (double)y.y * y.y;
          s2[2] += (double)y.z * y.z; s2[3] += (double)y.w * y.w;
        }
      }
    }
    __syncthreads();
  }
  if (tail.acc) {
    // the layer's BatchNorm finalised by the workgroup that arrives last (bn_tail.h): no slab, no launch
    double* red = reinterpret_cast<double*>(tile);
    reduce_stats<double, PF_NW>(s1, s2, red, red + PF_NW * 128);
    bn_tail_run(tail, red + PF_NW * 128, reinterpret_cast<int*>(red + PF_NW * 128 + 128), [](int) { return 0.f; });
  } else if (stat_slab) {
    reduce_stats<double, PF_NW>(s1, s2, reinterpret_cast<double*>(tile), stat_slab + (int64_t)blockIdx.x * 128);
  }
}

// ==========================================================================================
// backward
// ==========================================================================================
// POOLIN: this is the last layer and its incoming gradient is the readout's: dZ is not read
// from HBM but rebuilt per row as dP[graph]/(n_g+1e-8) * relu' * dropout' (models.py:57-59,
// 209-211 backward) -- saves writing and re-reading one [Nn,64] array per step.
struct PoolIn {
  const float* dP;               // [B,64]
  const int32_t* node_graph;     // [Nn]
  const int32_t* gptr;           // [B+1]
  const uint8_t* mask_cur;       // keep bits of THIS layer's activation (or NULL)
};


// XP0: the previous layer's output is layer 0's in factored form: its rows are rebuilt from the
// narrow aggregate l0.P0 (32 bytes per row) instead of being read from Xprev (256 bytes per row).
template <int MAXR, bool FIRST, bool POOLIN, bool XP0 = false, int NW = NWAVE>
__global__ void __launch_bounds__(NW * 64) k_gcn_bwd(
    cgnn_tiles t, PoolIn pin, cgnn_l0src l0, const float* __restrict__ dZ, const float* __restrict__ Y,
    const float* __restrict__ bn, const float* __restrict__ bwc,
    const float* __restrict__ Xprev /* Yprev [Nn,64] or X0 [Nn,F0] */, int F0,
    const float* __restrict__ bn_prev, DropCfg drop, int use_drop,
    const uint8_t* __restrict__ mask_prev, const float* __restrict__ W,
    float* __restrict__ dZprev, double* __restrict__ s_slab, float* __restrict__ dW_slab,
    double* __restrict__ db_slab, cgnn_bn_tail tail) {
  __shared__ __attribute__((aligned(16))) float tile[MAXR * HID];
  __shared__ __attribute__((aligned(16))) float stg_all[NW * STG_FLOATS];
  __shared__ uint4 wsp[FIRST ? 1 : WSP_FRAGS];          // split W panel for dX = dT W (24 KB)
  __shared__ float disl[MAXR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
  float* stg = stg_all + wave * STG_FLOATS;
  const uint4* ent = static_cast<const uint4*>(t.ent_src);

  if (!FIRST) stage_split_weight<true>(wsp, W, (NW * 64));
  // previous layer's BatchNorm block (a | b | mean | invstd) lives in LDS, not in 16 registers
  __shared__ __attribute__((aligned(16))) float bnl[FIRST ? 4 : 4 * HID];
  if (!FIRST) {
    for (int i = threadIdx.x; i < 4 * HID; i += (NW * 64)) bnl[i] = bn_prev[i];
  }
  // XP0: rows of Y0 rebuilt on the fp32 matrix pipe (l0src.h): D_t[row][col 4j + t]
  L0W w0;
  float4 b0q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (XP0) {
    w0 = l0w_quad(l0, threadIdx.x & 63);
    b0q = ld4(l0.b0 + 4 * (threadIdx.x & 15));
  }
  // dW accumulators: FIRST: dw[ti][0] only (16 input columns); else dw[ti][tj].
  f32x4 dw[4][FIRST ? 1 : 4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b2 = 0; b2 < (FIRST ? 1 : 4); ++b2) dw[a][b2] = f32x4{0, 0, 0, 0};
  // per-thread partial sums stay fp32 (a thread adds a few hundred terms over its tiles: error
  // ~1e-6 relative, far inside the 1e-5 bar); everything across threads/workgroups is fp64.
  // (64 dW accumulators leave no room for 28 registers of fp64 partials.)
  float db[4] = {0, 0, 0, 0};
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  CGNN_STAMP_DECL

  for (int tid = blockIdx.x; tid < t.num_tiles; tid += gridDim.x) {
    CGNN_STAMP_BEGIN()
    const int base = t.tile_ptr[tid];
    const int n = t.tile_ptr[tid + 1] - base;
    const int nblk = (n + 15) >> 4;
    const int gb0 = t.tile_blk[tid];
    // entry offset / width of a block: scalar loads (wave-uniform index)
    const int uw = cgnn_uniform(wave);
    int off0 = 0, width = 0;
    if (uw < nblk) {
      off0 = t.blk_off_src[gb0 + uw];
      width = (t.blk_off_src[gb0 + uw + 1] - off0) >> 4;
    }
    MetaRegs pre;
    if (wave < nblk) pre = meta_issue<FIRST>(ent + (off0 >> 1), width, q, j);

    // --------------------------- phase A: dis * dY, dY = BatchNorm'(dZ), into the tile
    {
      // phase-A constants of this thread's 4 columns: re-read per tile (L1/L2 hits) instead of
      // held in 20 registers across phase B, where dW's 64 accumulators need the room.
      const float4 ca = ld4(bn + 4 * j), cmean = ld4(bn + 2 * HID + 4 * j), cis = ld4(bn + 3 * HID + 4 * j);
      const float4 c1 = ld4(bwc + 4 * j), c2 = ld4(bwc + HID + 4 * j);
      const float4 cb = POOLIN ? ld4(bn + HID + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
#define CGNN_BWD_UNR 6
#define CGNN_BWD_UNR_POOL 6
      // rows requested per thread before the first is consumed: every batch is one exposed HBM
      // round trip of this phase (the dW accumulators leave no room to prefetch across tiles)
      constexpr int UNR = POOLIN ? CGNN_BWD_UNR_POOL : CGNN_BWD_UNR;
      for (int r0 = threadIdx.x >> 4; r0 < nblk * 16; r0 += (NW * 4) * UNR) {
        float4 zb[UNR], yb[UNR];
        float dv[UNR];
        int gid[UNR];
        uint32_t kb[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int row = r0 + (NW * 4) * u;
          zb[u] = yb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          dv[u] = 0.f;
          gid[u] = 0;
          kb[u] = 0xFu;
          if (row < n) {
            if (POOLIN) {
              gid[u] = pin.node_graph[base + row];
              if (use_drop) kb[u] = pin.mask_cur[(int64_t)(base + row) * 16 + j];
            } else {
              zb[u] = BW_LD(dZ + (int64_t)(base + row) * HID + 4 * j);
            }
            yb[u] = BW_LD(Y + (int64_t)(base + row) * HID + 4 * j);
            dv[u] = t.dis[base + row];
          }
        }
        if (POOLIN) {
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int row = r0 + (NW * 4) * u;
            if (row < n) {
              const float inv = 1.0f / ((float)(pin.gptr[gid[u] + 1] - pin.gptr[gid[u]]) + 1e-8f);
              zb[u] = scale4(ld4(pin.dP + (int64_t)gid[u] * HID + 4 * j), inv);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int row = r0 + (NW * 4) * u;
          if (row < nblk * 16) {
            float4 dy = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < n) {
              float4 dz = zb[u];
              const float4 y = yb[u];
              if (POOLIN) {
                float4 f;
                act4(y, ca, cb, kb[u], drop.scale, f);
                dz = make_float4(dz.x * f.x, dz.y * f.y, dz.z * f.z, dz.w * f.w);
              }
              dy.x = ca.x * (dz.x - c1.x - (y.x - cmean.x) * cis.x * c2.x);
              dy.y = ca.y * (dz.y - c1.y - (y.y - cmean.y) * cis.y * c2.y);
              dy.z = ca.z * (dz.z - c1.z - (y.z - cmean.z) * cis.z * c2.z);
              dy.w = ca.w * (dz.w - c1.w - (y.w - cmean.w) * cis.w * c2.w);
              db[0] += dy.x; db[1] += dy.y; db[2] += dy.z; db[3] += dy.w;
              dy = scale4(dy, dv[u]);
            }
            st4(tile + row * HID + 4 * j, dy);
            if (j == 0) disl[row] = dv[u];
          }
        }
      }
    }
    CGNN_STAMP(0)
    __syncthreads();
    CGNN_STAMP(1)

    // --------------------- phase B: dT = A_hat^T dY per block; dW += dT^T X; dZprev = ...
    for (int b = wave; b < nblk; b += NW) {
      float4 yp[4];
      float p0a = 0.f, p0b = 0.f;               // XP0: P0[row 16b + j][q], [4 + q] (MFMA operand)
      uint32_t keeps = 0u;                      // byte r: keep bits of row 4q+r (0 = no such row)
      {
        int off1 = 0, width1 = 0;
        {
          const int bn2 = cgnn_uniform(b) + NW;
          if (bn2 < nblk) {
            off1 = t.blk_off_src[gb0 + bn2];
            width1 = (t.blk_off_src[gb0 + bn2 + 1] - off1) >> 4;
          }
        }
        CGNN_STAMP(2)
        // previous layer's block (rows 4q+r, columns 4j..4j+3): requested AFTER the metadata
        // commit (so the commit does not wait on it) and BEFORE the aggregation (which hides
        // its HBM latency).  Only the raw pre-BatchNorm values and the keep bits stay live; X
        // (B operand of dW) and relu'/dropout'/xhat (epilogue) are re-derived where needed.
        if (!FIRST) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * b + 4 * q + r;
            yp[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < n) {
                if (!XP0) yp[r] = BW_LD(Xprev + (int64_t)(base + row) * HID + 4 * j);
                uint32_t kb = 0xFu;
                if (use_drop) kb = mask_prev[(int64_t)(base + row) * 16 + j];
                keeps |= kb << (8 * r);
            }
          }
        }
        if (!FIRST && XP0 && 16 * b + j < n) {
          const float* src = l0.P0 + (int64_t)(base + 16 * b + j) * L0_FP + q;
          p0a = src[0];
          p0b = src[4];
        }
        float4 ag[4];
        agg_block<FIRST ? 4 : BWD_G, FIRST>(tile, pre, ent + (off0 >> 1), width, q, j, ag);
        CGNN_STAMP(3)
        if (XP0) {
          // rows outside the tile rebuild to b0, harmless: their keep byte is 0 -> x = f = 0
          l0_f32x4 yt[4];
#pragma unroll
          for (int tq = 0; tq < 4; ++tq) {
            const float bt = tq == 0 ? b0q.x : tq == 1 ? b0q.y : tq == 2 ? b0q.z : b0q.w;
            yt[tq] = l0_mfma(p0a, p0b, w0.a[tq], w0.b[tq], l0_f32x4{bt, bt, bt, bt});
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) yp[r] = make_float4(yt[0][r], yt[1][r], yt[2][r], yt[3][r]);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
          st4(stg + (4 * q + it) * SLD + 4 * j, scale4(ag[it], disl[16 * b + 4 * q + it]));
        if (b + NW < nblk) pre = meta_issue<FIRST>(ent + (off1 >> 1), width1, q, j);
        off0 = off1; width = width1;
      }
      __builtin_amdgcn_wave_barrier();

      if (FIRST) {
        // B operand: X0[row 4q+s][col j] (zero beyond F0); dW0[o][jcol] tile ti: o = 16ti+4q+r
        float xv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int row = 16 * b + 4 * q + s;
          xv[s] = (row < n && j < F0) ? Xprev[(int64_t)(base + row) * F0 + j] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) {
            const float av = stg[(4 * q + s) * SLD + 16 * ti + j];
            dw[ti][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xv[s], dw[ti][0], 0, 0, 0);
          }
        }
        __builtin_amdgcn_wave_barrier();
        CGNN_STAMP(4)
        continue;
      }

      // dW[o][col] += sum_m dT[m][o] X[m][col]; k <-> m = 4q + s
      const float4 pa = ld4(bnl + 4 * j), pb = ld4(bnl + HID + 4 * j);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float4 f;
        const float4 x = act4(yp[s], pa, pb, (keeps >> (8 * s)) & 0xFu, drop.scale, f);
        const float bx[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
          const float av = stg[(4 * q + s) * SLD + 16 * ti + j];
#pragma unroll
          for (int tj = 0; tj < 4; ++tj)
            dw[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bx[tj], dw[ti][tj], 0, 0, 0);
        }
      }
      // dX[row][col] = sum_o dT[row][o] W[o][col] as split-bf16 products (see mfma_split): lane
      // (q, j) takes row j of the block, k-step ms covers o = 16*(2ms) + 4q.. and 16*(2ms+1) + 4q..
      f32x4 dx[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        const Split8 a = split8(ld4(stg + j * SLD + 16 * (2 * ms) + 4 * q),
                                ld4(stg + j * SLD + 16 * (2 * ms + 1) + 4 * q));
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          const uint4 bh = wsp[((0 * 2 + ms) * 4 + tj) * 64 + lane];
          const uint4 bm = wsp[((1 * 2 + ms) * 4 + tj) * 64 + lane];
          const uint4 bl = wsp[((2 * 2 + ms) * 4 + tj) * 64 + lane];
          dx[tj] = mfma_split(a, bh, bm, bl, dx[tj]);
        }
      }
      __builtin_amdgcn_wave_barrier();
      const float4 pa2 = ld4(bnl + 4 * j), pb2 = ld4(bnl + HID + 4 * j);
      const float4 pmean = ld4(bnl + 2 * HID + 4 * j), pis = ld4(bnl + 3 * HID + 4 * j);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * b + 4 * q + r;
        if (row < n) {
          float4 f;
          act4(yp[r], pa2, pb2, (keeps >> (8 * r)) & 0xFu, drop.scale, f);
          const float4 dzp = make_float4(dx[0][r] * f.x, dx[1][r] * f.y, dx[2][r] * f.z, dx[3][r] * f.w);
          BW_ST(dZprev + (int64_t)(base + row) * HID + 4 * j, dzp);
          s1[0] += dzp.x; s1[1] += dzp.y; s1[2] += dzp.z; s1[3] += dzp.w;
          s2[0] = fmaf(dzp.x, (yp[r].x - pmean.x) * pis.x, s2[0]);
          s2[1] = fmaf(dzp.y, (yp[r].y - pmean.y) * pis.y, s2[1]);
          s2[2] = fmaf(dzp.z, (yp[r].z - pmean.z) * pis.z, s2[2]);
          s2[3] = fmaf(dzp.w, (yp[r].w - pmean.w) * pis.w, s2[3]);
        }
      }
      CGNN_STAMP(4)
    }
    CGNN_STAMP(5)
    __syncthreads();
    CGNN_STAMP(6)
  }

  // ---------------------------------------------------------------- workgroup reductions
  // db: 32 threads share a chunk j (threadIdx % 16); reduce through the tile memory.
  {
    double* red = reinterpret_cast<double*>(tile);       // [32][64]
    __syncthreads();
    const int g = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) red[g * 64 + 4 * j + i] = (double)db[i];
    __syncthreads();
    if (threadIdx.x < 64) {
      double s = 0.0;
      for (int g2 = 0; g2 < (NW * 64) / 16; ++g2) s += red[g2 * 64 + threadIdx.x];
      db_slab[(int64_t)blockIdx.x * 64 + threadIdx.x] = s;
    }
    __syncthreads();
  }
  if (!FIRST) {
    double* red = reinterpret_cast<double*>(tile);
    if (tail.acc) {
      // BatchNorm-backward coefficients of the layer below from the last workgroup's tail (bn_tail.h)
      reduce_stats<float, NW>(s1, s2, red, red + NW * 128);
      bn_tail_run(tail, red + NW * 128, reinterpret_cast<int*>(red + NW * 128 + 128), [](int) { return 0.f; });
    } else {
      reduce_stats<float, NW>(s1, s2, red, s_slab + (int64_t)blockIdx.x * 128);
    }
  }
  // dW: tree over the 8 waves through LDS (fixed order), wave 0 writes the partial.
  {
    constexpr int NTJ = FIRST ? 1 : 4;
    constexpr int PER = 64 * 16 * NTJ;                    // floats per wave partial
    float* red = tile;                                    // up to 4 * 4096 floats = 64 KB
    __syncthreads();
    for (int half = NW / 2; half >= 1; half >>= 1) {
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
                                                   const float* __restrict__ bn, DropCfg drop_in,
                                                   int use_drop, uint8_t* __restrict__ mask_out,
                                                   const int32_t* __restrict__ gptr, int B,
                                                   float* __restrict__ P, float* __restrict__ F1,
                                                   float* __restrict__ F2) {
  // F1/F2 (training): per graph and column, sum over the graph's rows of the factor f =
  // relu'(z) * keep / (1-p) and of f * xhat.  The readout's gradient is constant per graph
  // (dP[g] / (n_g + 1e-8)), so the BatchNorm-backward sums of the last layer are
  // sum_g dP[g]/n_g * F1[g] and sum_g dP[g]/n_g * F2[g]: the backward never re-reads Y.
  const DropCfg drop = drop_resolve(drop_in);
  __shared__ float red[3 * 16 * HID];
  const int j = threadIdx.x & 15, rr = threadIdx.x >> 4;
  const float4 a = ld4(bn + 4 * j), b = ld4(bn + HID + 4 * j);
  const float4 mean = ld4(bn + 2 * HID + 4 * j), is = ld4(bn + 3 * HID + 4 * j);
  for (int g = blockIdx.x; g < B; g += gridDim.x) {
    const int rbeg = gptr[g], rend = gptr[g + 1];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), f1 = s, f2 = s;
#define CGNN_POOL_U 4
    constexpr int U = CGNN_POOL_U;             // rows in flight per thread (latency-bound otherwise)
    for (int row0 = rbeg + rr; row0 < rend; row0 += 16 * U) {
      float4 yb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = row0 + 16 * u;
        yb[u] = row < rend ? ldnt4(Y + (int64_t)row * HID + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = row0 + 16 * u;
        if (row < rend) {
          uint32_t keep = 0xFu;
          if (use_drop) {
            keep = drop_bits(drop, (uint32_t)row, (uint32_t)j);
            if (mask_out) mask_out[(int64_t)row * 16 + j] = (uint8_t)keep;
          }
          float4 f;
          const float4 y = yb[u];
          const float4 x = act4(y, a, b, keep, drop.scale, f);
          s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
          if (F1) {
            f1.x += f.x; f1.y += f.y; f1.z += f.z; f1.w += f.w;
            f2.x = fmaf(f.x, (y.x - mean.x) * is.x, f2.x); f2.y = fmaf(f.y, (y.y - mean.y) * is.y, f2.y);
            f2.z = fmaf(f.z, (y.z - mean.z) * is.z, f2.z); f2.w = fmaf(f.w, (y.w - mean.w) * is.w, f2.w);
          }
        }
      }
    }
    st4(red + rr * HID + 4 * j, s);
    if (F1) {
      st4(red + (16 + rr) * HID + 4 * j, f1);
      st4(red + (32 + rr) * HID + 4 * j, f2);
    }
    __syncthreads();
    if (threadIdx.x < HID) {
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += red[k * HID + threadIdx.x];
      P[(int64_t)g * HID + threadIdx.x] = tot / ((float)(rend - rbeg) + 1e-8f);
    } else if (F1 && threadIdx.x < 3 * HID) {
      const int which = threadIdx.x / HID, col = threadIdx.x % HID;          // 1: F1, 2: F2
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += (double)red[(16 * which + k) * HID + col];
      (which == 1 ? F1 : F2)[(int64_t)g * HID + col] = (float)tot;
    }
    __syncthreads();
  }
}

// BatchNorm-backward sums of the last layer from the per-graph factor sums of k_pool_fwd:
// slab[wg][0..63] = sum_g dP[g]/(n_g+1e-8) * F1[g], slab[wg][64..127] = ... * F2[g]  (fp64).
__global__ void __launch_bounds__(128) k_pool_bwd_sums(const float* __restrict__ dP,
                                                       const float* __restrict__ F1,
                                                       const float* __restrict__ F2,
                                                       const int32_t* __restrict__ gptr, int B,
                                                       double* __restrict__ s_slab) {
  const int col = threadIdx.x & 63;
  const float* F = threadIdx.x < 64 ? F1 : F2;
  double acc = 0.0;
  for (int g = blockIdx.x; g < B; g += gridDim.x) {
    const float inv = 1.0f / ((float)(gptr[g + 1] - gptr[g]) + 1e-8f);
    acc += (double)(dP[(int64_t)g * HID + col] * inv) * (double)F[(int64_t)g * HID + col];
  }
  s_slab[(int64_t)blockIdx.x * 128 + threadIdx.x] = acc;
}

// The same sums folded over ALL graphs by one block per channel, finalised on the spot (per-rank
// BatchNorm: no exchange between the sums and the coefficients) -- k_pool_bwd_sums +
// k_bn_bwd_stats in one launch.
__device__ __forceinline__ double block_sum256(double v, double* sh);
__global__ void __launch_bounds__(256) k_pool_bwd_finalize(const float* __restrict__ dP,
                                                           const float* __restrict__ F1,
                                                           const float* __restrict__ F2,
                                                           const int32_t* __restrict__ gptr, int B,
                                                           double count, int zero_coef,
                                                           float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta,
                                                           float* __restrict__ bwc) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  double a1 = 0.0, a2 = 0.0;
  for (int g = threadIdx.x; g < B; g += 256) {
    const float inv = 1.0f / ((float)(gptr[g + 1] - gptr[g]) + 1e-8f);
    const double d = (double)(dP[(int64_t)g * HID + c] * inv);
    a1 += d * (double)F1[(int64_t)g * HID + c];
    a2 += d * (double)F2[(int64_t)g * HID + c];
  }
  const double S1 = block_sum256(a1, sh), S2 = block_sum256(a2, sh);
  if (threadIdx.x == 0) {
    dbeta[c] = (float)S1;
    dgamma[c] = (float)S2;
    bwc[c] = zero_coef ? 0.f : (float)(S1 / count);
    bwc[HID + c] = zero_coef ? 0.f : (float)(S2 / count);
  }
}

constexpr int PBTHR = 1024;  // readout backward: 64 row-lanes x 16 chunks (16 waves per CU)

__global__ void __launch_bounds__(PBTHR) k_pool_bwd(const float* __restrict__ dP,
                                                   const float* __restrict__ Y,
                                                   const float* __restrict__ bn, DropCfg drop,
                                                   int use_drop, const uint8_t* __restrict__ mask,
                                                   const int32_t* __restrict__ gptr, int B,
                                                   float* __restrict__ dZ,
                                                   double* __restrict__ s_slab) {
  __shared__ double red[(PBTHR / 16) * 128];
  const int j = threadIdx.x & 15, rr = threadIdx.x >> 4;
  const float4 a = ld4(bn + 4 * j), b = ld4(bn + HID + 4 * j);
  const float4 mean = ld4(bn + 2 * HID + 4 * j), is = ld4(bn + 3 * HID + 4 * j);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int g = blockIdx.x; g < B; g += gridDim.x) {
    const int rbeg = gptr[g], rend = gptr[g + 1];
    const float inv = 1.0f / ((float)(rend - rbeg) + 1e-8f);
    float4 gp = ld4(dP + (int64_t)g * HID + 4 * j);
    gp.x *= inv; gp.y *= inv; gp.z *= inv; gp.w *= inv;
    constexpr int U = 3;                       // rows in flight per thread
    constexpr int RS = PBTHR / 16;             // row stride
    for (int row0 = rbeg + rr; row0 < rend; row0 += RS * U) {
      float4 yb[U];
      uint32_t kb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = row0 + RS * u;
        yb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        kb[u] = 0u;
        if (row < rend) {
          yb[u] = ld4(Y + (int64_t)row * HID + 4 * j);
          kb[u] = use_drop ? mask[(int64_t)row * 16 + j] : 0xFu;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = row0 + RS * u;
        if (row < rend) {
          const float4 y = yb[u];
          float4 f;
          act4(y, a, b, kb[u], drop.scale, f);
          const float4 dz = make_float4(gp.x * f.x, gp.y * f.y, gp.z * f.z, gp.w * f.w);
          if (dZ) st4(dZ + (int64_t)row * HID + 4 * j, dz);
          s1[0] += dz.x; s1[1] += dz.y; s1[2] += dz.z; s1[3] += dz.w;
          s2[0] += (double)dz.x * ((y.x - mean.x) * is.x); s2[1] += (double)dz.y * ((y.y - mean.y) * is.y);
          s2[2] += (double)dz.z * ((y.z - mean.z) * is.z); s2[3] += (double)dz.w * ((y.w - mean.w) * is.w);
        }
      }
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
    for (int k = 0; k < PBTHR / 16; ++k) tot += red[k * 128 + threadIdx.x];
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
                                                     int take_cols, int ld_out,
                                                     float* __restrict__ out_tail = nullptr, int split = 0) {
  // one block per 4 output elements: 64 threads (one wave) per element, fixed-order tree
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  if (e < width)
    for (int r = lane; r < rows; r += 64) s += (double)slab[(int64_t)r * width + e];
  s = cgnn_wave_sum(s);
  if (e < width && lane == 0) {
    if (out_d) out_d[e] = s;
    if (out_tail && e >= split) {
      out_tail[e - split] = (float)s;
    } else if (out_f) {
      const int rr = e / out_cols, cc = e % out_cols;
      if (cc < take_cols) out_f[(int64_t)rr * ld_out + cc] = (float)s;
    }
  }
}

// several f64 slab -> f32 vector reductions in one launch (blockIdx.y = job): the bias gradients of
// all layers of a backward pass
__global__ void __launch_bounds__(256) k_slab_reduce_multi(cgnn_reduce_jobs jobs) {
  const int jb = blockIdx.y;
  const int width = jobs.width[jb], rows = jobs.rows[jb];
  const double* __restrict__ slab = jobs.slab[jb];
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (e >= width) return;                                  // (wave-uniform)
  double s = 0.0;
  for (int r = lane; r < rows; r += 64) s += slab[(int64_t)r * width + e];
  s = cgnn_wave_sum(s);
  if (lane == 0) jobs.out[jb][e] = (float)s;
}

__global__ void k_bn_finalize(const double* __restrict__ sums, double count,
                              const double* __restrict__ count_dev,
                              const float* __restrict__ gamma, const float* __restrict__ beta,
                              float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                              float eps, int training, float* __restrict__ bn_out,
                              const float* __restrict__ mean_offset) {
  const int c = threadIdx.x;
  if (c >= HID) return;
  if (count_dev) count = count_dev[0];
  // mean_offset (nullable): the statistics are those of y - mean_offset[c] (the factored layer 0 is
  // handed on without its constant term); the module's running mean is that of y
  const double off = mean_offset ? (double)mean_offset[c] : 0.0;
  float mean, var;
  if (training) {
    const double m = sums[c] / count;
    double v = sums[HID + c] / count - m * m;            // biased batch variance
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
    rmean[c] = (1.0f - momentum) * rmean[c] + momentum * (float)(m + off);
    rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unbiased;
  } else {
    mean = (float)((double)rmean[c] - off);
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
                                  const double* __restrict__ count_dev, int zero_coef,
                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                  float* __restrict__ bwc) {
  const int c = threadIdx.x;
  if (c >= HID) return;
  if (count_dev) count = count_dev[0];
  dbeta[c] = (float)sums[c];
  dgamma[c] = (float)sums[HID + c];
  bwc[c] = zero_coef ? 0.f : (float)(sums[c] / count);
  bwc[HID + c] = zero_coef ? 0.f : (float)(sums[HID + c] / count);
}

// ---- merged "reduce the per-workgroup partials + finalise" kernels (single-GPU fast path) ----
// Block-wide fixed-order sum of v over 256 threads (4 waves).
__device__ __forceinline__ double block_sum256(double v, double* sh) {
  v = cgnn_wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const double t = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return t;
}

// one block per channel c: S1 = sum_r slab[r][c], S2 = sum_r slab[r][64+c], then finalise
__global__ void __launch_bounds__(256) k_bn_fwd_stats(
    const double* __restrict__ slab, int rows, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
    float momentum, float eps, long long* __restrict__ tracked, float* __restrict__ bn_out,
    uint32_t* __restrict__ rng_state, int rng_n, const float* __restrict__ mean_offset) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  // graph replay: refresh the device dropout words here (cgnn_rng_advance's arithmetic) -- this
  // launch runs after every consumer of the previous step's words and before the first of this one
  if (rng_state && c == 0 && (int)threadIdx.x < rng_n)
    rng_state[threadIdx.x] = mix32(rng_state[threadIdx.x] + 0x9E3779B9u * (uint32_t)(threadIdx.x + 1));
  double a1 = 0.0, a2 = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) {
    a1 += slab[(int64_t)r * 128 + c];
    a2 += slab[(int64_t)r * 128 + HID + c];
  }
  const double S1 = block_sum256(a1, sh), S2 = block_sum256(a2, sh);
  if (threadIdx.x == 0) {
    const double m = S1 / count;
    double v = S2 / count - m * m;
    if (v < 0.0) v = 0.0;
    const float mean = (float)m, var = (float)v;
    const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
    // (mean_offset: the statistics are those of y - offset, the running mean is that of y)
    const float mean_y = mean_offset ? (float)(m + (double)mean_offset[c]) : mean;
    rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean_y;
    rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unbiased;
    const float invstd = 1.0f / sqrtf(var + eps);
    const float a = gamma[c] * invstd;
    bn_out[c] = a;
    bn_out[HID + c] = beta[c] - mean * a;
    bn_out[2 * HID + c] = mean;
    bn_out[3 * HID + c] = invstd;
    if (c == 0 && tracked) *tracked += 1;
  }
}

__global__ void __launch_bounds__(256) k_bn_bwd_stats(const double* __restrict__ slab, int rows,
                                                      double count, int zero_coef,
                                                      float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta,
                                                      float* __restrict__ bwc) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  double a1 = 0.0, a2 = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) {
    a1 += slab[(int64_t)r * 128 + c];
    a2 += slab[(int64_t)r * 128 + HID + c];
  }
  const double S1 = block_sum256(a1, sh), S2 = block_sum256(a2, sh);
  if (threadIdx.x == 0) {
    dbeta[c] = (float)S1;
    dgamma[c] = (float)S2;
    bwc[c] = zero_coef ? 0.f : (float)(S1 / count);
    bwc[HID + c] = zero_coef ? 0.f : (float)(S2 / count);
  }
}

// dW (f32 slab [rows][64*out_cols]) and db (f64 slab [rows][64]) in one launch.  A block folds
// RD_C consecutive output elements: thread (cc, rg) adds rows rg, rg + RD_G, ... of element cc
// (every load instruction reads 64-byte row pieces, up to 16 in flight per thread; 64 row groups, since
// the layer-0 slab has 2048 rows and the fold is a latency chain), then the RD_G partials are
// combined in fixed order.  [one wave per element with lane = row touched 64 lines
// per load: 32 us for the three layers of a step]
constexpr int RD_C = 16, RD_G = 64;      // 1024 threads
__host__ __device__ inline int dw_db_blocks(int out_cols) { return (HID * out_cols + HID) / RD_C; }
__device__ __forceinline__ void dw_db_reduce_block(const float* __restrict__ dw_slab,
                                                   const double* __restrict__ db_slab, int rows,
                                                   int out_cols, int take_cols,
                                                   float* __restrict__ dW, int ldw,
                                                   float* __restrict__ db, int block) {
  __shared__ double sh[RD_G][RD_C];
  const int nw = HID * out_cols;                    // a multiple of RD_C: a block is all dW or all db
  const int cc = threadIdx.x % RD_C, rg = threadIdx.x / RD_C;
  const int e = block * RD_C + cc;
  double s = 0.0;
  if (e < nw) {
    int r = rg;
    for (; r + 15 * RD_G < rows; r += 16 * RD_G) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = dw_slab[(int64_t)(r + u * RD_G) * nw + e];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    for (; r + 3 * RD_G < rows; r += 4 * RD_G) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = dw_slab[(int64_t)(r + u * RD_G) * nw + e];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += (double)v[u];
    }
    for (; r < rows; r += RD_G) s += (double)dw_slab[(int64_t)r * nw + e];
  } else {
    int r = rg;
    for (; r + 7 * RD_G < rows; r += 8 * RD_G) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = db_slab[(int64_t)(r + u * RD_G) * HID + (e - nw)];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < rows; r += RD_G) s += db_slab[(int64_t)r * HID + (e - nw)];
  }
  sh[rg][cc] = s;
  __syncthreads();
  __shared__ double sh2[RD_G / 8][RD_C];
  if (rg < RD_G / 8) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sh[8 * rg + k][cc];
    sh2[rg][cc] = t;
  }
  __syncthreads();
  if (rg == 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < RD_G / 8; ++k) t += sh2[k][cc];
    if (e < nw) {
      const int o = e / out_cols, c2 = e % out_cols;
      if (c2 < take_cols) dW[(int64_t)o * ldw + c2] = (float)t;
    } else {
      db[e - nw] = (float)t;
    }
  }
}

__global__ void __launch_bounds__(RD_C * RD_G) k_dw_db_reduce(const float* __restrict__ dw_slab,
                                                      const double* __restrict__ db_slab, int rows,
                                                      int out_cols, int take_cols,
                                                      float* __restrict__ dW, int ldw,
                                                      float* __restrict__ db) {
  dw_db_reduce_block(dw_slab, db_slab, rows, out_cols, take_cols, dW, ldw, db, blockIdx.x);
}

// several layers' slabs in ONE launch (the reductions do not feed the backward chain, so they can
// all wait for its end: one launch instead of one per layer)
__global__ void __launch_bounds__(RD_C * RD_G) k_dw_db_reduce_multi(cgnn_dw_jobs jobs) {
  int block = blockIdx.x;
#pragma unroll
  for (int i = 0; i < CGNN_DW_MAX_JOBS; ++i) {
    if (i >= jobs.n) return;
    const int nb = dw_db_blocks(jobs.out_cols[i]);
    if (block < nb) {
      dw_db_reduce_block(jobs.dw_slab[i], jobs.db_slab[i], jobs.rows[i], jobs.out_cols[i], jobs.take_cols[i],
                         jobs.dW[i], jobs.take_cols[i], jobs.db[i], block);
      return;
    }
    block -= nb;
  }
}

int g_grid_cache[CGNN_MAX_DEVICES] = {};
int g_grid_override = 0;   // cgnn_set_fused_grid (test hook): > 0 replaces the CU count

int fused_grid() {
  if (g_grid_override > 0) return g_grid_override;
  const int dev = cgnn_device_ordinal();
  if (g_grid_cache[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      return 256;
    g_grid_cache[dev] = cus;
  }
  return g_grid_cache[dev];
}

DropCfg make_drop(float p, uint64_t seed, int* use_drop) {
  DropCfg d;
  *use_drop = (p > 0.f) ? 1 : 0;
  double thr = (double)p * 65536.0 + 0.5;
  if (thr > 65535.0) thr = 65535.0;
  d.thr16 = (uint32_t)thr;
  // the reference's scale, 1/(1-p) (aten::native_dropout), not 1/(realised keep rate): with
  // replayed keep bits the arithmetic then matches the oracle to rounding
  d.scale = p > 0.f ? (float)(1.0 / (1.0 - (double)p)) : 1.0f;
  d.key0 = (uint32_t)(seed & 0xFFFFFFFFu) * 0x9E3779B9u + 0x85EBCA6Bu;
  d.key1 = (uint32_t)(seed >> 32) ^ 0xC2B2AE35u;
  d.dev_key = nullptr;
  return d;
}

__global__ void k_rng_advance(uint32_t* state, int n) {
  const int i = threadIdx.x;
  if (i < n) state[i] = mix32(state[i] + 0x9E3779B9u * (uint32_t)(i + 1));
}

bool l0src_ok(const cgnn_l0src* l0) {
  return l0 && l0->P0 && l0->W0 && l0->b0 && l0->F0 >= 1 && l0->F0 <= L0_FP;
}

// a tail descriptor the kernels can run (NULL = none = fine)
bool tail_ok(const cgnn_bn_tail* tl, int mode) {
  if (!tl) return true;
  if (!tl->acc || tl->mode != mode || !(tl->count > 0.0) || (reinterpret_cast<uintptr_t>(tl->acc) & 7)) return false;
  if (mode == 0)
    return tl->gamma && tl->beta && tl->running_mean && tl->running_var && tl->bn_out && tl->rng_n >= 0 &&
           tl->rng_n <= 64 && (tl->rng_n == 0 || tl->rng_state);
  return tl->dgamma && tl->dbeta && tl->bwc;
}

bool tiles_ok(const cgnn_tiles* t) {
  return t && t->num_tiles >= 0 && t->num_nodes >= 0 && t->max_tile_rows <= CGNN_FUSED_MAX_ROWS &&
         (t->num_tiles == 0 || (t->tile_ptr && t->tile_blk && t->blk_off_dst && t->ent_dst &&
                                t->blk_off_src && t->ent_src && t->dis));
}

}  // namespace

extern "C" {

#ifdef CGNN_STAMPS
// diagnostic build only: copy the stamp accumulators to the host and clear them
int cgnn_debug_stamps(unsigned long long* out_host) {
  if (hipDeviceSynchronize() != hipSuccess) return CGNN_ELAUNCH;
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return CGNN_ELAUNCH;
  static unsigned long long zeros[1024 * 8 * 8];
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros)) != hipSuccess) return CGNN_ELAUNCH;
  return CGNN_OK;
}
#endif

int cgnn_fused_grid(void) { return fused_grid(); }

int cgnn_set_fused_grid(int32_t workgroups) {
  if (workgroups < 0 || workgroups > 65535) return CGNN_EINVAL;
  g_grid_override = workgroups;
  return CGNN_OK;
}

int cgnn_gcn_fused_fwd_first(const cgnn_tiles* t, const float* X0, int32_t F0, const float* W0,
                             const float* bias, float* Y, double* stat_slab, int64_t stat_slab_bytes, void* stream) {
  if (!tiles_ok(t) || F0 <= 0 || F0 > CGNN_FUSED_MAX_F0) return t && t->max_tile_rows > CGNN_FUSED_MAX_ROWS ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (!X0 || !W0 || !bias || !Y) return CGNN_EINVAL;
  CGNN_NEED_BYTES(stat_slab, stat_slab_bytes, (int64_t)fused_grid() * 128 * (int64_t)sizeof(double));
  DropCfg d{};
  k_gcn_fwd<CGNN_FUSED_MAX_ROWS, true><<<fused_grid(), NTHR, 0, cgnn_stream(stream)>>>(
      *t, X0, F0, nullptr, d, 0, nullptr, W0, bias, Y, stat_slab);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_rng_advance(uint32_t* state, int32_t n, void* stream) {
  if (!state || n <= 0 || n > 64) return CGNN_EINVAL;
  k_rng_advance<<<1, 64, 0, cgnn_stream(stream)>>>(state, n);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_fwd(const cgnn_tiles* t, const float* Yprev, const cgnn_l0src* l0,
                       const float* bn_prev, float p_drop, uint64_t seed, const uint32_t* seed_dev,
                       uint8_t* mask_out, const float* W, const float* bias, float* Y,
                       double* stat_slab, int64_t stat_slab_bytes, const cgnn_bn_tail* tail, void* stream) {
  if (!tiles_ok(t)) return t && t->max_tile_rows > CGNN_FUSED_MAX_ROWS ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if ((!Yprev && !l0src_ok(l0)) || !bn_prev || !W || !bias || !Y || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (!tail_ok(tail, 0)) return CGNN_EINVAL;
  const cgnn_bn_tail tl = tail ? *tail : cgnn_bn_tail{};
  CGNN_NEED_BYTES(stat_slab, stat_slab_bytes, (int64_t)fused_grid() * 128 * (int64_t)sizeof(double));
  int use_drop;
  DropCfg d = make_drop(p_drop, seed, &use_drop);
  d.dev_key = seed_dev;
  if (Yprev)
    k_gcn_fwd_pf<CGNN_FUSED_MAX_ROWS, false><<<fused_grid(), PF_NTHR, 0, cgnn_stream(stream)>>>(
        *t, Yprev, cgnn_l0src{}, bn_prev, d, use_drop, mask_out, W, bias, Y, stat_slab, tl);
  else
    k_gcn_fwd_pf<CGNN_FUSED_MAX_ROWS, true><<<fused_grid(), PF_NTHR, 0, cgnn_stream(stream)>>>(
        *t, nullptr, *l0, bn_prev, d, use_drop, mask_out, W, bias, Y, stat_slab, tl);
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

int cgnn_bn_finalize(const double* sums, double count, const double* count_dev, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, int32_t training, float* bn_out, const float* mean_offset, void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || !bn_out) return CGNN_EINVAL;
  if (training && (!sums || (!count_dev && count <= 0.0))) return CGNN_EINVAL;
  k_bn_finalize<<<1, 64, 0, cgnn_stream(stream)>>>(sums, count, count_dev, gamma, beta, running_mean,
                                                   running_var, momentum, eps, training, bn_out, mean_offset);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_pool_fwd(const float* Y, const float* bn, float p_drop, uint64_t seed,
                            const uint32_t* seed_dev, uint8_t* mask_out, const int32_t* gptr,
                            int32_t num_graphs, float* P, float* F1, float* F2, void* stream) {
  if (num_graphs < 0 || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (num_graphs == 0) return CGNN_OK;
  if (!Y || !bn || !gptr || !P || (!F1) != (!F2)) return CGNN_EINVAL;
  int use_drop;
  DropCfg d = make_drop(p_drop, seed, &use_drop);
  d.dev_key = seed_dev;
  const int grid = num_graphs < 8 * fused_grid() ? num_graphs : 8 * fused_grid();
  k_pool_fwd<<<grid, PTHR, 0, cgnn_stream(stream)>>>(Y, bn, d, use_drop, mask_out, gptr, num_graphs, P,
                                                     F1, F2);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_pool_bwd_sums(const float* dP, const float* F1, const float* F2,
                                 const int32_t* gptr, int32_t num_graphs, double* s_slab, int64_t s_slab_bytes,
                                 void* stream) {
  if (num_graphs < 0 || !dP || !F1 || !F2 || !gptr || !s_slab) return CGNN_EINVAL;
  CGNN_NEED_BYTES(s_slab, s_slab_bytes, (int64_t)fused_grid() * 128 * (int64_t)sizeof(double));
  // exactly cgnn_fused_grid() workgroups so that the slab has the documented row count
  k_pool_bwd_sums<<<fused_grid(), 128, 0, cgnn_stream(stream)>>>(dP, F1, F2, gptr, num_graphs, s_slab);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_pool_bwd(const float* dP, const float* Y, const float* bn, float p_drop,
                            const uint8_t* mask, const int32_t* gptr, int32_t num_graphs,
                            float* dZ, double* s_slab, int64_t s_slab_bytes, void* stream) {
  if (num_graphs < 0 || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (!dP || !Y || !bn || !gptr || !s_slab) return CGNN_EINVAL;   /* dZ may be NULL: sums only */
  if (p_drop > 0.f && !mask) return CGNN_EINVAL;
  CGNN_NEED_BYTES(s_slab, s_slab_bytes, (int64_t)fused_grid() * 128 * (int64_t)sizeof(double));
  int use_drop;
  DropCfg d = make_drop(p_drop, 0, &use_drop);
  // exactly cgnn_fused_grid() workgroups so that the slab has the documented row count
  k_pool_bwd<<<fused_grid(), PBTHR, 0, cgnn_stream(stream)>>>(dP, Y, bn, d, use_drop, mask, gptr,
                                                             num_graphs, dZ, s_slab);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_bn_bwd_finalize(const double* sums, double count, const double* count_dev,
                         int32_t zero_coef, float* dgamma, float* dbeta, float* bwc, void* stream) {
  if (!sums || !dgamma || !dbeta || !bwc || (!count_dev && count <= 0.0)) return CGNN_EINVAL;
  k_bn_bwd_finalize<<<1, 64, 0, cgnn_stream(stream)>>>(sums, count, count_dev, zero_coef, dgamma, dbeta, bwc);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_bwd(const cgnn_tiles* t, const float* dZ, const float* Y, const float* bn,
                       const float* bwc, const float* Yprev, const cgnn_l0src* l0,
                       const float* bn_prev, float p_drop, const uint8_t* mask_prev, const float* W,
                       float* dZprev, double* s_slab_prev, int64_t s_slab_prev_bytes, float* dW_slab, int64_t dW_slab_bytes, double* db_slab, int64_t db_slab_bytes,
                       const float* dP, const int32_t* node_graph, const int32_t* gptr,
                       const uint8_t* mask_cur, const cgnn_bn_tail* tail, void* stream) {
  if (!tiles_ok(t)) return t && t->max_tile_rows > CGNN_FUSED_MAX_ROWS ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (!Y || !bn || !bwc || (!Yprev && !l0src_ok(l0)) || !bn_prev || !W || !dZprev || (!s_slab_prev && !tail) ||
      !dW_slab || !db_slab || p_drop < 0.f || p_drop >= 1.f)
    return CGNN_EINVAL;
  if (!tail_ok(tail, 1)) return CGNN_EINVAL;
  const cgnn_bn_tail tl = tail ? *tail : cgnn_bn_tail{};
  if (p_drop > 0.f && !mask_prev) return CGNN_EINVAL;
  if (dP ? (!node_graph || !gptr || (p_drop > 0.f && !mask_cur)) : !dZ) return CGNN_EINVAL;
  CGNN_NEED_BYTES(s_slab_prev, s_slab_prev_bytes, (int64_t)fused_grid() * 128 * (int64_t)sizeof(double));
  CGNN_NEED_BYTES(dW_slab, dW_slab_bytes, (int64_t)fused_grid() * HID * HID * (int64_t)sizeof(float));
  CGNN_NEED_BYTES(db_slab, db_slab_bytes, (int64_t)fused_grid() * HID * (int64_t)sizeof(double));
  int use_drop;
  DropCfg d = make_drop(p_drop, 0, &use_drop);
  PoolIn pin{dP, node_graph, gptr, mask_cur};
  const cgnn_l0src src = Yprev ? cgnn_l0src{} : *l0;
#define CGNN_BWD_NW 8
#define CGNN_BWD_LAUNCH(POOL, XP)                                                                      \
  k_gcn_bwd<CGNN_FUSED_MAX_ROWS, false, POOL, XP, CGNN_BWD_NW><<<fused_grid(), CGNN_BWD_NW * 64, 0, cgnn_stream(stream)>>>( \
      *t, pin, src, dZ, Y, bn, bwc, Yprev, 0, bn_prev, d, use_drop, mask_prev, W, dZprev, s_slab_prev, \
      dW_slab, db_slab, tl)
  if (dP) { if (Yprev) CGNN_BWD_LAUNCH(true, false); else CGNN_BWD_LAUNCH(true, true); }
  else    { if (Yprev) CGNN_BWD_LAUNCH(false, false); else CGNN_BWD_LAUNCH(false, true); }
#undef CGNN_BWD_LAUNCH
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_bwd_first(const cgnn_tiles* t, const float* dZ, const float* Y, const float* bn,
                             const float* bwc, const float* X0, int32_t F0, float* dW_slab, int64_t dW_slab_bytes,
                             double* db_slab, int64_t db_slab_bytes, float p_drop, const float* dP,
                             const int32_t* node_graph, const int32_t* gptr,
                             const uint8_t* mask_cur, void* stream) {
  if (!tiles_ok(t) || F0 <= 0 || F0 > CGNN_FUSED_MAX_F0) return t && t->max_tile_rows > CGNN_FUSED_MAX_ROWS ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (!Y || !bn || !bwc || !X0 || !dW_slab || !db_slab || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (dP ? (!node_graph || !gptr || (p_drop > 0.f && !mask_cur)) : !dZ) return CGNN_EINVAL;
  CGNN_NEED_BYTES(dW_slab, dW_slab_bytes, (int64_t)fused_grid() * HID * 16 * (int64_t)sizeof(float));
  CGNN_NEED_BYTES(db_slab, db_slab_bytes, (int64_t)fused_grid() * HID * (int64_t)sizeof(double));
  int use_drop;
  DropCfg d = make_drop(dP ? p_drop : 0.f, 0, &use_drop);
  PoolIn pin{dP, node_graph, gptr, mask_cur};
  if (dP)
    k_gcn_bwd<CGNN_FUSED_MAX_ROWS, true, true><<<fused_grid(), NTHR, 0, cgnn_stream(stream)>>>(
        *t, pin, cgnn_l0src{}, dZ, Y, bn, bwc, X0, F0, nullptr, d, use_drop, nullptr, nullptr, nullptr, nullptr,
        dW_slab, db_slab, cgnn_bn_tail{});
  else
    k_gcn_bwd<CGNN_FUSED_MAX_ROWS, true, false><<<fused_grid(), NTHR, 0, cgnn_stream(stream)>>>(
        *t, pin, cgnn_l0src{}, dZ, Y, bn, bwc, X0, F0, nullptr, d, use_drop, nullptr, nullptr, nullptr, nullptr,
        dW_slab, db_slab, cgnn_bn_tail{});
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_bn_stats_finalize(const double* slab, int32_t rows, double count, const float* gamma,
                           const float* beta, float* running_mean, float* running_var,
                           float momentum, float eps, int64_t* num_batches_tracked, float* bn_out,
                           void* stream) {
  if (!slab || rows <= 0 || count <= 0.0 || !gamma || !beta || !running_mean || !running_var || !bn_out)
    return CGNN_EINVAL;
  k_bn_fwd_stats<<<HID, 256, 0, cgnn_stream(stream)>>>(
      slab, rows, count, gamma, beta, running_mean, running_var, momentum, eps,
      reinterpret_cast<long long*>(num_batches_tracked), bn_out, nullptr, 0, nullptr);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_bn_stats_finalize_rng(const double* slab, int32_t rows, double count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, int64_t* num_batches_tracked, float* bn_out,
                               uint32_t* rng_state, int32_t rng_n, const float* mean_offset, void* stream) {
  if (!slab || rows <= 0 || count <= 0.0 || !gamma || !beta || !running_mean || !running_var || !bn_out)
    return CGNN_EINVAL;
  if (rng_n < 0 || rng_n > 64 || (rng_n > 0 && !rng_state)) return CGNN_EINVAL;
  k_bn_fwd_stats<<<HID, 256, 0, cgnn_stream(stream)>>>(
      slab, rows, count, gamma, beta, running_mean, running_var, momentum, eps,
      reinterpret_cast<long long*>(num_batches_tracked), bn_out, rng_state, rng_n, mean_offset);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_fused_pool_bwd_finalize(const float* dP, const float* F1, const float* F2,
                                     const int32_t* gptr, int32_t num_graphs, double count,
                                     int32_t zero_coef, float* dgamma, float* dbeta, float* bwc,
                                     void* stream) {
  if (num_graphs < 0 || count <= 0.0 || !dP || !F1 || !F2 || !gptr || !dgamma || !dbeta || !bwc)
    return CGNN_EINVAL;
  k_pool_bwd_finalize<<<HID, 256, 0, cgnn_stream(stream)>>>(dP, F1, F2, gptr, num_graphs, count, zero_coef,
                                                           dgamma, dbeta, bwc);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_bn_bwd_stats_finalize(const double* slab, int32_t rows, double count, int32_t zero_coef,
                               float* dgamma, float* dbeta, float* bwc, void* stream) {
  if (!slab || rows <= 0 || count <= 0.0 || !dgamma || !dbeta || !bwc) return CGNN_EINVAL;
  k_bn_bwd_stats<<<HID, 256, 0, cgnn_stream(stream)>>>(slab, rows, count, zero_coef, dgamma, dbeta, bwc);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_dw_db_reduce(const float* dw_slab, const double* db_slab, int32_t rows, int32_t out_cols,
                      int32_t take_cols, float* dW, int32_t ld_dw, float* db, void* stream) {
  if (!dw_slab || !db_slab || !dW || !db || rows <= 0 || out_cols <= 0 || take_cols <= 0 ||
      take_cols > out_cols || ld_dw < take_cols)
    return CGNN_EINVAL;
  k_dw_db_reduce<<<(HID * out_cols + HID) / RD_C, RD_C * RD_G, 0, cgnn_stream(stream)>>>(dw_slab, db_slab, rows, out_cols,
                                                                 take_cols, dW, ld_dw, db);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_dw_db_reduce_multi(const cgnn_dw_jobs* jobs, void* stream) {
  if (!jobs || jobs->n < 1 || jobs->n > CGNN_DW_MAX_JOBS) return CGNN_EINVAL;
  int total = 0;
  for (int i = 0; i < jobs->n; ++i) {
    if (!jobs->dw_slab[i] || !jobs->db_slab[i] || !jobs->dW[i] || !jobs->db[i] || jobs->rows[i] <= 0 ||
        jobs->out_cols[i] <= 0 || jobs->take_cols[i] <= 0 || jobs->take_cols[i] > jobs->out_cols[i])
      return CGNN_EINVAL;
    total += (HID * jobs->out_cols[i] + HID) / RD_C;
  }
  k_dw_db_reduce_multi<<<total, RD_C * RD_G, 0, cgnn_stream(stream)>>>(*jobs);
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

int cgnn_slab_reduce_f32_split(const float* slab, int32_t rows, int32_t width, int32_t split, float* out,
                               float* out_tail, void* stream) {
  if (!slab || !out || !out_tail || rows <= 0 || width <= 0 || split <= 0 || split >= width) return CGNN_EINVAL;
  k_slab_reduce<float><<<(width + 3) / 4, 256, 0, cgnn_stream(stream)>>>(slab, rows, width, nullptr, out, width,
                                                                        width, width, out_tail, split);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_slab_reduce_f64_multi(const cgnn_reduce_jobs* jobs, void* stream) {
  if (!jobs || jobs->n < 0 || jobs->n > CGNN_REDUCE_MAX_JOBS) return CGNN_EINVAL;
  if (jobs->n == 0) return CGNN_OK;
  int wmax = 0;
  for (int j = 0; j < jobs->n; ++j) {
    if (!jobs->slab[j] || !jobs->out[j] || jobs->rows[j] <= 0 || jobs->width[j] <= 0) return CGNN_EINVAL;
    wmax = jobs->width[j] > wmax ? jobs->width[j] : wmax;
  }
  k_slab_reduce_multi<<<dim3((wmax + 3) / 4, jobs->n), 256, 0, cgnn_stream(stream)>>>(*jobs);
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
