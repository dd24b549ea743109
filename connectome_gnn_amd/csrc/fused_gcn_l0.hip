// fused_gcn_l0.hip -- layer 0 of the fused GCN path, specialised for narrow input features.
//
// The reference evaluates every layer as  A_hat (X W^T) + b  (models.py:111-114).  For layer 0
// the input has F0 = 5 columns, so by associativity
//       Y0  = (A_hat X0) W0^T + b            dW0 = dY0^T (A_hat X0)
// only the NARROW aggregate P0 = A_hat X0 ([Nn, F0], 8x less LDS traffic than a 64-wide one) is
// ever needed; it is computed once in forward and kept (32 bytes per node).  The layer-0
// backward then needs no aggregation at all: it is a streaming reduction over (dY0, P0).
//
//   k_l0_fwd : per tile, X0*dis -> LDS [rows][8]; one thread per destination row walks its
//              blocked-ELL entries; P0 -> LDS + HBM; Y0 = P0 W0^T + b on the vector ALUs
//              (5 FMAs per output) for the BatchNorm sums in the epilogue.  Y0 itself is
//              written only on request (Y != NULL): the fused path rebuilds its rows from P0
//              wherever they are needed (l0src.h), 32 instead of 256 bytes per node.
//              25 KB of LDS per workgroup -> 6 workgroups per CU hide the gather latency.
//   k_l0_bwd : dY0 = BatchNorm'(dZ0); dW0 += dY0^T P0; db0 += dY0  (no LDS tile, no metadata;
//              Y0 read from HBM or rebuilt from the P0 row it loads anyway).
#include "common.h"
#include "l0src.h"
#include "bn_tail.h"

namespace {

constexpr int HID = CGNN_FUSED_HIDDEN;
constexpr int FP = 8;                 // padded feature count (F0 <= 8)
constexpr int MAXR = CGNN_FUSED_MAX_ROWS;
constexpr int L0THR = 384;            // forward: one thread per row of a <=384-row tile
constexpr int L0BTHR = 256;           // backward (streaming)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
#define L0NT ldnt4
#define L0NTE(p) ldnt(p)
#define L0STNT stnt4
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

#define CGNN_L0_MINW 8
template <bool WRITE_Y>
__global__ void __launch_bounds__(L0THR, WRITE_Y ? 5 : CGNN_L0_MINW) k_l0_fwd(cgnn_tiles t, const float* __restrict__ X0, int F0,
                                                  const float* __restrict__ W0,
                                                  const float* __restrict__ bias,
                                                  float* __restrict__ P0, float* __restrict__ Y,
                                                  double* __restrict__ stat_slab,
                                                  const float* __restrict__ center,
                                                  float* __restrict__ w_eff,
                                                  float* __restrict__ mean_offset, cgnn_bn_tail tail) {
  __shared__ __attribute__((aligned(16))) float smem[2 * MAXR * FP];   // 24 KB -> 6 WGs per CU
  float* xs = smem;
  float* ps = smem + MAXR * FP;
  double* red = reinterpret_cast<double*>(smem);                      // reused after the loop
  const int j = threadIdx.x & 15, rr = threadIdx.x >> 4;       // rr in [0, 24)
  const uint2* ent = static_cast<const uint2*>(t.ent_dst);
  // W0^T in LDS: wl[k][col] = W0[col][k] (zero for k >= F0); occupancy matters more here than
  // 32 registers of weights (the row gather is hidden by many resident waves, not by ILP)
  __shared__ __attribute__((aligned(16))) float wl[WRITE_Y ? L0_LDS_FLOATS : 4];
  if (WRITE_Y) l0_stage(wl, cgnn_l0src{nullptr, W0, bias, F0}, L0THR);
  // WRITE_Y = false (the fused path: Y0 is never written): the BatchNorm sums of Y0 = P0 W0^T + b
  // come from the second moments of P0 -- M[k][l] = sum_rows p_k p_l over the 9 "columns"
  // (p_0..p_7, 1) -- accumulated on the matrix pipe (v_mfma_f32_16x16x4_f32, A = B = a 4-row
  // slice of P0, an exact fmaf chain), flushed to fp64 after every tile, and turned into
  // sum y / sum y^2 once per workgroup at the end.  No per-element VALU work, no W reads, 12
  // registers: the kernel fits 64 VGPRs = 8 waves per SIMD (the Y0-rebuilding form below needs 89
  // and was LDS-bound on its W reads: 158 -> see DESIGN 5).
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  // per-wave fp64 accumulators of the 9 x 9 moments live in LDS (4 KB): registers are what caps
  // the occupancy of this kernel
  constexpr int NM = FP + 1, NWV = L0THR / 64;
  __shared__ double macc[WRITE_Y ? 1 : NWV * NM * NM];
  if (!WRITE_Y)
    for (int i = threadIdx.x; i < NWV * NM * NM; i += L0THR) macc[i] = 0.0;
  // CENTRED form (center != NULL, F0 <= 7).  With c = center[0..F0) (a per-column constant near the
  // column means of X0) and rbar = center[7] (near the mean of r = A_hat 1, the row sums of the
  // normalised operator) -- cgnn_gcn_l0_center --
  //     A_hat X0 = A_hat (X0 - 1 c^T) + r c^T
  //     P0' = [A_hat (X0 - 1 c^T) | r - rbar | 0..]   (column F0 = the aggregated ones column, fp64 sum)
  //     W'  = [W0 | W0 c]                              ([64][F0 + 1], `w_eff`)
  //     Y0  = P0 W0^T + b = P0' W'^T + (b + rbar W0 c)
  // The layer is handed on WITHOUT its constant term: consumers rebuild y_c = P0' W'^T (bias 0) and
  // BatchNorm, which is invariant under a per-channel shift, is finalised on y_c; the constant
  // b + rbar W0 c (`mean_offset`, fp64 -> fp32, written by workgroup 0) only enters the running mean.
  // Same function of the parameters, but everything that is aggregated, stored, squared, multiplied
  // or summed in fp32 lives at the scale of the features' SPREAD: raw features far from zero
  // (un-normalised strength / degree columns, mean >> sigma) otherwise put rounding of the size
  // 2^-24 * mean into every aggregated row and every rebuilt y, and lose log2((mean/sigma)^2) bits
  // of the BatchNorm variance to cancellation.
  constexpr int RC = FP - 1;                         // the ones column
  __shared__ float cvec[FP];
  if (!WRITE_Y) {
    if (threadIdx.x < FP)
      cvec[threadIdx.x] = (center && (threadIdx.x < F0 || threadIdx.x == RC)) ? center[threadIdx.x] : 0.f;
    if (w_eff && blockIdx.x == 0 && threadIdx.x < HID) {
      double last = 0.0;
      for (int k = 0; k < F0; ++k) {
        const float wv = W0[threadIdx.x * F0 + k];
        w_eff[threadIdx.x * (F0 + 1) + k] = wv;
        last += (double)wv * (double)center[k];
      }
      w_eff[threadIdx.x * (F0 + 1) + F0] = (float)last;
      if (mean_offset)
        mean_offset[threadIdx.x] = (float)((double)bias[threadIdx.x] + (double)(float)last * (double)center[RC]);
    }
    __syncthreads();
  }
  const bool centred = !WRITE_Y && center != nullptr;
  // per-thread partial sums: a thread sees <= ~16 rows per tile, a few tiles; sum y^2 in fp64 (it
  // cancels against mean^2 downstream), the cross-thread / cross-workgroup combination is fp64
  float s1[4] = {0, 0, 0, 0};
  double s2[4] = {0, 0, 0, 0};

  for (int tid = blockIdx.x; tid < t.num_tiles; tid += gridDim.x) {
    const int base = t.tile_ptr[tid];
    const int n = t.tile_ptr[tid + 1] - base;
    const int gb0 = t.tile_blk[tid];
    // 1. dis * X0 -> LDS (zero-padded to 8 columns)
    {
      constexpr int NI = MAXR * FP / L0THR;            // 8 elements per thread
      constexpr int NB = WRITE_Y ? NI : NI / 2;         // loads in flight (register budget)
#pragma unroll
      for (int u0 = 0; u0 < NI; u0 += NB) {
        float xv[NB], dvv[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int idx = threadIdx.x + L0THR * (u0 + u), r = idx >> 3, k = idx & 7;
          xv[u] = dvv[u] = 0.f;
          if (r < n && (k < F0 || (centred && k == RC))) {
            xv[u] = k < F0 ? X0[(int64_t)(base + r) * F0 + k] - (WRITE_Y ? 0.f : cvec[k]) : 1.f;
            dvv[u] = t.dis[base + r];
          }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) xs[threadIdx.x + L0THR * (u0 + u)] = xv[u] * dvv[u];
      }
    }
    __syncthreads();
    // 2. narrow aggregate, one thread per destination row (entries of 16 rows are contiguous)
    for (int r = threadIdx.x; r < n; r += L0THR) {
      const int b = r >> 4, i = r & 15;
      const int off0 = t.blk_off_dst[gb0 + b];
      const int width = (t.blk_off_dst[gb0 + b + 1] - off0) >> 4;
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
      double ar = 0.0;                                 // the ones column: r = A_hat 1, summed in fp64
      const uint2* e = ent + off0 + i;
      // entries stream from HBM: fetch EB steps at a time (independent loads), then consume
#define CGNN_L0_EB 2
      constexpr int EB = WRITE_Y ? 6 : CGNN_L0_EB;
      for (int s0 = 0; s0 < width; s0 += EB) {
        uint2 eb[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) eb[u] = s0 + u < width ? L0NTE(e + 16 * (s0 + u)) : make_uint2(0u, 0u);
#pragma unroll
        for (int u = 0; u < EB; ++u) {
          const float w = __uint_as_float(eb[u].y);          // padding: weight 0, row 0
          const float* src = xs + (eb[u].x >> 8) * FP;       // entry offset = 256 * local row
          const float4 v0 = ld4(src), v1 = ld4(src + 4);
          a0.x = fmaf(w, v0.x, a0.x); a0.y = fmaf(w, v0.y, a0.y); a0.z = fmaf(w, v0.z, a0.z); a0.w = fmaf(w, v0.w, a0.w);
          a1.x = fmaf(w, v1.x, a1.x); a1.y = fmaf(w, v1.y, a1.y); a1.z = fmaf(w, v1.z, a1.z);
          if (WRITE_Y) a1.w = fmaf(w, v1.w, a1.w);
          else ar = fma((double)w, (double)v1.w, ar);
        }
      }
      const float dv = t.dis[base + r];
      a0 = make_float4(a0.x * dv, a0.y * dv, a0.z * dv, a0.w * dv);
      a1 = make_float4(a1.x * dv, a1.y * dv, a1.z * dv, a1.w * dv);
      if (!WRITE_Y) {
        // the ones column was aggregated in LDS column 7; in P0' it sits right behind the features
        // (column F0), so that consumers rebuild rows from F0 + 1 columns
        if (centred) {
          const float rc = (float)(ar * (double)dv - (double)cvec[RC]);
          a0.y = F0 == 1 ? rc : a0.y; a0.z = F0 == 2 ? rc : a0.z; a0.w = F0 == 3 ? rc : a0.w;
          a1.x = F0 == 4 ? rc : a1.x; a1.y = F0 == 5 ? rc : a1.y; a1.z = F0 == 6 ? rc : a1.z;
          a1.w = F0 == 7 ? rc : 0.f;
        } else {
          a1.w = (float)(ar * (double)dv);           // (an 8th feature column, if there is one)
        }
      }
      st4(ps + r * FP, a0); st4(ps + r * FP + 4, a1);
      L0STNT(P0 + (int64_t)(base + r) * FP, a0); L0STNT(P0 + (int64_t)(base + r) * FP + 4, a1);
    }
    __syncthreads();
    if (!WRITE_Y) {
      // 3. second moments of this tile's P0 rows: wave w takes the 16-row blocks w, w + 6, ...
      if (stat_slab || tail.acc) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ci = lane & 15, kq = lane >> 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int rb = wave; 16 * rb < n; rb += L0THR / 64) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * rb + 4 * g + kq;
            float v = 0.f;
            if (row < n) v = ci < FP ? ps[row * FP + ci] : (ci == FP ? 1.f : 0.f);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v, v, acc, 0, 0, 0);
          }
        }
        // lane (ci, kq) holds M[4*kq + r][ci], r = 0..3 (wave-private slots: plain read-add-write)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 4 * kq + r;
          if (m < NM && ci < NM) macc[(wave * NM + m) * NM + ci] += (double)acc[r];
        }
      }
      continue;                                      // (the next tile's phase-1 barrier orders ps)
    }
    // 3. Y0 = P0 W0^T + b: thread (row rr + 16*it, columns 4j..4j+3)
    // (two rows per pass share every W0^T read: this phase is LDS-bound)
    constexpr int RS = L0THR / 16;                   // 24 row lanes
    for (int r = rr; r < n; r += 2 * RS) {
      const int r2 = r + RS;
      const bool two = r2 < n;
      const float4 p0 = ld4(ps + r * FP), p1 = ld4(ps + r * FP + 4);
      const float4 q0 = two ? ld4(ps + r2 * FP) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 q1 = two ? ld4(ps + r2 * FP + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 y = l0_rebuild4(p0, p1, wl, 4 * j, F0);
      const float4 z = l0_rebuild4(q0, q1, wl, 4 * j, F0);
      st4(Y + (int64_t)(base + r) * HID + 4 * j, y);
      s1[0] += y.x; s1[1] += y.y; s1[2] += y.z; s1[3] += y.w;
      s2[0] += (double)y.x * y.x; s2[1] += (double)y.y * y.y;
      s2[2] += (double)y.z * y.z; s2[3] += (double)y.w * y.w;
      if (two) {
        st4(Y + (int64_t)(base + r2) * HID + 4 * j, z);
        s1[0] += z.x; s1[1] += z.y; s1[2] += z.z; s1[3] += z.w;
        s2[0] += (double)z.x * z.x; s2[1] += (double)z.y * z.y;
        s2[2] += (double)z.z * z.z; s2[3] += (double)z.w * z.w;
      }
    }
    __syncthreads();
  }
  if (!stat_slab && !tail.acc) return;
  __syncthreads();
  if (!WRITE_Y) {
    // moments -> sums: the waves' accumulators folded in fixed order, then, with the moments M of
    // the stored rows p' and the weights W' the consumers use (fp32-rounded last column), in fp64:
    //   S1[c] = sum_k W'[c][k] M[k][8] + M[8][8] b[c]                 (b = 0 in the centred form:
    //   S2[c] = sum_kl W'[c][k] W'[c][l] M[k][l] + 2 b[c] sum_k W'[c][k] M[k][8] + M[8][8] b[c]^2    sums of y_c)
    if (threadIdx.x < NM * NM) {
      double tsum = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < NWV; ++w2) tsum += macc[w2 * NM * NM + threadIdx.x];
      red[threadIdx.x] = tsum;                       // M[k][l] at red[k * NM + l]
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int c = threadIdx.x & 63;
      const double bc = centred ? 0.0 : (double)bias[c], cnt = red[FP * NM + FP];
      double* wc = red + 128 + FP * threadIdx.x;      // (smem is dead here; M occupies red[0..80])
      double last = 0.0;
#pragma unroll 1
      for (int k = 0; k < F0; ++k) {
        wc[k] = (double)W0[c * F0 + k];
        last += wc[k] * (double)cvec[k];
      }
      if (centred) wc[F0] = (double)(float)last;      // exactly the w_eff value
      const int FA = centred ? F0 + 1 : F0;
      double lin = 0.0;                              // sum_k W'[c][k] M1[k]
#pragma unroll 1
      for (int k = 0; k < FA; ++k) lin += wc[k] * red[k * NM + FP];
      double out = lin + cnt * bc;
      if (threadIdx.x >= 64) {
        double quad = 0.0;
#pragma unroll 1
        for (int k = 0; k < FA; ++k) {
          double rowsum = 0.0;
#pragma unroll 1
          for (int l = 0; l < FA; ++l) rowsum += wc[l] * red[k * NM + l];
          quad += wc[k] * rowsum;
        }
        out = quad + 2.0 * bc * lin + cnt * bc * bc;
      }
      if (tail.acc) red[1280 + threadIdx.x] = out;     // (red[0..1152) hold M and the threads' weight rows)
      else stat_slab[(int64_t)blockIdx.x * 128 + threadIdx.x] = out;
    }
    if (tail.acc) {
      // the layer's BatchNorm finalised by the workgroup that arrives last (bn_tail.h).  The centred form's
      // constant b + rbar W0 c -- which the statistics were taken without and only the running mean sees --
      // is recomputed by the tail's threads exactly as workgroup 0 wrote it to `mean_offset`
      __syncthreads();
      bn_tail_run(tail, red + 1280, reinterpret_cast<int*>(red + 1408), [&](int c) {
        if (!centred) return 0.f;
        double last = 0.0;
        for (int k = 0; k < F0; ++k) last += (double)W0[c * F0 + k] * (double)center[k];
        return (float)((double)bias[c] + (double)(float)last * (double)center[RC]);
      });
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    red[rr * 128 + 4 * j + i] = (double)s1[i];
    red[rr * 128 + 64 + 4 * j + i] = s2[i];
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < L0THR / 16; ++k) tot += red[k * 128 + threadIdx.x];
    stat_slab[(int64_t)blockIdx.x * 128 + threadIdx.x] = tot;
  }
}

// dW0[o][k] = sum_rows dY0[row][o] * P0[row][k];  db0[o] = sum_rows dY0[row][o]
template <bool REBUILD>
__global__ void __launch_bounds__(L0BTHR) k_l0_bwd(const float* __restrict__ dZ,
                                                  const float* __restrict__ Y, cgnn_l0src l0,
                                                  const float* __restrict__ bn,
                                                  const float* __restrict__ bwc,
                                                  const float* __restrict__ P0, int64_t nn,
                                                  float* __restrict__ dW_slab,
                                                  double* __restrict__ db_slab,
                                                  const float* __restrict__ center) {
  __shared__ float redw[16 * HID * FP];          // 32 KB
  __shared__ double redb[16 * HID];
  __shared__ __attribute__((aligned(16))) float wl[REBUILD ? L0_LDS_FLOATS : 4];
  if (REBUILD) {
    l0_stage(wl, l0, L0BTHR);
    __syncthreads();
  }
  const int j = threadIdx.x & 15, rr = threadIdx.x >> 4;
  const float4 ca = ld4(bn + 4 * j), cmean = ld4(bn + 2 * HID + 4 * j), cis = ld4(bn + 3 * HID + 4 * j);
  const float4 c1 = ld4(bwc + 4 * j), c2 = ld4(bwc + HID + 4 * j);
  // centred form (see k_l0_fwd): P0 holds P0' (column FT = r - rbar, FT the true feature count), l0
  // the weights W' ([64][FT + 1]) and a zero bias; with y = P0' W'^T + b + rbar W0 c the weight
  // gradient is     dW0[:, k] = dW'[:, k] + c[k] (dW'[:, FT] + rbar db0),
  // formed in the fp64 fold below (columns >= FT of the slab are dropped by the caller's reduce).
  __shared__ float shs[FP];
  const int FT = center ? l0.F0 - 1 : FP - 1;
  if (threadIdx.x < FP) shs[threadIdx.x] = center ? center[threadIdx.x] : 0.f;
  float dw[4][FP];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < FP; ++k) dw[c][k] = 0.f;
  float db[4] = {0.f, 0.f, 0.f, 0.f};
#define CGNN_L0B_U 4
  constexpr int U = CGNN_L0B_U;                  // rows in flight per thread
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t row0 = (int64_t)blockIdx.x * 16 + rr; row0 < nn; row0 += stride * U) {
    float4 zb[U], yb[U], pa[U], pb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + stride * u;
      zb[u] = yb[u] = pa[u] = pb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nn) {
        zb[u] = L0NT(dZ + row * HID + 4 * j);
        if (!REBUILD) yb[u] = ld4(Y + row * HID + 4 * j);
        pa[u] = L0NT(P0 + row * FP);
        pb[u] = L0NT(P0 + row * FP + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + stride * u;
      if (row < nn) {
        const float4 dz = zb[u];
        const float4 y = REBUILD ? l0_rebuild4(pa[u], pb[u], wl, 4 * j, l0.F0) : yb[u];
        const float dy[4] = {ca.x * (dz.x - c1.x - (y.x - cmean.x) * cis.x * c2.x),
                             ca.y * (dz.y - c1.y - (y.y - cmean.y) * cis.y * c2.y),
                             ca.z * (dz.z - c1.z - (y.z - cmean.z) * cis.z * c2.z),
                             ca.w * (dz.w - c1.w - (y.w - cmean.w) * cis.w * c2.w)};
        const float pv[FP] = {pa[u].x, pa[u].y, pa[u].z, pa[u].w, pb[u].x, pb[u].y, pb[u].z, pb[u].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          db[c] += dy[c];
#pragma unroll
          for (int k = 0; k < FP; ++k) dw[c][k] = fmaf(dy[c], pv[k], dw[c][k]);
        }
      }
    }
  }
  // reduce over the 16 row-lanes that share a column chunk (fixed order)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    redb[rr * HID + 4 * j + c] = (double)db[c];
#pragma unroll
    for (int k = 0; k < FP; ++k) redw[(rr * HID + 4 * j + c) * FP + k] = dw[c][k];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < HID * FP; e += L0BTHR) {
    double tot = 0.0, ext = 0.0, dbt = 0.0;          // this element; the ones column and db of its row
    const int col = e / FP, k = e % FP;
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) {
      tot += (double)redw[r2 * HID * FP + e];
      ext += (double)redw[r2 * HID * FP + col * FP + FT];
      dbt += redb[r2 * HID + col];
    }
    if (center && k < FT) tot += (double)shs[k] * (ext + (double)shs[FP - 1] * dbt);
    dW_slab[(int64_t)blockIdx.x * HID * FP + e] = (float)tot;
  }
  if (threadIdx.x < HID) {
    double tot = 0.0;
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) tot += redb[r2 * HID + threadIdx.x];
    db_slab[(int64_t)blockIdx.x * HID + threadIdx.x] = tot;
  }
}

// Centring constants of the factored layer 0, from the batch's first non-empty tile:
// center[k] = mean of X0[:, k] (k < F0), center[7] = mean of r = A_hat 1 (the normalised operator's
// row sums); any finite values are exact, values near the true means are accurate.
__global__ void __launch_bounds__(L0THR) k_l0_center(cgnn_tiles t, const float* __restrict__ X0, int F0,
                                                     float* __restrict__ center) {
  __shared__ float ds[MAXR];
  __shared__ double part[L0THR / 64][FP];
  const uint2* ent = static_cast<const uint2*>(t.ent_dst);
  int tid = 0;
  while (tid < t.num_tiles && t.tile_ptr[tid + 1] == t.tile_ptr[tid]) ++tid;
  if (tid >= t.num_tiles) {
    if (threadIdx.x < FP) center[threadIdx.x] = threadIdx.x == FP - 1 ? 1.f : 0.f;
    return;
  }
  const int base = t.tile_ptr[tid], n = t.tile_ptr[tid + 1] - base, gb0 = t.tile_blk[tid];
  for (int r = threadIdx.x; r < MAXR; r += L0THR) ds[r] = r < n ? t.dis[base + r] : 0.f;
  __syncthreads();
  double a[FP];
#pragma unroll
  for (int k = 0; k < FP; ++k) a[k] = 0.0;
  for (int r = threadIdx.x; r < n; r += L0THR) {
#pragma unroll
    for (int k = 0; k < FP - 1; ++k)
      if (k < F0) a[k] += (double)X0[(int64_t)(base + r) * F0 + k];
    const int b = r >> 4, i = r & 15;
    const int off0 = t.blk_off_dst[gb0 + b];
    const int width = (t.blk_off_dst[gb0 + b + 1] - off0) >> 4;
    double acc = 0.0;
    for (int s0 = 0; s0 < width; ++s0) {
      const uint2 e = ent[off0 + i + 16 * s0];
      acc += (double)__uint_as_float(e.y) * (double)ds[e.x >> 8];
    }
    a[FP - 1] += acc * (double)ds[r];
  }
#pragma unroll
  for (int k = 0; k < FP; ++k) {
    const double v = cgnn_wave_sum(a[k]);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < FP) {
    double tot = 0.0;
    for (int w2 = 0; w2 < L0THR / 64; ++w2) tot += part[w2][threadIdx.x];
    center[threadIdx.x] = (float)(tot / (double)n);
  }
}

#define CGNN_L0_GRID_MULT 8
// workgroups (= slab rows) of both kernels for a batch of `nn` nodes: one per 256 nodes, between
// one and CGNN_L0_GRID_MULT per CU (a small batch leaves fewer slab rows to fold afterwards)
int l0_grid(int64_t nn) {
  const int64_t cus = cgnn_fused_grid(), want = (nn + 255) / 256;
  return (int)(want < cus ? cus : (want > CGNN_L0_GRID_MULT * cus ? CGNN_L0_GRID_MULT * cus : want));
}

}  // namespace

extern "C" {

int cgnn_l0_grid(int64_t num_nodes) { return num_nodes < 0 ? CGNN_EINVAL : l0_grid(num_nodes); }

int cgnn_gcn_l0_center(const cgnn_tiles* t, const float* X0, int32_t F0, float* center, void* stream) {
  if (!t || F0 <= 0 || F0 > FP || t->max_tile_rows > CGNN_FUSED_MAX_ROWS) return t && F0 > FP ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (!center) return CGNN_EINVAL;
  if (t->num_tiles > 0 && (!X0 || !t->tile_ptr || !t->tile_blk || !t->blk_off_dst || !t->ent_dst || !t->dis))
    return CGNN_EINVAL;
  k_l0_center<<<1, L0THR, 0, cgnn_stream(stream)>>>(*t, X0, F0, center);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_l0_fwd(const cgnn_tiles* t, const float* X0, int32_t F0, const float* W0,
                    const float* bias, float* P0, float* Y, double* stat_slab, int64_t stat_slab_bytes, const float* center,
                    float* w_eff, float* mean_offset, const cgnn_bn_tail* tail, void* stream) {
  if (!t || F0 <= 0 || F0 > FP || t->max_tile_rows > CGNN_FUSED_MAX_ROWS) return t && F0 > FP ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (Y && (center || w_eff || mean_offset)) return CGNN_EINVAL;   // the centred form belongs to the factored layer
  if (center && (!w_eff || !mean_offset || F0 >= FP)) return CGNN_EINVAL;   // column 7 must be spare
  if (!center && (w_eff || mean_offset)) return CGNN_EINVAL;
  if (t->num_tiles == 0) return CGNN_OK;
  CGNN_NEED_BYTES(stat_slab, stat_slab_bytes, (int64_t)l0_grid(t->num_nodes) * 128 * (int64_t)sizeof(double));
  // (a tail finalises the FACTORED layer only: the statistics then come from the moments of P0)
  if (tail && (Y || !tail->acc || tail->mode != 0 || !(tail->count > 0.0) || !tail->gamma || !tail->beta ||
               !tail->running_mean || !tail->running_var || !tail->bn_out || tail->rng_n < 0 || tail->rng_n > 64 ||
               (tail->rng_n > 0 && !tail->rng_state)))
    return CGNN_EINVAL;
  const cgnn_bn_tail tl = tail ? *tail : cgnn_bn_tail{};
  if (!X0 || !W0 || !bias || !P0 || !t->tile_ptr || !t->tile_blk || !t->blk_off_dst ||
      !t->ent_dst || !t->dis)
    return CGNN_EINVAL;
  if (Y)
    k_l0_fwd<true><<<l0_grid(t->num_nodes), L0THR, 0, cgnn_stream(stream)>>>(*t, X0, F0, W0, bias, P0, Y, stat_slab,
                                                                             nullptr, nullptr, nullptr, cgnn_bn_tail{});
  else
    k_l0_fwd<false><<<l0_grid(t->num_nodes), L0THR, 0, cgnn_stream(stream)>>>(*t, X0, F0, W0, bias, P0, Y, stat_slab,
                                                                              center, w_eff, mean_offset, tl);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_l0_bwd(const float* dZ, const float* Y, const cgnn_l0src* l0, const float* bn,
                    const float* bwc, const float* P0, int64_t num_nodes, float* dW_slab, int64_t dW_slab_bytes,
                    double* db_slab, int64_t db_slab_bytes, const float* center, void* stream) {
  if (num_nodes < 0 || !dZ || !bn || !bwc || !P0 || !dW_slab || !db_slab) return CGNN_EINVAL;
  if (!Y && !(l0 && l0->W0 && l0->b0 && l0->F0 >= 1 && l0->F0 <= FP)) return CGNN_EINVAL;
  if (center && (Y || l0->F0 < 2)) return CGNN_EINVAL;
  CGNN_NEED_BYTES(dW_slab, dW_slab_bytes, (int64_t)l0_grid(num_nodes) * HID * FP * (int64_t)sizeof(float));
  CGNN_NEED_BYTES(db_slab, db_slab_bytes, (int64_t)l0_grid(num_nodes) * HID * (int64_t)sizeof(double));
  if (Y)
    k_l0_bwd<false><<<l0_grid(num_nodes), L0BTHR, 0, cgnn_stream(stream)>>>(dZ, Y, cgnn_l0src{}, bn, bwc, P0,
                                                                   num_nodes, dW_slab, db_slab, nullptr);
  else
    k_l0_bwd<true><<<l0_grid(num_nodes), L0BTHR, 0, cgnn_stream(stream)>>>(dZ, nullptr, *l0, bn, bwc, P0,
                                                                  num_nodes, dW_slab, db_slab, center);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
