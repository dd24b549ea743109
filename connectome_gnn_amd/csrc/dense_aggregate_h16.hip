// dense_aggregate_h16.hip -- the aggregation of dense parcellations on the fp16 matrix cores
// (BASELINE config 5: 1000-ROI graphs at 10 % density, hidden 256, fp16 storage / fp32 accumulate).
//
//   Y_g = M_g X_g          M_g[d][s] = sum of the normalised coefficients of the edges s -> d
//                          (+ the self-loop coefficient on the diagonal), one [P x P] half matrix
//                          per graph, built once per batch by cgnn_dense_adj_f16 and stored in
//                          MFMA-fragment-major order (opaque to the caller)
//
// Same reduction as models.py:112-114 (GCN: c_e = dis[s] w_e dis[d]) / :146-149 (SAGE:
// w_e / (wsum[d] + 1e-8)) and, from the source-sorted CSR, their autograd transposes.  At ~100
// neighbours per node the per-edge forms (aggregate.hip, aggregate_tiled_h16.hip) are bound by
// VALU/LDS work per edge (8 % of the HBM roofline); as a dense product the 10x redundant flops are
// free on the matrix pipe (v_mfma_f32_32x32x16_f16, fp32 accumulate) and the structure costs
// 2 B/entry of M instead of 8 B/edge -- 2.5x the bytes, read at streaming rate.
//
// One persistent workgroup per (graph, 64-column slice): the slice of X is transposed into LDS
// ([64 cols][k], so a lane's B fragment = 8 consecutive k = one ds_read_b128); every wave owns
// 32-row blocks of M and streams them HBM/L2 -> registers as A fragments (1 KB contiguous per
// step thanks to the fragment-major layout, 16 steps ahead).  The 4 slices of a graph are placed on the same XCD so that M_g is
// read from HBM once and re-read from that XCD's L2.
#include <hip/hip_fp16.h>
#include "common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CGNN_D_NW 12
#define CGNN_D_AHEAD 16
constexpr int D_NW = CGNN_D_NW;
constexpr int D_THR = D_NW * 64;
constexpr int D_MAXP = 1024;
constexpr int D_KPAD = 24;                      // halves of padding per transposed row (>= 16: the
                                                // B prefetch reads one step past the last)
constexpr int D_AHEAD = CGNN_D_AHEAD;                    // A fragments in flight per wave

// row pitch of the transposed slice: k is padded to whole pipeline rounds (zero-filled)
__host__ __device__ inline int d_kp(int P) {
  return (P + 16 * D_AHEAD - 1) / (16 * D_AHEAD) * (16 * D_AHEAD) + D_KPAD;
}

// ------------------------------------------------------------------ builder: CSR -> dense half
// one block per (graph, row): accumulate the row in LDS (duplicate edges add up), write halves
__global__ void __launch_bounds__(256) k_dense_adj(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ coef, const float* __restrict__ selfc,
    const int32_t* __restrict__ gptr, int P, __half* __restrict__ M) {
  __shared__ float row[D_MAXP];
  const int g = blockIdx.y, d = blockIdx.x;
  const int base = gptr[g], n = gptr[g + 1] - base;
  for (int i = threadIdx.x; i < P; i += 256) row[i] = 0.f;
  __syncthreads();
  if (d < n) {
    const int r = base + d;
    if (threadIdx.x == 0) {                    // serial, in CSR (= COO) order: bit-reproducible
      for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
        const int s = col[e] - base;
        if (s >= 0 && s < n) row[s] += coef[e];
      }
      if (selfc) row[d] += selfc[r];
    }
  }
  __syncthreads();
  // fragment-major layout: [row block d/32][step k/16][lane = d%32 + 32*((k/8)%2)][k%8], i.e. the
  // A operand of one v_mfma_f32_32x32x16_f16 is 1 KB contiguous (see k_dense_agg)
  __half* out = M + (int64_t)g * P * P;
  const int S = P >> 4;
  for (int k8 = threadIdx.x; k8 < (P >> 3); k8 += 256) {          // 8 halves = 16 bytes at a time
    const int k = 8 * k8;
    const int lane = (d & 31) + 32 * (k8 & 1);
    __half2 h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = __floats2half2_rn(row[k + 2 * i], row[k + 2 * i + 1]);
    *reinterpret_cast<uint4*>(out + ((((int64_t)(d >> 5) * S + (k >> 4)) * 64 + lane) << 3)) =
        *reinterpret_cast<const uint4*>(h);
  }
}

// ------------------------------------------------------------------------------ Y_g = M_g X_g
__global__ void __launch_bounds__(D_THR) k_dense_agg(
    const __half* __restrict__ M, int P, const int32_t* __restrict__ gptr, int B,
    const __half* __restrict__ X, int64_t ldx, int nslices, const float* __restrict__ bias,
    __half* __restrict__ Y, int64_t ldy, double* __restrict__ stat_slab) {
  extern __shared__ __attribute__((aligned(16))) __half Xt[];       // [64][KP] (+ statistics scratch)
  const int KP = d_kp(P);
  // BatchNorm statistics of the (half-rounded) output, optional: per-wave partials -> red, the
  // workgroup's running column sums -> wacc [2][64 * nslices], one slab row per workgroup at the end
  double* red = reinterpret_cast<double*>(Xt + 64 * KP);             // [D_NW][128]
  double* wacc = red + D_NW * 128;                                   // [2][64 * nslices]
  if (stat_slab)
    for (int i = threadIdx.x; i < 128 * nslices; i += D_THR) wacc[i] = 0.0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int units = B * nslices;
  // unit order: the slices of one graph sit on workgroups of the same XCD (blockIdx % 8)
  const bool xcd_map = (gridDim.x % 8 == 0) && ((gridDim.x / 8) % nslices == 0);
  for (int it = 0;; ++it) {
    int u;
    if (xcd_map) {
      const int per_xcd = gridDim.x / 8;                           // workgroups per XCD
      const int x = blockIdx.x % 8, i = blockIdx.x / 8;
      const int g = it * (gridDim.x / nslices) + x * (per_xcd / nslices) + i / nslices;
      u = g * nslices + i % nslices;
      if (it * (int)gridDim.x >= units) break;
      if (g >= B) continue;
    } else {
      u = it * gridDim.x + blockIdx.x;
      if (u >= units) break;
    }
    const int g = u / nslices, slice = u - g * nslices;
    const int base = gptr[g], n = gptr[g + 1] - base;
    const int ksteps = (n + 15) >> 4;

    // ---- transpose the [n x 64] slice of X into LDS; k in [n, kfill) is zero-filled
    const int kfill = (ksteps + D_AHEAD - 1) / D_AHEAD * D_AHEAD * 16;
    __syncthreads();
    {
      const int piece = threadIdx.x & 7;
      constexpr int SU = 8;                         // rows in flight per thread (one latency per 8)
      for (int k0 = threadIdx.x >> 3; k0 < kfill; k0 += SU * (D_THR / 8)) {
        uint4 v[SU];
#pragma unroll
        for (int q = 0; q < SU; ++q) {
          const int k = k0 + q * (D_THR / 8);
          v[q] = make_uint4(0u, 0u, 0u, 0u);
          if (k < n) v[q] = *reinterpret_cast<const uint4*>(X + (int64_t)(base + k) * ldx + 64 * slice + 8 * piece);
        }
#pragma unroll
        for (int q = 0; q < SU; ++q) {
          const int k = k0 + q * (D_THR / 8);
          if (k < kfill) {
            const __half* hv = reinterpret_cast<const __half*>(&v[q]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {          // column c = 8*piece + i lives in LDS row c/2 + 32*(c%2)
              const int c = 8 * piece + i;
              Xt[((c >> 1) + 32 * (c & 1)) * KP + k] = hv[i];
            }
          }
        }
      }
    }
    __syncthreads();

    const __half* Mg = M + (int64_t)g * P * P;
    double st1[2] = {0.0, 0.0}, st2[2] = {0.0, 0.0};
    for (int rb = wave; 32 * rb < n; rb += D_NW) {
      // rows >= n of M are zero
      // M is stored fragment-major: the A operand of (row block rb, step s) is 1 KB contiguous,
      // lane l's 16 bytes at offset 16*l -> every wave load is 8 full 128-byte lines
      const __half* arow = Mg + ((int64_t)rb * (P >> 4) * 64 + lane) * 8;
      // output tile t holds the columns {2*r + t}: a lane ends up with 2 adjacent columns of each
      // of its rows = one 4-byte store
      const __half* b0 = Xt + r * KP + 8 * h;       // column 2r   (LDS row c/2 + 32*(c%2))
      const __half* b1 = b0 + 32 * KP;              // column 2r+1
      f32x16 acc0 = {0}, acc1 = {0};
      // Branch-free software pipeline: D_AHEAD A fragments in flight from HBM/L2, B fragments one
      // step ahead from LDS.  Steps past `ksteps` re-read the last A fragment (finite values)
      // against the zero-filled tail of Xt, so the steady state is one basic block and the
      // loads' wait counts stay exact.
      h8 a[D_AHEAD];
      const int last = ksteps - 1;
#pragma unroll
      for (int p = 0; p < D_AHEAD; ++p) a[p] = *reinterpret_cast<const h8*>(arow + 512 * min(p, last));
      h8 bc0 = *reinterpret_cast<const h8*>(b0), bc1 = *reinterpret_cast<const h8*>(b1);
      for (int kk = 0; kk < ksteps; kk += D_AHEAD) {
#pragma unroll
        for (int p = 0; p < D_AHEAD; ++p) {
          const h8 av = a[p];
          a[p] = *reinterpret_cast<const h8*>(arow + 512 * min(kk + p + D_AHEAD, last));
          // next step's B (the row tail up to KP is zero-filled, so kk + p + 1 is always readable)
          const h8 bn0 = *reinterpret_cast<const h8*>(b0 + 16 * (kk + p + 1));
          const h8 bn1 = *reinterpret_cast<const h8*>(b1 + 16 * (kk + p + 1));
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bc0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bc1, acc1, 0, 0, 0);
          bc0 = bn0; bc1 = bn1;
          // pin the order: without this the scheduler sinks every load next to its use (a
          // 2-deep pipeline, one exposed memory round trip per step)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      float2 bia = make_float2(0.f, 0.f);
      if (bias) bia = *reinterpret_cast<const float2*>(bias + 64 * slice + 2 * r);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 32 * rb + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (row < n) {
          const __half2 hv = __floats2half2_rn(acc0[q] + bia.x, acc1[q] + bia.y);
          *reinterpret_cast<__half2*>(Y + (int64_t)(base + row) * ldy + 64 * slice + 2 * r) = hv;
          if (stat_slab) {
            const float2 fv = __half22float2(hv);
            st1[0] += fv.x; st1[1] += fv.y;
            st2[0] += (double)fv.x * fv.x; st2[1] += (double)fv.y * fv.y;
          }
        }
      }
    }
    if (stat_slab) {
      // fold the two row halves, the waves (fixed order), then add to the workgroup's sums
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        st1[t] += __shfl_xor(st1[t], 32, 64);
        st2[t] += __shfl_xor(st2[t], 32, 64);
        if (h == 0) { red[wave * 128 + 2 * r + t] = st1[t]; red[wave * 128 + 64 + 2 * r + t] = st2[t]; }
      }
      __syncthreads();
      if (threadIdx.x < 128) {
        double tot = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < D_NW; ++w2) tot += red[w2 * 128 + threadIdx.x];
        wacc[(threadIdx.x >> 6) * 64 * nslices + 64 * slice + (threadIdx.x & 63)] += tot;
      }
    }
  }
  if (stat_slab) {
    __syncthreads();
    for (int i = threadIdx.x; i < 128 * nslices; i += D_THR) stat_slab[(int64_t)blockIdx.x * 128 * nslices + i] = wacc[i];
  }
}

}  // namespace

extern "C" {

int cgnn_dense_adj_f16(const int32_t* rowptr, const int32_t* col, const float* coef,
                       const float* selfc, const int32_t* gptr, int32_t num_graphs, int32_t P,
                       void* M, void* stream) {
  if (num_graphs < 0 || P <= 0 || P > D_MAXP || P % 64) return P > D_MAXP ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (num_graphs == 0) return CGNN_OK;
  if (num_graphs > 65535) return CGNN_EUNSUPPORTED;          // grid.y of the builder
  if (!rowptr || !col || !coef || !gptr || !M) return CGNN_EINVAL;
  k_dense_adj<<<dim3((unsigned)P, (unsigned)num_graphs), 256, 0, cgnn_stream(stream)>>>(
      rowptr, col, coef, selfc, gptr, P, static_cast<__half*>(M));
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_dense_aggregate_f16(const void* M, int32_t P, const int32_t* gptr, int32_t num_graphs,
                             const void* X, int64_t ldx, int32_t F, const float* bias, void* Y,
                             int64_t ldy, double* stat_slab, int64_t stat_slab_bytes, void* stream) {
  if (num_graphs < 0 || P <= 0 || F <= 0 || ldx < F || ldy < F) return CGNN_EINVAL;
  if (P > D_MAXP || P % 64 || F % 64 || ldx % 8) return CGNN_EUNSUPPORTED;
  if ((size_t)64 * d_kp(P) * sizeof(__half) + (size_t)(D_NW * 128 + 2 * F) * sizeof(double) > 160 * 1024)
    return CGNN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(M)) & 15) return CGNN_EUNSUPPORTED;
  if (num_graphs == 0) return CGNN_OK;
  if (!M || !gptr || !X || !Y) return CGNN_EINVAL;
  CGNN_NEED_BYTES(stat_slab, stat_slab_bytes, (int64_t)cgnn_fused_grid() * 2 * F * (int64_t)sizeof(double));
  const size_t lds = (size_t)64 * d_kp(P) * sizeof(__half) + (size_t)(D_NW * 128 + 2 * F) * sizeof(double);
  static bool attr_set_dev[CGNN_MAX_DEVICES] = {};
  bool& attr_set = attr_set_dev[cgnn_device_ordinal()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dense_agg),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return CGNN_ELAUNCH;
    attr_set = true;
  }
  k_dense_agg<<<cgnn_fused_grid(), D_THR, lds, cgnn_stream(stream)>>>(
      static_cast<const __half*>(M), P, gptr, num_graphs, static_cast<const __half*>(X), ldx, F / 64,
      bias, static_cast<__half*>(Y), ldy, stat_slab);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
