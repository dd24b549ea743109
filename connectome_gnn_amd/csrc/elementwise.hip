// elementwise.hip -- BatchNorm1d (+ReLU) + dropout for the layered path, any power-of-two width.
//
// Replaces, per layer, torch's batch_norm (3 forward + 3 backward kernels), relu, dropout and
// their mask bookkeeping (models.py:208-210 / :260-261) by two streaming passes forward
// (column statistics; normalise+activate+drop) and two backward (BatchNorm-backward sums;
// apply).  Statistics are accumulated in fp64 across threads/blocks exactly like the fused
// path; dropout keep-bits are one byte per 4-column chunk.
//
//   X' = drop( act( a*Y + b ) ),  a = gamma*invstd, b = beta - mean*a,  act = ReLU or identity
//   dY = a * ( dZ - c1 - xhat*c2 ),  dZ = dX' * drop' * act'
#include "common.h"
#include "drop_ew.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
// the activation arrays are float or IEEE half (fp16 storage, fp32 arithmetic: BASELINE config 5);
// four consecutive elements at a time, 16 or 8 bytes
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const _Float16* p) {
  const h4 v = *reinterpret_cast<const h4*>(p);
  return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}
__device__ __forceinline__ void st4(_Float16* p, const float4& v) {
  h4 h;
  h.x = (_Float16)v.x; h.y = (_Float16)v.y; h.z = (_Float16)v.z; h.w = (_Float16)v.w;
  *reinterpret_cast<h4*>(p) = h;
}

// streamed activation arrays (common.h: non-temporal); knobs for the A/B builds
using ::ldnt4;
using ::stnt4;
__device__ __forceinline__ float4 ldnt4(const _Float16* p) {
  const h4 v = __builtin_nontemporal_load(reinterpret_cast<const h4*>(p));
  return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}
__device__ __forceinline__ void stnt4(_Float16* p, const float4& v) {
  h4 h;
  h.x = (_Float16)v.x; h.y = (_Float16)v.y; h.z = (_Float16)v.z; h.w = (_Float16)v.w;
  __builtin_nontemporal_store(h, reinterpret_cast<h4*>(p));
}
// (measured on cfg3: non-temporal loads in the apply passes -4 us per launch; in the statistics
// passes they take the array away from the apply pass that re-reads it, +4 us there)
#define EW_LDS ld4
#define EW_LD ldnt4
#define EW_ST st4

// Readout gradient in place of a stored dX' (models.py:57-59 backward): the row's gradient is
// dP[graph] / (n_graph + 1e-8), rebuilt on the fly instead of written out and read back.
struct PoolGrad {
  const float* dP;               // [B, N] or NULL (= read dX')
  const int32_t* node_graph;     // [M]
  const int32_t* gptr;           // [B+1]
};

__device__ __forceinline__ float4 pool_grad(const PoolGrad& pg, int64_t r, int N, int c) {
  const int g = pg.node_graph[r];
  const float inv = 1.0f / ((float)(pg.gptr[g + 1] - pg.gptr[g]) + 1e-8f);
  const float4 v = ld4(pg.dP + (int64_t)g * N + 4 * c);
  return make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
}

constexpr int ROWS = 256;     // most rows per block of the reduction kernels (fewer for short arrays)
constexpr int THR = 256;

// per-thread fp32 partials over <= ROWS*nch/THR rows, fp64 across threads -> slab[block][2N]
template <bool BWD, typename T>
__global__ void __launch_bounds__(THR) k_colstats(
    const T* __restrict__ A /* Y (fwd) or dX' (bwd) */, const T* __restrict__ Y,
    const uint8_t* __restrict__ mask, const float* __restrict__ coef, int relu, DropCfg drop,
    int use_drop, int64_t M, int N, double* __restrict__ slab, PoolGrad pg, int rows_per_block) {
  extern __shared__ double red[];                    // [rpp][2N]
  const int nch = N >> 2;
  const int c = threadIdx.x % nch, rr = threadIdx.x / nch, rpp = THR / nch;
  const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block, rend = min(M, rbeg + rows_per_block);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  double q2[4] = {0.0, 0.0, 0.0, 0.0};    // forward: sum y^2 in fp64 (it cancels against mean^2 downstream)
  float4 ca = s1, cb = s1, cm = s1, ci = s1;
  if (BWD) {
    ca = ld4(coef + 4 * c); cb = ld4(coef + N + 4 * c);
    cm = ld4(coef + 2 * N + 4 * c); ci = ld4(coef + 3 * N + 4 * c);
  }
  if (rr < rpp) {
    for (int64_t r = rbeg + rr; r < rend; r += rpp) {
      const float4 a = (BWD && pg.dP) ? pool_grad(pg, r, N, c) : EW_LDS(A + r * N + 4 * c);
      if (!BWD) {
        s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
        q2[0] += (double)a.x * a.x; q2[1] += (double)a.y * a.y;
        q2[2] += (double)a.z * a.z; q2[3] += (double)a.w * a.w;
      } else {
        const float4 y = EW_LDS(Y + r * N + 4 * c);
        const uint32_t kb = use_drop ? mask[r * nch + c] : 0xFu;
        const float zx = fmaf(ca.x, y.x, cb.x), zy = fmaf(ca.y, y.y, cb.y);
        const float zz = fmaf(ca.z, y.z, cb.z), zw = fmaf(ca.w, y.w, cb.w);
        const float fx = ((!relu || zx > 0.f) && (kb & 1u)) ? drop.scale : 0.f;
        const float fy = ((!relu || zy > 0.f) && (kb & 2u)) ? drop.scale : 0.f;
        const float fz = ((!relu || zz > 0.f) && (kb & 4u)) ? drop.scale : 0.f;
        const float fw = ((!relu || zw > 0.f) && (kb & 8u)) ? drop.scale : 0.f;
        const float dx = a.x * fx, dy = a.y * fy, dz = a.z * fz, dw = a.w * fw;
        s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
        s2.x = fmaf(dx, (y.x - cm.x) * ci.x, s2.x); s2.y = fmaf(dy, (y.y - cm.y) * ci.y, s2.y);
        s2.z = fmaf(dz, (y.z - cm.z) * ci.z, s2.z); s2.w = fmaf(dw, (y.w - cm.w) * ci.w, s2.w);
      }
    }
  }
  if (rr < rpp) {
    double* p = red + (int64_t)rr * 2 * N;
    p[4 * c + 0] = s1.x; p[4 * c + 1] = s1.y; p[4 * c + 2] = s1.z; p[4 * c + 3] = s1.w;
    if (BWD) {
      p[N + 4 * c + 0] = s2.x; p[N + 4 * c + 1] = s2.y; p[N + 4 * c + 2] = s2.z; p[N + 4 * c + 3] = s2.w;
    } else {
      p[N + 4 * c + 0] = q2[0]; p[N + 4 * c + 1] = q2[1]; p[N + 4 * c + 2] = q2[2]; p[N + 4 * c + 3] = q2[3];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * N; e += THR) {
    double t = 0.0;
    for (int q = 0; q < rpp; ++q) t += red[(int64_t)q * 2 * N + e];
    slab[(int64_t)blockIdx.x * 2 * N + e] = t;
  }
}

__device__ __forceinline__ double block_sum256(double v, double* sh) {
  v = cgnn_wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const double t = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return t;
}

// Column sums of a slab [rows][2N] fp64 for FIN_C consecutive channels per block: thread (rg, cc)
// adds rows rg, rg + FIN_G, ... of columns c and N + c (each load instruction reads FIN_C
// consecutive doubles of a row: coalesced; one block per channel read 4 KB-strided columns), then
// the FIN_G partials are combined in fixed order.  Results in s1[cc], s2[cc] for threads rg == 0.
constexpr int FIN_C = 8, FIN_G = 128;      // 1024 threads: 64-byte row pieces, 128 row groups
__device__ __forceinline__ void slab_colsum2(const double* __restrict__ slab, int rows, int N, int c0,
                                             double (*sh)[FIN_C][2], double& S1, double& S2) {
  const int cc = threadIdx.x % FIN_C, rg = threadIdx.x / FIN_C, c = c0 + cc;
  double a1 = 0.0, a2 = 0.0;
  if (c < N) {
    int r = rg;
    for (; r + 3 * FIN_G < rows; r += 4 * FIN_G) {        // 8 independent loads in flight
      double v1[4], v2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v1[u] = slab[(int64_t)(r + u * FIN_G) * 2 * N + c];
        v2[u] = slab[(int64_t)(r + u * FIN_G) * 2 * N + N + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a1 += v1[u]; a2 += v2[u]; }
    }
    for (; r < rows; r += FIN_G) {
      a1 += slab[(int64_t)r * 2 * N + c];
      a2 += slab[(int64_t)r * 2 * N + N + c];
    }
  }
  sh[rg][cc][0] = a1;
  sh[rg][cc][1] = a2;
  __syncthreads();
  // fixed-order two-level fold (a serial chain of FIN_G dependent LDS reads costs ~3 us)
  __shared__ double sh2[FIN_G / 8][FIN_C][2];
  if (rg < FIN_G / 8) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { t1 += sh[8 * rg + k][cc][0]; t2 += sh[8 * rg + k][cc][1]; }
    sh2[rg][cc][0] = t1;
    sh2[rg][cc][1] = t2;
  }
  __syncthreads();
  S1 = S2 = 0.0;
  if (rg == 0) {
#pragma unroll
    for (int k = 0; k < FIN_G / 8; ++k) { S1 += sh2[k][cc][0]; S2 += sh2[k][cc][1]; }
  }
}

__global__ void __launch_bounds__(FIN_C * FIN_G) k_bn_fwd_finalize_n(
    const double* __restrict__ slab, int rows, int N, double count_host,
    const double* __restrict__ count_dev, int training,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
    float* __restrict__ rvar, float momentum, float eps, long long* __restrict__ tracked,
    float* __restrict__ coef) {
  __shared__ double sh[FIN_G][FIN_C][2];
  const double count = count_dev ? count_dev[0] : count_host;
  const int c = blockIdx.x * FIN_C + threadIdx.x % FIN_C;
  const bool owner = threadIdx.x < FIN_C && c < N;
  float mean = 0.f, var = 1.f;
  if (training) {
    double S1, S2;
    slab_colsum2(slab, rows, N, blockIdx.x * FIN_C, sh, S1, S2);
    if (owner) {
      const double m = S1 / count;
      double v = S2 / count - m * m;
      if (v < 0.0) v = 0.0;
      mean = (float)m;
      var = (float)v;
      const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
      rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unbiased;
      if (c == 0 && tracked) *tracked += 1;
    }
  } else if (owner) {
    mean = rmean[c];
    var = rvar[c];
  }
  if (owner) {
    const float invstd = 1.0f / sqrtf(var + eps);
    const float a = gamma[c] * invstd;
    coef[c] = a;
    coef[N + c] = beta[c] - mean * a;
    coef[2 * N + c] = mean;
    coef[3 * N + c] = invstd;
  }
}

__global__ void __launch_bounds__(FIN_C * FIN_G) k_bn_bwd_finalize_n(const double* __restrict__ slab, int rows,
                                                           int N, double count_host,
                                                           const double* __restrict__ count_dev,
                                                           int zero_coef, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta,
                                                           float* __restrict__ bwc) {
  __shared__ double sh[FIN_G][FIN_C][2];
  const double count = count_dev ? count_dev[0] : count_host;
  const int c = blockIdx.x * FIN_C + threadIdx.x % FIN_C;
  double S1, S2;
  slab_colsum2(slab, rows, N, blockIdx.x * FIN_C, sh, S1, S2);
  if (threadIdx.x < FIN_C && c < N) {
    dbeta[c] = (float)S1;
    dgamma[c] = (float)S2;
    bwc[c] = zero_coef ? 0.f : (float)(S1 / count);
    bwc[N + c] = zero_coef ? 0.f : (float)(S2 / count);
  }
}

// BatchNorm-backward sums of a pooled last layer from the forward pass's factor sums (see
// k_bn_act_pool_fwd): S1[c] = sum_g dP[g][c] / (n_g + 1e-8) * F1[g][c], S2 likewise with F2; then
// exactly k_bn_bwd_finalize_n.  fp64 accumulation over the graphs, fixed order.
__global__ void __launch_bounds__(FIN_C * FIN_G) k_bn_pool_bwd_finalize(
    const float* __restrict__ dP, const float* __restrict__ Fsum, const int32_t* __restrict__ gptr, int B,
    int N, double count, int zero_coef, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ bwc) {
  __shared__ double sh[FIN_G][FIN_C][2];
  __shared__ double sh2[FIN_G / 8][FIN_C][2];
  const int cc = threadIdx.x % FIN_C, rg = threadIdx.x / FIN_C, c = blockIdx.x * FIN_C + cc;
  double a1 = 0.0, a2 = 0.0;
  if (c < N)
    for (int g = rg; g < B; g += FIN_G) {
      const float inv = 1.0f / ((float)(gptr[g + 1] - gptr[g]) + 1e-8f);
      const float d = dP[(int64_t)g * N + c] * inv;
      a1 += (double)(d * Fsum[(int64_t)g * N + c]);
      a2 += (double)(d * Fsum[(int64_t)(B + g) * N + c]);
    }
  sh[rg][cc][0] = a1;
  sh[rg][cc][1] = a2;
  __syncthreads();
  if (rg < FIN_G / 8) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { t1 += sh[8 * rg + k][cc][0]; t2 += sh[8 * rg + k][cc][1]; }
    sh2[rg][cc][0] = t1;
    sh2[rg][cc][1] = t2;
  }
  __syncthreads();
  if (rg == 0 && c < N) {
    double S1 = 0.0, S2 = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_G / 8; ++k) { S1 += sh2[k][cc][0]; S2 += sh2[k][cc][1]; }
    dbeta[c] = (float)S1;
    dgamma[c] = (float)S2;
    bwc[c] = zero_coef ? 0.f : (float)(S1 / count);
    bwc[N + c] = zero_coef ? 0.f : (float)(S2 / count);
  }
}

// forward apply (BWD=false): X' = drop(act(a*Y+b)), keep bytes out
// backward apply (BWD=true): dY = a*(dZ - c1 - xhat*c2), dZ = dX'*drop'*act'
template <bool BWD, typename T>
__global__ void __launch_bounds__(256) k_bn_act_apply(
    const T* __restrict__ Y, const T* __restrict__ dXp, const float* __restrict__ coef,
    const float* __restrict__ bwc, int relu, DropCfg drop, int use_drop,
    uint8_t* __restrict__ mask_out, const uint8_t* __restrict__ mask_in, T* __restrict__ out,
    int64_t M, int N, int relu_in, double* __restrict__ colsum_slab, PoolGrad pg) {
  if (drop.dev_key) drop.key1 ^= drop.dev_key[0];
  const int nch = N >> 2;
  const int64_t total = M * nch;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);   // column sums of dY: this thread's chunk is fixed
                                                 // (gridDim*blockDim is a multiple of nch)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % nch);
    const float4 y = EW_LD(Y + 4 * i);
    const float4 ca = ld4(coef + 4 * c), cb = ld4(coef + N + 4 * c);
    const float zx = fmaf(ca.x, y.x, cb.x), zy = fmaf(ca.y, y.y, cb.y);
    const float zz = fmaf(ca.z, y.z, cb.z), zw = fmaf(ca.w, y.w, cb.w);
    uint32_t kb = 0xFu;
    if (use_drop) {
      if (BWD) kb = mask_in[i];
      else {
        kb = drop_bits(drop, (uint32_t)i);
        if (mask_out) mask_out[i] = (uint8_t)kb;
      }
    }
    const float fx = ((!relu || zx > 0.f) && (kb & 1u)) ? drop.scale : 0.f;
    const float fy = ((!relu || zy > 0.f) && (kb & 2u)) ? drop.scale : 0.f;
    const float fz = ((!relu || zz > 0.f) && (kb & 4u)) ? drop.scale : 0.f;
    const float fw = ((!relu || zw > 0.f) && (kb & 8u)) ? drop.scale : 0.f;
    if (!BWD) {
      EW_ST(out + 4 * i, make_float4(zx * fx, zy * fy, zz * fz, zw * fw));
    } else {
      const float4 g = pg.dP ? pool_grad(pg, i / nch, N, c) : EW_LD(dXp + 4 * i);
      const float4 cm = ld4(coef + 2 * N + 4 * c), ci = ld4(coef + 3 * N + 4 * c);
      const float4 c1 = ld4(bwc + 4 * c), c2 = ld4(bwc + N + 4 * c);
      float4 d = make_float4(bn_bwd_dy(ca.x, g.x, fx, c1.x, y.x, cm.x, ci.x, c2.x),
                             bn_bwd_dy(ca.y, g.y, fy, c1.y, y.y, cm.y, ci.y, c2.y),
                             bn_bwd_dy(ca.z, g.z, fz, c1.z, y.z, cm.z, ci.z, c2.z),
                             bn_bwd_dy(ca.w, g.w, fw, c1.w, y.w, cm.w, ci.w, c2.w));
      if (relu_in) {
        d.x = y.x > 0.f ? d.x : 0.f; d.y = y.y > 0.f ? d.y : 0.f;
        d.z = y.z > 0.f ? d.z : 0.f; d.w = y.w > 0.f ? d.w : 0.f;
      }
      EW_ST(out + 4 * i, d);
      cs.x += d.x; cs.y += d.y; cs.z += d.z; cs.w += d.w;
    }
  }
  if (BWD && colsum_slab) {
    __shared__ double red[256 * 4];
    red[4 * threadIdx.x + 0] = cs.x; red[4 * threadIdx.x + 1] = cs.y;
    red[4 * threadIdx.x + 2] = cs.z; red[4 * threadIdx.x + 3] = cs.w;
    __syncthreads();
    // thread tid owns chunk tid % nch (nch is a power of two <= 256, so it divides the strides)
    for (int e = threadIdx.x; e < N; e += 256) {
      double t = 0.0;
      for (int k = e >> 2; k < 256; k += nch) t += red[4 * k + (e & 3)];
      colsum_slab[(int64_t)blockIdx.x * N + e] = t;
    }
  }
}

// Readout fused with the last layer's BatchNorm (+act) + dropout: P[g] = mean_rows drop(act(a*Y+b)).
// One 1024-thread block per graph at a time; X' itself is never written.
constexpr int PTHR = 1024;

// grid = num_graphs * CS workgroups (capped): workgroup (g, cs) pools columns [cs, cs + 1) * N / CS of
// graph g -- with few large graphs (64 x 1000-ROI) one workgroup per graph leaves most CUs idle
// FSUM: also the per-graph factor sums F1[g][c] = sum_rows f, F2[g][c] = sum_rows f * xhat with
// f = act' * keep / (1-p): the readout's gradient is constant per graph, so the BatchNorm-backward
// sums of this layer are sum_g dP[g]/n_g * F1[g] and ... * F2[g] (k_bn_pool_bwd_finalize) -- the
// backward statistics pass over Y is not needed.  Fsum = [2][B][N].
template <typename T, bool FSUM>
__global__ void __launch_bounds__(PTHR) k_bn_act_pool_fwd(
    const T* __restrict__ Y, const float* __restrict__ coef, int relu, DropCfg drop, int use_drop,
    uint8_t* __restrict__ mask_out, const int32_t* __restrict__ gptr, int B, float* __restrict__ P,
    int N, int CS, float* __restrict__ Fsum) {
  if (drop.dev_key) drop.key1 ^= drop.dev_key[0];
  extern __shared__ float pred[];                    // [rpp][N / CS] (x 3 with factor sums)
  const int nch = N >> 2, nchb = nch / CS, NB = N / CS;
  const int cl = threadIdx.x % nchb, rr = threadIdx.x / nchb, rpp = PTHR / nchb;
  for (int u = blockIdx.x; u < B * CS; u += gridDim.x) {
    const int g = u / CS, c = (u - g * CS) * nchb + cl;
    const float4 ca = ld4(coef + 4 * c), cb = ld4(coef + N + 4 * c);
    float4 cm = make_float4(0.f, 0.f, 0.f, 0.f), ci = cm;
    if (FSUM) { cm = ld4(coef + 2 * N + 4 * c); ci = ld4(coef + 3 * N + 4 * c); }
    const int rbeg = gptr[g], rend = gptr[g + 1];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), f1 = s, f2 = s;
    constexpr int U = 8;                             // rows in flight per thread
    for (int row0 = rbeg + rr; row0 < rend; row0 += U * rpp) {
      float4 yb[U];
#pragma unroll
      for (int u2 = 0; u2 < U; ++u2) {
        const int row = row0 + u2 * rpp;
        yb[u2] = row < rend ? ldnt4(Y + ((int64_t)row * nch + c) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u2 = 0; u2 < U; ++u2) {
        const int row = row0 + u2 * rpp;
        if (row < rend) {
          const int64_t i = (int64_t)row * nch + c;
          const float zx = fmaf(ca.x, yb[u2].x, cb.x), zy = fmaf(ca.y, yb[u2].y, cb.y);
          const float zz = fmaf(ca.z, yb[u2].z, cb.z), zw = fmaf(ca.w, yb[u2].w, cb.w);
          uint32_t kb = 0xFu;
          if (use_drop) {
            kb = drop_bits(drop, (uint32_t)i);
            if (mask_out) mask_out[i] = (uint8_t)kb;
          }
          const float fx = ((!relu || zx > 0.f) && (kb & 1u)) ? drop.scale : 0.f;
          const float fy = ((!relu || zy > 0.f) && (kb & 2u)) ? drop.scale : 0.f;
          const float fz = ((!relu || zz > 0.f) && (kb & 4u)) ? drop.scale : 0.f;
          const float fw = ((!relu || zw > 0.f) && (kb & 8u)) ? drop.scale : 0.f;
          s.x += fx != 0.f ? zx * drop.scale : 0.f;
          s.y += fy != 0.f ? zy * drop.scale : 0.f;
          s.z += fz != 0.f ? zz * drop.scale : 0.f;
          s.w += fw != 0.f ? zw * drop.scale : 0.f;
          if (FSUM) {
            f1.x += fx; f1.y += fy; f1.z += fz; f1.w += fw;
            f2.x = fmaf(fx, (yb[u2].x - cm.x) * ci.x, f2.x); f2.y = fmaf(fy, (yb[u2].y - cm.y) * ci.y, f2.y);
            f2.z = fmaf(fz, (yb[u2].z - cm.z) * ci.z, f2.z); f2.w = fmaf(fw, (yb[u2].w - cm.w) * ci.w, f2.w);
          }
        }
      }
    }
    if (rr < rpp) {
      st4(pred + rr * NB + 4 * cl, s);
      if (FSUM) {
        st4(pred + (rpp + rr) * NB + 4 * cl, f1);
        st4(pred + (2 * rpp + rr) * NB + 4 * cl, f2);
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (FSUM ? 3 : 1) * NB; e += PTHR) {
      const int part = e / NB, ee = e - part * NB;
      float tot = 0.f;
      for (int k = 0; k < rpp; ++k) tot += pred[(part * rpp + k) * NB + ee];
      const int64_t o = (int64_t)g * N + (u - g * CS) * NB + ee;
      if (part == 0) P[o] = tot / ((float)(rend - rbeg) + 1e-8f);
      else Fsum[(int64_t)(part - 1) * B * N + o] = tot;
    }
    __syncthreads();
  }
}

bool width_ok(int N) { return N >= 4 && N <= 1024 && (N & (N - 1)) == 0; }
// rows per block: ROWS for long arrays, fewer (down to 32) when that would leave the chip with less
// than ~2048 blocks -- a 64000-row array at 256 rows per block is 250 blocks = one per CU, and the
// pass is then bound by the latency of each block's own loads
int stat_rows(int64_t M) {
  int r = ROWS;
#define EW_STAT_BLOCKS 1024
  while (r > 32 && (M + r - 1) / r < EW_STAT_BLOCKS) r >>= 1;
  return r;
}
int stat_blocks(int64_t M) { const int r = stat_rows(M); return (int)((M + r - 1) / r); }
unsigned apply_blocks(int64_t M, int N, bool colsum = false) {
  const int64_t total = M * (N >> 2);
  int64_t grid = (total + 255) / 256;
  // with column sums every block leaves a partial row in the slab: fewer, longer-running blocks
  const int64_t cap = colsum ? 1024 : 256 * 32;
  if (grid > cap) grid = cap;
  return (unsigned)(grid < 1 ? 1 : grid);
}

}  // namespace

namespace {

template <typename T>
int bn_act_fwd_stats_t(const T* Y, int64_t M, int32_t N, double* slab, int64_t slab_bytes, void* stream) {
  if (M < 0 || !width_ok(N) || !slab) return width_ok(N) ? CGNN_EINVAL : CGNN_EUNSUPPORTED;
  if (M == 0) return CGNN_OK;
  if (!Y) return CGNN_EINVAL;
  CGNN_NEED_BYTES(slab, slab_bytes, (int64_t)stat_blocks(M) * 2 * N * (int64_t)sizeof(double));
  DropCfg d{};
  const int rpp = THR / (N >> 2);
  k_colstats<false, T><<<stat_blocks(M), THR, (size_t)rpp * 2 * N * sizeof(double), cgnn_stream(stream)>>>(
      Y, (const T*)nullptr, nullptr, nullptr, 0, d, 0, M, N, slab, PoolGrad{nullptr, nullptr, nullptr}, stat_rows(M));
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}


template <typename T>
int bn_act_fwd_apply_t(const T* Y, const float* coef, int32_t relu, float p_drop, uint64_t seed,
                       const uint32_t* seed_dev, uint8_t* mask_out, T* X, int64_t M, int32_t N,
                       void* stream) {
  if (M < 0 || !width_ok(N) || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (M == 0) return CGNN_OK;
  if (!Y || !coef || !X) return CGNN_EINVAL;
  int use_drop;
  DropCfg d = make_drop(p_drop, seed, &use_drop);
  d.dev_key = seed_dev;
  k_bn_act_apply<false, T><<<apply_blocks(M, N), 256, 0, cgnn_stream(stream)>>>(
      Y, (const T*)nullptr, coef, nullptr, relu, d, use_drop, mask_out, nullptr, X, M, N, 0, nullptr,
      PoolGrad{nullptr, nullptr, nullptr});
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

template <typename T>
int bn_act_pool_fwd_t(const T* Y, const float* coef, int32_t relu, float p_drop, uint64_t seed,
                      const uint32_t* seed_dev, uint8_t* mask_out, const int32_t* gptr,
                      int32_t num_graphs, float* P, int32_t N, float* Fsum, void* stream) {
  if (num_graphs < 0 || !width_ok(N) || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (num_graphs == 0) return CGNN_OK;
  if (!Y || !coef || !gptr || !P) return CGNN_EINVAL;
  int use_drop;
  DropCfg d = make_drop(p_drop, seed, &use_drop);
  d.dev_key = seed_dev;
  // column splits: enough workgroups to fill the chip when the batch has few graphs
  int cs = 1;
  while (cs < 8 && num_graphs * cs < cgnn_fused_grid() && (N >> 2) / (2 * cs) >= 4) cs *= 2;
  const int rpp = PTHR / ((N >> 2) / cs);
  const int64_t units = (int64_t)num_graphs * cs;
  const unsigned grid = (unsigned)(units < 2048 ? units : 2048);
  if (Fsum)
    k_bn_act_pool_fwd<T, true><<<grid, PTHR, (size_t)3 * rpp * (N / cs) * sizeof(float), cgnn_stream(stream)>>>(
        Y, coef, relu, d, use_drop, mask_out, gptr, num_graphs, P, N, cs, Fsum);
  else
    k_bn_act_pool_fwd<T, false><<<grid, PTHR, (size_t)rpp * (N / cs) * sizeof(float), cgnn_stream(stream)>>>(
        Y, coef, relu, d, use_drop, mask_out, gptr, num_graphs, P, N, cs, nullptr);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

template <typename T>
int bn_act_bwd_stats_t(const T* dX, const T* Y, const uint8_t* mask, const float* coef,
                       int32_t relu, float p_drop, int64_t M, int32_t N, double* slab, int64_t slab_bytes,
                       const float* dP, const int32_t* node_graph, const int32_t* gptr,
                       void* stream) {
  if (M < 0 || !width_ok(N) || !slab || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (M == 0) return CGNN_OK;
  CGNN_NEED_BYTES(slab, slab_bytes, (int64_t)stat_blocks(M) * 2 * N * (int64_t)sizeof(double));
  if ((!dX && !dP) || !Y || !coef || (p_drop > 0.f && !mask)) return CGNN_EINVAL;
  if (dP && (!node_graph || !gptr)) return CGNN_EINVAL;
  int use_drop;
  DropCfg d = make_drop(p_drop, 0, &use_drop);
  const int rpp = THR / (N >> 2);
  k_colstats<true, T><<<stat_blocks(M), THR, (size_t)rpp * 2 * N * sizeof(double), cgnn_stream(stream)>>>(
      dX, Y, mask, coef, relu, d, use_drop, M, N, slab, PoolGrad{dP, node_graph, gptr}, stat_rows(M));
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}



template <typename T>
int bn_act_bwd_apply_t(const T* dX, const T* Y, const uint8_t* mask, const float* coef,
                       const float* bwc, int32_t relu, float p_drop, int32_t relu_in,
                       double* colsum_slab, int64_t colsum_slab_bytes, T* dY, int64_t M, int32_t N, const float* dP,
                       const int32_t* node_graph, const int32_t* gptr, void* stream) {
  if (M < 0 || !width_ok(N) || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (M == 0) return CGNN_OK;
  CGNN_NEED_BYTES(colsum_slab, colsum_slab_bytes, (int64_t)apply_blocks(M, N, true) * N * (int64_t)sizeof(double));
  if ((!dX && !dP) || !Y || !coef || !bwc || !dY || (p_drop > 0.f && !mask)) return CGNN_EINVAL;
  if (dP && (!node_graph || !gptr)) return CGNN_EINVAL;
  int use_drop;
  DropCfg d = make_drop(p_drop, 0, &use_drop);
  k_bn_act_apply<true, T><<<apply_blocks(M, N, colsum_slab != nullptr), 256, 0, cgnn_stream(stream)>>>(
      Y, dX, coef, bwc, relu, d, use_drop, nullptr, mask, dY, M, N, relu_in, colsum_slab,
      PoolGrad{dP, node_graph, gptr});
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // namespace

extern "C" {

int cgnn_bn_act_width_ok(int32_t N) { return width_ok(N) ? 1 : 0; }

int64_t cgnn_bn_act_slab_rows(int64_t M) { return M < 0 ? CGNN_EINVAL : (M == 0 ? 1 : stat_blocks(M)); }

int cgnn_bn_act_finalize(const double* slab, int32_t rows, int32_t N, double count,
                         const double* count_dev, int32_t training,
                         const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps,
                         int64_t* num_batches_tracked, float* coef, void* stream) {
  if (!width_ok(N) || !gamma || !beta || !running_mean || !running_var || !coef) return CGNN_EINVAL;
  if (training && (!slab || rows <= 0 || (!count_dev && count <= 0.0))) return CGNN_EINVAL;
  k_bn_fwd_finalize_n<<<(N + FIN_C - 1) / FIN_C, FIN_C * FIN_G, 0, cgnn_stream(stream)>>>(
      slab, rows, N, count, count_dev, training, gamma, beta, running_mean, running_var, momentum, eps,
      reinterpret_cast<long long*>(num_batches_tracked), coef);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_bn_act_bwd_finalize(const double* slab, int32_t rows, int32_t N, double count,
                             const double* count_dev, int32_t zero_coef, float* dgamma, float* dbeta,
                             float* bwc, void* stream) {
  if (!width_ok(N) || !slab || rows <= 0 || (!count_dev && count <= 0.0) || !dgamma || !dbeta || !bwc)
    return CGNN_EINVAL;
  k_bn_bwd_finalize_n<<<(N + FIN_C - 1) / FIN_C, FIN_C * FIN_G, 0, cgnn_stream(stream)>>>(slab, rows, N, count, count_dev, zero_coef, dgamma,
                                                          dbeta, bwc);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int64_t cgnn_bn_act_apply_blocks(int64_t M, int32_t N) {
  return (M < 0 || !width_ok(N)) ? CGNN_EINVAL : (int64_t)apply_blocks(M, N, true);
}


// ---- fp32 storage
int cgnn_bn_act_fwd_stats(const float* Y, int64_t M, int32_t N, double* slab, int64_t slab_bytes, void* stream) {
  return bn_act_fwd_stats_t<float>(Y, M, N, slab, slab_bytes, stream);
}
int cgnn_bn_act_fwd_apply(const float* Y, const float* coef, int32_t relu, float p_drop, uint64_t seed,
                          const uint32_t* seed_dev, uint8_t* mask_out, float* X, int64_t M, int32_t N,
                          void* stream) {
  return bn_act_fwd_apply_t<float>(Y, coef, relu, p_drop, seed, seed_dev, mask_out, X, M, N, stream);
}
int cgnn_bn_act_pool_fwd(const float* Y, const float* coef, int32_t relu, float p_drop, uint64_t seed,
                         const uint32_t* seed_dev, uint8_t* mask_out, const int32_t* gptr,
                         int32_t num_graphs, float* P, int32_t N, float* Fsum, void* stream) {
  return bn_act_pool_fwd_t<float>(Y, coef, relu, p_drop, seed, seed_dev, mask_out, gptr, num_graphs, P, N, Fsum,
                                  stream);
}

int cgnn_bn_act_pool_bwd_finalize(const float* dP, const float* Fsum, const int32_t* gptr, int32_t num_graphs,
                                  int32_t N, double count, int32_t zero_coef, float* dgamma, float* dbeta,
                                  float* bwc, void* stream) {
  if (num_graphs < 0 || !width_ok(N) || count <= 0.0 || !dP || !Fsum || !gptr || !dgamma || !dbeta || !bwc)
    return CGNN_EINVAL;
  k_bn_pool_bwd_finalize<<<(N + FIN_C - 1) / FIN_C, FIN_C * FIN_G, 0, cgnn_stream(stream)>>>(
      dP, Fsum, gptr, num_graphs, N, count, zero_coef, dgamma, dbeta, bwc);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}
int cgnn_bn_act_bwd_stats(const float* dX, const float* Y, const uint8_t* mask, const float* coef,
                          int32_t relu, float p_drop, int64_t M, int32_t N, double* slab, int64_t slab_bytes,
                          const float* dP, const int32_t* node_graph, const int32_t* gptr,
                          void* stream) {
  return bn_act_bwd_stats_t<float>(dX, Y, mask, coef, relu, p_drop, M, N, slab, slab_bytes, dP, node_graph, gptr, stream);
}
int cgnn_bn_act_bwd_apply(const float* dX, const float* Y, const uint8_t* mask, const float* coef,
                          const float* bwc, int32_t relu, float p_drop, int32_t relu_in,
                          double* colsum_slab, int64_t colsum_slab_bytes, float* dY, int64_t M, int32_t N, const float* dP,
                          const int32_t* node_graph, const int32_t* gptr, void* stream) {
  return bn_act_bwd_apply_t<float>(dX, Y, mask, coef, bwc, relu, p_drop, relu_in, colsum_slab, colsum_slab_bytes, dY, M, N, dP,
                                   node_graph, gptr, stream);
}

// ---- fp16 storage (IEEE half arrays, fp32 arithmetic, fp64 statistics): same semantics
typedef _Float16 cgnn_h;
int cgnn_bn_act_fwd_stats_f16(const void* Y, int64_t M, int32_t N, double* slab, int64_t slab_bytes, void* stream) {
  return bn_act_fwd_stats_t<cgnn_h>(static_cast<const cgnn_h*>(Y), M, N, slab, slab_bytes, stream);
}
int cgnn_bn_act_fwd_apply_f16(const void* Y, const float* coef, int32_t relu, float p_drop, uint64_t seed,
                              const uint32_t* seed_dev, uint8_t* mask_out, void* X, int64_t M,
                              int32_t N, void* stream) {
  return bn_act_fwd_apply_t<cgnn_h>(static_cast<const cgnn_h*>(Y), coef, relu, p_drop, seed, seed_dev,
                                    mask_out, static_cast<cgnn_h*>(X), M, N, stream);
}
int cgnn_bn_act_pool_fwd_f16(const void* Y, const float* coef, int32_t relu, float p_drop, uint64_t seed,
                             const uint32_t* seed_dev, uint8_t* mask_out, const int32_t* gptr,
                             int32_t num_graphs, float* P, int32_t N, float* Fsum, void* stream) {
  return bn_act_pool_fwd_t<cgnn_h>(static_cast<const cgnn_h*>(Y), coef, relu, p_drop, seed, seed_dev, mask_out,
                                   gptr, num_graphs, P, N, Fsum, stream);
}
int cgnn_bn_act_bwd_stats_f16(const void* dX, const void* Y, const uint8_t* mask, const float* coef,
                              int32_t relu, float p_drop, int64_t M, int32_t N, double* slab, int64_t slab_bytes,
                              const float* dP, const int32_t* node_graph, const int32_t* gptr,
                              void* stream) {
  return bn_act_bwd_stats_t<cgnn_h>(static_cast<const cgnn_h*>(dX), static_cast<const cgnn_h*>(Y), mask, coef,
                                    relu, p_drop, M, N, slab, slab_bytes, dP, node_graph, gptr, stream);
}
int cgnn_bn_act_bwd_apply_f16(const void* dX, const void* Y, const uint8_t* mask, const float* coef,
                              const float* bwc, int32_t relu, float p_drop, int32_t relu_in,
                              double* colsum_slab, int64_t colsum_slab_bytes, void* dY, int64_t M, int32_t N, const float* dP,
                              const int32_t* node_graph, const int32_t* gptr, void* stream) {
  return bn_act_bwd_apply_t<cgnn_h>(static_cast<const cgnn_h*>(dX), static_cast<const cgnn_h*>(Y), mask, coef,
                                    bwc, relu, p_drop, relu_in, colsum_slab, colsum_slab_bytes, static_cast<cgnn_h*>(dY), M, N,
                                    dP, node_graph, gptr, stream);
}

}  // extern "C"
