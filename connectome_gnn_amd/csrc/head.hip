// head.hip -- the graph-level classifier head in one kernel each way (gfx950, fp32).
//
//   logits = Linear2( dropout( relu( Linear1(P) ) ) )        reference models.py:196-201, 213-216
//
// P is [B, H] (one row per GRAPH, B = 512..4096), so the head is a few MFLOP: through generic
// GEMM libraries it is six launches of 5-36 us each (tile shapes meant for large problems); here
// it is one forward and one backward kernel of a few us.  Weights live in LDS, one thread per
// (row, output unit); the backward leaves per-workgroup partial parameter gradients in a slab
// [grid][H2*H + H2 + C*H2 + C] that cgnn_slab_reduce_f32 combines in fixed order.
#include "common.h"

namespace {

constexpr int HTHR = 256;
constexpr int HEAD_MAX_H = 256;          // input width (wider heads stay on the generic GEMMs);
                                         // 256: W1 [128 x 256] is 128 KB of the 160 KB LDS
constexpr int HEAD_MAX_C = 16;           // classes

__device__ __forceinline__ uint32_t hmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

struct HeadDrop {
  uint32_t thr16;      // keep iff 16-bit hash >= thr16
  float scale;         // 1 / (1 - p)
  uint32_t key0, key1;
  const uint32_t* dev_key;
};

// ---------------------------------------------------------------------------------- forward
// block: rows [r0, r0 + RB), RB = HTHR / H2 (H2 <= 128 -> RB >= 2)
__global__ void __launch_bounds__(HTHR) k_head_fwd(
    const float* __restrict__ P, int B, int H, int H2, int C, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
    HeadDrop drop, int use_drop, float* __restrict__ H1, float* __restrict__ fac,
    float* __restrict__ logits) {
  extern __shared__ float sm[];
  float* w1 = sm;                          // [H2][H + 1]
  float* pl = w1 + H2 * (H + 1);           // [RB][H]
  float* hl = pl + (HTHR / H2) * H;        // [RB][H2]
  if (drop.dev_key) drop.key1 ^= drop.dev_key[0];
  const int RB = HTHR / H2;
  // (16-byte loads, eight in flight: the scalar one-at-a-time form was a chain of 128 round trips
  // for a 128 x 256 weight -- 22 us of a 64-graph batch's step)
  if ((H & 3) == 0 && (reinterpret_cast<uintptr_t>(W1) & 15) == 0) {
    const int n4 = H2 * H / 4;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * HTHR) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = i0 + u * HTHR < n4 ? reinterpret_cast<const float4*>(W1)[i0 + u * HTHR] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = 4 * (i0 + u * HTHR);
        if (i < H2 * H) {
          float* d = w1 + (i / H) * (H + 1) + i % H;
          d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < H2 * H; i += HTHR) w1[(i / H) * (H + 1) + i % H] = W1[i];
  }
  const int j = threadIdx.x % H2, rr = threadIdx.x / H2;
  const float bj = b1[j];
  for (int r0 = blockIdx.x * RB; r0 < B; r0 += gridDim.x * RB) {
    __syncthreads();
    for (int i = threadIdx.x; i < RB * H; i += HTHR) {
      const int r = r0 + i / H;
      pl[i] = r < B ? P[(int64_t)r * H + i % H] : 0.f;
    }
    __syncthreads();
    const int r = r0 + rr;
    if (rr < RB) {
      float z = bj;
      const float* wr = w1 + j * (H + 1);
      const float* pr = pl + rr * H;
      for (int k = 0; k < H; ++k) z = fmaf(pr[k], wr[k], z);
      float f = z > 0.f ? 1.f : 0.f;
      if (use_drop) {
        const uint32_t e = (uint32_t)r * (uint32_t)H2 + (uint32_t)j;
        const uint32_t hsh = hmix32(hmix32(e ^ drop.key0) + drop.key1);
        f = ((hsh & 0xFFFFu) >= drop.thr16) ? f * drop.scale : 0.f;
      }
      const float hv = z * f;
      hl[rr * H2 + j] = hv;
      if (r < B) {
        H1[(int64_t)r * H2 + j] = hv;
        fac[(int64_t)r * H2 + j] = f;
      }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < RB * C; t += HTHR) {
      const int rl = t / C, c = t % C, row = r0 + rl;
      if (row < B) {
        float acc = b2[c];
        for (int k = 0; k < H2; ++k) acc = fmaf(hl[rl * H2 + k], W2[c * H2 + k], acc);
        logits[(int64_t)row * C + c] = acc;
      }
    }
  }
}

// --------------------------------------------------------------------------------- backward
// slab row layout: dW1 [H2*H] | db1 [H2] | dW2 [C*H2] | db2 [C]
__global__ void __launch_bounds__(HTHR) k_head_bwd(
    const float* __restrict__ dL, const float* __restrict__ P, const float* __restrict__ H1,
    const float* __restrict__ fac, int B, int H, int H2, int C, const float* __restrict__ W1,
    const float* __restrict__ W2, float* __restrict__ dP, float* __restrict__ slab) {
  extern __shared__ float sm[];
  float* w1 = sm;                          // [H2][H]  (row-major: dP reads along k)
  float* pl = w1 + H2 * H;                 // [RB][H]
  float* dh = pl + (HTHR / H2) * H;        // [RB][H2]
  float* hl = dh + (HTHR / H2) * H2;       // [RB][H2]
  float* dl = hl + (HTHR / H2) * H2;       // [RB][C]
  const int RB = HTHR / H2;
  const int WD = H2 * H + H2 + C * H2 + C;
  for (int i = threadIdx.x; i < H2 * H; i += HTHR) w1[i] = W1[i];
  // parameter-gradient accumulators of this thread: elements threadIdx + HTHR*u of the slab row
  constexpr int MAXE = (HEAD_MAX_H / 2 * HEAD_MAX_H + HEAD_MAX_H / 2 + HEAD_MAX_C * HEAD_MAX_H / 2 + HEAD_MAX_C + HTHR - 1) / HTHR;
  float g[MAXE];
#pragma unroll
  for (int u = 0; u < MAXE; ++u) g[u] = 0.f;
  const int j = threadIdx.x % H2, rr = threadIdx.x / H2;
  for (int r0 = blockIdx.x * RB; r0 < B; r0 += gridDim.x * RB) {
    __syncthreads();
    // every global load of this row group is issued before the first barrier (the kernel is a
    // chain of dependent round trips otherwise)
    const int rmine = r0 + rr;
    float fv = 0.f, hv = 0.f;
    if (rr < RB && rmine < B) {
      fv = fac[(int64_t)rmine * H2 + j];
      hv = H1[(int64_t)rmine * H2 + j];
    }
    for (int i = threadIdx.x; i < RB * H; i += HTHR) {
      const int r = r0 + i / H;
      pl[i] = r < B ? P[(int64_t)r * H + i % H] : 0.f;
    }
    for (int i = threadIdx.x; i < RB * C; i += HTHR) {
      const int r = r0 + i / C;
      dl[i] = r < B ? dL[(int64_t)r * C + i % C] : 0.f;
    }
    __syncthreads();
    if (rr < RB) {
      float d = 0.f;
      if (rmine < B) {
        for (int c = 0; c < C; ++c) d = fmaf(dl[rr * C + c], W2[c * H2 + j], d);
        d *= fv;
      }
      dh[rr * H2 + j] = d;
      hl[rr * H2 + j] = hv;
    }
    __syncthreads();
    // dP[r][k] = sum_j dh[r][j] W1[j][k]
    for (int t = threadIdx.x; t < RB * H; t += HTHR) {
      const int rl = t / H, k = t % H, row = r0 + rl;
      if (row < B) {
        float acc = 0.f;
        for (int jj = 0; jj < H2; ++jj) acc = fmaf(dh[rl * H2 + jj], w1[jj * H + k], acc);
        dP[(int64_t)row * H + k] = acc;
      }
    }
    // parameter gradients: this block's rows into the thread's accumulators
#pragma unroll
    for (int u = 0; u < MAXE; ++u) {
      const int e = threadIdx.x + HTHR * u;
      if (e < WD) {
        float acc = 0.f;
        if (e < H2 * H) {                                   // dW1[jj][k]
          const int jj = e / H, k = e % H;
          for (int rl = 0; rl < RB; ++rl) acc = fmaf(dh[rl * H2 + jj], pl[rl * H + k], acc);
        } else if (e < H2 * H + H2) {                       // db1[jj]
          const int jj = e - H2 * H;
          for (int rl = 0; rl < RB; ++rl) acc += dh[rl * H2 + jj];
        } else if (e < H2 * H + H2 + C * H2) {              // dW2[c][jj]
          const int q = e - H2 * H - H2, c = q / H2, jj = q % H2;
          for (int rl = 0; rl < RB; ++rl) acc = fmaf(dl[rl * C + c], hl[rl * H2 + jj], acc);
        } else {                                            // db2[c]
          const int c = e - H2 * H - H2 - C * H2;
          for (int rl = 0; rl < RB; ++rl) acc += dl[rl * C + c];
        }
        g[u] += acc;
      }
    }
  }
  float* out = slab + (int64_t)blockIdx.x * WD;
#pragma unroll
  for (int u = 0; u < MAXE; ++u) {
    const int e = threadIdx.x + HTHR * u;
    if (e < WD) out[e] = g[u];
  }
}

// ---------------------------------------------------------------- backward, register-tiled
// The same arithmetic as k_head_bwd for the head widths the models use (H = 2*H2 in {32, 64, 128}):
// a workgroup takes 16-row chunks; dh and dP are formed per chunk, and the weight-gradient partials
// live in registers as a JT x KT patch per thread (16 x 16 threads cover [H2 x H]), fed by one
// LDS broadcast per dh value and 16-byte reads of the P rows -- k_head_bwd spends its time in
// per-element integer divisions and two LDS reads per multiply-add, 38 us per 4096-graph batch.
// Slab row layout unchanged: dW1 [H2*H] | db1 [H2] | dW2 [C*H2] | db2 [C].
constexpr int HB_R = 16;                  // rows per chunk
constexpr int HB_MAX_GRID = 256;

template <int H, int H2, int C>
__global__ void __launch_bounds__(256) k_head_bwd_t(
    const float* __restrict__ dL, const float* __restrict__ P, const float* __restrict__ H1,
    const float* __restrict__ fac, int B, const float* __restrict__ W1, const float* __restrict__ W2,
    float* __restrict__ dP, float* __restrict__ slab) {
  constexpr int KT = H / 16, JT = H2 / 16;
  constexpr int HBR = H > 128 ? 8 : HB_R;             // rows per chunk (H = 256: W1 alone is 128 KB of LDS)
  constexpr int TPR = 256 / HBR;                      // threads per row in the dP pass
  constexpr int NSM = H2 + C * H2 + C;                // db1 | dW2 | db2: up to two per thread
  static_assert(H % 16 == 0 && H2 % 16 == 0 && KT % 2 == 0 && NSM <= 512, "head shape");
  __shared__ __attribute__((aligned(16))) float w1[H2 * H];
  __shared__ float w2[C * H2];
  __shared__ __attribute__((aligned(16))) float pl[HBR * H];
  __shared__ float dh[HBR * H2], hl[HBR * H2], dl[HBR * C];
  const int t = threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(W1) & 15) == 0) {     // 16-byte loads, eight in flight
    constexpr int N4 = H2 * H / 4;
    for (int i0 = t; i0 < N4; i0 += 8 * 256) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = i0 + u * 256 < N4 ? reinterpret_cast<const float4*>(W1)[i0 + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * 256 < N4) reinterpret_cast<float4*>(w1)[i0 + u * 256] = v[u];
    }
  } else {
    for (int i = t; i < H2 * H; i += 256) w1[i] = W1[i];
  }
  for (int i = t; i < C * H2; i += 256) w2[i] = W2[i];
  const int tj = t >> 4, tk = t & 15;                 // dW1 patch: rows JT*tj.., columns KT*tk..
  float gw1[JT][KT];
#pragma unroll
  for (int a = 0; a < JT; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b) gw1[a][b] = 0.f;
  float gsm[2] = {0.f, 0.f};                           // elements t and t + 256 of db1 | dW2 | db2
  for (int r0 = blockIdx.x * HBR; r0 < B; r0 += gridDim.x * HBR) {
    __syncthreads();
    for (int i = t; i < HBR * H / 4; i += 256) {
      const int r = r0 + i / (H / 4);
      *reinterpret_cast<float4*>(pl + 4 * i) =
          r < B ? *reinterpret_cast<const float4*>(P + (int64_t)r0 * H + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = t; i < HBR * H2; i += 256) {
      const int r = r0 + i / H2;
      hl[i] = r < B ? H1[(int64_t)r0 * H2 + i] : 0.f;
    }
    for (int i = t; i < HBR * C; i += 256) dl[i] = (r0 + i / C) < B ? dL[(int64_t)r0 * C + i] : 0.f;
    __syncthreads();
    for (int i = t; i < HBR * H2; i += 256) {
      const int rl = i / H2, j = i % H2, r = r0 + rl;
      float d = 0.f;
      if (r < B) {
#pragma unroll
        for (int c = 0; c < C; ++c) d = fmaf(dl[rl * C + c], w2[c * H2 + j], d);
        d *= fac[(int64_t)r * H2 + j];
      }
      dh[i] = d;
    }
    __syncthreads();
    // dP[r][4k4..] = sum_j dh[r][j] W1[j][4k4..]: thread (row, 4-column piece)
#pragma unroll
    for (int pc = 0; pc < (H / 4 + TPR - 1) / TPR; ++pc) {
      const int rl = t / TPR, k4 = (t % TPR) + TPR * pc;
      if (r0 + rl < B && k4 < H / 4) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int j = 0; j < H2; ++j) {
          const float d = dh[rl * H2 + j];
          const float4 w = *reinterpret_cast<const float4*>(w1 + j * H + 4 * k4);
          a.x = fmaf(d, w.x, a.x); a.y = fmaf(d, w.y, a.y); a.z = fmaf(d, w.z, a.z); a.w = fmaf(d, w.w, a.w);
        }
        *reinterpret_cast<float4*>(dP + (int64_t)(r0 + rl) * H + 4 * k4) = a;
      }
    }
    // parameter gradients of this chunk (rows past B contribute zeros: dh, dl, hl are zero there)
#pragma unroll 4
    for (int rl = 0; rl < HBR; ++rl) {
      float dv[JT], pv[KT];
#pragma unroll
      for (int a = 0; a < JT; ++a) dv[a] = dh[rl * H2 + JT * tj + a];
#pragma unroll
      for (int b = 0; b < KT; b += 2) {
        const float2 p2 = *reinterpret_cast<const float2*>(pl + rl * H + KT * tk + b);
        pv[b] = p2.x; pv[b + 1] = p2.y;
      }
#pragma unroll
      for (int a = 0; a < JT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b) gw1[a][b] = fmaf(dv[a], pv[b], gw1[a][b]);
    }
#pragma unroll
    for (int u = 0; u < (NSM + 255) / 256; ++u) {
      const int e = t + 256 * u;
      if (e < H2) {
        for (int rl = 0; rl < HBR; ++rl) gsm[u] += dh[rl * H2 + e];
      } else if (e < H2 + C * H2) {
        const int q = e - H2, c = q / H2, j = q % H2;
        for (int rl = 0; rl < HBR; ++rl) gsm[u] = fmaf(dl[rl * C + c], hl[rl * H2 + j], gsm[u]);
      } else if (e < NSM) {
        const int c = e - H2 - C * H2;
        for (int rl = 0; rl < HBR; ++rl) gsm[u] += dl[rl * C + c];
      }
    }
  }
  float* out = slab + (int64_t)blockIdx.x * (H2 * H + NSM);
#pragma unroll
  for (int a = 0; a < JT; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b) out[(JT * tj + a) * H + KT * tk + b] = gw1[a][b];
#pragma unroll
  for (int u = 0; u < (NSM + 255) / 256; ++u)
    if (t + 256 * u < NSM) out[H2 * H + t + 256 * u] = gsm[u];   // db1 | dW2 | db2 follow dW1 in this order
}

bool head_tiled(int H, int H2, int C) { return C == 2 && H == 2 * H2 && (H == 32 || H == 64 || H == 128 || H == 256); }
int head_bwd_grid(int B, int H, int H2, int C) {
  if (!head_tiled(H, H2, C)) return -1;
  const int rows = H > 128 ? 8 : HB_R;               // = HBR of k_head_bwd_t
  const int g = (B + rows - 1) / rows;
  return g < 1 ? 1 : (g > HB_MAX_GRID ? HB_MAX_GRID : g);
}

// ----------------------------------------------------------- mean cross-entropy over the batch
// loss = mean_i ( logsumexp(logits[i,:]) - logits[i, label_i] )   (torch CrossEntropyLoss defaults,
// reference train.py:39,49); also leaves dlogits for a unit upstream gradient:
// (softmax(logits[i,:]) - onehot(label_i)) / V.  torch's defaults include ignore_index = -100:
// such rows add nothing, get a zero gradient, and V counts the other rows (V = 0 -> NaN like
// torch).  Any other label outside [0, C) makes torch raise; a kernel cannot, so the loss and
// that row's gradient become NaN -- loud, and visible at the epoch's single read-back.
// One workgroup; rows summed in fp64, fixed order.
constexpr int CETHR = 1024;
constexpr int CE_IGNORE = -100;

__global__ void __launch_bounds__(CETHR) k_ce_fwd(const float* __restrict__ logits,
                                                  const int64_t* __restrict__ labels, int B, int C,
                                                  float* __restrict__ loss,
                                                  float* __restrict__ dlogits) {
  __shared__ double red[CETHR];
  __shared__ int cnt[CETHR];
  double acc = 0.0;
  int valid = 0, bad = 0;
  for (int i = threadIdx.x; i < B; i += CETHR) {
    const int64_t lab = labels[i];
    if (lab == CE_IGNORE) continue;
    if (lab < 0 || lab >= C) { bad = 1; continue; }
    const float* row = logits + (int64_t)i * C;
    float m = row[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(row[c] - m);
    acc += (double)(m + logf(se) - row[lab]);
    ++valid;
  }
  red[threadIdx.x] = acc;
  cnt[threadIdx.x] = valid | (bad << 30);
  __syncthreads();
  for (int s = CETHR / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[threadIdx.x] += red[threadIdx.x + s];
      const int a = cnt[threadIdx.x], b = cnt[threadIdx.x + s];
      cnt[threadIdx.x] = ((a & 0x3FFFFFFF) + (b & 0x3FFFFFFF)) | ((a | b) & (1 << 30));
    }
    __syncthreads();
  }
  const int total = cnt[0] & 0x3FFFFFFF;
  const bool any_bad = (cnt[0] >> 30) & 1;
  const float nanv = __int_as_float(0x7FC00000);
  if (threadIdx.x == 0) loss[0] = any_bad ? nanv : (float)(red[0] / (double)total);
  const float invv = 1.0f / (float)total;
  for (int i = threadIdx.x; i < B; i += CETHR) {
    const int64_t lab = labels[i];
    const float* row = logits + (int64_t)i * C;
    float* drow = dlogits + (int64_t)i * C;
    if (lab == CE_IGNORE) {
      for (int c = 0; c < C; ++c) drow[c] = 0.f;
    } else if (lab < 0 || lab >= C) {
      for (int c = 0; c < C; ++c) drow[c] = nanv;
    } else {
      float m = row[0];
      for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(row[c] - m);
      const float lse = m + logf(se);
      for (int c = 0; c < C; ++c) drow[c] = (expf(row[c] - lse) - (c == (int)lab ? 1.f : 0.f)) * invv;
    }
  }
}

// ------------------------------------------------ head forward + cross-entropy + head backward
// One launch for the whole classifier step when the loss is the batch-mean cross-entropy and its
// upstream gradient is the unit (the Trainer's step, reference train.py:46-51): a workgroup takes
// 16-row chunks through k_head_fwd's arithmetic (z, relu', dropout hash, logits), k_ce_fwd's per-row
// log-sum-exp and gradient -- the divisor V (labels that are not -100) is counted by every workgroup
// from the label array itself, so no workgroup waits on another -- and k_head_bwd_t's backward, and
// leaves its parameter-gradient partials AND its share of the loss (sum of its rows' losses / V) in
// its slab row: the slab fold that produces the gradients also produces the loss.  Three launches
// (7.5 + 4.8 + 7.0 us at 512 graphs) become one.
// Slab row: dW1 [H2*H] | db1 [H2] | dW2 [C*H2] | db2 [C] | loss share [1].
template <int H, int H2, int C, int HBR>
__global__ void __launch_bounds__(256) k_head_loss_t(
    const float* __restrict__ P, int B, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ b2, const int64_t* __restrict__ labels,
    HeadDrop drop, int use_drop, float* __restrict__ H1, float* __restrict__ fac,
    float* __restrict__ logits, float* __restrict__ dP, float* __restrict__ slab) {
  constexpr int KT = H / 16, JT = H2 / 16;
  constexpr int TPR = 256 / HBR;
  constexpr int NSM = H2 + C * H2 + C;
  constexpr int LDW = H + 4;                          // padded weight rows: lanes j -> banks 4j + k
  static_assert(H % 16 == 0 && H2 % 16 == 0 && KT % 2 == 0 && NSM <= 512, "head shape");
  __shared__ __attribute__((aligned(16))) float w1[H2 * LDW];
  __shared__ float w2[C * H2];
  __shared__ __attribute__((aligned(16))) float pl[HBR * H];
  __shared__ float dh[HBR * H2], hl[HBR * H2], fl[HBR * H2], dl[HBR * C], lg[HBR * C];
  __shared__ int cnt[256];
  __shared__ double lred[HBR];
  const int t = threadIdx.x;
  if (drop.dev_key) drop.key1 ^= drop.dev_key[0];
  if ((reinterpret_cast<uintptr_t>(W1) & 15) == 0) {     // 16-byte loads, eight in flight
    constexpr int N4 = H2 * H / 4;
    for (int i0 = t; i0 < N4; i0 += 8 * 256) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = i0 + u * 256 < N4 ? reinterpret_cast<const float4*>(W1)[i0 + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = 4 * (i0 + u * 256);
        if (i < H2 * H) *reinterpret_cast<float4*>(w1 + (i / H) * LDW + i % H) = v[u];
      }
    }
  } else {
    for (int i = t; i < H2 * H; i += 256) w1[(i / H) * LDW + i % H] = W1[i];
  }
  for (int i = t; i < C * H2; i += 256) w2[i] = W2[i];
  // V = number of rows that count (k_ce_fwd), bit 30 = a label outside [0, C) other than -100
  {
    int valid = 0, bad = 0;
    for (int i = t; i < B; i += 256) {
      const int64_t lab = labels[i];
      if (lab == CE_IGNORE) continue;
      if (lab < 0 || lab >= C) bad = 1; else ++valid;
    }
    cnt[t] = valid | (bad << 30);
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
      if (t < sft) {
        const int a = cnt[t], b = cnt[t + sft];
        cnt[t] = ((a & 0x3FFFFFFF) + (b & 0x3FFFFFFF)) | ((a | b) & (1 << 30));
      }
      __syncthreads();
    }
  }
  const int total = cnt[0] & 0x3FFFFFFF;
  const bool any_bad = (cnt[0] >> 30) & 1;
  const float nanv = __int_as_float(0x7FC00000);
  const float invv = 1.0f / (float)total;
  double lacc = 0.0;                                   // thread rl < HBR: loss of its rows

  const int tj = t >> 4, tk = t & 15;                 // dW1 patch: rows JT*tj.., columns KT*tk..
  float gw1[JT][KT];
#pragma unroll
  for (int a = 0; a < JT; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b) gw1[a][b] = 0.f;
  float gsm[2] = {0.f, 0.f};                           // elements t and t + 256 of db1 | dW2 | db2
  for (int r0 = blockIdx.x * HBR; r0 < B; r0 += gridDim.x * HBR) {
    __syncthreads();
    for (int i = t; i < HBR * H / 4; i += 256) {
      const int r = r0 + i / (H / 4);
      *reinterpret_cast<float4*>(pl + 4 * i) =
          r < B ? *reinterpret_cast<const float4*>(P + (int64_t)r0 * H + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // ---- forward: h = drop(relu(P W1^T + b1))   (k_head_fwd: the same fmaf chain over k, the same hash)
    for (int i = t; i < HBR * H2; i += 256) {
      const int rl = i / H2, j = i % H2, r = r0 + rl;
      float z = b1[j];
      const float* wr = w1 + j * LDW;
      const float* pr = pl + rl * H;
#pragma unroll 4
      for (int k = 0; k < H; k += 4) {
        const float4 pv = *reinterpret_cast<const float4*>(pr + k);
        const float4 wv = *reinterpret_cast<const float4*>(wr + k);
        z = fmaf(pv.x, wv.x, z); z = fmaf(pv.y, wv.y, z); z = fmaf(pv.z, wv.z, z); z = fmaf(pv.w, wv.w, z);
      }
      float f = z > 0.f ? 1.f : 0.f;
      if (use_drop) {
        const uint32_t e = (uint32_t)r * (uint32_t)H2 + (uint32_t)j;
        const uint32_t hsh = hmix32(hmix32(e ^ drop.key0) + drop.key1);
        f = ((hsh & 0xFFFFu) >= drop.thr16) ? f * drop.scale : 0.f;
      }
      const float hv = r < B ? z * f : 0.f;
      hl[i] = hv;
      fl[i] = r < B ? f : 0.f;
      if (r < B) {
        H1[(int64_t)r * H2 + j] = hv;
        fac[(int64_t)r * H2 + j] = f;
      }
    }
    __syncthreads();
    // ---- logits = h W2^T + b2
    if (t < HBR * C) {
      const int rl = t / C, c = t % C, row = r0 + rl;
      float acc = b2[c];
      for (int k = 0; k < H2; ++k) acc = fmaf(hl[rl * H2 + k], w2[c * H2 + k], acc);
      lg[t] = acc;
      if (row < B) logits[(int64_t)row * C + c] = acc;
    }
    __syncthreads();
    // ---- cross-entropy of the chunk's rows and its gradient (k_ce_fwd's formulas)
    if (t < HBR) {
      const int row = r0 + t;
      float* drow = dl + t * C;
#pragma unroll
      for (int c = 0; c < C; ++c) drow[c] = 0.f;
      if (row < B) {
        const int64_t lab = labels[row];
        const float* lr = lg + t * C;
        if (lab == CE_IGNORE) {
        } else if (lab < 0 || lab >= C) {
#pragma unroll
          for (int c = 0; c < C; ++c) drow[c] = nanv;
        } else {
          float m = lr[0];
#pragma unroll
          for (int c = 1; c < C; ++c) m = fmaxf(m, lr[c]);
          float se = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) se += expf(lr[c] - m);
          const float lse = m + logf(se);
          lacc += (double)(lse - lr[lab]);
#pragma unroll
          for (int c = 0; c < C; ++c) drow[c] = (expf(lr[c] - lse) - (c == (int)lab ? 1.f : 0.f)) * invv;
        }
      }
    }
    __syncthreads();
    // ---- backward (k_head_bwd_t)
    for (int i = t; i < HBR * H2; i += 256) {
      const int rl = i / H2, j = i % H2;
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) d = fmaf(dl[rl * C + c], w2[c * H2 + j], d);
      dh[i] = d * fl[i];
    }
    __syncthreads();
#pragma unroll
    for (int pc = 0; pc < (H / 4 + TPR - 1) / TPR; ++pc) {
      const int rl = t / TPR, k4 = (t % TPR) + TPR * pc;
      if (r0 + rl < B && k4 < H / 4) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int j = 0; j < H2; ++j) {
          const float d = dh[rl * H2 + j];
          const float4 w = *reinterpret_cast<const float4*>(w1 + j * LDW + 4 * k4);
          a.x = fmaf(d, w.x, a.x); a.y = fmaf(d, w.y, a.y); a.z = fmaf(d, w.z, a.z); a.w = fmaf(d, w.w, a.w);
        }
        *reinterpret_cast<float4*>(dP + (int64_t)(r0 + rl) * H + 4 * k4) = a;
      }
    }
#pragma unroll 4
    for (int rl = 0; rl < HBR; ++rl) {
      float dv[JT], pv[KT];
#pragma unroll
      for (int a = 0; a < JT; ++a) dv[a] = dh[rl * H2 + JT * tj + a];
#pragma unroll
      for (int b = 0; b < KT; b += 2) {
        const float2 p2 = *reinterpret_cast<const float2*>(pl + rl * H + KT * tk + b);
        pv[b] = p2.x; pv[b + 1] = p2.y;
      }
#pragma unroll
      for (int a = 0; a < JT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b) gw1[a][b] = fmaf(dv[a], pv[b], gw1[a][b]);
    }
#pragma unroll
    for (int u = 0; u < (NSM + 255) / 256; ++u) {
      const int e = t + 256 * u;
      if (e < H2) {
        for (int rl = 0; rl < HBR; ++rl) gsm[u] += dh[rl * H2 + e];
      } else if (e < H2 + C * H2) {
        const int q = e - H2, c = q / H2, j = q % H2;
        for (int rl = 0; rl < HBR; ++rl) gsm[u] = fmaf(dl[rl * C + c], hl[rl * H2 + j], gsm[u]);
      } else if (e < NSM) {
        const int c = e - H2 - C * H2;
        for (int rl = 0; rl < HBR; ++rl) gsm[u] += dl[rl * C + c];
      }
    }
  }
  if (t < HBR) lred[t] = lacc;
  __syncthreads();
  float* out = slab + (int64_t)blockIdx.x * (H2 * H + NSM + 1);
#pragma unroll
  for (int a = 0; a < JT; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b) out[(JT * tj + a) * H + KT * tk + b] = gw1[a][b];
#pragma unroll
  for (int u = 0; u < (NSM + 255) / 256; ++u)
    if (t + 256 * u < NSM) out[H2 * H + t + 256 * u] = gsm[u];
  if (t == 0) {
    double sum = 0.0;
    for (int i = 0; i < HBR; ++i) sum += lred[i];
    out[H2 * H + NSM] = any_bad ? nanv : (float)(sum / (double)total);
  }
}

constexpr size_t HEAD_LDS_MAX = 160 * 1024;
size_t head_fwd_lds(int H, int H2) {
  const int RB = HTHR / H2;
  return sizeof(float) * ((size_t)H2 * (H + 1) + (size_t)RB * H + (size_t)RB * H2);
}
size_t head_bwd_lds(int H, int H2, int C) {
  const int RB = HTHR / H2;
  return sizeof(float) * ((size_t)H2 * H + (size_t)RB * H + 2 * (size_t)RB * H2 + (size_t)RB * C);
}
bool head_ok(int H, int H2, int C) {
  return H >= 1 && H <= HEAD_MAX_H && H2 >= 1 && H2 <= HEAD_MAX_H / 2 && HTHR % H2 == 0 && C >= 1 &&
         C <= HEAD_MAX_C && head_fwd_lds(H, H2) <= HEAD_LDS_MAX && head_bwd_lds(H, H2, C) <= HEAD_LDS_MAX;
}
// dynamic LDS above 64 KB has to be allowed per kernel (once per device)
bool head_allow_lds() {
  static bool done[CGNN_MAX_DEVICES] = {};
  bool& d = done[cgnn_device_ordinal()];
  if (!d) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)HEAD_LDS_MAX) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_bwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)HEAD_LDS_MAX) != hipSuccess)
      return false;
    d = true;
  }
  return true;
}

// rows per chunk of k_head_loss_t: H = 256 keeps the whole 128 KB weight in LDS, so few rows fit beside it;
// small batches take 2-row chunks so that a 64-graph batch spreads over 32 workgroups instead of 8
int head_loss_rows(int B, int H) { return H > 128 ? (B <= 128 ? 2 : 8) : HB_R; }
int head_loss_grid(int B, int H) {
  const int rows = head_loss_rows(B, H);
  const int g = (B + rows - 1) / rows;
  return g < 1 ? 1 : (g > HB_MAX_GRID ? HB_MAX_GRID : g);
}

int head_grid(int B, int H2) {
  const int RB = HTHR / H2;
  int g = (B + RB - 1) / RB;
  const int cap = 4 * cgnn_fused_grid();   // several short workgroups per CU: the kernels are chains
                                           // of dependent loads, overlap comes from co-residency
  return g < 1 ? 1 : (g > cap ? cap : g);
}

}  // namespace

extern "C" {

int cgnn_head_supported(int32_t H, int32_t H2, int32_t C) { return head_ok(H, H2, C) ? 1 : 0; }
int cgnn_head_grid(int32_t B, int32_t H, int32_t H2, int32_t C) {
  // rows of the slab cgnn_head_bwd_f32 fills (H = 2*H2, C = 2: the register-tiled kernel; other
  // supported shapes: the generic one)
  if (B < 0 || !head_ok(H, H2, C)) return CGNN_EINVAL;
  const int g = head_bwd_grid(B, H, H2, C);
  return g > 0 ? g : head_grid(B, H2);
}

int cgnn_head_fwd_f32(const float* P, int32_t B, int32_t H, int32_t H2, int32_t C, const float* W1,
                      const float* b1, const float* W2, const float* b2, float p_drop,
                      uint64_t seed, const uint32_t* seed_dev, float* H1, float* fac, float* logits,
                      void* stream) {
  if (B < 0 || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (!head_ok(H, H2, C)) return CGNN_EUNSUPPORTED;
  if (B == 0) return CGNN_OK;
  if (!P || !W1 || !b1 || !W2 || !b2 || !H1 || !fac || !logits) return CGNN_EINVAL;
  HeadDrop d;
  double thr = (double)p_drop * 65536.0 + 0.5;
  if (thr > 65535.0) thr = 65535.0;
  d.thr16 = (uint32_t)thr;
  d.scale = p_drop > 0.f ? (float)(1.0 / (1.0 - (double)p_drop)) : 1.0f;   // reference: 1/(1-p)
  d.key0 = (uint32_t)(seed & 0xFFFFFFFFu) * 0x9E3779B9u + 0x7F4A7C15u;
  d.key1 = (uint32_t)(seed >> 32) ^ 0x94D049BBu;
  d.dev_key = seed_dev;
  const size_t lds = head_fwd_lds(H, H2);
  if (lds > 64 * 1024 && !head_allow_lds()) return CGNN_ELAUNCH;
  k_head_fwd<<<head_grid(B, H2), HTHR, lds, cgnn_stream(stream)>>>(
      P, B, H, H2, C, W1, b1, W2, b2, d, p_drop > 0.f ? 1 : 0, H1, fac, logits);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_cross_entropy_f32(const float* logits, const int64_t* labels, int32_t B, int32_t C,
                           float* loss, float* dlogits, void* stream) {
  if (B <= 0 || C <= 0 || !logits || !labels || !loss || !dlogits) return CGNN_EINVAL;
  k_ce_fwd<<<1, CETHR, 0, cgnn_stream(stream)>>>(logits, labels, B, C, loss, dlogits);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_head_bwd_f32(const float* dlogits, const float* P, const float* H1, const float* fac,
                      int32_t B, int32_t H, int32_t H2, int32_t C, const float* W1, const float* W2,
                      float* dP, float* slab, int64_t slab_bytes, void* stream) {
  if (B <= 0) return CGNN_EINVAL;
  if (!head_ok(H, H2, C)) return CGNN_EUNSUPPORTED;
  if (!dlogits || !P || !H1 || !fac || !W1 || !W2 || !dP || !slab) return CGNN_EINVAL;
  const int tg = head_bwd_grid(B, H, H2, C);
  const int64_t wd = (int64_t)H2 * H + H2 + (int64_t)C * H2 + C;      // dW1 | db1 | dW2 | db2 per slab row
  CGNN_NEED_BYTES(slab, slab_bytes, (int64_t)(tg > 0 ? tg : head_grid(B, H2)) * wd * (int64_t)sizeof(float));
  if (tg > 0) {
    if (H == 64) k_head_bwd_t<64, 32, 2><<<tg, 256, 0, cgnn_stream(stream)>>>(dlogits, P, H1, fac, B, W1, W2, dP, slab);
    else if (H == 128) k_head_bwd_t<128, 64, 2><<<tg, 256, 0, cgnn_stream(stream)>>>(dlogits, P, H1, fac, B, W1, W2, dP, slab);
    else if (H == 256) k_head_bwd_t<256, 128, 2><<<tg, 256, 0, cgnn_stream(stream)>>>(dlogits, P, H1, fac, B, W1, W2, dP, slab);
    else k_head_bwd_t<32, 16, 2><<<tg, 256, 0, cgnn_stream(stream)>>>(dlogits, P, H1, fac, B, W1, W2, dP, slab);
    CGNN_CHECK_LAUNCH();
    return CGNN_OK;
  }
  const size_t lds = head_bwd_lds(H, H2, C);
  if (lds > 64 * 1024 && !head_allow_lds()) return CGNN_ELAUNCH;
  k_head_bwd<<<head_grid(B, H2), HTHR, lds, cgnn_stream(stream)>>>(dlogits, P, H1, fac, B, H, H2, C, W1,
                                                                 W2, dP, slab);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_head_loss_grid(int32_t B, int32_t H, int32_t H2, int32_t C) {
  if (B < 0 || !head_ok(H, H2, C) || !head_tiled(H, H2, C)) return CGNN_EINVAL;
  return head_loss_grid(B, H);
}

int cgnn_head_loss_f32(const float* P, int32_t B, int32_t H, int32_t H2, int32_t C, const float* W1,
                       const float* b1, const float* W2, const float* b2, const int64_t* labels,
                       float p_drop, uint64_t seed, const uint32_t* seed_dev, float* H1, float* fac,
                       float* logits, float* dP, float* slab, int64_t slab_bytes, void* stream) {
  if (B <= 0 || p_drop < 0.f || p_drop >= 1.f) return CGNN_EINVAL;
  if (!head_ok(H, H2, C) || !head_tiled(H, H2, C)) return CGNN_EUNSUPPORTED;
  if (!P || !W1 || !b1 || !W2 || !b2 || !labels || !H1 || !fac || !logits || !dP || !slab) return CGNN_EINVAL;
  HeadDrop d;
  double thr = (double)p_drop * 65536.0 + 0.5;
  if (thr > 65535.0) thr = 65535.0;
  d.thr16 = (uint32_t)thr;
  d.scale = p_drop > 0.f ? (float)(1.0 / (1.0 - (double)p_drop)) : 1.0f;
  d.key0 = (uint32_t)(seed & 0xFFFFFFFFu) * 0x9E3779B9u + 0x7F4A7C15u;
  d.key1 = (uint32_t)(seed >> 32) ^ 0x94D049BBu;
  d.dev_key = seed_dev;
  const int use = p_drop > 0.f ? 1 : 0;
  const int rows = head_loss_rows(B, H);
  const int tg = head_loss_grid(B, H);
  CGNN_NEED_BYTES(slab, slab_bytes,
                  (int64_t)tg * ((int64_t)H2 * H + H2 + (int64_t)C * H2 + C + 1) * (int64_t)sizeof(float));
  hipStream_t st = cgnn_stream(stream);
#define CGNN_HL(HH, R) k_head_loss_t<HH, HH / 2, 2, R><<<tg, 256, 0, st>>>(P, B, W1, b1, W2, b2, labels, d, use, H1, fac, logits, dP, slab)
  if (H == 64) CGNN_HL(64, 16);
  else if (H == 128) CGNN_HL(128, 16);
  else if (H == 256 && rows == 2) CGNN_HL(256, 2);
  else if (H == 256) CGNN_HL(256, 8);
  else CGNN_HL(32, 16);
#undef CGNN_HL
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
