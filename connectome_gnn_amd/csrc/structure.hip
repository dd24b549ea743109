// structure.hip -- batch structure kernels: dst-/src-sorted CSR of the block-diagonal COO,
// GCN and GraphSAGE normalisation coefficients.  Integer work is bit-exact by construction:
// slots inside a row are in COO order (stable bucket sort = count, scan, fill, per-row sort).
//
// Reference arithmetic replaced: models.py:94-108 (GCN norm), :146-149 (SAGE w_sum), and the
// index plumbing of scatter_add_/index (models.py:50-54,112-113).
#include "common.h"

namespace {

constexpr int kScanItems = 8;             // items per thread
constexpr int kScanBlock = 256;
constexpr int kScanTile = kScanItems * kScanBlock;

__global__ void k_zero_i32(int32_t* p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

// counts: cnt_dst[dst]++, cnt_src[src]++ ; flags[0] out-of-range, flags[1] cross-graph
__global__ void k_count(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                        const int64_t* __restrict__ node_graph, int64_t nn, int64_t ne,
                        int32_t* cnt_dst, int32_t* cnt_src, int32_t* flags) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  int64_t s = src[e], d = dst[e];
  if (s < 0 || s >= nn || d < 0 || d >= nn) {
    atomicAdd(&flags[0], 1);
    return;
  }
  if (node_graph && node_graph[s] != node_graph[d]) atomicAdd(&flags[1], 1);
  atomicAdd(&cnt_dst[d], 1);
  atomicAdd(&cnt_src[s], 1);
}

// block-level exclusive scan helpers ------------------------------------------------------
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* lds /*>= 4+1 ints*/) {
  // 256 threads = 4 waves
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) lds[wid] = x;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wid; ++w) base += lds[w];
  int tot = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  *total = tot;
  return base + x - v;
}

__global__ void __launch_bounds__(kScanBlock) k_scan_tile_sums(const int32_t* in, int64_t n,
                                                                int32_t* tile_sums) {
  __shared__ int lds[8];
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) s += in[base + i];
  int tot;
  block_excl_scan(s, &tot, lds);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kScanBlock) k_scan_top(int32_t* tile_sums, int nt) {
  __shared__ int lds[8];
  int carry = 0;
  for (int b = 0; b < nt; b += kScanBlock) {
    int i = b + threadIdx.x;
    int v = i < nt ? tile_sums[i] : 0;
    int tot;
    int ex = block_excl_scan(v, &tot, lds);
    if (i < nt) tile_sums[i] = carry + ex;
    carry += tot;
  }
}

__global__ void __launch_bounds__(kScanBlock) k_scan_apply(const int32_t* in, int64_t n,
                                                            const int32_t* tile_offs,
                                                            int32_t* out) {
  __shared__ int lds[8];
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int v[kScanItems];
  int s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    v[i] = base + i < n ? in[base + i] : 0;
    s += v[i];
  }
  int tot;
  int ex = block_excl_scan(s, &tot, lds) + tile_offs[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
  }
}

__global__ void k_fill(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                       int64_t nn, int64_t ne, const int32_t* __restrict__ rowptr_dst,
                       const int32_t* __restrict__ rowptr_src, int32_t* cur_dst,
                       int32_t* cur_src, int32_t* eid_dst, int32_t* eid_src) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  int64_t s = src[e], d = dst[e];
  if (s < 0 || s >= nn || d < 0 || d >= nn) return;
  eid_dst[rowptr_dst[d] + atomicAdd(&cur_dst[d], 1)] = (int32_t)e;
  eid_src[rowptr_src[s] + atomicAdd(&cur_src[s], 1)] = (int32_t)e;
}

// One WAVE per row: restore COO order inside the row (the atomic fill order is arbitrary), then emit the
// neighbour id of every slot.  Edge ids are distinct, so an entry's place is the number of smaller ids in
// its row: the row's ids go to LDS, every lane ranks its entries against all of them (broadcast reads) and
// writes them straight to their slots.  One thread per row with an in-place insertion sort (below, kept
// for rows longer than SR_MAX) took 1.5 ms per ordering on 64 x 1000-ROI graphs at 100 edges per row.
constexpr int SR_MAX = 1024;
__global__ void __launch_bounds__(256) k_sort_rows_wave(const int32_t* __restrict__ rowptr, int32_t* eid,
                                                        const int64_t* __restrict__ other_end,
                                                        int32_t* __restrict__ col, int64_t nn, int32_t* max_deg) {
  __shared__ __attribute__((aligned(16))) int32_t keys[4][SR_MAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int dmax = 0;
  for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < nn; r0 += (int64_t)gridDim.x * 4) {   // (uniform trip count)
    const int64_t r = r0 + w;
    int b = 0, deg = 0;
    if (r < nn) {
      b = rowptr[r];
      deg = rowptr[r + 1] - b;
    }
    dmax = max(dmax, deg);
    const bool in_lds = deg <= SR_MAX;
    const int deg8 = (deg + 7) & ~7;                      // padded with INT_MAX: never smaller than a key
    if (in_lds)
      for (int i = lane; i < deg8; i += 64) keys[w][i] = i < deg ? eid[b + i] : 0x7FFFFFFF;
    __syncthreads();
    if (in_lds) {
      for (int i = lane; i < deg; i += 64) {
        const int key = keys[w][i];
        int rank = 0;
        for (int j = 0; j < deg8; j += 8) {               // two broadcast 16-byte reads per step
          const int4 a = *reinterpret_cast<const int4*>(&keys[w][j]);
          const int4 c = *reinterpret_cast<const int4*>(&keys[w][j + 4]);
          rank += (a.x < key) + (a.y < key) + (a.z < key) + (a.w < key) + (c.x < key) + (c.y < key) + (c.z < key) +
                  (c.w < key);
        }
        eid[b + rank] = key;
        col[b + rank] = (int32_t)other_end[key];
      }
    } else {
      if (lane == 0) {
        for (int i = b + 1; i < b + deg; ++i) {
          const int key = eid[i];
          int j = i - 1;
          while (j >= b && eid[j] > key) {
            eid[j + 1] = eid[j];
            --j;
          }
          eid[j + 1] = key;
        }
      }
      __threadfence();                                 // lane 0's stores before the wave's reads of them
      for (int i = lane; i < deg; i += 64) col[b + i] = (int32_t)other_end[eid[b + i]];
    }
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, __shfl_xor(dmax, o, 64));
  if (lane == 0 && dmax > 0) atomicMax(max_deg, dmax);
}

// One thread per row, insertion sort in place: for short rows (in-degree ~k <= 16 of the 84- / 360-ROI
// Watts-Strogatz graphs) the cheapest stable choice.
__global__ void k_sort_rows(const int32_t* __restrict__ rowptr, int32_t* eid,
                            const int64_t* __restrict__ other_end, int32_t* col, int64_t nn,
                            int32_t* max_deg) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int deg = 0;
  if (r < nn) {
    int b = rowptr[r], e = rowptr[r + 1];
    deg = e - b;
    for (int i = b + 1; i < e; ++i) {
      int key = eid[i];
      int j = i - 1;
      while (j >= b && eid[j] > key) {
        eid[j + 1] = eid[j];
        --j;
      }
      eid[j + 1] = key;
    }
    for (int i = b; i < e; ++i) col[i] = (int32_t)other_end[eid[i]];
  }
  // wave max -> one atomic per wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) deg = max(deg, __shfl_xor(deg, o, 64));
  if ((threadIdx.x & 63) == 0 && deg > 0) atomicMax(max_deg, deg);
}

// ------------------------------------------------------------------ grouped (per-graph) build
// When the edges of graph g are the contiguous run [eptr[g], eptr[g+1]) of the COO (what
// collate_graphs / the resident assembler produce) the whole build of a graph -- count, scan,
// fill, in-row ordering by edge id, neighbour ids, for BOTH orderings -- is done by one workgroup
// in LDS: one pass over the int64 COO, coalesced int32 outputs, no global atomics.  Same output
// bits as the generic builder above (tests compare them).
//   LDS: cnt/cur for dst and src [4][nmax+1] int; edge slots of both orderings and local
//        endpoints [4][emax] uint16 (graphs of <= 1024 nodes and < 65535 edges).
__global__ void __launch_bounds__(256) k_csr_grouped(
    const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
    const int32_t* __restrict__ gptr, const int32_t* __restrict__ eptr, int num_graphs, int64_t nn,
    int64_t ne, int nmax, int emax, int32_t* __restrict__ rowptr_dst, int32_t* __restrict__ eid_dst,
    int32_t* __restrict__ col_dst, int32_t* __restrict__ rowptr_src, int32_t* __restrict__ eid_src,
    int32_t* __restrict__ col_src, int32_t* __restrict__ flags) {
  extern __shared__ int32_t lds[];
  int32_t* cnt_d = lds;                       // [nmax+1] counts -> exclusive offsets
  int32_t* cnt_s = cnt_d + (nmax + 1);
  int32_t* cur_d = cnt_s + (nmax + 1);        // [nmax+1] fill cursors
  int32_t* cur_s = cur_d + (nmax + 1);
  uint16_t* slot_d = reinterpret_cast<uint16_t*>(cur_s + (nmax + 1));   // [emax] local edge index
  uint16_t* slot_s = slot_d + emax;                                    //        per ordered slot
  uint16_t* loc_s = slot_s + emax;            // [emax] local src of each edge
  uint16_t* loc_d = loc_s + emax;
  __shared__ int wave_tot[4];
  __shared__ int blk_max[2];
  for (int g = blockIdx.x; g < num_graphs; g += gridDim.x) {
    const int nb = gptr[g], n = gptr[g + 1] - nb;
    const int e0 = eptr[g], m = eptr[g + 1] - e0;
    for (int i = threadIdx.x; i <= n; i += 256) cnt_d[i] = cnt_s[i] = 0;
    if (threadIdx.x < 2) blk_max[threadIdx.x] = 0;
    __syncthreads();
    // 1. counts + local endpoints (invalid edges: flagged, the caller rebuilds generically)
    for (int i = threadIdx.x; i < m; i += 256) {
      const int64_t sg = src[e0 + i], dg = dst[e0 + i];
      const bool in_range = sg >= 0 && sg < nn && dg >= 0 && dg < nn;
      const bool local = in_range && sg >= nb && sg < nb + n && dg >= nb && dg < nb + n;
      if (!local) {
        atomicAdd(&flags[in_range ? 1 : 0], 1);
        loc_s[i] = loc_d[i] = 0xFFFFu;
        continue;
      }
      loc_s[i] = (uint16_t)(sg - nb);
      loc_d[i] = (uint16_t)(dg - nb);
      atomicAdd(&cnt_d[dg - nb], 1);
      atomicAdd(&cnt_s[sg - nb], 1);
    }
    __syncthreads();
    // 2. exclusive scans (n <= nmax <= 1024: 4 elements per thread), degree maxima
    for (int which = 0; which < 2; ++which) {
      int32_t* cnt = which ? cnt_s : cnt_d;
      int32_t* cur = which ? cur_s : cur_d;
      int v[4], sum = 0, mx = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = 4 * threadIdx.x + u;
        v[u] = i < n ? cnt[i] : 0;
        mx = max(mx, v[u]);
        sum += v[u];
      }
      int x = sum;
      const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
      if (lane == 63) wave_tot[wid] = x;
      if (lane == 0 && mx > 0) atomicMax(&blk_max[which], mx);
      __syncthreads();
      int base = x - sum;
      for (int w = 0; w < wid; ++w) base += wave_tot[w];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = 4 * threadIdx.x + u;
        if (i < n) {
          cnt[i] = base;
          cur[i] = base;
          (which ? rowptr_src : rowptr_dst)[nb + i] = e0 + base;
        }
        base += v[u];
      }
      __syncthreads();
    }
    if (g == num_graphs - 1 && threadIdx.x == 0) {
      rowptr_dst[nn] = (int32_t)ne;
      rowptr_src[nn] = (int32_t)ne;
    }
    if (threadIdx.x < 2 && blk_max[threadIdx.x] > 0) atomicMax(&flags[2 + threadIdx.x], blk_max[threadIdx.x]);
    // 3. fill (arbitrary order inside a row) ...
    for (int i = threadIdx.x; i < m; i += 256) {
      if (loc_s[i] == 0xFFFFu) continue;
      slot_d[atomicAdd(&cur_d[loc_d[i]], 1)] = (uint16_t)i;
      slot_s[atomicAdd(&cur_s[loc_s[i]], 1)] = (uint16_t)i;
    }
    __syncthreads();
    // 4. ... then restore COO order inside every row (rows are short: insertion sort)
    for (int r = threadIdx.x; r < 2 * n; r += 256) {
      const bool which = r >= n;
      const int row = which ? r - n : r;
      uint16_t* slot = which ? slot_s : slot_d;
      const int b = (which ? cnt_s : cnt_d)[row], e = (which ? cur_s : cur_d)[row];
      for (int i = b + 1; i < e; ++i) {
        const uint16_t key = slot[i];
        int j = i - 1;
        while (j >= b && slot[j] > key) {
          slot[j + 1] = slot[j];
          --j;
        }
        slot[j + 1] = key;
      }
    }
    __syncthreads();
    // 5. coalesced outputs: global edge ids and neighbour ids of both orderings
    for (int p = threadIdx.x; p < m; p += 256) {
      const int a = slot_d[p], c = slot_s[p];
      eid_dst[e0 + p] = e0 + a;
      col_dst[e0 + p] = nb + loc_s[a];
      eid_src[e0 + p] = e0 + c;
      col_src[e0 + p] = nb + loc_d[c];
    }
    __syncthreads();
  }
}

// GCN: deg (source side) + self loop, dis, selfc -- one thread per node, COO order sum.
__global__ void k_gcn_deg(const float* __restrict__ w, const int32_t* __restrict__ rowptr_src,
                          const int32_t* __restrict__ eid_src, int64_t nn, float* dis,
                          float* selfc) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nn) return;
  float deg = 0.f;
  int s = rowptr_src[i];
  const int e = rowptr_src[i + 1];
  for (; s + 8 <= e; s += 8) {                  // eight gathers in flight, the sum still in COO order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = w[eid_src[s + u]];
#pragma unroll
    for (int u = 0; u < 8; ++u) deg += v[u];
  }
  for (; s < e; ++s) deg += w[eid_src[s]];
  deg += 1.0f;                                  // self-loop appended last (models.py:97-100)
  float d = 1.0f / sqrtf(deg + 1e-8f);          // pow(-0.5) == rsqrt on the CPU path
  dis[i] = d;
  selfc[i] = d * 1.0f * d;
}

__global__ void k_gcn_coef(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                           const float* __restrict__ w, const float* __restrict__ dis,
                           const int32_t* __restrict__ eid_dst,
                           const int32_t* __restrict__ eid_src, int64_t nslots,
                           float* coef_dst, float* coef_src) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  int e = eid_dst[s];
  coef_dst[s] = dis[src[e]] * w[e] * dis[dst[e]];   // models.py:108, left to right
  e = eid_src[s];
  coef_src[s] = dis[src[e]] * w[e] * dis[dst[e]];
}

// 16 lanes per row: coalesced walks of the row's CSR slots, fixed-order fold of the 16 partials
__global__ void __launch_bounds__(256) k_sage_den(const float* __restrict__ w, const int32_t* __restrict__ rowptr_dst,
                                                  const int32_t* __restrict__ eid_dst, int64_t nn, float* den,
                                                  float* w_dst) {
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  float sum = 0.f;
  if (i < nn) {
    const int e = rowptr_dst[i + 1];
    for (int s = rowptr_dst[i] + l; s < e; s += 16) {
      const float v = w[eid_dst[s]];
      w_dst[s] = v;
      sum += v;
    }
  }
#pragma unroll
  for (int o = 8; o; o >>= 1) sum += __shfl_xor(sum, o, 16);
  if (i < nn && l == 0) den[i] = sum + 1e-8f;    // models.py:149
}

__global__ void k_sage_coef_bwd(const int64_t* __restrict__ dst, const float* __restrict__ w,
                                const float* __restrict__ den,
                                const int32_t* __restrict__ eid_src, int64_t nslots,
                                float* coef_src_bwd) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  int e = eid_src[s];
  coef_src_bwd[s] = w[e] / den[dst[e]];
}


// ---------------------------------------------------------------- blocked-ELL (fused path)
// One thread per 16-row block: width = max degree of its rows + 1 (the self-loop entry).
__global__ void k_bell_width(const int32_t* __restrict__ tile_ptr,
                             const int32_t* __restrict__ tile_blk, int num_tiles,
                             const int32_t* __restrict__ rowptr, int32_t* __restrict__ counts) {
  const int t = blockIdx.x;
  if (t >= num_tiles) return;
  const int base = tile_ptr[t], n = tile_ptr[t + 1] - base;
  const int b0 = tile_blk[t], nb = tile_blk[t + 1] - b0;
  for (int lb = threadIdx.x; lb < nb; lb += blockDim.x) {
    int w = 0;
    for (int i = 0; i < 16; ++i) {
      const int r = 16 * lb + i;
      if (r < n) w = max(w, rowptr[base + r + 1] - rowptr[base + r]);
    }
    counts[b0 + lb] = (w + 1) * 16;             // entries of this block
  }
}

// One 16-lane group per block row-set: lane i of the group owns row i of the block and writes
// entry (s, i) for every step s -> 128-byte coalesced lines per step.
__global__ void __launch_bounds__(256) k_bell_fill(
    const int32_t* __restrict__ tile_ptr, const int32_t* __restrict__ tile_blk, int num_tiles,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ eid, const float* __restrict__ w, float self_weight,
    const int32_t* __restrict__ blk_off, uint2* __restrict__ ent) {
  const int t = blockIdx.x;
  if (t >= num_tiles) return;
  const int base = tile_ptr[t], n = tile_ptr[t + 1] - base;
  const int b0 = tile_blk[t], nb = tile_blk[t + 1] - b0;
  const int i = threadIdx.x & 15;
  for (int lb = threadIdx.x >> 4; lb < nb; lb += blockDim.x >> 4) {
    const int off = blk_off[b0 + lb];
    const int width = (blk_off[b0 + lb + 1] - off) >> 4;
    const int r = 16 * lb + i;
    int beg = 0, deg = -1;                      // deg = -1: row does not exist -> all padding
    if (r < n) {
      beg = rowptr[base + r];
      deg = rowptr[base + r + 1] - beg;
    }
    // 8 steps at a time: all neighbour ids / edge ids first, then the weight gathers, then the
    // stores (three dependent round trips per 8 entries instead of per entry)
    for (int s0 = 0; s0 < width; s0 += 8) {
      int cv[8], ev[8];
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u;
        cv[u] = s < deg ? col[beg + s] : 0;
        ev[u] = s < deg ? eid[beg + s] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = ev[u] >= 0 ? w[ev[u]] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u;
        if (s < width) {
          uint2 e = make_uint2(0u, 0u);           // padding: row 0 of the tile, weight +0
          if (s < deg) {
            e.x = (uint32_t)(cv[u] - base) * 256u;
            e.y = __float_as_uint(wv[u]);
          } else if (s == deg) {
            e.x = (uint32_t)r * 256u;             // the appended self-loop (GCN: weight 1), last
            e.y = __float_as_uint(self_weight);
          }
          ent[(int64_t)off + 16 * s + i] = e;
        }
      }
    }
  }
}

__global__ void k_gather_f32(const float* __restrict__ src, const int32_t* __restrict__ idx,
                             int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

// dis from edge weights already in src-CSR slot order (contiguous per row, COO order).
__global__ void k_gcn_dis(const float* __restrict__ w_src, const int32_t* __restrict__ rowptr_src,
                          int64_t nn, float* __restrict__ dis) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nn) return;
  const int beg = rowptr_src[i], end = rowptr_src[i + 1];
  float deg = 0.f;
  // same summation order as a plain loop, but 16 independent loads in flight per pass
  for (int s0 = beg; s0 < end; s0 += 16) {
    float wv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) wv[u] = s0 + u < end ? w_src[s0 + u] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) deg += wv[u];
  }
  deg += 1.0f;
  dis[i] = 1.0f / sqrtf(deg + 1e-8f);
}

inline unsigned blocks_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// in-place exclusive scan of counts[0..n) (n = Nn+1, counts[Nn] = 0) using tile_sums scratch
int scan_i32(int32_t* counts, int64_t n, int32_t* tile_sums, hipStream_t st) {
  int nt = (int)((n + kScanTile - 1) / kScanTile);
  k_scan_tile_sums<<<nt, kScanBlock, 0, st>>>(counts, n, tile_sums);
  k_scan_top<<<1, kScanBlock, 0, st>>>(tile_sums, nt);
  k_scan_apply<<<nt, kScanBlock, 0, st>>>(counts, n, tile_sums, counts);
  return hipGetLastError() == hipSuccess ? CGNN_OK : CGNN_ELAUNCH;
}

// bookkeeping of a replayed epoch, one thread: the next batch's position and the running loss tally
__global__ void k_epoch_advance(int64_t* cursor, int64_t step, const float* __restrict__ loss, float weight,
                                float* tally) {
  if (tally) tally[0] += loss[0] * weight;
  cursor[0] += step;
}

// Several row gathers by ONE id list in one launch (resident datasets: node features, labels, block
// offsets and `dis` of a batch's subjects): dst_j[i] = src_j[ids[i]], rows of row_bytes_j bytes.
// blockIdx.y = job, blockIdx.x strides over the ids; 16-byte words when a job's rows allow it.
__global__ void __launch_bounds__(256) k_gather_rows(cgnn_gather_jobs jobs, const int64_t* __restrict__ ids, int nids,
                                                     const int64_t* __restrict__ ids_offset) {
  if (ids_offset) ids += ids_offset[0];          // the batch's position in a longer id list (device cursor)
  const int jb = blockIdx.y;
  const int64_t rb = jobs.row_bytes[jb];
  const char* src = static_cast<const char*>(jobs.src[jb]);
  char* dst = static_cast<char*>(jobs.dst[jb]);
  const bool wide = (rb & 15) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  for (int i = blockIdx.x; i < nids; i += gridDim.x) {
    const int64_t id = ids[i];
    const char* s = src + id * rb;
    char* d = dst + (int64_t)i * rb;
    if (wide) {
      for (int64_t o = 16 * (int64_t)threadIdx.x; o < rb; o += 16 * 256)
        *reinterpret_cast<uint4*>(d + o) = *reinterpret_cast<const uint4*>(s + o);
    } else {
      for (int64_t o = 4 * (int64_t)threadIdx.x; o < rb; o += 4 * 256)
        *reinterpret_cast<uint32_t*>(d + o) = *reinterpret_cast<const uint32_t*>(s + o);
    }
  }
}

}  // namespace

extern "C" {

int cgnn_abi_version(void) { return CGNN_ABI_VERSION; }
const char* cgnn_build_target(void) { return "gfx950"; }

int64_t cgnn_csr_workspace_bytes(int64_t nn, int64_t ne) {
  if (nn < 0 || ne < 0) return CGNN_EINVAL;
  int64_t nt = (nn + 1 + kScanTile - 1) / kScanTile;
  // two cursor arrays [nn] + two tile-sum arrays [nt]
  return cgnn_align_up((2 * nn + 2 * nt + 8) * (int64_t)sizeof(int32_t), 256);
}

int cgnn_csr_build(const int64_t* edge_index, const int64_t* node_graph, int64_t nn, int64_t ne,
                   int32_t* rowptr_dst, int32_t* eid_dst, int32_t* col_dst,
                   int32_t* rowptr_src, int32_t* eid_src, int32_t* col_src, int32_t* flags,
                   void* workspace, int64_t workspace_bytes, void* stream) {
  if (nn < 0 || ne < 0 || nn >= INT32_MAX || ne >= INT32_MAX) return CGNN_EINVAL;
  if (!rowptr_dst || !rowptr_src || !flags || !workspace) return CGNN_EINVAL;
  if (ne > 0 && (!edge_index || !eid_dst || !col_dst || !eid_src || !col_src)) return CGNN_EINVAL;
  CGNN_NEED_BYTES(workspace, workspace_bytes,
                  (2 * nn + 2 * ((nn + 1 + kScanTile - 1) / kScanTile)) * (int64_t)sizeof(int32_t));
  hipStream_t st = cgnn_stream(stream);
  const int64_t* src = edge_index;
  const int64_t* dst = edge_index + ne;
  int32_t* cur_dst = static_cast<int32_t*>(workspace);
  int32_t* cur_src = cur_dst + nn;
  int32_t* tiles_a = cur_src + nn;
  int64_t nt = (nn + 1 + kScanTile - 1) / kScanTile;
  int32_t* tiles_b = tiles_a + nt;

  k_zero_i32<<<blocks_for(nn + 1, 256), 256, 0, st>>>(rowptr_dst, nn + 1);
  k_zero_i32<<<blocks_for(nn + 1, 256), 256, 0, st>>>(rowptr_src, nn + 1);
  k_zero_i32<<<blocks_for(2 * nn, 256), 256, 0, st>>>(cur_dst, 2 * nn);
  k_zero_i32<<<1, 64, 0, st>>>(flags, 4);
  CGNN_CHECK_LAUNCH();
  if (ne > 0) {
    k_count<<<blocks_for(ne, 256), 256, 0, st>>>(src, dst, node_graph, nn, ne, rowptr_dst,
                                                 rowptr_src, flags);
    CGNN_CHECK_LAUNCH();
  }
  int rc = scan_i32(rowptr_dst, nn + 1, tiles_a, st);
  if (rc) return rc;
  rc = scan_i32(rowptr_src, nn + 1, tiles_b, st);
  if (rc) return rc;
  if (ne > 0) {
    k_fill<<<blocks_for(ne, 256), 256, 0, st>>>(src, dst, nn, ne, rowptr_dst, rowptr_src, cur_dst,
                                                cur_src, eid_dst, eid_src);
    CGNN_CHECK_LAUNCH();
    if (nn > 0) {
      if (ne > 24 * nn) {                                  // long rows: a wave per row
        const unsigned wg = (unsigned)((nn + 3) / 4 < 16384 ? (nn + 3) / 4 : 16384);
        k_sort_rows_wave<<<wg, 256, 0, st>>>(rowptr_dst, eid_dst, src, col_dst, nn, flags + 2);
        k_sort_rows_wave<<<wg, 256, 0, st>>>(rowptr_src, eid_src, dst, col_src, nn, flags + 3);
      } else {
        k_sort_rows<<<blocks_for(nn, 256), 256, 0, st>>>(rowptr_dst, eid_dst, src, col_dst, nn,
                                                         flags + 2);
        k_sort_rows<<<blocks_for(nn, 256), 256, 0, st>>>(rowptr_src, eid_src, dst, col_src, nn,
                                                         flags + 3);
      }
      CGNN_CHECK_LAUNCH();
    }
  }
  return CGNN_OK;
}

int cgnn_csr_build_grouped(const int64_t* edge_index, const int32_t* gptr, const int32_t* eptr,
                           int32_t num_graphs, int64_t nn, int64_t ne, int32_t max_nodes,
                           int32_t max_edges, int32_t* rowptr_dst, int32_t* eid_dst, int32_t* col_dst,
                           int32_t* rowptr_src, int32_t* eid_src, int32_t* col_src, int32_t* flags,
                           void* stream) {
  if (nn < 0 || ne < 0 || nn >= INT32_MAX || ne >= INT32_MAX || num_graphs < 0 || max_nodes < 0 ||
      max_edges < 0)
    return CGNN_EINVAL;
  if (!rowptr_dst || !rowptr_src || !flags || !gptr || !eptr) return CGNN_EINVAL;
  if (ne > 0 && (!edge_index || !eid_dst || !col_dst || !eid_src || !col_src)) return CGNN_EINVAL;
  const size_t lds = sizeof(int32_t) * 4 * ((size_t)max_nodes + 1) + sizeof(uint16_t) * 4 * (size_t)max_edges;
  constexpr size_t kMaxLds = 128 * 1024;
  if (max_nodes > 1024 || max_edges >= 65535 || lds > kMaxLds || num_graphs == 0) return CGNN_EUNSUPPORTED;
  static bool attr_set_dev[CGNN_MAX_DEVICES] = {};
  bool& attr_set = attr_set_dev[cgnn_device_ordinal()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_csr_grouped),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds) != hipSuccess)
      return CGNN_ELAUNCH;
    attr_set = true;
  }
  hipStream_t st = cgnn_stream(stream);
  k_zero_i32<<<1, 64, 0, st>>>(flags, 4);
  CGNN_CHECK_LAUNCH();
  const int cus = cgnn_fused_grid();
  const int grid = num_graphs < 8 * cus ? num_graphs : 8 * cus;
  k_csr_grouped<<<grid, 256, lds, st>>>(edge_index, edge_index + ne, gptr, eptr, num_graphs, nn, ne,
                                         max_nodes, max_edges, rowptr_dst, eid_dst, col_dst, rowptr_src,
                                         eid_src, col_src, flags);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_norm(const int64_t* edge_index, const float* w, int64_t nn, int64_t ne,
                  const int32_t* rowptr_dst, const int32_t* eid_dst, const int32_t* rowptr_src,
                  const int32_t* eid_src, float* dis, float* selfc, float* coef_dst,
                  float* coef_src, void* stream) {
  if (nn < 0 || ne < 0) return CGNN_EINVAL;
  if (nn > 0 && (!rowptr_src || !dis || !selfc)) return CGNN_EINVAL;
  if (ne > 0 && (!edge_index || !w || !eid_dst || !eid_src || !coef_dst || !coef_src))
    return CGNN_EINVAL;
  (void)rowptr_dst;
  hipStream_t st = cgnn_stream(stream);
  if (nn > 0) {
    k_gcn_deg<<<blocks_for(nn, 256), 256, 0, st>>>(w, rowptr_src, eid_src, nn, dis, selfc);
    CGNN_CHECK_LAUNCH();
  }
  if (ne > 0) {
    k_gcn_coef<<<blocks_for(ne, 256), 256, 0, st>>>(edge_index, edge_index + ne, w, dis, eid_dst,
                                                    eid_src, ne, coef_dst, coef_src);
    CGNN_CHECK_LAUNCH();
  }
  return CGNN_OK;
}

int cgnn_sage_norm(const int64_t* edge_index, const float* w, int64_t nn, int64_t ne,
                   const int32_t* rowptr_dst, const int32_t* eid_dst, const int32_t* rowptr_src,
                   const int32_t* eid_src, float* den, float* w_dst, float* coef_src_bwd,
                   void* stream) {
  if (nn < 0 || ne < 0) return CGNN_EINVAL;
  if (nn > 0 && (!rowptr_dst || !den)) return CGNN_EINVAL;
  if (ne > 0 && (!edge_index || !w || !eid_dst || !w_dst || (coef_src_bwd && !eid_src)))
    return CGNN_EINVAL;
  (void)rowptr_src;
  hipStream_t st = cgnn_stream(stream);
  if (nn > 0) {
    k_sage_den<<<blocks_for(nn, 16), 256, 0, st>>>(w, rowptr_dst, eid_dst, nn, den, w_dst);
    CGNN_CHECK_LAUNCH();
  }
  if (ne > 0 && coef_src_bwd) {
    k_sage_coef_bwd<<<blocks_for(ne, 256), 256, 0, st>>>(edge_index + ne, w, den, eid_src, ne,
                                                         coef_src_bwd);
    CGNN_CHECK_LAUNCH();
  }
  return CGNN_OK;
}

int cgnn_bell_plan(const int32_t* tile_ptr, const int32_t* tile_blk, int32_t num_tiles,
                   int32_t num_blocks, const int32_t* rowptr, int32_t* blk_off, int32_t* scratch, int64_t scratch_bytes,
                   void* stream) {
  if (num_tiles < 0 || num_blocks < 0 || !blk_off || !scratch) return CGNN_EINVAL;
  if (num_tiles > 0 && (!tile_ptr || !tile_blk || !rowptr)) return CGNN_EINVAL;
  CGNN_NEED_BYTES(scratch, scratch_bytes,
                  (((int64_t)num_blocks + 1 + kScanTile - 1) / kScanTile) * (int64_t)sizeof(int32_t));
  hipStream_t st = cgnn_stream(stream);
  k_zero_i32<<<blocks_for(num_blocks + 1, 256), 256, 0, st>>>(blk_off, num_blocks + 1);
  if (num_tiles > 0) k_bell_width<<<num_tiles, 64, 0, st>>>(tile_ptr, tile_blk, num_tiles, rowptr, blk_off);
  CGNN_CHECK_LAUNCH();
  return scan_i32(blk_off, (int64_t)num_blocks + 1, scratch, st);
}

int cgnn_bell_fill(const int32_t* tile_ptr, const int32_t* tile_blk, int32_t num_tiles,
                   const int32_t* rowptr, const int32_t* col, const int32_t* eid,
                   const float* edge_weight, float self_weight, const int32_t* blk_off,
                   void* entries, void* stream) {
  if (num_tiles < 0) return CGNN_EINVAL;
  if (num_tiles == 0) return CGNN_OK;
  if (!tile_ptr || !tile_blk || !rowptr || !blk_off || !entries) return CGNN_EINVAL;
  k_bell_fill<<<num_tiles, 256, 0, cgnn_stream(stream)>>>(tile_ptr, tile_blk, num_tiles, rowptr, col,
                                                         eid, edge_weight, self_weight, blk_off,
                                                         static_cast<uint2*>(entries));
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gather_rows(const cgnn_gather_jobs* jobs, const int64_t* ids, int32_t num_ids,
                     const int64_t* ids_offset, void* stream) {
  if (!jobs || jobs->n < 0 || jobs->n > CGNN_GATHER_MAX_JOBS || num_ids < 0) return CGNN_EINVAL;
  if (jobs->n == 0 || num_ids == 0) return CGNN_OK;
  if (!ids) return CGNN_EINVAL;
  for (int j = 0; j < jobs->n; ++j)
    if (!jobs->src[j] || !jobs->dst[j] || jobs->row_bytes[j] <= 0 || (jobs->row_bytes[j] & 3) ||
        ((reinterpret_cast<uintptr_t>(jobs->src[j]) | reinterpret_cast<uintptr_t>(jobs->dst[j])) & 3))
      return CGNN_EINVAL;
  const int gx = num_ids < 2048 ? num_ids : 2048;
  k_gather_rows<<<dim3(gx, jobs->n), 256, 0, cgnn_stream(stream)>>>(*jobs, ids, num_ids, ids_offset);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_epoch_advance(int64_t* cursor, int64_t step, const float* loss, float weight, float* tally,
                       void* stream) {
  if (!cursor || (tally && !loss)) return CGNN_EINVAL;
  k_epoch_advance<<<1, 1, 0, cgnn_stream(stream)>>>(cursor, step, loss, weight, tally);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gather_f32(const float* src, const int32_t* idx, int64_t n, float* out, void* stream) {
  if (n < 0) return CGNN_EINVAL;
  if (n == 0) return CGNN_OK;
  if (!src || !idx || !out) return CGNN_EINVAL;
  k_gather_f32<<<blocks_for(n, 256), 256, 0, cgnn_stream(stream)>>>(src, idx, n, out);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_gcn_dis(const float* w_src, const int32_t* rowptr_src, int64_t nn, float* dis,
                 void* stream) {
  if (nn < 0) return CGNN_EINVAL;
  if (nn == 0) return CGNN_OK;
  if (!rowptr_src || !dis) return CGNN_EINVAL;
  k_gcn_dis<<<blocks_for(nn, 256), 256, 0, cgnn_stream(stream)>>>(w_src, rowptr_src, nn, dis);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
