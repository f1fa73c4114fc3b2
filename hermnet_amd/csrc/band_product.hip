// Batched products with one 32-wide side: rbf_proj of the TRAINING path on the bucketed basis (hermnet_amd/trainops.py:
// BucketedBasis -- /root/reference/HermNet/rmnet.py:55 on 32-centre windows, differentiated twice by the step of
// /root/reference/example/dist_train.py:86-99).  Per chunk c of C sorted edges (A = the basis window, B = the weight window):
//
//   P   out[c]  = A[c] B[c] + bias[c]                    A [nc,C,32]   B [nc,32,N]   bias [nc,N] | null     out [nc,C,N]
//   Q   gA[c]   = (g1[c] + g2[c]) B[c]^T                 g1, g2 [nc,C,N] (g2 may be null)                    gA  [nc,C,32]
//   S   gB[c]   = A[c]^T (g1[c] + g2[c]),  gbias[c] = the column sums of g1[c] + g2[c]                       gB  [nc,32,N], gbias [nc,N] | null
//
// The three are closed under differentiation (trainops.py: BandP / BandQ / BandS), so the step's first and second order
// passes run on them alone.  The second addend exists because the [E,3H] radial array has two consumers in the autograd graph
// (the message algebra and its backward): their two gradients arrive separately and are added while they are read, instead of
// by an elementwise launch over 610 MB per layer.
//
// fp32 values on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation, a fixed summation order).  The [nc,C,N] side
// is 610 MB per launch at configs[4]'s batch (397,312 padded edges, N = 384): HBM-streaming kernels whose matrix work -- 64
// FLOP per streamed float -- sits just under the fp32 matrix peak (62 us of pipe time against 110 us of HBM time at 5.5 TB/s).
//
// Lane conventions of the 32x32x2 product D = X Y (X [32,2], Y [2,32]): lane l supplies X[l & 31][l >> 5] and Y[l >> 5][l & 31] and
// holds D[8 g + 4 (l >> 5) + e][l & 31] in register 4 g + e.  The reduction index is free, so a lane that holds four
// consecutive reduction entries (one 16-byte load) spends them on four products: step e pairs entry 4 h + e of both halves h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hermnet_hip.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

#ifndef HN_P_TI
#define HN_P_TI 3      // column tiles of the product kernel in flight per wave (measured: 1: 215 us, 2: 198, 3: 187, 4: 192)
#endif
#define HN_LAUNCH_END return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH

__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f16 mma(float x, float y, f16 acc) { return __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0); }

// The chunk's weight window [32][N] -> LDS rows of `ld` floats.  Four 16-byte loads per thread are in flight before the first LDS
// write (a load-then-write loop waits out one memory latency per iteration: 12 of them at N = 384, ~25 us per workgroup).
template <int THREADS>
__device__ __forceinline__ void stage_weights(float* Bl, const float* Bc, int N, int ld, int tid) {
  const int n4 = N / 4, total = 8 * N;
  for (int i0 = tid; i0 < total; i0 += 4 * THREADS) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * THREADS;
      if (i < total) v[u] = *reinterpret_cast<const f4*>(Bc + 4 * (size_t)i);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * THREADS;
      if (i < total) *reinterpret_cast<f4*>(Bl + (i / n4) * ld + 4 * (i % n4)) = v[u];
    }
  }
}

template <int TI>
__device__ __forceinline__ void p_tiles(const f4 (&a)[4], const float* Bl, const float* bc, float* o, int N, int t, int j, int h) {
  f16 acc[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const float b0 = bc ? bc[(t + i) * 32 + j] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[i][v] = b0;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TI; ++i) acc[i] = mma(a[q][e], Bl[(8 * q + 4 * h + e) * N + (t + i) * 32 + j], acc[i]);
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __builtin_nontemporal_store(acc[i][4 * g + e], o + (size_t)(8 * g + e) * N + (t + i) * 32);
      }
}

// ---- P: one workgroup = `rows_wg` rows of one chunk (the chunk's weight window in LDS), a wave = 32 rows x all columns --------
__global__ __launch_bounds__(256) void band_p_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                     const float* __restrict__ bias, float* __restrict__ out, int C, int N,
                                                     int rows_wg) {
  extern __shared__ __align__(16) float Bl[];                       // [32][N]
  const int parts = C / rows_wg, c = blockIdx.x / parts, part = blockIdx.x % parts;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const float* Bc = B + (size_t)c * 32 * N;
  // (a row block's basis rows are requested one block ahead -- the first before the weights are staged: fetched at the head of
  // their own block, its products wait out the trip to memory)
  const size_t wg_row0 = (size_t)c * C + (size_t)part * rows_wg;
  f4 an[4] = {};
  if (wave < rows_wg / 32) {                            // (a workgroup of fewer than 128 rows leaves waves without a block)
#pragma unroll
    for (int q = 0; q < 4; ++q) an[q] = ld4(A + (wg_row0 + wave * 32 + j) * 32 + 8 * q + 4 * h);
  }
  stage_weights<256>(Bl, Bc, N, N, tid);
  __syncthreads();
  const float* bc = bias ? bias + (size_t)c * N : nullptr;
  const int t0 = 0, t1 = N / 32;
  for (int rb = wave; rb < rows_wg / 32; rb += 4) {
    const size_t row0 = wg_row0 + rb * 32;
    f4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = an[q];
    if (rb + 4 < rows_wg / 32) {
#pragma unroll
      for (int q = 0; q < 4; ++q) an[q] = ld4(A + (row0 + 128 + j) * 32 + 8 * q + 4 * h);
    }
    float* o = out + (row0 + 4 * h) * N + j;
    // HN_P_TI column tiles at a time: their products interleave, so that no product waits for the one it accumulates onto
    int t = t0;
    for (; t + HN_P_TI <= t1; t += HN_P_TI) p_tiles<HN_P_TI>(a, Bl, bc, o, N, t, j, h);
    for (; t < t1; ++t) p_tiles<1>(a, Bl, bc, o, N, t, j, h);
  }
}

// ---- Q: one workgroup = `rows_wg` rows of one chunk, a wave = 32 rows; the reduction runs over the N columns ------------------
template <bool TWO>
__global__ __launch_bounds__(256) void band_q_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                     const float* __restrict__ B, float* __restrict__ gA, int C, int N,
                                                     int rows_wg) {
  extern __shared__ __align__(16) float Bl[];                       // [32][N + 4]: 16-byte reads of 8 lanes hit 32 banks
  const int LD = N + 4;
  const int parts = C / rows_wg, c = blockIdx.x / parts, part = blockIdx.x % parts;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const float* Bc = B + (size_t)c * 32 * N;
  stage_weights<256>(Bl, Bc, N, LD, tid);
  __syncthreads();
  const float* bl = Bl + j * LD + 4 * h;
  constexpr int U = 4;                                               // 16-byte loads in flight per lane and operand
  for (int rb = wave; rb < rows_wg / 32; rb += 4) {
    const size_t row0 = (size_t)c * C + (size_t)part * rows_wg + rb * 32;
    const float* p1 = g1 + (row0 + j) * N + 4 * h;
    const float* p2 = TWO ? g2 + (row0 + j) * N + 4 * h : nullptr;
    f16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    f4 cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cur[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p1 + 8 * u));
      if (TWO) cur[u] += __builtin_nontemporal_load(reinterpret_cast<const f4*>(p2 + 8 * u));
    }
    const int nq = N / 8;
    for (int q0 = 0; q0 < nq; q0 += U) {
      if (q0 + U < nq) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          nxt[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p1 + 8 * (q0 + U + u)));
          if (TWO) nxt[u] += __builtin_nontemporal_load(reinterpret_cast<const f4*>(p2 + 8 * (q0 + U + u)));
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const f4 b = *reinterpret_cast<const f4*>(bl + 8 * (q0 + u));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mma(cur[u][e], b[e], acc);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    float* o = gA + (row0 + 4 * h) * 32 + j;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[(8 * g + e) * 32] = acc[4 * g + e];
  }
}

// ---- S: one wave = one 32-column tile of one chunk, all C rows in pairs; 4 tiles per workgroup ------------------------------
template <bool TWO>
__global__ __launch_bounds__(256) void band_s_kernel(const float* __restrict__ A, const float* __restrict__ g1,
                                                     const float* __restrict__ g2, float* __restrict__ gB,
                                                     float* __restrict__ gbias, int C, int N) {
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const int tile = blockIdx.y * 4 + wave;
  if (tile * 32 >= N) return;
  const float* a = A + ((size_t)c * C + h) * 32 + j;
  const float* p1 = g1 + ((size_t)c * C + h) * N + tile * 32 + j;
  const float* p2 = TWO ? g2 + ((size_t)c * C + h) * N + tile * 32 + j : nullptr;
  f16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  float colsum = 0.f;
  constexpr int U = 8;                                               // row pairs per round: 8 + 8 (+ 8) loads in flight
  for (int s0 = 0; s0 < C / 2; s0 += U) {
    float av[U], gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      av[u] = a[(size_t)(2 * (s0 + u)) * 32];
      gv[u] = __builtin_nontemporal_load(p1 + (size_t)(2 * (s0 + u)) * N);
      if (TWO) gv[u] += __builtin_nontemporal_load(p2 + (size_t)(2 * (s0 + u)) * N);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc = mma(av[u], gv[u], acc);
      colsum += gv[u];
    }
  }
  float* o = gB + ((size_t)c * 32 + 4 * h) * N + tile * 32 + j;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) o[(size_t)(8 * g + e) * N] = acc[4 * g + e];
  if (gbias) {
    colsum += __shfl_xor(colsum, 32, 64);
    if (h == 0) gbias[(size_t)c * N + tile * 32 + j] = colsum;
  }
}

// ---- Q and S from ONE pass over g1 (+ g2): the pass of the step that differentiates the parameters needs both -----------------
// One workgroup = one chunk.  The chunk streams through LDS in tiles of 64 rows x 128 columns (g1 + g2 summed on the way in);
// a tile is read twice from LDS, once per product, in the lane pattern each one needs:
//   Q  wave w owns row block (w & 1) of the slab and column half (w >> 1) of the tile; accumulates over the N / 128 tiles of
//      the slab, the two column halves are added at the slab's end (a fixed order) and 32 x 32 results stored
//   S  wave w owns column tile w of the tile, all 64 rows; one accumulator per column panel, kept over the whole chunk
// 64 products per wave and tile against 32 KB (64 KB with two addends) of HBM traffic: the same ratio as the single kernels,
// with half of their traffic.  LDS: tile [64][132] + weight panel [32][132] + basis slab [64][32] = 58.9 KB, two workgroups
// per CU.  DO_S = false: Q alone with the same staging (the 16-byte loads of band_q_kernel's lanes walk 32 rows at once).
constexpr int kLG = 132;
template <int NP, bool TWO, bool DO_S>      // N = 128 NP
__global__ __launch_bounds__(256, 2) void band_qs_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         const float* __restrict__ g1, const float* __restrict__ g2,
                                                         float* __restrict__ gA, float* __restrict__ gB,
                                                         float* __restrict__ gbias, int C) {
  constexpr int N = 128 * NP;
  __shared__ __align__(16) float lds[64 * kLG + 32 * kLG + (DO_S ? 64 * 32 : 0)];
  float* Gt = lds;
  float* Bp = lds + 64 * kLG;
  float* As = Bp + 32 * kLG;
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const int rb = wave & 1, ch = wave >> 1;
  const float* Bc = B + (size_t)c * 32 * N;
  f16 accS[DO_S ? NP : 1];
  float colsum[DO_S ? NP : 1];
  if (DO_S) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      colsum[p] = 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) accS[p][v] = 0.f;
    }
  }
  // the tile a thread stages: 8 x 16 bytes, rows (tid >> 5) + 8 i, columns 4 (tid & 31) .. + 3
  // (a uniform chunk pointer + 32-bit offsets: one offset register per load, shared by both addends)
  const int lr = tid >> 5, lc = (tid & 31) * 4;
  const float* gc1 = g1 + (size_t)c * C * N;
  const float* gc2 = TWO ? g2 + (size_t)c * C * N : nullptr;
  f4 t1[8], t2[TWO ? 8 : 1];
  auto fetch = [&](int slab, int p) {
    const unsigned base = (unsigned)((slab * 64 + lr) * N + 128 * p + lc);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned off = base + (unsigned)(8 * i * N);
      t1[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(gc1 + off));
      if (TWO) t2[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(gc2 + off));
    }
  };
  const int slabs = C / 64;
  fetch(0, 0);
  for (int slab = 0; slab < slabs; ++slab) {
    const size_t row0 = (size_t)c * C + (size_t)slab * 64;
    f16 accQ;
#pragma unroll
    for (int v = 0; v < 16; ++v) accQ[v] = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      __syncthreads();                                   // the previous tile (and the slab's exchange) has been read
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<f4*>(Gt + (lr + 8 * i) * kLG + lc) = TWO ? t1[i] + t2[i] : t1[i];
#pragma unroll
      for (int i = 0; i < 4; ++i)                        // weight panel [32][128]: 1024 x 16 bytes
        *reinterpret_cast<f4*>(Bp + (lr + 8 * i) * kLG + lc) = ld4(Bc + (size_t)(lr + 8 * i) * N + 128 * p + lc);
      if (DO_S && p == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) reinterpret_cast<f4*>(As)[tid + 256 * i] = ld4(A + row0 * 32 + 4 * (tid + 256 * i));
      }
      __syncthreads();
      if (p + 1 < NP) fetch(slab, p + 1);
      else if (slab + 1 < slabs) fetch(slab + 1, 0);
      {                                                  // Q: 32 rows x 64 columns of reduction
        const float* gq = Gt + (32 * rb + j) * kLG + 64 * ch + 4 * h;
        const float* bq = Bp + j * kLG + 64 * ch + 4 * h;
#pragma unroll 4
        for (int q = 0; q < 8; ++q) {
          const f4 gv = *reinterpret_cast<const f4*>(gq + 8 * q), bv = *reinterpret_cast<const f4*>(bq + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) accQ = mma(gv[e], bv[e], accQ);
        }
      }
      if (DO_S) {                                        // S: 64 rows of reduction x 32 columns
        const float* as = As + h * 32 + j;
        const float* gs = Gt + h * kLG + 32 * wave + j;
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
          const float gv = gs[2 * s * kLG];
          accS[p] = mma(as[2 * s * 32], gv, accS[p]);
          colsum[p] += gv;
        }
      }
    }
    // the two column halves of Q: half 1 -> LDS -> added by half 0, stored
    __syncthreads();
    float* x = Gt + rb * (16 * 64);
    if (ch == 1) {
#pragma unroll
      for (int v = 0; v < 16; ++v) x[v * 64 + lane] = accQ[v];
    }
    __syncthreads();
    if (ch == 0) {
      float* o = gA + (row0 + 32 * rb + 4 * h) * 32 + j;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[(8 * g + e) * 32] = accQ[4 * g + e] + x[(4 * g + e) * 64 + lane];
    }
  }
  if (DO_S) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      float* o = gB + ((size_t)c * 32 + 4 * h) * N + 128 * p + 32 * wave + j;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[(size_t)(8 * g + e) * N] = accS[p][4 * g + e];
      if (gbias) {
        float cs = colsum[p];
        cs += __shfl_xor(cs, 32, 64);
        if (h == 0) gbias[(size_t)c * N + 128 * p + 32 * wave + j] = cs;
      }
    }
  }
}

template <int NP>
void launch_qs(const float* a, const float* b, const float* g1, const float* g2, long nc, int C, float* ga, float* gb, float* gbias,
               hipStream_t s) {
  const dim3 grid((unsigned)nc), block(256);
  if (gb) {
    if (g2) hipLaunchKernelGGL((band_qs_kernel<NP, true, true>), grid, block, 0, s, a, b, g1, g2, ga, gb, gbias, C);
    else hipLaunchKernelGGL((band_qs_kernel<NP, false, true>), grid, block, 0, s, a, b, g1, g2, ga, gb, gbias, C);
  } else {
    if (g2) hipLaunchKernelGGL((band_qs_kernel<NP, true, false>), grid, block, 0, s, a, b, g1, g2, ga, gb, gbias, C);
    else hipLaunchKernelGGL((band_qs_kernel<NP, false, false>), grid, block, 0, s, a, b, g1, g2, ga, gb, gbias, C);
  }
}
inline bool staged_shape(int C, int N) { return C % 64 == 0 && (N == 128 || N == 256 || N == 384); }

// ---- the basis window itself: phi[r][k] = w[c][k] e(u_r) exp(coeff (u_r - mu[c][k])^2), c = r / C the row's chunk ------------------
// (rmnet.py:168-172 on the chunk's 32 centres; e = the polynomial envelope 1 + a u^p + b u^(p+1) + c u^(p+2) for u < 1 on rows
// that hold an edge, 0 otherwise).  Left to torch the expression and its two derivatives are ~60 elementwise launches per step
// over [rows, 32] and [rows] arrays.  Eight lanes per row, a lane four centres.
//   order 0   phi [rows,32]                              from u [rows]
//   order 1   g_u [rows] = sum_k g[r][k] dphi/du          from g [rows,32], u
//   order 2   d_g [rows,32] = c_u[r] dphi/du,  d_u [rows] = c_u[r] sum_k g[r][k] d2phi/du2     from c_u [rows], g, u
struct BasisArgs {
  const float *u, *mu, *w, *g, *cu;
  const long* src;
  float *o0, *o1;
  long rows, num_edges;
  int C, p;
  float coeff;
};

template <int ORDER>
__global__ __launch_bounds__(256) void basis_window_kernel(BasisArgs a) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long r = i >> 3;
  if (r >= a.rows) return;
  const int l = (int)(i & 7);
  const long c = r / a.C;
  const float u = a.u[r];
  const bool alive = a.src[r] < a.num_edges && u < 1.0f;
  const int p = a.p;
  const float ca = -(p + 1) * (p + 2) * 0.5f, cb = (float)(p * (p + 2)), cc = -p * (p + 1) * 0.5f;
  float up2 = 1.0f;                                   // u^(p-2) (p >= 2)
  for (int k = 0; k < p - 2; ++k) up2 *= u;
  const float up1 = p >= 2 ? up2 * u : 1.0f, up0 = up1 * u;                    // u^(p-1), u^p   (p >= 1)
  const float e = alive ? 1.0f + up0 * (ca + u * (cb + u * cc)) : 0.0f;
  const float e1 = alive ? up1 * (ca * p + u * (cb * (p + 1) + u * cc * (p + 2))) : 0.0f;
  const float e2 = alive ? (p >= 2 ? ca * p * (p - 1) * up2 : 0.0f) + up1 * (cb * (p + 1) * p + u * cc * (p + 2) * (p + 1)) : 0.0f;
  const f4 mu = ld4(a.mu + c * 32 + 4 * l), w = ld4(a.w + c * 32 + 4 * l);
  f4 G, t;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    t[k] = u - mu[k];
    G[k] = w[k] * __expf(a.coeff * t[k] * t[k]);
  }
  const float c2 = 2.0f * a.coeff;
  if (ORDER == 0) {
    *reinterpret_cast<f4*>(a.o0 + r * 32 + 4 * l) = G * e;
    return;
  }
  const f4 d1 = G * (e1 + c2 * t * e);                 // dphi/du
  const f4 g = ld4(a.g + r * 32 + 4 * l);
  if (ORDER == 1) {
    const f4 v = g * d1;
    float sum = (v.x + v.y) + (v.z + v.w);
    sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
    if (l == 0) a.o0[r] = sum;
    return;
  }
  const float cu = a.cu[r];
  *reinterpret_cast<f4*>(a.o0 + r * 32 + 4 * l) = d1 * cu;
  const f4 d2 = G * (e2 + 2.0f * c2 * t * e1 + (c2 + c2 * c2 * t * t) * e);     // d2phi/du2
  const f4 v = g * d2;
  float sum = (v.x + v.y) + (v.z + v.w);
  sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
  if (l == 0) a.o1[r] = cu * sum;
}

inline int rows_per_workgroup(int C) { return C % 256 == 0 ? 256 : (C % 128 == 0 ? 128 : C); }
inline bool shape_ok(long nc, int C, int N) { return nc >= 0 && C > 0 && C % 32 == 0 && N > 0 && N % 32 == 0 && N <= 480 && nc * (C / 32) < (1l << 30); }

}  // namespace

extern "C" int hermnet_band_product_supported(int rows_per_chunk, int width) { return shape_ok(0, rows_per_chunk, width) ? 1 : 0; }

extern "C" int hermnet_band_product(const float* a, const float* b, const float* bias, long num_chunks, int rows_per_chunk,
                                    int width, float* out, void* stream) {
  if (!shape_ok(num_chunks, rows_per_chunk, width)) return HN_ERR_BAD_ARG;
  if (num_chunks == 0) return HN_OK;
  if (!a || !b || !out) return HN_ERR_BAD_ARG;
  const int rows_wg = rows_per_workgroup(rows_per_chunk);
  hipLaunchKernelGGL(band_p_kernel, dim3((unsigned)(num_chunks * (rows_per_chunk / rows_wg))), dim3(256),
                     (size_t)32 * width * sizeof(float), (hipStream_t)stream, a, b, bias, out, rows_per_chunk, width, rows_wg);
  HN_LAUNCH_END;
}

extern "C" int hermnet_band_product_grad_a(const float* g1, const float* g2, const float* b, long num_chunks,
                                           int rows_per_chunk, int width, float* ga, void* stream) {
  if (!shape_ok(num_chunks, rows_per_chunk, width)) return HN_ERR_BAD_ARG;
  if (num_chunks == 0) return HN_OK;
  if (!g1 || !b || !ga) return HN_ERR_BAD_ARG;
  if (staged_shape(rows_per_chunk, width)) {
    if (width == 384) launch_qs<3>(nullptr, b, g1, g2, num_chunks, rows_per_chunk, ga, nullptr, nullptr, (hipStream_t)stream);
    else if (width == 256) launch_qs<2>(nullptr, b, g1, g2, num_chunks, rows_per_chunk, ga, nullptr, nullptr, (hipStream_t)stream);
    else launch_qs<1>(nullptr, b, g1, g2, num_chunks, rows_per_chunk, ga, nullptr, nullptr, (hipStream_t)stream);
    HN_LAUNCH_END;
  }
  const int rows_wg = rows_per_workgroup(rows_per_chunk);
  const dim3 grid((unsigned)(num_chunks * (rows_per_chunk / rows_wg)));
  const size_t lds = (size_t)32 * (width + 4) * sizeof(float);
  if (g2) hipLaunchKernelGGL(band_q_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, g1, g2, b, ga, rows_per_chunk, width, rows_wg);
  else hipLaunchKernelGGL(band_q_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, g1, g2, b, ga, rows_per_chunk, width, rows_wg);
  HN_LAUNCH_END;
}

extern "C" int hermnet_band_product_grad_b(const float* a, const float* g1, const float* g2, long num_chunks,
                                           int rows_per_chunk, int width, float* gb, float* gbias, void* stream) {
  if (!shape_ok(num_chunks, rows_per_chunk, width) || rows_per_chunk % 16) return HN_ERR_BAD_ARG;
  if (num_chunks == 0) return HN_OK;
  if (!a || !g1 || !gb) return HN_ERR_BAD_ARG;
  const dim3 grid((unsigned)num_chunks, (unsigned)((width / 32 + 3) / 4));
  if (g2) hipLaunchKernelGGL(band_s_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, g1, g2, gb, gbias, rows_per_chunk, width);
  else hipLaunchKernelGGL(band_s_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, g1, g2, gb, gbias, rows_per_chunk, width);
  HN_LAUNCH_END;
}

extern "C" int hermnet_band_product_grads(const float* a, const float* b, const float* g1, const float* g2, long num_chunks,
                                          int rows_per_chunk, int width, float* ga, float* gb, float* gbias, void* stream) {
  if (!shape_ok(num_chunks, rows_per_chunk, width) || rows_per_chunk % 16) return HN_ERR_BAD_ARG;
  if (num_chunks == 0) return HN_OK;
  if (!a || !b || !g1 || !ga || !gb) return HN_ERR_BAD_ARG;
  if (!staged_shape(rows_per_chunk, width)) {
    const int rc = hermnet_band_product_grad_a(g1, g2, b, num_chunks, rows_per_chunk, width, ga, stream);
    return rc != HN_OK ? rc : hermnet_band_product_grad_b(a, g1, g2, num_chunks, rows_per_chunk, width, gb, gbias, stream);
  }
  if (width == 384) launch_qs<3>(a, b, g1, g2, num_chunks, rows_per_chunk, ga, gb, gbias, (hipStream_t)stream);
  else if (width == 256) launch_qs<2>(a, b, g1, g2, num_chunks, rows_per_chunk, ga, gb, gbias, (hipStream_t)stream);
  else launch_qs<1>(a, b, g1, g2, num_chunks, rows_per_chunk, ga, gb, gbias, (hipStream_t)stream);
  HN_LAUNCH_END;
}

extern "C" int hermnet_basis_window(int order, const float* u, const long* src, long num_edges, const float* mu, const float* w,
                                    long num_chunks, int rows_per_chunk, float coeff, int env_p, const float* g, const float* cu,
                                    float* out0, float* out1, void* stream) {
  if (order < 0 || order > 2 || num_chunks < 0 || rows_per_chunk <= 0 || env_p < 1 || env_p > 64) return HN_ERR_BAD_ARG;
  if (num_chunks == 0) return HN_OK;
  if (!u || !src || !mu || !w || !out0 || (order >= 1 && !g) || (order == 2 && (!cu || !out1))) return HN_ERR_BAD_ARG;
  BasisArgs a = {u, mu, w, g, cu, src, out0, out1, num_chunks * rows_per_chunk, num_edges, rows_per_chunk, env_p, coeff};
  const dim3 grid((unsigned)((a.rows * 8 + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  if (order == 0) hipLaunchKernelGGL(basis_window_kernel<0>, grid, dim3(256), 0, s, a);
  else if (order == 1) hipLaunchKernelGGL(basis_window_kernel<1>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(basis_window_kernel<2>, grid, dim3(256), 0, s, a);
  HN_LAUNCH_END;
}
