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

#define HN_LAUNCH_END return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH

__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f16 mma(float x, float y, f16 acc) { return __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0); }

// ---- P: one workgroup = `rows_wg` rows of one chunk (the chunk's weight window in LDS), a wave = 32 rows x all columns --------
__global__ __launch_bounds__(256) void band_p_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                     const float* __restrict__ bias, float* __restrict__ out, int C, int N,
                                                     int rows_wg) {
  extern __shared__ __align__(16) float Bl[];                       // [32][N]
  const int parts = C / rows_wg, c = blockIdx.x / parts, part = blockIdx.x % parts;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  const float* Bc = B + (size_t)c * 32 * N;
  for (int i = tid; i < 8 * N; i += 256) reinterpret_cast<f4*>(Bl)[i] = ld4(Bc + 4 * i);
  __syncthreads();
  const float* bc = bias ? bias + (size_t)c * N : nullptr;
  for (int rb = wave; rb < rows_wg / 32; rb += 4) {
    const size_t row0 = (size_t)c * C + (size_t)part * rows_wg + rb * 32;
    f4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = ld4(A + (row0 + j) * 32 + 8 * q + 4 * h);
    float* o = out + (row0 + 4 * h) * N + j;
    for (int t = 0; t < N / 32; ++t) {
      const float b0 = bc ? bc[t * 32 + j] : 0.f;
      f16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = b0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mma(a[q][e], Bl[(8 * q + 4 * h + e) * N + t * 32 + j], acc);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) __builtin_nontemporal_store(acc[4 * g + e], o + (size_t)(8 * g + e) * N + t * 32);
    }
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
  for (int i = tid; i < 8 * N; i += 256) {
    const int k = i / (N / 4), n4 = i % (N / 4);
    *reinterpret_cast<f4*>(Bl + k * LD + 4 * n4) = ld4(Bc + 4 * i);
  }
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
