// gfx950 fp32 MFMA GEMM for the node-level feature-mixing linears of the hot path (reference: rmnet.py:52
// `x_proj`, rmnet.py:94-107 `vec_proj` / `xvec_proj`, and their input gradients), with the elementwise stages that
// surround them fused into the operand load and the result store:
//
//     C[M, N] = epilogue( prologue(A)[M, K] . Bt[N, K]^T )
//
// Both operands are K-contiguous (nn.Linear keeps weights [out, in]; the host keeps transposed copies for the backward
// products), so every product of the layer has this one form.  Shapes here are skinny: K = H, 2H or 3H (128..384) and
// N = H..3H with M = thousands of rows, batched over the relation blocks -- a regime where the library's generic tiles
// reach ~45 % of the fp32 MFMA rate (profiles/r02_v1_counters.json: mfma_busy_share of the Cijk_* kernels).
//
// Tile: 64 x 64 outputs per 256-thread workgroup, K in chunks of KC (64 or 128) staged in LDS (rows padded by 4 floats:
// conflict-free ds_read_b128 at consecutive rows); each of the 4 waves owns a 32 x 32 block and issues
// v_mfma_f32_32x32x2_f32.  A lane reads 4 consecutive k of its row per ds_read_b128 and feeds 4 MFMAs with them; the
// k-assignment (lanes 0-31: k = 8q..8q+3, lanes 32-63: k = 8q+4..8q+7) is the same for A and Bt, which is all the
// instruction needs.  64 KB (KC = 128) or 34 KB of LDS per workgroup: two or four workgroups per CU overlap one
// workgroup's loads with another's matrix work.
//
// Prologues (applied while the A tile is staged):   none | ScaledSiLU(A + pbias[k])        (rmnet.py:110-117)
// Epilogues (applied to the accumulators):          store (+ bias[n]) | multiply by ScaledSiLU'(E + ebias[n]) (the
//                                                   activation's backward) | accumulate into C
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hermnet_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, PADK = 4;

struct GemmArgs {
  const float* A; long lda; long strideA;
  const float* Bt; long ldb; long strideB;
  float* C; long ldc; long strideC;
  int M, N, K;
  const float* bias; long strideBias;        // epilogue 0: + bias[n]
  const float* pbias; long stridePbias;      // prologue 1: ScaledSiLU(A + pbias[k])
  const float* E; long lde; long strideE;    // epilogue 1: * ScaledSiLU'(E[m, n] + ebias[n])
  const float* ebias; long strideEbias;
};

__device__ __forceinline__ float ssilu(float x) {
  const float s = 1.0f / (1.0f + __expf(-x));
  return x * s * (1.0f / 0.6f);
}
__device__ __forceinline__ float dssilu(float x) {
  const float s = 1.0f / (1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s)) * (1.0f / 0.6f);
}

template <int KC, int PRO, int EPI>
__global__ __launch_bounds__(256) void node_gemm_kernel(GemmArgs g) {
  extern __shared__ __align__(16) float lds[];
  constexpr int LD = KC + PADK;
  float* As = lds;                 // [BM][LD]
  float* Bs = lds + BM * LD;       // [BN][LD]

  const int b = blockIdx.z;
  const float* A = g.A + (long)b * g.strideA;
  const float* Bt = g.Bt + (long)b * g.strideB;
  float* C = g.C + (long)b * g.strideC;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;       // this wave's 32 x 32 block inside the tile
  const int r = lane & 31, hk = (lane >> 5) * 4;               // operand row, k offset of this lane half

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // K chunks are software-pipelined through registers: the global loads of chunk c+1 are issued before the matrix
  // work of chunk c and land in LDS after it, so only the first chunk's round trip is exposed.
  constexpr int V = KC / 4;                                  // float4 per row
  constexpr int IT = BM * V / 256;                           // float4 per thread and operand
  float4 va[IT], vb[IT];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / V, c4 = idx % V;
      va[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      vb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + row < g.M) va[it] = *reinterpret_cast<const float4*>(A + (long)(m0 + row) * g.lda + kc + c4 * 4);
      if (n0 + row < g.N) vb[it] = *reinterpret_cast<const float4*>(Bt + (long)(n0 + row) * g.ldb + kc + c4 * 4);
    }
  };
  auto commit = [&](int kc) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / V, c4 = idx % V;
      float4 v = va[it];
      if (PRO == 1 && m0 + row < g.M) {
        float4 pb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.pbias) pb = *reinterpret_cast<const float4*>(g.pbias + (long)b * g.stridePbias + kc + c4 * 4);
        v = make_float4(ssilu(v.x + pb.x), ssilu(v.y + pb.y), ssilu(v.z + pb.z), ssilu(v.w + pb.w));
      }
      *reinterpret_cast<float4*>(As + row * LD + c4 * 4) = v;
      *reinterpret_cast<float4*>(Bs + row * LD + c4 * 4) = vb[it];
    }
  };
  fetch(0);
  for (int kc = 0; kc < g.K; kc += KC) {
    if (kc > 0) __syncthreads();                             // the previous chunk's tiles are consumed
    commit(kc);
    __syncthreads();
    if (kc + KC < g.K) fetch(kc + KC);                       // in flight during the matrix work below
    // ---- 32 x 32 x KC on the matrix pipe
    const float* ap = As + (wm + r) * LD + hk;
    const float* bp = Bs + (wn + r) * LD + hk;
#pragma unroll 4
    for (int q = 0; q < KC / 8; ++q) {
      const float4 a4 = *reinterpret_cast<const float4*>(ap + q * 8);
      const float4 b4 = *reinterpret_cast<const float4*>(bp + q * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
  }
  // ---- epilogue.  Accumulator layout (32x32): col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
  const int col = n0 + wn + (lane & 31);
  if (col >= g.N) return;
  float bcol = 0.f, ebcol = 0.f;
  if (EPI == 0 && g.bias) bcol = g.bias[(long)b * g.strideBias + col];
  if (EPI == 1 && g.ebias) ebcol = g.ebias[(long)b * g.strideEbias + col];
  const float* E = EPI == 1 ? g.E + (long)b * g.strideE : nullptr;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = m0 + wm + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    if (row >= g.M) continue;
    float v = acc[i];
    float* cp = C + (long)row * g.ldc + col;
    if (EPI == 0) v += bcol;
    else if (EPI == 1) v *= dssilu(E[(long)row * g.lde + col] + ebcol);
    else v += *cp;
    *cp = v;
  }
}

typedef void (*gemm_kern_t)(GemmArgs);

template <int KC>
gemm_kern_t pick(int pro, int epi) {
  if (pro == 0) return epi == 0 ? node_gemm_kernel<KC, 0, 0> : (epi == 1 ? node_gemm_kernel<KC, 0, 1> : node_gemm_kernel<KC, 0, 2>);
  return epi == 0 ? node_gemm_kernel<KC, 1, 0> : (epi == 1 ? node_gemm_kernel<KC, 1, 1> : node_gemm_kernel<KC, 1, 2>);
}

}  // namespace

extern "C" int hermnet_node_gemm(const float* A, long lda, long strideA, const float* Bt, long ldb, long strideB,
                                 float* C, long ldc, long strideC, int M, int N, int K, int batch,
                                 int prologue, const float* pbias, long stridePbias,
                                 int epilogue, const float* bias_or_ebias, long strideBias,
                                 const float* E, long lde, long strideE, void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || batch < 0 || (K & 63) || prologue < 0 || prologue > 1 || epilogue < 0 || epilogue > 2)
    return HN_ERR_BAD_ARG;
  if (M == 0 || batch == 0) return HN_OK;
  if (!A || !Bt || !C || (lda & 3) || (ldb & 3) || (epilogue == 1 && !E)) return HN_ERR_BAD_ARG;
  GemmArgs g = {};
  g.A = A; g.lda = lda; g.strideA = strideA; g.Bt = Bt; g.ldb = ldb; g.strideB = strideB;
  g.C = C; g.ldc = ldc; g.strideC = strideC; g.M = M; g.N = N; g.K = K;
  g.pbias = pbias; g.stridePbias = stridePbias;
  if (epilogue == 0) { g.bias = bias_or_ebias; g.strideBias = strideBias; }
  else if (epilogue == 1) { g.ebias = bias_or_ebias; g.strideEbias = strideBias; g.E = E; g.lde = lde; g.strideE = strideE; }
  const int KC = (K % 128 == 0) ? 128 : 64;
  gemm_kern_t k = KC == 128 ? pick<128>(prologue, epilogue) : pick<64>(prologue, epilogue);
  const size_t lds = (size_t)(BM + BN) * (KC + PADK) * sizeof(float);
  // > 64 KB of dynamic LDS needs the opt-in once per kernel
  static gemm_kern_t done[16];
  static int ndone = 0;
  bool seen = false;
  for (int i = 0; i < ndone; ++i) seen |= done[i] == k;
  if (!seen) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return HN_ERR_LDS;
    if (ndone < 16) done[ndone++] = k;
  }
  dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN), (unsigned)batch);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, (hipStream_t)stream, g);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}
