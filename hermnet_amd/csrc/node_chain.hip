// gfx950: the node-level dense algebra of one HeteroVertexConv layer as FOUR chain kernels on the fp32 matrix pipe.
//
//   node_pre_fwd     x -> LayerNorm -> [H -> H] -> ScaledSiLU -> [H -> 3H]  = xh[t]        /root/reference/HermNet/rmnet.py:52
//   node_pre_bwd     gxh[t] -> [3H -> H] -> * ScaledSiLU' -> [H -> H]       = gn[t]        (its input gradient)
//   node_update_fwd  (x1, vec1) -> vec_proj, vec_dot, |v2|, xvec_proj MLP, dx / dvec, residual   rmnet.py:94-107, 29-31
//   node_update_bwd  (gx_out, gvec_out) -> (gx1, gvec1)                                         (its input gradient)
//
// Why one kernel per chain: the layer's linears are skinny (K = H .. 3H, thousands of rows); as separate GEMM launches
// every hidden activation makes a round trip through HBM, every launch quantises the chip on its own, and the library's
// generic tiles reach ~0.45 of the fp32 MFMA rate on these shapes (profiles/r02_v6_kernel_stats.csv).  Here a workgroup
// owns a tile of TR rows for the WHOLE chain: hidden activations live in an LDS tile (the A operand of the next product),
// elementwise stages run on the accumulators, and only what the backward needs is stored.
//
// Data flow of one product  C[TR, Nout] = A[TR, K] . W[Nout, K]^T :
//   * A: an LDS tile, rows padded by 4 floats; lane l reads row (l & 31), k = 8q + 4 (l >> 5) .. +3 with one conflict-free
//     ds_read_b128 and feeds four v_mfma_f32_32x32x2_f32 with it;
//   * W: never staged in LDS.  The host keeps every weight in FRAGMENT ORDER, wf[(cb * K/8 + q) * 64 + l] = the float4
//     W[32 cb + (l & 31)][8q + 4 (l >> 5) .. +3], i.e. exactly the B operand registers of lane l: one coalesced 1 KiB
//     global_load_dwordx4 per wave, straight from L2 (a layer's weights are < 1 MB) into the MFMA operand, two k-groups
//     ahead of use.  Waves of a workgroup split the OUTPUT COLUMNS, so no weight byte is loaded twice per workgroup;
//   * C: 32 x 32 accumulator blocks; a wave owns the same 32-channel slices of every part of an output (s|a|b, v1|v2,
//     p|q|r), so products of parts (vec_dot, q * vdot, r * v1) are lane-local.
// Two workgroups per CU (<= 68 KB of LDS, <= 256 VGPRs): one's elementwise / staging phases run beside the other's
// matrix phases.  fp32 MFMA is exact fp32 (a k-ordered fmaf chain): results equal a library GEMM's to rounding.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hermnet_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kInvSqrt2 = 0.70710678118654752f;
constexpr float kSiluScale = 1.0f / 0.6f;

// (v_rcp_f32: 1 ulp; an IEEE division costs ten instructions per element in the epilogues)
__device__ __forceinline__ float sigmoid_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ssilu(float x) { return x * sigmoid_(x) * kSiluScale; }
__device__ __forceinline__ float dssilu(float x) {
  const float s = sigmoid_(x);
  return s * (1.0f + x * (1.0f - s)) * kSiluScale;
}
// An empty volatile asm that "reads and writes" x: the value must exist HERE (pure arithmetic is otherwise free to sink
// below barriers to its first use, which keeps every accumulator it reads alive until then and spills).
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
// Nothing moves across this point: the "memory" clobber orders the compiler's loads and stores (IR and selection DAG),
// the scheduling barrier the machine scheduler.
__device__ __forceinline__ void fence_sched() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// accumulator elements per epilogue batch (all global loads of a batch are requested before the first is consumed)
constexpr int G = 4;

// Geometry of a TR-row tile of width H on a 256-thread workgroup (4 waves).
template <int H_, int TR_>
struct Cfg {
  static constexpr int H = H_, TR = TR_;
  static constexpr int CB = H / 32;                 // 32-channel blocks
  static constexpr int WC = CB < 4 ? CB : 4;        // waves along the channels
  static constexpr int WR = 4 / WC;                 // waves along the rows
  static constexpr int CPW = CB / WC;               // channel blocks per wave
  static constexpr int RB = TR / 32 / WR;           // 32-row blocks per wave
  static constexpr int LD = H + 4;                  // LDS row stride of a [TR][H] tile
  static constexpr int F4 = TR * H / 4 / 256;       // float4 per thread of a cooperative [TR][H] tile copy
  static_assert(CB % WC == 0 && (TR / 32) % WR == 0 && RB >= 1 && F4 >= 1, "unsupported tile");
};

// ---- B operand: a ring of weight fragments, PF = 2 k-groups ahead ------------------------------------------------
template <int NJ, int RS>
struct BRing { f32x4 v[RS][NJ]; };

// first two k-groups of a stream (call it early: before the barrier / epilogue that precedes the product)
template <int NJ, int RS>
__device__ __forceinline__ void b_preload(BRing<NJ, RS>& r, const f32x4* const (&bp)[NJ]) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) { r.v[0][j] = bp[j][0]; r.v[1][j] = bp[j][64]; }
}

// acc[rb][j] += A[rows of block rb][k-groups 0 .. KP/8) . W_j, W_j streamed from bp[j] (this lane's pointer at group 0
// of the panel).  `As`: this lane's LDS read pointer, &tile[(first row of the wave + (l & 31)) * LD + 4 (l >> 5)].
// The ring holds groups 0 and 1 on entry.  MORE: the stream continues behind this panel (next panel of the same
// product): groups KP/8 and KP/8 + 1 are requested too and sit in slots 0 and 1 on exit (needs KP/8 % RS == 0).
template <int KP, int LD, int RB, int NJ, int RS, bool MORE>
__device__ __forceinline__ void mma_panel(f32x16 (&acc)[RB][NJ], const float* As, const f32x4* const (&bp)[NJ],
                                          BRing<NJ, RS>& ring) {
  constexpr int NQ = KP / 8;
  static_assert(NQ >= 2 && (!MORE || NQ % RS == 0), "panel / ring mismatch");
  f32x4 a[2][RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) a[0][rb] = *reinterpret_cast<const f32x4*>(As + rb * 32 * LD);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (MORE || q + 2 < NQ) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) ring.v[(q + 2) % RS][j] = bp[j][(q + 2) * 64];
    }
    if (q + 1 < NQ) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a[(q + 1) & 1][rb] = *reinterpret_cast<const f32x4*>(As + rb * 32 * LD + 8 * (q + 1));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[rb][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][rb][i], ring.v[q % RS][j][i], acc[rb][j], 0, 0, 0);
  }
}

template <int RB, int NJ>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[RB][NJ]) {
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][j][i] = 0.f;
}

// Per-tile buffer descriptors: every per-lane global access of a tile is `uniform base + 32-bit offset` through a raw
// buffer instruction whose range check replaces the row guards (rows past the tile's last valid row load 0 and drop
// their stores) -- no exec-mask branches and no 64-bit address arithmetic in the epilogues.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t tile_rsrc(const float* base, int valid_floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, valid_floats > 0 ? valid_floats * 4 : 0, 0x00020000);
}
__device__ __forceinline__ float bld(rsrc_t r, int off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off * 4, 0, 0));
}
__device__ __forceinline__ void bst(rsrc_t r, int off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off * 4, 0, 0);
}
__device__ __forceinline__ f32x4 bld4(rsrc_t r, int off) {
  const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, off * 4, 0, 0);
  return __builtin_bit_cast(f32x4, u);
}
__device__ __forceinline__ void bst4(rsrc_t r, int off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off * 4, 0, 0);
}

// accumulator element i of a 32 x 32 block: row offset inside the block
__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// cooperative copy of a [TR][W] tile (row stride `ld_src` floats in global memory) through registers
template <int TR, int W>
struct TileRegs { f32x4 v[TR * W / 4 / 256]; };

// `src`: descriptor of the tile's rows (valid range = nrows * ld_src floats), `off0`: float offset of the first column
template <int TR, int W>
__device__ __forceinline__ void tile_load(TileRegs<TR, W>& r, rsrc_t src, int ld_src, int off0, int tid) {
  constexpr int V = W / 4, F4 = TR * W / 4 / 256;
#pragma unroll
  for (int it = 0; it < F4; ++it) {
    const int idx = tid + it * 256, row = idx / V, c4 = idx % V;
    r.v[it] = bld4(src, row * ld_src + off0 + c4 * 4);
  }
}
template <int TR, int W, int LD>
__device__ __forceinline__ void tile_store(float* tile, const TileRegs<TR, W>& r, int tid) {
  constexpr int V = W / 4, F4 = TR * W / 4 / 256;
#pragma unroll
  for (int it = 0; it < F4; ++it) {
    const int idx = tid + it * 256, row = idx / V, c4 = idx % V;
    *reinterpret_cast<f32x4*>(tile + row * LD + c4 * 4) = r.v[it];
  }
}

// =====================================================================================================================
// node_pre_fwd: one workgroup = (TR source rows, relation t)
// =====================================================================================================================
struct PreFwdArgs {
  const float* x;      // [Ns, H]
  const float* w1f;    // [T] fragments of W1_t [H, H]   (LayerNorm affine folded in)
  const float* b1;     // [T, H]
  const float* w2f;    // [T] fragments of W2_t [3H, H]
  const float* b2;     // [T, 3H]
  float* hb;           // [T, Ns, H]   pre-activation incl. bias (saved for the backward)
  float* xh;           // [T, Ns, 3H]  incl. bias
  float* mean;         // [Ns]
  float* rstd;         // [Ns]
  int Ns, T, Hr;
  float eps;
};

template <int H, int TR>
__global__ __launch_bounds__(256, 2) void node_pre_fwd_kernel(PreFwdArgs a) {
  using C = Cfg<H, TR>;
  constexpr int LD = C::LD, RB = C::RB, CPW = C::CPW, CB = C::CB;
  extern __shared__ __align__(16) float tile[];           // [TR][LD]
  const int t = blockIdx.y, row0 = blockIdx.x * TR;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % C::WC, wr = wave / C::WC;
  const int nrows = min(TR, a.Ns - row0);
  const rsrc_t hb_r = tile_rsrc(a.hb + ((size_t)t * a.Ns + row0) * H, nrows * H);
  const rsrc_t xh_r = tile_rsrc(a.xh + ((size_t)t * a.Ns + row0) * 3 * H, nrows * 3 * H);

  // weight streams of this wave (requested before anything else: they do not depend on the rows)
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.w1f + (size_t)t * H * H) + lane;
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.w2f + (size_t)t * 3 * H * H) + lane;
  const f32x4* bp1[CPW];
  const f32x4* bp2[3 * CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) bp1[j] = w1 + (size_t)(wc * CPW + j) * (H / 8) * 64;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int j = 0; j < CPW; ++j) bp2[p * CPW + j] = w2 + (size_t)(p * CB + wc * CPW + j) * (H / 8) * 64;
  BRing<CPW, 3> r1;
  b_preload(r1, bp1);

  // ---- LayerNorm without affine (rmnet.py:52), a wave per row, statistics over the first Hr channels
  constexpr int EPL = H / 64;
  const float inv_hr = 1.0f / (float)a.Hr;
#pragma unroll 4
  for (int i = 0; i < TR / 4; ++i) {
    const int lr = wave * (TR / 4) + i, row = row0 + lr;
    float v[EPL], m[EPL];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int c = lane * EPL + e;
      m[e] = c < a.Hr ? 1.f : 0.f;
      v[e] = row < a.Ns ? a.x[(size_t)row * H + c] * m[e] : 0.f;
      s += v[e];
    }
    const float mu = wave_sum(s) * inv_hr;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { v[e] = (v[e] - mu) * m[e]; q = fmaf(v[e], v[e], q); }
    const float rs = rsqrtf(wave_sum(q) * inv_hr + a.eps);
#pragma unroll
    for (int e = 0; e < EPL; ++e) tile[lr * LD + lane * EPL + e] = v[e] * rs;
    if (t == 0 && lane == 0 && row < a.Ns) { a.mean[row] = mu; a.rstd[row] = rs; }
  }
  __syncthreads();

  // ---- h = n W1^T
  const float* As = tile + (wr * RB * 32 + (lane & 31)) * LD + 4 * (lane >> 5);
  f32x16 acc1[RB][CPW];
  zero_acc(acc1);
  mma_panel<H, LD, RB, CPW, 3, false>(acc1, As, bp1, r1);
  BRing<3 * CPW, 3> r2;
  b_preload(r2, bp2);
  __syncthreads();                                   // every wave has read n
  // ---- + b1, save, ScaledSiLU -> the tile becomes the A operand of the second product
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int col = (wc * CPW + j) * 32 + (lane & 31);
      const float bias = a.b1[(size_t)t * H + col];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
        const float hv = acc1[rb][j][i] + bias;
        bst(hb_r, lr * H + col, hv);
        tile[lr * LD + col] = ssilu(hv);
      }
    }
  __syncthreads();
  // ---- xh = a W2^T + b2
  f32x16 acc2[RB][3 * CPW];
  zero_acc(acc2);
  mma_panel<H, LD, RB, 3 * CPW, 3, false>(acc2, As, bp2, r2);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int jj = 0; jj < 3 * CPW; ++jj) {
      const int col = (jj / CPW) * H + (wc * CPW + jj % CPW) * 32 + (lane & 31);
      const float bias = a.b2[(size_t)t * 3 * H + col];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
        bst(xh_r, lr * 3 * H + col, acc2[rb][jj][i] + bias);
      }
    }
}

// =====================================================================================================================
// node_pre_bwd: one workgroup = (TR source rows, relation t) -> gn[t] (the sum over t and the LayerNorm backward follow
// in layernorm_bwd_parts_kernel)
// =====================================================================================================================
struct PreBwdArgs {
  const float* gxh;    // [T, Ns, 3H]
  const float* hb;     // [T, Ns, H]
  const float* w2tf;   // [T] fragments of W2_t^T [H, 3H]
  const float* w1tf;   // [T] fragments of W1_t^T [H, H]
  float* gn;           // [T, Ns, H]
  int Ns, T;
};

template <int H, int TR>
__global__ __launch_bounds__(256, 2) void node_pre_bwd_kernel(PreBwdArgs a) {
  using C = Cfg<H, TR>;
  constexpr int LD = C::LD, RB = C::RB, CPW = C::CPW;
  constexpr int KC = H < 128 ? H : 128;            // K chunk of the first product (K = 3H)
  constexpr int NCH = 3 * H / KC, LDC = KC + 4;
  static_assert(2 * TR * LDC >= TR * LD, "the gh tile must fit the two chunk buffers");
  extern __shared__ __align__(16) float lds[];            // 2 x [TR][LDC]
  const int t = blockIdx.y, row0 = blockIdx.x * TR;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % C::WC, wr = wave / C::WC;
  const int nrows = min(TR, a.Ns - row0);

  const f32x4* w2t = reinterpret_cast<const f32x4*>(a.w2tf + (size_t)t * 3 * H * H) + lane;
  const f32x4* w1t = reinterpret_cast<const f32x4*>(a.w1tf + (size_t)t * H * H) + lane;
  const f32x4* bpa[CPW];
  const f32x4* bpb[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    bpa[j] = w2t + (size_t)(wc * CPW + j) * (3 * H / 8) * 64;
    bpb[j] = w1t + (size_t)(wc * CPW + j) * (H / 8) * 64;
  }
  BRing<CPW, 4> ra;
  b_preload(ra, bpa);
  const rsrc_t gxh_r = tile_rsrc(a.gxh + ((size_t)t * a.Ns + row0) * 3 * H, nrows * 3 * H);
  const rsrc_t hb_r = tile_rsrc(a.hb + ((size_t)t * a.Ns + row0) * H, nrows * H);
  const rsrc_t gn_r = tile_rsrc(a.gn + ((size_t)t * a.Ns + row0) * H, nrows * H);
  TileRegs<TR, KC> regs;
  tile_load<TR, KC>(regs, gxh_r, 3 * H, 0, tid);
  // ScaledSiLU'(hb) of this lane's accumulator positions, requested now
  float dact[RB][CPW][16];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int col = (wc * CPW + j) * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
        dact[rb][j][i] = bld(hb_r, lr * H + col);
      }
    }

  // ---- ga = gxh W2   (K = 3H in chunks, double-buffered through registers)
  f32x16 acc[RB][CPW];
  zero_acc(acc);
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) {
    float* buf = lds + (kc & 1) * TR * LDC;
    tile_store<TR, KC, LDC>(buf, regs, tid);
    __syncthreads();
    if (kc + 1 < NCH) tile_load<TR, KC>(regs, gxh_r, 3 * H, (kc + 1) * KC, tid);
    const float* As = buf + (wr * RB * 32 + (lane & 31)) * LDC + 4 * (lane >> 5);
    const f32x4* bpk[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpk[j] = bpa[j] + (size_t)kc * (KC / 8) * 64;
    if (kc + 1 < NCH) mma_panel<KC, LDC, RB, CPW, 4, true>(acc, As, bpk, ra);
    else mma_panel<KC, LDC, RB, CPW, 4, false>(acc, As, bpk, ra);
  }
  BRing<CPW, 3> rb_;
  b_preload(rb_, bpb);
  __syncthreads();                                   // the chunk buffers are free
  // ---- gh = ga * ScaledSiLU'(hb) -> tile
  float* tile = lds;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int col = (wc * CPW + j) * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
        tile[lr * LD + col] = acc[rb][j][i] * dssilu(dact[rb][j][i]);
      }
    }
  __syncthreads();
  // ---- gn_t = gh W1
  f32x16 acc2[RB][CPW];
  zero_acc(acc2);
  const float* As = tile + (wr * RB * 32 + (lane & 31)) * LD + 4 * (lane >> 5);
  mma_panel<H, LD, RB, CPW, 3, false>(acc2, As, bpb, rb_);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int col = (wc * CPW + j) * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
        bst(gn_r, lr * H + col, acc2[rb][j][i]);
      }
    }
}

// gx = LayerNorm'(x)^T (sum_p g[p]) + add : the backward of the LayerNorm in front of the T relations' projections
__global__ __launch_bounds__(256) void layernorm_bwd_parts_kernel(const float* __restrict__ g, int nparts, long part_stride,
                                                                  const float* __restrict__ x, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const float* __restrict__ add,
                                                                  float* __restrict__ gx, int rows, int H, int Hr) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float mu = mean[r], rs = rstd[r];
  constexpr int KMAX = 4;                        // H <= 1024
  f32x4 gv[KMAX], nh[KMAX];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int c = (k * 64 + lane) * 4;
    gv[k] = nh[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c < H) {
      f32x4 gg = *reinterpret_cast<const f32x4*>(g + (size_t)r * H + c);
      for (int p = 1; p < nparts; ++p) gg += *reinterpret_cast<const f32x4*>(g + p * part_stride + (size_t)r * H + c);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)r * H + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float m = c + e < Hr ? 1.f : 0.f;
        gv[k][e] = gg[e] * m;
        nh[k][e] = (xv[e] - mu) * rs * m;
        s1 += gv[k][e];
        s2 = fmaf(gv[k][e], nh[k][e], s2);
      }
    }
  }
  const float m1 = wave_sum(s1) / (float)Hr, m2 = wave_sum(s2) / (float)Hr;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < H) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = c + e < Hr ? rs * (gv[k][e] - m1 - nh[k][e] * m2) : 0.f;
      if (add != nullptr) o += *reinterpret_cast<const f32x4*>(add + (size_t)r * H + c);
      *reinterpret_cast<f32x4*>(gx + (size_t)r * H + c) = o;
    }
  }
}

// =====================================================================================================================
// Tiles of the TARGET rows: relation blocks [type_rowptr[t], type_rowptr[t+1]) cut into TR-row tiles, then the rows of
// unknown elements [type_rowptr[T], N) (zero rows, hermnet.py:51).
// =====================================================================================================================
struct TileInfo { int t, row0, nrows; };

__device__ __forceinline__ TileInfo find_tile(const int* __restrict__ type_rowptr, int T, int N, int TR, int tile) {
  int first = 0;
  for (int t = 0; t < T; ++t) {
    const int lo = type_rowptr[t], hi = type_rowptr[t + 1];
    const int nt = (hi - lo + TR - 1) / TR;
    if (tile < first + nt) {
      const int row0 = lo + (tile - first) * TR;
      return {t, row0, min(TR, hi - row0)};
    }
    first += nt;
  }
  const int row0 = type_rowptr[T] + (tile - first) * TR;
  return {T, row0, min(TR, N - row0)};
}

// =====================================================================================================================
// node_update_fwd (rmnet.py:94-107 + the residual of rmnet.py:29-31 + the zero rows of hermnet.py:51,56-57)
// =====================================================================================================================
struct UpdFwdArgs {
  const float* x1;          // [N, H]
  const float* vec1;        // [N, 3, H]
  const float* wvf;         // [T] fragments of vec_proj.weight [2H, H]
  const float* wx0f;        // [T] fragments of xvec_proj[0].weight [H, 2H]
  const float* bx0;         // [T, H]
  const float* wx2f;        // [T] fragments of xvec_proj[2].weight [3H, H]
  const float* bx2;         // [T, 3H]
  const float* row_active;  // [N] or null
  const int* type_rowptr;   // [T+1]
  float* vp;                // [N, 3, 2H]  (v1 | v2), saved
  float* h2b;               // [N, H]      xvec_proj[0] output incl. bias, saved
  float* q23;               // [N, 2H]     (q | r) incl. bias, saved
  float* x_out;             // [N, H]
  float* vec_out;           // [N, 3, H]
  int N, T;
};

template <int H, int TR, int MINW>
__global__ __launch_bounds__(256, MINW) void node_update_fwd_kernel(UpdFwdArgs a) {
  using C = Cfg<H, TR>;
  constexpr int LD = C::LD, RB = C::RB, CPW = C::CPW, CB = C::CB;
  extern __shared__ __align__(16) float lds[];            // 2 x [TR][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % C::WC, wr = wave / C::WC;
  const TileInfo ti = find_tile(a.type_rowptr, a.T, a.N, TR, blockIdx.x);
  const int row0 = ti.row0, nrows = ti.nrows, t = ti.t;
  if (t >= a.T) {                                 // rows of unknown elements: zero
    constexpr int V = H / 4;
    for (int idx = tid; idx < nrows * V; idx += 256) {
      const int r = row0 + idx / V, c = (idx % V) * 4;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(a.x_out + (size_t)r * H + c) = z;
#pragma unroll
      for (int d = 0; d < 3; ++d) *reinterpret_cast<f32x4*>(a.vec_out + ((size_t)r * 3 + d) * H + c) = z;
    }
    return;
  }
  const f32x4* wv = reinterpret_cast<const f32x4*>(a.wvf + (size_t)t * 2 * H * H) + lane;
  const f32x4* wx0 = reinterpret_cast<const f32x4*>(a.wx0f + (size_t)t * 2 * H * H) + lane;
  const f32x4* wx2 = reinterpret_cast<const f32x4*>(a.wx2f + (size_t)t * 3 * H * H) + lane;
  const f32x4* bpv[2 * CPW];
  const f32x4* bpx[CPW];
  const f32x4* bpq[3 * CPW];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpv[p * CPW + j] = wv + (size_t)(p * CB + wc * CPW + j) * (H / 8) * 64;
#pragma unroll
  for (int j = 0; j < CPW; ++j) bpx[j] = wx0 + (size_t)(wc * CPW + j) * (2 * H / 8) * 64;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpq[p * CPW + j] = wx2 + (size_t)(p * CB + wc * CPW + j) * (H / 8) * 64;

  const int lrow = wr * RB * 32 + (lane & 31);
  // uniform bases of this tile's rows: every per-lane address below is base + a 32-bit offset inside the tile
  const rsrc_t x1_r = tile_rsrc(a.x1 + (size_t)row0 * H, nrows * H);
  const rsrc_t vec1_r = tile_rsrc(a.vec1 + (size_t)row0 * 3 * H, nrows * 3 * H);
  const rsrc_t act_r = tile_rsrc(a.row_active ? a.row_active + row0 : a.x1, a.row_active ? nrows : 0);
  const bool all_on = a.row_active == nullptr;
  const rsrc_t vp_r = tile_rsrc(a.vp + (size_t)row0 * 6 * H, nrows * 6 * H);
  const rsrc_t h2b_r = tile_rsrc(a.h2b + (size_t)row0 * H, nrows * H);
  const rsrc_t q23_r = tile_rsrc(a.q23 + (size_t)row0 * 2 * H, nrows * 2 * H);
  const rsrc_t xo_r = tile_rsrc(a.x_out + (size_t)row0 * H, nrows * H);
  const rsrc_t vo_r = tile_rsrc(a.vec_out + (size_t)row0 * 3 * H, nrows * 3 * H);
  float dot[RB][CPW][16], sq[RB][CPW][16];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) dot[rb][j][i] = sq[rb][j][i] = 0.f;

  // ---- vp[d] = vec1[d] Wv^T for the three Cartesian components; vec_dot and |v2|^2 accumulate in registers
  TileRegs<TR, H> regs;
  tile_load<TR, H>(regs, vec1_r, 3 * H, 0, tid);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float* buf = lds + (d & 1) * TR * LD;
    tile_store<TR, H, LD>(buf, regs, tid);
    BRing<2 * CPW, 3> rv;
    b_preload(rv, bpv);
    __syncthreads();
    if (d < 2) tile_load<TR, H>(regs, vec1_r, 3 * H, (d + 1) * H, tid);
    else tile_load<TR, H>(regs, x1_r, H, 0, tid);
    f32x16 accv[RB][2 * CPW];
    zero_acc(accv);
    mma_panel<H, LD, RB, 2 * CPW, 3, false>(accv, buf + lrow * LD + 4 * (lane >> 5), bpv, rv);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        const int col = (wc * CPW + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if ((i & 7) == 0) fence_sched();
          const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
          const float v1 = accv[rb][j][i], v2 = accv[rb][CPW + j][i];
          dot[rb][j][i] = fmaf(v1, v2, dot[rb][j][i]);
          sq[rb][j][i] = fmaf(v2, v2, sq[rb][j][i]);
          pin(dot[rb][j][i]);
          pin(sq[rb][j][i]);
          bst(vp_r, (lr * 3 + d) * 2 * H + col, v1);
          bst(vp_r, (lr * 3 + d) * 2 * H + H + col, v2);
        }
      }
  }
  // ---- xin = [x1 | sqrt(|v2|^2 + 1e-8)]: x1 -> buffer 1 (free since the product of d = 1), the norm -> buffer 0
  float* bufx = lds + TR * LD;
  float* bufn = lds;
  tile_store<TR, H, LD>(bufx, regs, tid);
  BRing<CPW, 4> rx;
  b_preload(rx, bpx);
  __syncthreads();                                   // every wave has finished the product of d = 2 (buffer 0)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int col = (wc * CPW + j) * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 16; ++i) bufn[((wr * RB + rb) * 32 + acc_row(i, lane)) * LD + col] = sqrtf(sq[rb][j][i] + 1e-8f);
    }
  __syncthreads();
  // ---- h2 = xin Wx0^T + bx0 (K = 2H: two panels)
  f32x16 acch[RB][CPW];
  zero_acc(acch);
  {
    const f32x4* bpx1[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpx1[j] = bpx[j] + (size_t)(H / 8) * 64;
    mma_panel<H, LD, RB, CPW, 4, true>(acch, bufx + lrow * LD + 4 * (lane >> 5), bpx, rx);
    mma_panel<H, LD, RB, CPW, 4, false>(acch, bufn + lrow * LD + 4 * (lane >> 5), bpx1, rx);
  }
  BRing<3 * CPW, 3> rq;
  b_preload(rq, bpq);
  __syncthreads();                                   // both buffers are free
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int col = (wc * CPW + j) * 32 + (lane & 31);
      const float bias = a.bx0[(size_t)t * H + col];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
        const float hv = acch[rb][j][i] + bias;
        bst(h2b_r, lr * H + col, hv);
        lds[lr * LD + col] = ssilu(hv);
      }
    }
  __syncthreads();
  // ---- (p | q | r) = a2 Wx2^T + bx2, then the update and the residual
  f32x16 accq[RB][3 * CPW];
  zero_acc(accq);
  mma_panel<H, LD, RB, 3 * CPW, 3, false>(accq, lds + lrow * LD + 4 * (lane >> 5), bpq, rq);
  const float inv_sqrt_h = rsqrtf((float)H);
  // Batches of G accumulator elements: all loads of a batch are requested, then consumed (one L2 round trip per batch
  // instead of one per element); the scheduling barriers keep hipcc from pulling later batches' loads in front.
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int col = (wc * CPW + j) * 32 + (lane & 31);
      const float bp_ = a.bx2[(size_t)t * 3 * H + col], bq_ = a.bx2[(size_t)t * 3 * H + H + col],
                  br_ = a.bx2[(size_t)t * 3 * H + 2 * H + col];
#pragma unroll
      for (int g = 0; g < 16 / G; ++g) {
        fence_sched();
        float v1[G][3], vv[G][3], onf[G];
#pragma unroll
        for (int e = 0; e < G; ++e) {
          const int lr = (wr * RB + rb) * 32 + acc_row(G * g + e, lane);
          onf[e] = bld(act_r, lr);
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            v1[e][d] = bld(vp_r, (lr * 3 + d) * 2 * H + col);
            vv[e][d] = bld(vec1_r, (lr * 3 + d) * H + col);
          }
        }
        fence_sched();
#pragma unroll
        for (int e = 0; e < G; ++e) {
          const int i = G * g + e;
          const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
          const bool on = all_on | (onf[e] != 0.f);
          const float p = accq[rb][j][i] + bp_, q = accq[rb][CPW + j][i] + bq_, r = accq[rb][2 * CPW + j][i] + br_;
          bst(q23_r, lr * 2 * H + col, q);
          bst(q23_r, lr * 2 * H + H + col, r);
          const float xv = bufx[lr * LD + col];          // x1: the tile is still in buffer 1
          bst(xo_r, lr * H + col, on ? xv + (p + q * dot[rb][j][i] * inv_sqrt_h) * kInvSqrt2 : 0.f);
#pragma unroll
          for (int d = 0; d < 3; ++d) bst(vo_r, (lr * 3 + d) * H + col, on ? fmaf(r, v1[e][d], vv[e][d]) : 0.f);
        }
      }
    }
}

// =====================================================================================================================
// node_update_bwd: (gx_out, gvec_out) -> (gx1, gvec1), parameters are constants
// =====================================================================================================================
struct UpdBwdArgs {
  const float* gxo;         // [N, H]
  const float* gvo;         // [N, 3, H]
  const float* vp;          // [N, 3, 2H]
  const float* h2b;         // [N, H]
  const float* q23;         // [N, 2H]
  const float* wx2tf;       // [T] fragments of xvec_proj[2].weight^T [H, 3H]
  const float* wx0tf;       // [T] fragments of xvec_proj[0].weight^T [2H, H]
  const float* wvtf;        // [T] fragments of vec_proj.weight^T [H, 2H]
  const float* row_active;  // [N] or null
  const int* type_rowptr;
  float* gx1;               // [N, H]
  float* gvec1;             // [N, 3, H]
  int N, T;
};

template <int H, int TR, int MINW>
__global__ __launch_bounds__(256, MINW) void node_update_bwd_kernel(UpdBwdArgs a) {
  using C = Cfg<H, TR>;
  constexpr int LD = C::LD, RB = C::RB, CPW = C::CPW, CB = C::CB, F4 = C::F4, V = H / 4;
  extern __shared__ __align__(16) float lds[];            // 2 x [TR][LD]
  float* buf0 = lds;
  float* buf1 = lds + TR * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % C::WC, wr = wave / C::WC;
  const TileInfo ti = find_tile(a.type_rowptr, a.T, a.N, TR, blockIdx.x);
  const int row0 = ti.row0, nrows = ti.nrows, t = ti.t;
  if (t >= a.T) {
    for (int idx = tid; idx < nrows * V; idx += 256) {
      const int r = row0 + idx / V, c = (idx % V) * 4;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(a.gx1 + (size_t)r * H + c) = z;
#pragma unroll
      for (int d = 0; d < 3; ++d) *reinterpret_cast<f32x4*>(a.gvec1 + ((size_t)r * 3 + d) * H + c) = z;
    }
    return;
  }
  const f32x4* wx2t = reinterpret_cast<const f32x4*>(a.wx2tf + (size_t)t * 3 * H * H) + lane;
  const f32x4* wx0t = reinterpret_cast<const f32x4*>(a.wx0tf + (size_t)t * 2 * H * H) + lane;
  const f32x4* wvt = reinterpret_cast<const f32x4*>(a.wvtf + (size_t)t * 2 * H * H) + lane;
  const f32x4* bpa[CPW];
  const f32x4* bpx[2 * CPW];
  const f32x4* bpg[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    bpa[j] = wx2t + (size_t)(wc * CPW + j) * (3 * H / 8) * 64;
    bpg[j] = wvt + (size_t)(wc * CPW + j) * (2 * H / 8) * 64;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpx[p * CPW + j] = wx0t + (size_t)(p * CB + wc * CPW + j) * (H / 8) * 64;
  BRing<CPW, 4> ra;
  b_preload(ra, bpa);
  const int lrow = wr * RB * 32 + (lane & 31);
  const float inv_sqrt_h = rsqrtf((float)H);
  const rsrc_t gxo_r = tile_rsrc(a.gxo + (size_t)row0 * H, nrows * H);
  const rsrc_t gvo_r = tile_rsrc(a.gvo + (size_t)row0 * 3 * H, nrows * 3 * H);
  const rsrc_t vp_r = tile_rsrc(a.vp + (size_t)row0 * 6 * H, nrows * 6 * H);
  const rsrc_t h2b_r = tile_rsrc(a.h2b + (size_t)row0 * H, nrows * H);
  const rsrc_t q23_r = tile_rsrc(a.q23 + (size_t)row0 * 2 * H, nrows * 2 * H);
  const rsrc_t act_r = tile_rsrc(a.row_active ? a.row_active + row0 : a.gxo, a.row_active ? nrows : 0);
  const bool all_on = a.row_active == nullptr;
  const rsrc_t gx1_r = tile_rsrc(a.gx1 + (size_t)row0 * H, nrows * H);
  const rsrc_t gvec1_r = tile_rsrc(a.gvec1 + (size_t)row0 * 3 * H, nrows * 3 * H);

  // ---- gq = (gx/sqrt2 | gx vdot/sqrt2 | sum_d gv[d] v1[d]) elementwise, a float4 per thread and position
  TileRegs<TR, H> g3;
#pragma unroll
  for (int it = 0; it < F4; ++it) {
    if ((it & 1) == 0) fence_sched();     // two positions (20 float4 loads) in flight, not all F4
    const int idx = tid + it * 256, lr = idx / V, c = (idx % V) * 4;
    f32x4 g1 = {0.f, 0.f, 0.f, 0.f}, g2 = g1, gq3 = g1;
    {
      // (rows past the tile's end load zeros; inactive rows are multiplied by zero: their saved values are finite)
      const float on = (all_on | (bld(act_r, lr) != 0.f)) ? 1.f : 0.f;
      const f32x4 gx = bld4(gxo_r, lr * H + c) * on;
      f32x4 vd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const f32x4 v1 = bld4(vp_r, (lr * 3 + d) * 2 * H + c);
        const f32x4 v2 = bld4(vp_r, (lr * 3 + d) * 2 * H + H + c);
        const f32x4 gv = bld4(gvo_r, (lr * 3 + d) * H + c) * on;
        vd += v1 * v2;
        gq3 += gv * v1;
      }
      g1 = gx * kInvSqrt2;
      g2 = gx * vd * (inv_sqrt_h * kInvSqrt2);
    }
    *reinterpret_cast<f32x4*>(buf0 + lr * LD + c) = g1;
    *reinterpret_cast<f32x4*>(buf1 + lr * LD + c) = g2;
    g3.v[it] = gq3;
  }
  __syncthreads();
  // ---- ga2 = gq Wx2  (K = 3H: three panels)
  f32x16 acc[RB][CPW];
  zero_acc(acc);
  {
    const f32x4* bp1[CPW];
    const f32x4* bp2[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) { bp1[j] = bpa[j] + (size_t)(H / 8) * 64; bp2[j] = bpa[j] + (size_t)(2 * H / 8) * 64; }
    mma_panel<H, LD, RB, CPW, 4, true>(acc, buf0 + lrow * LD + 4 * (lane >> 5), bpa, ra);
    __syncthreads();                                 // buffer 0 is free
    tile_store<TR, H, LD>(buf0, g3, tid);
    mma_panel<H, LD, RB, CPW, 4, true>(acc, buf1 + lrow * LD + 4 * (lane >> 5), bp1, ra);
    __syncthreads();                                 // third part in place, buffer 1 free
    mma_panel<H, LD, RB, CPW, 4, false>(acc, buf0 + lrow * LD + 4 * (lane >> 5), bp2, ra);
  }
  BRing<2 * CPW, 3> rx;
  b_preload(rx, bpx);
  fence_sched();
  // ---- gh2 = ga2 * ScaledSiLU'(h2b) -> buffer 1
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int col = (wc * CPW + j) * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
        const float hv = bld(h2b_r, lr * H + col);
        buf1[lr * LD + col] = acc[rb][j][i] * dssilu(hv);
      }
    }
  __syncthreads();
  // ---- gxin = gh2 Wx0   (gx1 part | g|v2| part); the gx1 part accumulates onto the identity term gx
  f32x16 accx[RB][2 * CPW];
  float gxv[RB][CPW][16];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int col = (wc * CPW + j) * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
        const float on = (all_on | (bld(act_r, lr) != 0.f)) ? 1.f : 0.f;
        gxv[rb][j][i] = bld(gxo_r, lr * H + col) * on;
        accx[rb][j][i] = gxv[rb][j][i];
        accx[rb][CPW + j][i] = 0.f;
      }
    }
  mma_panel<H, LD, RB, 2 * CPW, 3, false>(accx, buf1 + lrow * LD + 4 * (lane >> 5), bpx, rx);
  fence_sched();
  // per accumulator position: s = gvdot / sqrt(H) = gx q / sqrt(2H), gnn = g|v2| / |v2|   (batches of G, loads first)
  float s_[RB][CPW][16], gnn[RB][CPW][16];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int col = (wc * CPW + j) * 32 + (lane & 31);
#pragma unroll
      for (int g = 0; g < 16 / G; ++g) {
        fence_sched();
        float v2[G][3], qv[G];
#pragma unroll
        for (int e = 0; e < G; ++e) {
          const int lr = (wr * RB + rb) * 32 + acc_row(G * g + e, lane);
          qv[e] = bld(q23_r, lr * 2 * H + col);
#pragma unroll
          for (int d = 0; d < 3; ++d) v2[e][d] = bld(vp_r, (lr * 3 + d) * 2 * H + H + col);
        }
        fence_sched();
#pragma unroll
        for (int e = 0; e < G; ++e) {
          const int i = G * g + e;
          const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
          bst(gx1_r, lr * H + col, accx[rb][j][i]);
          const float sq = fmaf(v2[e][0], v2[e][0], fmaf(v2[e][1], v2[e][1], v2[e][2] * v2[e][2]));
          s_[rb][j][i] = gxv[rb][j][i] * qv[e] * (kInvSqrt2 * inv_sqrt_h);
          gnn[rb][j][i] = accx[rb][CPW + j][i] * rsqrtf(sq + 1e-8f);
          pin(s_[rb][j][i]);
          pin(gnn[rb][j][i]);
        }
      }
    }
  // ---- gvec1[d] = gv[d] + (gv1[d] | gv2[d]) Wv,  gv1 = gv q3 + s v2,  gv2 = s v1 + gnn v2  (the accumulator starts at gv[d])
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    BRing<CPW, 4> rg;
    b_preload(rg, bpg);
    f32x16 accg[RB][CPW];
    __syncthreads();                                 // the previous product has read both buffers
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        const int col = (wc * CPW + j) * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 16 / G; ++g) {
          fence_sched();
          float gv[G], q3[G], v1[G], v2[G], onf[G];
#pragma unroll
          for (int e = 0; e < G; ++e) {
            const int lr = (wr * RB + rb) * 32 + acc_row(G * g + e, lane);
            onf[e] = bld(act_r, lr);
            gv[e] = bld(gvo_r, (lr * 3 + d) * H + col);
            q3[e] = bld(q23_r, lr * 2 * H + H + col);
            v1[e] = bld(vp_r, (lr * 3 + d) * 2 * H + col);
            v2[e] = bld(vp_r, (lr * 3 + d) * 2 * H + H + col);
          }
          fence_sched();
#pragma unroll
          for (int e = 0; e < G; ++e) {
            const int i = G * g + e;
            const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
            const float gvm = (all_on | (onf[e] != 0.f)) ? gv[e] : 0.f;
            accg[rb][j][i] = gvm;
            buf0[lr * LD + col] = fmaf(gvm, q3[e], s_[rb][j][i] * v2[e]);
            buf1[lr * LD + col] = fmaf(s_[rb][j][i], v1[e], gnn[rb][j][i] * v2[e]);
          }
        }
      }
    __syncthreads();
    const f32x4* bpg1[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpg1[j] = bpg[j] + (size_t)(H / 8) * 64;
    mma_panel<H, LD, RB, CPW, 4, true>(accg, buf0 + lrow * LD + 4 * (lane >> 5), bpg, rg);
    mma_panel<H, LD, RB, CPW, 4, false>(accg, buf1 + lrow * LD + 4 * (lane >> 5), bpg1, rg);
    fence_sched();
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        const int col = (wc * CPW + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int lr = (wr * RB + rb) * 32 + acc_row(i, lane);
          bst(gvec1_r, (lr * 3 + d) * H + col, accg[rb][j][i]);
        }
      }
  }
}

int tiles_of(const int* rp_host, int T, int N, int TR) {
  int n = 0;
  for (int t = 0; t < T; ++t) n += (rp_host[t + 1] - rp_host[t] + TR - 1) / TR;
  return n + (N - rp_host[T] + TR - 1) / TR;
}

// Launch with `lds_bytes` of dynamic LDS (> 64 KB needs the opt-in, once per kernel).
template <typename Args>
int launch_chain(void (*kernel)(Args), dim3 grid, size_t lds_bytes, void* stream, const Args& args) {
  static void (*done[8])(Args);
  static int ndone = 0;
  bool seen = false;
  for (int i = 0; i < ndone; ++i) seen |= done[i] == kernel;
  if (!seen) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return HN_ERR_LDS;
    if (ndone < 8) done[ndone++] = kernel;
  }
  hipLaunchKernelGGL(kernel, grid, dim3(256), lds_bytes, (hipStream_t)stream, args);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

// LDS floats of a kernel family for tile (H, TR): NB buffers of [TR][H + 4]
#define HN_CHAIN_DISPATCH(KERNEL, GRID, NB, ARGS)                                                          \
  switch (hidden) {                                                                                        \
    case 64: return launch_chain(KERNEL<64, 64>, GRID(64), (size_t)NB * 64 * 68 * 4, stream, ARGS);        \
    case 128: return launch_chain(KERNEL<128, 64>, GRID(64), (size_t)NB * 64 * 132 * 4, stream, ARGS);     \
    case 256: return launch_chain(KERNEL<256, 32>, GRID(32), (size_t)NB * 32 * 260 * 4, stream, ARGS);     \
    default: return HN_ERR_BAD_ARG;                                                                        \
  }

// Update kernels: tile (H, TR) and register budget.  H = 128 has two instances: 64-row tiles keep ~300 values per lane
// live (one wave per SIMD, 512 registers) and load every weight fragment once per 64 rows -- the choice while the grid
// has at most one tile per CU anyway; 32-row tiles fit 256 registers, so two or three workgroups share a CU and one's
// elementwise phases run beside another's matrix phases -- the choice for large grids.
#define HN_TILES(TR) tiles_of(type_rowptr_host, num_rel, num_nodes, TR)
#define HN_UPDATE_DISPATCH(KERNEL, ARGS)                                                                              \
  switch (hidden) {                                                                                                   \
    case 64: return launch_chain(KERNEL<64, 64, 2>, dim3((unsigned)HN_TILES(64)), (size_t)2 * 64 * 68 * 4, stream, ARGS); \
    case 128:                                                                                                         \
      if (HN_TILES(64) <= 320)                                                                                        \
        return launch_chain(KERNEL<128, 64, 1>, dim3((unsigned)HN_TILES(64)), (size_t)2 * 64 * 132 * 4, stream, ARGS); \
      return launch_chain(KERNEL<128, 32, 2>, dim3((unsigned)HN_TILES(32)), (size_t)2 * 32 * 132 * 4, stream, ARGS);  \
    case 256: return launch_chain(KERNEL<256, 32, 1>, dim3((unsigned)HN_TILES(32)), (size_t)2 * 32 * 260 * 4, stream, ARGS); \
    default: return HN_ERR_BAD_ARG;                                                                                   \
  }

}  // namespace

extern "C" int hermnet_node_chain_supported(int hidden) { return hidden == 64 || hidden == 128 || hidden == 256; }

extern "C" int hermnet_node_pre_fwd(const float* x, const float* w1_frag, const float* b1, const float* w2_frag,
                                    const float* b2, float* hb, float* xh, float* mean, float* rstd, int num_src,
                                    int num_rel, int hidden, int hidden_real, float eps, void* stream) {
  if (num_src < 0 || num_rel <= 0 || hidden_real > hidden) return HN_ERR_BAD_ARG;
  if (!hermnet_node_chain_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_src == 0) return HN_OK;
  if (!x || !w1_frag || !b1 || !w2_frag || !b2 || !hb || !xh || !mean || !rstd) return HN_ERR_BAD_ARG;
  PreFwdArgs a = {x, w1_frag, b1, w2_frag, b2, hb, xh, mean, rstd, num_src, num_rel, hidden_real > 0 ? hidden_real : hidden, eps};
#define HN_GRID(TR) dim3((unsigned)((num_src + TR - 1) / TR), (unsigned)num_rel)
  HN_CHAIN_DISPATCH(node_pre_fwd_kernel, HN_GRID, 1, a);
}

extern "C" int hermnet_node_pre_bwd(const float* gxh, const float* hb, const float* w2t_frag, const float* w1t_frag,
                                    float* gn_parts, const float* x, const float* mean, const float* rstd,
                                    const float* add, float* gx, int num_src, int num_rel, int hidden, int hidden_real,
                                    void* stream) {
  if (num_src < 0 || num_rel <= 0 || hidden_real > hidden) return HN_ERR_BAD_ARG;
  if (!hermnet_node_chain_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_src == 0) return HN_OK;
  if (!gxh || !hb || !w2t_frag || !w1t_frag || !gn_parts || !x || !mean || !rstd || !gx) return HN_ERR_BAD_ARG;
  PreBwdArgs a = {gxh, hb, w2t_frag, w1t_frag, gn_parts, num_src, num_rel};
  // chunk buffers: 2 x [TR][min(H,128) + 4]
  int rc;
  switch (hidden) {
    case 64: rc = launch_chain(node_pre_bwd_kernel<64, 64>, HN_GRID(64), (size_t)2 * 64 * 68 * 4, stream, a); break;
    case 128: rc = launch_chain(node_pre_bwd_kernel<128, 64>, HN_GRID(64), (size_t)2 * 64 * 132 * 4, stream, a); break;
    default: rc = launch_chain(node_pre_bwd_kernel<256, 32>, HN_GRID(32), (size_t)2 * 32 * 132 * 4, stream, a); break;
  }
#undef HN_GRID
  if (rc != HN_OK) return rc;
  hipLaunchKernelGGL(layernorm_bwd_parts_kernel, dim3((unsigned)((num_src + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     gn_parts, num_rel, (long)num_src * hidden, x, mean, rstd, add, gx, num_src, hidden,
                     hidden_real > 0 ? hidden_real : hidden);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_node_update_fwd(const float* x1, const float* vec1, const float* wv_frag, const float* wx0_frag,
                                       const float* bx0, const float* wx2_frag, const float* bx2,
                                       const float* row_active, const int* type_rowptr, const int* type_rowptr_host,
                                       float* vp, float* h2b, float* q23, float* x_out, float* vec_out, int num_nodes,
                                       int num_rel, int hidden, void* stream) {
  if (num_nodes < 0 || num_rel <= 0 || !type_rowptr_host) return HN_ERR_BAD_ARG;
  if (!hermnet_node_chain_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_nodes == 0) return HN_OK;
  if (!x1 || !vec1 || !wv_frag || !wx0_frag || !bx0 || !wx2_frag || !bx2 || !type_rowptr || !vp || !h2b || !q23 ||
      !x_out || !vec_out || type_rowptr_host[num_rel] > num_nodes)
    return HN_ERR_BAD_ARG;
  UpdFwdArgs a = {x1, vec1, wv_frag, wx0_frag, bx0, wx2_frag, bx2, row_active, type_rowptr, vp, h2b, q23, x_out, vec_out,
                  num_nodes, num_rel};
  HN_UPDATE_DISPATCH(node_update_fwd_kernel, a);
}

extern "C" int hermnet_node_update_bwd(const float* gx_out, const float* gvec_out, const float* vp, const float* h2b,
                                       const float* q23, const float* wx2t_frag, const float* wx0t_frag,
                                       const float* wvt_frag, const float* row_active, const int* type_rowptr,
                                       const int* type_rowptr_host, float* gx1, float* gvec1, int num_nodes, int num_rel,
                                       int hidden, void* stream) {
  if (num_nodes < 0 || num_rel <= 0 || !type_rowptr_host) return HN_ERR_BAD_ARG;
  if (!hermnet_node_chain_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_nodes == 0) return HN_OK;
  if (!gx_out || !gvec_out || !vp || !h2b || !q23 || !wx2t_frag || !wx0t_frag || !wvt_frag || !type_rowptr || !gx1 ||
      !gvec1 || type_rowptr_host[num_rel] > num_nodes)
    return HN_ERR_BAD_ARG;
  UpdBwdArgs a = {gx_out, gvec_out, vp, h2b, q23, wx2t_frag, wx0t_frag, wvt_frag, row_active, type_rowptr, gx1, gvec1,
                  num_nodes, num_rel};
  HN_UPDATE_DISPATCH(node_update_bwd_kernel, a);
}
