// gfx950 (MI355X / CDNA4) kernels for HermNet's relational message + segmented scatter-add.
//
// Replaces, per HeteroVertexConv layer and for all relations at once (files under /root/reference):
//   RadialBasis.forward               rmnet.py:168-172   (Gaussian taps * envelope, in registers)
//   rbf_proj Linear(R, 3H) on edges   rmnet.py:55        (banded contraction against an LDS weight tile)
//   MessagePassing.propagate gather   rmnet.py:58        (row gathers of xh[src], vec[src])
//   PaiNNMessage.message              rmnet.py:61-67     (in registers)
//   PaiNNMessage.aggregate scatter    rmnet.py:69-73     (segmented sum over the CSR row, no atomics)
//   residual                          rmnet.py:24-26     (epilogue, single write of x1 / vec1)
//   in_subgraph                       utils.py:11-24     (replaced by the relation-ordered CSR/CSC)
//
// Mapping (wave = 64 lanes).  A wave owns one target (fwd) or source (bwd) row at a time.  A lane
// owns VW adjacent channels of a 64-channel column block, so 64/VW lanes cover the block and the
// wave processes VW consecutive edges of the row's segment at once (VW = 2: two 32-lane halves,
// VW = 4: four 16-lane quarters).  A lane group moves one 256-byte row segment per load and reads
// the LDS weight tile with conflict-free ds_read_b64 / ds_read_b128.  Groups are combined once per
// row; nothing is accumulated in HBM, there are no atomics, results are bit-reproducible.
//
// The kernels are VALU-issue bound (rocprofv3: VALU busy 56-72 %, LDS < 15 %), so the variants
// trade registers (waves per SIMD) against per-edge instruction overhead; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/hermnet_hip.h"
#include "hermnet_math.h"
#include "message_bwd_cl.h"
int hn_option(int option);      // host_api.cpp

#define HN_LDS_ROW (3 * HN_CB)      // floats per tap row in LDS: [part][64]
// Phase fence for the instruction scheduler: without it hipcc hoists the LDS reads of all three
// parts (and the next edges' loads) to the top of the iteration and spills.
#ifndef HN_SB
#define HN_SB __builtin_amdgcn_sched_barrier(0)
#endif

#if defined(HN_STAMPS)
// Diagnostic build only (tools/build_variant.sh stamps -DHN_STAMPS): per-phase cycle sums of the backward kernel,
// read back with hermnet_debug_stamps().  The stamps drain outstanding memory operations, so such a build says
// where the cycles go, not how long the real kernel runs.
__device__ unsigned long long hn_dbg[8];
#define HN_T(var) unsigned long long var; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define HN_TNW(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#endif

namespace {

struct MsgArgs {
  // graph
  int N, E, T;
  int Nsrc;              // rows of the source space (xh, vec gathers); = N unless the targets are virtual rows (HTNet)
  const int* res_row;    // [N] source row entering the residual of target row r, or null (= r)
  const int* type_rowptr;
  const int* csr_rowptr;
  const int* csr_src;
  const int* csc_rowptr;
  const int* csc_tgt;
  const int* csc_pos;
  // radial basis
  const float* offset;
  int R;
  float inv_rc, coeff;
  int env_kind, env_p;
  // tensors
  int H;
  const float* xh;     // [T,N,3H]
  const float* vec;    // [N,3,H] or null
  const float* x;      // [N,H]
  const float* wt;     // [T,R,3H]
  const float* brbf;   // [T,3H]
  const float4* edge;  // [E] (rx,ry,rz,d)
  float* x1;           // fwd out
  float* vec1;         // fwd out
  const float* xh_bias;  // [T,3H] or null: bias of the x_proj output, added on load (xh holds the GEMM without it)
  const float* gx1;    // bwd in
  const float* gvec1;  // bwd in
  float* gxh;          // bwd out [T,N,3H]
  float* gvec;         // bwd out [N,3,H] or null
  float* gx;           // bwd out [N,H]
  float4* gedge;       // bwd out [H/64,E]
  int rows_per_block;
  int xcd_remap;       // 1 = XCD-contiguous block order (see xcd_contiguous)
  int split_t;         // bwd: 1 = one relation per workgroup (blockIdx.z), gvec is [T,N,3,H] partial sums
  const int* row_ranges;   // fwd: [T][2] target rows [lo, hi) of every relation this launch covers, or null = whole blocks
  int zero_unknown;        // fwd: 1 = this launch also zeroes the rows of unknown-element atoms
  // fwd, tap-row windows (num_rbf too large for one LDS tile; see hermnet_message_scatter_fwd): the launch stages the
  // padded tile rows [win_base, win_base + win_rows) and owns the edges whose first tap row lies in [win_lo, win_hi);
  // win_accumulate = 1: its sums are added to the rows the launch over the other window wrote (residual included there)
  int win_base, win_rows, win_lo, win_hi, win_accumulate;
};

// ---- VW-wide per-lane vectors ------------------------------------------------------------------
template <int VW> struct VecT;
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<4> { typedef float4 type; };

template <int VW>
struct Vec {
  float v[VW];
  __device__ __forceinline__ static Vec zero() {
    Vec r;
#pragma unroll
    for (int i = 0; i < VW; ++i) r.v[i] = 0.f;
    return r;
  }
  __device__ __forceinline__ static Vec load(const float* p) {
    typename VecT<VW>::type t = *reinterpret_cast<const typename VecT<VW>::type*>(p);
    Vec r;
    const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (int i = 0; i < VW; ++i) r.v[i] = f[i];
    return r;
  }
  __device__ __forceinline__ void store(float* p) const {
    typename VecT<VW>::type t;
    float* f = reinterpret_cast<float*>(&t);
#pragma unroll
    for (int i = 0; i < VW; ++i) f[i] = v[i];
    *reinterpret_cast<typename VecT<VW>::type*>(p) = t;
  }
};

// Elementwise ops are written on explicit <2 x float> pairs so that they select the packed fp32
// instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, scalars broadcast by op_sel): the SLP
// vectoriser packs only some of these chains on its own.
typedef float hn_f2 __attribute__((ext_vector_type(2)));
#define HN_PAIRWISE(expr)                                   \
  Vec<VW> r;                                                \
  _Pragma("unroll") for (int i = 0; i < VW; i += 2) {       \
    const hn_f2 rv = (expr);                                \
    r.v[i] = rv.x;                                          \
    r.v[i + 1] = rv.y;                                      \
  }                                                         \
  return r;
#define HN_P(x) (hn_f2{(x).v[i], (x).v[i + 1]})
#define HN_S(x) (hn_f2{(x), (x)})

template <int VW>
__device__ __forceinline__ Vec<VW> v_fma(const Vec<VW>& a, const Vec<VW>& b, const Vec<VW>& c) {   // a * b + c
  HN_PAIRWISE(__builtin_elementwise_fma(HN_P(a), HN_P(b), HN_P(c)))
}
template <int VW>
__device__ __forceinline__ Vec<VW> v_sfma(float s, const Vec<VW>& a, const Vec<VW>& c) {   // s * a + c
  HN_PAIRWISE(__builtin_elementwise_fma(HN_S(s), HN_P(a), HN_P(c)))
}
template <int VW>
__device__ __forceinline__ Vec<VW> v_mul(const Vec<VW>& a, const Vec<VW>& b) {
  HN_PAIRWISE(HN_P(a) * HN_P(b))
}
template <int VW>
__device__ __forceinline__ Vec<VW> v_scale(const Vec<VW>& a, float s) {
  HN_PAIRWISE(HN_P(a) * HN_S(s))
}
template <int VW>
__device__ __forceinline__ Vec<VW> v_add(const Vec<VW>& a, const Vec<VW>& b) {
  HN_PAIRWISE(HN_P(a) + HN_P(b))
}
template <int VW>
__device__ __forceinline__ float v_hsum(const Vec<VW>& a) {
  float s = a.v[0];
#pragma unroll
  for (int i = 1; i < VW; ++i) s += a.v[i];
  return s;
}

// ---- cross-lane sums ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// Sum over the VW lane groups of the wave (same channel, different edge slot); all lanes get the total.
template <int VW>
__device__ __forceinline__ float groups_sum1(float v) {
  if (VW == 4) v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
template <int VW>
__device__ __forceinline__ Vec<VW> groups_sum(Vec<VW> a) {
#pragma unroll
  for (int i = 0; i < VW; ++i) a.v[i] = groups_sum1<VW>(a.v[i]);
  return a;
}

// VW = 4 (four 16-lane rows): row g of the result holds the sum over the four rows of a.v[g] -- three
// v_permlane swaps (gfx950) and three adds on the VALU instead of eight LDS-routed shuffles.  The caller then
// owns channel `first channel + g` of the reduced vector: a dword store per lane, 256 contiguous bytes per wave.
__device__ __forceinline__ float rows_reduce4(const Vec<4>& a) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  // permlane16_swap: odd rows of the first operand <-> even rows of the second
  const u2 s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.v[0]), __float_as_uint(a.v[1]), false, false);
  const u2 s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.v[2]), __float_as_uint(a.v[3]), false, false);
  const float u = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);   // rows: v0(r0+r1), v1(r0+r1), v0(r2+r3), v1(r2+r3)
  const float w = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);   //       v2(r0+r1), v3(r0+r1), v2(r2+r3), v3(r2+r3)
  // permlane32_swap: upper half of the first operand <-> lower half of the second
  const u2 h = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(w), false, false);
  return __uint_as_float(h[0]) + __uint_as_float(h[1]);               // rows: sum v0, sum v1, sum v2, sum v3
}

// Sum over the 64/VW lanes of each lane group (all channels of one edge); every lane gets its group's total.
template <int VW>
__device__ __forceinline__ float group_allsum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
  v += dpp_mov<0x141>(v);   // row_half_mirror      : other quad of the 8-lane group
  v += dpp_mov<0x140>(v);   // row_mirror           : other half of the 16-lane DPP row
  if (VW == 2) v += __shfl_xor(v, 16, 64);
  return v;
}

// ---- LDS staging ---------------------------------------------------------------------------------
// Zero-padded weight tile of (relation t, column block cb) and the tap centres:
//   wl[(k + HN_PAD) * 192 + part * 64 + c] = wt[t][k][part * H + cb * 64 + c]   (0 outside 0 <= k < R)
//   mu[k + HN_PAD] = offset[clamp(k)]
// (`row0`, `nrows`: the window of padded rows held by wl -- all of them unless the launch works on a tap-row window; mu
// always covers every row)
template <int NTHREADS>
__device__ __forceinline__ void stage_weights(const MsgArgs& a, int t, int cb, float* wl, float* mu, int row0, int nrows) {
  const int rows = a.R + 2 * HN_PAD + 1;
  const int n4 = nrows * (HN_LDS_ROW / 4);
  constexpr int INFLIGHT = 8;   // independent 16-byte loads per thread before the first LDS write
  for (int base = threadIdx.x; base < n4; base += NTHREADS * INFLIGHT) {
    float4 v[INFLIGHT];
#pragma unroll
    for (int q8 = 0; q8 < INFLIGHT; ++q8) {
      const int idx = base + q8 * NTHREADS;
      const int kk = idx / (HN_LDS_ROW / 4);
      const int q = idx - kk * (HN_LDS_ROW / 4);
      const int k = kk + row0 - HN_PAD;
      v[q8] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < n4 && k >= 0 && k < a.R) {
        // 16 float4 per 64-channel part: part = q >> 4, c4 = q & 15
        v[q8] = *reinterpret_cast<const float4*>(
            a.wt + ((size_t)(t * a.R + k) * 3 * a.H + (q >> 4) * a.H + cb * HN_CB + (q & 15) * 4));
      }
    }
#pragma unroll
    for (int q8 = 0; q8 < INFLIGHT; ++q8) {
      const int idx = base + q8 * NTHREADS;
      if (idx < n4) *reinterpret_cast<float4*>(wl + (size_t)idx * 4) = v[q8];   // LDS image is idx-linear
    }
  }
  for (int kk = threadIdx.x; kk < rows; kk += NTHREADS) {
    int k = kk - HN_PAD;
    k = k < 0 ? 0 : (k >= a.R ? a.R - 1 : k);
    mu[kk] = a.offset[k];
  }
}

// ---- radial basis ----------------------------------------------------------------------------------
// Tap values of one edge, evaluated cooperatively: lane m < 12 of the edge's lane group computes
// tap m (same fp32 operation order as the reference, rmnet.py:156-172:  g = exp(coeff (u - mu_m)^2))
// and parks {g, g (u - mu_m)} in the wave's LDS scratch; every lane then reads the 12 pairs back
// as broadcasts.  One exp per edge and lane instead of twelve.
__device__ __forceinline__ void coop_taps(const float* mu, float2* tb, int lo, float u, float coeff, int gl) {
  if (gl < HN_TAPS) {
    const float diff = u - mu[lo + HN_PAD + gl];
    const float g = __expf(coeff * (diff * diff));
    tb[gl] = make_float2(g, g * diff);
  }
}

template <bool WITH_DER>
__device__ __forceinline__ void rbf_taps(const float2* tb, float (&g)[HN_TAPS], float (&gd)[HN_TAPS]) {
#pragma unroll
  for (int m = 0; m < HN_TAPS; ++m) {
    const float2 t = tb[m];
    g[m] = t.x;
    if (WITH_DER) gd[m] = t.y;
  }
}

// Banded contraction of one part (s, a or b) for this lane's VW channels:
//   S0 = sum_m g_m W[lo+m][part],  (WITH_DER) S1 = sum_m gd_m W[lo+m][part].
// One conflict-free LDS read per tap, double buffered behind the FMAs of the previous tap.
template <bool WITH_DER, int VW>
__device__ __forceinline__ void rbf_part(const float* wcol, const float (&g)[HN_TAPS], const float (&gd)[HN_TAPS],
                                         Vec<VW>& S0, Vec<VW>& S1) {
  S0 = Vec<VW>::zero();
  if (WITH_DER) S1 = Vec<VW>::zero();
#if defined(HN_SIMPLE_PART)
#pragma unroll
  for (int m = 0; m < HN_TAPS; ++m) {
    const Vec<VW> w = Vec<VW>::load(wcol + m * HN_LDS_ROW);
    S0 = v_sfma(g[m], w, S0);
    if (WITH_DER) S1 = v_sfma(gd[m], w, S1);
  }
#elif defined(HN_BURST_PART)
  // all HN_BURST_PART reads of a group in flight before the first FMA: the LDS latency is paid once per group
  // instead of once per tap (the double-buffered form below waits on every read one tap after issuing it)
#pragma unroll
  for (int m0 = 0; m0 < HN_TAPS; m0 += HN_BURST_PART) {
    Vec<VW> w[HN_BURST_PART];
#pragma unroll
    for (int q = 0; q < HN_BURST_PART; ++q) w[q] = Vec<VW>::load(wcol + (m0 + q) * HN_LDS_ROW);
#pragma unroll
    for (int q = 0; q < HN_BURST_PART; ++q) {
      S0 = v_sfma(g[m0 + q], w[q], S0);
      if (WITH_DER) S1 = v_sfma(gd[m0 + q], w[q], S1);
    }
  }
#else
  Vec<VW> wa = Vec<VW>::load(wcol), wb;
#pragma unroll
  for (int m = 0; m < HN_TAPS; m += 2) {
    wb = Vec<VW>::load(wcol + (m + 1) * HN_LDS_ROW);
    S0 = v_sfma(g[m], wa, S0);
    if (WITH_DER) S1 = v_sfma(gd[m], wa, S1);
    if (m + 2 < HN_TAPS) wa = Vec<VW>::load(wcol + (m + 2) * HN_LDS_ROW);
    S0 = v_sfma(g[m + 1], wb, S0);
    if (WITH_DER) S1 = v_sfma(gd[m + 1], wb, S1);
  }
#endif
}

// All three parts in one pass over the taps (no g[] arrays): the register-lean form used by the
// 16-wave (<= 128 VGPR) variants.
template <bool WITH_DER, int VW>
__device__ __forceinline__ void rbf_all(const float* wcol, const float2* tb, Vec<VW> (&S0)[3], Vec<VW> (&S1)[3]) {
#pragma unroll
  for (int p = 0; p < 3; ++p) { S0[p] = Vec<VW>::zero(); if (WITH_DER) S1[p] = Vec<VW>::zero(); }
#pragma unroll
  for (int m = 0; m < HN_TAPS; ++m) {
    const float2 t = tb[m];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const Vec<VW> w = Vec<VW>::load(wcol + m * HN_LDS_ROW + p * HN_CB);
      S0[p] = v_sfma(t.x, w, S0[p]);
      if (WITH_DER) S1[p] = v_sfma(t.y, w, S1[p]);
    }
  }
}

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an XCD and its
// 4 MiB L2), so with the identity mapping every XCD walks the WHOLE row range and sees every gather table in full
// (15-46 MB at config 2: L2 misses, Infinity-Cache hits).  Rows are in (relation, atom id) order and atom ids of a
// structure are spatially coherent, so giving XCD k the k-th contiguous eighth of the blocks confines its gathers to
// the rows near one slab of the structure -- a few MB that its L2 can hold.  Speed only; any mapping is correct.
__device__ __forceinline__ int xcd_contiguous(int b, int nb, int on) {
  const int per = nb >> 3;
  if (!on || b >= per * 8) return b;        // (the nb % 8 trailing blocks keep their place)
  return (b & 7) * per + (b >> 3);
}

// Decode blockIdx.x -> (relation t, first row, one-past-last row) for blocks of `rpb` rows
// laid out relation after relation.  Returns -1 for a work block, else the index of the
// surplus block (0, 1, ...) past the last work block.
__device__ __forceinline__ int decode_block(const MsgArgs& a, int bx, int& t, int& r0, int& r1) {
  for (t = 0; t < a.T; ++t) {
    // (atom shards: the targets that read no halo row run while the halo exchange is in flight, the others after
    // it -- two launches over complementary [lo, hi) ranges of every relation's row block)
    const int lo = a.row_ranges ? a.row_ranges[2 * t] : a.type_rowptr[t];
    const int hi = a.row_ranges ? a.row_ranges[2 * t + 1] : a.type_rowptr[t + 1];
    const int nb = hi > lo ? (hi - lo + a.rows_per_block - 1) / a.rows_per_block : 0;
    if (bx < nb) {
      r0 = lo + bx * a.rows_per_block;
      r1 = min(r0 + a.rows_per_block, hi);
      return -1;
    }
    bx -= nb;
  }
  return bx;
}

// ------------------------------------------------------------------------------------------
// Forward: one workgroup = (relation, column block, chunk of target rows).
// ------------------------------------------------------------------------------------------
template <bool HAS_VEC, int VW>
struct FwdIn {
  float4 g;        // (rx, ry, rz, d)
  Vec<VW> xs, xa, xb;
  Vec<VW> vj[3];
  bool live;
};

// WIN: the launch works on a window of the tap rows (MsgArgs::win_*): an edge it does not own is a padding slot (lv = 0)
// reading a clamped tile row; with win_accumulate the epilogue adds to the other launch's rows.
template <bool HAS_VEC, int NW, int VW, int PF, bool FUSED, bool WIN = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void message_scatter_fwd_kernel(MsgArgs a) {
  static_assert(!WIN || VW == 4, "the windowed form exists for the default (VW = 4) variants");
  extern __shared__ __align__(16) float lds[];
  const int tile_rows = WIN ? a.win_rows : a.R + 2 * HN_PAD + 1;
  float* wl = lds;
  float* mu = lds + tile_rows * HN_LDS_ROW;
  float2* tapbase = reinterpret_cast<float2*>(mu + ((a.R + 2 * HN_PAD + 1 + 3) & ~3));
  constexpr int LPE = 64 / VW;   // lanes per edge

  const int cb = blockIdx.y;
  int t, r0, r1;
  const int surplus = decode_block(a, xcd_contiguous(blockIdx.x, gridDim.x, a.xcd_remap), t, r0, r1);
  if (surplus >= 0) {
    // the first surplus block zeroes the rows of unknown-type atoms (they are never targets)
    if (surplus == 0 && a.zero_unknown) {
      const int c = cb * HN_CB + (threadIdx.x & 63);
      for (int r = a.type_rowptr[a.T] + (int)(threadIdx.x >> 6); r < a.N; r += NW) {
        a.x1[(size_t)r * a.H + c] = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) a.vec1[((size_t)r * 3 + d) * a.H + c] = 0.f;
      }
    }
    return;
  }
  stage_weights<NW * 64>(a, t, cb, wl, mu, WIN ? a.win_base : 0, tile_rows);
  // more than 8 waves per workgroup live on <= 168 VGPRs: with vec rows in flight the six bias vectors (24 registers
  // at VW = 4) then stay in LDS and are re-read where they are used (layer 0, without vec rows, fits 16 waves as is)
  constexpr bool LEAN = HAS_VEC && NW > 8 && VW == 4;
  float* lbias = reinterpret_cast<float*>(tapbase + 16 * 4 * 16);
  if (LEAN && threadIdx.x < HN_CB) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const size_t o = (size_t)t * 3 * a.H + p * a.H + cb * HN_CB + threadIdx.x;
      lbias[p * HN_CB + threadIdx.x] = a.brbf[o];
      lbias[(3 + p) * HN_CB + threadIdx.x] = a.xh_bias ? a.xh_bias[o] : 0.f;
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / LPE;            // lane group = edge slot
  const int gl = lane % LPE;
  float2* tb = tapbase + (wave * VW + grp) * 16;   // this lane group's tap scratch
  const int H = a.H;
  const int col = cb * HN_CB + VW * gl;  // first of this lane's VW channels
  const float inv_sqrt3h = 0.57735026918962576f * rsqrtf((float)H);  // (1/sqrt3)(1/sqrtH)
  const float inv_sqrth = rsqrtf((float)H);
  const float* xh_t = a.xh + (size_t)t * a.Nsrc * 3 * H;

  Vec<VW> bias_r[3], xbias_r[3];
  if (!LEAN) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      bias_r[p] = Vec<VW>::load(a.brbf + (size_t)t * 3 * H + p * H + col);
      xbias_r[p] = a.xh_bias ? Vec<VW>::load(a.xh_bias + (size_t)t * 3 * H + p * H + col) : Vec<VW>::zero();
    }
  }
  auto bias_of = [&](int p) { return LEAN ? Vec<VW>::load(lbias + p * HN_CB + VW * gl) : bias_r[p]; };
  auto xbias_of = [&](int p) { return LEAN ? Vec<VW>::load(lbias + (3 + p) * HN_CB + VW * gl) : xbias_r[p]; };

  for (int r = r0 + wave; r < r1; r += NW) {
    const int beg = a.csr_rowptr[r], end = a.csr_rowptr[r + 1];
    Vec<VW> ax = Vec<VW>::zero();
    Vec<VW> av[3] = {Vec<VW>::zero(), Vec<VW>::zero(), Vec<VW>::zero()};

    for (int base = beg; base < end; base += 64) {
      const int cnt = min(64, end - base);
      // one coalesced index load per 64 edges; lane groups pick their entries with a lane shuffle
      const int my_src = lane < cnt ? a.csr_src[base + lane] : 0;
      const int nit = (cnt + VW - 1) / VW;

      // PF = 0: rows and geometry of the next edges are requested at the end of an iteration;
      // PF = 1: both at the top (full prefetch, costs a second register set);
      // PF = 2: geometry (16 B, decides taps and LDS rows) at the top, rows at the end -- the tap and
      //         S0 arithmetic of the next iteration then starts at once while its rows are in flight.
      auto load_geom = [&](int it, FwdIn<HAS_VEC, VW>& in) {
        const int idx = VW * it + grp;
        in.live = idx < cnt;
        in.g = a.edge[base + (in.live ? idx : 0)];
      };
      auto load_rows = [&](int it, FwdIn<HAS_VEC, VW>& in) {
        const int j = __shfl(my_src, VW * it + grp, 64);
        const float* xr = xh_t + (size_t)j * 3 * H + col;
        // (masked at use, not here: touching the registers now would wait for the loads)
        in.xs = Vec<VW>::load(xr); in.xa = Vec<VW>::load(xr + H); in.xb = Vec<VW>::load(xr + 2 * H);
        if (HAS_VEC) {
          const float* vr = a.vec + (size_t)j * 3 * H + col;
#pragma unroll
          for (int d = 0; d < 3; ++d) in.vj[d] = Vec<VW>::load(vr + d * H);
        }
      };
      auto load_edges = [&](int it) {
        FwdIn<HAS_VEC, VW> in;
        load_geom(it, in);
        load_rows(it, in);
        return in;
      };

      FwdIn<HAS_VEC, VW> cur = load_edges(0);
      for (int it = 0; it < nit; ++it) {
        FwdIn<HAS_VEC, VW> nxt;
        if (PF == 1) {
          // (the last iteration re-loads its own: no branch, so the loads stay asynchronous until the
          // register copy at the end of the iteration)
          nxt = load_edges(min(it + 1, nit - 1));
          __builtin_amdgcn_sched_barrier(0);
        } else if (PF == 2) {
          load_geom(min(it + 1, nit - 1), nxt);
          __builtin_amdgcn_sched_barrier(0);
        }
        const float u = cur.g.w * a.inv_rc;
        const HnEnv env = hn_envelope(u, a.env_kind, a.env_p);
        const int lo = hn_window_lo(u, a.R);
        int trow = lo + HN_PAD;                               // padded tile row of tap 0
        bool own = true;
        if (WIN) { own = trow >= a.win_lo && trow < a.win_hi; trow = min(max(trow - a.win_base, 0), a.win_rows - HN_TAPS); }
        const float* wcol = wl + trow * HN_LDS_ROW + VW * gl;
        float g[HN_TAPS], gd[HN_TAPS];
        Vec<VW> S0p[3], S1p[3];
        coop_taps(mu, tb, lo, u, a.coeff, gl);
        if (FUSED) rbf_all<false, VW>(wcol, tb, S0p, S1p);
        else rbf_taps<false>(tb, g, gd);
        HN_SB;
        // padding slots (segment length not a multiple of VW) contribute nothing: every term is
        // linear in rbfh = bias + env * S0 (rmnet.py:55), so scale it by 0 for them
        const float lv = (cur.live && own) ? 1.0f : 0.0f;
        const float ev = env.val * lv;
        Vec<VW> S0, S1;
        // message (rmnet.py:61-67), one part at a time to keep the register footprint small
        if (FUSED) S0 = S0p[0]; else rbf_part<false, VW>(wcol, g, gd, S0, S1);   // part s -> dx
        ax = v_fma(v_add(cur.xs, xbias_of(0)), v_sfma(ev, S0, v_scale(bias_of(0), lv)), ax);
        HN_SB;
        if (FUSED) S0 = S0p[2]; else rbf_part<false, VW>(wcol + 2 * HN_CB, g, gd, S0, S1);   // part b -> rhat term
        const Vec<VW> mb = v_scale(v_mul(v_add(cur.xb, xbias_of(2)), v_sfma(ev, S0, v_scale(bias_of(2), lv))), inv_sqrth);
        av[0] = v_sfma(cur.g.x, mb, av[0]);
        av[1] = v_sfma(cur.g.y, mb, av[1]);
        av[2] = v_sfma(cur.g.z, mb, av[2]);
        HN_SB;
        if (HAS_VEC) {
          if (FUSED) S0 = S0p[1]; else rbf_part<false, VW>(wcol + HN_CB, g, gd, S0, S1);   // part a -> vec_j term
          const Vec<VW> ma = v_scale(v_mul(v_add(cur.xa, xbias_of(1)), v_sfma(ev, S0, v_scale(bias_of(1), lv))), inv_sqrt3h);
#pragma unroll
          for (int d = 0; d < 3; ++d) av[d] = v_fma(cur.vj[d], ma, av[d]);
        }
        HN_SB;
        if (PF == 1) cur = nxt;
        else if (PF == 2) { load_rows(min(it + 1, nit - 1), cur); cur.g = nxt.g; cur.live = nxt.live; }
        else if (it + 1 < nit) cur = load_edges(it + 1);
      }
    }
    // combine the lane groups, then the residual epilogue (rmnet.py:24-26).
    const int rres = a.res_row ? a.res_row[r] : r;      // row of (x, vec) that enters the residual
    if constexpr (VW == 4) {
      // rows_reduce4 leaves ONE channel (col + grp) of every reduced row in each lane: dword accesses, all lanes
      const int c1 = col + grp;
      const size_t xo = (size_t)r * H + c1;
      if (WIN && a.win_accumulate) {          // the other launch wrote residual + its edges' sums
        a.x1[xo] += rows_reduce4(ax) * 0.70710678118654752f;
#pragma unroll
        for (int d = 0; d < 3; ++d) a.vec1[((size_t)r * 3 + d) * H + c1] += rows_reduce4(av[d]);
        continue;
      }
      a.x1[xo] = (a.x[(size_t)rres * H + c1] + rows_reduce4(ax)) * 0.70710678118654752f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const size_t vo = ((size_t)r * 3 + d) * H + c1;
        a.vec1[vo] = (HAS_VEC ? a.vec[((size_t)rres * 3 + d) * H + c1] : 0.f) + rows_reduce4(av[d]);
      }
    } else {
      // the four output rows (x1, vec1[0..2]) are spread over the lane groups (256 B per group and row)
      ax = groups_sum<VW>(ax);
#pragma unroll
      for (int d = 0; d < 3; ++d) av[d] = groups_sum<VW>(av[d]);
      for (int o = grp; o < 4; o += VW) {     // o = 0: x1, o = 1..3: vec1[o-1]
        if (o == 0) {
          const size_t xo = (size_t)r * H + col;
          v_scale(v_add(Vec<VW>::load(a.x + (size_t)rres * H + col), ax), 0.70710678118654752f).store(a.x1 + xo);
        } else {
          const size_t vo = ((size_t)r * 3 + (o - 1)) * H + col;
          const Vec<VW> acc = o == 1 ? av[0] : (o == 2 ? av[1] : av[2]);
          Vec<VW> v0 = Vec<VW>::zero();
          if (HAS_VEC) v0 = Vec<VW>::load(a.vec + ((size_t)rres * 3 + (o - 1)) * H + col);
          v_add(v0, acc).store(a.vec1 + vo);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward (forces): one workgroup = (column block, chunk of SOURCE rows); it loops over the
// relations of the targets, restaging the weight tile per relation.  Sums keyed by the source row
// stay in registers for the whole CSC segment: gxh[t][row] is written once per relation; gvec[row]
// is written at t = 0 (with the residual's identity term) and read-modify-written by the same
// lanes for t > 0.  The per-edge cross-lane sum (dE/dD, 3 floats over one lane group) is 4 DPP
// steps (+ one cross-row exchange for VW = 2).
// ------------------------------------------------------------------------------------------
template <int VW>
struct BwdIn {
  float4 g;         // (rx, ry, rz, d)
  Vec<VW> gx1;      // d/dx1 of the target row
  Vec<VW> gd[3];    // d/dvec1 of the target row
  int pos;
  bool live;
};

template <bool HAS_VEC, int NW, int VW, int PF, bool FUSED>
__global__ __launch_bounds__(NW * 64, NW / 4) void message_scatter_bwd_kernel(MsgArgs a) {
  extern __shared__ __align__(16) float lds[];
  float* wl = lds;
  float* mu = lds + (a.R + 2 * HN_PAD + 1) * HN_LDS_ROW;
  float2* tapbase = reinterpret_cast<float2*>(mu + ((a.R + 2 * HN_PAD + 1 + 3) & ~3));
  constexpr int LPE = 64 / VW;

  const int cb = blockIdx.y;
  const int r0 = blockIdx.x * a.rows_per_block;
  const int r1 = min(r0 + a.rows_per_block, a.N);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / LPE;
  const int gl = lane % LPE;
  float2* tb = tapbase + (wave * VW + grp) * 16;
  const int H = a.H;
  const int col = cb * HN_CB + VW * gl;
  const int nk = a.type_rowptr[a.T];     // rows below nk have a known type (are targets)
  const float inv_sqrt3h = 0.57735026918962576f * rsqrtf((float)H);
  const float inv_sqrth = rsqrtf((float)H);
  const float inv_sqrt2 = 0.70710678118654752f;
  float4* gedge = a.gedge + (size_t)cb * a.E;

  const int t_lo = a.split_t ? blockIdx.z : 0, t_hi = a.split_t ? blockIdx.z + 1 : a.T;
#if defined(HN_STAMPS)
  unsigned long long st_stage = 0, st_pro = 0, st_iter = 0, st_epi = 0, st_segs = 0;
  HN_T(st_k0);
#endif
  for (int t = t_lo; t < t_hi; ++t) {
#if defined(HN_STAMPS)
    HN_T(st_s0);
#endif
    __syncthreads();   // previous tile no longer in use
    stage_weights<NW * 64>(a, t, cb, wl, mu, 0, a.R + 2 * HN_PAD + 1);
    __syncthreads();
#if defined(HN_STAMPS)
    HN_T(st_s1);
    st_stage += st_s1 - st_s0;
#endif
    const float* xh_t = a.xh + (size_t)t * a.N * 3 * H;
    float* gxh_t = a.gxh + (size_t)t * a.N * 3 * H;
    Vec<VW> bias[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bias[p] = Vec<VW>::load(a.brbf + (size_t)t * 3 * H + p * H + col);

    for (int r = r0 + wave; r < r1; r += NW) {
#if defined(HN_STAMPS)
      HN_T(st_a);
#endif
      const int beg = a.csc_rowptr[(size_t)t * a.N + r], end = a.csc_rowptr[(size_t)t * a.N + r + 1];
      const float* xr = xh_t + (size_t)r * 3 * H + col;
      Vec<VW> xs = Vec<VW>::load(xr), xa = Vec<VW>::load(xr + H), xb = Vec<VW>::load(xr + 2 * H);
      if (a.xh_bias) {   // per source segment: free next to the per-edge work
        const float* xbp = a.xh_bias + (size_t)t * 3 * H + col;
        xs = v_add(xs, Vec<VW>::load(xbp)); xa = v_add(xa, Vec<VW>::load(xbp + H)); xb = v_add(xb, Vec<VW>::load(xbp + 2 * H));
      }
      Vec<VW> vj[3] = {Vec<VW>::zero(), Vec<VW>::zero(), Vec<VW>::zero()};
      if (HAS_VEC) {
        const float* vr = a.vec + (size_t)r * 3 * H + col;
#pragma unroll
        for (int d = 0; d < 3; ++d) vj[d] = Vec<VW>::load(vr + d * H);
      }
      Vec<VW> gs = Vec<VW>::zero(), ga = Vec<VW>::zero(), gb = Vec<VW>::zero();
      Vec<VW> gv[3] = {Vec<VW>::zero(), Vec<VW>::zero(), Vec<VW>::zero()};
      // gvec[row] is read-modify-written across the relations (t = 0 starts from the residual's identity term):
      // the old value is requested here, a whole segment ahead of its use
      float prev1[3] = {0.f, 0.f, 0.f};
      if (HAS_VEC && VW == 4) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const size_t vo = ((size_t)r * 3 + d) * H + col + grp;
          if (t == 0) prev1[d] = r < nk ? a.gvec1[vo] : 0.f;
          else if (!a.split_t) prev1[d] = a.gvec[vo];
        }
      }

      for (int base = beg; base < end; base += 64) {
        const int cnt = min(64, end - base);
        const int my_tgt = lane < cnt ? a.csc_tgt[base + lane] : 0;
        const int my_pos = lane < cnt ? a.csc_pos[base + lane] : 0;
        const int nit = (cnt + VW - 1) / VW;

        auto load_geom = [&](int it, BwdIn<VW>& in) {
          const int idx = VW * it + grp;
          in.pos = __shfl(my_pos, idx, 64);
          in.live = idx < cnt;
          in.g = a.edge[in.pos];
        };
        auto load_rows = [&](int it, BwdIn<VW>& in) {
          const int i = __shfl(my_tgt, VW * it + grp, 64);
          in.gx1 = Vec<VW>::load(a.gx1 + (size_t)i * H + col);
          const float* gvr = a.gvec1 + (size_t)i * 3 * H + col;
#pragma unroll
          for (int d = 0; d < 3; ++d) in.gd[d] = Vec<VW>::load(gvr + d * H);   // padding slots are masked at use
        };
        auto load_edges = [&](int it) {
          BwdIn<VW> in;
          load_geom(it, in);
          load_rows(it, in);
          return in;
        };

        BwdIn<VW> cur = load_edges(0);
#if defined(HN_STAMPS)
        HN_T(st_b);
        st_pro += st_b - st_a;
#endif
        for (int it = 0; it < nit; ++it) {
          BwdIn<VW> nxt;
          if (PF == 1) {
            nxt = load_edges(min(it + 1, nit - 1));   // prefetch modes: see the forward kernel
            __builtin_amdgcn_sched_barrier(0);
          } else if (PF == 2) {
            load_geom(min(it + 1, nit - 1), nxt);
            __builtin_amdgcn_sched_barrier(0);
          }
          const float4 g = cur.g;
          const float u = g.w * a.inv_rc;
          const HnEnv env = hn_envelope(u, a.env_kind, a.env_p);
          const int lo = hn_window_lo(u, a.R);
          const float* wcol = wl + (lo + HN_PAD) * HN_LDS_ROW + VW * gl;
          float g_[HN_TAPS], gd_[HN_TAPS];
          Vec<VW> S0p[3], S1p[3];
          coop_taps(mu, tb, lo, u, a.coeff, gl);
          if (FUSED) rbf_all<true, VW>(wcol, tb, S0p, S1p);
          else rbf_taps<true>(tb, g_, gd_);
          HN_SB;
          // d rbfh / d d = inv_rc * (env' S0 + env * 2 coeff S1)
          const float c0 = a.inv_rc * env.der, c1 = a.inv_rc * env.val * 2.0f * a.coeff;
          const float rd[3] = {g.x, g.y, g.z};
          // padding slots: every term is linear in (gx1, gvec1), so scale those by 0
          const float lv = cur.live ? 1.0f : 0.0f;
          const Vec<VW> gdx = v_scale(cur.gx1, lv * inv_sqrt2);
          const Vec<VW> g0 = v_scale(cur.gd[0], lv), g1 = v_scale(cur.gd[1], lv), g2 = v_scale(cur.gd[2], lv);
          Vec<VW> pdv;                       // partial dE/dd per channel
          Vec<VW> S0, S1;
          // ---- part s: dx = sum xs * rs
          if (FUSED) { S0 = S0p[0]; S1 = S1p[0]; } else rbf_part<true, VW>(wcol, g_, gd_, S0, S1);
          {
            const Vec<VW> rs = v_sfma(env.val, S0, bias[0]);
            const Vec<VW> drs = v_sfma(c0, S0, v_scale(S1, c1));
            gs = v_fma(gdx, rs, gs);
            pdv = v_mul(v_mul(gdx, xs), drs);
          }
          HN_SB;
          // ---- part a: dvec += vec_j * (xa * ra) / sqrt(3H)
          if (HAS_VEC) {
            if (FUSED) { S0 = S0p[1]; S1 = S1p[1]; } else rbf_part<true, VW>(wcol + HN_CB, g_, gd_, S0, S1);
            const Vec<VW> ra = v_sfma(env.val, S0, bias[1]);
            const Vec<VW> dra = v_sfma(c0, S0, v_scale(S1, c1));
            const Vec<VW> A = v_scale(v_fma(g0, vj[0], v_fma(g1, vj[1], v_mul(g2, vj[2]))), inv_sqrt3h);
            ga = v_fma(A, ra, ga);
            const Vec<VW> w = v_scale(v_mul(xa, ra), inv_sqrt3h);
            gv[0] = v_fma(g0, w, gv[0]);
            gv[1] = v_fma(g1, w, gv[1]);
            gv[2] = v_fma(g2, w, gv[2]);
            pdv = v_fma(v_mul(A, xa), dra, pdv);
          }
          HN_SB;
          // ---- part b: dvec += rhat * (xb * rb) / sqrt(H)
          Vec<VW> q;
          {
            if (FUSED) { S0 = S0p[2]; S1 = S1p[2]; } else rbf_part<true, VW>(wcol + 2 * HN_CB, g_, gd_, S0, S1);
            const Vec<VW> rb = v_sfma(env.val, S0, bias[2]);
            const Vec<VW> drb = v_sfma(c0, S0, v_scale(S1, c1));
            const Vec<VW> B = v_scale(v_sfma(rd[0], g0, v_sfma(rd[1], g1, v_scale(g2, rd[2]))), inv_sqrth);
            gb = v_fma(B, rb, gb);
            pdv = v_fma(v_mul(B, xb), drb, pdv);
            q = v_scale(v_mul(xb, rb), inv_sqrth);
          }
          HN_SB;
          const float pd = v_hsum(pdv);
          const float pr[3] = {v_hsum(v_mul(g0, q)), v_hsum(v_mul(g1, q)), v_hsum(v_mul(g2, q))};   // dE/d rhat
          // Cartesian gradient w.r.t. D (d = |D|, rhat = D/d): gD = pd rhat + (pr - (pr.rhat) rhat)/d
          const float dotp = pr[0] * rd[0] + pr[1] * rd[1] + pr[2] * rd[2];
          const float invd = __builtin_amdgcn_rcpf(g.w);   // 1 ulp; the full-precision divide costs ~10 instructions
          float gD[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) gD[d] = group_allsum<VW>(fmaf(pd - dotp * invd, rd[d], pr[d] * invd));
          if (cur.live && gl == 0) gedge[cur.pos] = make_float4(gD[0], gD[1], gD[2], 0.f);
          HN_SB;
          if (PF == 1) cur = nxt;
          else if (PF == 2) { load_rows(min(it + 1, nit - 1), cur); cur.g = nxt.g; cur.pos = nxt.pos; cur.live = nxt.live; }
          else if (it + 1 < nit) cur = load_edges(it + 1);
        }
#if defined(HN_STAMPS)
        HN_T(st_c);
        st_iter += st_c - st_b;
        st_a = st_c;
#endif
      }
      // combine lane groups; the output rows (gxh s/a/b, gx | gvec[0..2]) are spread over the groups
      const bool known = r < nk;
      if constexpr (VW == 4) {
        // every lane ends up with ONE channel (col + grp) of each reduced row: dword stores, all lanes active
        const int c1 = col + grp;
        float* go = gxh_t + (size_t)r * 3 * H + c1;
        go[0] = rows_reduce4(gs);
        go[H] = rows_reduce4(ga);
        go[2 * H] = rows_reduce4(gb);
        if (t == 0) a.gx[(size_t)r * H + c1] = (known ? a.gx1[(size_t)r * H + c1] : 0.f) * inv_sqrt2;
        if (HAS_VEC) {
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const size_t vo = ((size_t)r * 3 + d) * H + c1;
            a.gvec[(a.split_t ? (size_t)t * a.N * 3 * H : 0) + vo] =
                prev1[d] + rows_reduce4(gv[d]);
          }
        }
      } else {
        gs = groups_sum<VW>(gs); ga = groups_sum<VW>(ga); gb = groups_sum<VW>(gb);
        if (grp == 0) {
          float* go = gxh_t + (size_t)r * 3 * H + col;
          gs.store(go); ga.store(go + H); gb.store(go + 2 * H);
          if (t == 0) {   // residual identity: gx = gx1 / sqrt2 on rows that are targets
            const Vec<VW> g1 = known ? Vec<VW>::load(a.gx1 + (size_t)r * H + col) : Vec<VW>::zero();
            v_scale(g1, inv_sqrt2).store(a.gx + (size_t)r * H + col);
          }
        }
        if (HAS_VEC) {
#pragma unroll
          for (int d = 0; d < 3; ++d) gv[d] = groups_sum<VW>(gv[d]);
          // gvec[d] handled by lane group (d + 1) % VW  (VW = 2: groups 1,0,1)
          for (int d = 0; d < 3; ++d) {
            if (((d + 1) % VW) != grp) continue;
            const size_t vo = ((size_t)r * 3 + d) * H + col;
            Vec<VW> prev;
            if (t == 0) prev = known ? Vec<VW>::load(a.gvec1 + vo) : Vec<VW>::zero();
            else if (a.split_t) prev = Vec<VW>::zero();
            else prev = Vec<VW>::load(a.gvec + vo);
            const Vec<VW> add = d == 0 ? gv[0] : (d == 1 ? gv[1] : gv[2]);
            v_add(prev, add).store(a.gvec + (a.split_t ? (size_t)t * a.N * 3 * H : 0) + vo);
          }
        }
      }
#if defined(HN_STAMPS)
      HN_TNW(st_d);
      st_epi += st_d - st_a;
      st_segs += 1;
#endif
    }
  }
#if defined(HN_STAMPS)
  HN_T(st_k1);
  if (lane == 0) {
    atomicAdd(&hn_dbg[0], st_k1 - st_k0); atomicAdd(&hn_dbg[1], st_stage); atomicAdd(&hn_dbg[2], st_pro);
    atomicAdd(&hn_dbg[3], st_iter); atomicAdd(&hn_dbg[4], st_epi); atomicAdd(&hn_dbg[5], st_segs);
    atomicAdd(&hn_dbg[6], 1ull);
  }
#endif
}

int fill_args(const hn_graph* g, const hn_rbf_desc* rbf, int hidden, MsgArgs& a) {
  if (!g || !rbf || hidden <= 0 || hidden % HN_CB != 0) return HN_ERR_BAD_ARG;
  if (g->num_rel <= 0 || rbf->num_rbf < 2) return HN_ERR_BAD_ARG;
  a.N = g->num_nodes; a.E = g->num_edges; a.T = g->num_rel;
  a.Nsrc = g->num_src > 0 ? g->num_src : g->num_nodes; a.res_row = g->res_row;
  a.type_rowptr = g->type_rowptr; a.csr_rowptr = g->csr_rowptr; a.csr_src = g->csr_src;
  a.csc_rowptr = g->csc_rowptr; a.csc_tgt = g->csc_tgt; a.csc_pos = g->csc_pos;
  a.offset = rbf->offset; a.R = rbf->num_rbf; a.inv_rc = rbf->inv_rc; a.coeff = rbf->coeff;
  a.env_kind = rbf->env_kind; a.env_p = rbf->env_p;
  a.H = hidden;
  return HN_OK;
}

// tuning knobs and alternative forms: process-wide options (host_api.cpp: hermnet_set_option), read at every launch

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

// Rows per workgroup such that the grid is a whole number of "rounds" of one workgroup per CU
// (the 116 KB weight tile allows one resident workgroup per CU): `work` = rows x column blocks,
// `slack` = workgroups lost to per-relation rounding.  Targets ~64 rows per workgroup (each workgroup stages
// the 116 KB weight tile once per relation: fewer, longer workgroups amortise it; measured on config 2).
int pick_rows(int rows, int ncb, int slack, int override_rows) {
  if (override_rows > 0) return override_rows;
  const long work = (long)rows * ncb;
  const int cus = num_cus();
  long rounds = (work + (long)cus * 32) / ((long)cus * 64);
  if (rounds < 1) rounds = 1;
  long wgs = (long)cus * rounds - slack;
  if (wgs < 1) wgs = 1;
  long rpb = (work + wgs - 1) / wgs;
  if (rpb < 8) rpb = 8;
  return (int)rpb;
}

// LDS image: weight tile [rows][192] | tap centres mu [rows, padded to 4] | per-wave tap scratch
// (float2 {g, g*diff} x 16 per lane group).
// (`tile_rows`: rows of the weight tile held at once -- all of them unless the launch works on a tap-row window)
size_t lds_bytes(int R, int tile_rows = 0) {
  const size_t rows = (size_t)(R + 2 * HN_PAD + 1);
  const size_t held = tile_rows > 0 ? (size_t)tile_rows : rows;
  return (held * HN_LDS_ROW + ((rows + 3) & ~(size_t)3)) * sizeof(float) + 16 * 4 * 16 * sizeof(float2) +
         6 * HN_CB * sizeof(float);      // + the column block's rbf_proj / x_proj biases (register-lean variants)
}

typedef void (*kern_t)(MsgArgs);

// Opt in to > 64 KB of dynamic LDS once per kernel (not per launch: it is a host-side attribute, and
// per-launch calls are neither free nor welcome while a stream is being captured into a hipGraph).
int ensure_lds(kern_t k, size_t lds) {
  constexpr int kMax = 64;
  static kern_t done[kMax];
  static size_t done_lds[kMax];
  static int ndone = 0;
  for (int i = 0; i < ndone; ++i)
    if (done[i] == k && done_lds[i] >= lds) return HN_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
      hipSuccess)
    return HN_ERR_LDS;
  if (ndone < kMax) { done[ndone] = k; done_lds[ndone] = lds; ++ndone; }
  return HN_OK;
}

// variant = waves * 1000 + VW * 100 + prefetch * 10 + fused-tap-loop
template <bool HAS_VEC>
kern_t pick_fwd(int variant, int& nw) {
  switch (variant) {
    case 16200: nw = 16; return message_scatter_fwd_kernel<HAS_VEC, 16, 2, 0, false>;
    case 16201: nw = 16; return message_scatter_fwd_kernel<HAS_VEC, 16, 2, 0, true>;
    case 16210: nw = 16; return message_scatter_fwd_kernel<HAS_VEC, 16, 2, 1, false>;
    case 16211: nw = 16; return message_scatter_fwd_kernel<HAS_VEC, 16, 2, 1, true>;
    case 8210:  nw = 8;  return message_scatter_fwd_kernel<HAS_VEC, 8, 2, 1, false>;
    case 8400:  nw = 8;  return message_scatter_fwd_kernel<HAS_VEC, 8, 4, 0, false>;
    case 8420:  nw = 8;  return message_scatter_fwd_kernel<HAS_VEC, 8, 4, 2, false>;
    case 12420: nw = 12; return message_scatter_fwd_kernel<HAS_VEC, 12, 4, 2, false>;
    case 12400: nw = 12; return message_scatter_fwd_kernel<HAS_VEC, 12, 4, 0, false>;
    case 12421: nw = 12; return message_scatter_fwd_kernel<HAS_VEC, 12, 4, 2, true>;
    case 12410: nw = 12; return message_scatter_fwd_kernel<HAS_VEC, 12, 4, 1, false>;
    case 16420: nw = 16; return message_scatter_fwd_kernel<HAS_VEC, 16, 4, 2, false>;
    case 16400: nw = 16; return message_scatter_fwd_kernel<HAS_VEC, 16, 4, 0, false>;
    case 16221: nw = 16; return message_scatter_fwd_kernel<HAS_VEC, 16, 2, 2, true>;
    case 16220: nw = 16; return message_scatter_fwd_kernel<HAS_VEC, 16, 2, 2, false>;
    default:    nw = 8;  return message_scatter_fwd_kernel<HAS_VEC, 8, 4, 1, false>;    // 8410
  }
}

template <bool HAS_VEC>
kern_t pick_bwd(int variant, int& nw) {
  switch (variant) {
    case 16200: nw = 16; return message_scatter_bwd_kernel<HAS_VEC, 16, 2, 0, false>;
    case 16201: nw = 16; return message_scatter_bwd_kernel<HAS_VEC, 16, 2, 0, true>;
    case 16210: nw = 16; return message_scatter_bwd_kernel<HAS_VEC, 16, 2, 1, false>;
    case 16211: nw = 16; return message_scatter_bwd_kernel<HAS_VEC, 16, 2, 1, true>;
    case 8400:  nw = 8;  return message_scatter_bwd_kernel<HAS_VEC, 8, 4, 0, false>;
    case 8420:  nw = 8;  return message_scatter_bwd_kernel<HAS_VEC, 8, 4, 2, false>;
    case 16221: nw = 16; return message_scatter_bwd_kernel<HAS_VEC, 16, 2, 2, true>;
    case 16220: nw = 16; return message_scatter_bwd_kernel<HAS_VEC, 16, 2, 2, false>;
    case 8410:  nw = 8;  return message_scatter_bwd_kernel<HAS_VEC, 8, 4, 1, false>;
    case 8201:  nw = 8;  return message_scatter_bwd_kernel<HAS_VEC, 8, 2, 0, true>;
    default:    nw = 8;  return message_scatter_bwd_kernel<HAS_VEC, 8, 2, 1, false>;    // 8210
  }
}

}  // namespace

extern "C" int hermnet_message_scatter_fwd(const hn_graph* g, const hn_rbf_desc* rbf, int hidden,
                                           const float* xh, const float* xh_bias, const float* vec, const float* x,
                                           const float* wt, const float* brbf, const float* edge,
                                           float* x1, float* vec1, const int* target_ranges, int zero_unknown_rows,
                                           int range_rows, void* stream) {
  MsgArgs a = {};
  int rc = fill_args(g, rbf, hidden, a);
  if (rc) return rc;
  if (!xh || !x || !wt || !brbf || !x1 || !vec1 || (g->num_edges > 0 && !edge)) return HN_ERR_BAD_ARG;
  if (a.N == 0) return HN_OK;
  a.xh = xh; a.xh_bias = xh_bias; a.vec = vec; a.x = x; a.wt = wt; a.brbf = brbf;
  a.edge = reinterpret_cast<const float4*>(edge);
  a.x1 = x1; a.vec1 = vec1;
  a.row_ranges = target_ranges; a.zero_unknown = (target_ranges == nullptr) || zero_unknown_rows;
  const int rpb_fwd = hn_option(HN_OPT_FWD_ROWS);
  // defaults from tools/kbench.py on MI355X (config 2): see DESIGN.md "Kernel variants"
  const int variant_vec = hn_option(HN_OPT_FWD_VARIANT);
  const int variant_l0 = hn_option(HN_OPT_FWD_VARIANT_L0);
  const int variant = vec ? variant_vec : variant_l0;
  // (a launch over row ranges sizes its workgroups for the rows it covers: a workgroup of the full launch lives as long
  // as the whole kernel, so a launch over a tenth of the rows with the same chunking would take just as long)
  const int rows = (target_ranges && range_rows > 0 && range_rows < a.N) ? range_rows : a.N;
  a.rows_per_block = pick_rows(rows, hidden / HN_CB, a.T * (hidden / HN_CB), rpb_fwd);
  a.xcd_remap = 1;
  // blocks: sum_t ceil(N_t / rpb) <= N / rpb + T, plus one surplus block that zeroes unknown rows
  dim3 grid((unsigned)(rows / a.rows_per_block + a.T + 1), (unsigned)(hidden / HN_CB));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  size_t lds = lds_bytes(a.R);
  if (lds > 160 * 1024) {
    // num_rbf > 176: the tile of one (relation, column block) no longer fits the LDS.  Two launches over tap-row windows
    // (the same split as the backward's, message_bwd_cl.hip: hn_bwd_cl_launch): the first owns the edges whose first tap
    // row is below `split`, writes residual + their sums; the second owns the rest and adds its sums.
    const int rows_all = a.R + 2 * HN_PAD + 1, split = (rows_all - HN_PAD + 1) / 2;
    const int held = split + HN_PAD > rows_all - split ? split + HN_PAD : rows_all - split;
    lds = lds_bytes(a.R, held);
    if (lds > 160 * 1024) return HN_ERR_LDS;
    kern_t k = vec ? message_scatter_fwd_kernel<true, 8, 4, 2, false, true> : message_scatter_fwd_kernel<false, 16, 4, 2, false, true>;
    const int nw = vec ? 8 : 16;
    if (ensure_lds(k, lds) != HN_OK) return HN_ERR_LDS;
    a.win_base = 0; a.win_rows = split + HN_PAD; a.win_lo = 0; a.win_hi = split; a.win_accumulate = 0;
    hipLaunchKernelGGL(k, grid, dim3(nw * 64), lds, s, a);
    a.win_base = split; a.win_rows = rows_all - split; a.win_lo = split; a.win_hi = 0x7fffffff; a.win_accumulate = 1;
    a.zero_unknown = 0;
    hipLaunchKernelGGL(k, grid, dim3(nw * 64), lds, s, a);
    return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
  }
  int nw = 16;
  kern_t k = vec ? pick_fwd<true>(variant, nw) : pick_fwd<false>(variant, nw);
  if (ensure_lds(k, lds) != HN_OK) return HN_ERR_LDS;
  hipLaunchKernelGGL(k, grid, dim3(nw * 64), lds, s, a);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_message_scatter_bwd(const hn_graph* g, const hn_rbf_desc* rbf, int hidden,
                                           const float* xh, const float* xh_bias, const float* vec,
                                           const float* wt, const float* brbf, const float* edge,
                                           const float* gx1, const float* gvec1,
                                           float* gxh, float* gvec, float* gx, float* gedge,
                                           int split_t, const float* edge_table, float* gvec_partials,
                                           const int* source_ranges, const int* source_ranges_host, int num_ranges,
                                           void* stream) {
  MsgArgs a = {};
  int rc = fill_args(g, rbf, hidden, a);
  if (rc) return rc;
  if (!xh || !wt || !brbf || !gx1 || !gvec1 || !gxh || (g->num_edges > 0 && !gedge)) return HN_ERR_BAD_ARG;
  if (vec && !gvec && gx) return HN_ERR_BAD_ARG;
  const bool no_finish = gx == nullptr;        // the consumer sums the partials (include/hermnet_hip.h)
  if (g->num_edges > 0 && !edge) return HN_ERR_BAD_ARG;
  if (a.N == 0) return HN_OK;
  const int use_cl = hn_option(HN_OPT_BWD_LANES16) == 0;
  const bool virtual_targets = g->num_src > 0 || g->res_row != nullptr;
  const size_t gather_bytes = (size_t)a.N * 3 * hidden * sizeof(float);      // the channel-per-lane form gathers through
  const size_t src_rows = g->num_src > 0 ? (size_t)g->num_src : (size_t)a.N;
  const bool cl_ok = edge_table && !split_t && gather_bytes < 0xffffffffull &&  // 32-bit buffer offsets
                     src_rows * 3 * hidden < 0x7fffffffull &&                   // 32-bit row offsets (elements)
                     (!vec || a.T == 1 || gvec_partials);
  if (virtual_targets && !cl_ok) return HN_ERR_BAD_ARG;
  if (num_ranges < 0 || (num_ranges > 0 && !(cl_ok && (use_cl || virtual_targets)))) return HN_ERR_BAD_ARG;   // ranges: that form only
  // (ranged launches without the finishing launch: the "proj" halo exchange, ABI v12 -- the halo source rows first, the rest
  // while their partial sums travel; the consumer sums the partials as for the unranged form)
  if (no_finish && !(cl_ok && use_cl && !virtual_targets && (!vec || gvec_partials)))
    return HN_ERR_BAD_ARG;
  if (cl_ok && (use_cl || virtual_targets)) {
    HnBwdClArgs b = {};
    b.N = a.N; b.Nsrc = g->num_src > 0 ? g->num_src : a.N; b.E = a.E; b.T = a.T;
    b.identity = virtual_targets ? 0 : 1;
    b.csc_rowptr = a.csc_rowptr; b.csc_tgt = a.csc_tgt; b.csc_pos = a.csc_pos;
    b.R = a.R; b.H = hidden; b.table = edge_table; b.edge = reinterpret_cast<const float4*>(edge);
    b.xh = xh; b.xh_bias = xh_bias; b.vec = vec; b.wt = wt; b.brbf = brbf; b.gx1 = gx1; b.gvec1 = gvec1;
    b.gxh = gxh; b.gvec = (a.T == 1 && !no_finish) ? gvec : gvec_partials; b.gvec_out = gvec; b.gx = gx;
    b.gedge = reinterpret_cast<float4*>(gedge);
    b.type_rowptr = g->type_rowptr;
    const int rpb_cl = hn_option(HN_OPT_BWD_CL_ROWS);
    b.xcd_remap = 1;
    b.src_ranges = source_ranges; b.num_ranges = num_ranges;
    return hn_bwd_cl_launch(b, vec != nullptr, rpb_cl, source_ranges_host, reinterpret_cast<hipStream_t>(stream));
  }
  a.xh = xh; a.xh_bias = xh_bias; a.vec = vec; a.wt = wt; a.brbf = brbf;
  a.edge = reinterpret_cast<const float4*>(edge);
  a.gx1 = gx1; a.gvec1 = gvec1; a.gxh = gxh; a.gvec = gvec; a.gx = gx;
  a.gedge = reinterpret_cast<float4*>(gedge);
  const int rpb_bwd = hn_option(HN_OPT_BWD_ROWS);
  const int variant_vec = hn_option(HN_OPT_BWD_VARIANT);
  const int variant_l0 = hn_option(HN_OPT_BWD_VARIANT_L0);
  const int variant = vec ? variant_vec : variant_l0;
  a.split_t = split_t ? 1 : 0;
  a.rows_per_block = pick_rows(a.N, (hidden / HN_CB) * (a.split_t ? a.T : 1), 0, rpb_bwd) * (a.split_t ? a.T : 1);
  const size_t lds = lds_bytes(a.R);
  if (lds > 160 * 1024) return HN_ERR_LDS;
  dim3 grid((unsigned)((a.N + a.rows_per_block - 1) / a.rows_per_block), (unsigned)(hidden / HN_CB),
            (unsigned)(a.split_t ? a.T : 1));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int nw = 16;
  kern_t k = vec ? pick_bwd<true>(variant, nw) : pick_bwd<false>(variant, nw);
  if (ensure_lds(k, lds) != HN_OK) return HN_ERR_LDS;
  hipLaunchKernelGGL(k, grid, dim3(nw * 64), lds, s, a);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

#if defined(HN_STAMPS)
// diagnostic builds only: read and reset the stamp sums (total, staging, segment prologue, iterations, epilogue,
// segments, waves)
extern "C" int hermnet_debug_stamps(unsigned long long* out8) {
  if (hipDeviceSynchronize() != hipSuccess) return HN_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(hn_dbg), 8 * sizeof(unsigned long long)) != hipSuccess) return HN_ERR_LAUNCH;
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(hn_dbg), z, sizeof(z)) != hipSuccess) return HN_ERR_LAUNCH;
  return HN_OK;
}
#endif
