// Shared pieces of the node chain kernels (node_chain.hip: widths 64 / 128 / 256; node_chain_wide.hip: every
// multiple of 64 up to 512): MFMA panel loop with streamed weight fragments, buffer-descriptor tile I/O, the
// wave-private LDS transposes of the epilogues, tile bookkeeping and the launcher.  Included inside each
// translation unit's anonymous namespace user: everything here is `static`/inline by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/hermnet_hip.h"

#ifndef HN_NT_SAVED_LOADS
#define HN_NT_SAVED_LOADS 0      // (the read-once side of the same tensors: measured, no gain -- profiles/r06_nt_saved_ab.log)
#endif
// Argument blocks of the four chain kernels (plain structs shared by both translation units)
namespace hn_chain {
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
  const int* src_ranges;   // [T][4] or null: relation t only ever gathers source rows [r0, r1) and [r2, r3)
  int Ns, T, Hr;
  float eps;
  const int* windows;      // [nwin][2] row windows or null (see tile_selected)
  int nwin, wmode;
};

struct PreBwdArgs {
  const float* gxh;    // [T, Ns, 3H]
  const float* hb;     // [T, Ns, H]
  const float* w2tf;   // [T] fragments of W2_t^T [H, 3H]
  const float* w1tf;   // [T] fragments of W1_t^T [H, H]
  float* gn;           // [T, Ns, H]
  const int* src_ranges;   // as in PreFwdArgs; skipped tiles contribute zero rows to gn[t]
  int Ns, T;
  const int* windows;      // as in PreFwdArgs
  int nwin, wmode;
};

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
  float* nrm;               // [N, H]      sqrt(sum_d v2^2 + 1e-8), saved
  float* x_out;             // [N, H]
  float* vec_out;           // [N, 3, H]
  int N, T;
};

// Incoming gradients that still sit in the partial sums of the layer ABOVE (include/hermnet_hip.h: hn_pending_grads): the
// update backward forms them for its own rows first (materialise_pending) instead of two small launches doing it.
struct PendingGrads {
  const float* gn;          // [nparts, N, H]     node_pre_bwd's per-relation partial sums (null: nothing pending)
  const float* gv;          // [nparts, N, 3, H]  message_scatter_bwd's per-relation partial sums
  const float* x;           // [N, H]  that layer's LayerNorm input, mean / rstd [N] its statistics
  const float* mean;
  const float* rstd;
  const float* gx1;         // [N, H]     that layer's update-backward results: the residual's identity terms
  const float* gvec1;       // [N, 3, H]
  int nparts, Hr;
  // fused form (node_chain16.hip: node_pre_bwd of the layer above inside this launch): gn is null, the chain runs on
  const float* gxh;         // [nparts, N, 3H]  that layer's message-backward result (null: gn holds the partial sums)
  const float* hb;          // [nparts, N, H]   its saved pre-activations
  const float* w2tf;        // [nparts] frag16 of W2_t^T [H, 3H]
  const float* w1tf;        // [nparts] frag16 of W1_t^T [H, H]
};

struct UpdBwdArgs {
  const float* gxo;         // [N, H]     (with `pend`: written by this launch before it is read)
  const float* gvo;         // [N, 3, H]
  const float* vp;          // [N, 3, 2H]
  const float* h2b;         // [N, H]
  const float* q23;         // [N, 2H]
  const float* nrm;         // [N, H]
  const float* wx2tf;       // [T] fragments of xvec_proj[2].weight^T [H, 3H]
  const float* wx0tf;       // [T] fragments of xvec_proj[0].weight^T [2H, H]
  const float* wvtf;        // [T] fragments of vec_proj.weight^T [H, 2H]
  const float* row_active;  // [N] or null
  const int* type_rowptr;
  float* gx1;               // [N, H]
  float* gvec1;             // [N, 3, H]
  int N, T;
  PendingGrads pend;
};
}  // namespace hn_chain
using namespace hn_chain;

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

// Diagnostic build (-DHN_STAMPS): wave 0 of every workgroup records the shader clock at phase boundaries into a buffer
// of its own (tools/chain_stamps.py reads it back); no stamp executes in the product build.
#ifdef HN_STAMPS
__device__ unsigned long long hn_stamps[8192 * 16];
#define STAMP_HWID()                                                                                 \
  do {                                                                                               \
    if (threadIdx.x == 0) {                                                                          \
      unsigned hw, xcc;                                                                              \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                               \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                             \
      hn_stamps[((blockIdx.x + gridDim.x * blockIdx.y) & 8191) * 16 + 15] = ((unsigned long long)xcc << 32) | hw; \
    }                                                                                                \
  } while (0)
#define STAMP(k)                                                                                     \
  do {                                                                                               \
    if (threadIdx.x == 0)                                                                            \
      hn_stamps[((blockIdx.x + gridDim.x * blockIdx.y) & 8191) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define STAMP(k)
#define STAMP_HWID()
#endif

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

// ---- B operand: a ring of RS weight fragments per column block, RS - 1 steps ahead of use ----------------------------
// (written for the fp32 MFMAs: a k-group was RB * NJ * 4 MFMAs = RB * NJ * 256 cycles, an L2 hit under load 500-800: three
// groups ahead.  Since the products run as bf16 splits a slot holds one STEP: see mma_panel)
template <int NJ, int RS>
struct BRing { f32x4 v[RS][NJ]; };

// ring slots of a product with RB x NJ accumulator blocks per wave
// (three slots: a k-group is three steps, so the slots of a group's steps are compile-time constants in a loop rolled per GROUP --
// with four the loop body had to be four groups, twelve steps, and the update kernels spilled)
#ifndef HN_RS
#define HN_RS 3
#endif
constexpr int kRS = HN_RS;
constexpr int ring_size(int rb_nj, bool more) { return kRS; }

// Experiment knob (-DHN_STAGGER=n): the workgroup that lands in an ODD wave slot of its SIMD (the second of two
// co-resident ones) starts n x 8128 cycles late, so that the two do not walk through their matrix phases in lockstep.
__device__ __forceinline__ void stagger_start() {
#ifdef HN_STAGGER
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (hw & 1)
    for (int i = 0; i < HN_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 products on the BF16 matrix pipe (round 5).  The fp32 MFMAs (v_mfma_f32_32x32x2_f32 / 16x16x4) deliver 64 FLOP/clk/SIMD,
// the bf16 ones (32x32x16 / 16x16x32) 1024.  An fp32 value split THREE ways into bf16 planes, x = x0 + x1 + x2 EXACTLY (3 x 8
// significant bits; each plane the round-to-nearest-even of what the planes before it leave), and the six largest of the nine
// partial products -- w2 x0, w1 x1, w1 x0, w0 x2, w0 x1, w0 x0, smallest first, accumulated in fp32 by the MFMA; the three dropped
// ones are each <= 2^-24 of the product -- cost 6 / 16 of the fp32 instruction's pipe time at fp32 accuracy (the "BF16x6 / x9"
// emulation of fp32 GEMMs).  Weights are split once on the host (nodeops.weight_fragments / weight_fragments16: three planes in
// stream order, 6 bytes per weight instead of 4); activations stay fp32 in the LDS tiles and are split by the consuming wave in
// registers (~37 VALU instructions per eight values, issued between the MFMAs of the previous k-group).
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 hn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hn_f32x2 __attribute__((ext_vector_type(2)));
#ifndef HN_SPLIT_SCALAR
#define HN_SPLIT_SCALAR 0
#endif
struct Split8 { hn_bf16x8 p[3]; };          // p[0] + p[1] + p[2] == the eight fp32 values (exactly, barring under/overflow)

__device__ __forceinline__ void split8(const f32x4& lo, const f32x4& hi, Split8& o) {
  float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const hn_f32x2 v = {x[e], x[e + 1]};
      const hn_bf16x2 h = __builtin_convertvector(v, hn_bf16x2);          // v_cvt_pk_bf16_f32: round to nearest even
      o.p[p][e] = h[0]; o.p[p][e + 1] = h[1];
      if (p < 2) {
        const hn_f32x2 back = __builtin_convertvector(h, hn_f32x2);
#if HN_SPLIT_SCALAR
        // (A/B, VERDICT r5 item 3: MI355X_MICROARCH.md prices a packed fp32 op beside MFMAs at +22-26 cycles per gap; the
        // residuals as two v_sub_f32 -- the pins keep the SLP vectoriser from re-packing them)
        float r0 = x[e] - back[0], r1 = x[e + 1] - back[1];
        pin(r0); pin(r1);
        x[e] = r0; x[e + 1] = r1;
#else
        x[e] -= back[0]; x[e + 1] -= back[1];                             // exact: the residual fits fp32
#endif
      }
    }
  }
}

// float4s of one 32-channel block's fragment stream over K inputs (K/16 k-groups x 3 planes x 64 lanes); floats of a
// fragment-ordered [O, K] weight
constexpr int frag_f4(int K) { return K * 12; }

// first RS - 1 steps of a stream (call it early: before the barrier / epilogue that precedes the product)
template <int NJ, int RS>
__device__ __forceinline__ void b_preload(BRing<NJ, RS>& r, const f32x4* const (&bp)[NJ]) {
#pragma unroll
  for (int g = 0; g < RS - 1; ++g)
#pragma unroll
    for (int j = 0; j < NJ; ++j) r.v[g][j] = bp[j][g * 64];
}

// acc[rb][j] += W_j[:, 0 .. KP) . A[rows of block rb]^T, W_j streamed from bp[j] (this lane's pointer at step 0 of the panel).
// A block's fragments are consumed as STEPS of one 1-KiB load each (lane l: 16 bytes = 8 bf16 of one weight plane,
// W_p[32 cb + (l & 31)][16 Q + 8 (l >> 5) .. +7]) in the order [k-group Q of 16][plane 2, 1, 0]:
//     frag(W)[((cb * K/16 + Q) * 3 + s) * 64 + l],   step s of a group runs s + 1 MFMAs per accumulator block
// (activation planes s .. 0) -- three steps per 16 k where the fp32 form took two 8-deep groups: 1.5 x the requests and bytes.
// `As`: this lane's LDS read pointer, &tile[(first row of the wave + (l & 31)) * LD + 4 (l >> 5)] (the tile is fp32).
// The ring holds steps 0 .. RS-2 on entry.  MORE: the stream continues behind this panel (next panel of the same
// product): its first RS - 1 steps are requested too and sit in slots 0 .. RS-2 on exit.
// PIN: a step's requests stay in front of its MFMAs (see below); the pre-forward kernels, whose loops hipcc leaves alone
// anyway, ran 5x slower with the fence in place (found by measurement, not understood) and pass false.
#ifndef HN_PIN_LOADS
#define HN_PIN_LOADS 1
#endif
#ifndef HN_PIN_PRE
#define HN_PIN_PRE false     // the pre-forward kernels' products
#endif
// DB: the activation planes of the NEXT k-group are split beside this group's MFMAs (a second set of planes: 12 RB more
// registers -- the projection kernels have them, the update kernels, at their 256-register budgets, do not and pass false).
#ifndef HN_SPLIT_DB
#define HN_SPLIT_DB 1
#endif
template <int KP, int LD, int RB, int NJ, int RS, bool MORE, bool PIN = true, bool DBT = true>
__device__ __forceinline__ void mma_panel(f32x16 (&acc)[RB][NJ], const float* As, const f32x4* const (&bp)[NJ],
                                          BRing<NJ, RS>& ring) {
  constexpr bool DB = DBT && HN_SPLIT_DB;
  // k-groups; groups / steps of one pass of the rolled loop: the fewest groups whose steps fill whole turns of the ring -- an
  // EVEN number of them when the planes are double-buffered (a group's buffer is then a compile-time constant too)
  constexpr int CH0 = RS % 3 == 0 ? RS / 3 : RS;
  constexpr int GT = KP / 16, PF = RS - 1, CH = (DB && CH0 % 2) ? 2 * CH0 : CH0, CS = 3 * CH;
  static_assert(KP % (16 * CH) == 0 && CS % RS == 0 && GT >= CH, "panel / ring mismatch");
  const float* A8 = As + 4 * ((threadIdx.x & 63) >> 5);             // &tile[row * LD + 8 (l >> 5)]
  // The planes of the current k-group and (DB) of the next one: its split runs beside this group's MFMAs instead of in front
  // of its own (measured: H = 512 node_pre_fwd 412 -> 388 us; nothing at H = 128, where two or three workgroups per CU cover
  // each other's splits anyway)
  constexpr int NX = DB ? 2 : 1;
  Split8 X[NX][RB];
  f32x4 lo[RB], hi[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    lo[rb] = *reinterpret_cast<const f32x4*>(A8 + rb * 32 * LD);
    hi[rb] = *reinterpret_cast<const f32x4*>(A8 + rb * 32 * LD + 4);
    if (DB) split8(lo[rb], hi[rb], X[0][rb]);
  }
  // CH k-groups per pass of a ROLLED loop (ring slots are then compile-time constants): a fully unrolled chain kernel is more
  // code than the instruction cache holds, and hipcc's scheduler, handed a whole panel as one region, spills what the caller
  // keeps in flight around the product.
  auto chunk = [&](int g0, bool last) {
#pragma unroll
    for (int gq = 0; gq < CH; ++gq) {
      const bool has_next = !last || gq + 1 < CH;                   // the panel has a group behind this one
      if (!DB) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) split8(lo[rb], hi[rb], X[0][rb]);
      }
      if (has_next) {                                               // its tile values: LDS reads one group ahead
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          lo[rb] = *reinterpret_cast<const f32x4*>(A8 + rb * 32 * LD + 16 * (g0 + gq + 1));
          hi[rb] = *reinterpret_cast<const f32x4*>(A8 + rb * 32 * LD + 16 * (g0 + gq + 1) + 4);
        }
      }
#pragma unroll
      for (int ps = 0; ps < 3; ++ps) {                              // the step of weight plane 2 - ps
        const int sl = 3 * gq + ps;
        if (MORE || !last || sl + PF < CS) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) ring.v[(sl + PF) % RS][j] = bp[j][(3 * g0 + sl + PF) * 64];
        }
        // the requests stay HERE, in front of the step's MFMAs: left alone, hipcc's scheduler sinks a weight request towards
        // its use (fewer live registers) and the product waits for L2 every few steps (round 4: read off the ISA)
        if (PIN && HN_PIN_LOADS) fence_sched();
#pragma unroll
        for (int m = ps; m >= 0; --m)                               // activation planes ps .. 0
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[rb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hn_bf16x8, ring.v[sl % RS][j]),
                                                                   X[gq & (NX - 1)][rb].p[m], acc[rb][j], 0, 0, 0);
        if (DB && ps == 1 && has_next) {                   // (VALU work beside the matrix pipe's)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) split8(lo[rb], hi[rb], X[(gq + 1) & (NX - 1)][rb]);
        }
      }
    }
  };
#pragma unroll 1
  for (int g0 = 0; g0 < GT - CH; g0 += CH) chunk(g0, false);
  chunk(GT - CH, true);
}

// Accumulators that START at the bias (acc[rb][j] = bias of the block's channels, every row): the bias loads are then
// issued in front of the product, where nothing is queued in front of them -- an epilogue's bias load sits behind the
// previous block's stores, and vmcnt retires in order -- and the epilogue has no add left.  bias[j][g]: the four channel
// runs of block j of this lane (see run4).
template <int RB, int NJ>
__device__ __forceinline__ void bias_acc(f32x16 (&acc)[RB][NJ], const f32x4 (&bias)[NJ][4]) {
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        acc[rb][j][4 * g] = bias[j][g][0]; acc[rb][j][4 * g + 1] = bias[j][g][1];
        acc[rb][j][4 * g + 2] = bias[j][g][2]; acc[rb][j][4 * g + 3] = bias[j][g][3];
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
#ifdef HN_KO_LOADS     // diagnostic build: no epilogue / staging loads (results wrong, time meaningful)
  return 1.0f;
#endif
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off * 4, 0, 0));
}
__device__ __forceinline__ void bst(rsrc_t r, int off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off * 4, 0, 0);
}
#ifdef HN_KO_STORES    // diagnostic build: a store only when the value is NaN-patterned (never), so the work stays live
#define HN_STORE_IF(v) if (__builtin_expect((v)[0] == 1.2345e-30f, 0))
#else
#define HN_STORE_IF(v)
#endif
__device__ __forceinline__ f32x4 bld4(rsrc_t r, int off) {
#ifdef HN_KO_LOADS
  return (f32x4){1.f, 0.5f, 0.25f, 2.f};
#endif
  const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, off * 4, 0, 0);
  return __builtin_bit_cast(f32x4, u);
}
// non-temporal load: saved tensors a backward kernel reads exactly once
__device__ __forceinline__ f32x4 bld4_nt(rsrc_t r, int off) {
#ifdef HN_KO_LOADS
  return (f32x4){1.f, 0.5f, 0.25f, 2.f};
#endif
  const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, off * 4, 0, HN_NT_SAVED_LOADS ? 2 : 0);
  return __builtin_bit_cast(f32x4, u);
}
__device__ __forceinline__ void bst4(rsrc_t r, int off, f32x4 v) {
  HN_STORE_IF(v)
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off * 4, 0, 0);
}
// non-temporal store (aux bit 1 = nt on gfx940+): for the tensors a forward kernel only SAVES for the backward (nobody reads
// them for several milliseconds: keeping their lines would evict what the next kernels do read)
// Round 6, measured (profiles/r06_nt_saved_ab.log, interleaved in one job): configs[1] 2.819 -> 2.808 ms/step with the update
// forward's save-only stores non-temporal; inside a step these kernels run like after a 512 MB cache flush (update forward 37 us
// on hot buffers, 50 in the step and behind a flush: profiles/r06_chain_thrash.log), i.e. their inputs come from HBM.
#ifndef HN_NT_SAVED
#define HN_NT_SAVED 1
#endif
__device__ __forceinline__ void bst4_nt(rsrc_t r, int off, f32x4 v) {
  HN_STORE_IF(v)
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off * 4, 0, HN_NT_SAVED ? 2 : 0);
}

// Coalesced store of one 32 x 32 accumulator block through a wave-private LDS transpose.  A lane owns a ROW of the
// block: stored straight from the registers, one instruction would write 64 16-byte pieces into 32 different rows
// (measured: ~750 cycles per instruction, 5 B/clk/CU -- the epilogues took as long as the products).  Instead the four
// runs go to this wave's scratch [32][36] (ds_write_b128, lane = row) and come back as 4 x (8 rows x 128 contiguous
// bytes): the same number of store instructions, each writing whole 128-byte lines.  LDS operations of one wave execute
// in order, so no barrier or wait is needed between the two halves.  `off_block`: float offset of the block's first
// row / first channel in the destination, LDG its row stride.
constexpr int kScrLd = 36, kScrFloats = 32 * kScrLd;
template <int LDG, bool NT = false>
__device__ __forceinline__ void store_block(float* scr, int lane, const f32x4 (&v)[4], rsrc_t r, int off_block) {
  const int m = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(scr + m * kScrLd + 8 * g + 4 * h) = v[g];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int rr = it * 8 + (lane >> 3), c = (lane & 7) * 4;
    if (NT) bst4_nt(r, off_block + rr * LDG + c, *reinterpret_cast<const f32x4*>(scr + rr * kScrLd + c));
    else bst4(r, off_block + rr * LDG + c, *reinterpret_cast<const f32x4*>(scr + rr * kScrLd + c));
  }
}

// The same for loads, in two steps so that the memory round trip hides behind a product: `issue` requests the block as
// 4 x (8 rows x 128 contiguous bytes) into registers, `finish` (any time later) passes it through the scratch and
// returns the four runs of this lane's row.
struct BlockLoad { f32x4 v[4]; };
template <int LDG, bool NT = false>
__device__ __forceinline__ void issue_block(BlockLoad& b, int lane, rsrc_t r, int off_block) {
#pragma unroll
  for (int it = 0; it < 4; ++it)
    b.v[it] = NT ? bld4_nt(r, off_block + (it * 8 + (lane >> 3)) * LDG + (lane & 7) * 4)
                 : bld4(r, off_block + (it * 8 + (lane >> 3)) * LDG + (lane & 7) * 4);
}
__device__ __forceinline__ void finish_block(float* scr, int lane, const BlockLoad& b, f32x4 (&out)[4]) {
#pragma unroll
  for (int it = 0; it < 4; ++it)
    *reinterpret_cast<f32x4*>(scr + (it * 8 + (lane >> 3)) * kScrLd + (lane & 7) * 4) = b.v[it];
#pragma unroll
  for (int g = 0; g < 4; ++g)
    out[g] = *reinterpret_cast<const f32x4*>(scr + (lane & 31) * kScrLd + 8 * g + 4 * (lane >> 5));
}

// Accumulator block (rb, cb) of D^T = W . A^T: lane l holds tile row  (first row of the wave) + 32 rb + (l & 31)  and the
// channels  32 cb + 8 g + 4 (l >> 5) + e  in register 4 g + e  (g, e < 4): run g of the block as one float4.
__device__ __forceinline__ f32x4 run4(const f32x16& acc, int g) {
  return (f32x4){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
}
__device__ __forceinline__ void set_run4(f32x16& acc, int g, f32x4 v) {
  acc[4 * g] = v[0]; acc[4 * g + 1] = v[1]; acc[4 * g + 2] = v[2]; acc[4 * g + 3] = v[3];
}
__device__ __forceinline__ f32x4 ssilu4(f32x4 v) { return (f32x4){ssilu(v[0]), ssilu(v[1]), ssilu(v[2]), ssilu(v[3])}; }
__device__ __forceinline__ f32x4 dssilu4(f32x4 v) { return (f32x4){dssilu(v[0]), dssilu(v[1]), dssilu(v[2]), dssilu(v[3])}; }
__device__ __forceinline__ f32x4 ld4g(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void pin4(f32x4& v) { asm volatile("" : "+v"(v)); }

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


// Row windows (atom shards, sharding.py: the halo rows close every relation's row block): a launch with wmode 1 runs
// only the tiles that touch a window, wmode 2 only the others -- the rows that do not wait for the halo exchange are
// projected while it is in flight, the rest after it (backward: the halo tiles first, so that their gradients travel
// while the others are computed).  The two launches partition the tiles; wmode 0 runs all of them.
__device__ __forceinline__ bool tile_selected(const int* __restrict__ win, int nwin, int wmode, int row0, int TR) {
  if (wmode == 0) return true;
  bool inside = false;
  for (int k = 0; k < nwin; ++k) {
    const int lo = win[2 * k], hi = win[2 * k + 1];
    inside |= hi > lo && row0 < hi && row0 + TR > lo;
  }
  return inside == (wmode == 1);
}

// HTNet: relation (c; p, q) gathers source rows of elements p and q only -- tiles outside both row ranges are skipped
__device__ __forceinline__ bool tile_wanted(const int* __restrict__ ranges, int t, int row0, int TR) {
  if (ranges == nullptr) return true;
  const int r0 = ranges[4 * t], r1 = ranges[4 * t + 1], r2 = ranges[4 * t + 2], r3 = ranges[4 * t + 3];
  return (row0 < r1 && row0 + TR > r0) || (row0 < r3 && row0 + TR > r2);
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

// The two elementwise halves of the LayerNorm backward, shared by every kernel that runs it (they must agree bit for bit:
// the two products of ln_out are made opaque -- pin -- so that no context contracts one of them with the final sum).
__device__ __forceinline__ void ln_prep(const f32x4& gg, const f32x4& xv, float mu, float rs, int c, int Hr, f32x4& gv,
                                        f32x4& nh, float& s1, float& s2) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float m = c + e < Hr ? 1.f : 0.f;
    gv[e] = gg[e] * m;
    nh[e] = (xv[e] - mu) * rs * m;
    s1 += gv[e];
    s2 = fmaf(gv[e], nh[e], s2);
  }
}
__device__ __forceinline__ f32x4 ln_out(const f32x4& gv, const f32x4& nh, float m1, float m2, float rs, int c, int Hr,
                                        bool has_add, const f32x4& av, float add_scale) {
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = rs * fmaf(-nh[e], m2, gv[e] - m1);
    pin(v);
    o[e] = c + e < Hr ? v : 0.f;
  }
  if (has_add) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float b = av[e] * add_scale;
      pin(b);
      o[e] += b;
    }
  }
  return o;
}

// One row of gx = LayerNorm'(x)^T (sum_p g[p]) + add_scale * add, a wave per row (lane l: channels 4 (64 k + l) .. + 3).
__device__ __forceinline__ void layernorm_bwd_row(const float* __restrict__ g, int nparts, long part_stride,
                                                  const float* __restrict__ x, float mu, float rs,
                                                  const float* __restrict__ add, float add_scale, float* __restrict__ gx,
                                                  size_t r, int H, int Hr, int lane) {
  constexpr int KMAX = 4;                        // H <= 1024
  f32x4 gv[KMAX], nh[KMAX];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int c = (k * 64 + lane) * 4;
    gv[k] = nh[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c < H) {
      f32x4 gg = *reinterpret_cast<const f32x4*>(g + r * H + c);
      for (int p = 1; p < nparts; ++p) gg += *reinterpret_cast<const f32x4*>(g + p * part_stride + r * H + c);
      ln_prep(gg, *reinterpret_cast<const f32x4*>(x + r * H + c), mu, rs, c, Hr, gv[k], nh[k], s1, s2);
    }
  }
  const float m1 = wave_sum(s1) / (float)Hr, m2 = wave_sum(s2) / (float)Hr;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < H) {
      f32x4 av = {0.f, 0.f, 0.f, 0.f};
      if (add != nullptr) av = *reinterpret_cast<const f32x4*>(add + r * H + c);
      *reinterpret_cast<f32x4*>(gx + r * H + c) = ln_out(gv[k], nh[k], m1, m2, rs, c, Hr, add != nullptr, av, add_scale);
    }
  }
}

// gx = LayerNorm'(x)^T (sum_p g[p]) + add : the backward of the LayerNorm in front of the T relations' projections
__global__ __launch_bounds__(256) void layernorm_bwd_parts_kernel(const float* __restrict__ g, int nparts, long part_stride,
                                                                  const float* __restrict__ x, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const float* __restrict__ add,
                                                                  float* __restrict__ gx, int rows, int H, int Hr,
                                                                  const int* __restrict__ windows, int nwin, int wmode,
                                                                  int TR) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  if (!tile_selected(windows, nwin, wmode, r / TR * TR, TR)) return;     // the rows of the pre kernel's tiles
  layernorm_bwd_row(g, nparts, part_stride, x, mean[r], rstd[r], add, 1.f, gx, (size_t)r, H, Hr, threadIdx.x & 63);
}

// The prologue of the update backward when its incoming gradients are still `pending` (UpdBwdArgs::pend): what
// message_bwd_finish_kernel and layernorm_bwd_parts_kernel would have written for the rows [row0, row0 + nrows) of this
// workgroup's tile, in the same order of operations (bit-identical), into gxo / gvo; the barrier at the end makes the rows
// visible to the whole workgroup, which then reads them like any other input.  Rows >= identity_rows (atoms of an unknown
// element) have no residual term.
// TILE (fused form, H = 128): the sum over the relations of gn for the rows of this tile already sits in `gn_tile` ([TR][ld_tile]
// in LDS, written by the chain that ran in this launch) instead of p.gn's partial sums in memory.
template <int H, int TR, bool TILE = false>
__device__ __forceinline__ void materialise_pending(const UpdBwdArgs& a, int row0, int nrows, int tid,
                                                    const float* gn_tile = nullptr, int ld_tile = 0) {
  static_assert(!TILE || H == 128, "the fused form exists at width 128");
  const PendingGrads& p = a.pend;
  const int identity_rows = a.type_rowptr[a.T];
  float* gxo = const_cast<float*>(a.gxo);
  const int lane = tid & 63, wave = tid >> 6;
  // ---- gvec_out: every load of a relation's slice is issued before the first one is used (one round trip per relation)
  constexpr int V3 = 3 * H / 4, NIT = (TR * V3 + 255) / 256;
  const long stride4 = (long)a.N * V3;
  const f32x4* part = reinterpret_cast<const f32x4*>(p.gv) + (long)row0 * V3;
  const f32x4* g1 = reinterpret_cast<const f32x4*>(p.gvec1) + (long)row0 * V3;
  f32x4* gvo = reinterpret_cast<f32x4*>(const_cast<float*>(a.gvo)) + (long)row0 * V3;
  const int nv = nrows * V3;
  f32x4 acc[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = tid + it * 256;
    acc[it] = part[idx < nv ? idx : 0];
  }
  // ---- gx_out (requested behind the first slice of gvec, finished while the others arrive)
  if (H == 128) {
    // half a wave per row: lanes 0-31 / 32-63 hold one row each (channels 4 (l & 31) .. + 3), TR / 8 passes; the sums over
    // a row are the xor tree of wave_sum without its first step, whose partner is zero at this width: the same bits
    constexpr int RP = TR / 8;
    const int c = (lane & 31) * 4;
    const long ps = (long)a.N * H;
    f32x4 gg[RP], xv[RP], av[RP];
    float mu[RP], rs[RP];
    int rr[RP];
#pragma unroll
    for (int it = 0; it < RP; ++it) {
      const int lr = it * 8 + wave * 2 + (lane >> 5);
      rr[it] = lr < nrows ? row0 + lr : -1;
      const size_t r = (size_t)(rr[it] >= 0 ? rr[it] : row0);
      if (TILE) gg[it] = *reinterpret_cast<const f32x4*>(gn_tile + (rr[it] >= 0 ? lr : 0) * ld_tile + c);
      else gg[it] = *reinterpret_cast<const f32x4*>(p.gn + r * H + c);
      xv[it] = *reinterpret_cast<const f32x4*>(p.x + r * H + c);
      av[it] = *reinterpret_cast<const f32x4*>(p.gx1 + r * H + c);
      mu[it] = p.mean[r];
      rs[it] = p.rstd[r];
    }
    if (!TILE) {
      for (int t = 1; t < p.nparts; ++t) {
#pragma unroll
        for (int it = 0; it < RP; ++it) {
          const size_t r = (size_t)(rr[it] >= 0 ? rr[it] : row0);
          gg[it] += *reinterpret_cast<const f32x4*>(p.gn + t * ps + r * H + c);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < RP; ++it) {
      f32x4 gv, nh;
      float s1 = 0.f, s2 = 0.f;
      ln_prep(gg[it], xv[it], mu[it], rs[it], c, p.Hr, gv, nh, s1, s2);
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
      }
      const float m1 = s1 / (float)p.Hr, m2 = s2 / (float)p.Hr;
      const f32x4 o = ln_out(gv, nh, m1, m2, rs[it], c, p.Hr, rr[it] >= 0 && rr[it] < identity_rows, av[it], kInvSqrt2);
      if (rr[it] >= 0) *reinterpret_cast<f32x4*>(gxo + (size_t)rr[it] * H + c) = o;
    }
  } else {
    for (int lr = wave; lr < nrows; lr += 4) {
      const int r = row0 + lr;
      layernorm_bwd_row(p.gn, p.nparts, (long)a.N * H, p.x, p.mean[r], p.rstd[r], r < identity_rows ? p.gx1 : nullptr,
                        kInvSqrt2, gxo, (size_t)r, H, p.Hr, lane);
    }
  }
  for (int t = 1; t < p.nparts; ++t) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256;
      acc[it] += part[t * stride4 + (idx < nv ? idx : 0)];
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = tid + it * 256;
    if (idx < nv) {
      if (row0 + idx / V3 < identity_rows) acc[it] += g1[idx];
      gvo[idx] = acc[it];
    }
  }
  __syncthreads();
}

}  // namespace
