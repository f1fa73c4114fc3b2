// gfx950: the PaiNNUpdate chain (node_update_fwd / node_update_bwd of node_chain.hip; rmnet.py:94-107, 29-31) on 16-ROW
// tiles -- for grids that 32-row tiles quantise badly.  (Written for v_mfma_f32_16x16x4_f32; since the second half of round 5
// the products run as three-way bf16 splits on v_mfma_f32_16x16x32_bf16 at fp32 accuracy: see mma16_panel.)
//
// Why: BASELINE configs[1] has 10,041 target rows = 314 tiles of 32 rows on 256 CUs: 58 CUs get two workgroups, 198 get
// one, the launch lasts as long as the CUs with two (a makespan of 64 rows against 39 rows per CU on average: 0.61).  In
// 16-row tiles the same rows are 628 workgroups, three or two per CU: 48 rows against 39 (0.82), and a CU's two or three
// co-resident workgroups fill each other's epilogue phases.  The 16 x 16 x 4 form has the same FLOP rate as 32 x 32 x 2
// (64 FLOP/clk/SIMD); what it costs is weight traffic (every fragment serves 16 rows instead of 32: twice the bytes
// from L2), which is why the 32-row kernels stay in charge of large grids (node_chain.hip picks per launch).
//
// Data flow as in node_chain.hip: weights streamed from L2 in MFMA operand order -- here frag16(W)[(b * K/16 + Q) * 64 + l]
// = float4 W[16 b + (l & 15)][16 Q + 4 (l >> 4) .. +3], one coalesced 1 KiB load per 16-channel block and 16-deep k-group --,
// activations in [16][H + 8] LDS tiles (ds_read_b128 at row l & 15, k = 16 Q + 4 (l >> 4): conflict-free with the +8 pad),
// accumulators transposed: lane l holds tile row l & 15 and the channels 16 b + 4 (l >> 4) .. +3 of block b as ONE float4.
// A wave owns 32 channels (two blocks) of every output part, so vec_dot, q * vdot and r * v1 stay lane-local.
//
// Round 5: ONE node launch per layer boundary, each way.  A tile's update (layer l) and the node projection of the rows it has
// just produced (layer l + 1: LayerNorm -> [H -> H] -> ScaledSiLU -> [H -> 3H] for every relation; rmnet.py:52 behind
// rmnet.py:29-31, 94-107) are row-local, so `node_update_pre_fwd16_kernel` runs both: x_out never leaves the chip between
// them.  The mirror `node_pre_update_bwd16_kernel` runs the projection's backward of layer l + 1 in front of the update
// backward of layer l: the per-relation sums stay in registers (no [T, N, H] partial sums in memory), the LayerNorm backward
// runs on the tile.  The same phases also exist as kernels of their own (`node_pre_fwd16_kernel`, `node_pre_bwd16_kernel`):
// the unfused form is the A/B and the bit-for-bit check of the fused one.
#include "node_chain_common.h"

// Workgroups per CU the register budget is cut for (launch bounds).  Round 5: every kernel here FITS three (<= 168 registers,
// no spills) -- what it took: the vec_dot sums pinned where they are formed (hipcc had sunk them behind the last product and
// kept v1 / v2 of all three components alive, spilled), v1 re-read from vp instead of held in registers, a weight ring of two
// for four-block products.  Measured (tools/chain_bench.py, 10,041 rows, same box, alternating builds, profiles/r05_chain_bench1.log):
// the backward kernels gain a little from the third workgroup (fused 100.0 vs 101.8 us), the forward kernels LOSE (fused 102.0
// vs 97.2 us) -- with three resident the products of each run slower and the launch ends no earlier: the matrix pipe is not
// what a third workgroup's waves were waiting for.  So: forward 2, backward 3.
#ifndef HN_U16_MINW_FWD
#define HN_U16_MINW_FWD 2
#endif
#ifndef HN_U16_MINW_BWD
#define HN_U16_MINW_BWD 3
#endif

namespace {

constexpr int kTR16 = 16;
constexpr int kScr16Ld = 36, kScr16Floats = 16 * kScr16Ld;     // per-wave transpose scratch [16][36]

// k-groups of weight fragments in flight per block: a group is 4 NB MFMAs of 32 cycles, an L2 hit 500-800 cycles
constexpr int ring16(int nb) { return nb >= 4 ? 2 : 4; }     // (64 registers for a ring of four at NB = 4 kept a third workgroup off the CU)

template <int NB>
struct Ring16 { f32x4 v[ring16(NB)][NB]; };

// Diagnostic builds (results wrong, time meaningful): -DHN_KO_BSTREAM=1 every weight request reads the stream's FIRST fragment
// (same instructions, every request an L1 hit: what the weight stream's bytes cost), =2 no weight request at all (what its
// instructions cost).  Round 5, 10,041 rows (profiles/r05_chain_ko.log): fused forward 98.4 / 96.2 / 75.2 us, fused backward
// 99.9 / 94.8 / 74.3 us for 0 / 1 / 2 -- the REQUESTS (one 1-KiB load per four MFMAs at 16 rows), not their bytes or their
// latency, are a quarter of these kernels' time.  Halving them with two row blocks per fragment (mixed 32- / 16-row tiles so
// that the busiest CU still holds three blocks) was prototyped on the projection kernel: bit-identical, 64.0 -> 59.6 us -- the
// 64-row kernel of node_chain.hip does the same work in 48 us; not pursued.
#ifndef HN_KO_BSTREAM
#define HN_KO_BSTREAM 0
#endif
__device__ __forceinline__ f32x4 wfrag(const f32x4* p, int idx) {
#if HN_KO_BSTREAM == 1
  return p[idx & 0];
#elif HN_KO_BSTREAM == 2
  return (f32x4){0.01f, -0.02f, 0.03f, 0.015f};
#else
  return p[idx];
#endif
}

template <int NB>
__device__ __forceinline__ void b16_preload(Ring16<NB>& r, const f32x4* const (&bp)[NB]) {
#pragma unroll
  for (int g = 0; g < ring16(NB) - 1; ++g)
#pragma unroll
    for (int j = 0; j < NB; ++j) r.v[g][j] = wfrag(bp[j], g * 64);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5 (second half): the products run on the BF16 matrix pipe at fp32 accuracy -- three-way bf16 splits, six partial
// products per k-group on v_mfma_f32_16x16x32_bf16 (node_chain_common.h: split8, "fp32 products on the BF16 matrix pipe").
// Measured before it was built, with the instruction mix alone (a diagnostic build, results wrong, time meaningful:
// profiles/r05_split_timing.log): update_fwd 55 -> 32 us, pre_fwd16 64 -> 30, fused forward 98 -> 61.
//
// Stream: a block's fragments are consumed as STEPS of one 1-KiB load each (lane l: 16 bytes = 8 bf16 of one weight plane,
// W_p[16 b + (l & 15)][32 Q + 8 (l >> 4) .. +7]), in the order [k-group Q of 32][plane 2, 1, 0]: three steps per 32 k where
// the fp32 form took two 16-deep groups -- 1.5 x the requests and bytes.  frag16(W)[((b * K/32 + Q) * 3 + s) * 64 + l].
// ---------------------------------------------------------------------------------------------------------------------
// float4s of one 16-channel block's fragment stream over K inputs (3 K / 32 steps of 64 lanes), floats of an [O, K] weight
constexpr int frag16_f4(int K) { return K * 6; }
constexpr size_t frag16_floats(size_t O, size_t K) { return O * K * 3 / 2; }

// acc[j] += W_j[:, 0 .. KP) . A^T; `As`: this lane's LDS pointer &tile[(l & 15) * LD + 4 (l >> 4)] (the tile is fp32).
// The ring holds steps 0 .. RS-2 on entry; MORE: the stream continues behind the panel (its first RS-1 steps are
// requested and sit in slots 0 .. RS-2 on exit).
// RB row blocks share every weight fragment: acc[rb][j] += W_j . A_rb^T for RB activation tiles (`As[rb]`: this lane's pointer
// into tile rb).  A fragment then feeds RB x (1 .. 3) MFMAs -- the weight stream, which is what a 16-row tile's time is made of
// (see the knock-outs above), is paid once for RB tiles.  RB > 1 keeps ONE set of planes per tile (split in front of the
// group's MFMAs); RB = 1 double-buffers them (the split runs beside the previous group's MFMAs).
template <int KP, int NB, int RB, bool MORE>
__device__ __forceinline__ void mma16_panel_rb(f32x4 (&acc)[RB][NB], const float* const (&As)[RB], const f32x4* const (&bp)[NB],
                                               Ring16<NB>& ring) {
  constexpr int GT = KP / 32, NSTEP = 3 * GT, RS = ring16(NB), PF = RS - 1, NX = RB == 1 ? 2 : 1;
  static_assert(KP % 32 == 0 && (!MORE || NSTEP % RS == 0), "panel / ring mismatch");
  const int g4 = 4 * ((threadIdx.x & 63) >> 4);                 // As + g4 = &tile[(l & 15) * LD + 8 (l >> 4)]
  f32x4 lo[RB], hi[RB];
  Split8 X[NX][RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    lo[rb] = *reinterpret_cast<const f32x4*>(As[rb] + g4);
    hi[rb] = *reinterpret_cast<const f32x4*>(As[rb] + g4 + 4);
    if (NX == 2) split8(lo[rb], hi[rb], X[0][rb]);
  }
#pragma unroll
  for (int Q = 0; Q < GT; ++Q) {
    if (NX == 1) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) split8(lo[rb], hi[rb], X[0][rb]);
    }
    if (Q + 1 < GT) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        lo[rb] = *reinterpret_cast<const f32x4*>(As[rb] + g4 + 32 * (Q + 1));
        hi[rb] = *reinterpret_cast<const f32x4*>(As[rb] + g4 + 32 * (Q + 1) + 4);
      }
    }
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {                 // the step of weight plane 2 - ps
      const int n = 3 * Q + ps;
      if (MORE || n + PF < NSTEP) {
#pragma unroll
        for (int j = 0; j < NB; ++j) ring.v[(n + PF) % RS][j] = wfrag(bp[j], (n + PF) * 64);
      }
      if (HN_PIN_LOADS) fence_sched();               // (the requests stay in front of the step's MFMAs: node_chain_common.h)
#pragma unroll
      for (int m = ps; m >= 0; --m)                  // activation planes ps .. 0: the smaller partial products first
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int j = 0; j < NB; ++j)
            acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hn_bf16x8, ring.v[n % RS][j]),
                                                                 X[Q & (NX - 1)][rb].p[m], acc[rb][j], 0, 0, 0);
      if (NX == 2 && ps == 1 && Q + 1 < GT) {        // (VALU work beside the matrix pipe's)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) split8(lo[rb], hi[rb], X[(Q + 1) & 1][rb]);
      }
    }
  }
}

template <int KP, int NB, bool MORE>
__device__ __forceinline__ void mma16_panel(f32x4 (&acc)[NB], const float* As, const f32x4* const (&bp)[NB], Ring16<NB>& ring) {
  const float* const as1[1] = {As};
  mma16_panel_rb<KP, NB, 1, MORE>(reinterpret_cast<f32x4 (&)[1][NB]>(acc), as1, bp, ring);
}

// Coalesced store / load of this wave's [16 rows][32 channels] block pair through the wave-private scratch: lane l owns
// row l & 15 and the channels 16 s + 4 (l >> 4) .. +3 of sub-block s (v[s]); memory sees 2 x (8 rows x 128 bytes).
template <int LDG, bool NT = false>
__device__ __forceinline__ void store16(float* scr, int lane, const f32x4 (&v)[2], rsrc_t r, int off) {
  const int m = lane & 15, g = lane >> 4;
#pragma unroll
  for (int s = 0; s < 2; ++s) *reinterpret_cast<f32x4*>(scr + m * kScr16Ld + 16 * s + 4 * g) = v[s];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int rr = it * 8 + (lane >> 3), c = (lane & 7) * 4;
    if (NT) bst4_nt(r, off + rr * LDG + c, *reinterpret_cast<const f32x4*>(scr + rr * kScr16Ld + c));
    else bst4(r, off + rr * LDG + c, *reinterpret_cast<const f32x4*>(scr + rr * kScr16Ld + c));
  }
}
struct Load16 { f32x4 v[2]; };
template <int LDG, bool NT = false>
__device__ __forceinline__ void issue16(Load16& b, int lane, rsrc_t r, int off) {
#pragma unroll
  for (int it = 0; it < 2; ++it)
    b.v[it] = NT ? bld4_nt(r, off + (it * 8 + (lane >> 3)) * LDG + (lane & 7) * 4)
                 : bld4(r, off + (it * 8 + (lane >> 3)) * LDG + (lane & 7) * 4);
}
__device__ __forceinline__ void finish16(float* scr, int lane, const Load16& b, f32x4 (&out)[2]) {
#pragma unroll
  for (int it = 0; it < 2; ++it)
    *reinterpret_cast<f32x4*>(scr + (it * 8 + (lane >> 3)) * kScr16Ld + (lane & 7) * 4) = b.v[it];
#pragma unroll
  for (int s = 0; s < 2; ++s)
    out[s] = *reinterpret_cast<const f32x4*>(scr + (lane & 15) * kScr16Ld + 16 * s + 4 * (lane >> 4));
}

__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 sqrt4(f32x4 v) { return (f32x4){sqrtf(v[0]), sqrtf(v[1]), sqrtf(v[2]), sqrtf(v[3])}; }
__device__ __forceinline__ f32x4 rcp_clamped4(f32x4 n) {
  return (f32x4){__builtin_amdgcn_rcpf(fmaxf(n[0], 1e-4f)), __builtin_amdgcn_rcpf(fmaxf(n[1], 1e-4f)),
                 __builtin_amdgcn_rcpf(fmaxf(n[2], 1e-4f)), __builtin_amdgcn_rcpf(fmaxf(n[3], 1e-4f))};
}

// cooperative [16][H] tile copy through registers (H / 64 float4 per thread)
template <int H>
struct Tile16Regs { f32x4 v[kTR16 * H / 4 / 256]; };
template <int H>
__device__ __forceinline__ void tile16_load(Tile16Regs<H>& r, rsrc_t src, int ld_src, int off0, int tid) {
  constexpr int V = H / 4, F4 = kTR16 * H / 4 / 256;
#pragma unroll
  for (int it = 0; it < F4; ++it) {
    const int idx = tid + it * 256, row = idx / V, c4 = idx % V;
    r.v[it] = bld4(src, row * ld_src + off0 + c4 * 4);
  }
}
template <int H, int LD>
__device__ __forceinline__ void tile16_store(float* tile, const Tile16Regs<H>& r, int tid) {
  constexpr int V = H / 4, F4 = kTR16 * H / 4 / 256;
#pragma unroll
  for (int it = 0; it < F4; ++it) {
    const int idx = tid + it * 256, row = idx / V, c4 = idx % V;
    *reinterpret_cast<f32x4*>(tile + row * LD + c4 * 4) = r.v[it];
  }
}

// =====================================================================================================================
// The node projection of one 16-row tile (rmnet.py:52 for every relation): the tile's LayerNorm input sits in `bufN`.
// =====================================================================================================================
// LayerNorm without affine of a [16][H] LDS tile, in place; statistics over the first Hr channels.  Sixteen adjacent lanes share
// a row, its float4s dealt round-robin (a wave normalises four rows: the four waves work side by side -- one wave doing the
// whole tile with four lanes per row took 7.9k cycles of a workgroup's life in which the other three waited at the barrier).
// The caller puts a barrier in front (tile written) and behind (tile normalised).
template <int H>
__device__ __forceinline__ void layernorm_tile16(float* tile, int tid, int nrows, int Hr, float eps, float* mean0, float* rstd0) {
  constexpr int LD = H + 8, TPR = 16, NF = H / 4 / TPR;
  const int lr = tid / TPR, q = tid % TPR;
  f32x4 v[NF];
#pragma unroll
  for (int k = 0; k < NF; ++k) v[k] = *reinterpret_cast<const f32x4*>(tile + lr * LD + (k * TPR + q) * 4);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NF; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if ((k * TPR + q) * 4 + e >= Hr) v[k][e] = 0.f;
      s += v[k][e];
    }
#pragma unroll
  for (int m = 1; m < TPR; m <<= 1) s += __shfl_xor(s, m, 64);
  const float inv_h = 1.0f / (float)Hr;
  float mu = s * inv_h;
  // corrected two-pass statistics (round 6; see node_chain.hip): the mean of the deviations is the rounding error of `mu`
  float qq = 0.f, dd = 0.f;
#pragma unroll
  for (int k = 0; k < NF; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[k][e] = (k * TPR + q) * 4 + e < Hr ? v[k][e] - mu : 0.f;
      dd += v[k][e];
      qq = fmaf(v[k][e], v[k][e], qq);
    }
#pragma unroll
  for (int m = 1; m < TPR; m <<= 1) {
    qq += __shfl_xor(qq, m, 64);
    dd += __shfl_xor(dd, m, 64);
  }
  const float dm = dd * inv_h;
  mu += dm;
#pragma unroll
  for (int k = 0; k < NF; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((k * TPR + q) * 4 + e < Hr) v[k][e] -= dm;
  float var = qq * inv_h;
  pin(var);                       // (the product stays a product: no contraction with the sum behind it, in any kernel)
  float dm2 = dm * dm;
  pin(dm2);
  var = fmaxf(var - dm2, 0.f);
  const float rs = rsqrtf(var + eps);
#pragma unroll
  for (int k = 0; k < NF; ++k) *reinterpret_cast<f32x4*>(tile + lr * LD + (k * TPR + q) * 4) = v[k] * rs;
  if (q == 0 && lr < nrows) { mean0[lr] = mu; rstd0[lr] = rs; }
}

// xh[t] = x_proj_t(LayerNorm(x)) for every relation t of `p`, rows [row0, row0 + nrows) -- their x in bufN, raw, behind a
// barrier.  bufA0 / bufA1: two more [16][H + 8] tiles (the hidden activations of even / odd relations: one barrier per
// relation).  Saves hb (pre-activations incl. bias), mean, rstd like node_pre_fwd_kernel.  Per wave and relation 64 + 192 MFMAs.
template <int H>
__device__ __forceinline__ void pre_fwd16_phase(const PreFwdArgs& p, float* bufN, float* bufA0, float* bufA1, float* scr,
                                                int row0, int nrows, int tid, int lane, int wave) {
  constexpr int LD = H + 8, NB16 = H / 16;
  const int mrow = lane & 15, ch = 4 * (lane >> 4), cw = 32 * wave;
  const f32x4* bp1[2];
  Ring16<2> r1;
  f32x4 acc1[2];
  // the first product's operands of relation t: requested in front of whatever stores precede the product (vmcnt is in order)
  auto request1 = [&](int t) {
    const f32x4* w1 = reinterpret_cast<const f32x4*>(p.w1f + (size_t)t * frag16_floats(H, H)) + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) bp1[s] = w1 + (size_t)(2 * wave + s) * frag16_f4(H);
    b16_preload(r1, bp1);
#pragma unroll
    for (int s = 0; s < 2; ++s) acc1[s] = ld4g(p.b1 + (size_t)t * H + cw + 16 * s + ch);
  };
  request1(0);
  layernorm_tile16<H>(bufN, tid, nrows, p.Hr, p.eps, p.mean + row0, p.rstd + row0);
  __syncthreads();
  STAMP(10);
  const float* An = bufN + mrow * LD + ch;
#pragma unroll 1
  for (int t = 0; t < p.T; ++t) {
    float* bufA = (t & 1) ? bufA1 : bufA0;
    const rsrc_t hb_r = tile_rsrc(p.hb + ((size_t)t * p.Ns + row0) * H, nrows * H);
    const rsrc_t xh_r = tile_rsrc(p.xh + ((size_t)t * p.Ns + row0) * 3 * H, nrows * 3 * H);
    // ---- h = n W1^T + b1
    mma16_panel<H, 2, false>(acc1, An, bp1, r1);
    const f32x4* w2 = reinterpret_cast<const f32x4*>(p.w2f + (size_t)t * frag16_floats(3 * H, H)) + lane;
    const f32x4* bp2[6];
#pragma unroll
    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
      for (int s = 0; s < 2; ++s) bp2[2 * pp + s] = w2 + (size_t)(pp * NB16 + 2 * wave + s) * frag16_f4(H);
    Ring16<6> r2;
    b16_preload(r2, bp2);
    f32x4 acc2[6];
#pragma unroll
    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
      for (int s = 0; s < 2; ++s) acc2[2 * pp + s] = ld4g(p.b2 + (size_t)t * 3 * H + pp * H + cw + 16 * s + ch);
    fence_sched();
    {
      f32x4 hv[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        hv[s] = acc1[s];                               // (incl. b1)
        *reinterpret_cast<f32x4*>(bufA + mrow * LD + cw + 16 * s + ch) = ssilu4(hv[s]);
      }
      store16<H, true>(scr, lane, hv, hb_r, cw);
    }
    __syncthreads();
    // ---- xh = a W2^T + b2
    mma16_panel<H, 6, false>(acc2, bufA + mrow * LD + ch, bp2, r2);
    if (t + 1 < p.T) request1(t + 1);
    fence_sched();
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
      const f32x4 v[2] = {acc2[2 * pp], acc2[2 * pp + 1]};
      store16<3 * H>(scr, lane, v, xh_r, pp * H + cw);
    }
    STAMP(11 + (t < 3 ? t : 3));
  }
}

// The projection as a kernel of its own (the unfused form: A/B and bit-for-bit check of the fused kernel's second half)
template <int H>
__global__ __launch_bounds__(256, HN_U16_MINW_FWD) void node_pre_fwd16_kernel(PreFwdArgs p) {
  constexpr int TR = kTR16, LD = H + 8;
  extern __shared__ __align__(16) float lds[];               // 3 x [TR][LD], then 4 x [16][36] scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * TR, nrows = min(TR, p.Ns - row0);
  const rsrc_t x_r = tile_rsrc(p.x + (size_t)row0 * H, nrows * H);
  Tile16Regs<H> regs;
  tile16_load<H>(regs, x_r, H, 0, tid);
  tile16_store<H, LD>(lds + 2 * TR * LD, regs, tid);
  __syncthreads();
  pre_fwd16_phase<H>(p, lds + 2 * TR * LD, lds, lds + TR * LD, lds + 3 * TR * LD + wave * kScr16Floats, row0, nrows, tid, lane, wave);
}

// =====================================================================================================================
// node_update_fwd on 16-row tiles (H = 128: four waves x 32 channels); FUSE: + the next layer's node projection of the tile
// =====================================================================================================================
template <int H, bool FUSE>
__global__ __launch_bounds__(256, HN_U16_MINW_FWD) void node_update_fwd16_kernel(UpdFwdArgs a, PreFwdArgs p) {
  static_assert(H == 128, "four waves x 32 channels");
  constexpr int TR = kTR16, LD = H + 8, NB16 = H / 16;       // NB16: 16-channel blocks per part
  extern __shared__ __align__(16) float lds[];               // 3 x [TR][LD], then 4 x [16][36] scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* scr = lds + 3 * TR * LD + wave * kScr16Floats;
  const TileInfo ti = find_tile(a.type_rowptr, a.T, a.N, TR, blockIdx.x);
  const int row0 = ti.row0, nrows = ti.nrows, t = ti.t;
  if (t >= a.T) {                                 // rows of unknown elements: zero
    constexpr int V = H / 4;
    for (int idx = tid; idx < nrows * V; idx += 256) {
      const int r = row0 + idx / V, c = (idx % V) * 4;
      *reinterpret_cast<f32x4*>(a.x_out + (size_t)r * H + c) = zero4();
#pragma unroll
      for (int d = 0; d < 3; ++d) *reinterpret_cast<f32x4*>(a.vec_out + ((size_t)r * 3 + d) * H + c) = zero4();
    }
    if constexpr (FUSE) {                         // (they are sources all the same: their projection runs on x = 0)
      for (int idx = tid; idx < TR * (H / 4); idx += 256)
        *reinterpret_cast<f32x4*>(lds + 2 * TR * LD + (idx / (H / 4)) * LD + (idx % (H / 4)) * 4) = zero4();
      __syncthreads();
      pre_fwd16_phase<H>(p, lds + 2 * TR * LD, lds, lds + TR * LD, scr, row0, nrows, tid, lane, wave);
    }
    return;
  }
  const f32x4* wv = reinterpret_cast<const f32x4*>(a.wvf + (size_t)t * frag16_floats(2 * H, H)) + lane;
  const f32x4* wx0 = reinterpret_cast<const f32x4*>(a.wx0f + (size_t)t * frag16_floats(H, 2 * H)) + lane;
  const f32x4* wx2 = reinterpret_cast<const f32x4*>(a.wx2f + (size_t)t * frag16_floats(3 * H, H)) + lane;
  // this wave's blocks of part p: b = p * NB16 + 2 wave + s
  const f32x4* bpv[4];
  const f32x4* bpx[2];
  const f32x4* bpq[6];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) bpv[2 * p + s] = wv + (size_t)(p * NB16 + 2 * wave + s) * frag16_f4(H);
#pragma unroll
  for (int s = 0; s < 2; ++s) bpx[s] = wx0 + (size_t)(2 * wave + s) * frag16_f4(2 * H);
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) bpq[2 * p + s] = wx2 + (size_t)(p * NB16 + 2 * wave + s) * frag16_f4(H);

  const int mrow = lane & 15, ch = 4 * (lane >> 4), cw = 32 * wave;
  const rsrc_t x1_r = tile_rsrc(a.x1 + (size_t)row0 * H, nrows * H);
  const rsrc_t vec1_r = tile_rsrc(a.vec1 + (size_t)row0 * 3 * H, nrows * 3 * H);
  const rsrc_t act_r = tile_rsrc(a.row_active ? a.row_active + row0 : a.x1, a.row_active ? nrows : 0);
  const bool all_on = a.row_active == nullptr;
  const rsrc_t vp_r = tile_rsrc(a.vp + (size_t)row0 * 6 * H, nrows * 6 * H);
  const rsrc_t h2b_r = tile_rsrc(a.h2b + (size_t)row0 * H, nrows * H);
  const rsrc_t q23_r = tile_rsrc(a.q23 + (size_t)row0 * 2 * H, nrows * 2 * H);
  const rsrc_t xo_r = tile_rsrc(a.x_out + (size_t)row0 * H, nrows * H);
  const rsrc_t vo_r = tile_rsrc(a.vec_out + (size_t)row0 * 3 * H, nrows * 3 * H);
  const rsrc_t nrm_r = tile_rsrc(a.nrm + (size_t)row0 * H, nrows * H);
  f32x4 dot[2] = {zero4(), zero4()}, sq[2] = {zero4(), zero4()};

  // ---- vp[d] = vec1[d] Wv^T; vec_dot and |v2|^2 accumulate in registers (v1 is re-read from vp for dvec = r v1: round 5 --
  // 24 registers held across the whole kernel were what kept a third workgroup off the CU)
  STAMP(0);
  STAMP_HWID();
  // Round 5, second half: the three components in ONE product (three activation tiles per weight fragment): Wv was streamed
  // once per component -- 144 of the kernel's 264 fragment requests per wave, and the requests are what a tile's time is made of.
  Tile16Regs<H> regs;
  Ring16<4> rv;
  b16_preload(rv, bpv);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    tile16_load<H>(regs, vec1_r, 3 * H, d * H, tid);
    tile16_store<H, LD>(lds + d * TR * LD, regs, tid);
  }
  __syncthreads();
  tile16_load<H>(regs, x1_r, H, 0, tid);               // (in flight during the product)
  // (accumulators START at their bias -- a lane's float4 of a block is exactly the bias float4 of its channels --, loaded
  // in front of the preceding epilogue's stores: vmcnt retires in order, a bias load issued behind stores waits for them)
  f32x4 acch[2], accq[6];
  {
    f32x4 accv[3][4];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int b = 0; b < 4; ++b) accv[d][b] = zero4();
    const float* const asv[3] = {lds + mrow * LD + ch, lds + TR * LD + mrow * LD + ch, lds + 2 * TR * LD + mrow * LD + ch};
    mma16_panel_rb<H, 4, 3, false>(accv, asv, bpv, rv);
#pragma unroll
    for (int s = 0; s < 2; ++s) acch[s] = ld4g(a.bx0 + (size_t)t * H + cw + 16 * s + ch);
    fence_sched();
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      f32x4 v1[2] = {accv[d][0], accv[d][1]}, v2[2] = {accv[d][2], accv[d][3]};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        dot[s] += v1[s] * v2[s];
        sq[s] += v2[s] * v2[s];
        pin4(dot[s]);
        pin4(sq[s]);
      }
      store16<6 * H>(scr, lane, v1, vp_r, d * 2 * H + cw);
      store16<6 * H, true>(scr, lane, v2, vp_r, d * 2 * H + H + cw);
    }
    STAMP(3);
  }
  // ---- xin = [x1 | sqrt(|v2|^2 + 1e-8)]: x1 -> buffer 1, the norm -> buffer 0
  float* bufx = lds + TR * LD;
  float* bufn = lds;
  Ring16<2> rx;
  b16_preload(rx, bpx);
  __syncthreads();                                   // every wave has finished the product over the three vec tiles
  tile16_store<H, LD>(bufx, regs, tid);
  {
    f32x4 nv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      nv[s] = sqrt4(sq[s] + 1e-8f);
      *reinterpret_cast<f32x4*>(bufn + mrow * LD + cw + 16 * s + ch) = nv[s];
    }
    store16<H, true>(scr, lane, nv, nrm_r, cw);
  }
  __syncthreads();
  STAMP(4);
  // ---- h2 = xin Wx0^T + bx0 (K = 2H: two panels)
  {
    const f32x4* bpx1[2] = {bpx[0] + (size_t)frag16_f4(H), bpx[1] + (size_t)frag16_f4(H)};
    mma16_panel<H, 2, true>(acch, bufx + mrow * LD + ch, bpx, rx);
    mma16_panel<H, 2, false>(acch, bufn + mrow * LD + ch, bpx1, rx);
  }
  STAMP(5);
  Ring16<6> rq;
  b16_preload(rq, bpq);
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) accq[2 * p + s] = ld4g(a.bx2 + (size_t)t * 3 * H + p * H + cw + 16 * s + ch);
  __syncthreads();                                   // buffer 0 is free (x1 stays in buffer 1)
  {
    f32x4 hv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c0 = cw + 16 * s + ch;
      hv[s] = acch[s];                               // (incl. bx0)
      *reinterpret_cast<f32x4*>(lds + mrow * LD + c0) = ssilu4(hv[s]);
    }
    store16<H, true>(scr, lane, hv, h2b_r, cw);
  }
  __syncthreads();
  // ---- (p | q | r) = a2 Wx2^T + bx2, then the update and the residual; vec1 is requested before the product
  // vec1 and v1 of component 0, requested before the product; the components behind it one epilogue step ahead.  (v1 was
  // stored by this wave's own lanes: the wait makes the stores of the three vp epilogues final before it is read back.)
  Load16 lvv, lv1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  issue16<3 * H>(lvv, lane, vec1_r, cw);
  issue16<6 * H>(lv1, lane, vp_r, cw);
  const float on = (all_on | (bld(act_r, mrow) != 0.f)) ? 1.f : 0.f;
  fence_sched();
  STAMP(6);
  mma16_panel<H, 6, false>(accq, lds + mrow * LD + ch, bpq, rq);
  fence_sched();
  STAMP(7);
  const float inv_sqrt_h = rsqrtf((float)H);
  f32x4 q[2], r[2], xo[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int c0 = cw + 16 * s + ch;
    const f32x4 p = accq[s];                         // (incl. bx2)
    q[s] = accq[2 + s];
    r[s] = accq[4 + s];
    const f32x4 x1v = *reinterpret_cast<const f32x4*>(bufx + mrow * LD + c0);
    xo[s] = (x1v + (p + q[s] * dot[s] * inv_sqrt_h) * kInvSqrt2) * on;
  }
  store16<2 * H, true>(scr, lane, q, q23_r, cw);
  store16<2 * H, true>(scr, lane, r, q23_r, H + cw);
  store16<H>(scr, lane, xo, xo_r, cw);
  if constexpr (FUSE) {
    // the rows this tile has just produced are the next layer's LayerNorm input: a third tile takes them (buffers 0 / 1 may
    // still be read by slower waves), the barrier behind the vec_out stores publishes it
#pragma unroll
    for (int s = 0; s < 2; ++s) *reinterpret_cast<f32x4*>(lds + 2 * TR * LD + mrow * LD + cw + 16 * s + ch) = xo[s];
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    f32x4 vo[2], v1[2];
    finish16(scr, lane, lvv, vo);
    finish16(scr, lane, lv1, v1);
    if (d < 2) {
      issue16<3 * H>(lvv, lane, vec1_r, (d + 1) * H + cw);
      issue16<6 * H>(lv1, lane, vp_r, (d + 1) * 2 * H + cw);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) vo[s] = (vo[s] + r[s] * v1[s]) * on;
    store16<3 * H>(scr, lane, vo, vo_r, d * H + cw);
  }
  STAMP(8);
  if constexpr (FUSE) {
    __syncthreads();
    STAMP(9);
    pre_fwd16_phase<H>(p, lds + 2 * TR * LD, lds, lds + TR * LD, scr, row0, nrows, tid, lane, wave);
  }
}

// =====================================================================================================================
// The backward of the node projection on one 16-row tile:  gn_t = ((gxh[t] W2_t) * ScaledSiLU'(hb[t])) W1_t  per relation.
// =====================================================================================================================
// LDS floats of the phase: the gxh tile [16][3H + 8] (3H + 8 = 8 mod 64 like H + 8: the same conflict-free ds_read_b128 pattern),
// then the gh tile [16][H + 8]
constexpr int pre_bwd16_tile_floats(int H) { return kTR16 * (3 * H + 8); }

template <int H>
struct Tile16Regs3 { f32x4 v[kTR16 * 3 * H / 4 / 256]; };

// `parts` != null: gn_t goes to parts[t] (the unfused form: the consumer sums the relations and runs the LayerNorm backward);
// null: gsum = (gn_0 + gn_1) + gn_2 ... in this lane's accumulator positions (the same additions in the same order as the
// consumer's).  Per wave and relation 192 + 64 MFMAs, two barriers; relation t + 1's gxh tile travels during relation t's products.
template <int H>
__device__ __forceinline__ void pre_bwd16_phase(const float* gxh, const float* hb, const float* w2tf, const float* w1tf,
                                                float* parts, int Ns, int T, float* tileA, float* bufG, float* scr,
                                                int row0, int nrows, int tid, int lane, int wave, f32x4 (&gsum)[2]) {
  constexpr int LD = H + 8, LDA = 3 * H + 8, V3 = 3 * H / 4, F4 = kTR16 * 3 * H / 4 / 256;
  const int mrow = lane & 15, ch = 4 * (lane >> 4), cw = 32 * wave;
  Tile16Regs3<H> regs;
  Load16 lhb;
  Ring16<2> ra;
  const f32x4* bpa[2];
  auto request = [&](int t) {          // relation t's gxh tile (cooperative, coalesced), hb block and first weight groups
    const rsrc_t g_r = tile_rsrc(gxh + ((size_t)t * Ns + row0) * 3 * H, nrows * 3 * H);
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + it * 256;
      regs.v[it] = bld4(g_r, (idx / V3) * 3 * H + (idx % V3) * 4);
    }
    issue16<H>(lhb, lane, tile_rsrc(hb + ((size_t)t * Ns + row0) * H, nrows * H), cw);
  };
  auto request_w = [&](int t) {
    const f32x4* w2t = reinterpret_cast<const f32x4*>(w2tf + (size_t)t * frag16_floats(H, 3 * H)) + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) bpa[s] = w2t + (size_t)(2 * wave + s) * frag16_f4(3 * H);
    b16_preload(ra, bpa);
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      const int idx = tid + it * 256;
      *reinterpret_cast<f32x4*>(tileA + (idx / V3) * LDA + (idx % V3) * 4) = regs.v[it];
    }
  };
  request(0);
  request_w(0);
  stage();
  __syncthreads();
  STAMP(1);
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    Load16 hb_now = lhb;
    if (t + 1 < T) request(t + 1);                   // in flight during the product below
    fence_sched();
    // ---- ga = gxh[t] W2_t   (K = 3H)
    f32x4 acc[2] = {zero4(), zero4()};
    mma16_panel<3 * H, 2, false>(acc, tileA + mrow * LDA + ch, bpa, ra);
    const f32x4* w1t = reinterpret_cast<const f32x4*>(w1tf + (size_t)t * frag16_floats(H, H)) + lane;
    const f32x4* bpb[2] = {w1t + (size_t)(2 * wave) * frag16_f4(H), w1t + (size_t)(2 * wave + 1) * frag16_f4(H)};
    Ring16<2> rb;
    b16_preload(rb, bpb);
    fence_sched();
    __syncthreads();                                 // every wave has read the gxh tile (and the previous relation's gh tile)
    {
      f32x4 hbv[2];
      finish16(scr, lane, hb_now, hbv);
#pragma unroll
      for (int s = 0; s < 2; ++s) *reinterpret_cast<f32x4*>(bufG + mrow * LD + cw + 16 * s + ch) = acc[s] * dssilu4(hbv[s]);
    }
    if (t + 1 < T) { stage(); request_w(t + 1); }
    fence_sched();
    __syncthreads();
    // ---- gn_t = gh W1_t
    f32x4 acc2[2] = {zero4(), zero4()};
    mma16_panel<H, 2, false>(acc2, bufG + mrow * LD + ch, bpb, rb);
    if (parts != nullptr) {
      store16<H>(scr, lane, acc2, tile_rsrc(parts + ((size_t)t * Ns + row0) * H, nrows * H), cw);
    } else if (t == 0) {
      gsum[0] = acc2[0]; gsum[1] = acc2[1];
    } else {
      gsum[0] += acc2[0]; gsum[1] += acc2[1];
    }
    STAMP(2 + (t < 3 ? t : 3));
  }
}

template <int H>
__global__ __launch_bounds__(256, HN_U16_MINW_BWD) void node_pre_bwd16_kernel(PreBwdArgs a) {
  constexpr int TR = kTR16, LD = H + 8;
  extern __shared__ __align__(16) float lds[];               // [TR][3H + 8], [TR][LD], then 4 x [16][36] scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * TR, nrows = min(TR, a.Ns - row0);
  f32x4 unused[2];
  pre_bwd16_phase<H>(a.gxh, a.hb, a.w2tf, a.w1tf, a.gn, a.Ns, a.T, lds, lds + pre_bwd16_tile_floats(H),
                     lds + pre_bwd16_tile_floats(H) + TR * LD + wave * kScr16Floats, row0, nrows, tid, lane, wave, unused);
}

// =====================================================================================================================
// node_update_bwd on 16-row tiles; FUSE: the backward of the layer ABOVE's node projection runs in front, on the same rows
// =====================================================================================================================
template <int H, bool FUSE>
__global__ __launch_bounds__(256, HN_U16_MINW_BWD) void node_update_bwd16_kernel(UpdBwdArgs a) {
  static_assert(H == 128, "four waves x 32 channels");
  constexpr int TR = kTR16, LD = H + 8, NB16 = H / 16, V = H / 4, F4 = TR * H / 4 / 256;
  // 2 x [TR][LD], then 4 x [16][36] scratch; FUSE: [TR][3H + 8] (later the two tiles), [TR][LD], then the scratch
  extern __shared__ __align__(16) float lds[];
  float* buf0 = lds;
  float* buf1 = lds + TR * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* scr = lds + (FUSE ? pre_bwd16_tile_floats(H) + TR * LD : 2 * TR * LD) + wave * kScr16Floats;
  const TileInfo ti = find_tile(a.type_rowptr, a.T, a.N, TR, blockIdx.x);
  const int row0 = ti.row0, nrows = ti.nrows, t = ti.t;
  if (t >= a.T) {
    for (int idx = tid; idx < nrows * V; idx += 256) {
      const int r = row0 + idx / V, c = (idx % V) * 4;
      *reinterpret_cast<f32x4*>(a.gx1 + (size_t)r * H + c) = zero4();
#pragma unroll
      for (int d = 0; d < 3; ++d) *reinterpret_cast<f32x4*>(a.gvec1 + ((size_t)r * 3 + d) * H + c) = zero4();
    }
    return;
  }
  STAMP(0);
  STAMP_HWID();
  if constexpr (FUSE) {
    // the projection's backward of the layer above on these rows: its sum over the relations stays in registers, goes through
    // one LDS tile into the LayerNorm backward, and gx_out / gvec_out of the tile are formed as in the unfused form
    f32x4 gsum[2];
    float* bufG = lds + pre_bwd16_tile_floats(H);
    pre_bwd16_phase<H>(a.pend.gxh, a.pend.hb, a.pend.w2tf, a.pend.w1tf, nullptr, a.N, a.pend.nparts, lds, bufG, scr, row0, nrows,
                       tid, lane, wave, gsum);
    // (into the gxh tile's space: its last readers -- the last relation's first product -- are two barriers back)
#pragma unroll
    for (int s = 0; s < 2; ++s) *reinterpret_cast<f32x4*>(lds + (lane & 15) * LD + 32 * wave + 16 * s + 4 * (lane >> 4)) = gsum[s];
    __syncthreads();
    materialise_pending<H, TR, true>(a, row0, nrows, tid, lds, LD);
    STAMP(6);
  } else {
    if (a.pend.gn != nullptr) materialise_pending<H, TR>(a, row0, nrows, tid);   // (incoming gradients still in partial sums)
  }
  const f32x4* wx2t = reinterpret_cast<const f32x4*>(a.wx2tf + (size_t)t * frag16_floats(H, 3 * H)) + lane;   // [H, 3H]
  const f32x4* wx0t = reinterpret_cast<const f32x4*>(a.wx0tf + (size_t)t * frag16_floats(2 * H, H)) + lane;   // [2H, H]
  const f32x4* wvt = reinterpret_cast<const f32x4*>(a.wvtf + (size_t)t * frag16_floats(H, 2 * H)) + lane;     // [H, 2H]
  const f32x4* bpa[2];
  const f32x4* bpx[4];
  const f32x4* bpg[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bpa[s] = wx2t + (size_t)(2 * wave + s) * frag16_f4(3 * H);
    bpg[s] = wvt + (size_t)(2 * wave + s) * frag16_f4(2 * H);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) bpx[2 * p + s] = wx0t + (size_t)(p * NB16 + 2 * wave + s) * frag16_f4(H);
  Ring16<2> ra;
  b16_preload(ra, bpa);
  const int mrow = lane & 15, ch = 4 * (lane >> 4), cw = 32 * wave;
  const float inv_sqrt_h = rsqrtf((float)H);
  const rsrc_t gxo_r = tile_rsrc(a.gxo + (size_t)row0 * H, nrows * H);
  const rsrc_t gvo_r = tile_rsrc(a.gvo + (size_t)row0 * 3 * H, nrows * 3 * H);
  const rsrc_t vp_r = tile_rsrc(a.vp + (size_t)row0 * 6 * H, nrows * 6 * H);
  const rsrc_t h2b_r = tile_rsrc(a.h2b + (size_t)row0 * H, nrows * H);
  const rsrc_t q23_r = tile_rsrc(a.q23 + (size_t)row0 * 2 * H, nrows * 2 * H);
  const rsrc_t nrm_r = tile_rsrc(a.nrm + (size_t)row0 * H, nrows * H);
  const rsrc_t act_r = tile_rsrc(a.row_active ? a.row_active + row0 : a.gxo, a.row_active ? nrows : 0);
  const bool all_on = a.row_active == nullptr;
  const rsrc_t gx1_r = tile_rsrc(a.gx1 + (size_t)row0 * H, nrows * H);
  const rsrc_t gvec1_r = tile_rsrc(a.gvec1 + (size_t)row0 * 3 * H, nrows * 3 * H);

  // ---- gq = (gx/sqrt2 | gx vdot/sqrt2 | sum_d gv[d] v1[d]) elementwise, a float4 per thread and position
  Tile16Regs<H> g3;
#pragma unroll
  for (int it = 0; it < F4; ++it) {
    const int idx = tid + it * 256, lr = idx / V, c = (idx % V) * 4;
    const float on = (all_on | (bld(act_r, lr) != 0.f)) ? 1.f : 0.f;
    const f32x4 gx = bld4(gxo_r, lr * H + c) * on;
    f32x4 vd = zero4(), gq3 = zero4();
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const f32x4 v1 = bld4(vp_r, (lr * 3 + d) * 2 * H + c);
      const f32x4 v2 = bld4(vp_r, (lr * 3 + d) * 2 * H + H + c);
      const f32x4 gv = bld4(gvo_r, (lr * 3 + d) * H + c) * on;
      vd += v1 * v2;
      gq3 += gv * v1;
    }
    *reinterpret_cast<f32x4*>(buf0 + lr * LD + c) = gx * kInvSqrt2;
    *reinterpret_cast<f32x4*>(buf1 + lr * LD + c) = gx * vd * (inv_sqrt_h * kInvSqrt2);
    g3.v[it] = gq3;
  }
  // h2b, gx and q of this wave's accumulator positions: requested now, transposed behind the first product
  Load16 lh2, lgx, lq2;
  issue16<H, true>(lh2, lane, h2b_r, cw);
  issue16<H>(lgx, lane, gxo_r, cw);
  issue16<2 * H, true>(lq2, lane, q23_r, cw);
  __syncthreads();
  STAMP(7);
  // ---- ga2 = gq Wx2  (K = 3H: three panels)
  f32x4 acc[2] = {zero4(), zero4()};
  {
    const f32x4* bp1[2] = {bpa[0] + (size_t)frag16_f4(H), bpa[1] + (size_t)frag16_f4(H)};
    const f32x4* bp2[2] = {bpa[0] + (size_t)frag16_f4(2 * H), bpa[1] + (size_t)frag16_f4(2 * H)};
    mma16_panel<H, 2, true>(acc, buf0 + mrow * LD + ch, bpa, ra);
    __syncthreads();                                 // buffer 0 is free
    tile16_store<H, LD>(buf0, g3, tid);
    mma16_panel<H, 2, true>(acc, buf1 + mrow * LD + ch, bp1, ra);
    __syncthreads();                                 // third part in place, buffer 1 free
    mma16_panel<H, 2, false>(acc, buf0 + mrow * LD + ch, bp2, ra);
  }
  STAMP(8);
  Ring16<4> rx;
  b16_preload(rx, bpx);
  fence_sched();
  // ---- gh2 = ga2 * ScaledSiLU'(h2b) -> buffer 1;  s = gvdot / sqrt(H) = gx q / sqrt(2H) per accumulator position
  const float onr = (all_on | (bld(act_r, mrow) != 0.f)) ? 1.f : 0.f;
  f32x4 s_[2], accx[4];
  {
    f32x4 ph2[2], gxv[2], q2[2];
    finish16(scr, lane, lh2, ph2);
    finish16(scr, lane, lgx, gxv);
    finish16(scr, lane, lq2, q2);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      *reinterpret_cast<f32x4*>(buf1 + mrow * LD + cw + 16 * s + ch) = acc[s] * dssilu4(ph2[s]);
      gxv[s] *= onr;
      s_[s] = gxv[s] * q2[s] * (kInvSqrt2 * inv_sqrt_h);
      accx[s] = gxv[s];                  // gxin's gx1 part accumulates onto the identity term gx
      accx[2 + s] = zero4();
    }
  }
  // inputs of the later epilogues, requested one product ahead
  Load16 lnr, lgv, lq3, lw1, lw2;
  issue16<H, true>(lnr, lane, nrm_r, cw);
  issue16<3 * H>(lgv, lane, gvo_r, cw);
  issue16<2 * H, true>(lq3, lane, q23_r, H + cw);
  issue16<6 * H>(lw1, lane, vp_r, cw);
  issue16<6 * H>(lw2, lane, vp_r, H + cw);
  fence_sched();
  __syncthreads();
  STAMP(9);
  // ---- gxin = gh2 Wx0 (gx1 part | g|v2| part)
  mma16_panel<H, 4, false>(accx, buf1 + mrow * LD + ch, bpx, rx);
  STAMP(10);
  Ring16<2> rg;
  b16_preload(rg, bpg);
  fence_sched();
  f32x4 gnn[2], pgv[2], pq3[2], pw1[2], pw2[2];
  {
    f32x4 nv[2], gx1v[2];
    finish16(scr, lane, lnr, nv);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      gnn[s] = accx[2 + s] * rcp_clamped4(nv[s]);     // (rows past the tile's end load 0: their products are zero anyway)
      gx1v[s] = accx[s];
    }
    store16<H>(scr, lane, gx1v, gx1_r, cw);
  }
  finish16(scr, lane, lgv, pgv);
  finish16(scr, lane, lq3, pq3);
  finish16(scr, lane, lw1, pw1);
  finish16(scr, lane, lw2, pw2);
  // ---- gvec1[d] = gv[d] + (gv1[d] | gv2[d]) Wv,  gv1 = gv q3 + s v2,  gv2 = s v1 + gnn v2  (the accumulator starts at gv[d])
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    f32x4 accg[2];
    __syncthreads();                                 // the previous product has read both buffers
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c0 = cw + 16 * s + ch;
      const f32x4 gv = pgv[s] * onr;
      accg[s] = gv;
      *reinterpret_cast<f32x4*>(buf0 + mrow * LD + c0) = gv * pq3[s] + s_[s] * pw2[s];
      *reinterpret_cast<f32x4*>(buf1 + mrow * LD + c0) = s_[s] * pw1[s] + gnn[s] * pw2[s];
    }
    fence_sched();
    Load16 ngv, nw1, nw2;
    if (d < 2) {                                     // in flight during the product below
      issue16<3 * H>(ngv, lane, gvo_r, (d + 1) * H + cw);
      issue16<6 * H>(nw1, lane, vp_r, (d + 1) * 2 * H + cw);
      issue16<6 * H>(nw2, lane, vp_r, (d + 1) * 2 * H + H + cw);
    }
    fence_sched();
    __syncthreads();
    const f32x4* bpg1[2] = {bpg[0] + (size_t)frag16_f4(H), bpg[1] + (size_t)frag16_f4(H)};
    mma16_panel<H, 2, true>(accg, buf0 + mrow * LD + ch, bpg, rg);
    mma16_panel<H, 2, false>(accg, buf1 + mrow * LD + ch, bpg1, rg);
    if (d < 2) b16_preload(rg, bpg);
    fence_sched();
    if (d < 2) {
      finish16(scr, lane, ngv, pgv);
      finish16(scr, lane, nw1, pw1);
      finish16(scr, lane, nw2, pw2);
    }
    store16<3 * H>(scr, lane, accg, gvec1_r, d * H + cw);
    STAMP(11 + d);
  }
}

// =====================================================================================================================
// The read-out (hermnet.py:112-116, 129) on 16-row tiles: Linear(H -> C) on the matrix pipe, ScaledSiLU and the C -> 1 Linear in
// the epilogue; H = 128, C = 64.  Round 5: the VALU form of csrc/node_kernels.hip (every workgroup stages the whole weight in LDS)
// took 24 us forward + 16 us backward at 10k rows for 0.3 GFLOP.
// =====================================================================================================================
struct HeadArgs {
  const float* x;        // fwd: x [N, H];  bwd: h [N, C] (saved pre-activations)
  const float* wf;       // fwd: frag16(W0 [C, H]);  bwd: frag16(W0^T [H, C])
  const float* b0;       // [C]
  const float* w2;       // [C]
  const float* b2;       // [1] or null
  const float* ge;       // bwd: [N]
  const float* mask;     // [N] or null
  float* h;              // fwd: [N, C] saved
  float* out;            // fwd: e [N];  bwd: gx [N, H]
  int N;
};

template <int H, int C>
__global__ __launch_bounds__(256) void energy_head16_fwd_kernel(HeadArgs a) {
  static_assert(H == 128 && C == 64, "four waves x 16 output channels");
  constexpr int TR = kTR16, LD = H + 8;
  extern __shared__ __align__(16) float lds[];               // [TR][LD], then [4][16] partial row sums
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * TR, nrows = min(TR, a.N - row0);
  const int mrow = lane & 15, ch = 4 * (lane >> 4);
  const f32x4* bp[1] = {reinterpret_cast<const f32x4*>(a.wf) + (size_t)wave * frag16_f4(H) + lane};
  Ring16<1> ring;
  b16_preload(ring, bp);
  f32x4 acc[1] = {ld4g(a.b0 + 16 * wave + ch)};
  const f32x4 w2v = ld4g(a.w2 + 16 * wave + ch);
  Tile16Regs<H> regs;
  tile16_load<H>(regs, tile_rsrc(a.x + (size_t)row0 * H, nrows * H), H, 0, tid);
  tile16_store<H, LD>(lds, regs, tid);
  __syncthreads();
  mma16_panel<H, 1, false>(acc, lds + mrow * LD + ch, bp, ring);
  const f32x4 hv = acc[0];
  if (mrow < nrows) *reinterpret_cast<f32x4*>(a.h + (size_t)(row0 + mrow) * C + 16 * wave + ch) = hv;
  const f32x4 av = ssilu4(hv);                               // (incl. the 1 / 0.6)
  float s = av[0] * w2v[0];
  s = fmaf(av[1], w2v[1], s);
  s = fmaf(av[2], w2v[2], s);
  s = fmaf(av[3], w2v[3], s);
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  float* part = lds + TR * LD;
  if (lane < 16) part[wave * 16 + lane] = s;
  __syncthreads();
  if (tid < nrows) {
    const float e = ((part[tid] + part[16 + tid]) + (part[32 + tid] + part[48 + tid])) + (a.b2 ? a.b2[0] : 0.f);
    a.out[row0 + tid] = e * (a.mask ? a.mask[row0 + tid] : 1.0f);
  }
}

template <int H, int C>
__global__ __launch_bounds__(256) void energy_head16_bwd_kernel(HeadArgs a) {
  static_assert(H == 128 && C == 64, "four waves x 32 output channels");
  constexpr int TR = kTR16, LDC = C + 8;
  extern __shared__ __align__(16) float lds[];               // [TR][LDC], then 4 x [16][36] scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * TR, nrows = min(TR, a.N - row0);
  const int mrow = lane & 15, ch = 4 * (lane >> 4);
  float* scr = lds + TR * LDC + wave * kScr16Floats;
  const f32x4* w = reinterpret_cast<const f32x4*>(a.wf) + lane;
  const f32x4* bp[2] = {w + (size_t)(2 * wave) * frag16_f4(C), w + (size_t)(2 * wave + 1) * frag16_f4(C)};
  Ring16<2> ring;
  b16_preload(ring, bp);
  {   // gh = ge mask w2 ScaledSiLU'(h): one float4 per thread
    const int lr = tid / (C / 4), c = (tid % (C / 4)) * 4;
    f32x4 g = zero4();
    if (lr < nrows) {
      const int row = row0 + lr;
      const float f = a.ge[row] * (a.mask ? a.mask[row] : 1.0f);
      g = dssilu4(ld4g(a.x + (size_t)row * C + c)) * ld4g(a.w2 + c) * f;
    }
    *reinterpret_cast<f32x4*>(lds + lr * LDC + c) = g;
  }
  __syncthreads();
  f32x4 acc[2] = {zero4(), zero4()};
  mma16_panel<C, 2, false>(acc, lds + mrow * LDC + ch, bp, ring);
  store16<H>(scr, lane, acc, tile_rsrc(a.out + (size_t)row0 * H, nrows * H), 32 * wave);
}

}  // namespace

int hn_head16_supported(int hidden, int cols) { return hidden == 128 && cols == 64; }

int hn_head16_fwd(const float* x, const float* w0_frag16, const float* b0, const float* w2, const float* b2, const float* mask,
                  float* h, float* e, int rows, void* stream) {
  HeadArgs a = {x, w0_frag16, b0, w2, b2, nullptr, mask, h, e, rows};
  return launch_chain(energy_head16_fwd_kernel<128, 64>, dim3((unsigned)((rows + 15) / 16)), (size_t)(16 * 136 + 64) * 4, stream, a);
}

int hn_head16_bwd(const float* ge, const float* h, const float* w0t_frag16, const float* w2, const float* mask, float* gx,
                  int rows, void* stream) {
  HeadArgs a = {h, w0t_frag16, nullptr, w2, nullptr, ge, mask, nullptr, gx, rows};
  return launch_chain(energy_head16_bwd_kernel<128, 64>, dim3((unsigned)((rows + 15) / 16)), (size_t)(16 * 72 + 4 * kScr16Floats) * 4,
                      stream, a);
}

namespace {
}  // namespace

// Entry points of this translation unit (called by node_chain.hip's dispatch).  The weight pointers of the argument blocks
// must hold frag16 copies (include/hermnet_hip.h: hermnet_node_update_fwd, "16-row form").
int hn_update16_supported(int hidden) { return hidden == 128; }

namespace {
constexpr size_t kLdsUpd16 = (size_t)(2 * 16 * 136 + 4 * kScr16Floats) * 4;
constexpr size_t kLdsFwdFused16 = (size_t)(3 * 16 * 136 + 4 * kScr16Floats) * 4;
constexpr size_t kLdsBwdFused16 = (size_t)(pre_bwd16_tile_floats(128) + 16 * 136 + 4 * kScr16Floats) * 4;

int launch_fwd16(bool fuse, int tiles, size_t lds_bytes, void* stream, const UpdFwdArgs& a, const PreFwdArgs& p) {
  if (fuse) hipLaunchKernelGGL((node_update_fwd16_kernel<128, true>), dim3((unsigned)tiles), dim3(256), lds_bytes, (hipStream_t)stream, a, p);
  else hipLaunchKernelGGL((node_update_fwd16_kernel<128, false>), dim3((unsigned)tiles), dim3(256), lds_bytes, (hipStream_t)stream, a, p);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}
}  // namespace

int hn_update16_fwd(int hidden, const UpdFwdArgs& a, int tiles, void* stream) {
  if (hidden != 128) return HN_ERR_BAD_ARG;
  return launch_fwd16(false, tiles, kLdsFwdFused16, stream, a, PreFwdArgs{});
}

// update of layer l + node projection of layer l + 1 on the same 16-row tiles (p.x is unused: the rows come from the update)
int hn_update16_pre_fwd(int hidden, const UpdFwdArgs& a, const PreFwdArgs& p, int tiles, void* stream) {
  if (hidden != 128 || p.Ns != a.N || p.src_ranges != nullptr || p.wmode != 0) return HN_ERR_BAD_ARG;
  return launch_fwd16(true, tiles, kLdsFwdFused16, stream, a, p);
}

int hn_update16_bwd(int hidden, const UpdBwdArgs& a, int tiles, void* stream) {
  if (hidden != 128) return HN_ERR_BAD_ARG;
  if (a.pend.gxh != nullptr)
    return launch_chain(node_update_bwd16_kernel<128, true>, dim3((unsigned)tiles), kLdsBwdFused16, stream, a);
  return launch_chain(node_update_bwd16_kernel<128, false>, dim3((unsigned)tiles), kLdsUpd16, stream, a);
}

int hn_pre16_fwd(int hidden, const PreFwdArgs& p, void* stream) {
  if (hidden != 128 || p.src_ranges != nullptr || p.wmode != 0) return HN_ERR_BAD_ARG;
  return launch_chain(node_pre_fwd16_kernel<128>, dim3((unsigned)((p.Ns + 15) / 16)), kLdsFwdFused16, stream, p);
}

int hn_pre16_bwd(int hidden, const PreBwdArgs& a, void* stream) {
  if (hidden != 128 || a.src_ranges != nullptr || a.wmode != 0) return HN_ERR_BAD_ARG;
  return launch_chain(node_pre_bwd16_kernel<128>, dim3((unsigned)((a.Ns + 15) / 16)), kLdsBwdFused16, stream, a);
}

#ifdef HN_STAMPS
extern "C" int hermnet_debug_stamps16(unsigned long long* out_host, int count) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(hn_stamps), (size_t)count * sizeof(unsigned long long)) == hipSuccess ? 0 : 3;
}
#endif
