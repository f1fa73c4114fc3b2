// gfx950: the four node chain kernels (node_chain.hip) for EVERY hidden width that is a multiple of 64 up to 512 -- the
// reference's default is hidden_channels = 512 (/root/reference/HermNet/hermnet.py:84-88), its examples use it.
//
//   node_pre_fwd     x -> LayerNorm -> [H -> H] -> ScaledSiLU -> [H -> 3H]  = xh[t]        rmnet.py:52
//   node_pre_bwd     gxh[t] -> [3H -> H] -> * ScaledSiLU' -> [H -> H]       = gn[t]
//   node_update_fwd  (x1, vec1) -> vec_proj, vec_dot, |v2|, xvec_proj MLP, dx / dvec, residual   rmnet.py:94-107, 29-31
//   node_update_bwd  (gx_out, gvec_out) -> (gx1, gvec1)
//
// Same data flow as node_chain.hip (weights streamed from L2 in MFMA fragment order, activations in LDS tiles, transposed
// 32 x 32 accumulators, coalesced epilogue I/O through wave-private LDS transposes); what differs is how a 256-thread
// workgroup covers a wide row:
//   * a tile is 32 rows; wave w owns the 32-channel blocks {w, w + 4, w + 8, ...} (NP = ceil(H / 128) of them; the last
//     round is partly empty when H is an odd multiple of 64: those waves compute on a clamped block and store nothing);
//   * a product with ONE output part per block (H -> H, 2H -> H, 3H -> H) runs on all NP blocks of the wave at once
//     (NP accumulators share every activation read, the weight stream is continuous across K panels);
//   * a product with 2 or 3 parts per block (v1|v2, s|a|b, p|q|r, gx1|g|v2|) runs block by block -- NP passes with 2 or 3
//     accumulators -- so the register budget does not grow with H;
//   * nothing is carried in registers between products except NP accumulator blocks (|v2|^2, s, g|v2| / |v2|): v1, v2,
//     vec_dot, x1 ... are re-read (L2-hot, coalesced) where the narrow kernels keep them -- at these widths a product is
//     tens of thousands of cycles, a block load a few hundred.
// LDS: one or two [32][H + 4] tiles + 18 KB of transpose scratch: 150 KB at H = 512 (one workgroup per CU).
#include <type_traits>
#include "node_chain_common.h"

// ring slots of the wide family's products: one workgroup per CU, so a wave has nobody to hide a late weight fragment behind --
// five steps ahead (the 32/64-row kernels of node_chain.hip, two or more workgroups per CU, take two: node_chain_common.h)
#ifndef HN_RS_WIDE
#define HN_RS_WIDE 6
#endif
constexpr int kRSW = HN_RS_WIDE;

namespace {

template <int H_>
struct WCfg {
  static constexpr int H = H_, TR = 32;
  static constexpr int CB = H / 32;                 // 32-channel blocks
  static constexpr int NP = (CB + 3) / 4;           // blocks per wave
  static constexpr int LD = H + 4;
  static_assert(H % 64 == 0 && H >= 128 && H <= 512, "widths: multiples of 64 from 128 to 512");
};

// blocks of this wave: cb[j] = wave + 4 j; ok[j] = cb[j] < CB; cbc[j] = a valid block to compute on when !ok[j]
template <int CB, int NP>
struct Blocks {
  int cb[NP];
  bool ok[NP];
  int cbc[NP];
  __device__ __forceinline__ explicit Blocks(int wave) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      cb[j] = wave + 4 * j;
      ok[j] = cb[j] < CB;
      cbc[j] = ok[j] ? cb[j] : wave;
    }
  }
};

// acc[0][j] += W_{blocks} . A^T over K panel `KP` for all NP blocks at once
template <int KP, int LD, int NP, bool MORE, bool PIN = true>
__device__ __forceinline__ void mma_all(f32x16 (&acc)[1][NP], const float* As, const f32x4* const (&bp)[NP], BRing<NP, kRSW>& ring) {
  mma_panel<KP, LD, 1, NP, kRSW, MORE, PIN>(acc, As, bp, ring);
}

// =====================================================================================================================
template <int H>
__global__ __launch_bounds__(256, 1) void node_pre_fwd_wide_kernel(PreFwdArgs a) {
  using C = WCfg<H>;
  constexpr int TR = C::TR, LD = C::LD, CB = C::CB, NP = C::NP;
  extern __shared__ __align__(16) float tile[];           // [TR][LD], then 4 x [32][36] scratch
  const int t = blockIdx.y, row0 = blockIdx.x * TR;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* scr = tile + TR * LD + wave * kScrFloats;
  const int nrows = min(TR, a.Ns - row0);
  if (!tile_selected(a.windows, a.nwin, a.wmode, row0, TR)) return;
  if (!tile_wanted(a.src_ranges, t, row0, TR)) return;
  const Blocks<CB, NP> B(wave);
  const rsrc_t x_r = tile_rsrc(a.x + (size_t)row0 * H, nrows * H);
  const rsrc_t hb_r = tile_rsrc(a.hb + ((size_t)t * a.Ns + row0) * H, nrows * H);
  const rsrc_t xh_r = tile_rsrc(a.xh + ((size_t)t * a.Ns + row0) * 3 * H, nrows * 3 * H);
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.w1f + (size_t)t * H * H * 3 / 2) + lane;
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.w2f + (size_t)t * 3 * H * H * 3 / 2) + lane;
  const f32x4* bp1[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) bp1[j] = w1 + (size_t)B.cbc[j] * frag_f4(H);
  BRing<NP, kRSW> r1;
  b_preload(r1, bp1);

  // ---- LayerNorm without affine (rmnet.py:52): 8 adjacent lanes share a row, statistics over the first Hr channels
  {
    constexpr int TPR = 256 / TR, NF = H / 4 / TPR;
    const int lr = tid / TPR, q = tid % TPR;
    f32x4 v[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) v[k] = bld4(x_r, lr * H + (k * TPR + q) * 4);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NF; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if ((k * TPR + q) * 4 + e >= a.Hr) v[k][e] = 0.f;
        s += v[k][e];
      }
#pragma unroll
    for (int m = 1; m < TPR; m <<= 1) s += __shfl_xor(s, m, 64);
    float mu = s / (float)a.Hr;
    // corrected two-pass statistics (round 6): a lane sums NF * 4 values in sequence, so on a row whose mean dwarfs its spread
    // (1e3 under unit noise) `mu` carries an error of ~1e-4 of the spread -- 20 x what torch's pairwise sum leaves
    // (tests/test_gpu_parity.py::test_node_chain_kernels_on_adversarial_operands).  The mean of the DEVIATIONS is that error,
    // measured at the deviations' own scale: it is taken out of the deviations, the mean and the variance.
    float qq = 0.f, dd = 0.f;
#pragma unroll
    for (int k = 0; k < NF; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[k][e] = (k * TPR + q) * 4 + e < a.Hr ? v[k][e] - mu : 0.f;
        dd += v[k][e];
        qq = fmaf(v[k][e], v[k][e], qq);
      }
#pragma unroll
    for (int m = 1; m < TPR; m <<= 1) {
      qq += __shfl_xor(qq, m, 64);
      dd += __shfl_xor(dd, m, 64);
    }
    const float dm = dd / (float)a.Hr;
    mu += dm;
#pragma unroll
    for (int k = 0; k < NF; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((k * TPR + q) * 4 + e < a.Hr) v[k][e] -= dm;
    float var = qq / (float)a.Hr;
    var = fmaf(-dm, dm, var);
    const float rs = rsqrtf(fmaxf(var, 0.f) + a.eps);
#pragma unroll
    for (int k = 0; k < NF; ++k) *reinterpret_cast<f32x4*>(tile + lr * LD + (k * TPR + q) * 4) = v[k] * rs;
    if ((t == 0 || a.src_ranges != nullptr) && q == 0 && lr < nrows) { a.mean[row0 + lr] = mu; a.rstd[row0 + lr] = rs; }
  }
  __syncthreads();

  // ---- h = n W1^T on all blocks of the wave
  const int mrow = lane & 31, ch = 4 * (lane >> 5);
  const float* As = tile + mrow * LD + ch;
  f32x16 acc1[1][NP];
  zero_acc(acc1);
  mma_all<H, LD, NP, false, HN_PIN_PRE>(acc1, As, bp1, r1);
  __syncthreads();                                   // every wave has read n
  // ---- + b1, save, ScaledSiLU -> the tile becomes the A operand of the second product
#pragma unroll
  for (int j = 0; j < NP; ++j)
    if (B.ok[j]) {
      f32x4 hv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = B.cb[j] * 32 + 8 * g + ch;
        hv[g] = run4(acc1[0][j], g) + ld4g(a.b1 + (size_t)t * H + c0);
        *reinterpret_cast<f32x4*>(tile + mrow * LD + c0) = ssilu4(hv[g]);
      }
      store_block<H>(scr, lane, hv, hb_r, B.cb[j] * 32);
    }
  __syncthreads();
  // ---- xh = a W2^T + b2: block by block, the three parts (s | a | b) of a block together
#pragma unroll
  for (int j = 0; j < NP; ++j)
    if (B.ok[j]) {
      const f32x4* bp2[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bp2[p] = w2 + (size_t)(p * CB + B.cb[j]) * frag_f4(H);
      BRing<3, kRSW> r2;
      b_preload(r2, bp2);
      f32x16 acc2[1][3];
      zero_acc(acc2);
      mma_panel<H, LD, 1, 3, kRSW, false, HN_PIN_PRE>(acc2, As, bp2, r2);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const int cblk = p * H + B.cb[j] * 32;
        f32x4 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = run4(acc2[0][p], g) + ld4g(a.b2 + (size_t)t * 3 * H + cblk + 8 * g + ch);
        store_block<3 * H>(scr, lane, v, xh_r, cblk);
      }
    }
}

// =====================================================================================================================
constexpr int pre_bwd_wide_kc(int H) { return H % 128 == 0 ? 128 : 64; }
constexpr int pre_bwd_wide_lds_floats(int H) {
  const int chunks = 2 * 32 * (pre_bwd_wide_kc(H) + 4), tile = 32 * (H + 4);
  return (chunks > tile ? chunks : tile) + 4 * kScrFloats;
}

template <int H>
__global__ __launch_bounds__(256, 1) void node_pre_bwd_wide_kernel(PreBwdArgs a) {
  using C = WCfg<H>;
  constexpr int TR = C::TR, LD = C::LD, CB = C::CB, NP = C::NP;
  constexpr int KC = pre_bwd_wide_kc(H), NCH = 3 * H / KC, LDC = KC + 4;
  constexpr int BODY = 2 * TR * LDC > TR * LD ? 2 * TR * LDC : TR * LD;
  extern __shared__ __align__(16) float lds[];            // two [TR][LDC] chunk buffers / the gh tile, then the scratch
  const int t = blockIdx.y, row0 = blockIdx.x * TR;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* scr = lds + BODY + wave * kScrFloats;
  const int nrows = min(TR, a.Ns - row0);
  const int mrow = lane & 31, ch = 4 * (lane >> 5);
  if (!tile_selected(a.windows, a.nwin, a.wmode, row0, TR)) return;
  if (!tile_wanted(a.src_ranges, t, row0, TR)) {
    float* g0 = a.gn + ((size_t)t * a.Ns + row0) * H;
    for (int i = tid; i < nrows * (H / 4); i += 256) reinterpret_cast<f32x4*>(g0)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    return;
  }
  const Blocks<CB, NP> B(wave);
  const f32x4* w2t = reinterpret_cast<const f32x4*>(a.w2tf + (size_t)t * 3 * H * H * 3 / 2) + lane;
  const f32x4* w1t = reinterpret_cast<const f32x4*>(a.w1tf + (size_t)t * H * H * 3 / 2) + lane;
  const f32x4* bpa[NP];
  const f32x4* bpb[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    bpa[j] = w2t + (size_t)B.cbc[j] * frag_f4(3 * H);
    bpb[j] = w1t + (size_t)B.cbc[j] * frag_f4(H);
  }
  BRing<NP, kRSW> ra;
  b_preload(ra, bpa);
  const rsrc_t gxh_r = tile_rsrc(a.gxh + ((size_t)t * a.Ns + row0) * 3 * H, nrows * 3 * H);
  const rsrc_t hb_r = tile_rsrc(a.hb + ((size_t)t * a.Ns + row0) * H, nrows * H);
  const rsrc_t gn_r = tile_rsrc(a.gn + ((size_t)t * a.Ns + row0) * H, nrows * H);
  TileRegs<TR, KC> regs;
  tile_load<TR, KC>(regs, gxh_r, 3 * H, 0, tid);

  // ---- ga = gxh W2   (K = 3H in chunks, double-buffered through registers)
  f32x16 acc[1][NP];
  zero_acc(acc);
  auto chunk = [&](int kc, auto more) {
    float* buf = lds + (kc & 1) * TR * LDC;
    tile_store<TR, KC, LDC>(buf, regs, tid);
    __syncthreads();
    if (kc + 1 < NCH) tile_load<TR, KC>(regs, gxh_r, 3 * H, (kc + 1) * KC, tid);
    const float* As = buf + mrow * LDC + ch;
    const f32x4* bpk[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) bpk[j] = bpa[j] + (size_t)kc * frag_f4(KC);
    mma_panel<KC, LDC, 1, NP, kRSW, decltype(more)::value>(acc, As, bpk, ra);
  };
#pragma unroll 1
  for (int kc = 0; kc < NCH - 1; ++kc) chunk(kc, std::true_type());     // the weight stream continues behind the chunk
  chunk(NCH - 1, std::false_type());
  BRing<NP, kRSW> rb_;
  b_preload(rb_, bpb);
  __syncthreads();                                   // the chunk buffers are free
  // ---- gh = ga * ScaledSiLU'(hb) -> tile
  float* tile = lds;
#pragma unroll
  for (int j = 0; j < NP; ++j)
    if (B.ok[j]) {
      BlockLoad lhb;
      issue_block<H>(lhb, lane, hb_r, B.cb[j] * 32);
      f32x4 hbv[4];
      finish_block(scr, lane, lhb, hbv);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(tile + mrow * LD + B.cb[j] * 32 + 8 * g + ch) = run4(acc[0][j], g) * dssilu4(hbv[g]);
    }
  __syncthreads();
  // ---- gn_t = gh W1
  f32x16 acc2[1][NP];
  zero_acc(acc2);
  mma_all<H, LD, NP, false>(acc2, tile + mrow * LD + ch, bpb, rb_);
#pragma unroll
  for (int j = 0; j < NP; ++j)
    if (B.ok[j]) {
      const f32x4 v[4] = {run4(acc2[0][j], 0), run4(acc2[0][j], 1), run4(acc2[0][j], 2), run4(acc2[0][j], 3)};
      store_block<H>(scr, lane, v, gn_r, B.cb[j] * 32);
    }
}

// =====================================================================================================================
template <int H>
__global__ __launch_bounds__(256, 1) void node_update_fwd_wide_kernel(UpdFwdArgs a) {
  using C = WCfg<H>;
  constexpr int TR = C::TR, LD = C::LD, CB = C::CB, NP = C::NP;
  extern __shared__ __align__(16) float lds[];            // 2 x [TR][LD], then 4 x [32][36] scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* scr = lds + 2 * TR * LD + wave * kScrFloats;
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
  const Blocks<CB, NP> B(wave);
  const f32x4* wv = reinterpret_cast<const f32x4*>(a.wvf + (size_t)t * 2 * H * H * 3 / 2) + lane;
  const f32x4* wx0 = reinterpret_cast<const f32x4*>(a.wx0f + (size_t)t * 2 * H * H * 3 / 2) + lane;
  const f32x4* wx2 = reinterpret_cast<const f32x4*>(a.wx2f + (size_t)t * 3 * H * H * 3 / 2) + lane;
  const int mrow = lane & 31, ch = 4 * (lane >> 5);
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
  f32x4 sq[NP][4];
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) sq[j][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- vp[d] = vec1[d] Wv^T for the three Cartesian components (v1 | v2 of a block together); |v2|^2 accumulates
  TileRegs<TR, H> regs;
  tile_load<TR, H>(regs, vec1_r, 3 * H, 0, tid);
#pragma unroll 1
  for (int d = 0; d < 3; ++d) {
    float* buf = lds + (d & 1) * TR * LD;
    tile_store<TR, H, LD>(buf, regs, tid);
    __syncthreads();                                 // (also: every wave has finished the product of d - 1)
    if (d < 2) tile_load<TR, H>(regs, vec1_r, 3 * H, (d + 1) * H, tid);
    else tile_load<TR, H>(regs, x1_r, H, 0, tid);
#pragma unroll
    for (int j = 0; j < NP; ++j)
      if (B.ok[j]) {
        const f32x4* bpv[2] = {wv + (size_t)B.cb[j] * frag_f4(H), wv + (size_t)(CB + B.cb[j]) * frag_f4(H)};
        BRing<2, kRSW> rv;
        b_preload(rv, bpv);
        f32x16 accv[1][2];
        zero_acc(accv);
        mma_panel<H, LD, 1, 2, kRSW, false, true, false>(accv, buf + mrow * LD + ch, bpv, rv);
        f32x4 v1[4], v2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          v1[g] = run4(accv[0][0], g);
          v2[g] = run4(accv[0][1], g);
          sq[j][g] += v2[g] * v2[g];
        }
        const int ob = d * 2 * H + B.cb[j] * 32;       // the block inside the tile's [32][3][2H] rows
        store_block<6 * H>(scr, lane, v1, vp_r, ob);
        store_block<6 * H>(scr, lane, v2, vp_r, ob + H);
      }
  }
  // ---- xin = [x1 | sqrt(|v2|^2 + 1e-8)]: x1 -> buffer 1 (last read by the product of d = 1), the norm -> buffer 0
  float* bufx = lds + TR * LD;
  float* bufn = lds;
  tile_store<TR, H, LD>(bufx, regs, tid);
  __syncthreads();                                   // every wave has finished the product of d = 2 (buffer 0)
#pragma unroll
  for (int j = 0; j < NP; ++j)
    if (B.ok[j]) {
      f32x4 nv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 s2 = sq[j][g] + 1e-8f;
        nv[g] = (f32x4){sqrtf(s2[0]), sqrtf(s2[1]), sqrtf(s2[2]), sqrtf(s2[3])};
        *reinterpret_cast<f32x4*>(bufn + mrow * LD + B.cb[j] * 32 + 8 * g + ch) = nv[g];
      }
      store_block<H>(scr, lane, nv, nrm_r, B.cb[j] * 32);
    }
  const f32x4* bpx[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) bpx[j] = wx0 + (size_t)B.cbc[j] * frag_f4(2 * H);
  BRing<NP, kRSW> rx;
  b_preload(rx, bpx);
  __syncthreads();
  // ---- h2 = xin Wx0^T + bx0 (K = 2H: two panels) on all blocks of the wave
  f32x16 acch[1][NP];
  zero_acc(acch);
  {
    const f32x4* bpx1[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) bpx1[j] = bpx[j] + (size_t)frag_f4(H);
    mma_all<H, LD, NP, true>(acch, bufx + mrow * LD + ch, bpx, rx);
    mma_all<H, LD, NP, false>(acch, bufn + mrow * LD + ch, bpx1, rx);
  }
  __syncthreads();                                   // buffer 0 is free (x1 stays in buffer 1 for the last epilogue)
#pragma unroll
  for (int j = 0; j < NP; ++j)
    if (B.ok[j]) {
      f32x4 hv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = B.cb[j] * 32 + 8 * g + ch;
        hv[g] = run4(acch[0][j], g) + ld4g(a.bx0 + (size_t)t * H + c0);
        *reinterpret_cast<f32x4*>(lds + mrow * LD + c0) = ssilu4(hv[g]);
      }
      store_block<H>(scr, lane, hv, h2b_r, B.cb[j] * 32);
    }
  __syncthreads();
  // ---- (p | q | r) = a2 Wx2^T + bx2 block by block, then the update and the residual.  v1, v2 and vec1 of the block
  // are requested before the product (coalesced) and transposed behind it; x1 is still in buffer 1.
  const float on = (all_on | (bld(act_r, mrow) != 0.f)) ? 1.f : 0.f;
  const float inv_sqrt_h = rsqrtf((float)H);
#pragma unroll 1
  for (int j = 0; j < NP; ++j) {
    const int cb = wave + 4 * j;
    if (cb >= CB) break;
    const int cblk = cb * 32;
    const f32x4* bpq[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bpq[p] = wx2 + (size_t)(p * CB + cb) * frag_f4(H);
    BRing<3, kRSW> rq;
    b_preload(rq, bpq);
    BlockLoad lvv[3], lv1[3], lv2[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      issue_block<3 * H>(lvv[d], lane, vec1_r, d * H + cblk);
      issue_block<6 * H>(lv1[d], lane, vp_r, d * 2 * H + cblk);
      issue_block<6 * H>(lv2[d], lane, vp_r, d * 2 * H + H + cblk);
    }
    fence_sched();
    f32x16 accq[1][3];
    zero_acc(accq);
    mma_panel<H, LD, 1, 3, kRSW, false, true, false>(accq, lds + mrow * LD + ch, bpq, rq);
    fence_sched();
    f32x4 q[4], r[4], xo[4], dot[4], v1d[3][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) dot[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      f32x4 v2[4];
      finish_block(scr, lane, lv1[d], v1d[d]);
      finish_block(scr, lane, lv2[d], v2);
#pragma unroll
      for (int g = 0; g < 4; ++g) dot[g] += v1d[d][g] * v2[g];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = cblk + 8 * g + ch;
      const float* bb = a.bx2 + (size_t)t * 3 * H + c0;
      const f32x4 p = run4(accq[0][0], g) + ld4g(bb);
      q[g] = run4(accq[0][1], g) + ld4g(bb + H);
      r[g] = run4(accq[0][2], g) + ld4g(bb + 2 * H);
      const f32x4 x1v = *reinterpret_cast<const f32x4*>(bufx + mrow * LD + c0);
      xo[g] = (x1v + (p + q[g] * dot[g] * inv_sqrt_h) * kInvSqrt2) * on;
    }
    store_block<2 * H>(scr, lane, q, q23_r, cblk);
    store_block<2 * H>(scr, lane, r, q23_r, H + cblk);
    store_block<H>(scr, lane, xo, xo_r, cblk);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      f32x4 vo[4];
      finish_block(scr, lane, lvv[d], vo);
#pragma unroll
      for (int g = 0; g < 4; ++g) vo[g] = (vo[g] + r[g] * v1d[d][g]) * on;
      store_block<3 * H>(scr, lane, vo, vo_r, d * H + cblk);
    }
  }
}

// =====================================================================================================================
template <int H>
__global__ __launch_bounds__(256, 1) void node_update_bwd_wide_kernel(UpdBwdArgs a) {
  using C = WCfg<H>;
  constexpr int TR = C::TR, LD = C::LD, CB = C::CB, NP = C::NP, V = H / 4, F4 = TR * H / 4 / 256;
  extern __shared__ __align__(16) float lds[];            // 2 x [TR][LD], then 4 x [32][36] scratch
  float* buf0 = lds;
  float* buf1 = lds + TR * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* scr = lds + 2 * TR * LD + wave * kScrFloats;
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
  if (a.pend.gn != nullptr) materialise_pending<H, TR>(a, row0, nrows, tid);   // (incoming gradients still in partial sums)
  const Blocks<CB, NP> B(wave);
  const f32x4* wx2t = reinterpret_cast<const f32x4*>(a.wx2tf + (size_t)t * 3 * H * H * 3 / 2) + lane;
  const f32x4* wx0t = reinterpret_cast<const f32x4*>(a.wx0tf + (size_t)t * 2 * H * H * 3 / 2) + lane;
  const f32x4* wvt = reinterpret_cast<const f32x4*>(a.wvtf + (size_t)t * 2 * H * H * 3 / 2) + lane;
  const f32x4* bpa[NP];
  const f32x4* bpg[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    bpa[j] = wx2t + (size_t)B.cbc[j] * frag_f4(3 * H);
    bpg[j] = wvt + (size_t)B.cbc[j] * frag_f4(2 * H);
  }
  BRing<NP, kRSW> ra;
  b_preload(ra, bpa);
  const int mrow = lane & 31, ch = 4 * (lane >> 5);
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

  // ---- gq = (gx/sqrt2 | gx vdot/sqrt2 | sum_d gv[d] v1[d]) elementwise, a float4 per thread and position; the first
  // two parts now, the third into buffer 0 once the first panel has been read (nothing is parked in registers across
  // a product at these widths: 64 registers per lane at H = 512)
#pragma unroll
  for (int it = 0; it < F4; ++it) {
    if ((it & 1) == 0) fence_sched();     // two positions in flight, not all F4
    const int idx = tid + it * 256, lr = idx / V, c = (idx % V) * 4;
    const float on = (all_on | (bld(act_r, lr) != 0.f)) ? 1.f : 0.f;
    const f32x4 gx = bld4(gxo_r, lr * H + c) * on;
    f32x4 vd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) vd += bld4(vp_r, (lr * 3 + d) * 2 * H + c) * bld4(vp_r, (lr * 3 + d) * 2 * H + H + c);
    *reinterpret_cast<f32x4*>(buf0 + lr * LD + c) = gx * kInvSqrt2;
    *reinterpret_cast<f32x4*>(buf1 + lr * LD + c) = gx * vd * (inv_sqrt_h * kInvSqrt2);
  }
  __syncthreads();
  // ---- ga2 = gq Wx2  (K = 3H: three panels) on all blocks of the wave
  f32x16 acc[1][NP];
  zero_acc(acc);
  {
    const f32x4* bp1[NP];
    const f32x4* bp2[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { bp1[j] = bpa[j] + (size_t)frag_f4(H); bp2[j] = bpa[j] + (size_t)frag_f4(2 * H); }
    mma_all<H, LD, NP, true>(acc, buf0 + mrow * LD + ch, bpa, ra);
    __syncthreads();                                 // buffer 0 is free
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      if ((it & 1) == 0) fence_sched();
      const int idx = tid + it * 256, lr = idx / V, c = (idx % V) * 4;
      const float on = (all_on | (bld(act_r, lr) != 0.f)) ? 1.f : 0.f;
      f32x4 gq3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 3; ++d) gq3 += bld4(gvo_r, (lr * 3 + d) * H + c) * bld4(vp_r, (lr * 3 + d) * 2 * H + c);
      *reinterpret_cast<f32x4*>(buf0 + lr * LD + c) = gq3 * on;
    }
    mma_all<H, LD, NP, true>(acc, buf1 + mrow * LD + ch, bp1, ra);
    __syncthreads();                                 // third part in place, buffer 1 free
    mma_all<H, LD, NP, false>(acc, buf0 + mrow * LD + ch, bp2, ra);
  }
  // ---- gh2 = ga2 * ScaledSiLU'(h2b) -> buffer 1;  s = gvdot / sqrt(H) = gx q / sqrt(2H) per accumulator position
  const float onr = (all_on | (bld(act_r, mrow) != 0.f)) ? 1.f : 0.f;
  f32x4 s_[NP][4];
#pragma unroll
  for (int j = 0; j < NP; ++j)
    if (B.ok[j]) {
      BlockLoad lh2, lgx, lq2;
      issue_block<H>(lh2, lane, h2b_r, B.cb[j] * 32);
      issue_block<H>(lgx, lane, gxo_r, B.cb[j] * 32);
      issue_block<2 * H>(lq2, lane, q23_r, B.cb[j] * 32);
      f32x4 ph2[4], gxv[4], q2[4];
      finish_block(scr, lane, lh2, ph2);
      finish_block(scr, lane, lgx, gxv);
      finish_block(scr, lane, lq2, q2);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<f32x4*>(buf1 + mrow * LD + B.cb[j] * 32 + 8 * g + ch) = run4(acc[0][j], g) * dssilu4(ph2[g]);
        s_[j][g] = gxv[g] * onr * q2[g] * (kInvSqrt2 * inv_sqrt_h);
        pin4(s_[j][g]);
      }
    }
  __syncthreads();
  // ---- gxin = gh2 Wx0 (gx1 part | g|v2| part) block by block; the gx1 part accumulates onto the identity term gx
  f32x4 gnn[NP][4];                                   // g|v2| / |v2| per accumulator position
#pragma unroll 1
  for (int j = 0; j < NP; ++j) {
    const int cb = wave + 4 * j;
    if (cb >= CB) break;
    const f32x4* bpx[2] = {wx0t + (size_t)cb * frag_f4(H), wx0t + (size_t)(CB + cb) * frag_f4(H)};
    BRing<2, kRSW> rx;
    b_preload(rx, bpx);
    BlockLoad lgx, lnr;
    issue_block<H>(lgx, lane, gxo_r, cb * 32);
    issue_block<H>(lnr, lane, nrm_r, cb * 32);
    fence_sched();
    f32x4 gxv[4], nv[4];
    finish_block(scr, lane, lgx, gxv);
    finish_block(scr, lane, lnr, nv);
    f32x16 accx[1][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      set_run4(accx[0][0], g, gxv[g] * onr);
      set_run4(accx[0][1], g, (f32x4){0.f, 0.f, 0.f, 0.f});
    }
    mma_panel<H, LD, 1, 2, kRSW, false, true, false>(accx, buf1 + mrow * LD + ch, bpx, rx);
    f32x4 v[4], gn_[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      v[g] = run4(accx[0][0], g);
      const f32x4 n = nv[g];       // (rows past the tile's end load 0: their products are zero anyway)
      gn_[g] = run4(accx[0][1], g) * (f32x4){__builtin_amdgcn_rcpf(fmaxf(n[0], 1e-4f)), __builtin_amdgcn_rcpf(fmaxf(n[1], 1e-4f)),
                                             __builtin_amdgcn_rcpf(fmaxf(n[2], 1e-4f)), __builtin_amdgcn_rcpf(fmaxf(n[3], 1e-4f))};
    }
    store_block<H>(scr, lane, v, gx1_r, cb * 32);
    // gnn[j] with a runtime j: NP is small, select by comparison (keeps the array in registers)
#pragma unroll
    for (int jj = 0; jj < NP; ++jj)
      if (jj == j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gnn[jj][g] = gn_[g];
      }
  }
  // ---- gvec1[d] = gv[d] + (gv1[d] | gv2[d]) Wv,  gv1 = gv q3 + s v2,  gv2 = s v1 + gnn v2  (the accumulator starts at gv[d])
  BRing<NP, kRSW> rg;
#pragma unroll 1
  for (int d = 0; d < 3; ++d) {
    f32x16 accg[1][NP];
    zero_acc(accg);
    __syncthreads();                                 // the previous product has read both buffers
    b_preload(rg, bpg);
#pragma unroll
    for (int j = 0; j < NP; ++j)
      if (B.ok[j]) {
        BlockLoad lgv, lq3, lw1, lw2;
        issue_block<3 * H>(lgv, lane, gvo_r, d * H + B.cb[j] * 32);
        issue_block<2 * H>(lq3, lane, q23_r, H + B.cb[j] * 32);
        issue_block<6 * H>(lw1, lane, vp_r, d * 2 * H + B.cb[j] * 32);
        issue_block<6 * H>(lw2, lane, vp_r, d * 2 * H + H + B.cb[j] * 32);
        f32x4 gv[4], q3[4], v1[4], v2[4];
        finish_block(scr, lane, lgv, gv);
        finish_block(scr, lane, lq3, q3);
        finish_block(scr, lane, lw1, v1);
        finish_block(scr, lane, lw2, v2);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = B.cb[j] * 32 + 8 * g + ch;
          gv[g] *= onr;
          set_run4(accg[0][j], g, gv[g]);
          *reinterpret_cast<f32x4*>(buf0 + mrow * LD + c0) = gv[g] * q3[g] + s_[j][g] * v2[g];
          *reinterpret_cast<f32x4*>(buf1 + mrow * LD + c0) = s_[j][g] * v1[g] + gnn[j][g] * v2[g];
        }
      }
    __syncthreads();
    const f32x4* bpg1[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) bpg1[j] = bpg[j] + (size_t)frag_f4(H);
    mma_all<H, LD, NP, true>(accg, buf0 + mrow * LD + ch, bpg, rg);
    mma_all<H, LD, NP, false>(accg, buf1 + mrow * LD + ch, bpg1, rg);
#pragma unroll
    for (int j = 0; j < NP; ++j)
      if (B.ok[j]) {
        const f32x4 v[4] = {run4(accg[0][j], 0), run4(accg[0][j], 1), run4(accg[0][j], 2), run4(accg[0][j], 3)};
        store_block<3 * H>(scr, lane, v, gvec1_r, d * H + B.cb[j] * 32);
      }
  }
}

#define HN_WIDE_SWITCH(CALL)          \
  switch (hidden) {                   \
    case 128: CALL(128); break;       \
    case 192: CALL(192); break;       \
    case 256: CALL(256); break;       \
    case 320: CALL(320); break;       \
    case 384: CALL(384); break;       \
    case 448: CALL(448); break;       \
    case 512: CALL(512); break;       \
    default: return HN_ERR_BAD_ARG;   \
  }

}  // namespace

// Entry points of this translation unit (called by the extern "C" functions of node_chain.hip)
int hn_wide_pre_fwd(int hidden, const PreFwdArgs& a, void* stream) {
  const dim3 grid((unsigned)((a.Ns + 31) / 32), (unsigned)a.T);
  int rc = HN_ERR_BAD_ARG;
#define HN_CALL(HH) rc = launch_chain(node_pre_fwd_wide_kernel<HH>, grid, (size_t)(32 * (HH + 4) + 4 * kScrFloats) * 4, stream, a)
  HN_WIDE_SWITCH(HN_CALL)
#undef HN_CALL
  return rc;
}

int hn_wide_pre_bwd(int hidden, const PreBwdArgs& a, void* stream) {
  const dim3 grid((unsigned)((a.Ns + 31) / 32), (unsigned)a.T);
  int rc = HN_ERR_BAD_ARG;
#define HN_CALL(HH) rc = launch_chain(node_pre_bwd_wide_kernel<HH>, grid, (size_t)pre_bwd_wide_lds_floats(HH) * 4, stream, a)
  HN_WIDE_SWITCH(HN_CALL)
#undef HN_CALL
  return rc;
}

int hn_wide_update_fwd(int hidden, const UpdFwdArgs& a, int tiles, void* stream) {
  const dim3 grid((unsigned)tiles);
  int rc = HN_ERR_BAD_ARG;
#define HN_CALL(HH) rc = launch_chain(node_update_fwd_wide_kernel<HH>, grid, (size_t)(2 * 32 * (HH + 4) + 4 * kScrFloats) * 4, stream, a)
  HN_WIDE_SWITCH(HN_CALL)
#undef HN_CALL
  return rc;
}

int hn_wide_update_bwd(int hidden, const UpdBwdArgs& a, int tiles, void* stream) {
  const dim3 grid((unsigned)tiles);
  int rc = HN_ERR_BAD_ARG;
#define HN_CALL(HH) rc = launch_chain(node_update_bwd_wide_kernel<HH>, grid, (size_t)(2 * 32 * (HH + 4) + 4 * kScrFloats) * 4, stream, a)
  HN_WIDE_SWITCH(HN_CALL)
#undef HN_CALL
  return rc;
}
