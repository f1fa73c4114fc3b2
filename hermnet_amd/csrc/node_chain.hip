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
//   * C: 32 x 32 accumulator blocks, TRANSPOSED (the weight fragment is the MFMA's row operand, the activation its column
//     operand): a lane holds ONE tile row and 16 channels as four runs of 4 consecutive channels, so every epilogue
//     access to a row-major [rows][channels] array is a 16-byte load / store at `row offset + constant` (4 per block
//     instead of 16 dwords, one address register per array), LDS tiles are written with ds_write_b128, and per-row
//     quantities (row flags) are per-lane.  A wave owns the same 32-channel slices of every part of an output
//     (s|a|b, v1|v2, p|q|r), so products of parts (vec_dot, q * vdot, r * v1) are lane-local.
// Two workgroups per CU (<= 68 KB of LDS, <= 256 VGPRs): one's elementwise / staging phases run beside the other's
// matrix phases.  fp32 MFMA is exact fp32 (a k-ordered fmaf chain): results equal a library GEMM's to rounding.
#include "node_chain_common.h"
int hn_option(int option);      // host_api.cpp: process-wide options (hermnet_set_option)

namespace {
// =====================================================================================================================
// node_pre_fwd: one workgroup = (TR source rows, relation t)
// =====================================================================================================================

template <int H, int TR>
__global__ __launch_bounds__(256, 2) void node_pre_fwd_kernel(PreFwdArgs a) {
  using C = Cfg<H, TR>;
  constexpr int LD = C::LD, RB = C::RB, CPW = C::CPW, CB = C::CB;
  extern __shared__ __align__(16) float tile[];           // [TR][LD], then 4 x [32][36] store scratch
  stagger_start();
  const int t = blockIdx.y, row0 = blockIdx.x * TR;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % C::WC, wr = wave / C::WC;
  float* scr = tile + TR * LD + wave * kScrFloats;
  const int nrows = min(TR, a.Ns - row0);
  if (!tile_selected(a.windows, a.nwin, a.wmode, row0, TR)) return;
  if (!tile_wanted(a.src_ranges, t, row0, TR)) return;     // xh[t] / hb[t] of these rows are never read
  const rsrc_t x_r = tile_rsrc(a.x + (size_t)row0 * H, nrows * H);
  const rsrc_t hb_r = tile_rsrc(a.hb + ((size_t)t * a.Ns + row0) * H, nrows * H);
  const rsrc_t xh_r = tile_rsrc(a.xh + ((size_t)t * a.Ns + row0) * 3 * H, nrows * 3 * H);

  // weight streams of this wave (requested before anything else: they do not depend on the rows)
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.w1f + (size_t)t * H * H * 3 / 2) + lane;
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.w2f + (size_t)t * 3 * H * H * 3 / 2) + lane;
  const f32x4* bp1[CPW];
  const f32x4* bp2[3 * CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) bp1[j] = w1 + (size_t)(wc * CPW + j) * frag_f4(H);
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int j = 0; j < CPW; ++j) bp2[p * CPW + j] = w2 + (size_t)(p * CB + wc * CPW + j) * frag_f4(H);
  BRing<CPW, ring_size(RB * CPW, false)> r1;
  b_preload(r1, bp1);
  const int ch = 4 * (lane >> 5);
  f32x4 bias1[CPW][4];                                // b1 of this wave's channels: the first product starts from it
#pragma unroll
  for (int j = 0; j < CPW; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) bias1[j][g] = ld4g(a.b1 + (size_t)t * H + (wc * CPW + j) * 32 + 8 * g + ch);
  STAMP(0);
  STAMP_HWID();

  // ---- LayerNorm without affine (rmnet.py:52): TPR adjacent lanes share a row (its float4s dealt round-robin, so a
  // load instruction reads 16 TPR contiguous bytes per row); statistics over the first Hr channels, two passes in
  // registers, the cross-lane sums are two or three DPP steps
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
    // (every relation that wants the tile writes the same statistics: with source ranges relation 0 may skip it)
    if ((t == 0 || a.src_ranges != nullptr) && q == 0 && lr < nrows) { a.mean[row0 + lr] = mu; a.rstd[row0 + lr] = rs; }
  }
  STAMP(1);
  __syncthreads();
  STAMP(2);

  // ---- h = n W1^T
  const int mrow = wr * RB * 32 + (lane & 31);
  const float* As = tile + mrow * LD + ch;
  f32x16 acc1[RB][CPW];
  bias_acc(acc1, bias1);
  mma_panel<H, LD, RB, CPW, ring_size(RB * CPW, false), false, HN_PIN_PRE>(acc1, As, bp1, r1);
  STAMP(3);
  BRing<3 * CPW, ring_size(RB * 3 * CPW, false)> r2;
  b_preload(r2, bp2);
  f32x4 bias2[3 * CPW][4];                           // b2, requested in front of the epilogue's stores
#pragma unroll
  for (int jj = 0; jj < 3 * CPW; ++jj)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bias2[jj][g] = ld4g(a.b2 + (size_t)t * 3 * H + (jj / CPW) * H + (wc * CPW + jj % CPW) * 32 + 8 * g + ch);
  __syncthreads();                                   // every wave has read n
  STAMP(4);
  // ---- + b1, save, ScaledSiLU -> the tile becomes the A operand of the second product
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      f32x4 hv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int lr = mrow + rb * 32, c0 = (wc * CPW + j) * 32 + 8 * g + ch;
        hv[g] = run4(acc1[rb][j], g);                 // (incl. b1)
        *reinterpret_cast<f32x4*>(tile + lr * LD + c0) = ssilu4(hv[g]);
      }
      store_block<H, true>(scr, lane, hv, hb_r, (wr * RB + rb) * 32 * H + (wc * CPW + j) * 32);       // (saved for the backward only)
    }
  STAMP(5);
  __syncthreads();
  STAMP(6);
  // ---- xh = a W2^T + b2
  f32x16 acc2[RB][3 * CPW];
  bias_acc(acc2, bias2);
  mma_panel<H, LD, RB, 3 * CPW, ring_size(RB * 3 * CPW, false), false, HN_PIN_PRE>(acc2, As, bp2, r2);
  STAMP(7);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int jj = 0; jj < 3 * CPW; ++jj) {
      const int cblk = (jj / CPW) * H + (wc * CPW + jj % CPW) * 32;
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) v[g] = run4(acc2[rb][jj], g);            // (incl. b2)
      store_block<3 * H>(scr, lane, v, xh_r, (wr * RB + rb) * 32 * 3 * H + cblk);
    }
  STAMP(8);
}

// =====================================================================================================================
// node_pre_bwd: one workgroup = (TR source rows, relation t) -> gn[t] (the sum over t and the LayerNorm backward follow
// in layernorm_bwd_parts_kernel)
// =====================================================================================================================
// LDS floats of node_pre_bwd: two K-chunk buffers, later the gh tile + the store scratch in the same space
constexpr int pre_bwd_lds_floats(int H, int TR) {
  const int kc = H < 128 ? H : 128, chunks = 2 * TR * (kc + 4), tail = TR * (H + 4) + 4 * kScrFloats;
  return chunks > tail ? chunks : tail;
}


template <int H, int TR>
__global__ __launch_bounds__(256, 2) void node_pre_bwd_kernel(PreBwdArgs a) {
  using C = Cfg<H, TR>;
  constexpr int LD = C::LD, RB = C::RB, CPW = C::CPW;
  constexpr int KC = H < 128 ? H : 128;            // K chunk of the first product (K = 3H)
  constexpr int NCH = 3 * H / KC, LDC = KC + 4;
  static_assert(2 * TR * LDC >= TR * LD, "the gh tile must fit the two chunk buffers");   // (+ scratch: pre_bwd_lds_floats)
  extern __shared__ __align__(16) float lds[];            // 2 x [TR][LDC], then 4 x [32][36] store scratch
  stagger_start();
  const int t = blockIdx.y, row0 = blockIdx.x * TR;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % C::WC, wr = wave / C::WC;
  float* scr = lds + TR * LD + wave * kScrFloats;     // the chunk buffers' upper part: free once the gh tile is written
  const int nrows = min(TR, a.Ns - row0);
  const int mrow = wr * RB * 32 + (lane & 31), ch = 4 * (lane >> 5);
  if (!tile_selected(a.windows, a.nwin, a.wmode, row0, TR)) return;
  if (!tile_wanted(a.src_ranges, t, row0, TR)) {
    float* g0 = a.gn + ((size_t)t * a.Ns + row0) * H;
    for (int i = tid; i < nrows * (H / 4); i += 256) reinterpret_cast<f32x4*>(g0)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    return;
  }

  const f32x4* w2t = reinterpret_cast<const f32x4*>(a.w2tf + (size_t)t * 3 * H * H * 3 / 2) + lane;
  const f32x4* w1t = reinterpret_cast<const f32x4*>(a.w1tf + (size_t)t * H * H * 3 / 2) + lane;
  const f32x4* bpa[CPW];
  const f32x4* bpb[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    bpa[j] = w2t + (size_t)(wc * CPW + j) * frag_f4(3 * H);
    bpb[j] = w1t + (size_t)(wc * CPW + j) * frag_f4(H);
  }
  BRing<CPW, ring_size(RB * CPW, true)> ra;
  b_preload(ra, bpa);
  const rsrc_t gxh_r = tile_rsrc(a.gxh + ((size_t)t * a.Ns + row0) * 3 * H, nrows * 3 * H);
  const rsrc_t hb_r = tile_rsrc(a.hb + ((size_t)t * a.Ns + row0) * H, nrows * H);
  const rsrc_t gn_r = tile_rsrc(a.gn + ((size_t)t * a.Ns + row0) * H, nrows * H);
  TileRegs<TR, KC> regs;
  tile_load<TR, KC>(regs, gxh_r, 3 * H, 0, tid);
  // hb of this wave's accumulator blocks, requested now (coalesced), transposed behind the first product
  BlockLoad lhb[RB][CPW];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) issue_block<H, true>(lhb[rb][j], lane, hb_r, (wr * RB + rb) * 32 * H + (wc * CPW + j) * 32);

  // ---- ga = gxh W2   (K = 3H in chunks, double-buffered through registers)
  f32x16 acc[RB][CPW];
  zero_acc(acc);
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) {
    float* buf = lds + (kc & 1) * TR * LDC;
    tile_store<TR, KC, LDC>(buf, regs, tid);
    __syncthreads();
    if (kc + 1 < NCH) tile_load<TR, KC>(regs, gxh_r, 3 * H, (kc + 1) * KC, tid);
    const float* As = buf + mrow * LDC + ch;
    const f32x4* bpk[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpk[j] = bpa[j] + (size_t)kc * frag_f4(KC);
    if (kc + 1 < NCH) mma_panel<KC, LDC, RB, CPW, ring_size(RB * CPW, true), true>(acc, As, bpk, ra);
    else mma_panel<KC, LDC, RB, CPW, ring_size(RB * CPW, true), false>(acc, As, bpk, ra);
  }
  BRing<CPW, ring_size(RB * CPW, false)> rb_;
  b_preload(rb_, bpb);
  __syncthreads();                                   // the chunk buffers are free
  // ---- gh = ga * ScaledSiLU'(hb) -> tile
  float* tile = lds;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      f32x4 hbv[4];
      finish_block(scr, lane, lhb[rb][j], hbv);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(tile + (mrow + rb * 32) * LD + (wc * CPW + j) * 32 + 8 * g + ch) =
            run4(acc[rb][j], g) * dssilu4(hbv[g]);
    }
  __syncthreads();
  // ---- gn_t = gh W1
  f32x16 acc2[RB][CPW];
  zero_acc(acc2);
  mma_panel<H, LD, RB, CPW, ring_size(RB * CPW, false), false>(acc2, tile + mrow * LD + ch, bpb, rb_);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const f32x4 v[4] = {run4(acc2[rb][j], 0), run4(acc2[rb][j], 1), run4(acc2[rb][j], 2), run4(acc2[rb][j], 3)};
      store_block<H>(scr, lane, v, gn_r, (wr * RB + rb) * 32 * H + (wc * CPW + j) * 32);
    }
}


// =====================================================================================================================
// node_update_fwd (rmnet.py:94-107 + the residual of rmnet.py:29-31 + the zero rows of hermnet.py:51,56-57)
// =====================================================================================================================

template <int H, int TR, int MINW>
__global__ __launch_bounds__(256, MINW) void node_update_fwd_kernel(UpdFwdArgs a) {
  using C = Cfg<H, TR>;
  constexpr int LD = C::LD, RB = C::RB, CPW = C::CPW, CB = C::CB;
  extern __shared__ __align__(16) float lds[];            // 2 x [TR][LD], then 4 x [32][36] store scratch
  stagger_start();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % C::WC, wr = wave / C::WC;
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
  const f32x4* wv = reinterpret_cast<const f32x4*>(a.wvf + (size_t)t * 2 * H * H * 3 / 2) + lane;
  const f32x4* wx0 = reinterpret_cast<const f32x4*>(a.wx0f + (size_t)t * 2 * H * H * 3 / 2) + lane;
  const f32x4* wx2 = reinterpret_cast<const f32x4*>(a.wx2f + (size_t)t * 3 * H * H * 3 / 2) + lane;
  const f32x4* bpv[2 * CPW];
  const f32x4* bpx[CPW];
  const f32x4* bpq[3 * CPW];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpv[p * CPW + j] = wv + (size_t)(p * CB + wc * CPW + j) * frag_f4(H);
#pragma unroll
  for (int j = 0; j < CPW; ++j) bpx[j] = wx0 + (size_t)(wc * CPW + j) * frag_f4(2 * H);
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpq[p * CPW + j] = wx2 + (size_t)(p * CB + wc * CPW + j) * frag_f4(H);

  const int mrow = wr * RB * 32 + (lane & 31), ch = 4 * (lane >> 5);
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
  f32x4 dot[RB][CPW][4], sq[RB][CPW][4], kv1[RB][CPW][4][3];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) dot[rb][j][g] = sq[rb][j][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- vp[d] = vec1[d] Wv^T for the three Cartesian components; vec_dot and |v2|^2 accumulate in registers
  // (every weight ring is requested BEFORE the stores of the epilogue in front of its product: vmcnt retires in order,
  // so a fragment requested behind a store would make the product's first MFMA wait for the store to drain)
  STAMP(0);
  STAMP_HWID();
  TileRegs<TR, H> regs;
  tile_load<TR, H>(regs, vec1_r, 3 * H, 0, tid);
  BRing<2 * CPW, ring_size(RB * 2 * CPW, false)> rv;
  BRing<CPW, ring_size(RB * CPW, true)> rx;
  b_preload(rv, bpv);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float* buf = lds + (d & 1) * TR * LD;
    tile_store<TR, H, LD>(buf, regs, tid);
    __syncthreads();
    STAMP(1 + 3 * d);
    if (d < 2) tile_load<TR, H>(regs, vec1_r, 3 * H, (d + 1) * H, tid);
    else tile_load<TR, H>(regs, x1_r, H, 0, tid);
    f32x16 accv[RB][2 * CPW];
    zero_acc(accv);
    mma_panel<H, LD, RB, 2 * CPW, ring_size(RB * 2 * CPW, false), false, true, false>(accv, buf + mrow * LD + ch, bpv, rv);
    STAMP(2 + 3 * d);
    if (d < 2) b_preload(rv, bpv);
    else b_preload(rx, bpx);
    fence_sched();
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        f32x4 v1[4], v2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          v1[g] = run4(accv[rb][j], g);
          v2[g] = run4(accv[rb][CPW + j], g);
          kv1[rb][j][g][d] = v1[g];                  // stays in registers for dvec = r v1
          dot[rb][j][g] += v1[g] * v2[g];
          sq[rb][j][g] += v2[g] * v2[g];
          pin4(dot[rb][j][g]);
          pin4(sq[rb][j][g]);
        }
        const int ob = ((wr * RB + rb) * 32 * 3 + d) * 2 * H + (wc * CPW + j) * 32;
        store_block<6 * H>(scr, lane, v1, vp_r, ob);
        store_block<6 * H, true>(scr, lane, v2, vp_r, ob + H);
      }
    STAMP(3 + 3 * d);
  }
  // ---- xin = [x1 | sqrt(|v2|^2 + 1e-8)]: x1 -> buffer 1 (free since the product of d = 1), the norm -> buffer 0
  float* bufx = lds + TR * LD;
  float* bufn = lds;
  tile_store<TR, H, LD>(bufx, regs, tid);
  __syncthreads();                                   // every wave has finished the product of d = 2 (buffer 0)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      f32x4 nv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 s2 = sq[rb][j][g] + 1e-8f;
        nv[g] = (f32x4){sqrtf(s2[0]), sqrtf(s2[1]), sqrtf(s2[2]), sqrtf(s2[3])};
        *reinterpret_cast<f32x4*>(bufn + (mrow + rb * 32) * LD + (wc * CPW + j) * 32 + 8 * g + ch) = nv[g];
      }
      store_block<H, true>(scr, lane, nv, nrm_r, (wr * RB + rb) * 32 * H + (wc * CPW + j) * 32);
    }
  __syncthreads();
  STAMP(10);
  // ---- h2 = xin Wx0^T + bx0 (K = 2H: two panels)
  f32x16 acch[RB][CPW];
  zero_acc(acch);
  {
    const f32x4* bpx1[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpx1[j] = bpx[j] + (size_t)frag_f4(H);
    mma_panel<H, LD, RB, CPW, ring_size(RB * CPW, true), true, true, false>(acch, bufx + mrow * LD + ch, bpx, rx);
    mma_panel<H, LD, RB, CPW, ring_size(RB * CPW, true), false, true, false>(acch, bufn + mrow * LD + ch, bpx1, rx);
  }
  STAMP(11);
  BRing<3 * CPW, ring_size(RB * 3 * CPW, false)> rq;
  b_preload(rq, bpq);
  __syncthreads();                                   // both buffers are free (x1 is re-read from memory below)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      f32x4 hv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int lr = mrow + rb * 32, c0 = (wc * CPW + j) * 32 + 8 * g + ch;
        hv[g] = run4(acch[rb][j], g) + ld4g(a.bx0 + (size_t)t * H + c0);
        *reinterpret_cast<f32x4*>(lds + lr * LD + c0) = ssilu4(hv[g]);
      }
      store_block<H, true>(scr, lane, hv, h2b_r, (wr * RB + rb) * 32 * H + (wc * CPW + j) * 32);
    }
  __syncthreads();
  // ---- (p | q | r) = a2 Wx2^T + bx2, then the update and the residual.  Of the epilogue's inputs, v1 is still in
  // registers, x1 still in buffer 1; vec1 is requested BEFORE the product (coalesced) and transposed behind it.
  BlockLoad lvv[RB][CPW][3];
  float pon[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    pon[rb] = bld(act_r, mrow + rb * 32);
#pragma unroll
    for (int j = 0; j < CPW; ++j)
#pragma unroll
      for (int d = 0; d < 3; ++d)
        issue_block<3 * H>(lvv[rb][j][d], lane, vec1_r, ((wr * RB + rb) * 32 * 3 + d) * H + (wc * CPW + j) * 32);
  }
  fence_sched();
  STAMP(12);
  f32x16 accq[RB][3 * CPW];
  zero_acc(accq);
  mma_panel<H, LD, RB, 3 * CPW, ring_size(RB * 3 * CPW, false), false, true, false>(accq, lds + mrow * LD + ch, bpq, rq);
  STAMP(13);
  fence_sched();
  const float inv_sqrt_h = rsqrtf((float)H);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const float on = (all_on | (pon[rb] != 0.f)) ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      f32x4 q[4], r[4], xo[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = (wc * CPW + j) * 32 + 8 * g + ch;
        const float* bb = a.bx2 + (size_t)t * 3 * H + c0;
        const f32x4 p = run4(accq[rb][j], g) + ld4g(bb);
        q[g] = run4(accq[rb][CPW + j], g) + ld4g(bb + H);
        r[g] = run4(accq[rb][2 * CPW + j], g) + ld4g(bb + 2 * H);
        const f32x4 x1v = *reinterpret_cast<const f32x4*>(bufx + (mrow + rb * 32) * LD + c0);
        xo[g] = (x1v + (p + q[g] * dot[rb][j][g] * inv_sqrt_h) * kInvSqrt2) * on;
      }
      const int rblk = (wr * RB + rb) * 32, cblk = (wc * CPW + j) * 32;
      store_block<2 * H, true>(scr, lane, q, q23_r, rblk * 2 * H + cblk);
      store_block<2 * H, true>(scr, lane, r, q23_r, rblk * 2 * H + H + cblk);
      store_block<H>(scr, lane, xo, xo_r, rblk * H + cblk);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        f32x4 vo[4];
        finish_block(scr, lane, lvv[rb][j][d], vo);          // vec1[d] of this lane's row
#pragma unroll
        for (int g = 0; g < 4; ++g) vo[g] = (vo[g] + r[g] * kv1[rb][j][g][d]) * on;
        store_block<3 * H>(scr, lane, vo, vo_r, (rblk * 3 + d) * H + cblk);
      }
    }
  }
  STAMP(14);
}

// =====================================================================================================================
// node_update_bwd: (gx_out, gvec_out) -> (gx1, gvec1), parameters are constants
// =====================================================================================================================

template <int H, int TR, int MINW>
__global__ __launch_bounds__(256, MINW) void node_update_bwd_kernel(UpdBwdArgs a) {
  using C = Cfg<H, TR>;
  constexpr int LD = C::LD, RB = C::RB, CPW = C::CPW, CB = C::CB, F4 = C::F4, V = H / 4;
  extern __shared__ __align__(16) float lds[];            // 2 x [TR][LD], then 4 x [32][36] store scratch
  stagger_start();
  float* buf0 = lds;
  float* buf1 = lds + TR * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % C::WC, wr = wave / C::WC;
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
  const f32x4* wx2t = reinterpret_cast<const f32x4*>(a.wx2tf + (size_t)t * 3 * H * H * 3 / 2) + lane;
  const f32x4* wx0t = reinterpret_cast<const f32x4*>(a.wx0tf + (size_t)t * 2 * H * H * 3 / 2) + lane;
  const f32x4* wvt = reinterpret_cast<const f32x4*>(a.wvtf + (size_t)t * 2 * H * H * 3 / 2) + lane;
  const f32x4* bpa[CPW];
  const f32x4* bpx[2 * CPW];
  const f32x4* bpg[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    bpa[j] = wx2t + (size_t)(wc * CPW + j) * frag_f4(3 * H);
    bpg[j] = wvt + (size_t)(wc * CPW + j) * frag_f4(2 * H);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpx[p * CPW + j] = wx0t + (size_t)(p * CB + wc * CPW + j) * frag_f4(H);
  BRing<CPW, ring_size(RB * CPW, true)> ra;
  b_preload(ra, bpa);
  const int mrow = wr * RB * 32 + (lane & 31), ch = 4 * (lane >> 5);
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
  TileRegs<TR, H> g3;
#pragma unroll
  for (int it = 0; it < F4; ++it) {
    if ((it & 1) == 0) fence_sched();     // two positions (20 float4 loads) in flight, not all F4
    const int idx = tid + it * 256, lr = idx / V, c = (idx % V) * 4;
    // (rows past the tile's end load zeros; inactive rows are multiplied by zero: their saved values are finite)
    const float on = (all_on | (bld(act_r, lr) != 0.f)) ? 1.f : 0.f;
    const f32x4 gx = bld4(gxo_r, lr * H + c) * on;
    f32x4 vd = {0.f, 0.f, 0.f, 0.f}, gq3 = vd;
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
  // h2b and gx of this wave's accumulator blocks: requested now (coalesced), transposed behind the first product
  BlockLoad lh2[RB][CPW], lgx[RB][CPW], lq2[RB][CPW];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int ob = (wr * RB + rb) * 32 * H + (wc * CPW + j) * 32;
      issue_block<H>(lh2[rb][j], lane, h2b_r, ob);
      issue_block<H>(lgx[rb][j], lane, gxo_r, ob);
      issue_block<2 * H>(lq2[rb][j], lane, q23_r, (wr * RB + rb) * 32 * 2 * H + (wc * CPW + j) * 32);
    }
  __syncthreads();
  // ---- ga2 = gq Wx2  (K = 3H: three panels)
  f32x16 acc[RB][CPW];
  zero_acc(acc);
  {
    const f32x4* bp1[CPW];
    const f32x4* bp2[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) { bp1[j] = bpa[j] + (size_t)frag_f4(H); bp2[j] = bpa[j] + (size_t)frag_f4(2 * H); }
    mma_panel<H, LD, RB, CPW, ring_size(RB * CPW, true), true, true, false>(acc, buf0 + mrow * LD + ch, bpa, ra);
    __syncthreads();                                 // buffer 0 is free
    tile_store<TR, H, LD>(buf0, g3, tid);
    mma_panel<H, LD, RB, CPW, ring_size(RB * CPW, true), true, true, false>(acc, buf1 + mrow * LD + ch, bp1, ra);
    __syncthreads();                                 // third part in place, buffer 1 free
    mma_panel<H, LD, RB, CPW, ring_size(RB * CPW, true), false, true, false>(acc, buf0 + mrow * LD + ch, bp2, ra);
  }
  BRing<2 * CPW, ring_size(RB * 2 * CPW, false)> rx;
  b_preload(rx, bpx);
  fence_sched();
  // ---- gh2 = ga2 * ScaledSiLU'(h2b) -> buffer 1
  f32x16 accx[RB][2 * CPW];
  f32x4 s_[RB][CPW][4];            // s = gvdot / sqrt(H) = gx q / sqrt(2H) per accumulator position
  float onr[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    onr[rb] = (all_on | (bld(act_r, mrow + rb * 32) != 0.f)) ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      f32x4 ph2[4], gxv[4], q2[4];
      finish_block(scr, lane, lh2[rb][j], ph2);
      finish_block(scr, lane, lgx[rb][j], gxv);
      finish_block(scr, lane, lq2[rb][j], q2);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<f32x4*>(buf1 + (mrow + rb * 32) * LD + (wc * CPW + j) * 32 + 8 * g + ch) =
            run4(acc[rb][j], g) * dssilu4(ph2[g]);
        gxv[g] *= onr[rb];
        s_[rb][j][g] = gxv[g] * q2[g] * (kInvSqrt2 * inv_sqrt_h);
        pin4(s_[rb][j][g]);
        // gxin = gh2 Wx0 (gx1 part | g|v2| part): the gx1 part accumulates onto the identity term gx
        set_run4(accx[rb][j], g, gxv[g]);
        set_run4(accx[rb][CPW + j], g, (f32x4){0.f, 0.f, 0.f, 0.f});
      }
    }
  }
  fence_sched();
  // Everything the next epilogues read from memory is requested one product ahead of its use (coalesced), and
  // transposed into this lane's row behind that product.
  BlockLoad lnr[RB][CPW], lgv[RB][CPW], lq3[RB][CPW], lw1[RB][CPW], lw2[RB][CPW];
// inputs of the staging of component d: gv, (q3), v1, v2
#define HN_REQUEST(d)                                                                                         \
  _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) _Pragma("unroll") for (int j = 0; j < CPW; ++j) {          \
    const int rblk = (wr * RB + rb) * 32, cblk = (wc * CPW + j) * 32;                                          \
    issue_block<3 * H>(lgv[rb][j], lane, gvo_r, (rblk * 3 + (d)) * H + cblk);                                  \
    if ((d) == 0) issue_block<2 * H>(lq3[rb][j], lane, q23_r, rblk * 2 * H + H + cblk);                        \
    issue_block<6 * H>(lw1[rb][j], lane, vp_r, (rblk * 3 + (d)) * 2 * H + cblk);                               \
    issue_block<6 * H>(lw2[rb][j], lane, vp_r, (rblk * 3 + (d)) * 2 * H + H + cblk);                           \
  }
  f32x4 pgv[RB][CPW][4], pq3[RB][CPW][4], pw1[RB][CPW][4], pw2[RB][CPW][4];
#define HN_ARRIVE(d)                                                                                          \
  _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) _Pragma("unroll") for (int j = 0; j < CPW; ++j) {          \
    finish_block(scr, lane, lgv[rb][j], pgv[rb][j]);                                                           \
    if ((d) == 0) finish_block(scr, lane, lq3[rb][j], pq3[rb][j]);                                             \
    finish_block(scr, lane, lw1[rb][j], pw1[rb][j]);                                                           \
    finish_block(scr, lane, lw2[rb][j], pw2[rb][j]);                                                           \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) pgv[rb][j][g] *= onr[rb];                                    \
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int rblk = (wr * RB + rb) * 32, cblk = (wc * CPW + j) * 32;
      issue_block<H>(lnr[rb][j], lane, nrm_r, rblk * H + cblk);
    }
  HN_REQUEST(0)
  fence_sched();
  __syncthreads();
  mma_panel<H, LD, RB, 2 * CPW, ring_size(RB * 2 * CPW, false), false, true, false>(accx, buf1 + mrow * LD + ch, bpx, rx);
  // (weight rings are requested BEFORE the stores of the epilogue in front of their product: vmcnt retires in order)
  BRing<CPW, ring_size(RB * CPW, true)> rg;
  b_preload(rg, bpg);
  fence_sched();
  f32x4 pnrm[RB][CPW][4];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      finish_block(scr, lane, lnr[rb][j], pnrm[rb][j]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 nv = pnrm[rb][j][g];       // (rows past the tile's end load 0: their products are zero anyway)
        pnrm[rb][j][g] = (f32x4){__builtin_amdgcn_rcpf(fmaxf(nv[0], 1e-4f)), __builtin_amdgcn_rcpf(fmaxf(nv[1], 1e-4f)),
                                 __builtin_amdgcn_rcpf(fmaxf(nv[2], 1e-4f)), __builtin_amdgcn_rcpf(fmaxf(nv[3], 1e-4f))};
      }
    }
  HN_ARRIVE(0)
  // per accumulator position: gnn = g|v2| / |v2|
  f32x4 gnn[RB][CPW][4];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[g] = run4(accx[rb][j], g);
        gnn[rb][j][g] = run4(accx[rb][CPW + j], g) * pnrm[rb][j][g];
        pin4(gnn[rb][j][g]);
      }
      store_block<H>(scr, lane, v, gx1_r, (wr * RB + rb) * 32 * H + (wc * CPW + j) * 32);
    }
  // ---- gvec1[d] = gv[d] + (gv1[d] | gv2[d]) Wv,  gv1 = gv q3 + s v2,  gv2 = s v1 + gnn v2  (the accumulator starts at gv[d])
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    f32x16 accg[RB][CPW];
    __syncthreads();                                 // the previous product has read both buffers
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int j = 0; j < CPW; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int lr = mrow + rb * 32, c0 = (wc * CPW + j) * 32 + 8 * g + ch;
          const f32x4 gv = pgv[rb][j][g], v1 = pw1[rb][j][g], v2 = pw2[rb][j][g];
          set_run4(accg[rb][j], g, gv);
          *reinterpret_cast<f32x4*>(buf0 + lr * LD + c0) = gv * pq3[rb][j][g] + s_[rb][j][g] * v2;
          *reinterpret_cast<f32x4*>(buf1 + lr * LD + c0) = s_[rb][j][g] * v1 + gnn[rb][j][g] * v2;
        }
    fence_sched();
    if (d < 2) { HN_REQUEST(d + 1) }                 // in flight during the product below
    fence_sched();
    __syncthreads();
    const f32x4* bpg1[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) bpg1[j] = bpg[j] + (size_t)frag_f4(H);
    mma_panel<H, LD, RB, CPW, ring_size(RB * CPW, true), true, true, false>(accg, buf0 + mrow * LD + ch, bpg, rg);
    mma_panel<H, LD, RB, CPW, ring_size(RB * CPW, true), false, true, false>(accg, buf1 + mrow * LD + ch, bpg1, rg);
    if (d < 2) b_preload(rg, bpg);
    fence_sched();
    if (d < 2) { HN_ARRIVE(d + 1) }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        const f32x4 v[4] = {run4(accg[rb][j], 0), run4(accg[rb][j], 1), run4(accg[rb][j], 2), run4(accg[rb][j], 3)};
        store_block<3 * H>(scr, lane, v, gvec1_r, ((wr * RB + rb) * 32 * 3 + d) * H + (wc * CPW + j) * 32);
      }
  }
}

// LDS floats of a kernel family for tile (H, TR): NB buffers of [TR][H + 4]
#define HN_CHAIN_DISPATCH(KERNEL, GRID, NB, ARGS)                                                          \
  switch (hidden) {                                                                                        \
    case 64: return launch_chain(KERNEL<64, 64>, GRID(64), (size_t)(NB * 64 * 68 + 4 * kScrFloats) * 4, stream, ARGS);        \
    case 128: return launch_chain(KERNEL<128, 64>, GRID(64), (size_t)(NB * 64 * 132 + 4 * kScrFloats) * 4, stream, ARGS);     \
    case 256: return launch_chain(KERNEL<256, 32>, GRID(32), (size_t)(NB * 32 * 260 + 4 * kScrFloats) * 4, stream, ARGS);     \
    default: return HN_ERR_BAD_ARG;                                                                        \
  }

// Update kernels: tile (H, TR) and register budget.  H = 128 has two instances: 32-row tiles fit 256 registers, so two
// workgroups share a CU and one's elementwise phases run beside the other's matrix phases (the default: measured
// 64 / 77 us against 75 / 99 us per launch at 10,000 rows); 64-row tiles keep ~400 values per lane live (one wave per SIMD,
// 512 registers) and load every weight fragment once per 64 rows (option HN_OPT_UPDATE_TILE64_MAX = largest grid, in 64-row
// tiles, that takes them).
inline int tile64_max() { return hn_option(HN_OPT_UPDATE_TILE64_MAX); }
#define HN_TILES(TR) tiles_of(type_rowptr_host, num_rel, num_nodes, TR)
#define HN_UPDATE_DISPATCH(KERNEL, ARGS)                                                                              \
  switch (hidden) {                                                                                                   \
    case 64: return launch_chain(KERNEL<64, 64, 2>, dim3((unsigned)HN_TILES(64)), (size_t)(2 * 64 * 68 + 4 * kScrFloats) * 4, stream, ARGS); \
    case 128:                                                                                                         \
      if (HN_TILES(64) <= tile64_max())                                                                               \
        return launch_chain(KERNEL<128, 64, 1>, dim3((unsigned)HN_TILES(64)), (size_t)(2 * 64 * 132 + 4 * kScrFloats) * 4, stream, ARGS); \
      return launch_chain(KERNEL<128, 32, 2>, dim3((unsigned)HN_TILES(32)), (size_t)(2 * 32 * 132 + 4 * kScrFloats) * 4, stream, ARGS);  \
    case 256: return launch_chain(KERNEL<256, 32, 1>, dim3((unsigned)HN_TILES(32)), (size_t)(2 * 32 * 260 + 4 * kScrFloats) * 4, stream, ARGS); \
    default: return HN_ERR_BAD_ARG;                                                                                   \
  }

}  // namespace

#ifdef HN_STAMPS
extern "C" int hermnet_debug_stamps(unsigned long long* out_host, int count) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(hn_stamps), (size_t)count * sizeof(unsigned long long)) == hipSuccess ? 0 : 3;
}
#endif

// node_chain_wide.hip: the same four chains for every multiple of 64 from 128 to 512
int hn_wide_pre_fwd(int hidden, const PreFwdArgs& a, void* stream);
int hn_wide_pre_bwd(int hidden, const PreBwdArgs& a, void* stream);
int hn_wide_update_fwd(int hidden, const UpdFwdArgs& a, int tiles, void* stream);
int hn_wide_update_bwd(int hidden, const UpdBwdArgs& a, int tiles, void* stream);

// node_chain16.hip: the update chain on 16-row tiles (v_mfma_f32_16x16x4_f32, frag16 weight copies)
int hn_update16_supported(int hidden);
int hn_update16_fwd(int hidden, const UpdFwdArgs& a, int tiles, void* stream);
int hn_update16_bwd(int hidden, const UpdBwdArgs& a, int tiles, void* stream);
int hn_update16_pre_fwd(int hidden, const UpdFwdArgs& a, const PreFwdArgs& p, int tiles, void* stream);
int hn_pre16_fwd(int hidden, const PreFwdArgs& p, void* stream);
int hn_pre16_bwd(int hidden, const PreBwdArgs& a, void* stream);
int hn_head16_supported(int hidden, int cols);
int hn_head16_fwd(const float* x, const float* w0_frag16, const float* b0, const float* w2, const float* b2, const float* mask,
                  float* h, float* e, int rows, void* stream);
int hn_head16_bwd(const float* ge, const float* h, const float* w0t_frag16, const float* w2, const float* mask, float* gx,
                  int rows, void* stream);

// Widths 64 / 128 / 256 have tuned instances in this file; every other multiple of 64 up to 512 -- the reference's default
// hidden_channels = 512 among them (hermnet.py:86) -- takes the panelled kernels.  Option HN_OPT_NODE_CHAIN_WIDE = 1 sends 128
// and 256 there too (tests, comparisons).
static bool use_wide(int hidden) {
  const int force = hn_option(HN_OPT_NODE_CHAIN_WIDE);
  if (hidden == 64) return false;
  return force != 0 || !(hidden == 128 || hidden == 256);
}
// rows per tile of the pre kernels (the row windows of the halo overlap are cut at these boundaries)
static int pre_tile_rows(int hidden) { return use_wide(hidden) ? 32 : (hidden == 256 ? 32 : 64); }

// 16 or 32: the tile height the update kernels should run this grid at.  16-row tiles cost twice the weight bytes from L2,
// so they are chosen only where they shorten the launch: the busiest CU's share (whole tiles) is at least ~15 % smaller.
extern "C" int hermnet_node_update_tile_rows(const int* type_rowptr_host, int num_nodes, int num_rel, int hidden) {
  const int force = hn_option(HN_OPT_UPDATE_TILE16);      // 0 never, 1 always, 2 auto
  if (!type_rowptr_host || !hermnet_node_chain_supported(hidden)) return 0;
  const int base = use_wide(hidden) ? 32 : (hidden == 64 ? 64 : 32);
  if (!hn_update16_supported(hidden) || use_wide(hidden) || force == 0) return base;
  if (force == 1) return 16;
  int cus = 256, dev = 0, v = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  const long t32 = tiles_of(type_rowptr_host, num_rel, num_nodes, 32), t16 = tiles_of(type_rowptr_host, num_rel, num_nodes, 16);
  const long m32 = 2 * ((t32 + cus - 1) / cus), m16 = (t16 + cus - 1) / cus;      // busiest CU, in 16-row units
  return (m16 * 100 <= m32 * 85) ? 16 : base;
}

extern "C" int hermnet_node_chain_supported(int hidden) { return hidden >= 64 && hidden <= 512 && hidden % 64 == 0; }
extern "C" int hermnet_node_chain_tile_rows(int hidden, int update) {
  if (!hermnet_node_chain_supported(hidden)) return 0;
  if (update) return use_wide(hidden) ? 32 : (hidden == 64 ? 64 : 32);
  return pre_tile_rows(hidden);
}

extern "C" int hermnet_node_pre_fwd(const float* x, const float* w1_frag, const float* b1, const float* w2_frag,
                                    const float* b2, float* hb, float* xh, float* mean, float* rstd,
                                    const int* src_ranges, int num_src, int num_rel, int hidden, int hidden_real,
                                    float eps, const int* row_windows, int num_windows, int window_mode, void* stream) {
  if (num_src < 0 || num_rel <= 0 || hidden_real > hidden) return HN_ERR_BAD_ARG;
  if (window_mode < 0 || window_mode > 2 || (window_mode && (!row_windows || num_windows < 0))) return HN_ERR_BAD_ARG;
  if (!hermnet_node_chain_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_src == 0) return HN_OK;
  if (!x || !w1_frag || !b1 || !w2_frag || !b2 || !hb || !xh || !mean || !rstd) return HN_ERR_BAD_ARG;
  PreFwdArgs a = {x, w1_frag, b1, w2_frag, b2, hb, xh, mean, rstd, src_ranges, num_src, num_rel, hidden_real > 0 ? hidden_real : hidden, eps,
                  row_windows, num_windows, window_mode};
  if (use_wide(hidden)) return hn_wide_pre_fwd(hidden, a, stream);
#define HN_GRID(TR) dim3((unsigned)((num_src + TR - 1) / TR), (unsigned)num_rel)
  HN_CHAIN_DISPATCH(node_pre_fwd_kernel, HN_GRID, 1, a);
}

extern "C" int hermnet_node_pre_bwd(const float* gxh, const float* hb, const float* w2t_frag, const float* w1t_frag,
                                    float* gn_parts, const float* x, const float* mean, const float* rstd,
                                    const float* add, float* gx, const int* src_ranges, int num_src, int num_rel,
                                    int hidden, int hidden_real, const int* row_windows, int num_windows,
                                    int window_mode, void* stream) {
  if (num_src < 0 || num_rel <= 0 || hidden_real > hidden) return HN_ERR_BAD_ARG;
  if (window_mode < 0 || window_mode > 2 || (window_mode && (!row_windows || num_windows < 0))) return HN_ERR_BAD_ARG;
  if (!hermnet_node_chain_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_src == 0) return HN_OK;
  if (!gxh || !hb || !w2t_frag || !w1t_frag || !gn_parts || (gx && (!x || !mean || !rstd))) return HN_ERR_BAD_ARG;
  if (!gx && window_mode) return HN_ERR_BAD_ARG;
  PreBwdArgs a = {gxh, hb, w2t_frag, w1t_frag, gn_parts, src_ranges, num_src, num_rel, row_windows, num_windows, window_mode};
  // chunk buffers: 2 x [TR][min(H,128) + 4]
  int rc;
  if (use_wide(hidden)) rc = hn_wide_pre_bwd(hidden, a, stream);
  else switch (hidden) {
    case 64: rc = launch_chain(node_pre_bwd_kernel<64, 64>, HN_GRID(64), (size_t)pre_bwd_lds_floats(64, 64) * 4, stream, a); break;
    case 128: rc = launch_chain(node_pre_bwd_kernel<128, 64>, HN_GRID(64), (size_t)pre_bwd_lds_floats(128, 64) * 4, stream, a); break;
    default: rc = launch_chain(node_pre_bwd_kernel<256, 32>, HN_GRID(32), (size_t)pre_bwd_lds_floats(256, 32) * 4, stream, a); break;
  }
#undef HN_GRID
  if (rc != HN_OK || !gx) return rc;              // (gx == NULL: the consumer runs the LayerNorm backward over the parts)
  hipLaunchKernelGGL(layernorm_bwd_parts_kernel, dim3((unsigned)((num_src + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     gn_parts, num_rel, (long)num_src * hidden, x, mean, rstd, add, gx, num_src, hidden,
                     hidden_real > 0 ? hidden_real : hidden, row_windows, num_windows, window_mode, pre_tile_rows(hidden));
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_node_update_fwd(const float* x1, const float* vec1, const float* wv_frag, const float* wx0_frag,
                                       const float* bx0, const float* wx2_frag, const float* bx2,
                                       const float* row_active, const int* type_rowptr, const int* type_rowptr_host,
                                       float* vp, float* h2b, float* q23, float* nrm, float* x_out, float* vec_out,
                                       int num_nodes, int num_rel, int hidden, int tile_rows, void* stream) {
  if (num_nodes < 0 || num_rel <= 0 || !type_rowptr_host) return HN_ERR_BAD_ARG;
  if (!hermnet_node_chain_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_nodes == 0) return HN_OK;
  if (!x1 || !vec1 || !wv_frag || !wx0_frag || !bx0 || !wx2_frag || !bx2 || !type_rowptr || !vp || !h2b || !q23 || !nrm ||
      !x_out || !vec_out || type_rowptr_host[num_rel] > num_nodes)
    return HN_ERR_BAD_ARG;
  UpdFwdArgs a = {x1, vec1, wv_frag, wx0_frag, bx0, wx2_frag, bx2, row_active, type_rowptr, vp, h2b, q23, nrm, x_out,
                  vec_out, num_nodes, num_rel};
  if (tile_rows == 16) return hn_update16_supported(hidden) ? hn_update16_fwd(hidden, a, HN_TILES(16), stream) : HN_ERR_BAD_ARG;
  if (use_wide(hidden)) return hn_wide_update_fwd(hidden, a, HN_TILES(32), stream);
  HN_UPDATE_DISPATCH(node_update_fwd_kernel, a);
}

extern "C" int hermnet_node_update_bwd(const float* gx_out, const float* gvec_out, const float* vp, const float* h2b,
                                       const float* q23, const float* nrm, const float* wx2t_frag, const float* wx0t_frag,
                                       const float* wvt_frag, const float* row_active, const int* type_rowptr,
                                       const int* type_rowptr_host, float* gx1, float* gvec1, int num_nodes, int num_rel,
                                       int hidden, int tile_rows, const hn_pending_grads* pending, void* stream) {
  if (num_nodes < 0 || num_rel <= 0 || !type_rowptr_host) return HN_ERR_BAD_ARG;
  if (!hermnet_node_chain_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_nodes == 0) return HN_OK;
  if (!gx_out || !gvec_out || !vp || !h2b || !q23 || !nrm || !wx2t_frag || !wx0t_frag || !wvt_frag || !type_rowptr || !gx1 ||
      !gvec1 || type_rowptr_host[num_rel] > num_nodes)
    return HN_ERR_BAD_ARG;
  UpdBwdArgs a = {gx_out, gvec_out, vp, h2b, q23, nrm, wx2t_frag, wx0t_frag, wvt_frag, row_active, type_rowptr, gx1, gvec1,
                  num_nodes, num_rel, {}};
  if (pending) {
    const hn_pending_grads& p = *pending;
    if ((!p.gn_parts && !p.gxh) || !p.gvec_parts || !p.x || !p.mean || !p.rstd || !p.gx1 || !p.gvec1 || p.num_parts < 1 ||
        p.hidden_real > hidden)
      return HN_ERR_BAD_ARG;
    a.pend = {p.gn_parts, p.gvec_parts, p.x, p.mean, p.rstd, p.gx1, p.gvec1, p.num_parts,
              p.hidden_real > 0 ? p.hidden_real : hidden, p.gxh, p.hb, p.w2t_frag16, p.w1t_frag16};
    // fused form: the projection's backward of the layer above runs inside this launch (16-row tiles only)
    if (p.gxh && (tile_rows != 16 || !p.hb || !p.w2t_frag16 || !p.w1t_frag16)) return HN_ERR_BAD_ARG;
  }
  if (tile_rows == 16) return hn_update16_supported(hidden) ? hn_update16_bwd(hidden, a, HN_TILES(16), stream) : HN_ERR_BAD_ARG;
  if (use_wide(hidden)) return hn_wide_update_bwd(hidden, a, HN_TILES(32), stream);
  HN_UPDATE_DISPATCH(node_update_bwd_kernel, a);
}

// ---- round 5: the layer boundary as ONE node launch each way (csrc/node_chain16.hip) ------------------------------------------
extern "C" int hermnet_node_fused_supported(int hidden) { return hn_update16_supported(hidden); }

extern "C" int hermnet_node_update_pre_fwd(const float* x1, const float* vec1, const float* wv_frag16, const float* wx0_frag16,
                                           const float* bx0, const float* wx2_frag16, const float* bx2,
                                           const float* row_active, const int* type_rowptr, const int* type_rowptr_host,
                                           float* vp, float* h2b, float* q23, float* nrm, float* x_out, float* vec_out,
                                           int num_nodes, int num_rel, int hidden, const float* w1_frag16, const float* b1,
                                           const float* w2_frag16, const float* b2, float* hb, float* xh, float* mean,
                                           float* rstd, int next_num_rel, int hidden_real, float eps, void* stream) {
  if (num_nodes < 0 || num_rel <= 0 || next_num_rel <= 0 || !type_rowptr_host || hidden_real > hidden) return HN_ERR_BAD_ARG;
  if (!hn_update16_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_nodes == 0) return HN_OK;
  if (!x1 || !vec1 || !wv_frag16 || !wx0_frag16 || !bx0 || !wx2_frag16 || !bx2 || !type_rowptr || !vp || !h2b || !q23 || !nrm ||
      !x_out || !vec_out || !w1_frag16 || !b1 || !w2_frag16 || !b2 || !hb || !xh || !mean || !rstd ||
      type_rowptr_host[num_rel] > num_nodes)
    return HN_ERR_BAD_ARG;
  UpdFwdArgs a = {x1, vec1, wv_frag16, wx0_frag16, bx0, wx2_frag16, bx2, row_active, type_rowptr, vp, h2b, q23, nrm, x_out,
                  vec_out, num_nodes, num_rel};
  PreFwdArgs p = {x_out, w1_frag16, b1, w2_frag16, b2, hb, xh, mean, rstd, nullptr, num_nodes, next_num_rel,
                  hidden_real > 0 ? hidden_real : hidden, eps, nullptr, 0, 0};
  return hn_update16_pre_fwd(hidden, a, p, HN_TILES(16), stream);
}

extern "C" int hermnet_node_pre_fwd16(const float* x, const float* w1_frag16, const float* b1, const float* w2_frag16,
                                      const float* b2, float* hb, float* xh, float* mean, float* rstd, int num_src,
                                      int num_rel, int hidden, int hidden_real, float eps, void* stream) {
  if (num_src < 0 || num_rel <= 0 || hidden_real > hidden || !hn_update16_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_src == 0) return HN_OK;
  if (!x || !w1_frag16 || !b1 || !w2_frag16 || !b2 || !hb || !xh || !mean || !rstd) return HN_ERR_BAD_ARG;
  PreFwdArgs p = {x, w1_frag16, b1, w2_frag16, b2, hb, xh, mean, rstd, nullptr, num_src, num_rel,
                  hidden_real > 0 ? hidden_real : hidden, eps, nullptr, 0, 0};
  return hn_pre16_fwd(hidden, p, stream);
}

extern "C" int hermnet_node_pre_bwd16(const float* gxh, const float* hb, const float* w2t_frag16, const float* w1t_frag16,
                                      float* gn_parts, int num_src, int num_rel, int hidden, void* stream) {
  if (num_src < 0 || num_rel <= 0 || !hn_update16_supported(hidden)) return HN_ERR_BAD_ARG;
  if (num_src == 0) return HN_OK;
  if (!gxh || !hb || !w2t_frag16 || !w1t_frag16 || !gn_parts) return HN_ERR_BAD_ARG;
  PreBwdArgs a = {gxh, hb, w2t_frag16, w1t_frag16, gn_parts, nullptr, num_src, num_rel, nullptr, 0, 0};
  return hn_pre16_bwd(hidden, a, stream);
}

// ---- the read-out on the matrix pipe (csrc/node_chain16.hip: 16-row tiles, hidden 128 -> 64) --------------------------------------
extern "C" int hermnet_energy_head16_supported(int hidden, int cols) { return hn_head16_supported(hidden, cols); }

extern "C" int hermnet_energy_head16_fwd(const float* x, const float* w0_frag16, const float* b0, const float* w2, const float* b2,
                                         const float* row_mask, float* h, float* e, int rows, int hidden, int cols, void* stream) {
  if (rows < 0 || !hn_head16_supported(hidden, cols)) return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!x || !w0_frag16 || !b0 || !w2 || !h || !e) return HN_ERR_BAD_ARG;
  return hn_head16_fwd(x, w0_frag16, b0, w2, b2, row_mask, h, e, rows, stream);
}

extern "C" int hermnet_energy_head16_bwd(const float* ge, const float* h, const float* w0t_frag16, const float* w2,
                                         const float* row_mask, float* gx, int rows, int hidden, int cols, void* stream) {
  if (rows < 0 || !hn_head16_supported(hidden, cols)) return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!ge || !h || !w0t_frag16 || !w2 || !gx) return HN_ERR_BAD_ARG;
  return hn_head16_bwd(ge, h, w0t_frag16, w2, row_mask, gx, rows, stream);
}
