// gfx950 backward (forces) message kernel, "channel per lane" form.
//
// Same contract as message_scatter_bwd_kernel (message_kernels.hip; reference: the backward pass of
// rmnet.py:55-73 + 24-26 for energy/force evaluation), different mapping:
//
//   * a wave works on ONE edge at a time; its 64 lanes are the 64 channels of the column block.  Everything that
//     depends only on the edge -- window start, the 12 Gaussian taps g_m and g_m (u - mu_m), envelope factors,
//     unit vector -- is wave-uniform and arrives through the SCALAR path from a per-edge table
//     (hermnet_edge_radial_table, computed once per step: geometry and radial basis are shared by all layers),
//     so no vector register holds tap values and no lane evaluates an exponential here;
//   * the banded contraction packs {value, derivative}:  (S0, S1) += (g_m, gd_m) * W[lo+m][part][c]  is one
//     v_pk_fma_f32 with an SGPR pair and a broadcast weight; the LDS tile is laid out [tap row][channel][s,a,b,0],
//     one conflict-free ds_read_b128 per tap;
//   * per lane there is one channel of state, ~107 VGPRs in all, so a 1024-thread workgroup puts 4 waves on every
//     SIMD (the VW = 4 form: 245 VGPRs, 2 waves per SIMD, each wave parked ~50 % of its life -- rocprofv3
//     SQ_WAIT_ANY / SQ_WAVE_CYCLES, profiles/r02_v1_counters.json -- and a lone wave issues fp32 VALU at only
//     ~40 % of the SIMD's rate);
//   * segment sums (gxh, gvec) need no cross-lane step at all; the per-edge dE/dD (a sum over channels) is converted
//     to Cartesian per channel (rhat, 1/d are scalars of the edge) and reduced for 4 edges at once with
//     v_permlane32/16 swaps + 4 DPP steps;
//   * a wave walks its share of the chunk's CSC edges as ONE software-pipelined stream: the next edge's record
//     (sequential: the table is in CSC order), its twelve weight rows and its target rows are requested while the
//     current edge is in its channel algebra; row switches cost scalar loads and buffer operations only.
//
// No atomics on HBM, fixed summation order: bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/hermnet_hip.h"
#include "hermnet_math.h"
#include "message_bwd_cl.h"

namespace {

typedef float hn_f2 __attribute__((ext_vector_type(2)));
typedef unsigned hn_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(4))) const float hn_cfloat;   // constant address space: uniform loads go scalar
typedef __attribute__((address_space(4))) const int hn_cint;

constexpr int kRec = HN_EDGE_TABLE_FLOATS;   // floats per edge record
// The table is in CSC order (the order the backward walks): record q belongs to CSC edge q = CSR edge csc_pos[q], so a
// wave's records are one sequential stream.  Layout (m < 12, g_m = exp(coeff (u - mu_m)^2)):
//   [2m]   = env(u) g_m                                   -> sum_m [2m]   W_m = rbfh - bias            (rmnet.py:55,168-172)
//   [2m+1] = (env'(u) g_m + 2 coeff env(u) g_m (u - mu_m)) / rc  -> sum_m [2m+1] W_m = d rbfh / d d
//   [24] padded tile row of tap 0 (int bits) | [25] the same of CSC edge q+1 (the kernel requests that edge's weight
//   rows while it still works on edge q) | [26,27] 0 | [28..30] rhat | [31] 1/d
// The envelope factors are folded into the tap pairs, so the contraction yields rbfh and its distance derivative
// directly (no per-channel envelope arithmetic in the message kernel).

// Records leave through a wave-private LDS transpose: a lane computes one record (128 B), and written straight from its
// registers every store instruction would touch 64 different 128-byte lines with 16 bytes each (measured 25.7 us for the 55 MB
// table of configs[1]); through the tile the wave writes its 64 records as eight fully coalesced 1-KiB stores.
// (measured, profiles/r06_nt_msg_ab.log: 2.832 -> 2.822 ms/step)
#ifndef HN_NT_TABLE
#define HN_NT_TABLE 1
#endif
__global__ __launch_bounds__(256) void edge_table_kernel(const float4* __restrict__ edge, const int* __restrict__ csc_pos,
                                                         int E, const float* __restrict__ offset, int R, float inv_rc,
                                                         float coeff, int env_kind, int env_p,
                                                         float* __restrict__ table, const int* __restrict__ csc_end) {
  constexpr int kLd = kRec + 4;                          // 36 floats: 16-byte aligned rows, 8 lanes per 128 B then a 144-B step
  __shared__ __align__(16) float stage[4][64 * kLd];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = (blockIdx.x * 4 + wave) * 64;           // first record of this wave
  if (q0 >= E) return;
  const int q = q0 + lane;
  const int n_csc = csc_end[0];
  float rec[kRec];
#pragma unroll
  for (int w = 0; w < kRec; ++w) rec[w] = 0.f;
  // CSC positions behind the last segment (edges into unknown-element rows, the NULL edges of a padded list) belong to
  // no row: csc_pos is not defined there.  Their records are never an edge's own record -- only the one a wave requests
  // AHEAD of its last edge -- so they hold a valid tile row and zeros.
  rec[24] = __int_as_float(HN_PAD);
  rec[25] = __int_as_float(HN_PAD);
  if (q < n_csc) {
    const float4 g = edge[csc_pos[q]];
    const float d_next = edge[csc_pos[min(q + 1, n_csc - 1)]].w;
    const float u = g.w * inv_rc;
    const HnEnv env = hn_envelope(u, env_kind, env_p);
    const int lo = hn_window_lo(u, R);
    const float c0 = inv_rc * env.der, c1 = inv_rc * env.val * 2.0f * coeff;    // d rbfh / d d = c0 S0 + c1 S1
#pragma unroll
    for (int m = 0; m < HN_TAPS; ++m) {
      int k = lo + m;
      k = k < 0 ? 0 : (k >= R ? R - 1 : k);                 // (rows outside [0, R) hold zero weights)
      const float diff = u - offset[k];
      const float gm = __expf(coeff * (diff * diff));        // same fp32 operation order as rmnet.py:156-172
      rec[2 * m] = env.val * gm;
      rec[2 * m + 1] = c0 * gm + c1 * (gm * diff);
    }
    rec[24] = __int_as_float(lo + HN_PAD);                  // padded tile row of tap 0
    rec[25] = __int_as_float(hn_window_lo(d_next * inv_rc, R) + HN_PAD);
    rec[28] = g.x; rec[29] = g.y; rec[30] = g.z;
    rec[31] = __builtin_amdgcn_rcpf(g.w);
  }
  float* st = stage[wave];
#pragma unroll
  for (int w = 0; w < kRec / 4; ++w)
    *reinterpret_cast<float4*>(st + lane * kLd + 4 * w) = make_float4(rec[4 * w], rec[4 * w + 1], rec[4 * w + 2], rec[4 * w + 3]);
  // (LDS operations of one wave execute in order: no barrier between its own writes and reads)
  const int nrec = min(64, E - q0);                       // records of this wave that exist
  float4* out = reinterpret_cast<float4*>(table + (size_t)q0 * kRec);
#pragma unroll
  for (int j = 0; j < kRec / 4; ++j) {
    const int idx = j * 64 + lane, r = idx / (kRec / 4), w = idx % (kRec / 4);
    // (non-temporal where HN_NT_TABLE: the table is written at the start of the step and first read by the backward pass)
    if (r < nrec) {
#if HN_NT_TABLE
      typedef float f4v __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(*reinterpret_cast<const f4v*>(st + r * kLd + 4 * w), reinterpret_cast<f4v*>(out + idx));
#else
      out[idx] = *reinterpret_cast<const float4*>(st + r * kLd + 4 * w);
#endif
    }
  }
  // record E: a copy of the last one -- the message kernel requests record q + 1 without a bounds check
  if (q0 + nrec == E && lane < kRec / 4)
    out[(size_t)nrec * (kRec / 4) + lane] = *reinterpret_cast<const float4*>(st + (nrec - 1) * kLd + 4 * lane);
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// 64-lane totals of one quantity of FOUR edges, in two stages so that only one register per quantity stays live
// between edge pairs: pair_fold(e0, e1) -> halves hold 32-lane partial sums of e0 | e1; quad_total(u01, u23) -> every
// lane of 16-lane row k holds the total of edge {0, 2, 1, 3}[k].
__device__ __forceinline__ float pair_fold(float e0, float e1) {
  const hn_u2 s = __builtin_amdgcn_permlane32_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
  return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
__device__ __forceinline__ float quad_total(float u, float w) {
  const hn_u2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(w), false, false);
  float v = __uint_as_float(s[0]) + __uint_as_float(s[1]);             // rows: e0, e2, e1, e3
  v += dpp_mov<0xB1>(v);    // lane ^ 1
  v += dpp_mov<0x4E>(v);    // lane ^ 2
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}

// The 12 tap rows of this lane's weight record are fetched as twelve ds_read_b128 (4 LDS cycles each, conflict-free) in
// three groups of four registers, all requested one edge AHEAD (see the kernel).  Written as asm because hipcc narrows a
// float4 LDS load whose .w is unused to ds_read_b96 (8 LDS cycles), and because the data must stay untouched in its
// registers until the one wait at the top of the next edge: that wait names all twelve registers as "+v" operands, so no
// use can be scheduled in front of it (and the generated code is checked for copies of registers in flight).
typedef float hn_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_issue4(unsigned addr, hn_f4 (&w)[4]) {
  asm volatile(
      "ds_read_b128 %0, %4\n\t"
      "ds_read_b128 %1, %4 offset:1024\n\t"
      "ds_read_b128 %2, %4 offset:2048\n\t"
      "ds_read_b128 %3, %4 offset:3072"
      : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
      : "v"(addr)
      : "memory");
}
// everything this wave has in flight on the LDS / scalar-memory counter has landed (the edge's record and its first two
// tap groups, requested during the previous edge); releases both groups
__device__ __forceinline__ void lds_wait_all(hn_f4 (&wa)[4], hn_f4 (&wb)[4], hn_f4 (&wc)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(wa[0]), "+v"(wa[1]), "+v"(wa[2]), "+v"(wa[3]), "+v"(wb[0]), "+v"(wb[1]), "+v"(wb[2]), "+v"(wb[3]),
                 "+v"(wc[0]), "+v"(wc[1]), "+v"(wc[2]), "+v"(wc[3])
               :: "memory");
}
static_assert(HN_CB * 16 == 1024 && HN_TAPS == 12, "the tap reader hard-codes the 1 KiB row pitch and 3 x 4 taps");

typedef __amdgpu_buffer_rsrc_t hn_rsrc;
__device__ __forceinline__ float buf_load(hn_rsrc r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

__device__ __forceinline__ void buf_store(hn_rsrc r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
// non-temporal forms (aux bit 1 = nt) for what this kernel touches exactly once: the source row's own xh / vec values (saved by
// the forward milliseconds ago: they come from HBM and nobody reads them again) and the per-edge gradient slots (read once, at
// the end of the backward pass)
// Measured (profiles/r06_nt_msg_ab.log, interleaved in one job): configs[1] 2.787 -> 2.765 ms/step.
#ifndef HN_NT_MSG
#define HN_NT_MSG 1
#endif
__device__ __forceinline__ float buf_load_once(hn_rsrc r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, HN_NT_MSG ? 2 : 0));
}
// descriptor of ONE row of a [rows, 3, H] array (wave-uniform base, 3H floats): the row's three parts are then
// addressed by scalar offsets and this lane's channel offset -- no per-lane 64-bit address arithmetic
__device__ __forceinline__ hn_rsrc row_rsrc(unsigned long long base, unsigned row_offset_elems, int H) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (unsigned long long)row_offset_elems * 4ull), 0, 3 * H * 4, 0x00020000);
}
// address of a wave-uniform pointer, pinned into scalar registers (hipcc does 64-bit multiplies of the relation / row
// offsets on the vector ALU; everything derived from the product would then live in VGPRs and every descriptor built
// from it would be wrapped in a waterfall loop)
__device__ __forceinline__ unsigned long long uniform_addr(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

struct EdgeIn {          // what one edge needs from memory (vector part)
  float gx1, g0, g1, g2;
};

// One workgroup = (relation t, 64-channel column block, chunk of SOURCE rows): it stages the relation's weight tile
// once and never synchronises again.  Its 16 waves split the chunk's CSC edge range into 16 equal shares at row
// boundaries (a row belongs to the wave in whose share its first edge lies), so the waves are balanced to within one
// segment without a queue, and each wave walks its edges as ONE continuous stream: the per-edge prefetch pipeline
// (indices 64 at a time, target rows and the scalar record one edge ahead) does not drain at the row boundaries --
// segments are short (~14 edges), and a wave that restarts its loads per segment spends a third of its time in
// dependent round trips.  Rows switch in-line: epilogue of the finished row, then the new row's own values.
// gvec is written per relation ([T, Nsrc, 3, H] partial sums: three workgroups never read-modify-write the same row);
// message_bwd_finish_kernel adds the T slices in a fixed order together with the residual's identity terms
// (rmnet.py:24-26: gx = gx1 / sqrt2, gvec += gvec1 on target rows).
//
// WIN = true (num_rbf > 137: the whole tile would take more than the 160 KiB of LDS): the launch stages a WINDOW of the
// tile's rows and owns the edges whose twelve taps lie inside it (HnBwdClArgs::win_*); two launches with windows that
// overlap by eleven rows cover every edge exactly once.  Ownership is wave-uniform like everything else about an edge: a
// foreign edge walks through the same pipeline with its target gradients multiplied by zero (every sum of the kernel is
// linear in them) and a clamped tile row; the second launch adds its row sums to the first one's.
#ifndef HN_KO_BWD
#define HN_KO_BWD 0      // 1: no contraction, 2: no channel algebra (diagnostic builds: tools/build_variant.sh)
#endif
template <bool HAS_VEC, bool WIN>
__global__ __launch_bounds__(1024, 4) void message_scatter_bwd_cl_kernel(HnBwdClArgs a) {
  extern __shared__ __align__(16) float4 tile[];     // [tap row][64] of (s, a, b, 0)
  const int tile_rows = WIN ? a.win_rows : a.R + 2 * HN_PAD + 1;
  const int tile_base = WIN ? a.win_base : 0;        // padded tile row held by tile[0..63]

  // XCD-aware order (see message_kernels.hip: xcd_contiguous): the dispatcher deals the linearised grid round-robin over
  // the 8 XCDs; XCD k gets the k-th contiguous eighth of (relation, column block, row chunk), i.e. a slab of source rows
  // whose targets' gradient rows (the gathers) are mostly its own slab's atoms.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd_remap) {
    const int total = gridDim.x * gridDim.y * gridDim.z, per = total >> 3;
    int L = bx + gridDim.x * (by + gridDim.y * bz);
    if (L < per * 8) {
      L = (L & 7) * per + (L >> 3);
      bx = L % gridDim.x; by = (L / gridDim.x) % gridDim.y; bz = L / (gridDim.x * gridDim.y);
    }
  }
  const int t = __builtin_amdgcn_readfirstlane(bz);
  const int cb = __builtin_amdgcn_readfirstlane(by);
  int r0, r1;
  if (a.num_ranges > 0) {      // chunk bx of the concatenated ranges (chunks do not straddle a range)
    int c = __builtin_amdgcn_readfirstlane(bx);
    r0 = r1 = 0;
    bool found = false;
    for (int k = 0; k < a.num_ranges && !found; ++k) {
      const int lo = a.src_ranges[2 * k], hi = a.src_ranges[2 * k + 1];
      const int nb = hi > lo ? (hi - lo + a.rows_per_block - 1) / a.rows_per_block : 0;
      if (c < nb) { r0 = lo + c * a.rows_per_block; r1 = min(r0 + a.rows_per_block, hi); found = true; }
      c -= nb;
    }
    if (!found) return;        // (the grid is sized from the host's copy of the ranges: no surplus chunks expected)
    r0 = __builtin_amdgcn_readfirstlane(r0);
    r1 = __builtin_amdgcn_readfirstlane(r1);
  } else {
    r0 = __builtin_amdgcn_readfirstlane(bx) * a.rows_per_block;
    r1 = min(r0 + a.rows_per_block, a.Nsrc);
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int H = a.H;
  const unsigned c = (unsigned)(cb * HN_CB + lane);  // this lane's channel (unsigned: zero-extended lane offset, so that
                                                     // `uniform_base[c]` is a scalar-base + 32-bit-offset access)
  const float inv_sqrt3h = 0.57735026918962576f * rsqrtf((float)H);
  const float inv_sqrth = rsqrtf((float)H);
  const float inv_sqrt2 = 0.70710678118654752f;
  float4* gedge = a.gedge + (size_t)cb * a.E;
  const int row16 = lane >> 4;                       // DPP row of this lane
  // LDS byte address of this lane's record in tap row 0 (the tile starts the dynamic region)
  const unsigned tile_lane = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)tile) + (unsigned)lane * 16u;

  // gathered target rows go through buffer descriptors (the host guarantees both arrays are < 4 GiB)
  const hn_rsrc rs_gx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gx1), 0, (int)((size_t)a.N * H * 4), 0x00020000);
  const hn_rsrc rs_gvec1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gvec1), 0, (int)((size_t)a.N * 3 * H * 4), 0x00020000);
  const unsigned c4 = c * 4u;

  // ---- stage the weight tile of (relation t, column block cb): rows outside [0, R) are zero
  for (int idx = threadIdx.x; idx < tile_rows * HN_CB; idx += blockDim.x) {
    const int k = idx / HN_CB + tile_base - HN_PAD, ch = idx % HN_CB;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k >= 0 && k < a.R) {
      const float* w = a.wt + ((size_t)(t * a.R + k) * 3 * H + cb * HN_CB + ch);
      v = make_float4(w[0], w[H], w[2 * H], 0.f);
    }
    tile[idx] = v;
  }

  const unsigned long long xh_t = uniform_addr(a.xh + (size_t)t * a.Nsrc * 3 * H);
  const unsigned long long gxh_t = uniform_addr(a.gxh + (size_t)t * a.Nsrc * 3 * H);
  const unsigned long long gvec_t = HAS_VEC ? uniform_addr(a.gvec + (size_t)t * a.Nsrc * 3 * H) : 0ull;
  const unsigned long long vec_a = HAS_VEC ? uniform_addr(a.vec) : 0ull;
  const int* rowptr_t = a.csc_rowptr + (size_t)t * a.Nsrc;
  const float* brow = a.brbf + (size_t)t * 3 * H;
  const float bs = brow[c], ba = (brow + H)[c], bb = (brow + 2 * H)[c];
  float xbs = 0.f, xba = 0.f, xbb = 0.f;
  if (a.xh_bias) {
    const float* xrow = a.xh_bias + (size_t)t * 3 * H;
    xbs = xrow[c]; xba = (xrow + H)[c]; xbb = (xrow + 2 * H)[c];
  }

  // ---- this wave's rows: those whose first edge lies in its share [es, ee) of the chunk's edge range
  const int E0 = rowptr_t[r0], E1 = rowptr_t[r1];
  const long span = (long)E1 - E0;
  const int es = E0 + (int)(span * wave / nwaves);
  const int ee = (wave + 1 == nwaves) ? E1 + 1 : E0 + (int)(span * (wave + 1) / nwaves);
  int ra = r0, rb = r0;                              // rows [ra, rb): counts of rows starting below es / ee
  for (int q = r0; q < r1; q += 64) {
    const int rq = q + lane;
    const int start = rq < r1 ? rowptr_t[rq] : 0x7fffffff;
    ra += __popcll(__ballot(start < es));
    rb += __popcll(__ballot(start < ee));
  }
  __syncthreads();                                   // the tile is staged

  if (ra < rb) {
    // row pointers through the scalar path, requested one row ahead: a row switch then costs no vector-memory wait
    // (a vector load + v_readlane made every switch drain the whole vector-memory queue, the gathers in flight included)
    const hn_cint* rowptr_c = (const hn_cint*)uniform_addr(rowptr_t);
    int row = ra;
    const int e_begin = rowptr_c[ra];
    const int e_end = rowptr_c[rb];
    int row_end = rowptr_c[row + 1];
    int row_end2 = rowptr_c[min(row + 2, a.Nsrc)];

    // row values, pre-multiplied by the constant factors they always meet (rmnet.py:24,63-66): xs / sqrt2,
    // xa / sqrt(3H) (and xa itself), xb / sqrt(H), vec_j / sqrt(3H); gs and gb are rescaled once per row.
    // A new row's values are only REQUESTED at the switch (raw registers); they are scaled after the first edge's
    // contraction, which needs none of them -- scaling at once would wait for the loads on the spot.
    float xs = 0.f, xa = 0.f, xas = 0.f, xb = 0.f, vj0 = 0.f, vj1 = 0.f, vj2 = 0.f;
    float raw_s = 0.f, raw_a = 0.f, raw_b = 0.f, raw_v0 = 0.f, raw_v1 = 0.f, raw_v2 = 0.f;
    bool fresh = false;
    float gs = 0.f, ga = 0.f, gb = 0.f, gv0 = 0.f, gv1 = 0.f, gv2 = 0.f;
    const unsigned h4 = (unsigned)(H * 4);
    auto row_request = [&](int r) {
      // (row offsets in 32 bits: the host routes larger arrays to the other kernel form)
      const unsigned ro = (unsigned)__builtin_amdgcn_readfirstlane(r * (3 * H));
      const hn_rsrc xr = row_rsrc(xh_t, ro, H);
      raw_s = buf_load_once(xr, c4, 0); raw_a = buf_load_once(xr, c4, h4); raw_b = buf_load_once(xr, c4, 2 * h4);
      if (HAS_VEC) {
        const hn_rsrc vr = row_rsrc(vec_a, ro, H);
        raw_v0 = buf_load_once(vr, c4, 0); raw_v1 = buf_load_once(vr, c4, h4); raw_v2 = buf_load_once(vr, c4, 2 * h4);
      }
      gs = 0.f; ga = 0.f; gb = 0.f; gv0 = 0.f; gv1 = 0.f; gv2 = 0.f;
      fresh = true;
    };
    auto row_scale = [&]() {
      xs = (raw_s + xbs) * inv_sqrt2; xa = raw_a + xba; xb = (raw_b + xbb) * inv_sqrth;
      xas = xa * inv_sqrt3h;
      if (HAS_VEC) { vj0 = raw_v0 * inv_sqrt3h; vj1 = raw_v1 * inv_sqrt3h; vj2 = raw_v2 * inv_sqrt3h; }
      fresh = false;
    };
    auto row_epilogue = [&](int r) {                 // every lane owns its channel: plain coalesced stores
      const unsigned ro = (unsigned)__builtin_amdgcn_readfirstlane(r * (3 * H));
      const hn_rsrc go = row_rsrc(gxh_t, ro, H);
      if constexpr (WIN) {
        float o0 = gs * inv_sqrt2, o1 = ga, o2 = gb * inv_sqrth;
        if (a.win_accumulate) { o0 += buf_load(go, c4, 0); o1 += buf_load(go, c4, h4); o2 += buf_load(go, c4, 2 * h4); }
        buf_store(go, c4, 0, o0); buf_store(go, c4, h4, o1); buf_store(go, c4, 2 * h4, o2);
      } else {
        buf_store(go, c4, 0, gs * inv_sqrt2); buf_store(go, c4, h4, ga); buf_store(go, c4, 2 * h4, gb * inv_sqrth);
      }
      if (HAS_VEC) {
        const hn_rsrc gvo = row_rsrc(gvec_t, ro, H);
        if constexpr (WIN) {
          if (a.win_accumulate) { gv0 += buf_load(gvo, c4, 0); gv1 += buf_load(gvo, c4, h4); gv2 += buf_load(gvo, c4, 2 * h4); }
        }
        buf_store(gvo, c4, 0, gv0); buf_store(gvo, c4, h4, gv1); buf_store(gvo, c4, 2 * h4, gv2);
      }
    };
    // (WIN) tile row of an edge inside this launch's window, clamped for the edges it does not own; and who owns it
    auto win_row = [&](int prow) { return min(max(prow - a.win_base, 0), a.win_rows - HN_TAPS); };
    auto win_owns = [&](int prow) { return (prow >= a.win_lo && prow < a.win_hi) ? 1u : 0u; };
    unsigned own_cur = 1u;                             // (WIN) 1 if this launch owns the edge the stream is at
    unsigned own_mask = 0u;                            // ... bit j: it owns edge j of the current group of four
    row_request(row);

    // The record stream and the weight-row reads run CONTINUOUSLY over the wave's edges: while edge q is in its channel
    // algebra, edge q+1's record (scalar loads; the table is in CSC order, so it is the next 128 bytes) and its twelve
    // weight rows (ds_read_b128; the tile row comes from record q, slot 25) are already in flight.  A wave then meets ONE
    // LDS / scalar-memory wait per edge, for data requested a whole algebra section earlier.
    // Instruction diet (the kernel's time is ~ instructions per edge x 3.8 cycles per SIMD): the record pointer is a
    // running scalar (the table has one spare record, so no clamp), the record's tail (next tile row, rhat, 1/d) lands
    // in one of two scalar sets by edge parity so that the algebra still reads edge q's tail while q+1's arrives,
    // gather offsets are pre-multiplied once per 64-edge batch, and whole groups of four edges run without
    // per-edge bounds checks.
    float rec[24];                                     // the 12 {value, derivative} tap pairs of the current edge
    float tl0[5], tl1[5];                              // (next tile row, rhat x y z, 1/d) of even / odd edges
    hn_f4 wA[4], wB[4], wC[4];
    const hn_cfloat* rp = (const hn_cfloat*)(a.table + (size_t)e_begin * kRec);
    auto load_record = [&](float (&tl)[5]) {
      // wave-uniform and read-only: through the constant address space these are scalar loads into SGPRs
#pragma unroll
      for (int w = 0; w < 24; ++w) rec[w] = rp[w];
      tl[0] = rp[25];
#pragma unroll
      for (int w = 0; w < 4; ++w) tl[1 + w] = rp[28 + w];
    };
    auto issue_weights = [&](unsigned waddr) {
      lds_issue4(waddr, wA);
      lds_issue4(waddr + 4 * (HN_CB * 16), wB);
      lds_issue4(waddr + 8 * (HN_CB * 16), wC);
    };
    if (e_begin < e_end) {
      load_record(tl0);
      if constexpr (WIN) {
        const int prow0 = __float_as_int(rp[24]);
        own_cur = win_owns(prow0);
        issue_weights((unsigned)win_row(prow0) * (HN_CB * 16) + tile_lane);
      } else {
        issue_weights((unsigned)__float_as_int(rp[24]) * (HN_CB * 16) + tile_lane);
      }
    }
    for (int base = e_begin; base < e_end; base += 64) {
      const int cnt = min(64, e_end - base);
      // one coalesced index load per 64 edges; an edge's indices are then wave-uniform (v_readlane)
      const unsigned my_off = lane < cnt ? (unsigned)a.csc_tgt[base + lane] * (unsigned)(H * 4) : 0u;   // byte offset in gx1
      const int my_pos = lane < cnt ? a.csc_pos[base + lane] : 0;

      auto load_edge = [&](int k) {
        EdgeIn in;
        // buffer loads: descriptor + scalar row offset + this lane's channel offset -- no per-lane 64-bit address math
        const unsigned so = (unsigned)__builtin_amdgcn_readlane((int)my_off, k);
        in.gx1 = buf_load(rs_gx1, c4, so);
        in.g0 = buf_load(rs_gvec1, c4, 3u * so);
        in.g1 = buf_load(rs_gvec1, c4, 3u * so + h4);
        in.g2 = buf_load(rs_gvec1, c4, 3u * so + 2 * h4);
        return in;
      };

      // two named buffers, alternating with the parity of the (unrolled) edge slot: no register moves to rotate them
      EdgeIn in0 = load_edge(0), in1 = in0;
      // per-channel Cartesian dE/dD of the current pair of edges / folded pairs (32-lane partial sums)
      float px[2], py[2], pz[2];
      float ux[2], uy[2], uz[2];
      auto edge_body = [&](auto J, int k) {
        constexpr int j = decltype(J)::value;
        while (base + k >= row_end) {                         // (wave-uniform) the stream enters the next row
          row_epilogue(row);
          ++row;
          row_end = row_end2;
          row_end2 = rowptr_c[min(row + 2, a.Nsrc)];
          row_request(row);
        }
        // the next edge's rows fly during this edge's math
        // (index lanes past the batch hold offset 0: a harmless extra read of row 0; only slot 3 can step past lane 63)
        const int kn = j == 3 ? min(k + 1, 63) : k + 1;
        if (j & 1) in0 = load_edge(kn); else in1 = load_edge(kn);
        const EdgeIn& cur = (j & 1) ? in1 : in0;
        float (&tl)[5] = (j & 1) ? tl1 : tl0;
        // ---- (S0, S1) of the three parts: 12 taps, one LDS read each, requested during the previous edge
        hn_f2 Ss = {0.f, 0.f}, Sa = {0.f, 0.f}, Sb = {0.f, 0.f};
        auto taps4 = [&](int m0, const hn_f4 (&w)[4]) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const hn_f2 g = {rec[2 * (m0 + q)], rec[2 * (m0 + q) + 1]};
            Ss = __builtin_elementwise_fma(g, hn_f2{w[q].x, w[q].x}, Ss);
            if (HAS_VEC) Sa = __builtin_elementwise_fma(g, hn_f2{w[q].y, w[q].y}, Sa);
            Sb = __builtin_elementwise_fma(g, hn_f2{w[q].z, w[q].z}, Sb);
          }
        };
        lds_wait_all(wA, wB, wC);
#if HN_KO_BWD == 1
        // diagnostic build (results wrong, time meaningful): the twelve-tap contraction replaced by three moves -- the weight
        // rows are still read and waited for, every other instruction of the edge is in place
        Ss = hn_f2{rec[0] * wA[0].x, rec[1]}; Sa = hn_f2{rec[2] * wB[0].y, rec[3]}; Sb = hn_f2{rec[4] * wC[0].z, rec[5]};
#else
        taps4(0, wA);
        taps4(4, wB);
        taps4(8, wC);
#endif
        unsigned waddr_next;
        float own = 1.0f;
        if constexpr (WIN) {
          const int prow_next = __float_as_int(tl[0]);
          waddr_next = (unsigned)win_row(prow_next) * (HN_CB * 16) + tile_lane;
          own = own_cur ? 1.0f : 0.0f;
          own_mask = j == 0 ? own_cur : (own_mask | (own_cur << j));
          own_cur = win_owns(prow_next);
        } else {
          waddr_next = (unsigned)__float_as_int(tl[0]) * (HN_CB * 16) + tile_lane;
        }
        // the taps are consumed: request the NEXT edge's record (taps into the same scalar registers, tail into the other
        // set) and its weight rows into the same vector registers; their latency hides behind the rest of this edge
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(Ss), "+v"(Sa), "+v"(Sb));      // (every tap of this edge has been issued)
        rp += kRec;
        if (j & 1) load_record(tl0); else load_record(tl1);
        issue_weights(waddr_next);
        __builtin_amdgcn_sched_barrier(0);
        if (fresh) row_scale();                               // (wave-uniform) first edge of a row
#if HN_KO_BWD == 2
        // diagnostic build: the channel algebra replaced by the fewest operations that keep the contraction's results and the
        // gathered rows alive (results wrong): what the contraction, the loads and the reduction cost without it
        {
          gs += Ss.x + Ss.y + cur.gx1; ga += Sa.x + Sa.y + cur.g0; gb += Sb.x + Sb.y + cur.g1 + cur.g2;
          px[j & 1] = gs; py[j & 1] = ga; pz[j & 1] = gb;
          return;
        }
#endif
        const float rx = tl[1], ry = tl[2], rz = tl[3], invd = tl[4];
        const float gx1 = WIN ? cur.gx1 * own : cur.gx1, g0 = WIN ? cur.g0 * own : cur.g0;
        const float g1 = WIN ? cur.g1 * own : cur.g1, g2 = WIN ? cur.g2 * own : cur.g2;
        // (the .x sums are rbfh - bias, the .y sums d rbfh / d d: see the record layout)
        // ---- part s: dx = sum xs * rs
        const float rs = Ss.x + bs;
        gs = fmaf(gx1, rs, gs);
        float pdv = gx1 * xs * Ss.y;
        // ---- part a: dvec += vec_j * (xa * ra) / sqrt(3H)
        if (HAS_VEC) {
          const float ra_ = Sa.x + ba;
          const float A = fmaf(g0, vj0, fmaf(g1, vj1, g2 * vj2));
          ga = fmaf(A, ra_, ga);
          const float w = xas * ra_;
          gv0 = fmaf(g0, w, gv0); gv1 = fmaf(g1, w, gv1); gv2 = fmaf(g2, w, gv2);
          pdv = fmaf(A * xa, Sa.y, pdv);
        }
        // ---- part b: dvec += rhat * (xb * rb) / sqrt(H)
        const float rb_ = Sb.x + bb;
        const float B = fmaf(rx, g0, fmaf(ry, g1, rz * g2));
        gb = fmaf(B, rb_, gb);
        pdv = fmaf(B * xb, Sb.y, pdv);
        // ---- this channel's share of dE/dD, already Cartesian (rhat and 1/d are wave-uniform scalars here):
        // d = |D|, rhat = D/d:  gD = pd rhat + (pr - (pr.rhat) rhat) / d  with  pr = q (g0,g1,g2), pr.rhat = q B
        const float qi = xb * rb_ * invd;
        const float tpar = fmaf(-B, qi, pdv);
        px[j & 1] = fmaf(tpar, rx, g0 * qi); py[j & 1] = fmaf(tpar, ry, g1 * qi); pz[j & 1] = fmaf(tpar, rz, g2 * qi);
      };
      auto fold = [&](int h) {
        ux[h] = pair_fold(px[0], px[1]); uy[h] = pair_fold(py[0], py[1]); uz[h] = pair_fold(pz[0], pz[1]);
      };
      // ---- channel sums of (up to) four edges, one 16-byte store per edge (DPP row k holds edge {0,2,1,3}[k])
      auto group_store = [&](int k4) {
        const float sx = quad_total(ux[0], ux[1]), sy = quad_total(uy[0], uy[1]), sz = quad_total(uz[0], uz[1]);
        const int je = (row16 == 0) ? 0 : (row16 == 1 ? 2 : (row16 == 2 ? 1 : 3));
        const int p0 = __builtin_amdgcn_readlane(my_pos, k4), p1 = __builtin_amdgcn_readlane(my_pos, min(k4 + 1, cnt - 1));
        const int p2 = __builtin_amdgcn_readlane(my_pos, min(k4 + 2, cnt - 1)), p3 = __builtin_amdgcn_readlane(my_pos, min(k4 + 3, cnt - 1));
        const int pw = (row16 == 0) ? p0 : (row16 == 1 ? p2 : (row16 == 2 ? p1 : p3));
        if constexpr (WIN) {
          if ((lane & 15) == 0 && k4 + je < cnt && ((own_mask >> je) & 1u)) gedge[pw] = make_float4(sx, sy, sz, 0.f);
        } else {
#if HN_NT_MSG
          if ((lane & 15) == 0 && k4 + je < cnt)
            __builtin_nontemporal_store(hn_f4{sx, sy, sz, 0.f}, reinterpret_cast<hn_f4*>(&gedge[pw]));
#else
          if ((lane & 15) == 0 && k4 + je < cnt) gedge[pw] = make_float4(sx, sy, sz, 0.f);
#endif
        }
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      int k4 = 0;
      for (; k4 + 4 <= cnt; k4 += 4) {                        // whole groups: no per-edge bounds checks
        edge_body(I0{}, k4); edge_body(I1{}, k4 + 1); fold(0);
        edge_body(I2{}, k4 + 2); edge_body(I3{}, k4 + 3); fold(1);
        group_store(k4);
      }
      if (k4 < cnt) {                                         // the stream's last, partial group: absent edges add zeros
        px[1] = 0.f; py[1] = 0.f; pz[1] = 0.f;
        edge_body(I0{}, k4);
        if (k4 + 1 < cnt) edge_body(I1{}, k4 + 1);
        fold(0);
        px[0] = 0.f; py[0] = 0.f; pz[0] = 0.f; px[1] = 0.f; py[1] = 0.f; pz[1] = 0.f;
        if (k4 + 2 < cnt) edge_body(I2{}, k4 + 2);
        fold(1);
        group_store(k4);
      }
    }
    // ---- the stream is exhausted: finish the current row and the rows without edges behind it
    for (;;) {
      row_epilogue(row);
      if (++row >= rb) break;
      gs = 0.f; ga = 0.f; gb = 0.f; gv0 = 0.f; gv1 = 0.f; gv2 = 0.f;
    }
  }
}

// gvec[r] = sum_t part[t][r] (+ gvec1[r] on target rows), gx[r] = gx1[r] / sqrt2 on target rows, 0 elsewhere.
__global__ __launch_bounds__(256) void message_bwd_finish_kernel(const float4* __restrict__ part, int T, long n4_vec,
                                                                 const float4* __restrict__ gvec1,
                                                                 const float4* __restrict__ gx1,
                                                                 const int* __restrict__ type_rowptr, int identity,
                                                                 int H, float4* __restrict__ gvec, float4* __restrict__ gx,
                                                                 long n4_x, const int* __restrict__ ranges, int num_ranges) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int identity_rows = identity ? type_rowptr[T] : 0;      // rows below are targets of a known type
  if (num_ranges > 0) {        // only the rows of this launch's ranges (the others belong to the other launch)
    bool vin = false, xin = false;
    const long rv = i / (3 * H / 4), rx = i / (H / 4);
    for (int k = 0; k < num_ranges; ++k) {
      const int lo = ranges[2 * k], hi = ranges[2 * k + 1];
      vin |= rv >= lo && rv < hi;
      xin |= rx >= lo && rx < hi;
    }
    if (!vin) n4_vec = 0;
    if (!xin) n4_x = 0;
  }
  if (part && i < n4_vec) {
    float4 acc = part[i];
    for (int t = 1; t < T; ++t) {
      const float4 v = part[(long)t * n4_vec + i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (i / (3 * H / 4) < identity_rows) {
      const float4 v = gvec1[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    gvec[i] = acc;
  }
  if (i < n4_x) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i / (H / 4) < identity_rows) {
      const float4 g = gx1[i];
      const float s = 0.70710678118654752f;
      v = make_float4(g.x * s, g.y * s, g.z * s, g.w * s);
    }
    gx[i] = v;
  }
}

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

}  // namespace

size_t hn_bwd_cl_lds_bytes(int R) {
  return (size_t)(R + 2 * HN_PAD + 1) * HN_CB * sizeof(float4);
}

int hn_bwd_cl_launch(HnBwdClArgs a, bool has_vec, int rows_override, const int* ranges_host, hipStream_t s) {
  // The whole tile (num_rbf + 23 rows of 1 KiB) fits the LDS up to num_rbf = 137.  Beyond that: two launches over
  // tap-row windows.  The first owns the edges whose first tap row is below `split` and stages rows [0, split + 11), the
  // second owns the rest and stages [split, rows): 145 rows each at num_rbf = 256, 160 at 286 (the limit).
  constexpr int kMaxRows = 160;
  const int rows_all = a.R + 2 * HN_PAD + 1;
  const bool windowed = rows_all > kMaxRows;
  const int split = (rows_all - HN_PAD + 1) / 2;
  if (windowed && (split + HN_PAD > kMaxRows || rows_all - split > kMaxRows)) return HN_ERR_LDS;
  const size_t lds = windowed ? (size_t)(split + HN_PAD > rows_all - split ? split + HN_PAD : rows_all - split) * HN_CB * sizeof(float4)
                              : hn_bwd_cl_lds_bytes(a.R);
  typedef void (*kern_t)(HnBwdClArgs);
  kern_t k = windowed ? (has_vec ? message_scatter_bwd_cl_kernel<true, true> : message_scatter_bwd_cl_kernel<false, true>)
                      : (has_vec ? message_scatter_bwd_cl_kernel<true, false> : message_scatter_bwd_cl_kernel<false, false>);
  static bool done[4] = {false, false, false, false};
  const int which = (windowed ? 2 : 0) + (has_vec ? 1 : 0);
  if (!done[which]) {      // (once per kernel, for the largest tile it can be given: a host-side attribute)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kMaxRows * HN_CB * (int)sizeof(float4)) != hipSuccess)
      return HN_ERR_LDS;
    done[which] = true;
  }
  // one (relation, column block, row chunk) per workgroup; the chunk is sized for a whole number of rounds of one
  // workgroup per CU (the 154 KB tile allows one resident workgroup): ~256 rows = 16 rows per wave
  const int ncb = a.H / HN_CB;
  int rpb = rows_override;
  long rows = a.Nsrc;
  if (a.num_ranges > 0) {
    if (!ranges_host || !a.src_ranges || a.num_ranges > 16) return HN_ERR_BAD_ARG;
    rows = 0;
    for (int k = 0; k < a.num_ranges; ++k) {
      const int lo = ranges_host[2 * k], hi = ranges_host[2 * k + 1];
      if (lo < 0 || hi > a.Nsrc || (k > 0 && lo < ranges_host[2 * k - 1])) return HN_ERR_BAD_ARG;
      if (hi > lo) rows += hi - lo;
    }
    if (rows == 0) return HN_OK;
  }
  if (rpb <= 0) {
    const long work = rows * ncb * a.T;
    const int cus = num_cus();
    long rounds = (work + (long)cus * 128) / ((long)cus * 256);
    if (rounds < 1) rounds = 1;
    long wgs_per_tc = cus * rounds / ((long)ncb * a.T);     // chunks per (t, cb), rounded DOWN: a grid one workgroup
    if (wgs_per_tc < 1) wgs_per_tc = 1;                      // over a whole round would double the kernel's time
    rpb = (int)((rows + wgs_per_tc - 1) / wgs_per_tc);
    if (rpb < 16) rpb = 16;
    if (a.num_ranges > 0) {
      // chunks do not straddle a range: every range rounds its own chunk count up, and a grid ONE workgroup over a whole
      // round (258 workgroups on 256 CUs, one resident each) takes two rounds -- measured on the self-peer step (round 6,
      // profiles/r06_selfpeer_trace.md): 320 us for 88 % of the rows where the unranged launch takes 275 us for all of them.
      // Grow the chunk until the ranges' chunks fit the rounds the launch was sized for.
      for (int it = 0; it < 64; ++it) {
        long c = 0;
        for (int k = 0; k < a.num_ranges; ++k) {
          const int lo = ranges_host[2 * k], hi = ranges_host[2 * k + 1];
          if (hi > lo) c += (hi - lo + rpb - 1) / rpb;
        }
        if (c <= wgs_per_tc) break;
        rpb += (rpb + 31) / 32;
      }
    }
  }
  a.rows_per_block = rpb;
  long chunks = (a.Nsrc + rpb - 1) / rpb;
  if (a.num_ranges > 0) {
    chunks = 0;
    for (int k = 0; k < a.num_ranges; ++k) {
      const int lo = ranges_host[2 * k], hi = ranges_host[2 * k + 1];
      if (hi > lo) chunks += (hi - lo + rpb - 1) / rpb;
    }
  }
  dim3 grid((unsigned)chunks, (unsigned)ncb, (unsigned)a.T);
  if (!windowed) {
    hipLaunchKernelGGL(k, grid, dim3(1024), lds, s, a);
  } else {
    a.win_base = 0; a.win_rows = split + HN_PAD; a.win_lo = 0; a.win_hi = split; a.win_accumulate = 0;
    hipLaunchKernelGGL(k, grid, dim3(1024), lds, s, a);
    a.win_base = split; a.win_rows = rows_all - split; a.win_lo = split; a.win_hi = 0x7fffffff; a.win_accumulate = 1;
    hipLaunchKernelGGL(k, grid, dim3(1024), lds, s, a);
  }
  if (a.gx == nullptr) return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;   // (left to the consumer)
  // partial sums over the relations + identity terms -> gvec, gx
  const long n4v = has_vec ? (long)a.Nsrc * 3 * a.H / 4 : 0, n4x = (long)a.Nsrc * a.H / 4;
  const long n4 = n4v > n4x ? n4v : n4x;
  hipLaunchKernelGGL(message_bwd_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                     has_vec ? reinterpret_cast<const float4*>(a.gvec) : nullptr, a.T, n4v,
                     reinterpret_cast<const float4*>(a.gvec1), reinterpret_cast<const float4*>(a.gx1),
                     a.type_rowptr, a.identity, a.H, reinterpret_cast<float4*>(a.gvec_out),
                     reinterpret_cast<float4*>(a.gx), n4x, a.src_ranges, a.num_ranges);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_edge_radial_table(const hn_graph* g, const hn_rbf_desc* rbf, const float* edge, float* table,
                                         void* stream) {
  if (!g || !rbf || rbf->num_rbf < 2 || g->num_edges < 0) return HN_ERR_BAD_ARG;
  const int num_edges = g->num_edges;
  if (num_edges == 0) return HN_OK;
  if (!edge || !table || !rbf->offset || !g->csc_pos || !g->csc_rowptr) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(edge_table_kernel, dim3((num_edges + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(edge), g->csc_pos, num_edges, rbf->offset, rbf->num_rbf, rbf->inv_rc,
                     rbf->coeff, rbf->env_kind, rbf->env_p, table,
                     g->csc_rowptr + (size_t)g->num_rel * (size_t)(g->num_src > 0 ? g->num_src : g->num_nodes));
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}
