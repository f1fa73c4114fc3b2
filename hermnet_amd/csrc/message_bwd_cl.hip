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
//   * per lane there is one channel of state, ~100 VGPRs in all, so a 1024-thread workgroup puts 4 waves on every
//     SIMD (the VW = 4 form: 245 VGPRs, 2 waves per SIMD, each wave parked ~50 % of its life -- rocprofv3
//     SQ_WAIT_ANY / SQ_WAVE_CYCLES, profiles/r02_v1_counters.json -- and a lone wave issues fp32 VALU at only
//     ~40 % of the SIMD's rate);
//   * segment sums (gxh, gvec) need no cross-lane step at all; the per-edge dE/dD (a sum over channels) is reduced
//     for 4 edges at once with v_permlane32/16 swaps + 4 DPP steps;
//   * source rows are handed to waves dynamically (LDS counter), so a workgroup's 16 waves stay balanced although
//     segments are short (~14 edges) and uneven.
//
// No atomics on HBM, fixed summation order: bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hermnet_hip.h"
#include "hermnet_math.h"
#include "message_bwd_cl.h"

namespace {

typedef float hn_f2 __attribute__((ext_vector_type(2)));
typedef unsigned hn_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(4))) const float hn_cfloat;   // constant address space: uniform loads go scalar

constexpr int kRec = HN_EDGE_TABLE_FLOATS;   // floats per edge record
// record layout (m < 12, g_m = exp(coeff (u - mu_m)^2)):
//   [2m]   = env(u) g_m                                   -> sum_m [2m]   W_m = rbfh - bias            (rmnet.py:55,168-172)
//   [2m+1] = (env'(u) g_m + 2 coeff env(u) g_m (u - mu_m)) / rc  -> sum_m [2m+1] W_m = d rbfh / d d
//   [24] padded tile row of tap 0 (int bits) | [25] env | [26] env'/rc | [27] 2 coeff env / rc | [28..30] rhat | [31] 1/d
// The envelope factors are folded into the tap pairs, so the contraction yields rbfh and its distance derivative
// directly (no per-channel envelope arithmetic in the message kernel).

__global__ __launch_bounds__(256) void edge_table_kernel(const float4* __restrict__ edge, int E,
                                                         const float* __restrict__ offset, int R, float inv_rc,
                                                         float coeff, int env_kind, int env_p,
                                                         float* __restrict__ table) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float4 g = edge[e];
  const float u = g.w * inv_rc;
  const HnEnv env = hn_envelope(u, env_kind, env_p);
  const int lo = hn_window_lo(u, R);
  float rec[kRec];
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
  rec[25] = env.val;
  rec[26] = c0;
  rec[27] = c1;
  rec[28] = g.x; rec[29] = g.y; rec[30] = g.z;
  rec[31] = __builtin_amdgcn_rcpf(g.w);
  float4* out = reinterpret_cast<float4*>(table + (size_t)e * kRec);
#pragma unroll
  for (int q = 0; q < kRec / 4; ++q) out[q] = make_float4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
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

// The 12 tap rows of this lane's weight record are fetched in three groups of four, each row ONE ds_read_b128
// (4 LDS cycles, conflict-free), two groups in flight: group g+1 is requested before group g is waited for, so only
// the first group's LDS latency is exposed.  Written as asm because (i) hipcc narrows a float4 LDS load whose .w is
// unused to ds_read_b96 (8 LDS cycles; the LDS port is the second-busiest unit of this kernel) and (ii) the counted
// waits must sit exactly between the groups.  No scalar-memory load is in flight while these run (the next edge's
// record is requested after the contraction), so lgkmcnt counts LDS reads only and they return in order.
// Every wait statement takes the registers it releases as "+v" operands: no use can be scheduled in front of it.
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
template <int PENDING>   // wait until at most PENDING newer LDS reads are outstanding; releases w
__device__ __forceinline__ void lds_wait(hn_f4 (&w)[4]) {
  if (PENDING == 4)
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) :: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) :: "memory");
}
static_assert(HN_CB * 16 == 1024 && HN_TAPS == 12, "the tap reader hard-codes the 1 KiB row pitch and 3 x 4 taps");

typedef __amdgpu_buffer_rsrc_t hn_rsrc;
__device__ __forceinline__ float buf_load(hn_rsrc r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
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
template <bool HAS_VEC>
__global__ __launch_bounds__(1024, 4) void message_scatter_bwd_cl_kernel(HnBwdClArgs a) {
  extern __shared__ __align__(16) float4 tile[];     // [tap row][64] of (s, a, b, 0)
  const int tile_rows = a.R + 2 * HN_PAD + 1;

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
  const int t = bz;
  const int cb = by;
  const int r0 = bx * a.rows_per_block;
  const int r1 = min(r0 + a.rows_per_block, a.Nsrc);
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
    const int k = idx / HN_CB - HN_PAD, ch = idx % HN_CB;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k >= 0 && k < a.R) {
      const float* w = a.wt + ((size_t)(t * a.R + k) * 3 * H + cb * HN_CB + ch);
      v = make_float4(w[0], w[H], w[2 * H], 0.f);
    }
    tile[idx] = v;
  }

  const float* xh_t = a.xh + (size_t)t * a.Nsrc * 3 * H;
  float* gxh_t = a.gxh + (size_t)t * a.Nsrc * 3 * H;
  float* gvec_t = HAS_VEC ? a.gvec + (size_t)t * a.Nsrc * 3 * H : nullptr;
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
    // row pointers of up to 64 rows at a time live in one VGPR (lane q: rowptr[row_base + q]); v_readlane on demand
    int row_base = ra;
    int rp_vec = rowptr_t[min(row_base + lane, a.Nsrc)];
    auto rowptr_of = [&](int r) {
      if (r - row_base >= 64) {
        row_base = r;
        rp_vec = rowptr_t[min(row_base + lane, a.Nsrc)];
      }
      return __builtin_amdgcn_readlane(rp_vec, r - row_base);
    };

    int row = ra;
    const int e_begin = rowptr_of(ra);
    const int e_end = rowptr_t[rb];                  // (scalar load, once)
    int row_end = rowptr_of(row + 1);

    // row values, pre-multiplied by the constant factors they always meet (rmnet.py:24,63-66): xs / sqrt2,
    // xa / sqrt(3H) (and xa itself), xb / sqrt(H), vec_j / sqrt(3H); gs and gb are rescaled once per row
    float xs, xa, xas, xb, vj0 = 0.f, vj1 = 0.f, vj2 = 0.f;
    float gs = 0.f, ga = 0.f, gb = 0.f, gv0 = 0.f, gv1 = 0.f, gv2 = 0.f;
    auto row_prologue = [&](int r) {
      // (wave-uniform base + this lane's channel: the base stays in SGPRs, one lane-offset register serves all arrays)
      const float* xr = xh_t + (size_t)r * 3 * H;
      xs = (xr[c] + xbs) * inv_sqrt2; xa = (xr + H)[c] + xba; xb = ((xr + 2 * H)[c] + xbb) * inv_sqrth;
      xas = xa * inv_sqrt3h;
      if (HAS_VEC) {
        const float* vr = a.vec + (size_t)r * 3 * H;
        vj0 = vr[c] * inv_sqrt3h; vj1 = (vr + H)[c] * inv_sqrt3h; vj2 = (vr + 2 * H)[c] * inv_sqrt3h;
      }
      gs = 0.f; ga = 0.f; gb = 0.f; gv0 = 0.f; gv1 = 0.f; gv2 = 0.f;
    };
    auto row_epilogue = [&](int r) {                 // every lane owns its channel: plain coalesced stores
      float* go = gxh_t + (size_t)r * 3 * H;
      go[c] = gs * inv_sqrt2; (go + H)[c] = ga; (go + 2 * H)[c] = gb * inv_sqrth;
      if (HAS_VEC) {
        float* gvo = gvec_t + (size_t)r * 3 * H;
        gvo[c] = gv0; (gvo + H)[c] = gv1; (gvo + 2 * H)[c] = gv2;
      }
    };
    row_prologue(row);

    for (int base = e_begin; base < e_end; base += 64) {
      const int cnt = min(64, e_end - base);
      // one coalesced index load per 64 edges; an edge's indices are then wave-uniform (v_readlane)
      const int my_tgt = lane < cnt ? a.csc_tgt[base + lane] : 0;
      const int my_pos = lane < cnt ? a.csc_pos[base + lane] : 0;

      auto load_edge = [&](int k) {
        const int i = __builtin_amdgcn_readlane(my_tgt, k);
        EdgeIn in;
        // buffer loads: descriptor + scalar row offset + this lane's channel offset -- no per-lane 64-bit address math
        const unsigned so = (unsigned)i * (unsigned)(H * 4);
        in.gx1 = buf_load(rs_gx1, c4, so);
        in.g0 = buf_load(rs_gvec1, c4, 3u * so);
        in.g1 = buf_load(rs_gvec1, c4, 3u * so + (unsigned)(H * 4));
        in.g2 = buf_load(rs_gvec1, c4, 3u * so + (unsigned)(2 * H * 4));
        return in;
      };
      // the record of an edge is wave-uniform and read-only here: through the constant address space its loads
      // are scalar (s_load_dwordx8 into SGPRs), not 64 lanes fetching the same bytes
      auto load_record = [&](int k, float (&rec)[kRec]) {
        const int p = __builtin_amdgcn_readlane(my_pos, k);
        const hn_cfloat* rp = (const hn_cfloat*)(a.table + (size_t)p * kRec);
#pragma unroll
        for (int q = 0; q < kRec; ++q) rec[q] = rp[q];
      };

      EdgeIn cur = load_edge(0);
      float rec[kRec];
      load_record(0, rec);
      for (int k4 = 0; k4 < cnt; k4 += 4) {
        float pd[2], px[2], py[2], pz[2];                  // the current pair's per-channel dE/dd, dE/drhat
        float ud[2], ux[2], uy[2], uz[2];                  // folded pairs (32-lane partial sums)
        // geometry of the edge this DPP row will write (rows hold edges 0, 2, 1, 3 of the group): requested now,
        // used after the group's arithmetic
        const int je = (row16 == 0) ? 0 : (row16 == 1 ? 2 : (row16 == 2 ? 1 : 3));
        const int pw = __shfl(my_pos, min(k4 + je, cnt - 1), 64);
        const float4 gw = a.edge[pw];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pd[j & 1] = 0.f; px[j & 1] = 0.f; py[j & 1] = 0.f; pz[j & 1] = 0.f;
          const int k = k4 + j;
          if (k < cnt) {                                          // wave-uniform
            while (base + k >= row_end) {                         // (wave-uniform) the stream enters the next row
              row_epilogue(row);
              ++row;
              row_end = rowptr_of(row + 1);
              row_prologue(row);
            }
            const EdgeIn nxt = load_edge(min(k + 1, cnt - 1));    // the next edge's rows fly during this edge's math
            // ---- (S0, S1) of the three parts: 12 taps, one LDS read each, in three pipelined groups
            const unsigned waddr = (unsigned)__float_as_int(rec[24]) * (HN_CB * 16) + tile_lane;
            hn_f2 Ss = {0.f, 0.f}, Sa = {0.f, 0.f}, Sb = {0.f, 0.f};
            hn_f4 wA[4], wB[4];
            auto taps4 = [&](int m0, const hn_f4 (&w)[4]) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const hn_f2 g = {rec[2 * (m0 + q)], rec[2 * (m0 + q) + 1]};
                Ss = __builtin_elementwise_fma(g, hn_f2{w[q].x, w[q].x}, Ss);
                if (HAS_VEC) Sa = __builtin_elementwise_fma(g, hn_f2{w[q].y, w[q].y}, Sa);
                Sb = __builtin_elementwise_fma(g, hn_f2{w[q].z, w[q].z}, Sb);
              }
            };
            lds_issue4(waddr, wA);
            lds_issue4(waddr + 4 * (HN_CB * 16), wB);
            lds_wait<4>(wA);
            taps4(0, wA);
            asm volatile("" : "+v"(Ss), "+v"(Sb));                // taps 0-3 are consumed before wA is refilled
            lds_issue4(waddr + 8 * (HN_CB * 16), wA);
            lds_wait<4>(wB);
            taps4(4, wB);
            lds_wait<0>(wA);
            taps4(8, wA);
            const float rx = rec[28], ry = rec[29], rz = rec[30];
            // the taps are consumed: request the NEXT edge's record into the same scalar registers; its latency
            // hides behind the rest of this edge (the six scalars still needed were copied above)
            __builtin_amdgcn_sched_barrier(0);
            load_record(min(k + 1, cnt - 1), rec);
            __builtin_amdgcn_sched_barrier(0);
            const float gx1 = cur.gx1, g0 = cur.g0, g1 = cur.g1, g2 = cur.g2;
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
            const float q = xb * rb_;
            pd[j & 1] = pdv; px[j & 1] = g0 * q; py[j & 1] = g1 * q; pz[j & 1] = g2 * q;
            cur = nxt;
          }
          if (j & 1) {
            ud[j >> 1] = pair_fold(pd[0], pd[1]); ux[j >> 1] = pair_fold(px[0], px[1]);
            uy[j >> 1] = pair_fold(py[0], py[1]); uz[j >> 1] = pair_fold(pz[0], pz[1]);
          }
        }
        // ---- dE/dD of these (up to) four edges: channel sums, then Cartesian form, one 16-byte store per edge
        const float sd = quad_total(ud[0], ud[1]), sx = quad_total(ux[0], ux[1]);
        const float sy = quad_total(uy[0], uy[1]), sz = quad_total(uz[0], uz[1]);
        if ((lane & 15) == 0 && k4 + je < cnt) {
          const float invd = __builtin_amdgcn_rcpf(gw.w);
          const float dotp = sx * gw.x + sy * gw.y + sz * gw.z;
          // d = |D|, rhat = D/d:  gD = pd rhat + (pr - (pr.rhat) rhat) / d
          const float tpar = sd - dotp * invd;
          gedge[pw] = make_float4(fmaf(tpar, gw.x, sx * invd), fmaf(tpar, gw.y, sy * invd), fmaf(tpar, gw.z, sz * invd), 0.f);
        }
      }
    }
    // ---- the stream is exhausted: finish the current row and the rows without edges behind it
    for (;;) {
      row_epilogue(row);
      if (++row >= rb) break;
      row_prologue(row);
    }
  }
}

// gvec[r] = sum_t part[t][r] (+ gvec1[r] on target rows), gx[r] = gx1[r] / sqrt2 on target rows, 0 elsewhere.
__global__ __launch_bounds__(256) void message_bwd_finish_kernel(const float4* __restrict__ part, int T, long n4_vec,
                                                                 const float4* __restrict__ gvec1,
                                                                 const float4* __restrict__ gx1,
                                                                 const int* __restrict__ type_rowptr, int identity,
                                                                 int H, float4* __restrict__ gvec, float4* __restrict__ gx,
                                                                 long n4_x) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int identity_rows = identity ? type_rowptr[T] : 0;      // rows below are targets of a known type
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

int hn_bwd_cl_launch(HnBwdClArgs a, bool has_vec, int rows_override, hipStream_t s) {
  const size_t lds = hn_bwd_cl_lds_bytes(a.R);
  if (lds > 160 * 1024) return HN_ERR_LDS;
  typedef void (*kern_t)(HnBwdClArgs);
  kern_t k = has_vec ? message_scatter_bwd_cl_kernel<true> : message_scatter_bwd_cl_kernel<false>;
  static bool done[2] = {false, false};
  if (!done[has_vec]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return HN_ERR_LDS;
    done[has_vec] = true;
  }
  // one (relation, column block, row chunk) per workgroup; the chunk is sized for a whole number of rounds of one
  // workgroup per CU (the 154 KB tile allows one resident workgroup): ~256 rows = 16 rows per wave
  const int ncb = a.H / HN_CB;
  int rpb = rows_override;
  if (rpb <= 0) {
    const long work = (long)a.Nsrc * ncb * a.T;
    const int cus = num_cus();
    long rounds = (work + (long)cus * 128) / ((long)cus * 256);
    if (rounds < 1) rounds = 1;
    long wgs_per_tc = cus * rounds / ((long)ncb * a.T);     // chunks per (t, cb), rounded DOWN: a grid one workgroup
    if (wgs_per_tc < 1) wgs_per_tc = 1;                      // over a whole round would double the kernel's time
    rpb = (int)((a.Nsrc + wgs_per_tc - 1) / wgs_per_tc);
    if (rpb < 16) rpb = 16;
  }
  a.rows_per_block = rpb;
  dim3 grid((unsigned)((a.Nsrc + rpb - 1) / rpb), (unsigned)ncb, (unsigned)a.T);
  hipLaunchKernelGGL(k, grid, dim3(1024), lds, s, a);
  // partial sums over the relations + identity terms -> gvec, gx
  const long n4v = has_vec ? (long)a.Nsrc * 3 * a.H / 4 : 0, n4x = (long)a.Nsrc * a.H / 4;
  const long n4 = n4v > n4x ? n4v : n4x;
  hipLaunchKernelGGL(message_bwd_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                     has_vec ? reinterpret_cast<const float4*>(a.gvec) : nullptr, a.T, n4v,
                     reinterpret_cast<const float4*>(a.gvec1), reinterpret_cast<const float4*>(a.gx1),
                     a.type_rowptr, a.identity, a.H, reinterpret_cast<float4*>(a.gvec_out),
                     reinterpret_cast<float4*>(a.gx), n4x);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_edge_radial_table(const hn_rbf_desc* rbf, const float* edge, int num_edges, float* table,
                                         void* stream) {
  if (!rbf || rbf->num_rbf < 2 || num_edges < 0) return HN_ERR_BAD_ARG;
  if (num_edges == 0) return HN_OK;
  if (!edge || !table || !rbf->offset) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(edge_table_kernel, dim3((num_edges + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(edge), num_edges, rbf->offset, rbf->num_rbf, rbf->inv_rc, rbf->coeff,
                     rbf->env_kind, rbf->env_p, table);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

