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
// Mapping (wave = 64 lanes): a wave owns one target (fwd) or source (bwd) row; its two 32-lane
// halves take alternate edges of the row's segment; a lane owns two adjacent channels (float2)
// of a 64-channel column block, so a half-wave moves one 256-byte row segment per load and reads
// the LDS weight tile with conflict-free ds_read_b64.  Halves are combined with one cross-half
// exchange per row; nothing is accumulated in HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/hermnet_hip.h"
#include "hermnet_math.h"

#define HN_WAVES 16                 // waves per workgroup (1024 threads, 4 per SIMD)
#define HN_THREADS (HN_WAVES * 64)
#define HN_LDS_ROW (3 * HN_CB)      // floats per tap row in LDS: [part][64]
#ifndef HN_TAP_UNROLL
#define HN_TAP_UNROLL 12
#endif

namespace {

struct MsgArgs {
  // graph
  int N, E, T;
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
  const float* gx1;    // bwd in
  const float* gvec1;  // bwd in
  float* gxh;          // bwd out [T,N,3H]
  float* gvec;         // bwd out [N,3,H] or null
  float* gx;           // bwd out [N,H]
  float4* gedge;       // bwd out [H/64,E]
  int rows_per_block;
};

__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }

// Sum over the two 32-lane halves of the wave (lane l <-> l^32).
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// Sum over the 32 lanes of each half; every lane of the half ends with the total.
__device__ __forceinline__ float half_allsum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

// Stage the zero-padded weight tile of (relation t, column block cb) and the tap centres into LDS.
//   wl[(k + HN_PAD) * 192 + part * 64 + c] = wt[t][k][part * H + cb * 64 + c]   (0 outside 0 <= k < R)
//   mu[k + HN_PAD] = offset[clamp(k)]
template <int NTHREADS>
__device__ __forceinline__ void stage_weights(const MsgArgs& a, int t, int cb, float* wl, float* mu) {
  const int rows = a.R + 2 * HN_PAD + 1;
  const int n4 = rows * (HN_LDS_ROW / 4);
  for (int idx = threadIdx.x; idx < n4; idx += NTHREADS) {
    const int kk = idx / (HN_LDS_ROW / 4);
    const int q = idx - kk * (HN_LDS_ROW / 4);
    const int part = q >> 4;          // 16 float4 per 64-channel part
    const int c4 = q & 15;
    const int k = kk - HN_PAD;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k >= 0 && k < a.R) {
      const float* src = a.wt + ((size_t)(t * a.R + k) * 3 * a.H + part * a.H + cb * HN_CB + c4 * 4);
      v = *reinterpret_cast<const float4*>(src);
    }
    *reinterpret_cast<float4*>(wl + kk * HN_LDS_ROW + part * HN_CB + c4 * 4) = v;
  }
  for (int kk = threadIdx.x; kk < rows; kk += NTHREADS) {
    int k = kk - HN_PAD;
    k = k < 0 ? 0 : (k >= a.R ? a.R - 1 : k);
    mu[kk] = a.offset[k];
  }
}

// Banded contraction for one edge: S0[part] = sum_m g_m W[lo+m][part], and (WITH_DER)
// S1[part] = sum_m g_m (u - mu_m) W[lo+m][part], g_m = exp(coeff (u - mu_m)^2)  (rmnet.py:156-172).
template <bool WITH_DER>
__device__ __forceinline__ void banded_rbf(const float* wl, const float* mu, int lo, float u, float coeff,
                                           int hl, float2 (&S0)[3], float2 (&S1)[3]) {
#pragma unroll
  for (int p = 0; p < 3; ++p) { S0[p] = make_float2(0.f, 0.f); if (WITH_DER) S1[p] = make_float2(0.f, 0.f); }
  const float* wrow = wl + (lo + HN_PAD) * HN_LDS_ROW + 2 * hl;
  const float* mrow = mu + (lo + HN_PAD);
#pragma unroll HN_TAP_UNROLL
  for (int m = 0; m < HN_TAPS; ++m) {
    const float diff = u - mrow[m];
    const float g = __expf(coeff * (diff * diff));
    const float gd = g * diff;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const float2 w = ld2(wrow + m * HN_LDS_ROW + p * HN_CB);
      S0[p].x = fmaf(g, w.x, S0[p].x);
      S0[p].y = fmaf(g, w.y, S0[p].y);
      if (WITH_DER) {
        S1[p].x = fmaf(gd, w.x, S1[p].x);
        S1[p].y = fmaf(gd, w.y, S1[p].y);
      }
    }
  }
}

// Decode blockIdx.x -> (relation t, first row, one-past-last row) for blocks of `rpb` rows
// laid out relation after relation.  Returns -1 for a work block, else the index of the
// surplus block (0, 1, ...) past the last work block.
__device__ __forceinline__ int decode_block(const MsgArgs& a, int bx, int& t, int& r0, int& r1) {
  for (t = 0; t < a.T; ++t) {
    const int lo = a.type_rowptr[t], hi = a.type_rowptr[t + 1];
    const int nb = (hi - lo + a.rows_per_block - 1) / a.rows_per_block;
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
template <bool HAS_VEC>
__global__ __launch_bounds__(HN_THREADS, 4) void message_scatter_fwd_kernel(MsgArgs a) {
  extern __shared__ __align__(16) float lds[];
  float* wl = lds;
  float* mu = lds + (a.R + 2 * HN_PAD + 1) * HN_LDS_ROW;

  const int cb = blockIdx.y;
  int t, r0, r1;
  const int surplus = decode_block(a, blockIdx.x, t, r0, r1);
  if (surplus >= 0) {
    // the first surplus block zeroes the rows of unknown-type atoms (they are never targets)
    if (surplus == 0) {
      const int c = cb * HN_CB + (threadIdx.x & 63);
      for (int r = a.type_rowptr[a.T] + (int)(threadIdx.x >> 6); r < a.N; r += HN_WAVES) {
        a.x1[(size_t)r * a.H + c] = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) a.vec1[((size_t)r * 3 + d) * a.H + c] = 0.f;
      }
    }
    return;
  }
  stage_weights<HN_THREADS>(a, t, cb, wl, mu);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int half = lane >> 5;
  const int hl = lane & 31;
  const int H = a.H;
  const int col = cb * HN_CB + 2 * hl;   // first of this lane's two channels
  const float inv_sqrt3h = 0.57735026918962576f * rsqrtf((float)H);  // (1/sqrt3)(1/sqrtH)
  const float inv_sqrth = rsqrtf((float)H);
  const float* xh_t = a.xh + (size_t)t * a.N * 3 * H;

  float2 bias[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) bias[p] = ld2(a.brbf + (size_t)t * 3 * H + p * H + col);

  for (int r = r0 + wave; r < r1; r += HN_WAVES) {
    const int beg = a.csr_rowptr[r], end = a.csr_rowptr[r + 1];
    float2 ax = make_float2(0.f, 0.f);
    float2 av[3] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};

    for (int e = beg + half; e < end; e += 2) {
      const int j = a.csr_src[e];
      const float4 g = a.edge[e];
      const float* xr = xh_t + (size_t)j * 3 * H + col;
      const float2 xs = ld2(xr), xa = ld2(xr + H), xb = ld2(xr + 2 * H);
      float2 vj[3];
      if (HAS_VEC) {
        const float* vr = a.vec + (size_t)j * 3 * H + col;
#pragma unroll
        for (int d = 0; d < 3; ++d) vj[d] = ld2(vr + d * H);
      }
      const float u = g.w * a.inv_rc;
      const HnEnv env = hn_envelope(u, a.env_kind, a.env_p);
      const int lo = hn_window_lo(u, a.R);
      float2 S0[3], S1[3];
      banded_rbf<false>(wl, mu, lo, u, a.coeff, hl, S0, S1);
      // rbfh = bias + env * S0 (rmnet.py:55); message (rmnet.py:61-67)
      float2 rs, ra, rb;
      rs.x = fmaf(env.val, S0[0].x, bias[0].x); rs.y = fmaf(env.val, S0[0].y, bias[0].y);
      ra.x = fmaf(env.val, S0[1].x, bias[1].x); ra.y = fmaf(env.val, S0[1].y, bias[1].y);
      rb.x = fmaf(env.val, S0[2].x, bias[2].x); rb.y = fmaf(env.val, S0[2].y, bias[2].y);
      ax.x = fmaf(xs.x, rs.x, ax.x);
      ax.y = fmaf(xs.y, rs.y, ax.y);
      const float2 mb = make_float2(xb.x * rb.x * inv_sqrth, xb.y * rb.y * inv_sqrth);
      const float rd[3] = {g.x, g.y, g.z};
      if (HAS_VEC) {
        const float2 ma = make_float2(xa.x * ra.x * inv_sqrt3h, xa.y * ra.y * inv_sqrt3h);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          av[d].x = fmaf(vj[d].x, ma.x, fmaf(mb.x, rd[d], av[d].x));
          av[d].y = fmaf(vj[d].y, ma.y, fmaf(mb.y, rd[d], av[d].y));
        }
      } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          av[d].x = fmaf(mb.x, rd[d], av[d].x);
          av[d].y = fmaf(mb.y, rd[d], av[d].y);
        }
      }
    }
    // combine the two halves, then residual epilogue (rmnet.py:24-26); half 0 writes x1 and
    // vec1[0], half 1 writes vec1[1], vec1[2].
    ax.x = xhalf_sum(ax.x); ax.y = xhalf_sum(ax.y);
#pragma unroll
    for (int d = 0; d < 3; ++d) { av[d].x = xhalf_sum(av[d].x); av[d].y = xhalf_sum(av[d].y); }
    const size_t xo = (size_t)r * H + col;
    const size_t vo = (size_t)r * 3 * H + col;
    if (half == 0) {
      const float2 x0 = ld2(a.x + xo);
      st2(a.x1 + xo, make_float2((x0.x + ax.x) * 0.70710678118654752f, (x0.y + ax.y) * 0.70710678118654752f));
      float2 v0 = make_float2(0.f, 0.f);
      if (HAS_VEC) v0 = ld2(a.vec + vo);
      st2(a.vec1 + vo, make_float2(v0.x + av[0].x, v0.y + av[0].y));
    } else {
      float2 v1 = make_float2(0.f, 0.f), v2 = make_float2(0.f, 0.f);
      if (HAS_VEC) { v1 = ld2(a.vec + vo + H); v2 = ld2(a.vec + vo + 2 * H); }
      st2(a.vec1 + vo + H, make_float2(v1.x + av[1].x, v1.y + av[1].y));
      st2(a.vec1 + vo + 2 * H, make_float2(v2.x + av[2].x, v2.y + av[2].y));
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward (forces): one workgroup = (column block, chunk of SOURCE rows); it loops over the
// relations of the targets, restaging the weight tile per relation.  Sums keyed by the source row
// stay in registers for the whole CSC segment: gxh[t][row] is written once per relation; gvec[row]
// is written at t = 0 (with the residual's identity term) and read-modify-written by the same
// lanes for t > 0.  Per-edge cross-lane sums (dE/dD, 3 floats) use DPP inside 16-lane rows.
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// Sum over the 32 lanes of each half-wave; every lane ends with its half's total.
__device__ __forceinline__ float half_allsum_dpp(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
  v += dpp_mov<0x141>(v);   // row_half_mirror      : other quad of the 8-lane group
  v += dpp_mov<0x140>(v);   // row_mirror           : other half of the 16-lane row
  v += __shfl_xor(v, 16, 64);
  return v;
}

struct BwdIn {
  float4 g;       // (rx, ry, rz, d)
  float2 gx1;     // d/dx1 of the target row
  float2 gd[3];   // d/dvec1 of the target row
  int pos;
  float lv;       // 1 for a live edge, 0 for the padding half of an odd segment
};

template <bool HAS_VEC, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void message_scatter_bwd_kernel(MsgArgs a) {
  extern __shared__ __align__(16) float lds[];
  float* wl = lds;
  float* mu = lds + (a.R + 2 * HN_PAD + 1) * HN_LDS_ROW;

  const int cb = blockIdx.y;
  const int r0 = blockIdx.x * a.rows_per_block;
  const int r1 = min(r0 + a.rows_per_block, a.N);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int half = lane >> 5;
  const int hl = lane & 31;
  const int H = a.H;
  const int col = cb * HN_CB + 2 * hl;
  const int nk = a.type_rowptr[a.T];     // rows below nk have a known type (are targets)
  const float inv_sqrt3h = 0.57735026918962576f * rsqrtf((float)H);
  const float inv_sqrth = rsqrtf((float)H);
  const float inv_sqrt2 = 0.70710678118654752f;
  float4* gedge = a.gedge + (size_t)cb * a.E;

  for (int t = 0; t < a.T; ++t) {
    __syncthreads();   // previous tile no longer in use
    stage_weights<NW * 64>(a, t, cb, wl, mu);
    __syncthreads();
    const float* xh_t = a.xh + (size_t)t * a.N * 3 * H;
    float* gxh_t = a.gxh + (size_t)t * a.N * 3 * H;
    float2 bias[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bias[p] = ld2(a.brbf + (size_t)t * 3 * H + p * H + col);

    for (int r = r0 + wave; r < r1; r += NW) {
      const int beg = a.csc_rowptr[(size_t)t * a.N + r], end = a.csc_rowptr[(size_t)t * a.N + r + 1];
      const float* xr = xh_t + (size_t)r * 3 * H + col;
      const float2 xs = ld2(xr), xa = ld2(xr + H), xb = ld2(xr + 2 * H);
      float2 vj[3] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
      if (HAS_VEC) {
        const float* vr = a.vec + (size_t)r * 3 * H + col;
#pragma unroll
        for (int d = 0; d < 3; ++d) vj[d] = ld2(vr + d * H);
      }
      float2 gs = make_float2(0.f, 0.f), ga = make_float2(0.f, 0.f), gb = make_float2(0.f, 0.f);
      float2 gv[3] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};

      for (int base = beg; base < end; base += 64) {
        const int cnt = min(64, end - base);
        // one coalesced index load per 64 edges; pairs pick their entries with a lane shuffle
        const int my_tgt = lane < cnt ? a.csc_tgt[base + lane] : 0;
        const int my_pos = lane < cnt ? a.csc_pos[base + lane] : 0;
        const int npair = (cnt + 1) >> 1;

        auto load_pair = [&](int it) {
          BwdIn in;
          const int idx = 2 * it + half;
          const int i = __shfl(my_tgt, idx, 64);
          in.pos = __shfl(my_pos, idx, 64);
          in.lv = idx < cnt ? 1.0f : 0.0f;
          in.g = a.edge[in.pos];
          in.gx1 = ld2(a.gx1 + (size_t)i * H + col);
          const float* gvr = a.gvec1 + (size_t)i * 3 * H + col;
#pragma unroll
          for (int d = 0; d < 3; ++d) in.gd[d] = ld2(gvr + d * H);
          return in;
        };

        BwdIn cur = load_pair(0);
        for (int it = 0; it < npair; ++it) {
          BwdIn nxt = cur;
          if (it + 1 < npair) nxt = load_pair(it + 1);   // wave-uniform: prefetch the next pair

          const float4 g = cur.g;
          const float u = g.w * a.inv_rc;
          const HnEnv env = hn_envelope(u, a.env_kind, a.env_p);
          const int lo = hn_window_lo(u, a.R);
          float2 S0[3], S1[3];
          banded_rbf<true>(wl, mu, lo, u, a.coeff, hl, S0, S1);
          const float lv = cur.lv;
          // d rbfh / d d = inv_rc * (env' S0 + env * 2 coeff S1)
          const float c0 = a.inv_rc * env.der, c1 = a.inv_rc * env.val * 2.0f * a.coeff;
          const float rd[3] = {g.x, g.y, g.z};
          float pd = 0.f;                    // partial dE/dd over this lane's channels
          float pr[3] = {0.f, 0.f, 0.f};     // partial dE/d rhat
#define HN_BWD_CH(C)                                                                         \
          {                                                                                  \
            const float rs = fmaf(env.val, S0[0].C, bias[0].C);                              \
            const float ra = fmaf(env.val, S0[1].C, bias[1].C);                              \
            const float rb = fmaf(env.val, S0[2].C, bias[2].C);                              \
            const float drs = fmaf(c0, S0[0].C, c1 * S1[0].C);                               \
            const float dra = fmaf(c0, S0[1].C, c1 * S1[1].C);                               \
            const float drb = fmaf(c0, S0[2].C, c1 * S1[2].C);                               \
            const float gdx = cur.gx1.C * inv_sqrt2 * lv;                                    \
            const float g0 = cur.gd[0].C * lv, g1 = cur.gd[1].C * lv, g2 = cur.gd[2].C * lv; \
            const float A = (g0 * vj[0].C + g1 * vj[1].C + g2 * vj[2].C) * inv_sqrt3h;       \
            const float B = (g0 * rd[0] + g1 * rd[1] + g2 * rd[2]) * inv_sqrth;              \
            gs.C = fmaf(gdx, rs, gs.C);                                                      \
            ga.C = fmaf(A, ra, ga.C);                                                        \
            gb.C = fmaf(B, rb, gb.C);                                                        \
            const float w = xa.C * ra * inv_sqrt3h;                                          \
            gv[0].C = fmaf(g0, w, gv[0].C);                                                  \
            gv[1].C = fmaf(g1, w, gv[1].C);                                                  \
            gv[2].C = fmaf(g2, w, gv[2].C);                                                  \
            pd += gdx * xs.C * drs + A * xa.C * dra + B * xb.C * drb;                        \
            const float q = xb.C * rb * inv_sqrth;                                           \
            pr[0] = fmaf(g0, q, pr[0]); pr[1] = fmaf(g1, q, pr[1]); pr[2] = fmaf(g2, q, pr[2]); \
          }
          HN_BWD_CH(x)
          HN_BWD_CH(y)
#undef HN_BWD_CH
          // Cartesian gradient w.r.t. D (d = |D|, rhat = D/d): gD = pd rhat + (pr - (pr.rhat) rhat)/d
          const float dotp = pr[0] * rd[0] + pr[1] * rd[1] + pr[2] * rd[2];
          const float invd = 1.0f / g.w;
          float gD[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) gD[d] = half_allsum_dpp(fmaf(pd - dotp * invd, rd[d], pr[d] * invd));
          if (lv != 0.0f && hl == 0) gedge[cur.pos] = make_float4(gD[0], gD[1], gD[2], 0.f);
          cur = nxt;
        }
      }
      // combine halves; half 0 stores this relation's gxh row, gvec/gx are split over both halves
      gs.x = xhalf_sum(gs.x); gs.y = xhalf_sum(gs.y);
      ga.x = xhalf_sum(ga.x); ga.y = xhalf_sum(ga.y);
      gb.x = xhalf_sum(gb.x); gb.y = xhalf_sum(gb.y);
      if (half == 0) {
        float* go = gxh_t + (size_t)r * 3 * H + col;
        st2(go, gs); st2(go + H, ga); st2(go + 2 * H, gb);
      }
      const bool known = r < nk;
      if (t == 0 && half == 0) {   // residual identity: gx = gx1 / sqrt2 on rows that are targets
        const float2 g1 = known ? ld2(a.gx1 + (size_t)r * H + col) : make_float2(0.f, 0.f);
        st2(a.gx + (size_t)r * H + col, make_float2(g1.x * inv_sqrt2, g1.y * inv_sqrt2));
      }
      if (HAS_VEC) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { gv[d].x = xhalf_sum(gv[d].x); gv[d].y = xhalf_sum(gv[d].y); }
        const size_t vo = (size_t)r * 3 * H + col;
        // component 0 by half 0, components 1 and 2 by half 1
        const int d_lo = half == 0 ? 0 : 1, d_hi = half == 0 ? 1 : 3;
        for (int d = d_lo; d < d_hi; ++d) {
          float2 prev;
          if (t == 0) prev = known ? ld2(a.gvec1 + vo + d * H) : make_float2(0.f, 0.f);
          else prev = ld2(a.gvec + vo + d * H);
          const float2 add = d == 0 ? gv[0] : (d == 1 ? gv[1] : gv[2]);
          st2(a.gvec + vo + d * H, make_float2(prev.x + add.x, prev.y + add.y));
        }
      }
    }
  }
}

int fill_args(const hn_graph* g, const hn_rbf_desc* rbf, int hidden, MsgArgs& a) {
  if (!g || !rbf || hidden <= 0 || hidden % HN_CB != 0) return HN_ERR_BAD_ARG;
  if (g->num_rel <= 0 || rbf->num_rbf < 2) return HN_ERR_BAD_ARG;
  a.N = g->num_nodes; a.E = g->num_edges; a.T = g->num_rel;
  a.type_rowptr = g->type_rowptr; a.csr_rowptr = g->csr_rowptr; a.csr_src = g->csr_src;
  a.csc_rowptr = g->csc_rowptr; a.csc_tgt = g->csc_tgt; a.csc_pos = g->csc_pos;
  a.offset = rbf->offset; a.R = rbf->num_rbf; a.inv_rc = rbf->inv_rc; a.coeff = rbf->coeff;
  a.env_kind = rbf->env_kind; a.env_p = rbf->env_p;
  a.H = hidden;
  return HN_OK;
}

// tuning knobs (environment, read once): HERMNET_BWD_WAVES = 8|16, HERMNET_ROWS_PER_BLOCK
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
int bwd_waves() { int w = env_int("HERMNET_BWD_WAVES", 8); return w == 16 ? 16 : 8; }

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
// `slack` = workgroups lost to per-relation rounding.  Targets ~32 rows per workgroup.
int pick_rows(int rows, int ncb, int slack, int override_rows) {
  if (override_rows > 0) return override_rows;
  const long work = (long)rows * ncb;
  const int cus = num_cus();
  long rounds = (work + (long)cus * 16) / ((long)cus * 32);
  if (rounds < 1) rounds = 1;
  long wgs = (long)cus * rounds - slack;
  if (wgs < 1) wgs = 1;
  long rpb = (work + wgs - 1) / wgs;
  if (rpb < 8) rpb = 8;
  return (int)rpb;
}

size_t lds_bytes(int R) {
  return (size_t)(R + 2 * HN_PAD + 1) * (HN_LDS_ROW + 1) * sizeof(float);
}

}  // namespace

extern "C" int hermnet_message_scatter_fwd(const hn_graph* g, const hn_rbf_desc* rbf, int hidden,
                                           const float* xh, const float* vec, const float* x,
                                           const float* wt, const float* brbf, const float* edge,
                                           float* x1, float* vec1, void* stream) {
  MsgArgs a = {};
  int rc = fill_args(g, rbf, hidden, a);
  if (rc) return rc;
  if (!xh || !x || !wt || !brbf || !x1 || !vec1 || (g->num_edges > 0 && !edge)) return HN_ERR_BAD_ARG;
  if (a.N == 0) return HN_OK;
  a.xh = xh; a.vec = vec; a.x = x; a.wt = wt; a.brbf = brbf;
  a.edge = reinterpret_cast<const float4*>(edge);
  a.x1 = x1; a.vec1 = vec1;
  static const int rpb_fwd = env_int("HERMNET_FWD_ROWS", 0);
  a.rows_per_block = pick_rows(a.N, hidden / HN_CB, a.T * (hidden / HN_CB), rpb_fwd);
  const size_t lds = lds_bytes(a.R);
  if (lds > 160 * 1024) return HN_ERR_LDS;
  // blocks: sum_t ceil(N_t / rpb) <= N / rpb + T, plus one surplus block that zeroes unknown rows
  dim3 grid((unsigned)(a.N / a.rows_per_block + a.T + 1), (unsigned)(hidden / HN_CB));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  auto k = vec ? message_scatter_fwd_kernel<true> : message_scatter_fwd_kernel<false>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess) return HN_ERR_LDS;
  hipLaunchKernelGGL(k, grid, dim3(HN_THREADS), lds, s, a);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_message_scatter_bwd(const hn_graph* g, const hn_rbf_desc* rbf, int hidden,
                                           const float* xh, const float* vec,
                                           const float* wt, const float* brbf, const float* edge,
                                           const float* gx1, const float* gvec1,
                                           float* gxh, float* gvec, float* gx, float* gedge,
                                           void* stream) {
  MsgArgs a = {};
  int rc = fill_args(g, rbf, hidden, a);
  if (rc) return rc;
  if (!xh || !wt || !brbf || !gx1 || !gvec1 || !gxh || !gx || !gedge) return HN_ERR_BAD_ARG;
  if (vec && !gvec) return HN_ERR_BAD_ARG;
  if (g->num_edges > 0 && !edge) return HN_ERR_BAD_ARG;
  if (a.N == 0) return HN_OK;
  a.xh = xh; a.vec = vec; a.wt = wt; a.brbf = brbf;
  a.edge = reinterpret_cast<const float4*>(edge);
  a.gx1 = gx1; a.gvec1 = gvec1; a.gxh = gxh; a.gvec = gvec; a.gx = gx;
  a.gedge = reinterpret_cast<float4*>(gedge);
  static const int rpb_bwd = env_int("HERMNET_BWD_ROWS", 0);
  a.rows_per_block = pick_rows(a.N, hidden / HN_CB, 0, rpb_bwd);
  const size_t lds = lds_bytes(a.R);
  if (lds > 160 * 1024) return HN_ERR_LDS;
  dim3 grid((unsigned)((a.N + a.rows_per_block - 1) / a.rows_per_block), (unsigned)(hidden / HN_CB));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static const int nw = bwd_waves();
  void (*k)(MsgArgs);
  if (nw == 8) k = vec ? message_scatter_bwd_kernel<true, 8> : message_scatter_bwd_kernel<false, 8>;
  else k = vec ? message_scatter_bwd_kernel<true, 16> : message_scatter_bwd_kernel<false, 16>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess) return HN_ERR_LDS;
  hipLaunchKernelGGL(k, grid, dim3(nw * 64), lds, s, a);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}
