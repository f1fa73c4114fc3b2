// gfx950 elementwise / row-wise kernels of the node-level part of a HeteroVertexConv layer.
// The dense contractions between them are library GEMMs; these kernels fuse everything else so
// that a layer costs a fixed, small number of launches.  HBM-streaming, float4 per lane.
//
//   hermnet_ssilu_fwd / _bwd        ScaledSiLU                      /root/reference/HermNet/rmnet.py:110-117
//   hermnet_update_mid  / _bwd      vec_dot, |vec2|, cat([x, |vec2|])   rmnet.py:95-100
//   hermnet_update_out  / _bwd      dx, dvec, residual (+ zero rows)    rmnet.py:101-107, 29-31; hermnet.py:51,56-61
#include <hip/hip_runtime.h>
#include "../../include/hermnet_hip.h"

namespace {

constexpr float kInvSqrt2 = 0.70710678118654752f;
constexpr float kSiluScale = 1.0f / 0.6f;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }


__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Bias row of `row` for a [rows, cols] operand: bias is [groups, cols], group = row / rows_per_bias
// (rows_per_bias <= 0: one row for everything).  The GEMM that produced the operand ran without its bias
// (a broadcast-bias GEMM costs a full extra write + read of the output); it is added on load here.
__device__ __forceinline__ float4 bias4(const float* bias, int rows_per_bias, long row, int cols, int col) {
  if (bias == nullptr) return make_float4(0.f, 0.f, 0.f, 0.f);
  const long grp = rows_per_bias > 0 ? row / rows_per_bias : 0;
  return ld4(bias + grp * cols + col);
}

// a[row, col] = silu(h + bias) / 0.6 ; h, a contiguous [rows, cols]
__global__ __launch_bounds__(256) void ssilu_fwd_kernel(const float* __restrict__ h, const float* __restrict__ bias,
                                                        int rows_per_bias, float* __restrict__ a, long n4, int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c4n = cols >> 2;
  const float4 v = add4(ld4(h + 4 * i), bias4(bias, rows_per_bias, i / c4n, cols, (int)(i % c4n) * 4));
  float4 o;
  o.x = v.x * sigmoidf_(v.x) * kSiluScale;
  o.y = v.y * sigmoidf_(v.y) * kSiluScale;
  o.z = v.z * sigmoidf_(v.z) * kSiluScale;
  o.w = v.w * sigmoidf_(v.w) * kSiluScale;
  st4(a + 4 * i, o);
}

__device__ __forceinline__ float dssilu(float x) {
  const float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s)) * kSiluScale;
}

// gh[n, t, c] = g[t?..] * d ssilu(h[n, t, c] + bias);  g is addressed as g[n * gs_n + t * gs_t + c]
// (gs_n = T*C, gs_t = C for the same layout as h; gs_n = C, gs_t = N*C for a [T,N,C] gradient);
// bias as in ssilu_fwd for the [N, T*C] view of h.
__global__ __launch_bounds__(256) void ssilu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ h,
                                                        const float* __restrict__ bias, int rows_per_bias,
                                                        float* __restrict__ gh, int N, int T, int C,
                                                        long gs_n, long gs_t) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index into h
  const int c4n = C >> 2;
  const long total = (long)N * T * c4n;
  if (i >= total) return;
  const int c4 = (int)(i % c4n);
  const long nt = i / c4n;
  const int t = (int)(nt % T);
  const long n = nt / T;
  const float4 hv = add4(ld4(h + 4 * i), bias4(bias, rows_per_bias, n, T * C, t * C + 4 * c4));
  const float4 gv = ld4(g + n * gs_n + t * gs_t + 4 * c4);
  float4 o;
  o.x = gv.x * dssilu(hv.x);
  o.y = gv.y * dssilu(hv.y);
  o.z = gv.z * dssilu(hv.z);
  o.w = gv.w * dssilu(hv.w);
  st4(gh + 4 * i, o);
}

// vp [N,3,2H] -> vdot [N,H], xin [N,2H] = [x1 | sqrt(sum_d v2^2 + 1e-8)]
__global__ __launch_bounds__(256) void update_mid_kernel(const float* __restrict__ vp, const float* __restrict__ x1,
                                                         float* __restrict__ vdot, float* __restrict__ xin,
                                                         int rows, int H) {
  const int h4n = H >> 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * h4n) return;
  const int c = (int)(i % h4n) * 4;
  const long r = i / h4n;
  const float inv_sqrt_h = rsqrtf((float)H);
  float4 dot = make_float4(0.f, 0.f, 0.f, 0.f), sq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float4 a = ld4(vp + (r * 3 + d) * 2 * H + c);
    const float4 b = ld4(vp + (r * 3 + d) * 2 * H + H + c);
    dot.x = fmaf(a.x, b.x, dot.x); dot.y = fmaf(a.y, b.y, dot.y); dot.z = fmaf(a.z, b.z, dot.z); dot.w = fmaf(a.w, b.w, dot.w);
    sq.x = fmaf(b.x, b.x, sq.x); sq.y = fmaf(b.y, b.y, sq.y); sq.z = fmaf(b.z, b.z, sq.z); sq.w = fmaf(b.w, b.w, sq.w);
  }
  st4(vdot + r * H + c, make_float4(dot.x * inv_sqrt_h, dot.y * inv_sqrt_h, dot.z * inv_sqrt_h, dot.w * inv_sqrt_h));
  st4(xin + r * 2 * H + c, ld4(x1 + r * H + c));
  st4(xin + r * 2 * H + H + c,
      make_float4(sqrtf(sq.x + 1e-8f), sqrtf(sq.y + 1e-8f), sqrtf(sq.z + 1e-8f), sqrtf(sq.w + 1e-8f)));
}

// x_out = x1 + (q1 + q2 vdot)/sqrt2 ; vec_out[d] = vec1[d] + q3 v1[d]; rows with m == 0 (or >= nk) are zero;
// q = q_in + bias (see bias4)
__global__ __launch_bounds__(256) void update_out_kernel(const float* __restrict__ q, const float* __restrict__ qbias,
                                                         int rows_per_bias, const float* __restrict__ vdot,
                                                         const float* __restrict__ vp, const float* __restrict__ x1,
                                                         const float* __restrict__ vec1, const float* __restrict__ mask,
                                                         float* __restrict__ xo, float* __restrict__ vo,
                                                         int N, int nk, int H) {
  const int h4n = H >> 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * h4n) return;
  const int c = (int)(i % h4n) * 4;
  const long r = i / h4n;
  const bool on = r < nk && (mask == nullptr || mask[r] != 0.0f);
  if (!on) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    st4(xo + r * H + c, z);
#pragma unroll
    for (int d = 0; d < 3; ++d) st4(vo + (r * 3 + d) * H + c, z);
    return;
  }
  const float4 q1 = add4(ld4(q + r * 3 * H + c), bias4(qbias, rows_per_bias, r, 3 * H, c));
  const float4 q2 = add4(ld4(q + r * 3 * H + H + c), bias4(qbias, rows_per_bias, r, 3 * H, H + c));
  const float4 q3 = add4(ld4(q + r * 3 * H + 2 * H + c), bias4(qbias, rows_per_bias, r, 3 * H, 2 * H + c));
  const float4 vd = ld4(vdot + r * H + c), xv = ld4(x1 + r * H + c);
  st4(xo + r * H + c, make_float4(xv.x + (q1.x + q2.x * vd.x) * kInvSqrt2, xv.y + (q1.y + q2.y * vd.y) * kInvSqrt2,
                                  xv.z + (q1.z + q2.z * vd.z) * kInvSqrt2, xv.w + (q1.w + q2.w * vd.w) * kInvSqrt2));
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float4 v1 = ld4(vp + (r * 3 + d) * 2 * H + c);
    const float4 vv = ld4(vec1 + (r * 3 + d) * H + c);
    st4(vo + (r * 3 + d) * H + c,
        make_float4(fmaf(q3.x, v1.x, vv.x), fmaf(q3.y, v1.y, vv.y), fmaf(q3.z, v1.z, vv.z), fmaf(q3.w, v1.w, vv.w)));
  }
}

// backward of update_out: gq [N,3H], gvdot [N,H], gvp[:, :, 0:H] (= gvo q3; the v2 half is written by
// update_mid_bwd), gx1o [N,H] (= masked gxo), gv1o [N,3,H] (= masked gvo)
__global__ __launch_bounds__(256) void update_out_bwd_kernel(
    const float* __restrict__ gxo, const float* __restrict__ gvo, const float* __restrict__ q,
    const float* __restrict__ qbias, int rows_per_bias,
    const float* __restrict__ vdot, const float* __restrict__ vp, const float* __restrict__ mask,
    float* __restrict__ gq, float* __restrict__ gvdot, float* __restrict__ gvp, float* __restrict__ gx1o,
    float* __restrict__ gv1o, int N, int nk, int H) {
  const int h4n = H >> 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * h4n) return;
  const int c = (int)(i % h4n) * 4;
  const long r = i / h4n;
  const bool on = r < nk && (mask == nullptr || mask[r] != 0.0f);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!on) {
    st4(gx1o + r * H + c, z);
#pragma unroll
    for (int d = 0; d < 3; ++d) st4(gv1o + (r * 3 + d) * H + c, z);
    if (r < nk) {   // inactive relation: keep the GEMM inputs finite
      st4(gq + r * 3 * H + c, z); st4(gq + r * 3 * H + H + c, z); st4(gq + r * 3 * H + 2 * H + c, z);
      st4(gvdot + r * H + c, z);
#pragma unroll
      for (int d = 0; d < 3; ++d) st4(gvp + (r * 3 + d) * 2 * H + c, z);
    }
    return;
  }
  const float4 gx = ld4(gxo + r * H + c);
  const float4 q2 = add4(ld4(q + r * 3 * H + H + c), bias4(qbias, rows_per_bias, r, 3 * H, H + c));
  const float4 q3 = add4(ld4(q + r * 3 * H + 2 * H + c), bias4(qbias, rows_per_bias, r, 3 * H, 2 * H + c));
  const float4 vd = ld4(vdot + r * H + c);
  st4(gx1o + r * H + c, gx);
  st4(gq + r * 3 * H + c, make_float4(gx.x * kInvSqrt2, gx.y * kInvSqrt2, gx.z * kInvSqrt2, gx.w * kInvSqrt2));
  st4(gq + r * 3 * H + H + c, make_float4(gx.x * vd.x * kInvSqrt2, gx.y * vd.y * kInvSqrt2, gx.z * vd.z * kInvSqrt2,
                                          gx.w * vd.w * kInvSqrt2));
  st4(gvdot + r * H + c, make_float4(gx.x * q2.x * kInvSqrt2, gx.y * q2.y * kInvSqrt2, gx.z * q2.z * kInvSqrt2,
                                     gx.w * q2.w * kInvSqrt2));
  float4 g3 = z;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float4 gv = ld4(gvo + (r * 3 + d) * H + c);
    const float4 v1 = ld4(vp + (r * 3 + d) * 2 * H + c);
    g3.x = fmaf(gv.x, v1.x, g3.x); g3.y = fmaf(gv.y, v1.y, g3.y); g3.z = fmaf(gv.z, v1.z, g3.z); g3.w = fmaf(gv.w, v1.w, g3.w);
    st4(gvp + (r * 3 + d) * 2 * H + c, make_float4(gv.x * q3.x, gv.y * q3.y, gv.z * q3.z, gv.w * q3.w));
    st4(gv1o + (r * 3 + d) * H + c, gv);
  }
  st4(gq + r * 3 * H + 2 * H + c, g3);
}

// backward of update_mid: completes gvp [N,3,2H] and gx1 = gx1o + gxin[:, :H]
__global__ __launch_bounds__(256) void update_mid_bwd_kernel(
    const float* __restrict__ gvdot, const float* __restrict__ gxin, const float* __restrict__ vp,
    const float* __restrict__ xin, float* __restrict__ gvp, float* __restrict__ gx1, int rows, int H) {
  const int h4n = H >> 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * h4n) return;
  const int c = (int)(i % h4n) * 4;
  const long r = i / h4n;
  const float s = rsqrtf((float)H);
  const float4 gd = ld4(gvdot + r * H + c);
  const float4 gn = ld4(gxin + r * 2 * H + H + c);
  const float4 nr = ld4(xin + r * 2 * H + H + c);
  const float4 gnn = make_float4(gn.x / nr.x, gn.y / nr.y, gn.z / nr.z, gn.w / nr.w);
  const float4 ge = ld4(gxin + r * 2 * H + c);
  const float4 g0 = ld4(gx1 + r * H + c);
  st4(gx1 + r * H + c, make_float4(g0.x + ge.x, g0.y + ge.y, g0.z + ge.z, g0.w + ge.w));
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const long o = (r * 3 + d) * 2 * H + c;
    const float4 v1 = ld4(vp + o), v2 = ld4(vp + o + H);
    const float4 p = ld4(gvp + o);
    st4(gvp + o, make_float4(fmaf(gd.x * s, v2.x, p.x), fmaf(gd.y * s, v2.y, p.y), fmaf(gd.z * s, v2.z, p.z),
                             fmaf(gd.w * s, v2.w, p.w)));
    st4(gvp + o + H, make_float4(fmaf(gd.x * s, v1.x, gnn.x * v2.x), fmaf(gd.y * s, v1.y, gnn.y * v2.y),
                                 fmaf(gd.z * s, v1.z, gnn.z * v2.z), fmaf(gd.w * s, v1.w, gnn.w * v2.w)));
  }
}

// ---- LayerNorm without affine (the affine is folded into the following Linear, hermnet_amd/layer.py) ----
// rmnet.py:52 `x_layernorm`; one wave per row, H <= 1024; two-pass statistics in registers.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

constexpr int kLnMaxPerLane = 4;   // float4 chunks per lane: H <= 4 * 64 * 4 = 1024

// Rows are H floats apart; the statistics run over the first Hr channels only (Hr < H: the row is zero-padded to a
// multiple of 64 channels for the message kernels' column blocks, hermnet_amd/layer.py; padded outputs are zero).
__device__ __forceinline__ float4 chan_mask4(int c, int Hr) {
  return make_float4(c < Hr ? 1.f : 0.f, c + 1 < Hr ? 1.f : 0.f, c + 2 < Hr ? 1.f : 0.f, c + 3 < Hr ? 1.f : 0.f);
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ n,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int rows, int H, int Hr, float eps) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float4 v[kLnMaxPerLane];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kLnMaxPerLane; ++k) {
    const int c = (k * 64 + lane) * 4;
    v[k] = c < H ? ld4(x + (size_t)r * H + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 m = chan_mask4(c, Hr);
    v[k] = make_float4(v[k].x * m.x, v[k].y * m.y, v[k].z * m.z, v[k].w * m.w);
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mu = wave_sum(s) / (float)Hr;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < kLnMaxPerLane; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < H) {
      const float4 m = chan_mask4(c, Hr);
      v[k] = make_float4((v[k].x - mu) * m.x, (v[k].y - mu) * m.y, (v[k].z - mu) * m.z, (v[k].w - mu) * m.w);
      q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)Hr + eps);
#pragma unroll
  for (int k = 0; k < kLnMaxPerLane; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < H) st4(n + (size_t)r * H + c, make_float4(v[k].x * rs, v[k].y * rs, v[k].z * rs, v[k].w * rs));
  }
  if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}

// gx = rstd * (g - mean(g) - nh * mean(g * nh)) + add   with nh = (x - mean) * rstd; `add` may be null
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ add, float* __restrict__ gx,
                                                            int rows, int H, int Hr) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float mu = mean[r], rs = rstd[r];
  float4 gv[kLnMaxPerLane], nh[kLnMaxPerLane];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < kLnMaxPerLane; ++k) {
    const int c = (k * 64 + lane) * 4;
    gv[k] = nh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < H) {
      const float4 m = chan_mask4(c, Hr);
      const float4 gg = ld4(g + (size_t)r * H + c);
      gv[k] = make_float4(gg.x * m.x, gg.y * m.y, gg.z * m.z, gg.w * m.w);
      const float4 xv = ld4(x + (size_t)r * H + c);
      nh[k] = make_float4((xv.x - mu) * rs * m.x, (xv.y - mu) * rs * m.y, (xv.z - mu) * rs * m.z, (xv.w - mu) * rs * m.w);
    }
    s1 += (gv[k].x + gv[k].y) + (gv[k].z + gv[k].w);
    s2 += (gv[k].x * nh[k].x + gv[k].y * nh[k].y) + (gv[k].z * nh[k].z + gv[k].w * nh[k].w);
  }
  const float m1 = wave_sum(s1) / (float)Hr, m2 = wave_sum(s2) / (float)Hr;
#pragma unroll
  for (int k = 0; k < kLnMaxPerLane; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < H) {
      const float4 m = chan_mask4(c, Hr);
      float4 o = make_float4(rs * (gv[k].x - m1 - nh[k].x * m2) * m.x, rs * (gv[k].y - m1 - nh[k].y * m2) * m.y,
                             rs * (gv[k].z - m1 - nh[k].z * m2) * m.z, rs * (gv[k].w - m1 - nh[k].w * m2) * m.w);
      if (add != nullptr) o = add4(o, ld4(add + (size_t)r * H + c));
      st4(gx + (size_t)r * H + c, o);
    }
  }
}

// ---- energy read-out head (hermnet.py:113-117,129): e[n] = sum_c ssilu(h[n,c]) w[c] + b -----------------
// h = out_energy[0](x) comes from a library GEMM; one wave per row, C <= 1024.
__global__ __launch_bounds__(256) void energy_head_fwd_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                              const float* __restrict__ b,
                                                              const float* __restrict__ mask, float* __restrict__ e,
                                                              int rows, int C) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 v = ld4(h + (size_t)r * C + c), wv = ld4(w + c);
    s += v.x * sigmoidf_(v.x) * wv.x + v.y * sigmoidf_(v.y) * wv.y + v.z * sigmoidf_(v.z) * wv.z +
         v.w * sigmoidf_(v.w) * wv.w;
  }
  s = wave_sum(s);
  if (lane == 0) e[r] = (s * kSiluScale + (b ? b[0] : 0.f)) * (mask ? mask[r] : 1.0f);
}

// gh[n,c] = ge[n] * w[c] * d ssilu(h[n,c])
__global__ __launch_bounds__(256) void energy_head_bwd_kernel(const float* __restrict__ ge, const float* __restrict__ h,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ mask, float* __restrict__ gh,
                                                              long n4, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c4n = C >> 2;
  const long r = i / c4n;
  const int c = (int)(i % c4n) * 4;
  const float g = ge[r] * (mask ? mask[r] : 1.0f);
  const float4 hv = ld4(h + 4 * i), wv = ld4(w + c);
  st4(gh + 4 * i, make_float4(g * wv.x * dssilu(hv.x), g * wv.y * dssilu(hv.y), g * wv.z * dssilu(hv.z),
                              g * wv.w * dssilu(hv.w)));
}

// ---- halo exchange packing (hermnet_amd/sharding.py) -------------------------------------------------------
// One packed row per halo atom: [ x (H) | vec (3H) ] = 4H floats; idx holds rows of x / vec.
// MODE 0: buf[k] = rows[idx[k]]            (pack)
// MODE 1: buf[k] = rows[idx[k]]; rows[idx[k]] = 0   (pack the gradients of overwritten halo rows, then clear them)
// MODE 2: rows[idx[k]] = buf[k]            (unpack; idx unique)
// (accumulating returned gradients at their owner is halo_accumulate_kernel below: ordered sums, no float atomics)
template <int MODE>
__global__ __launch_bounds__(256) void halo_rows_kernel(float* __restrict__ x, float* __restrict__ vec,
                                                        const long* __restrict__ idx, int n, int H,
                                                        float* __restrict__ buf) {
  const int q4 = H;                                   // float4 per packed row: 4H / 4
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * q4) return;
  const int k = (int)(i / q4), c = (int)(i % q4) * 4;  // c in [0, 4H)
  const long r = idx[k];
  float* row = c < H ? x + r * H + c : vec + r * 3 * H + (c - H);
  float* b = buf + (long)k * 4 * H + c;
  if (MODE == 0) {
    st4(b, ld4(row));
  } else if (MODE == 1) {
    st4(b, ld4(row));
    st4(row, make_float4(0.f, 0.f, 0.f, 0.f));
  } else {
    st4(row, ld4(b));
  }
}

// Deterministic accumulate at the owner: rows[seg_rows[u]] += sum_{q in [seg_ptr[u], seg_ptr[u+1])} buf[seg_pos[q]],
// in list order (an owned atom can be a halo atom of several peers; a fixed order keeps the sharded step
// bit-reproducible, which float atomics do not).
__global__ __launch_bounds__(256) void halo_accumulate_kernel(float* __restrict__ x, float* __restrict__ vec,
                                                              const long* __restrict__ seg_rows,
                                                              const long* __restrict__ seg_ptr,
                                                              const long* __restrict__ seg_pos, int nu, int H,
                                                              const float* __restrict__ buf) {
  const int q4 = H;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nu * q4) return;
  const int u = (int)(i / q4), c = (int)(i % q4) * 4;
  const long r = seg_rows[u];
  float* row = c < H ? x + r * H + c : vec + r * 3 * H + (c - H);
  float4 acc = ld4(row);
  for (long q = seg_ptr[u]; q < seg_ptr[u + 1]; ++q) {
    const float4 v = ld4(buf + seg_pos[q] * 4 * H + c);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  st4(row, acc);
}

// ---- halo exchange of PROJECTED rows (round 6; sharding.py / layer.py: the "proj" form of the exchange) ----------------------
// What a neighbour rank needs of a halo atom is not x but its projections xh[t] = x_proj_t(LayerNorm(x)) for the T relations
// and vec -- T + 1 blocks of W = 3H floats.  Sending THEM (12H floats per atom instead of 4H) means the receiver runs no node
// projection on halo rows at all, forward or backward: no second, windowed launch of the node chain kernels on the halo tiles
// (each one workgroup latency, ~20-50 us, for a few dozen tiles), no finishing launches in front of the gradient exchange.
// A packed row = [ a[0][r] | ... | a[S-1][r] | sum_s b[s][r] ]  with a[j] = a + j * a_seg_stride (row stride W), b[s] = b + s *
// b_slice_stride (row stride W): forward a = xh [T, N, 3H], b = vec [N, 3, H] (one slice); backward a = gxh, b = the
// per-relation partial sums of gvec [T, N, 3, H], summed over the relations while they are packed.
//   MODE 0  pack       MODE 1  pack, then clear the sources (every a[j][r] and every b[s][r])
//   MODE 2  unpack:    a[j][r] = block j, b[0][r] = block S      (idx unique)
template <int MODE>
__global__ __launch_bounds__(256) void halo_proj_rows_kernel(float* __restrict__ a, long a_seg_stride, int S,
                                                             float* __restrict__ b, long b_slice_stride, int nsum,
                                                             const long* __restrict__ idx, int n, int W,
                                                             float* __restrict__ buf) {
  const int q4 = W / 4, blocks = S + 1;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * blocks * q4) return;
  const int k = (int)(i / ((long)blocks * q4)), rem = (int)(i % ((long)blocks * q4));
  const int j = rem / q4, c = (rem % q4) * 4;
  const long r = idx[k];
  float* out = buf + ((long)k * blocks + j) * W + c;
  if (j < S) {
    float* row = a + (long)j * a_seg_stride + r * W + c;
    if (MODE == 2) {
      st4(row, ld4(out));
    } else {
      st4(out, ld4(row));
      if (MODE == 1) st4(row, make_float4(0.f, 0.f, 0.f, 0.f));
    }
  } else {
    float* row = b + r * W + c;
    if (MODE == 2) {
      st4(row, ld4(out));
    } else {
      float4 acc = ld4(row);
      if (MODE == 1) st4(row, make_float4(0.f, 0.f, 0.f, 0.f));
      for (int s = 1; s < nsum; ++s) {                 // (ascending relation: a fixed order)
        const float4 v = ld4(row + (long)s * b_slice_stride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        if (MODE == 1) st4(row + (long)s * b_slice_stride, make_float4(0.f, 0.f, 0.f, 0.f));
      }
      st4(out, acc);
    }
  }
}

// the owner's side of the return path: a[j][seg_rows[u]] += sum of block j of the returned rows of segment u, b[0][seg_rows[u]]
// += sum of their last blocks -- in list order, no atomics (as halo_accumulate_kernel)
__global__ __launch_bounds__(256) void halo_proj_accumulate_kernel(float* __restrict__ a, long a_seg_stride, int S,
                                                                   float* __restrict__ b, const long* __restrict__ seg_rows,
                                                                   const long* __restrict__ seg_ptr,
                                                                   const long* __restrict__ seg_pos, int nu, int W,
                                                                   const float* __restrict__ buf) {
  const int q4 = W / 4, blocks = S + 1;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nu * blocks * q4) return;
  const int u = (int)(i / ((long)blocks * q4)), rem = (int)(i % ((long)blocks * q4));
  const int j = rem / q4, c = (rem % q4) * 4;
  const long r = seg_rows[u];
  float* row = j < S ? a + (long)j * a_seg_stride + r * W + c : b + r * W + c;
  float4 acc = ld4(row);
  for (long q = seg_ptr[u]; q < seg_ptr[u + 1]; ++q) {
    const float4 v = ld4(buf + (seg_pos[q] * blocks + j) * W + c);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  st4(row, acc);
}

// ---- fused read-out (hermnet.py:113-117,129): e[n] = w2 . ScaledSiLU(W0 x[n] + b0) + b2, without a library GEMM --------
// 0.16 GFLOP at 10k atoms: not worth a matrix-core kernel, but worth two launches and an [N, H/2] round trip less.
// A 256-thread workgroup owns 32 rows: the weight matrix (K x M floats, <= 64 KB) and the rows' operand tile sit in
// LDS; a wave works on 8 rows at once, lane = output column (M/64 columns per lane), so a weight value read from LDS
// (conflict-free) is used for 8 rows and the operand values are LDS broadcasts.
//   FWD: operand x [N, H], weights W0^T [H, C], result h [N, C] (saved) and e [N];  K = H, M = C
//   BWD: operand gh [N, C] = ge mask w2 ScaledSiLU'(h) built on the fly, weights W0 [C, H], result gx [N, H];  K = C, M = H
template <bool FWD, int MPL>      // MPL = output columns per lane (M = 64 MPL)
__global__ __launch_bounds__(256) void energy_head_fused_kernel(const float* __restrict__ opnd, const float* __restrict__ wkm,
                                                                const float* __restrict__ b0, const float* __restrict__ w2,
                                                                const float* __restrict__ b2, const float* __restrict__ ge,
                                                                const float* __restrict__ mask, float* __restrict__ out_mat,
                                                                float* __restrict__ e, int rows, int K) {
  constexpr int M = 64 * MPL;
  extern __shared__ __align__(16) float lds[];
  float* wl = lds;                        // [K][M]
  float* xt = lds + (size_t)K * M;        // [32][K + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * 32;
  for (int i = tid; i < K * M / 4; i += 256) reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(wkm)[i];
  for (int i = tid; i < 32 * K; i += 256) {
    const int r = i / K, k = i % K, row = row0 + r;
    float v = 0.f;
    if (row < rows) {
      if (FWD) v = opnd[(size_t)row * K + k];
      else v = ge[row] * (mask ? mask[row] : 1.0f) * w2[k] * dssilu(opnd[(size_t)row * K + k]);
    }
    xt[r * (K + 1) + k] = v;
  }
  __syncthreads();
  float acc[8][MPL];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int m = 0; m < MPL; ++m) acc[r][m] = 0.f;
  const float* xr = xt + wave * 8 * (K + 1);
  for (int k = 0; k < K; ++k) {
    float w[MPL];
#pragma unroll
    for (int m = 0; m < MPL; ++m) w[m] = wl[k * M + m * 64 + lane];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float xv = xr[r * (K + 1) + k];
#pragma unroll
      for (int m = 0; m < MPL; ++m) acc[r][m] = fmaf(xv, w[m], acc[r][m]);
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = row0 + wave * 8 + r;
    if (row >= rows) continue;                       // (wave-uniform)
    if (FWD) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < MPL; ++m) {
        const int c = m * 64 + lane;
        const float hv = acc[r][m] + b0[c];
        out_mat[(size_t)row * M + c] = hv;
        s += hv * sigmoidf_(hv) * w2[c];
      }
      s = wave_sum(s);
      if (lane == 0) e[row] = (s * kSiluScale + (b2 ? b2[0] : 0.f)) * (mask ? mask[row] : 1.0f);
    } else {
#pragma unroll
      for (int m = 0; m < MPL; ++m) out_mat[(size_t)row * M + m * 64 + lane] = acc[r][m];
    }
  }
}

// ---- HTNet: combine a centre atom's P pair relations (DESIGN.md "HTNet") ----------------------------------------------
// Target rows are [Te][P][B] blocks of B rows; the layer's result for the atom in row c*B + i is the MEAN over its P
// virtual rows.  One packed pass over (x | vec): a float4 of [rows, 4H] per thread.
//   FWD: out[c*B + i] = (1/P) sum_k in[(c*P + k)*B + i]   for rows < Te*B, zero rows behind (elements outside `elems`)
//   BWD: gin[(c*P + k)*B + i] = gout[c*B + i] / P
//   ACC: out[c*B + i] += (sx | sv) * sum_k in[(c*P + k)*B + i]   (the residual's gradient: the kernel-side identity
//        term of hermnet_message_scatter_bwd is absent with virtual target rows, the host adds it summed over P)
template <int MODE>   // 0 FWD, 1 BWD, 2 ACC
__global__ __launch_bounds__(256) void pair_mean_kernel(const float* __restrict__ xin, const float* __restrict__ vin,
                                                        float* __restrict__ xout, float* __restrict__ vout, int Te, int P,
                                                        int B, int rows_out, int H, float sx, float sv,
                                                        const int* __restrict__ ranges, int num_ranges) {
  const int q4 = H;                                     // float4 per packed row of 4H floats
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long rows = MODE == 0 ? rows_out : (MODE == 1 ? (long)Te * P * B : (long)Te * B);
  if (i >= rows * q4) return;
  const long r = i / q4;
  if (MODE == 2 && num_ranges > 0) {       // only the output rows of these [lo, hi) ranges (atom shards: two launches
    bool in = false;                       // over complementary row ranges around the halo exchange)
    for (int k = 0; k < num_ranges; ++k) in |= r >= ranges[2 * k] && r < ranges[2 * k + 1];
    if (!in) return;
  }
  const int c = (int)(i % q4) * 4;                     // in [0, 4H)
  const float sc = c < H ? sx : sv;
  auto at = [&](const float* x, const float* v, long row) {
    return c < H ? ld4(x + row * H + c) : ld4(v + row * 3 * H + (c - H));
  };
  float* dst = c < H ? xout + r * H + c : vout + r * 3 * H + (c - H);
  if (MODE != 1) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < (long)Te * B) {
      const long ce = r / B, ii = r % B;
      for (int k = 0; k < P; ++k) acc = add4(acc, at(xin, vin, (ce * P + k) * B + ii));
      acc = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
    }
    if (MODE == 2) acc = add4(acc, ld4(dst));
    st4(dst, acc);
  } else {
    const long ce = r / ((long)P * B), ii = r % B;
    const float4 g = at(xin, vin, ce * B + ii);
    st4(dst, make_float4(g.x * sc, g.y * sc, g.z * sc, g.w * sc));
  }
}

inline dim3 grid_for(long n, int block) { return dim3((unsigned)((n + block - 1) / block)); }
#define HN_LAUNCH_END return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH

}  // namespace

extern "C" int hermnet_ssilu_fwd(const float* h, const float* bias, int rows_per_bias, float* a, long rows, int cols,
                                 void* stream) {
  if (rows < 0 || cols <= 0 || (cols & 3)) return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!h || !a) return HN_ERR_BAD_ARG;
  const long n4 = rows * (cols / 4);
  hipLaunchKernelGGL(ssilu_fwd_kernel, grid_for(n4, 256), dim3(256), 0, (hipStream_t)stream, h, bias, rows_per_bias,
                     a, n4, cols);
  HN_LAUNCH_END;
}

extern "C" int hermnet_ssilu_bwd(const float* g, const float* h, const float* bias, int rows_per_bias, float* gh,
                                 int N, int T, int C, long g_stride_n, long g_stride_t, void* stream) {
  if (N < 0 || T <= 0 || C <= 0 || (C & 3) || (g_stride_n & 3) || (g_stride_t & 3)) return HN_ERR_BAD_ARG;
  if (N == 0) return HN_OK;
  if (!g || !h || !gh) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(ssilu_bwd_kernel, grid_for((long)N * T * (C / 4), 256), dim3(256), 0, (hipStream_t)stream,
                     g, h, bias, rows_per_bias, gh, N, T, C, g_stride_n, g_stride_t);
  HN_LAUNCH_END;
}

extern "C" int hermnet_update_mid(const float* vp, const float* x1, float* vdot, float* xin, int rows, int hidden,
                                  void* stream) {
  if (rows < 0 || hidden <= 0 || (hidden & 3)) return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!vp || !x1 || !vdot || !xin) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(update_mid_kernel, grid_for((long)rows * (hidden / 4), 256), dim3(256), 0, (hipStream_t)stream,
                     vp, x1, vdot, xin, rows, hidden);
  HN_LAUNCH_END;
}

extern "C" int hermnet_update_out(const float* q, const float* qbias, int rows_per_bias, const float* vdot,
                                  const float* vp, const float* x1,
                                  const float* vec1, const float* row_mask, float* x_out, float* vec_out,
                                  int num_nodes, int num_known, int hidden, void* stream) {
  if (num_nodes < 0 || hidden <= 0 || (hidden & 3) || num_known > num_nodes) return HN_ERR_BAD_ARG;
  if (num_nodes == 0) return HN_OK;
  if (!q || !vdot || !vp || !x1 || !vec1 || !x_out || !vec_out) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(update_out_kernel, grid_for((long)num_nodes * (hidden / 4), 256), dim3(256), 0,
                     (hipStream_t)stream, q, qbias, rows_per_bias, vdot, vp, x1, vec1, row_mask, x_out, vec_out, num_nodes,
                     num_known, hidden);
  HN_LAUNCH_END;
}

extern "C" int hermnet_update_out_bwd(const float* gx_out, const float* gvec_out, const float* q,
                                      const float* qbias, int rows_per_bias, const float* vdot,
                                      const float* vp, const float* row_mask, float* gq, float* gvdot, float* gvp,
                                      float* gx1, float* gvec1, int num_nodes, int num_known, int hidden,
                                      void* stream) {
  if (num_nodes < 0 || hidden <= 0 || (hidden & 3) || num_known > num_nodes) return HN_ERR_BAD_ARG;
  if (num_nodes == 0) return HN_OK;
  if (!gx_out || !gvec_out || !q || !vdot || !vp || !gq || !gvdot || !gvp || !gx1 || !gvec1) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(update_out_bwd_kernel, grid_for((long)num_nodes * (hidden / 4), 256), dim3(256), 0,
                     (hipStream_t)stream, gx_out, gvec_out, q, qbias, rows_per_bias, vdot, vp, row_mask, gq, gvdot, gvp, gx1,
                     gvec1,
                     num_nodes, num_known, hidden);
  HN_LAUNCH_END;
}

extern "C" int hermnet_update_mid_bwd(const float* gvdot, const float* gxin, const float* vp, const float* xin,
                                      float* gvp, float* gx1, int rows, int hidden, void* stream) {
  if (rows < 0 || hidden <= 0 || (hidden & 3)) return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!gvdot || !gxin || !vp || !xin || !gvp || !gx1) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(update_mid_bwd_kernel, grid_for((long)rows * (hidden / 4), 256), dim3(256), 0,
                     (hipStream_t)stream, gvdot, gxin, vp, xin, gvp, gx1, rows, hidden);
  HN_LAUNCH_END;
}

extern "C" int hermnet_layernorm_fwd(const float* x, float* n, float* mean, float* rstd, int rows, int hidden,
                                     int hidden_real, float eps, void* stream) {
  if (hidden_real <= 0) hidden_real = hidden;
  if (rows < 0 || hidden <= 0 || (hidden & 3) || hidden > kLnMaxPerLane * 256 || hidden_real > hidden) return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!x || !n || !mean || !rstd) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     x, n, mean, rstd, rows, hidden, hidden_real, eps);
  HN_LAUNCH_END;
}

extern "C" int hermnet_layernorm_bwd(const float* g, const float* x, const float* mean, const float* rstd,
                                     const float* add, float* gx, int rows, int hidden, int hidden_real, void* stream) {
  if (hidden_real <= 0) hidden_real = hidden;
  if (rows < 0 || hidden <= 0 || (hidden & 3) || hidden > kLnMaxPerLane * 256 || hidden_real > hidden) return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!g || !x || !mean || !rstd || !gx) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     g, x, mean, rstd, add, gx, rows, hidden, hidden_real);
  HN_LAUNCH_END;
}

extern "C" int hermnet_energy_head_fwd(const float* h, const float* w, const float* b, const float* row_mask, float* e,
                                       int rows, int cols, void* stream) {
  if (rows < 0 || cols <= 0 || (cols & 3)) return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!h || !w || !e) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(energy_head_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     h, w, b, row_mask, e, rows, cols);
  HN_LAUNCH_END;
}

extern "C" int hermnet_energy_head_bwd(const float* ge, const float* h, const float* w, const float* row_mask, float* gh,
                                       int rows, int cols, void* stream) {
  if (rows < 0 || cols <= 0 || (cols & 3)) return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!ge || !h || !w || !gh) return HN_ERR_BAD_ARG;
  const long n4 = (long)rows * (cols / 4);
  hipLaunchKernelGGL(energy_head_bwd_kernel, grid_for(n4, 256), dim3(256), 0, (hipStream_t)stream, ge, h, w, row_mask, gh, n4,
                     cols);
  HN_LAUNCH_END;
}

// Fused read-out.  forward: x [rows, hidden] -> h [rows, cols] (pre-activation, saved for the backward) and e [rows];
// w0t = out_energy[0].weight^T [hidden, cols].  backward: ge [rows] -> gx [rows, hidden]; w0 = out_energy[0].weight
// [cols, hidden].  cols in {64, 128, 256} and hidden in {64, 128, 256} with hidden * cols * 4 <= 64 KiB; else HN_ERR_BAD_ARG
// (the caller then takes the library-GEMM form: hermnet_energy_head_fwd / _bwd).
extern "C" int hermnet_energy_head_fused_fwd(const float* x, const float* w0t, const float* b0, const float* w2,
                                             const float* b2, const float* row_mask, float* h, float* e, int rows,
                                             int hidden, int cols, void* stream) {
  if (rows < 0 || (cols != 64 && cols != 128 && cols != 256) || hidden <= 0 || (hidden & 3) ||
      (size_t)hidden * cols * 4 > 65536)
    return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!x || !w0t || !b0 || !w2 || !h || !e) return HN_ERR_BAD_ARG;
  const size_t lds = ((size_t)hidden * cols + 32 * (hidden + 1)) * sizeof(float);
  const dim3 grid((unsigned)((rows + 31) / 32));
  hipStream_t s = (hipStream_t)stream;
  auto go = [&](auto kern) -> int {
    static bool done = false;
    if (!done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 98304) != hipSuccess)
        return HN_ERR_LDS;
      done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, w0t, b0, w2, b2, (const float*)nullptr, row_mask, h, e, rows, hidden);
    HN_LAUNCH_END;
  };
  if (cols == 64) return go(energy_head_fused_kernel<true, 1>);
  if (cols == 128) return go(energy_head_fused_kernel<true, 2>);
  return go(energy_head_fused_kernel<true, 4>);
}

extern "C" int hermnet_energy_head_fused_bwd(const float* ge, const float* h, const float* w0, const float* w2,
                                             const float* row_mask, float* gx, int rows, int hidden, int cols,
                                             void* stream) {
  if (rows < 0 || (hidden != 64 && hidden != 128 && hidden != 256) || cols <= 0 || (cols & 3) ||
      (size_t)hidden * cols * 4 > 65536)
    return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  if (!ge || !h || !w0 || !w2 || !gx) return HN_ERR_BAD_ARG;
  const size_t lds = ((size_t)hidden * cols + 32 * (cols + 1)) * sizeof(float);
  const dim3 grid((unsigned)((rows + 31) / 32));
  hipStream_t s = (hipStream_t)stream;
  auto go = [&](auto kern) -> int {
    static bool done = false;
    if (!done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 98304) != hipSuccess)
        return HN_ERR_LDS;
      done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, h, w0, (const float*)nullptr, w2, (const float*)nullptr, ge, row_mask,
                       gx, (float*)nullptr, rows, cols);
    HN_LAUNCH_END;
  };
  if (hidden == 64) return go(energy_head_fused_kernel<false, 1>);
  if (hidden == 128) return go(energy_head_fused_kernel<false, 2>);
  return go(energy_head_fused_kernel<false, 4>);
}

extern "C" int hermnet_pair_mean(int mode, const float* x_in, const float* vec_in, float* x_out, float* vec_out,
                                 int num_elem, int pairs, int block, int rows_out, int hidden, float scale_x,
                                 float scale_vec, const int* row_ranges, int num_ranges, void* stream) {
  if (num_elem < 0 || pairs <= 0 || block < 0 || hidden <= 0 || (hidden & 3) || rows_out < (long)num_elem * block ||
      mode < 0 || mode > 2 || num_ranges < 0 || (num_ranges > 0 && (!row_ranges || mode != 2)))
    return HN_ERR_BAD_ARG;
  const long rows = mode == 0 ? (long)rows_out : (mode == 1 ? (long)num_elem * pairs * block : (long)num_elem * block);
  if (rows == 0) return HN_OK;
  if (!x_in || !vec_in || !x_out || !vec_out) return HN_ERR_BAD_ARG;
  const dim3 grid = grid_for(rows * hidden, 256);
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(pair_mean_kernel<0>, grid, dim3(256), 0, s, x_in, vec_in, x_out, vec_out, num_elem, pairs, block, rows_out, hidden, scale_x, scale_vec, (const int*)nullptr, 0); break;
    case 1: hipLaunchKernelGGL(pair_mean_kernel<1>, grid, dim3(256), 0, s, x_in, vec_in, x_out, vec_out, num_elem, pairs, block, rows_out, hidden, scale_x, scale_vec, (const int*)nullptr, 0); break;
    default: hipLaunchKernelGGL(pair_mean_kernel<2>, grid, dim3(256), 0, s, x_in, vec_in, x_out, vec_out, num_elem, pairs, block, rows_out, hidden, scale_x, scale_vec, row_ranges, num_ranges); break;
  }
  HN_LAUNCH_END;
}

extern "C" int hermnet_halo_rows(int mode, float* x, float* vec, const long* idx, int n, int hidden, float* buf,
                                 void* stream) {
  if (n < 0 || hidden <= 0 || (hidden & 3) || mode < 0 || mode > 2) return HN_ERR_BAD_ARG;
  if (n == 0) return HN_OK;
  if (!x || !vec || !idx || !buf) return HN_ERR_BAD_ARG;
  const dim3 grid = grid_for((long)n * hidden, 256);
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(halo_rows_kernel<0>, grid, dim3(256), 0, s, x, vec, idx, n, hidden, buf); break;
    case 1: hipLaunchKernelGGL(halo_rows_kernel<1>, grid, dim3(256), 0, s, x, vec, idx, n, hidden, buf); break;
    case 2: hipLaunchKernelGGL(halo_rows_kernel<2>, grid, dim3(256), 0, s, x, vec, idx, n, hidden, buf); break;
    default: return HN_ERR_BAD_ARG;
  }
  HN_LAUNCH_END;
}

extern "C" int hermnet_halo_proj_rows(int mode, float* a, long a_seg_stride, int num_seg, float* b, long b_slice_stride,
                                      int num_sum, const long* idx, int n, int width, float* buf, void* stream) {
  if (n < 0 || width <= 0 || (width & 3) || num_seg < 1 || num_sum < 1 || (mode == 2 && num_sum != 1)) return HN_ERR_BAD_ARG;
  if (n == 0) return HN_OK;
  if (!a || !b || !idx || !buf) return HN_ERR_BAD_ARG;
  const dim3 grid = grid_for((long)n * (num_seg + 1) * (width / 4), 256);
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(halo_proj_rows_kernel<0>, grid, dim3(256), 0, s, a, a_seg_stride, num_seg, b, b_slice_stride, num_sum, idx, n, width, buf); break;
    case 1: hipLaunchKernelGGL(halo_proj_rows_kernel<1>, grid, dim3(256), 0, s, a, a_seg_stride, num_seg, b, b_slice_stride, num_sum, idx, n, width, buf); break;
    case 2: hipLaunchKernelGGL(halo_proj_rows_kernel<2>, grid, dim3(256), 0, s, a, a_seg_stride, num_seg, b, b_slice_stride, num_sum, idx, n, width, buf); break;
    default: return HN_ERR_BAD_ARG;
  }
  HN_LAUNCH_END;
}

extern "C" int hermnet_halo_proj_accumulate(float* a, long a_seg_stride, int num_seg, float* b, const long* seg_rows,
                                            const long* seg_ptr, const long* seg_pos, int num_rows, int width,
                                            const float* buf, void* stream) {
  if (num_rows < 0 || width <= 0 || (width & 3) || num_seg < 1) return HN_ERR_BAD_ARG;
  if (num_rows == 0) return HN_OK;
  if (!a || !b || !seg_rows || !seg_ptr || !seg_pos || !buf) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(halo_proj_accumulate_kernel, grid_for((long)num_rows * (num_seg + 1) * (width / 4), 256), dim3(256), 0,
                     (hipStream_t)stream, a, a_seg_stride, num_seg, b, seg_rows, seg_ptr, seg_pos, num_rows, width, buf);
  HN_LAUNCH_END;
}

extern "C" int hermnet_halo_accumulate(float* x, float* vec, const long* seg_rows, const long* seg_ptr,
                                       const long* seg_pos, int num_rows, int hidden, const float* buf, void* stream) {
  if (num_rows < 0 || hidden <= 0 || (hidden & 3)) return HN_ERR_BAD_ARG;
  if (num_rows == 0) return HN_OK;
  if (!x || !vec || !seg_rows || !seg_ptr || !seg_pos || !buf) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(halo_accumulate_kernel, grid_for((long)num_rows * hidden, 256), dim3(256), 0, (hipStream_t)stream,
                     x, vec, seg_rows, seg_ptr, seg_pos, num_rows, hidden, buf);
  HN_LAUNCH_END;
}
