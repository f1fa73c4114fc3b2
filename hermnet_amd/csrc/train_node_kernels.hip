// gfx950 kernels of the TRAINING path, node level (train() mode: example/dist_train.py:86-99 differentiates the forces
// w.r.t. the parameters, so every op on the path is differentiated twice).  Between the node-level GEMMs of a layer
// (/root/reference/HermNet/rmnet.py:52, 94-107) sit four elementwise / row-wise stages; left to torch they are ~120 small
// launches per layer over the three orders of differentiation (views, slices and their zero-filled backward included).
// Here each stage is one launch per order:
//
//   LN    y = LayerNorm(x) without affine (the affine rides on the following weight)          rmnet.py:52
//   SILU  y = x sigmoid(x)              (ScaledSiLU's factor rides on the following weight)  rmnet.py:110-117
//   MID   vp [R,3,2H] = (v1 | v2), xt [R,H]  ->  vdot = c0 sum_d v1 v2,  xin [R,2H] = (xt | sqrt(sum_d v2^2 + c1))   rmnet.py:96-99
//   OUT   q [R,3H] = (q1 | q2 | q3)  ->  xo = m (xt + (q1 + q2 vdot) c0),  vo_d = m (vt_d + q3 v1_d)                rmnet.py:101-107, 29-31
//         (m [R]: the row mask, hermnet.py:51,56-57; NULL = ones)
//
// and for each: `bwd` = the cotangents of the inputs from the cotangents of the outputs, `bwd2` = the same for `bwd` itself
// (the second-order pass; nothing is differentiated a third time).  Formulas: tests/test_gpu_parity.py holds the torch
// expressions they are checked against (float64 autograd to second order).  One thread per (row, channel quad); the
// LayerNorm kernels one wave per row.  HBM-streaming work, no atomics, deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hermnet_hip.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

struct NodeOpArgs {
  const float* in[12];
  float* out[6];
  long rows;       // rows (MID / OUT / LN) or float4 elements (SILU)
  int H;
  float c0, c1;
};

__device__ __forceinline__ f4 ld(const float* p, size_t off) { return *reinterpret_cast<const f4*>(p + off); }
__device__ __forceinline__ f4 ldz(const float* p, size_t off) { return p ? *reinterpret_cast<const f4*>(p + off) : (f4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void st(float* p, size_t off, f4 v) { *reinterpret_cast<f4*>(p + off) = v; }
__device__ __forceinline__ f4 zero() { return (f4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f4 sqrt4(f4 v) { return (f4){sqrtf(v.x), sqrtf(v.y), sqrtf(v.z), sqrtf(v.w)}; }
__device__ __forceinline__ float sig(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---- SILU ------------------------------------------------------------------------------------------------------------
// bwd:  gx = gy f'(x),  f' = s (1 + x (1 - s))
__global__ __launch_bounds__(256) void silu_bwd_kernel(NodeOpArgs a) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.rows) return;
  const f4 gy = ld(a.in[0], 4 * i), x = ld(a.in[1], 4 * i);
  f4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float s = sig(x[e]);
    o[e] = gy[e] * (s * (1.0f + x[e] * (1.0f - s)));
  }
  st(a.out[0], 4 * i, o);
}
// bwd2: cotangent u of gx -> c_gy = u f'(x),  c_x = u gy f''(x),  f'' = s (1 - s) (2 + x (1 - 2 s))
__global__ __launch_bounds__(256) void silu_bwd2_kernel(NodeOpArgs a) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.rows) return;
  const f4 u = ld(a.in[0], 4 * i), gy = ld(a.in[1], 4 * i), x = ld(a.in[2], 4 * i);
  f4 cg, cx;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float s = sig(x[e]);
    cg[e] = u[e] * (s * (1.0f + x[e] * (1.0f - s)));
    cx[e] = u[e] * gy[e] * (s * (1.0f - s) * (2.0f + x[e] * (1.0f - 2.0f * s)));
  }
  st(a.out[0], 4 * i, cg);
  st(a.out[1], 4 * i, cx);
}

// ---- MID -------------------------------------------------------------------------------------------------------------
#define HN_ROW_QUAD                                              \
  const int Q = a.H >> 2;                                        \
  const long i = (long)blockIdx.x * 256 + threadIdx.x;           \
  if (i >= a.rows * Q) return;                                   \
  const size_t r = (size_t)(i / Q);                              \
  const int c = (int)(i % Q) * 4;                                \
  const int H = a.H

__global__ __launch_bounds__(256) void mid_fwd_kernel(NodeOpArgs a) {
  HN_ROW_QUAD;
  f4 vd = zero(), n2 = zero();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const f4 v1 = ld(a.in[0], (r * 3 + d) * 2 * H + c), v2 = ld(a.in[0], (r * 3 + d) * 2 * H + H + c);
    vd += v1 * v2;
    n2 += v2 * v2;
  }
  st(a.out[0], r * H + c, vd * a.c0);
  st(a.out[1], r * 2 * H + c, ld(a.in[1], r * H + c));
  st(a.out[1], r * 2 * H + H + c, sqrt4(n2 + a.c1));
}
// bwd:  a_ = g_vdot, b_ = g_xin[:, H:]:  g_v1_d = c0 a_ v2_d,  g_v2_d = c0 a_ v1_d + b_ v2_d / n,  g_xt = g_xin[:, :H]
__global__ __launch_bounds__(256) void mid_bwd_kernel(NodeOpArgs a) {
  HN_ROW_QUAD;
  const f4 ga = ld(a.in[0], r * H + c) * a.c0, gb = ld(a.in[1], r * 2 * H + H + c);
  f4 v1[3], v2[3], n2 = zero();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    v1[d] = ld(a.in[2], (r * 3 + d) * 2 * H + c);
    v2[d] = ld(a.in[2], (r * 3 + d) * 2 * H + H + c);
    n2 += v2[d] * v2[d];
  }
  const f4 bn = gb / sqrt4(n2 + a.c1);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    st(a.out[0], (r * 3 + d) * 2 * H + c, ga * v2[d]);
    st(a.out[0], (r * 3 + d) * 2 * H + H + c, ga * v1[d] + bn * v2[d]);
  }
  st(a.out[1], r * H + c, ld(a.in[1], r * 2 * H + c));
}
// bwd2: cotangents (u1 | u2) of g_vp, u_xt of g_xt:
//   c_gvdot = c0 sum_d (u1_d v2_d + u2_d v1_d);  c_gxin = (u_xt | S / n), S = sum_d u2_d v2_d
//   c_v1_d = c0 a_ u2_d;  c_v2_d = c0 a_ u1_d + b_ (u2_d / n - v2_d S / n^3)
__global__ __launch_bounds__(256) void mid_bwd2_kernel(NodeOpArgs a) {
  HN_ROW_QUAD;
  const f4 ga = ld(a.in[2], r * H + c) * a.c0, gb = ld(a.in[3], r * 2 * H + H + c);
  f4 v1[3], v2[3], u1[3], u2[3], n2 = zero(), S = zero(), cg = zero();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    v1[d] = ld(a.in[4], (r * 3 + d) * 2 * H + c);
    v2[d] = ld(a.in[4], (r * 3 + d) * 2 * H + H + c);
    u1[d] = ldz(a.in[0], (r * 3 + d) * 2 * H + c);
    u2[d] = ldz(a.in[0], (r * 3 + d) * 2 * H + H + c);
    n2 += v2[d] * v2[d];
    S += u2[d] * v2[d];
    cg += u1[d] * v2[d] + u2[d] * v1[d];
  }
  const f4 n = sqrt4(n2 + a.c1), rn = 1.0f / n, bn = gb * rn, bs = gb * S * rn * rn * rn;
  st(a.out[0], r * H + c, cg * a.c0);
  st(a.out[1], r * 2 * H + c, ldz(a.in[1], r * H + c));
  st(a.out[1], r * 2 * H + H + c, S * rn);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    st(a.out[2], (r * 3 + d) * 2 * H + c, ga * u2[d]);
    st(a.out[2], (r * 3 + d) * 2 * H + H + c, ga * u1[d] + bn * u2[d] - bs * v2[d]);
  }
}

// ---- OUT -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void out_fwd_kernel(NodeOpArgs a) {
  HN_ROW_QUAD;
  const float m = a.in[5] ? a.in[5][r] : 1.0f;
  const f4 q1 = ld(a.in[0], r * 3 * H + c), q2 = ld(a.in[0], r * 3 * H + H + c), q3 = ld(a.in[0], r * 3 * H + 2 * H + c);
  const f4 vd = ld(a.in[1], r * H + c);
  st(a.out[0], r * H + c, (ld(a.in[3], r * H + c) + (q1 + q2 * vd) * a.c0) * m);
#pragma unroll
  for (int d = 0; d < 3; ++d)
    st(a.out[1], (r * 3 + d) * H + c, (ld(a.in[4], (r * 3 + d) * H + c) + q3 * ld(a.in[2], (r * 3 + d) * 2 * H + c)) * m);
}
// bwd: (gx, gv) -> g_q = m (gx c0 | gx vdot c0 | sum_d gv_d v1_d), g_vdot = m gx q2 c0, g_vp = (m gv_d q3 | 0),
//      g_xt = m gx, g_vt = m gv  (outputs 3, 4; skipped when NULL: without a mask they are gx, gv themselves)
__global__ __launch_bounds__(256) void out_bwd_kernel(NodeOpArgs a) {
  HN_ROW_QUAD;
  const float m = a.in[5] ? a.in[5][r] : 1.0f;
  const f4 gx = ld(a.in[0], r * H + c) * m;
  const f4 q2 = ld(a.in[2], r * 3 * H + H + c), q3 = ld(a.in[2], r * 3 * H + 2 * H + c), vd = ld(a.in[3], r * H + c);
  f4 g3 = zero();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const f4 gv = ldz(a.in[1], (r * 3 + d) * H + c) * m;
    g3 += gv * ld(a.in[4], (r * 3 + d) * 2 * H + c);
    st(a.out[2], (r * 3 + d) * 2 * H + c, gv * q3);
    st(a.out[2], (r * 3 + d) * 2 * H + H + c, zero());
    if (a.out[4]) st(a.out[4], (r * 3 + d) * H + c, gv);
  }
  st(a.out[0], r * 3 * H + c, gx * a.c0);
  st(a.out[0], r * 3 * H + H + c, gx * vd * a.c0);
  st(a.out[0], r * 3 * H + 2 * H + c, g3);
  st(a.out[1], r * H + c, gx * q2 * a.c0);
  if (a.out[3]) st(a.out[3], r * H + c, gx);
}
// bwd2: cotangents c_gq = (k1 | k2 | k3), c_gvdot, c_gvp = (w1 | .), c_gxt, c_gvt (each may be NULL = zero):
//   d_gx = m (k1 c0 + k2 vdot c0 + c_gvdot q2 c0 + c_gxt);   d_gv_d = m (k3 v1_d + w1_d q3 + c_gvt_d)
//   d_q = (0 | m gx c0 c_gvdot | m sum_d w1_d gv_d);  d_vdot = m gx c0 k2;  d_vp = (m gv_d k3 | 0)
__global__ __launch_bounds__(256) void out_bwd2_kernel(NodeOpArgs a) {
  HN_ROW_QUAD;
  const float m = a.in[10] ? a.in[10][r] : 1.0f;
  const f4 k1 = ldz(a.in[0], r * 3 * H + c), k2 = ldz(a.in[0], r * 3 * H + H + c), k3 = ldz(a.in[0], r * 3 * H + 2 * H + c);
  const f4 cvd = ldz(a.in[1], r * H + c), cxt = ldz(a.in[3], r * H + c);
  const f4 gx = ld(a.in[5], r * H + c) * m;
  const f4 q2 = ld(a.in[7], r * 3 * H + H + c), q3 = ld(a.in[7], r * 3 * H + 2 * H + c), vd = ld(a.in[8], r * H + c);
  st(a.out[0], r * H + c, ((k1 + k2 * vd + cvd * q2) * a.c0 + cxt) * m);
  f4 dq3 = zero();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const f4 w1 = ldz(a.in[2], (r * 3 + d) * 2 * H + c), cvt = ldz(a.in[4], (r * 3 + d) * H + c);
    const f4 gv = ldz(a.in[6], (r * 3 + d) * H + c) * m;
    const f4 v1 = ld(a.in[9], (r * 3 + d) * 2 * H + c);
    st(a.out[1], (r * 3 + d) * H + c, (k3 * v1 + w1 * q3 + cvt) * m);
    dq3 += w1 * gv;
    st(a.out[4], (r * 3 + d) * 2 * H + c, gv * k3);
    st(a.out[4], (r * 3 + d) * 2 * H + H + c, zero());
  }
  st(a.out[2], r * 3 * H + c, zero());
  st(a.out[2], r * 3 * H + H + c, gx * cvd * a.c0);
  st(a.out[2], r * 3 * H + 2 * H + c, dq3);
  st(a.out[3], r * H + c, gx * k2 * a.c0);
}

// ---- RES: the residual behind the message (rmnet.py:24-26) with the row mask of hermnet.py:51:
//   x1 = m c0 (x + dx),  v1_d = m (v_d + dv_d)        (v may be NULL: layer 0, vec = 0)
// Linear, so its backward is the mask-and-scale  (g1, gv1) -> (m c0 g1, m gv1)  for BOTH summands, and that map is its own
// backward.
__global__ __launch_bounds__(256) void res_fwd_kernel(NodeOpArgs a) {
  HN_ROW_QUAD;
  const float m = a.in[4] ? a.in[4][r] : 1.0f;
  st(a.out[0], r * H + c, (ld(a.in[0], r * H + c) + ld(a.in[1], r * H + c)) * (m * a.c0));
#pragma unroll
  for (int d = 0; d < 3; ++d)
    st(a.out[1], (r * 3 + d) * H + c, (ldz(a.in[2], (r * 3 + d) * H + c) + ld(a.in[3], (r * 3 + d) * H + c)) * m);
}
__global__ __launch_bounds__(256) void mask_scale_kernel(NodeOpArgs a) {
  HN_ROW_QUAD;
  const float m = a.in[2] ? a.in[2][r] : 1.0f;
  st(a.out[0], r * H + c, ldz(a.in[0], r * H + c) * (m * a.c0));
#pragma unroll
  for (int d = 0; d < 3; ++d) st(a.out[1], (r * 3 + d) * H + c, ldz(a.in[1], (r * 3 + d) * H + c) * m);
}

// ---- LN (one wave per row; lane l: channel quads l, l + 64, ...; H <= 1024) --------------------------------------------
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
constexpr int kLnK = 4;
struct LnRow { f4 y[kLnK]; float r; };
// normalised row and 1/sigma from x (biased variance, eps = c0): what F.layer_norm computed in the forward
__device__ __forceinline__ LnRow ln_row(const float* x, size_t r, int H, float eps, int lane) {
  LnRow o;
  f4 xv[kLnK];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kLnK; ++k) {
    const int c = (k * 64 + lane) * 4;
    xv[k] = c < H ? ld(x, r * H + c) : zero();
    s += (xv[k].x + xv[k].y) + (xv[k].z + xv[k].w);
  }
  const float mu = wsum(s) / (float)H;
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < kLnK; ++k) {
    const int c = (k * 64 + lane) * 4;
    o.y[k] = c < H ? xv[k] - mu : zero();
    v += (o.y[k].x * o.y[k].x + o.y[k].y * o.y[k].y) + (o.y[k].z * o.y[k].z + o.y[k].w * o.y[k].w);
  }
  o.r = rsqrtf(wsum(v) / (float)H + eps);
#pragma unroll
  for (int k = 0; k < kLnK; ++k) o.y[k] *= o.r;
  return o;
}
__device__ __forceinline__ float dot4(f4 a, f4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
__device__ __forceinline__ float sum4(f4 a) { return (a.x + a.y) + (a.z + a.w); }

// bwd:  gx = r (gy - mean(gy) - y mean(gy y))
__global__ __launch_bounds__(256) void ln_bwd_kernel(NodeOpArgs a) {
  const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if ((long)r >= a.rows) return;
  const int lane = threadIdx.x & 63, H = a.H;
  const LnRow n = ln_row(a.in[1], r, H, a.c0, lane);
  f4 g[kLnK];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < kLnK; ++k) {
    const int c = (k * 64 + lane) * 4;
    g[k] = c < H ? ld(a.in[0], r * H + c) : zero();
    s1 += sum4(g[k]);
    s2 += dot4(g[k], n.y[k]);
  }
  const float ma = wsum(s1) / (float)H, mb = wsum(s2) / (float)H;
#pragma unroll
  for (int k = 0; k < kLnK; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < H) st(a.out[0], r * H + c, (g[k] - ma - n.y[k] * mb) * n.r);
  }
}
// bwd2: cotangent v of gx ->  c_gy = r (v - Vm - y Vy)
//   c_x = -r^2 [ y P + b (v - Vm - y Vy) + Vy (gy - a - y b) ],  a = mean(gy), b = mean(gy y), Vm = mean(v), Vy = mean(v y),
//   P = mean(v gy) - a Vm - b Vy
__global__ __launch_bounds__(256) void ln_bwd2_kernel(NodeOpArgs a) {
  const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if ((long)r >= a.rows) return;
  const int lane = threadIdx.x & 63, H = a.H;
  const LnRow n = ln_row(a.in[2], r, H, a.c0, lane);
  f4 v[kLnK], g[kLnK];
  float sa = 0.f, sb = 0.f, sv = 0.f, svy = 0.f, svg = 0.f;
#pragma unroll
  for (int k = 0; k < kLnK; ++k) {
    const int c = (k * 64 + lane) * 4;
    v[k] = c < H ? ld(a.in[0], r * H + c) : zero();
    g[k] = c < H ? ld(a.in[1], r * H + c) : zero();
    sa += sum4(g[k]); sb += dot4(g[k], n.y[k]); sv += sum4(v[k]); svy += dot4(v[k], n.y[k]); svg += dot4(v[k], g[k]);
  }
  const float ih = 1.0f / (float)H;
  const float ma = wsum(sa) * ih, mb = wsum(sb) * ih, Vm = wsum(sv) * ih, Vy = wsum(svy) * ih;
  const float P = wsum(svg) * ih - ma * Vm - mb * Vy;
  const float r2 = n.r * n.r;
#pragma unroll
  for (int k = 0; k < kLnK; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < H) {
      const f4 t = v[k] - Vm - n.y[k] * Vy;
      st(a.out[0], r * H + c, t * n.r);
      st(a.out[1], r * H + c, (n.y[k] * P + t * mb + (g[k] - ma - n.y[k] * mb) * Vy) * (-r2));
    }
  }
}

}  // namespace

// One entry point for the twelve kernels (include/hermnet_hip.h lists the operands of every op).
extern "C" int hermnet_train_node_op(int op, const float* const* in, int num_in, float* const* out, int num_out, long rows,
                                     int hidden, float c0, float c1, void* stream) {
  static const int need_in[13] = {0, 2, 3, 2, 3, 5, 6, 6, 11, 2, 3, 5, 3}, need_out[13] = {0, 1, 2, 2, 2, 3, 2, 5, 5, 1, 2, 2, 2};
  if (op < 1 || op > 12 || !in || !out || num_in != need_in[op] || num_out != need_out[op]) return HN_ERR_BAD_ARG;
  if (rows < 0 || hidden <= 0 || (hidden & 3) != 0 || hidden > 1024) return HN_ERR_BAD_ARG;
  if (rows == 0) return HN_OK;
  NodeOpArgs a = {};
  for (int i = 0; i < num_in; ++i) a.in[i] = in[i];
  for (int i = 0; i < num_out; ++i) a.out[i] = out[i];
  a.rows = rows; a.H = hidden; a.c0 = c0; a.c1 = c1;
  // operands that may be NULL (a zero cotangent, no mask, an output nobody wants)
  auto in_ok = [&](int i) {
    switch (op) {
      case 5: return i <= 1;                 // u_vp, u_xt
      case 6: return i == 5;                 // mask
      case 7: return i == 1 || i == 5;       // gv (the last layer's vec output feeds nothing), mask
      case 8: return i <= 4 || i == 6 || i == 10;      // the five cotangents, gv, mask
      case 11: return i == 2 || i == 4;                // vec (layer 0), mask
      case 12: return true;                            // either cotangent, mask
      default: return false;
    }
  };
  for (int i = 0; i < num_in; ++i)
    if (!in[i] && !in_ok(i)) return HN_ERR_BAD_ARG;
  for (int i = 0; i < num_out; ++i)
    if (!out[i] && !(op == 7 && i >= 3)) return HN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long quads = rows * (hidden >> 2);
  const unsigned gq = (unsigned)((quads + 255) / 256), ge = (unsigned)((rows + 255) / 256), gr = (unsigned)((rows + 3) / 4);
  switch (op) {
    case 1: hipLaunchKernelGGL(silu_bwd_kernel, dim3(ge), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(silu_bwd2_kernel, dim3(ge), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(mid_fwd_kernel, dim3(gq), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(mid_bwd_kernel, dim3(gq), dim3(256), 0, s, a); break;
    case 5: hipLaunchKernelGGL(mid_bwd2_kernel, dim3(gq), dim3(256), 0, s, a); break;
    case 6: hipLaunchKernelGGL(out_fwd_kernel, dim3(gq), dim3(256), 0, s, a); break;
    case 7: hipLaunchKernelGGL(out_bwd_kernel, dim3(gq), dim3(256), 0, s, a); break;
    case 8: hipLaunchKernelGGL(out_bwd2_kernel, dim3(gq), dim3(256), 0, s, a); break;
    case 11: hipLaunchKernelGGL(res_fwd_kernel, dim3(gq), dim3(256), 0, s, a); break;
    case 12: hipLaunchKernelGGL(mask_scale_kernel, dim3(gq), dim3(256), 0, s, a); break;
    case 9: hipLaunchKernelGGL(ln_bwd_kernel, dim3(gr), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(ln_bwd2_kernel, dim3(gr), dim3(256), 0, s, a); break;
  }
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}
