// gfx950 kernels of the TRAINING path (train() mode: example/dist_train.py:86-99 differentiates the forces w.r.t. the
// parameters, so every op needs a second derivative).  The per-edge message algebra of PaiNNMessage
// (/root/reference/HermNet/rmnet.py:58-66) is multilinear in its inputs, so its backward and the backward of its backward
// are again per-edge products and channel sums: three streaming kernels replace ~35 elementwise / reduction launches per
// layer of the autograd graph, each reading and writing every [E, 3H] operand once.
//
//   X [E,3H] = x_proj(LayerNorm(x))[source]  (parts Xs | Xa | Xb),  R [E,3H] = rbf_proj(rbf(d)) (Rs | Ra | Rb; the constant
//   factors 1/sqrt(3H), 1/sqrt(H) ride on the projection weights),  V [E,3,H] = vec[source] (NULL in layer 0),  U [E,3] = rhat
//
//   forward     S = Xs Rs                                   [E,H]     -> dx after the row sum
//               M_d = (Xb Rb) U_d + V_d (Xa Ra)             [E,3,H]   -> dvec after the row sum
//   backward    (GS, GM) -> gX, gR, gV, gU                  (first-order cotangents)
//   backward^2  (cX, cR, cV, cU) -> dGS, dGM, dX, dR, dV, dU (cotangents of the backward's outputs; any of c* may be NULL)
//
// Mapping: LPE lanes per edge (power of two >= H/4, at most 64), a lane owns channel quads q = l, l + LPE, ...; sums
// over the channels are shuffles inside the lane group: deterministic, no atomics.  HBM-streaming work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hermnet_hip.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
// streamed once per launch ([E,3H] radial rows and their gradients: 610 MB each at configs[4]'s batch): non-temporal, so that
// the gathered node rows -- read ~20 times each -- keep the caches
#ifndef HN_TRAIN_NT
#define HN_TRAIN_NT 1
#endif
__device__ __forceinline__ f4 ld4s(const float* p) {
  return HN_TRAIN_NT ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)) : *reinterpret_cast<const f4*>(p);
}
__device__ __forceinline__ void st4s(float* p, f4 v) {
  if (HN_TRAIN_NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
  else *reinterpret_cast<f4*>(p) = v;
}
__device__ __forceinline__ f4 ld4z(const float* p, size_t off) { return p ? *reinterpret_cast<const f4*>(p + off) : (f4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
__device__ __forceinline__ float hsum(f4 v) { return (v.x + v.y) + (v.z + v.w); }

template <int LPE>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = 1; m < LPE; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

struct EdgeMsgArgs {
  const float *X, *R, *V, *U;          // inputs of the forward
  const float *GS, *GM;                // cotangents of (S, M)
  const float *cX, *cR, *cV, *cU;      // cotangents of the backward's outputs (each may be null = zero)
  float *o0, *o1, *o2, *o3, *o4, *o5;  // outputs, see the kernels
  long E;
  int H;
  // row indices (each may be null = the edge's own row e): X / cX rows of edge e live at row xi[e] of their arrays,
  // V / cV at vi[e], GS / GM at ti[e] -- the gathers x_j = xh[(relation, source)], vec_j = vec[source] and the
  // cotangent gathers g[target] happen inside the kernels instead of materialising [E, .] copies
  // ri: R (and cR) of edge e is row ri[e]; gR / dR are written to that row as well (R may be stored in another edge order,
  // e.g. sorted by distance bucket: rmnet.BucketedBasis)
  const long *xi, *vi, *ti, *ri;
  // the *_rows kernels: one lane group per OUTPUT ROW g of a grouping of the edges -- its edges are gp[q] (or q itself when
  // gp is null) for q in [grp[g], grp[g+1]) -- and the quantities that are summed over that grouping never leave registers
  const long *grp, *gp;
  long G;
};

__device__ __forceinline__ size_t row_of(const long* idx, long e) { return (size_t)(idx ? idx[e] : e); }

template <int LPE, bool HAS_V>
__global__ __launch_bounds__(256) void edge_msg_fwd_kernel(EdgeMsgArgs a) {
  // o0 = S [E,H], o1 = M [E,3,H]
  constexpr int EPB = 256 / LPE;
  const int l = threadIdx.x % LPE;
  const long e = (long)blockIdx.x * EPB + threadIdx.x / LPE;
  if (e >= a.E) return;
  const int H = a.H, Q = H >> 2;
  const float u0 = a.U[3 * e], u1 = a.U[3 * e + 1], u2 = a.U[3 * e + 2];
  const float* X = a.X + row_of(a.xi, e) * 3 * H;
  const float* R = a.R + row_of(a.ri, e) * 3 * H;
  const size_t bv = row_of(a.vi, e) * 3 * H;
  for (int q = l; q < Q; q += LPE) {
    const int c = 4 * q;
    const f4 xs = ld4(X + c), xa = ld4(X + H + c), xb = ld4(X + 2 * H + c);
    const f4 rs = ld4(R + c), ra = ld4(R + H + c), rb = ld4(R + 2 * H + c);
    st4(a.o0 + (size_t)e * H + c, xs * rs);
    const f4 B = xb * rb;
    f4 m0 = B * u0, m1 = B * u1, m2 = B * u2;
    if (HAS_V) {
      const f4 A = xa * ra;
      const float* V = a.V + bv + c;
      m0 += ld4(V) * A; m1 += ld4(V + H) * A; m2 += ld4(V + 2 * H) * A;
    }
    float* M = a.o1 + (size_t)e * 3 * H + c;
    st4(M, m0); st4(M + H, m1); st4(M + 2 * H, m2);
  }
}

template <int LPE, bool HAS_V>
__global__ __launch_bounds__(256) void edge_msg_bwd_kernel(EdgeMsgArgs a) {
  // o0 = gX [E,3H], o1 = gR [E,3H], o2 = gV [E,3,H] (HAS_V), o3 = gU [E,3]
  constexpr int EPB = 256 / LPE;
  const int l = threadIdx.x % LPE;
  const long e = (long)blockIdx.x * EPB + threadIdx.x / LPE;
  const bool live = e < a.E;              // (dead lane groups still take part in the shuffles)
  const int H = a.H, Q = H >> 2;
  const long ee = live ? e : 0;
  const float u0 = a.U[3 * ee], u1 = a.U[3 * ee + 1], u2 = a.U[3 * ee + 2];
  const size_t b3 = (size_t)ee * 3 * H;                      // the edge's own row: R and every per-edge output
  const size_t bx = row_of(a.xi, ee) * 3 * H, bw = row_of(a.vi, ee) * 3 * H, bt = row_of(a.ti, ee);
  const size_t br = row_of(a.ri, ee) * 3 * H;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (live)
    for (int q = l; q < Q; q += LPE) {
      const int c = 4 * q;
      const f4 xs = ld4(a.X + bx + c), xa = ld4(a.X + bx + H + c), xb = ld4(a.X + bx + 2 * H + c);
      const f4 rs = ld4(a.R + br + c), ra = ld4(a.R + br + H + c), rb = ld4(a.R + br + 2 * H + c);
      const f4 gs = ld4(a.GS + bt * H + c);
      const f4 g0 = ld4(a.GM + bt * 3 * H + c), g1 = ld4(a.GM + bt * 3 * H + H + c), g2 = ld4(a.GM + bt * 3 * H + 2 * H + c);
      const f4 gB = g0 * u0 + g1 * u1 + g2 * u2;
      f4 gXa = (f4){0.f, 0.f, 0.f, 0.f}, gRa = gXa;
      if (HAS_V) {
        const f4 v0 = ld4(a.V + bw + c), v1 = ld4(a.V + bw + H + c), v2 = ld4(a.V + bw + 2 * H + c);
        const f4 gA = g0 * v0 + g1 * v1 + g2 * v2, A = xa * ra;
        gXa = gA * ra; gRa = gA * xa;
        st4(a.o2 + b3 + c, g0 * A); st4(a.o2 + b3 + H + c, g1 * A); st4(a.o2 + b3 + 2 * H + c, g2 * A);
      }
      st4(a.o0 + b3 + c, gs * rs); st4(a.o0 + b3 + H + c, gXa); st4(a.o0 + b3 + 2 * H + c, gB * rb);
      st4(a.o1 + br + c, gs * xs); st4(a.o1 + br + H + c, gRa); st4(a.o1 + br + 2 * H + c, gB * xb);
      const f4 B = xb * rb;
      s0 += hsum(g0 * B); s1 += hsum(g1 * B); s2 += hsum(g2 * B);
    }
  s0 = group_sum<LPE>(s0); s1 = group_sum<LPE>(s1); s2 = group_sum<LPE>(s2);
  if (live && l == 0) { a.o3[3 * e] = s0; a.o3[3 * e + 1] = s1; a.o3[3 * e + 2] = s2; }
}

template <int LPE, bool HAS_V>
__global__ __launch_bounds__(256) void edge_msg_bwd2_kernel(EdgeMsgArgs a) {
  // o0 = dGS [E,H], o1 = dGM [E,3,H], o2 = dX [E,3H], o3 = dR [E,3H], o4 = dV [E,3,H] (HAS_V), o5 = dU [E,3]
  constexpr int EPB = 256 / LPE;
  const int l = threadIdx.x % LPE;
  const long e = (long)blockIdx.x * EPB + threadIdx.x / LPE;
  const bool live = e < a.E;
  const int H = a.H, Q = H >> 2;
  const long ee = live ? e : 0;
  const float u0 = a.U[3 * ee], u1 = a.U[3 * ee + 1], u2 = a.U[3 * ee + 2];
  const float k0 = a.cU ? a.cU[3 * ee] : 0.f, k1 = a.cU ? a.cU[3 * ee + 1] : 0.f, k2 = a.cU ? a.cU[3 * ee + 2] : 0.f;
  const size_t b3 = (size_t)ee * 3 * H;                      // the edge's own row: R and every per-edge output
  const size_t bx = row_of(a.xi, ee) * 3 * H, bw = row_of(a.vi, ee) * 3 * H, bt = row_of(a.ti, ee);
  const size_t br = row_of(a.ri, ee) * 3 * H;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (live)
    for (int q = l; q < Q; q += LPE) {
      const int c = 4 * q;
      const f4 xs = ld4(a.X + bx + c), xa = ld4(a.X + bx + H + c), xb = ld4(a.X + bx + 2 * H + c);
      const f4 rs = ld4(a.R + br + c), ra = ld4(a.R + br + H + c), rb = ld4(a.R + br + 2 * H + c);
      const f4 gs = ld4(a.GS + bt * H + c);
      const f4 g0 = ld4(a.GM + bt * 3 * H + c), g1 = ld4(a.GM + bt * 3 * H + H + c), g2 = ld4(a.GM + bt * 3 * H + 2 * H + c);
      const f4 cXs = ld4z(a.cX, bx + c), cXa = ld4z(a.cX, bx + H + c), cXb = ld4z(a.cX, bx + 2 * H + c);
      const f4 cRs = ld4z(a.cR, br + c), cRa = ld4z(a.cR, br + H + c), cRb = ld4z(a.cR, br + 2 * H + c);
      st4(a.o0 + (size_t)ee * H + c, cXs * rs + cRs * xs);
      const f4 tB = cXb * rb + cRb * xb, B = xb * rb;
      const f4 gB = g0 * u0 + g1 * u1 + g2 * u2, sU = g0 * k0 + g1 * k1 + g2 * k2;
      f4 d0 = tB * u0 + B * k0, d1 = tB * u1 + B * k1, d2 = tB * u2 + B * k2;
      f4 dXa = (f4){0.f, 0.f, 0.f, 0.f}, dRa = dXa;
      if (HAS_V) {
        const f4 v0 = ld4(a.V + bw + c), v1 = ld4(a.V + bw + H + c), v2 = ld4(a.V + bw + 2 * H + c);
        const f4 w0 = ld4z(a.cV, bw + c), w1 = ld4z(a.cV, bw + H + c), w2 = ld4z(a.cV, bw + 2 * H + c);
        const f4 tA = cXa * ra + cRa * xa, A = xa * ra;
        const f4 gA = g0 * v0 + g1 * v1 + g2 * v2, sV = g0 * w0 + g1 * w1 + g2 * w2;
        d0 += tA * v0 + w0 * A; d1 += tA * v1 + w1 * A; d2 += tA * v2 + w2 * A;
        dXa = cRa * gA + ra * sV; dRa = cXa * gA + xa * sV;
        st4(a.o4 + b3 + c, tA * g0); st4(a.o4 + b3 + H + c, tA * g1); st4(a.o4 + b3 + 2 * H + c, tA * g2);
      }
      st4(a.o1 + b3 + c, d0); st4(a.o1 + b3 + H + c, d1); st4(a.o1 + b3 + 2 * H + c, d2);
      st4(a.o2 + b3 + c, cRs * gs); st4(a.o2 + b3 + H + c, dXa); st4(a.o2 + b3 + 2 * H + c, cRb * gB + rb * sU);
      st4(a.o3 + br + c, cXs * gs); st4(a.o3 + br + H + c, dRa); st4(a.o3 + br + 2 * H + c, cXb * gB + xb * sU);
      s0 += hsum(tB * g0); s1 += hsum(tB * g1); s2 += hsum(tB * g2);
    }
  s0 = group_sum<LPE>(s0); s1 = group_sum<LPE>(s1); s2 = group_sum<LPE>(s2);
  if (live && l == 0) { a.o5[3 * e] = s0; a.o5[3 * e + 1] = s1; a.o5[3 * e + 2] = s2; }
}




// ---- the same algebra with the row sums inside (MessageAlgebra: node-level inputs AND outputs).  The per-edge kernels
// above write [E, .] arrays that a segmented sum reads back; here a lane group owns an output row, walks the row's edge
// list and keeps the sums in registers: forward by TARGET row (both outputs are sums), backward and backward-of-backward by
// (relation, source) row -- the rows of xh -- so that gX / dX are sums, gV / dV partial sums per relation (the caller adds
// the T slices: node-sized), and what stays per edge (gR, gU; dGS, dGM, dR, dU) is written as before.
#define HN_GROUP_PROLOGUE                                                         \
  constexpr int GPB = 256 / LPE;                                                  \
  const int l = threadIdx.x % LPE;                                                \
  const long g = (long)blockIdx.x * GPB + threadIdx.x / LPE;                      \
  const bool live = g < a.G;                                                      \
  const int H = a.H, Q = H >> 2;                                                  \
  const long q0 = live ? a.grp[g] : 0, q1 = live ? a.grp[g + 1] : 0

template <int LPE, bool HAS_V>
__global__ __launch_bounds__(256) void edge_msg_fwd_rows_kernel(EdgeMsgArgs a) {
  // o0 = dx [G,H], o1 = dv [G,3,H]
  HN_GROUP_PROLOGUE;
  if (!live) return;
  for (int q = l; q < Q; q += LPE) {
    const int c = 4 * q;
    f4 as = (f4){0.f, 0.f, 0.f, 0.f}, m0 = as, m1 = as, m2 = as;
    for (long k = q0; k < q1; ++k) {
      const long e = a.gp ? a.gp[k] : k;
      const float u0 = a.U[3 * e], u1 = a.U[3 * e + 1], u2 = a.U[3 * e + 2];
      const float* X = a.X + row_of(a.xi, e) * 3 * H;
      const float* R = a.R + row_of(a.ri, e) * 3 * H;
      const f4 xs = ld4(X + c), xa = ld4(X + H + c), xb = ld4(X + 2 * H + c);
      const f4 rs = ld4s(R + c), ra = ld4s(R + H + c), rb = ld4s(R + 2 * H + c);
      as += xs * rs;
      const f4 B = xb * rb;
      m0 += B * u0; m1 += B * u1; m2 += B * u2;
      if (HAS_V) {
        const f4 A = xa * ra;
        const float* V = a.V + row_of(a.vi, e) * 3 * H + c;
        m0 += ld4(V) * A; m1 += ld4(V + H) * A; m2 += ld4(V + 2 * H) * A;
      }
    }
    st4(a.o0 + (size_t)g * H + c, as);
    float* M = a.o1 + (size_t)g * 3 * H + c;
    st4(M, m0); st4(M + H, m1); st4(M + 2 * H, m2);
  }
}

template <int LPE, bool HAS_V>
__global__ __launch_bounds__(256) void edge_msg_bwd_rows_kernel(EdgeMsgArgs a) {
  // o0 = gX summed [G,3H], o1 = gR [rows of R,3H] per edge, o2 = gV summed per group [G,3,H] (HAS_V), o3 = gU [E,3] per edge
  HN_GROUP_PROLOGUE;
  if (!live) return;          // (the channel sums below are shuffles INSIDE a lane group: groups need not walk in step)
  for (int q = l, pass = 0; q < ((Q + LPE - 1) / LPE) * LPE; q += LPE, ++pass) {
    const int c = 4 * q;
    const bool qok = q < Q;
    f4 sXs = (f4){0.f, 0.f, 0.f, 0.f}, sXa = sXs, sXb = sXs, sV0 = sXs, sV1 = sXs, sV2 = sXs;
    for (long k = q0; k < q1; ++k) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      const long e = a.gp ? a.gp[k] : k;
      {
        if (qok) {
          const float u0 = a.U[3 * e], u1 = a.U[3 * e + 1], u2 = a.U[3 * e + 2];
          const size_t bx = row_of(a.xi, e) * 3 * H, bt = row_of(a.ti, e), br = row_of(a.ri, e) * 3 * H;
          const f4 xs = ld4(a.X + bx + c), xa = ld4(a.X + bx + H + c), xb = ld4(a.X + bx + 2 * H + c);
          const f4 rs = ld4s(a.R + br + c), ra = ld4s(a.R + br + H + c), rb = ld4s(a.R + br + 2 * H + c);
          const f4 gs = ld4(a.GS + bt * H + c);
          const f4 g0 = ld4(a.GM + bt * 3 * H + c), g1 = ld4(a.GM + bt * 3 * H + H + c), g2 = ld4(a.GM + bt * 3 * H + 2 * H + c);
          const f4 gB = g0 * u0 + g1 * u1 + g2 * u2;
          f4 gRa = (f4){0.f, 0.f, 0.f, 0.f};
          if (HAS_V) {
            const size_t bw = row_of(a.vi, e) * 3 * H;
            const f4 v0 = ld4(a.V + bw + c), v1 = ld4(a.V + bw + H + c), v2 = ld4(a.V + bw + 2 * H + c);
            const f4 gA = g0 * v0 + g1 * v1 + g2 * v2, A = xa * ra;
            sXa += gA * ra; gRa = gA * xa;
            sV0 += g0 * A; sV1 += g1 * A; sV2 += g2 * A;
          }
          sXs += gs * rs; sXb += gB * rb;
          st4s(a.o1 + br + c, gs * xs); st4s(a.o1 + br + H + c, gRa); st4s(a.o1 + br + 2 * H + c, gB * xb);
          const f4 B = xb * rb;
          s0 = hsum(g0 * B); s1 = hsum(g1 * B); s2 = hsum(g2 * B);
        }
      }
      s0 = group_sum<LPE>(s0); s1 = group_sum<LPE>(s1); s2 = group_sum<LPE>(s2);
      if (l == 0) {
        if (pass == 0) { a.o3[3 * e] = s0; a.o3[3 * e + 1] = s1; a.o3[3 * e + 2] = s2; }
        else { a.o3[3 * e] += s0; a.o3[3 * e + 1] += s1; a.o3[3 * e + 2] += s2; }
      }
    }
    if (qok) {
      float* o = a.o0 + (size_t)g * 3 * H + c;
      st4(o, sXs); st4(o + H, sXa); st4(o + 2 * H, sXb);
      if (HAS_V) {
        float* v = a.o2 + (size_t)g * 3 * H + c;
        st4(v, sV0); st4(v + H, sV1); st4(v + 2 * H, sV2);
      }
    }
  }
}

template <int LPE, bool HAS_V>
__global__ __launch_bounds__(256) void edge_msg_bwd2_rows_kernel(EdgeMsgArgs a) {
  // o0 = dGS [E,H], o1 = dGM [E,3,H] per edge; o2 = dX summed [G,3H]; o3 = dR per edge; o4 = dV summed per group [G,3,H]
  // (HAS_V); o5 = dU [E,3] per edge
  HN_GROUP_PROLOGUE;
  if (!live) return;
  for (int q = l, pass = 0; q < ((Q + LPE - 1) / LPE) * LPE; q += LPE, ++pass) {
    const int c = 4 * q;
    const bool qok = q < Q;
    f4 sXs = (f4){0.f, 0.f, 0.f, 0.f}, sXa = sXs, sXb = sXs, sV0 = sXs, sV1 = sXs, sV2 = sXs;
    for (long k = q0; k < q1; ++k) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      const long e = a.gp ? a.gp[k] : k;
      {
        if (qok) {
          const float u0 = a.U[3 * e], u1 = a.U[3 * e + 1], u2 = a.U[3 * e + 2];
          const float k0 = a.cU ? a.cU[3 * e] : 0.f, k1 = a.cU ? a.cU[3 * e + 1] : 0.f, k2 = a.cU ? a.cU[3 * e + 2] : 0.f;
          const size_t b3 = (size_t)e * 3 * H;
          const size_t bx = row_of(a.xi, e) * 3 * H, bt = row_of(a.ti, e), br = row_of(a.ri, e) * 3 * H;
          const f4 xs = ld4(a.X + bx + c), xa = ld4(a.X + bx + H + c), xb = ld4(a.X + bx + 2 * H + c);
          const f4 rs = ld4s(a.R + br + c), ra = ld4s(a.R + br + H + c), rb = ld4s(a.R + br + 2 * H + c);
          const f4 gs = ld4(a.GS + bt * H + c);
          const f4 g0 = ld4(a.GM + bt * 3 * H + c), g1 = ld4(a.GM + bt * 3 * H + H + c), g2 = ld4(a.GM + bt * 3 * H + 2 * H + c);
          const f4 cXs = ld4z(a.cX, bx + c), cXa = ld4z(a.cX, bx + H + c), cXb = ld4z(a.cX, bx + 2 * H + c);
          const f4 cRs = ld4z(a.cR, br + c), cRa = ld4z(a.cR, br + H + c), cRb = ld4z(a.cR, br + 2 * H + c);
          st4s(a.o0 + (size_t)e * H + c, cXs * rs + cRs * xs);
          const f4 tB = cXb * rb + cRb * xb, B = xb * rb;
          const f4 gB = g0 * u0 + g1 * u1 + g2 * u2, sU = g0 * k0 + g1 * k1 + g2 * k2;
          f4 d0 = tB * u0 + B * k0, d1 = tB * u1 + B * k1, d2 = tB * u2 + B * k2;
          f4 dRa = (f4){0.f, 0.f, 0.f, 0.f};
          if (HAS_V) {
            const size_t bw = row_of(a.vi, e) * 3 * H;
            const f4 v0 = ld4(a.V + bw + c), v1 = ld4(a.V + bw + H + c), v2 = ld4(a.V + bw + 2 * H + c);
            const f4 w0 = ld4z(a.cV, bw + c), w1 = ld4z(a.cV, bw + H + c), w2 = ld4z(a.cV, bw + 2 * H + c);
            const f4 tA = cXa * ra + cRa * xa, A = xa * ra;
            const f4 gA = g0 * v0 + g1 * v1 + g2 * v2, sV = g0 * w0 + g1 * w1 + g2 * w2;
            d0 += tA * v0 + w0 * A; d1 += tA * v1 + w1 * A; d2 += tA * v2 + w2 * A;
            sXa += cRa * gA + ra * sV; dRa = cXa * gA + xa * sV;
            sV0 += tA * g0; sV1 += tA * g1; sV2 += tA * g2;
          }
          st4s(a.o1 + b3 + c, d0); st4s(a.o1 + b3 + H + c, d1); st4s(a.o1 + b3 + 2 * H + c, d2);
          sXs += cRs * gs; sXb += cRb * gB + rb * sU;
          st4s(a.o3 + br + c, cXs * gs); st4s(a.o3 + br + H + c, dRa); st4s(a.o3 + br + 2 * H + c, cXb * gB + xb * sU);
          s0 = hsum(tB * g0); s1 = hsum(tB * g1); s2 = hsum(tB * g2);
        }
      }
      s0 = group_sum<LPE>(s0); s1 = group_sum<LPE>(s1); s2 = group_sum<LPE>(s2);
      if (l == 0) {
        if (pass == 0) { a.o5[3 * e] = s0; a.o5[3 * e + 1] = s1; a.o5[3 * e + 2] = s2; }
        else { a.o5[3 * e] += s0; a.o5[3 * e + 1] += s1; a.o5[3 * e + 2] += s2; }
      }
    }
    if (qok) {
      float* o = a.o2 + (size_t)g * 3 * H + c;
      st4(o, sXs); st4(o + H, sXa); st4(o + 2 * H, sXb);
      if (HAS_V) {
        float* v = a.o4 + (size_t)g * 3 * H + c;
        st4(v, sV0); st4(v + H, sV1); st4(v + 2 * H, sV2);
      }
    }
  }
}

#define HN_EDGE_LAUNCH(KERNEL)                                                                                    \
  do {                                                                                                            \
    const int Q = a.H >> 2;                                                                                       \
    int lpe = 1;                                                                                                  \
    while (lpe < Q && lpe < 64) lpe <<= 1;                                                                        \
    const unsigned blocks = (unsigned)((a.E + (256 / lpe) - 1) / (256 / lpe));                                    \
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);                                                        \
    switch (lpe) {                                                                                                \
      case 1: if (has_v) hipLaunchKernelGGL((KERNEL<1, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<1, false>), dim3(blocks), dim3(256), 0, s, a); break;      \
      case 2: if (has_v) hipLaunchKernelGGL((KERNEL<2, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<2, false>), dim3(blocks), dim3(256), 0, s, a); break;      \
      case 4: if (has_v) hipLaunchKernelGGL((KERNEL<4, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<4, false>), dim3(blocks), dim3(256), 0, s, a); break;      \
      case 8: if (has_v) hipLaunchKernelGGL((KERNEL<8, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<8, false>), dim3(blocks), dim3(256), 0, s, a); break;      \
      case 16: if (has_v) hipLaunchKernelGGL((KERNEL<16, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<16, false>), dim3(blocks), dim3(256), 0, s, a); break;  \
      case 32: if (has_v) hipLaunchKernelGGL((KERNEL<32, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<32, false>), dim3(blocks), dim3(256), 0, s, a); break;  \
      default: if (has_v) hipLaunchKernelGGL((KERNEL<64, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<64, false>), dim3(blocks), dim3(256), 0, s, a); break; \
    }                                                                                                             \
    return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;                                               \
  } while (0)

bool bad_shape(long E, int H) { return E < 0 || H <= 0 || (H & 3) != 0 || (double)E * 3.0 * H >= 9.0e18; }

}  // namespace

extern "C" int hermnet_edge_message_fwd(const float* X, const float* R, const float* V, const float* U, long num_edges,
                                        int hidden, const long* x_rows, const long* v_rows, const long* r_rows, float* S,
                                        float* M, void* stream) {
  if (bad_shape(num_edges, hidden)) return HN_ERR_BAD_ARG;
  if (num_edges == 0) return HN_OK;
  if (!X || !R || !U || !S || !M) return HN_ERR_BAD_ARG;
  EdgeMsgArgs a = {};
  a.X = X; a.R = R; a.V = V; a.U = U; a.o0 = S; a.o1 = M; a.E = num_edges; a.H = hidden; a.xi = x_rows; a.vi = v_rows; a.ri = r_rows;
  const bool has_v = V != nullptr;
  HN_EDGE_LAUNCH(edge_msg_fwd_kernel);
}

extern "C" int hermnet_edge_message_bwd(const float* GS, const float* GM, const float* X, const float* R, const float* V,
                                        const float* U, long num_edges, int hidden, const long* x_rows,
                                        const long* v_rows, const long* t_rows, const long* r_rows, float* gX,
                                        float* gR, float* gV, float* gU, void* stream) {
  if (bad_shape(num_edges, hidden)) return HN_ERR_BAD_ARG;
  if (num_edges == 0) return HN_OK;
  if (!GS || !GM || !X || !R || !U || !gX || !gR || !gU || (V && !gV)) return HN_ERR_BAD_ARG;
  EdgeMsgArgs a = {};
  a.X = X; a.R = R; a.V = V; a.U = U; a.GS = GS; a.GM = GM;
  a.o0 = gX; a.o1 = gR; a.o2 = gV; a.o3 = gU; a.E = num_edges; a.H = hidden;
  a.xi = x_rows; a.vi = v_rows; a.ti = t_rows; a.ri = r_rows;
  const bool has_v = V != nullptr;
  HN_EDGE_LAUNCH(edge_msg_bwd_kernel);
}

extern "C" int hermnet_edge_message_bwd2(const float* cX, const float* cR, const float* cV, const float* cU,
                                         const float* GS, const float* GM, const float* X, const float* R,
                                         const float* V, const float* U, long num_edges, int hidden,
                                         const long* x_rows, const long* v_rows, const long* t_rows,
                                         const long* r_rows, float* dGS, float* dGM, float* dX, float* dR, float* dV,
                                         float* dU, void* stream) {
  if (bad_shape(num_edges, hidden)) return HN_ERR_BAD_ARG;
  if (num_edges == 0) return HN_OK;
  if (!GS || !GM || !X || !R || !U || !dGS || !dGM || !dX || !dR || !dU || (V && !dV)) return HN_ERR_BAD_ARG;
  EdgeMsgArgs a = {};
  a.X = X; a.R = R; a.V = V; a.U = U; a.GS = GS; a.GM = GM; a.cX = cX; a.cR = cR; a.cV = V ? cV : nullptr; a.cU = cU;
  a.o0 = dGS; a.o1 = dGM; a.o2 = dX; a.o3 = dR; a.o4 = dV; a.o5 = dU; a.E = num_edges; a.H = hidden;
  a.xi = x_rows; a.vi = v_rows; a.ti = t_rows; a.ri = r_rows;
  const bool has_v = V != nullptr;
  HN_EDGE_LAUNCH(edge_msg_bwd2_kernel);
}

#define HN_GROUP_LAUNCH(KERNEL)                                                                                   \
  do {                                                                                                            \
    const int Q = a.H >> 2;                                                                                       \
    int lpe = 1;                                                                                                  \
    while (lpe < Q && lpe < 64) lpe <<= 1;                                                                        \
    const unsigned blocks = (unsigned)((a.G + (256 / lpe) - 1) / (256 / lpe));                                    \
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);                                                        \
    switch (lpe) {                                                                                                \
      case 1: if (has_v) hipLaunchKernelGGL((KERNEL<1, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<1, false>), dim3(blocks), dim3(256), 0, s, a); break;      \
      case 2: if (has_v) hipLaunchKernelGGL((KERNEL<2, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<2, false>), dim3(blocks), dim3(256), 0, s, a); break;      \
      case 4: if (has_v) hipLaunchKernelGGL((KERNEL<4, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<4, false>), dim3(blocks), dim3(256), 0, s, a); break;      \
      case 8: if (has_v) hipLaunchKernelGGL((KERNEL<8, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<8, false>), dim3(blocks), dim3(256), 0, s, a); break;      \
      case 16: if (has_v) hipLaunchKernelGGL((KERNEL<16, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<16, false>), dim3(blocks), dim3(256), 0, s, a); break;  \
      case 32: if (has_v) hipLaunchKernelGGL((KERNEL<32, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<32, false>), dim3(blocks), dim3(256), 0, s, a); break;  \
      default: if (has_v) hipLaunchKernelGGL((KERNEL<64, true>), dim3(blocks), dim3(256), 0, s, a); else hipLaunchKernelGGL((KERNEL<64, false>), dim3(blocks), dim3(256), 0, s, a); break; \
    }                                                                                                             \
    return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;                                               \
  } while (0)

extern "C" int hermnet_edge_message_fwd_rows(const float* X, const float* R, const float* V, const float* U, long num_edges,
                                             int hidden, const long* x_rows, const long* v_rows, const long* r_rows,
                                             const long* group_rowptr, const long* group_edges, long num_groups, float* dx,
                                             float* dv, void* stream) {
  if (bad_shape(num_edges, hidden) || num_groups < 0) return HN_ERR_BAD_ARG;
  if (num_groups == 0) return HN_OK;
  if (!X || !R || !U || !dx || !dv || !group_rowptr) return HN_ERR_BAD_ARG;
  EdgeMsgArgs a = {};
  a.X = X; a.R = R; a.V = V; a.U = U; a.o0 = dx; a.o1 = dv; a.E = num_edges; a.H = hidden; a.xi = x_rows; a.vi = v_rows; a.ri = r_rows;
  a.grp = group_rowptr; a.gp = group_edges; a.G = num_groups;
  const bool has_v = V != nullptr;
  HN_GROUP_LAUNCH(edge_msg_fwd_rows_kernel);
}

extern "C" int hermnet_edge_message_bwd_rows(const float* GS, const float* GM, const float* X, const float* R, const float* V,
                                             const float* U, long num_edges, int hidden, const long* x_rows,
                                             const long* v_rows, const long* t_rows, const long* r_rows,
                                             const long* group_rowptr, const long* group_edges, long num_groups,
                                             float* gX_rows, float* gR, float* gV_rows, float* gU, void* stream) {
  if (bad_shape(num_edges, hidden) || num_groups < 0) return HN_ERR_BAD_ARG;
  if (num_groups == 0) return HN_OK;
  if (!GS || !GM || !X || !R || !U || !gX_rows || !gR || !gU || (V && !gV_rows) || !group_rowptr) return HN_ERR_BAD_ARG;
  EdgeMsgArgs a = {};
  a.X = X; a.R = R; a.V = V; a.U = U; a.GS = GS; a.GM = GM;
  a.o0 = gX_rows; a.o1 = gR; a.o2 = gV_rows; a.o3 = gU; a.E = num_edges; a.H = hidden;
  a.xi = x_rows; a.vi = v_rows; a.ti = t_rows; a.ri = r_rows;
  a.grp = group_rowptr; a.gp = group_edges; a.G = num_groups;
  const bool has_v = V != nullptr;
  HN_GROUP_LAUNCH(edge_msg_bwd_rows_kernel);
}

extern "C" int hermnet_edge_message_bwd2_rows(const float* cX, const float* cR, const float* cV, const float* cU,
                                              const float* GS, const float* GM, const float* X, const float* R,
                                              const float* V, const float* U, long num_edges, int hidden,
                                              const long* x_rows, const long* v_rows, const long* t_rows,
                                              const long* r_rows, const long* group_rowptr, const long* group_edges,
                                              long num_groups, float* dGS, float* dGM, float* dX_rows, float* dR,
                                              float* dV_rows, float* dU, void* stream) {
  if (bad_shape(num_edges, hidden) || num_groups < 0) return HN_ERR_BAD_ARG;
  if (num_groups == 0) return HN_OK;
  if (!GS || !GM || !X || !R || !U || !dGS || !dGM || !dX_rows || !dR || !dU || (V && !dV_rows) || !group_rowptr)
    return HN_ERR_BAD_ARG;
  EdgeMsgArgs a = {};
  a.X = X; a.R = R; a.V = V; a.U = U; a.GS = GS; a.GM = GM; a.cX = cX; a.cR = cR; a.cV = V ? cV : nullptr; a.cU = cU;
  a.o0 = dGS; a.o1 = dGM; a.o2 = dX_rows; a.o3 = dR; a.o4 = dV_rows; a.o5 = dU; a.E = num_edges; a.H = hidden;
  a.xi = x_rows; a.vi = v_rows; a.ti = t_rows; a.ri = r_rows;
  a.grp = group_rowptr; a.gp = group_edges; a.G = num_groups;
  const bool has_v = V != nullptr;
  HN_GROUP_LAUNCH(edge_msg_bwd2_rows_kernel);
}

// ---- segmented row sum with an optional gather: out[r] = sum_{q in [rowptr[r], rowptr[r+1])} x[perm ? perm[q] : q]
// (the adjoint of a row gather; rows of `width` floats, width a multiple of 4).  LPR lanes per output row, a lane owns
// column quads c = l, l + LPR, ...; the members of a row are added in list order: deterministic, no atomics, one pass
// over x (torch: index_select into a sorted copy, then segment_reduce).
namespace {

template <int LPR>
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ x, const long* __restrict__ perm,
                                                         const long* __restrict__ rowptr, float* __restrict__ out,
                                                         long n_rows, int width) {
  constexpr int RPB = 256 / LPR;
  const int l = threadIdx.x % LPR;
  const long r = (long)blockIdx.x * RPB + threadIdx.x / LPR;
  if (r >= n_rows) return;
  const long q0 = rowptr[r], q1 = rowptr[r + 1];
  const int Q = width >> 2;
  // (columns beyond the lanes of a row go to further workgroups, blockIdx.y: a sum of a few very wide rows -- the bucketed
  // projection's weight gradient, 21 rows of 12,288 floats -- was six workgroups walking 48 column groups each)
  for (int c = blockIdx.y * LPR + l; c < Q; c += LPR * gridDim.y) {
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
    long q = q0;
    for (; q + 1 < q1; q += 2) {            // two rows in flight
      const long i0 = perm ? perm[q] : q, i1 = perm ? perm[q + 1] : q + 1;
      const f4 v0 = ld4s(x + (size_t)i0 * width + 4 * c), v1 = ld4s(x + (size_t)i1 * width + 4 * c);   // (each row read once)
      acc += v0;
      acc += v1;
    }
    if (q < q1) acc += ld4s(x + (size_t)(perm ? perm[q] : q) * width + 4 * c);
    st4(out + (size_t)r * width + 4 * c, acc);
  }
}

}  // namespace

extern "C" int hermnet_segment_sum(const float* x, const long* perm, const long* rowptr, long num_rows, int width,
                                   float* out, void* stream) {
  if (num_rows < 0 || width <= 0 || (width & 3) != 0) return HN_ERR_BAD_ARG;
  if (num_rows == 0) return HN_OK;
  if (!x || !rowptr || !out) return HN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int Q = width >> 2;
  int lpr = 1;
  while (lpr < Q && lpr < 64) lpr <<= 1;
  const unsigned rblocks = (unsigned)((num_rows + (256 / lpr) - 1) / (256 / lpr));
  const unsigned cblocks = (unsigned)((Q + lpr - 1) / lpr);
  const dim3 blocks(rblocks, cblocks < 1024u ? cblocks : 1024u);
  switch (lpr) {
    case 1: hipLaunchKernelGGL(segment_sum_kernel<1>, blocks, dim3(256), 0, s, x, perm, rowptr, out, num_rows, width); break;
    case 2: hipLaunchKernelGGL(segment_sum_kernel<2>, blocks, dim3(256), 0, s, x, perm, rowptr, out, num_rows, width); break;
    case 4: hipLaunchKernelGGL(segment_sum_kernel<4>, blocks, dim3(256), 0, s, x, perm, rowptr, out, num_rows, width); break;
    case 8: hipLaunchKernelGGL(segment_sum_kernel<8>, blocks, dim3(256), 0, s, x, perm, rowptr, out, num_rows, width); break;
    case 16: hipLaunchKernelGGL(segment_sum_kernel<16>, blocks, dim3(256), 0, s, x, perm, rowptr, out, num_rows, width); break;
    case 32: hipLaunchKernelGGL(segment_sum_kernel<32>, blocks, dim3(256), 0, s, x, perm, rowptr, out, num_rows, width); break;
    default: hipLaunchKernelGGL(segment_sum_kernel<64>, blocks, dim3(256), 0, s, x, perm, rowptr, out, num_rows, width); break;
  }
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

// ---- edge unit vectors (hermnet.py:144-152 with the distance floor of :146-147) and their two derivatives --------------------
// D [E,3] -> U = D / d, d = max(|D|, 1e-6).  Left to torch (norm, where, div, cat and their first and second order backward)
// the step spends ~110 launches on [E,3] / [E] arrays here.  One thread per edge.
//   order 0   U [E,3], d [E]                                                        from D
//   order 1   gD = (gU - U (U.gU)) / d + U gd                                        from gU*, gd*, D     (P = I - U U^T)
//   order 2   cotangent C of gD:  c_gU = (C - U (U.C)) / d,  c_gd = U.C,
//             c_D = [gd P C - ((gU.C) U + (U.gU) P C + (U.C) P gU - (U.C)(U.gU) U) / d] / d          from C, gU*, gd*, D
// (* = may be NULL: zero).  An edge on the floor (|D| <= 1e-6): d is a constant there, U = D / 1e-6.
namespace {

struct UnitArgs {
  const float *D, *gU, *gd, *C;
  float *o0, *o1, *o2;
  long E;
};

template <int ORDER>
__global__ __launch_bounds__(256) void edge_unit_kernel(UnitArgs a) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= a.E) return;
  const float x = a.D[3 * e], y = a.D[3 * e + 1], z = a.D[3 * e + 2];
  const float n = sqrtf(x * x + y * y + z * z);
  const bool floor = n <= 1.0e-6f;
  const float d = floor ? 1.0e-6f : n, inv = 1.0f / d;
  const float u0 = x * inv, u1 = y * inv, u2 = z * inv;
  if (ORDER == 0) {
    a.o0[3 * e] = u0; a.o0[3 * e + 1] = u1; a.o0[3 * e + 2] = u2;
    a.o1[e] = d;
    return;
  }
  const float g0 = a.gU ? a.gU[3 * e] : 0.f, g1 = a.gU ? a.gU[3 * e + 1] : 0.f, g2 = a.gU ? a.gU[3 * e + 2] : 0.f;
  const float gd = a.gd ? a.gd[e] : 0.f;
  const float ug = floor ? 0.f : u0 * g0 + u1 * g1 + u2 * g2;          // (on the floor: no projection, no distance term)
  const float gdl = floor ? 0.f : gd;
  if (ORDER == 1) {
    a.o0[3 * e] = (g0 - u0 * ug) * inv + u0 * gdl;
    a.o0[3 * e + 1] = (g1 - u1 * ug) * inv + u1 * gdl;
    a.o0[3 * e + 2] = (g2 - u2 * ug) * inv + u2 * gdl;
    return;
  }
  const float c0 = a.C[3 * e], c1 = a.C[3 * e + 1], c2 = a.C[3 * e + 2];
  const float uc = floor ? 0.f : u0 * c0 + u1 * c1 + u2 * c2;
  a.o0[3 * e] = (c0 - u0 * uc) * inv; a.o0[3 * e + 1] = (c1 - u1 * uc) * inv; a.o0[3 * e + 2] = (c2 - u2 * uc) * inv;
  a.o1[e] = uc;
  if (floor) {
    a.o2[3 * e] = 0.f; a.o2[3 * e + 1] = 0.f; a.o2[3 * e + 2] = 0.f;
    return;
  }
  const float gc = g0 * c0 + g1 * c1 + g2 * c2;
  const float pc0 = c0 - u0 * uc, pc1 = c1 - u1 * uc, pc2 = c2 - u2 * uc;      // P C
  const float pg0 = g0 - u0 * ug, pg1 = g1 - u1 * ug, pg2 = g2 - u2 * ug;      // P gU
  const float k = gc - uc * ug;
  a.o2[3 * e] = (gd * pc0 - (k * u0 + ug * pc0 + uc * pg0) * inv) * inv;
  a.o2[3 * e + 1] = (gd * pc1 - (k * u1 + ug * pc1 + uc * pg1) * inv) * inv;
  a.o2[3 * e + 2] = (gd * pc2 - (k * u2 + ug * pc2 + uc * pg2) * inv) * inv;
}

}  // namespace

extern "C" int hermnet_edge_unit(int order, const float* D, const float* gU, const float* gd, const float* C, long num_edges,
                                 float* out0, float* out1, float* out2, void* stream) {
  if (order < 0 || order > 2 || num_edges < 0) return HN_ERR_BAD_ARG;
  if (num_edges == 0) return HN_OK;
  if (!D || !out0 || (order != 1 && !out1) || (order == 2 && (!C || !out2))) return HN_ERR_BAD_ARG;
  UnitArgs a = {D, gU, gd, C, out0, out1, out2, num_edges};
  const dim3 grid((unsigned)((num_edges + 255) / 256));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (order == 0) hipLaunchKernelGGL(edge_unit_kernel<0>, grid, dim3(256), 0, s, a);
  else if (order == 1) hipLaunchKernelGGL(edge_unit_kernel<1>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(edge_unit_kernel<2>, grid, dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

// ---- column sums of a tall [T, K, O] array over K (the bias gradients of the node-level linears: 40 per training step) --------
// torch's reduction over a non-innermost axis runs at ~1.2 TB/s on these shapes (25 us for 31 MB, plus a memset); here a
// workgroup sums `rows_per_block` rows of one slice for all O columns (16-byte loads, rows strided over the lane groups, an LDS
// fold at the end) into partial [T, blocks, O]; the caller adds the few partials (a fixed order: deterministic).
namespace {

__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, float* __restrict__ partial, long K, int O,
                                                      int rows_per_block) {
  __shared__ f4 fold[256];
  const int Q = O >> 2, rl = 256 / Q;                  // lane groups (rows in flight) per workgroup
  const int q = threadIdx.x % Q, r = threadIdx.x / Q;
  const long t = blockIdx.y, b = blockIdx.x;
  const long k0 = b * rows_per_block, k1 = min(k0 + (long)rows_per_block, K);
  f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
  if (r < rl) {
    const float* p = x + (t * K) * O + 4 * q;
    for (long k = k0 + r; k < k1; k += rl) acc += ld4s(p + k * O);
  }
  fold[threadIdx.x] = acc;
  __syncthreads();
  if (r == 0) {
    for (int i = 1; i < rl; ++i) acc += fold[i * Q + q];
    st4(partial + ((t * gridDim.x + b) * O) + 4 * q, acc);
  }
}

}  // namespace

extern "C" int hermnet_col_sum(const float* x, long num_slices, long rows, int width, int rows_per_block, float* partial,
                               void* stream) {
  if (num_slices < 0 || rows < 0 || width <= 0 || (width & 3) || width > 1024 || rows_per_block <= 0) return HN_ERR_BAD_ARG;
  if (num_slices == 0 || rows == 0) return HN_OK;
  if (!x || !partial) return HN_ERR_BAD_ARG;
  const long blocks = (rows + rows_per_block - 1) / rows_per_block;
  if (blocks > 65535 * 32 || num_slices > 65535) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(col_sum_kernel, dim3((unsigned)blocks, (unsigned)num_slices), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, partial, rows, width, rows_per_block);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}
