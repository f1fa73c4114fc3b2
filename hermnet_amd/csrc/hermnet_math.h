// Per-edge scalar math shared by the gfx950 kernels and the host checker.
//
// Everything here restates reference arithmetic (files under /root/reference):
//   envelope          rmnet.py:175-208   PolynomialEnvelope / ExponentialEnvelope
//   gaussian taps     rmnet.py:156-158   PyG GaussianSmearing(start=0, stop=1, num_gaussians=R)
//   radial basis      rmnet.py:168-172   env(u)[:,None] * rbf(u),  u = d * (1/rc)
// in the same fp32 operation order per tap (u - offset[k], square, * coeff, exp, * env),
// so a tap computed here differs from the reference's by the exp implementation only.
//
// Banding: the Gaussians have width sigma = spacing = 1/(R-1) in u, so a tap k contributes
// exp(-0.5 (t-k)^2) with t = u (R-1).  The device code keeps the 12 taps
// k = floor(t)-5 .. floor(t)+6; every dropped tap is < exp(-18) = 1.6e-8 of the largest one,
// i.e. below fp32 resolution of the sum.  (fp32 itself flushes taps beyond |t-k| > 14.4 to 0,
// SURVEY.md section 7 "Hard parts".)
#pragma once

#if defined(__HIPCC__)
#define HN_HD __host__ __device__ __forceinline__
#else
#define HN_HD inline
#include <cmath>
#endif

#define HN_TAPS 12      // taps kept per edge
#define HN_TAP_BELOW 5  // window = floor(t) - 5 .. floor(t) + 6
#define HN_PAD 11       // zero rows before/after the R weight rows in LDS (window clamp range)
#define HN_CB 64        // channels per column block (one lane group: 32 lanes x 2 or 16 lanes x 4 channels)

struct HnEnv {
  float val;   // env(u)
  float der;   // d env / d u
};

HN_HD float hn_powi(float u, int p) {
  float r = 1.0f;
  for (int i = 0; i < p; ++i) r *= u;
  return r;
}

// rmnet.py:186-193 (polynomial, exponent p) and rmnet.py:206-208 (exponential).
HN_HD HnEnv hn_envelope(float u, int kind, int p) {
  HnEnv e;
  if (!(u < 1.0f)) { e.val = 0.0f; e.der = 0.0f; return e; }
  if (kind == 0) {
    // 1 + a u^p + b u^(p+1) + c u^(p+2)  ==  1 - u^p (1 + p w + p(p+1)/2 w^2),  w = 1 - u  (exact identity).
    // The reference's left-to-right fp32 sum cancels terms of magnitude ~p^2 and carries ~2e-6 absolute
    // noise; this form has no internal cancellation (~1e-7).  d/du = -K u^(p-1) w^2, K = p(p+1)(p+2)/2.
    const float up1 = hn_powi(u, p - 1);   // u^(p-1)
    const float up = up1 * u;
    const float w = 1.0f - u;
    const float fp = (float)p;
    e.val = 1.0f - up * (1.0f + w * (fp + 0.5f * fp * (fp + 1.0f) * w));
    const float K = 0.5f * fp * (fp + 1.0f) * (fp + 2.0f);
    e.der = -K * up1 * w * w;
  } else {
    const float den = (1.0f - u) * (1.0f + u);
    const float q = -(u * u) / den;
#if defined(__HIP_DEVICE_COMPILE__)
    e.val = __expf(q);
#else
    e.val = expf(q);
#endif
    e.der = -(2.0f * u) / (den * den) * e.val;
  }
  return e;
}

// Lowest tap index of the window for scaled distance u (may be negative or >= R-1;
// the weight tile is zero-padded by HN_PAD rows on both sides).
HN_HD int hn_window_lo(float u, int R) {
  float t = u * (float)(R - 1);
  // clamp before the float->int conversion: u can be large for beyond-cutoff edges
  t = fminf(fmaxf(t, 0.0f), (float)(R + HN_TAP_BELOW));   // also maps NaN to 0
  int lo = (int)t - HN_TAP_BELOW;   // t >= 0 so the cast is floor
  return lo;                        // in [-5, R] -> padded row lo + HN_PAD in [6, R + 11]
}
