// Host-only entry points of libhermnet_hip.so: ABI probe and the CPU evaluation of the banded
// radial contraction (same header math as the device code) used by the GPU-less test-suite.
#include <cmath>
#include <cstddef>
#include "../../include/hermnet_hip.h"
#include "hermnet_math.h"

#define HN_ABI_VERSION 12
#ifndef HN_SRC_HASH
#define HN_SRC_HASH "unknown"
#endif
#define HN_STR2(x) #x
#define HN_STR(x) HN_STR2(x)

extern "C" int hermnet_abi_version(void) { return HN_ABI_VERSION; }

extern "C" const char* hermnet_build_info(void) {
  return "hermnet_hip abi=" HN_STR(HN_ABI_VERSION) " target=gfx950 taps=12 colblock=64 src=" HN_SRC_HASH " built " __DATE__ " " __TIME__;
}

extern "C" int hermnet_host_rbf_row(const float* offset, int R, float inv_rc, float coeff,
                                    int env_kind, int env_p, const float* wt, const float* b, int C,
                                    float d, float* rb, float* drb) {
  if (!offset || !wt || !b || !rb || !drb || R < 2 || C <= 0) return HN_ERR_BAD_ARG;
  const float u = d * inv_rc;
  const HnEnv env = hn_envelope(u, env_kind, env_p);
  const int lo = hn_window_lo(u, R);
  const float c0 = inv_rc * env.der, c1 = inv_rc * env.val * 2.0f * coeff;
  for (int c = 0; c < C; ++c) {
    float s0 = 0.f, s1 = 0.f;
    for (int m = 0; m < HN_TAPS; ++m) {
      const int k = lo + m;
      if (k < 0 || k >= R) continue;   // zero-padded rows on the device
      const float diff = u - offset[k];
      const float g = expf(coeff * (diff * diff));
      s0 = fmaf(g, wt[(size_t)k * C + c], s0);
      s1 = fmaf(g * diff, wt[(size_t)k * C + c], s1);
    }
    rb[c] = fmaf(env.val, s0, b[c]);
    drb[c] = fmaf(c0, s0, c1 * s1);
  }
  return HN_OK;
}
