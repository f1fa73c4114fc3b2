// Host-only entry points of libhermnet_hip.so: ABI probe and the CPU evaluation of the banded
// radial contraction (same header math as the device code) used by the GPU-less test-suite.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include "../../include/hermnet_hip.h"
#include "hermnet_math.h"

#define HN_ABI_VERSION 13
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

// ---- options (ABI v12): process-wide integers read at every launch; they replace the HERMNET_* environment variables the
// library used to read once per process.  Defaults = the measured choices of DESIGN.md; the others exist for tests and A/Bs.
static int hn_opts[HN_NUM_OPTIONS] = {
    8420,   // HN_OPT_FWD_VARIANT      message forward with vec rows: waves * 1000 + VW * 100 + prefetch * 10 + fused
    16420,  // HN_OPT_FWD_VARIANT_L0   layer 0 (vec == 0)
    8420,   // HN_OPT_BWD_VARIANT      the 16-lanes-per-edge backward
    8420,   // HN_OPT_BWD_VARIANT_L0
    0,      // HN_OPT_FWD_ROWS         rows per workgroup (0: sized by the launcher)
    0,      // HN_OPT_BWD_ROWS
    0,      // HN_OPT_BWD_CL_ROWS      channel-per-lane backward: source rows per workgroup (0: whole rounds of one per CU)
    0,      // HN_OPT_BWD_LANES16      1: the 16-lanes-per-edge backward also where the channel-per-lane form could run
    0,      // HN_OPT_NODE_CHAIN_WIDE  1: widths 128 / 256 on the panelled chain kernels (node_chain_wide.hip) too
    2,      // HN_OPT_UPDATE_TILE16    16-row update tiles: 0 never, 1 always, 2 where they shorten the launch
    0,      // HN_OPT_UPDATE_TILE64_MAX  largest grid (in 64-row tiles) that takes the 64-row update kernels at width 128
};
int hn_option(int option) { return hn_opts[option]; }

extern "C" int hermnet_set_option(int option, int value) {
  if (option < 0 || option >= HN_NUM_OPTIONS) return HN_ERR_BAD_ARG;
  hn_opts[option] = value;
  return HN_OK;
}

extern "C" int hermnet_get_option(int option, int* value) {
  if (option < 0 || option >= HN_NUM_OPTIONS || !value) return HN_ERR_BAD_ARG;
  *value = hn_opts[option];
  return HN_OK;
}

// ---- weight fragments (ABI v12): the chain kernels' weight stream from a row-major fp32 weight, on the host.  A binder of the C
// seam calls this once per weight (and again when the weight changes) and uploads the result; the three-plane layout below is
// STABLE from ABI v11 on and documented in include/hermnet_hip.h -- but nobody has to re-implement it.
static inline uint16_t hn_bf16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);      // NaN stays NaN (quiet)
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float hn_bf16_to_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

extern "C" int hermnet_weight_fragments(const float* w_host, int out_features, int in_features, int tile_rows,
                                        unsigned short* frag_host) {
  // tile_rows 32: frag(W)   -- v_mfma_f32_32x32x16_bf16 operands, 32-row blocks x 16-deep k-groups, lane l: row l & 31, k 8 (l >> 5)
  // tile_rows 16: frag16(W) -- v_mfma_f32_16x16x32_bf16 operands, 16-row blocks x 32-deep k-groups, lane l: row l & 15, k 8 (l >> 4)
  if (!w_host || !frag_host || (tile_rows != 32 && tile_rows != 16)) return HN_ERR_BAD_ARG;
  const int BR = tile_rows, KG = tile_rows == 32 ? 16 : 32, G = KG / 8;
  if (out_features <= 0 || in_features <= 0 || out_features % BR || in_features % KG) return HN_ERR_BAD_ARG;
  const int nq = in_features / KG;
  for (int b = 0; b < out_features / BR; ++b)
    for (int q = 0; q < nq; ++q)
      for (int g = 0; g < G; ++g)
        for (int m = 0; m < BR; ++m)
          for (int e = 0; e < 8; ++e) {
            const float w = w_host[(size_t)(b * BR + m) * in_features + q * KG + g * 8 + e];
            const uint16_t p0 = hn_bf16_rne(w);
            const float r1 = w - hn_bf16_to_f32(p0);
            const uint16_t p1 = hn_bf16_rne(r1);
            const uint16_t p2 = hn_bf16_rne(r1 - hn_bf16_to_f32(p1));
            const uint16_t planes[3] = {p2, p1, p0};                         // step s holds plane 2 - s: smallest first
            for (int st = 0; st < 3; ++st)
              frag_host[((((size_t)(b * nq + q) * 3 + st) * G + g) * BR + m) * 8 + e] = planes[st];
          }
  return HN_OK;
}
