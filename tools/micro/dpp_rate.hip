// v_fmac_f32 against v_fmac_f32_dpp (row_newbcast) issue rate on gfx950: hipcc --offload-arch=gfx950 tools/micro/dpp_rate.hip -o /tmp/dpp_rate && /tmp/dpp_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
  float r = threadIdx.x * 0.5f, w = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
        asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                     "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(r), "v"(w));
      } else {
        asm volatile("v_fmac_f32_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(r), "v"(w));
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
float run(float* out, int waves_per_simd) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4000, blocks = 256 * waves_per_simd;       // one 256-thread block = one wave per SIMD of a CU
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // instructions per SIMD: waves_per_simd * iters * 64
  return ms * 1e-3f * 2.4e9f / (waves_per_simd * (float)iters * 64.f);    // cycles per instruction per SIMD at 2.4 GHz
}
int main() {
  float* out; hipMalloc(&out, 256 * 256 * 8 * 4);
  for (int w = 1; w <= 4; w *= 2)
    printf("waves/SIMD %d: v_fmac_f32 %.2f cycles per wave-instruction, v_fmac_f32_dpp %.2f\n", w, run<0>(out, w), run<1>(out, w));
  return 0;
}
