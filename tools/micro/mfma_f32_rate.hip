// Issue rate of v_mfma_f32_32x32x2_f32 on gfx950: W waves per CU (one workgroup per CU), each wave CHAINS independent accumulator
// chains of dependent products, no memory traffic.  Prints cycles per product per SIMD at the reported clock and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O2 mfma_f32_rate.hip -o mfma_f32_rate && ./mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float s) {
  f16v acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int v = 0; v < 16; ++v) acc[c][v] = (float)(c + v);
  float a = s + threadIdx.x * 1e-3f, b = s * 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float r = 0.f;
  for (int c = 0; c < CHAINS; ++c) r += acc[c][0] + acc[c][15];
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int CHAINS>
void run(int waves, int iters) {
  float* out; (void)hipMalloc(&out, 8192);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<CHAINS>, dim3(cus), dim3(waves * 64), 0, 0, out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<CHAINS>, dim3(cus), dim3(waves * 64), 0, 0, out, iters, 1.0f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double n = (double)iters * 8 * CHAINS;                 // products per wave
  const double per_simd = n * (waves > 4 ? waves / 4.0 : 1.0); // products per SIMD (waves <= 4: one wave per SIMD)
  const double clk = p.clockRate * 1e3;
  const double tf = (double)cus * waves * n * 4096.0 / (ms * 1e-3) / 1e12;
  printf("waves/CU %2d  chains %d: %7.3f ms  %6.1f cycles per product per SIMD (at %.2f GHz)  %6.1f TFLOP/s\n", waves, CHAINS, ms,
         ms * 1e-3 * clk / per_simd, clk / 1e9, tf);
  hipFree(out);
}

int main() {
  for (int w : {4, 8, 16}) { run<1>(w, 20000); run<2>(w, 10000); run<4>(w, 5000); }
  return 0;
}
