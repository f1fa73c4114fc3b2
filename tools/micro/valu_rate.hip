// Issue-rate microbenchmark for the fp32 VALU forms the message kernels use (gfx950):
// v_fma_f32, v_pk_fma_f32 (plain and with an op_sel broadcast operand), ds_read_b128 interleaved.
// One workgroup of W waves per CU, independent accumulator chains, no memory traffic.
//   hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float s, f2 sg) {   // sg: kernel argument = wave-uniform, an SGPR pair
  f2 a0 = {1.f, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f},
     a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
  f2 w = {s, s * 1.0001f};
  f2 gg = {s * 0.5f, s * 0.25f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {   // scalar fma: 16 per unroll step (inline asm: the SLP vectoriser would pack them)
#define SFMA(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(w.x), "v"(w.y))
        SFMA(a0.x); SFMA(a0.y); SFMA(a1.x); SFMA(a1.y); SFMA(a2.x); SFMA(a2.y); SFMA(a3.x); SFMA(a3.y);
        SFMA(a4.x); SFMA(a4.y); SFMA(a5.x); SFMA(a5.y); SFMA(a6.x); SFMA(a6.y); SFMA(a7.x); SFMA(a7.y);
      } else if (MODE == 1) {   // packed fma, all operands packed: 8 per unroll step
        a0 = __builtin_elementwise_fma(a0, w, w); a1 = __builtin_elementwise_fma(a1, w, w);
        a2 = __builtin_elementwise_fma(a2, w, w); a3 = __builtin_elementwise_fma(a3, w, w);
        a4 = __builtin_elementwise_fma(a4, w, w); a5 = __builtin_elementwise_fma(a5, w, w);
        a6 = __builtin_elementwise_fma(a6, w, w); a7 = __builtin_elementwise_fma(a7, w, w);
      } else if (MODE == 3 || MODE == 4) {
        // the exact form of the backward kernel's tap loop: {value, derivative} pair (MODE 3: in an SGPR pair, MODE 4: in
        // VGPRs) times a broadcast weight
#define PKS(acc) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(sg), "v"(w))
#define PKV(acc) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(gg), "v"(w))
        if (MODE == 3) { PKS(a0); PKS(a1); PKS(a2); PKS(a3); PKS(a4); PKS(a5); PKS(a6); PKS(a7); }
        else { PKV(a0); PKV(a1); PKV(a2); PKV(a3); PKV(a4); PKV(a5); PKV(a6); PKV(a7); }
      } else {   // packed fma with one operand broadcast from a register half (op_sel), as in the tap loop
        const f2 b0 = {gg.x, gg.x}, b1 = {gg.y, gg.y};
        a0 = __builtin_elementwise_fma(b0, w, a0); a1 = __builtin_elementwise_fma(b1, w, a1);
        a2 = __builtin_elementwise_fma(b0, w, a2); a3 = __builtin_elementwise_fma(b1, w, a3);
        a4 = __builtin_elementwise_fma(b0, w, a4); a5 = __builtin_elementwise_fma(b1, w, a5);
        a6 = __builtin_elementwise_fma(b0, w, a6); a7 = __builtin_elementwise_fma(b1, w, a7);
      }
    }
  }
  const f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (r.x + r.y == 12345.678f) out[threadIdx.x] = r.x;
}

template <int MODE>
double run(int waves, int iters, int fma_per_iter_per_lane) {
  float* out; (void)hipMalloc(&out, 4096);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(waves * 64), 0, 0, out, 10, 1.0f, f2{0.5f, 0.25f});
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(waves * 64), 0, 0, out, iters, 1.0f, f2{0.5f, 0.25f});
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double fmas = (double)cus * waves * 64 * iters * fma_per_iter_per_lane;
  const double clk = p.clockRate * 1e3;   // Hz
  const double cyc = ms * 1e-3 * clk;
  // wave-instructions per SIMD = waves/4 per CU SIMD * iters * instr per iter
  printf("mode %d waves/CU %d: %.3f ms, %.1f TFLOP/s, %.2f FMA lanes/clk/SIMD (clock %.0f MHz)\n", MODE, waves, ms,
         2 * fmas / ms / 1e9, fmas / cyc / (cus * 4), clk / 1e6);
  hipFree(out);
  return ms;
}

int main() {
  for (int waves : {4, 8, 16}) {
    run<0>(waves, 20000, 8 * 16);
    run<1>(waves, 20000, 8 * 16);
    run<2>(waves, 20000, 8 * 16);
    run<3>(waves, 20000, 8 * 16);
    run<4>(waves, 20000, 8 * 16);
  }
  return 0;
}
