// What does the boundary between two dependent kernels cost on gfx950, and does it depend on how the first one stored?
// The step is a chain of ~60 dependent launches; the kernel trace shows 5.5-11 us between the end of a kernel with a
// large output and the start of the next one, and ~0 behind kernels that write little.  Hypothesis: the release at the
// end of a kernel writes the XCD's dirty L2 lines back; stores that do not leave dirty lines behind would shorten it.
// Producer A<MODE> writes `mb` MB (MODE 0: plain stores, 1: __builtin_nontemporal_store, 2: sc0 sc1 (system scope,
// write-through), 3: nt sc0 sc1), consumer B reads it all.  Prints the time of a chain of (A, B) pairs per pair;
// run under `rocprofv3 --kernel-trace` for the durations and the gaps themselves (tools/micro/kernel_gap.sh).
//   hipcc --offload-arch=gfx950 -O2 kernel_gap.hip -o kernel_gap && ./kernel_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void produce(f4* __restrict__ out, long n4, float s) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f4 v = {s, s + 1.f, s + 2.f, (float)i};
    if (MODE == 0) out[i] = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, out + i);
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(out + i), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(out + i), "v"(v) : "memory");
  }
}

__global__ __launch_bounds__(256) void consume(const f4* __restrict__ in, long n4, float* __restrict__ sink) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f4 v = in[i];
    acc += v.x + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;
}

// the same consumer with a large dynamic LDS allocation (one workgroup per CU, like the message kernels) and 1024 threads
__global__ __launch_bounds__(1024) void consume_lds(const f4* __restrict__ in, long n4, float* __restrict__ sink) {
  extern __shared__ float tile[];
  tile[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  float acc = tile[(threadIdx.x * 7) & 1023];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f4 v = in[i];
    acc += v.x + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;
}

void run_lds(f4* buf, long n4, float* sink, int pairs, int lds_bytes) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(consume_lds), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  for (int i = 0; i < 3; ++i) {
    hipLaunchKernelGGL(produce<0>, dim3(2048), dim3(256), 0, 0, buf, n4, 1.f);
    hipLaunchKernelGGL(consume_lds, dim3(512), dim3(1024), lds_bytes, 0, buf, n4, sink);
  }
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < pairs; ++i) {
    hipLaunchKernelGGL(produce<0>, dim3(2048), dim3(256), 0, 0, buf, n4, (float)i);
    hipLaunchKernelGGL(consume_lds, dim3(512), dim3(1024), lds_bytes, 0, buf, n4, sink);
  }
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("plain + consumer with %3d KB LDS %6.1f MB  %7.2f us per pair\n", lds_bytes / 1024, n4 * 16 / 1e6, ms * 1e3 / pairs);
}

template <int MODE>
void run(const char* name, f4* buf, long n4, float* sink, int pairs) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int grid = 256 * 8;
  for (int i = 0; i < 3; ++i) {
    hipLaunchKernelGGL(produce<MODE>, dim3(grid), dim3(256), 0, 0, buf, n4, 1.f);
    hipLaunchKernelGGL(consume, dim3(grid), dim3(256), 0, 0, buf, n4, sink);
  }
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < pairs; ++i) {
    hipLaunchKernelGGL(produce<MODE>, dim3(grid), dim3(256), 0, 0, buf, n4, (float)i);
    hipLaunchKernelGGL(consume, dim3(grid), dim3(256), 0, 0, buf, n4, sink);
  }
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s %6.1f MB  %7.2f us per (produce, consume) pair\n", name, n4 * 16 / 1e6, ms * 1e3 / pairs);
}

int main(int argc, char** argv) {
  const int pairs = 50;
  float* sink; (void)hipMalloc(&sink, 64);
  for (double mb : {2.0, 8.0, 46.0, 160.0}) {
    const long n4 = (long)(mb * 1e6 / 16);
    f4* buf; (void)hipMalloc(&buf, n4 * 16);
    run<0>("plain", buf, n4, sink, pairs);
    run<1>("nontemporal", buf, n4, sink, pairs);
    run<2>("sc0 sc1", buf, n4, sink, pairs);
    run<3>("sc0 sc1 nt", buf, n4, sink, pairs);
    run_lds(buf, n4, sink, pairs, 8 * 1024);
    run_lds(buf, n4, sink, pairs, 150 * 1024);
    (void)hipFree(buf);
  }
  return 0;
}
