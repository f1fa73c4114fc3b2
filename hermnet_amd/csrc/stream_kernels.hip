// gfx950: a float4 stream copy -- the yardstick `bench.py` prices "fraction of the MEASURED HBM roofline" against
// (MI355X_MICROARCH.md, HBM: 6.29 TB/s for a float4 copy; SURVEY.md 8(d): "measured copy bandwidth on the box").
// Not on the energy/force path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hermnet_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One workgroup streams contiguous 16 KiB pieces (256 lanes x 4 float4 in flight per lane), pieces dealt round-robin
// over the grid so that every memory channel is busy at any time; 16-byte loads and stores, nothing else.
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n4) {
  const size_t piece = (size_t)256 * UNROLL;
  for (size_t base = (size_t)blockIdx.x * piece; base < n4; base += (size_t)gridDim.x * piece) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t i = base + (size_t)u * 256 + threadIdx.x;
      if (i < n4) v[u] = __builtin_nontemporal_load(src + i);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t i = base + (size_t)u * 256 + threadIdx.x;
      if (i < n4) __builtin_nontemporal_store(v[u], dst + i);
    }
  }
}

}  // namespace

extern "C" int hermnet_stream_copy(const float* src, float* dst, size_t num_floats, int workgroups, void* stream) {
  if (!src || !dst || (num_floats & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return HN_ERR_BAD_ARG;
  if (num_floats == 0) return HN_OK;
  const size_t n4 = num_floats / 4;
  size_t pieces = (n4 + 1023) / 1024;
  unsigned grid = workgroups > 0 ? (unsigned)workgroups : 256u * 8u;
  if (pieces < grid) grid = (unsigned)pieces;
  hipLaunchKernelGGL(stream_copy_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst), n4);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}
