// gfx950 kernels for the per-step edge geometry and its force back-propagation.
//
//   hermnet_edge_geometry_fwd  <- HVNet.with_edge              /root/reference/HermNet/hermnet.py:133-152
//   hermnet_edge_geometry_bwd  <- autograd of the same w.r.t. pos (callers: plugin/ase_interface/calculator.py:77-83,
//                                 plugin/lmp_interface/lmp_calc.py:50-56)
// Both are tiny, HBM-streaming kernels (E * ~50 B); one thread per edge / one wave per atom.
#include <hip/hip_runtime.h>
#include "../../include/hermnet_hip.h"

namespace {

__global__ __launch_bounds__(256) void edge_geometry_fwd_kernel(
    const float* __restrict__ pos, const int* __restrict__ src_id, const int* __restrict__ tgt_id,
    const float* __restrict__ shift, const float* __restrict__ cell, const int* __restrict__ batch,
    int E, float4* __restrict__ edge) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int j = src_id[e], i = tgt_id[e];
  float dx = pos[3 * j + 0] - pos[3 * i + 0];
  float dy = pos[3 * j + 1] - pos[3 * i + 1];
  float dz = pos[3 * j + 2] - pos[3 * i + 2];
  if (shift != nullptr) {
    // einsum('ni,nij->nj', edge_shift, cell[batch[j]])  (hermnet.py:139)
    const float* c = cell + 9 * (batch ? batch[j] : 0);
    const float s0 = shift[3 * e + 0], s1 = shift[3 * e + 1], s2 = shift[3 * e + 2];
    dx += s0 * c[0] + s1 * c[3] + s2 * c[6];
    dy += s0 * c[1] + s1 * c[4] + s2 * c[7];
    dz += s0 * c[2] + s1 * c[5] + s2 * c[8];
  }
  float d = sqrtf(dx * dx + dy * dy + dz * dz);
  // isclose(d, 0, rtol=1e-5, atol=1e-6) -> 1e-6   (hermnet.py:146-147)
  if (fabsf(d) <= 1.0e-6f) d = 1.0e-6f;
  edge[e] = make_float4(dx / d, dy / d, dz / d, d);
}

// gpos[a] = sum_{e in out(a)} gD[e] - sum_{e in in(a)} gD[e]; one wave per atom, lanes over edges.
__global__ __launch_bounds__(256) void edge_geometry_bwd_kernel(
    const float4* __restrict__ gD, const int* __restrict__ in_rowptr, const int* __restrict__ in_edges,
    const int* __restrict__ out_rowptr, const int* __restrict__ out_edges, int N, float* __restrict__ gpos) {
  const int a = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (a >= N) return;
  const int lane = threadIdx.x & 63;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int k = out_rowptr[a] + lane; k < out_rowptr[a + 1]; k += 64) {
    const float4 g = gD[out_edges ? out_edges[k] : k];
    sx += g.x; sy += g.y; sz += g.z;
  }
  for (int k = in_rowptr[a] + lane; k < in_rowptr[a + 1]; k += 64) {
    const float4 g = gD[in_edges ? in_edges[k] : k];
    sx -= g.x; sy -= g.y; sz -= g.z;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    sx += __shfl_xor(sx, m, 64);
    sy += __shfl_xor(sy, m, 64);
    sz += __shfl_xor(sz, m, 64);
  }
  if (lane == 0) {
    gpos[3 * a + 0] = sx;
    gpos[3 * a + 1] = sy;
    gpos[3 * a + 2] = sz;
  }
}

// Same sums with the out-edges taken from the CSC order of the relation build (edges by (relation(target),
// row(source))): the out-adjacency of row a is the union of its T CSC segments, so no third edge order is needed.
// Edges whose target has an unknown element are in no segment; they carry no message, hence no gradient.
__global__ __launch_bounds__(256) void edge_geometry_bwd_csc_kernel(
    const float4* __restrict__ gD, const int* __restrict__ in_rowptr, const int* __restrict__ csc_rowptr,
    const int* __restrict__ csc_pos, int T, int N, float* __restrict__ gpos) {
  const int a = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (a >= N) return;
  const int lane = threadIdx.x & 63;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int t = 0; t < T; ++t) {
    const int beg = csc_rowptr[(size_t)t * N + a], end = csc_rowptr[(size_t)t * N + a + 1];
    for (int k = beg + lane; k < end; k += 64) {
      const float4 g = gD[csc_pos[k]];
      sx += g.x; sy += g.y; sz += g.z;
    }
  }
  for (int k = in_rowptr[a] + lane; k < in_rowptr[a + 1]; k += 64) {
    const float4 g = gD[k];
    sx -= g.x; sy -= g.y; sz -= g.z;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    sx += __shfl_xor(sx, m, 64);
    sy += __shfl_xor(sy, m, 64);
    sz += __shfl_xor(sz, m, 64);
  }
  if (lane == 0) {
    gpos[3 * a + 0] = sx;
    gpos[3 * a + 1] = sy;
    gpos[3 * a + 2] = sz;
  }
}

}  // namespace

extern "C" int hermnet_edge_geometry_fwd(const float* pos, const int* src_id, const int* tgt_id,
                                         const float* shift, const float* cell, const int* batch,
                                         int num_edges, float* edge, void* stream) {
  if (num_edges < 0) return HN_ERR_BAD_ARG;
  if (num_edges == 0) return HN_OK;
  if (!pos || !src_id || !tgt_id || !edge) return HN_ERR_BAD_ARG;
  if (shift && !cell) return HN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int block = 256;
  hipLaunchKernelGGL(edge_geometry_fwd_kernel, dim3((num_edges + block - 1) / block), dim3(block), 0, s,
                     pos, src_id, tgt_id, shift, cell, batch, num_edges, reinterpret_cast<float4*>(edge));
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_edge_geometry_bwd(const float* gD, const int* in_rowptr, const int* in_edges,
                                         const int* out_rowptr, const int* out_edges,
                                         int num_nodes, float* gpos, void* stream) {
  if (num_nodes < 0) return HN_ERR_BAD_ARG;
  if (num_nodes == 0) return HN_OK;
  if (!in_rowptr || !out_rowptr || !gpos) return HN_ERR_BAD_ARG;   // gD may be NULL for an edge-less graph (never read)
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int block = 256;  // 4 atoms per block
  hipLaunchKernelGGL(edge_geometry_bwd_kernel, dim3((num_nodes + 3) / 4), dim3(block), 0, s,
                     reinterpret_cast<const float4*>(gD), in_rowptr, in_edges, out_rowptr, out_edges,
                     num_nodes, gpos);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_edge_geometry_bwd_csc(const float* gD, const int* csr_rowptr, const int* csc_rowptr,
                                             const int* csc_pos, int num_rel, int num_nodes, float* gpos,
                                             void* stream) {
  if (num_nodes < 0 || num_rel <= 0) return HN_ERR_BAD_ARG;
  if (num_nodes == 0) return HN_OK;
  if (!csr_rowptr || !csc_rowptr || !gpos) return HN_ERR_BAD_ARG;   // gD / csc_pos may be NULL for an edge-less graph
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(edge_geometry_bwd_csc_kernel, dim3((num_nodes + 3) / 4), dim3(256), 0, s,
                     reinterpret_cast<const float4*>(gD), csr_rowptr, csc_rowptr, csc_pos, num_rel, num_nodes, gpos);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}
