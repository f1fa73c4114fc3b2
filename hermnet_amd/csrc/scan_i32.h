// Exclusive prefix sum of int32 in three plain launches, shared by the relation build and the neighbour search (each
// translation unit gets its own copy: everything here sits in an anonymous namespace).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/hermnet_hip.h"

namespace {

constexpr int kScanBlock = 256;

// ---- exclusive prefix sum of int32, three plain launches (block sums, scan of the block sums, local scan + offset).
// Deliberately not hipcub::DeviceScan: a captured step is replayed as a hipGraph, and the library scan's look-back
// state did not survive replays that were interleaved with eager runs (second replay: garbage row pointers ->
// out-of-bounds scatter).  These kernels keep all their state in `temp`, rewritten on every run.
constexpr int kScanTile = 1024;           // elements per block (256 threads x 4)

__device__ __forceinline__ int block_exclusive_scan(int v, int* lds, int& total) {
  // 256 threads: wave scans with shuffles, wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u = __shfl_up(inc, d, 64);
    if (lane >= d) inc += u;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += lds[w];
  total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + inc - v;
}

__global__ __launch_bounds__(kScanBlock) void scan_block_sums_kernel(const int* __restrict__ in, int n, int* __restrict__ sums) {
  __shared__ int lds[4];
  const int base = blockIdx.x * kScanTile + threadIdx.x * 4;
  int v = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) v += (base + q < n) ? in[base + q] : 0;
  int total;
  (void)block_exclusive_scan(v, lds, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanBlock) void scan_sums_kernel(int* __restrict__ sums, int nb) {
  // one block: every thread owns a contiguous chunk of the block sums
  __shared__ int lds[4];
  const int per = (nb + kScanBlock - 1) / kScanBlock;
  const int lo = threadIdx.x * per, hi = min(lo + per, nb);
  int v = 0;
  for (int i = lo; i < hi; ++i) v += sums[i];
  int total;
  int run = block_exclusive_scan(v, lds, total);
  for (int i = lo; i < hi; ++i) { const int x = sums[i]; sums[i] = run; run += x; }
}

__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const int* __restrict__ in, int n, const int* __restrict__ sums,
                                                           int* __restrict__ out) {
  __shared__ int lds[4];
  const int base = blockIdx.x * kScanTile + threadIdx.x * 4;
  int x[4], v = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { x[q] = (base + q < n) ? in[base + q] : 0; v += x[q]; }
  int total;
  int run = block_exclusive_scan(v, lds, total) + sums[blockIdx.x];
#pragma unroll
  for (int q = 0; q < 4; ++q) { if (base + q < n) out[base + q] = run; run += x[q]; }
}

size_t scan_temp_bytes(int n) { return ((size_t)(n + kScanTile - 1) / kScanTile + 1) * sizeof(int); }

int exclusive_scan_i32(const int* in, int* out, int n, void* temp, size_t temp_bytes, hipStream_t s) {
  if (n <= 0) return HN_OK;
  const int nb = (n + kScanTile - 1) / kScanTile;
  if (temp_bytes < scan_temp_bytes(n)) return HN_ERR_BAD_ARG;
  int* sums = reinterpret_cast<int*>(temp);
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(kScanBlock), 0, s, in, n, sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kScanBlock), 0, s, sums, nb);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kScanBlock), 0, s, in, n, sums, out);
  return HN_OK;
}


// the same scan with 64-bit results (edge offsets of the neighbour search)
__global__ __launch_bounds__(kScanBlock) void scan_apply_long_kernel(const int* __restrict__ in, int n, const int* __restrict__ sums,
                                                                    long* __restrict__ out) {
  __shared__ int lds[4];
  const int base = blockIdx.x * kScanTile + threadIdx.x * 4;
  int x[4], v = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { x[q] = (base + q < n) ? in[base + q] : 0; v += x[q]; }
  int total;
  int run = block_exclusive_scan(v, lds, total) + sums[blockIdx.x];
#pragma unroll
  for (int q = 0; q < 4; ++q) { if (base + q < n) out[base + q] = (long)run; run += x[q]; }
}

int exclusive_scan_i32_to_long(const int* in, long* out, int n, void* temp, size_t temp_bytes, hipStream_t s) {
  if (n <= 0) return HN_OK;
  const int nb = (n + kScanTile - 1) / kScanTile;
  if (temp_bytes < scan_temp_bytes(n)) return HN_ERR_BAD_ARG;
  int* sums = reinterpret_cast<int*>(temp);
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(kScanBlock), 0, s, in, n, sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kScanBlock), 0, s, sums, nb);
  hipLaunchKernelGGL(scan_apply_long_kernel, dim3(nb), dim3(kScanBlock), 0, s, in, n, sums, out);
  return HN_OK;
}

}  // namespace
