// gfx950 device-side construction of the relation-ordered graph: the build's replacement for
// `in_subgraph` (/root/reference/HermNet/utils.py:11-24, called per relation per layer at
// hermnet.py:52-54).  Instead of N_t O(E) scans per relation per layer, the edge list is grouped
// three times per neighbour list by counting sort (histogram -> scan -> scatter -> per-group rank
// sort; identical to a stable sort by key, deterministic) -- ~25 launches, no host sync, everything
// int32.  HBM-streaming integer work: E * ~100 B.
//
// Orders produced (see include/hermnet_hip.h):
//   rows : atoms sorted by (relation, id), each relation's block starting at row_start[t]
//   CSR  : edges by (row(target), edge id)
//   CSC  : edges by (relation(target), row(source), CSR position)
//   out  : CSR positions by row(source)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/hermnet_hip.h"

#include "scan_i32.h"

namespace {

constexpr int kBlock = 256;
inline dim3 grid_for(long n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

__device__ __forceinline__ int relation_of(long z, const int* __restrict__ zlist, int T) {
  for (int t = 0; t < T; ++t)
    if ((long)zlist[t] == z) return t;     // first matching element; T = "not in elems" (hermnet.py:53)
  return T;
}

__global__ __launch_bounds__(kBlock) void count_relations_kernel(const long* __restrict__ z, int NA,
                                                                const int* __restrict__ zlist, int T,
                                                                int* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < NA) atomicAdd(&counts[relation_of(z[i], zlist, T)], 1);   // integer atomics: exact, order-free
}

__global__ __launch_bounds__(kBlock) void atom_keys_kernel(const long* __restrict__ z, int NA,
                                                          const int* __restrict__ zlist, int T,
                                                          unsigned* __restrict__ key, int* __restrict__ val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < NA) { key[i] = (unsigned)relation_of(z[i], zlist, T); val[i] = i; }
}

// sorted position p -> row = p - first_sorted[rel] + row_start[rel]
__global__ __launch_bounds__(kBlock) void assign_rows_kernel(
    const unsigned* __restrict__ rel_sorted, const int* __restrict__ node_order, int NA,
    const int* __restrict__ row_start, int T, const long* __restrict__ z, int* __restrict__ row_of_node,
    int* __restrict__ z_rows, float* __restrict__ row_real) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= NA) return;
  const int rel = (int)rel_sorted[p];
  // first sorted position of this relation = number of sorted entries with a smaller relation:
  // binary search on the sorted relation list
  int lo = 0, hi = NA;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)rel_sorted[mid] < rel) lo = mid + 1; else hi = mid; }
  const int atom = node_order[p];
  const int row = p - lo + row_start[rel];
  row_of_node[atom] = row;
  z_rows[row] = (int)z[atom];
  row_real[row] = 1.0f;
}

__device__ __forceinline__ int relation_of_row(int row, const int* __restrict__ row_start, int T) {
  int t = 0;
  while (t < T && row >= row_start[t + 1]) ++t;   // row_start[T] = first unknown row
  return t;
}

// row_active = real row of a relation that receives >= 1 edge (hermnet.py:56-57), or the caller's override
__global__ __launch_bounds__(kBlock) void row_active_kernel(const int* __restrict__ csc_rowptr,
                                                           const int* __restrict__ row_start, int N, int T,
                                                           const unsigned char* __restrict__ rel_active,
                                                           const float* __restrict__ row_real,
                                                           float* __restrict__ row_active) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const int t = relation_of_row(r, row_start, T);
  float on = 0.0f;
  if (t < T) {
    const bool act = rel_active ? rel_active[t] != 0
                                : (csc_rowptr[(size_t)(t + 1) * N] - csc_rowptr[(size_t)t * N]) > 0;
    on = act ? row_real[r] : 0.0f;
  }
  row_active[r] = on;
}


// ---- grouping by key without a global sort -------------------------------------------------------
// "Group the indices 0..n-1 by key[i], ascending index inside a group" is what every edge order
// needs (keys: row(target) | relation(target)*N + row(source) | row(source)).  A counting sort does it
// in O(n): histogram (integer atomics), exclusive scan, scatter through per-key cursors (order inside a
// group arbitrary), then a rank sort of every group (groups are neighbour lists: tens of entries).
// The result is deterministic and identical to a stable sort by key.
__global__ __launch_bounds__(kBlock) void hist_kernel(const int* __restrict__ key, int n, int* __restrict__ hist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&hist[key[i]], 1);
}

// `shared_key` (or -1): a key that many entries carry (the closing row of a padded list's NULL edges): its lanes take their
// slots through ONE atomic per wave
__global__ __launch_bounds__(kBlock) void scatter_kernel(const int* __restrict__ key, int n, int* __restrict__ cursor,
                                                        int* __restrict__ slots, int shared_key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = i < n ? key[i] : -2;
  const unsigned long long same = __ballot(k == shared_key);
  if (k == shared_key) {
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)same) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&cursor[k], __popcll(same));
    base = __shfl(base, leader, 64);
    slots[base + __popcll(same & ((1ull << lane) - 1ull))] = i;
  } else if (i < n) {
    slots[atomicAdd(&cursor[k], 1)] = i;
  }
}

// one wave per group: out[rowptr[g] + rank(v)] = v, rank = number of smaller members (members are unique)
__global__ __launch_bounds__(kBlock) void group_rank_sort_kernel(const int* __restrict__ rowptr, int ngroups,
                                                                const int* __restrict__ slots, int* __restrict__ out) {
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (g >= ngroups) return;
  const int lane = threadIdx.x & 63;
  const int beg = rowptr[g], end = rowptr[g + 1];
  for (int a = beg + lane; a < end; a += 64) {
    const int v = slots[a];
    int rank = 0;
    for (int b = beg; b < end; ++b) rank += slots[b] < v ? 1 : 0;
    out[beg + rank] = v;
  }
}

// Zero fill and copy as kernels, not hipMemsetAsync / hipMemcpyAsync: under hipGraph replay (ROCm 7.2) the captured
// memset nodes of this build did not survive eager memsets issued between two replays -- the second replay left the
// histogram un-zeroed (bisected with tools/graph_probe.py: garbage row pointers, then an out-of-bounds rank sort).
__global__ __launch_bounds__(kBlock) void zero_i32_kernel(int* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = 0;
}

__global__ __launch_bounds__(kBlock) void copy_i32_kernel(const int* __restrict__ src, int n, int* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__global__ __launch_bounds__(kBlock) void iota_scaled_kernel(int* __restrict__ dst, int n, int scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = i * scale;
}

struct GroupWork {
  int* cursor;      // [max keys + 1]
  int* slots;       // [n]
  void* scan_temp;
  size_t scan_bytes;
};

// rowptr [nkeys+1], out [n]
int group_by_key(const int* key, int n, int nkeys, int* rowptr, int* out, const GroupWork& w, hipStream_t s) {
  hipLaunchKernelGGL(zero_i32_kernel, grid_for((long)nkeys + 1), dim3(kBlock), 0, s, w.cursor, (long)nkeys + 1);
  if (n > 0) hipLaunchKernelGGL(hist_kernel, grid_for(n), dim3(kBlock), 0, s, key, n, w.cursor);
  if (exclusive_scan_i32(w.cursor, rowptr, nkeys + 1, w.scan_temp, w.scan_bytes, s) != HN_OK) return HN_ERR_LAUNCH;
  if (n == 0) return HN_OK;
  hipLaunchKernelGGL(copy_i32_kernel, grid_for(nkeys), dim3(kBlock), 0, s, rowptr, nkeys, w.cursor);
  hipLaunchKernelGGL(scatter_kernel, grid_for(n), dim3(kBlock), 0, s, key, n, w.cursor, w.slots, -1);
  hipLaunchKernelGGL(group_rank_sort_kernel, dim3((unsigned)((nkeys + 3) / 4)), dim3(kBlock), 0, s, rowptr, nkeys, w.slots, out);
  return HN_OK;
}

// ---- fused edge orders (round 3): both histograms in one pass over the edge list, ONE scan over the concatenated
// counters [ row(target) : N + 1 | relation(target) * N + row(source) : (T + 1) N + 1 ], the CSC scatter inside the CSR
// gather, csc_tgt inside the CSC rank sort: 10 launches instead of 21 for the same (bit-identical) orders.
__global__ __launch_bounds__(kBlock) void edge_keys_hist_kernel(const long* __restrict__ edge_index, int E,
                                                               const int* __restrict__ row_of_node,
                                                               const int* __restrict__ row_start, int T, int N,
                                                               int NA, int* __restrict__ key1, int* __restrict__ key2,
                                                               int* __restrict__ hist) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  long tgt = edge_index[(size_t)E + e];
  const long src = edge_index[e];
  // (an endpoint outside [0, NA) is not followed: the edge is filed as a NULL edge -- a damaged list gives a wrong energy,
  // never an out-of-bounds read or counter)
  if (src < 0 || src >= NA || tgt >= NA) tgt = -1;
  // a NULL edge of a padded list (hermnet_neighbor_fill_padded): the counters' closing slots -- row N of the CSR
  // counters, key (T+1) N of the CSC ones -- so that it lands behind every row and in no segment.  Thousands of them
  // meet ONE counter: the wave adds its count once (an atomic per lane on one address costs ~10 ns each, serialised).
  const unsigned long long nulls = __ballot(tgt < 0);
  if (tgt < 0) {
    key1[e] = N;
    key2[e] = (T + 1) * N;
    if ((int)(threadIdx.x & 63) == __ffsll((long long)nulls) - 1) {
      atomicAdd(&hist[N], __popcll(nulls));
      atomicAdd(&hist[N + 1 + (T + 1) * N], __popcll(nulls));
    }
    return;
  }
  const int rs = row_of_node[src], rt = row_of_node[tgt];
  const int k2 = relation_of_row(rt, row_start, T) * N + rs;        // == T*N + rs for unknown-element targets
  key1[e] = rt;
  key2[e] = k2;
  atomicAdd(&hist[rt], 1);
  atomicAdd(&hist[N + 1 + k2], 1);
}

// scan_apply for the concatenated counters [ n1 + 1 | n2 + ... ]: the running offsets go to `all` (cursor copy for the
// scatters) and, in their final form, straight to csr_rowptr [n1 + 1] and csc_rowptr [n2 + 1] (second part: minus the E
// edges in front).  HVNet: n1 = N, n2 = T N; HTNet: n1 = target rows, n2 = relations x source rows.
// `cursor` (may be `in` itself: every thread has read its own four counters before it writes them): a second copy of the
// offsets, the one the scatters advance -- written here instead of by a copy launch of its own (round 6).
__global__ __launch_bounds__(kBlock) void scan_apply_orders_kernel(const int* in, int n, const int* __restrict__ sums,
                                                                  int* __restrict__ all, int n1, long n2, int E,
                                                                  int* __restrict__ csr_rowptr, int* __restrict__ csc_rowptr,
                                                                  int* cursor) {
  __shared__ int lds[4];
  const int base = blockIdx.x * kScanTile + threadIdx.x * 4;
  int x[4], v = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { x[q] = (base + q < n) ? in[base + q] : 0; v += x[q]; }
  int total;
  int run = block_exclusive_scan(v, lds, total) + sums[blockIdx.x];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = base + q;
    if (i < n) {
      all[i] = run;
      if (cursor != nullptr) cursor[i] = run;
      if (i <= n1) csr_rowptr[i] = run;
      else if (i - (n1 + 1) <= n2) csc_rowptr[i - (n1 + 1)] = run - E;
    }
    run += x[q];
  }
}

// CSR-ordered per-edge arrays, and the scatter of the CSR positions into their CSC groups
__global__ __launch_bounds__(kBlock) void csr_gather_scatter_kernel(
    const long* __restrict__ edge_index, const float* __restrict__ shift, int E, int N,
    const int* __restrict__ row_of_node, const int* __restrict__ csr_perm, const int* __restrict__ key2,
    int* __restrict__ csr_src, int* __restrict__ src_id, int* __restrict__ tgt_id, float* __restrict__ shift_csr,
    int* __restrict__ rt_csr, int* __restrict__ cursor2 /* counters of the second part */, int* __restrict__ slots2,
    const int* __restrict__ csr_rowptr_end /* &csr_rowptr[N] = number of real edges */) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E) return;
  if (k >= csr_rowptr_end[0]) {
    // CSR positions behind the last row: the NULL edges of a padded list.  Benign values (atom 0 onto itself, no
    // shift): the edge-parallel kernels (geometry, radial table) may compute on them, nothing reads the results.
    csr_src[k] = 0; src_id[k] = 0; tgt_id[k] = 0; rt_csr[k] = 0;
    if (shift != nullptr) { shift_csr[3 * k + 0] = 0.f; shift_csr[3 * k + 1] = 0.f; shift_csr[3 * k + 2] = 0.f; }
    return;
  }
  const int e = csr_perm[k];
  const int s = (int)edge_index[e], t = (int)edge_index[(size_t)E + e];
  csr_src[k] = row_of_node[s];
  src_id[k] = s;
  tgt_id[k] = t;
  rt_csr[k] = row_of_node[t];
  if (shift != nullptr) {
    shift_csr[3 * k + 0] = shift[3 * e + 0];
    shift_csr[3 * k + 1] = shift[3 * e + 1];
    shift_csr[3 * k + 2] = shift[3 * e + 2];
  }
  slots2[atomicAdd(&cursor2[key2[e]], 1) - E] = k;
}

// rank sort of the CSC groups (group g = [rp[g] - E, rp[g+1] - E) of the concatenated scan) + csc_tgt
__global__ __launch_bounds__(kBlock) void csc_rank_sort_kernel(const int* __restrict__ rp2, int E, int ngroups,
                                                              const int* __restrict__ slots, const int* __restrict__ rt_csr,
                                                              int* __restrict__ csc_pos, int* __restrict__ csc_tgt) {
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (g >= ngroups) return;
  const int lane = threadIdx.x & 63;
  const int beg = rp2[g] - E, end = rp2[g + 1] - E;
  for (int a = beg + lane; a < end; a += 64) {
    const int v = slots[a];
    int rank = 0;
    for (int b = beg; b < end; ++b) rank += slots[b] < v ? 1 : 0;
    csc_pos[beg + rank] = v;
    csc_tgt[beg + rank] = rt_csr[v];
  }
}

// ---- HTNet (round 3): the triadic relation orders with the same counting sort.  Relation (c; {p, q}) = centre element c,
// unordered pair of neighbour elements; P = T (T + 1) / 2 pairs, enumerated p-major (k(p, q) = p T - p (p - 1) / 2 + q - p).
// A directed edge j -> i is listed once for every pair that contains element(j): expanded edge x = e T + m stands for the
// pair {element(j), m}.  SOURCE rows: the atoms in (element, id) order, blocks of B rows (Ns rows); TARGET rows: block
// (c P + k) B + (position of i inside its element), one block per relation.  All atoms must be of listed elements.
struct TriMap { int T, P, B, Ns; };

__device__ __forceinline__ void tri_keys(const TriMap& m, int rs, int rt, int mm, int& vt, int& k2) {
  const int a = rs / m.B, c = rt / m.B, loc = rt - c * m.B;
  const int p = a < mm ? a : mm, q = a < mm ? mm : a;
  const int rel = c * m.P + p * m.T - p * (p - 1) / 2 + (q - p);
  vt = rel * m.B + loc;
  k2 = rel * m.Ns + rs;
}

__global__ __launch_bounds__(kBlock) void tri_keys_hist_kernel(const long* __restrict__ edge_index, int E0, TriMap m,
                                                              const int* __restrict__ row_of_node, int Nt,
                                                              int* __restrict__ key1, int* __restrict__ hist) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= E0 * m.T) return;
  const int e = x / m.T;
  int vt, k2;
  tri_keys(m, row_of_node[edge_index[e]], row_of_node[edge_index[(size_t)E0 + e]], x - e * m.T, vt, k2);
  key1[x] = vt;
  atomicAdd(&hist[vt], 1);
  atomicAdd(&hist[Nt + 1 + k2], 1);
}

// CSR-ordered per-edge arrays of the expanded list (csr_perm = ORIGINAL edge id), and the scatter into the CSC groups
__global__ __launch_bounds__(kBlock) void tri_gather_scatter_kernel(
    const long* __restrict__ edge_index, const float* __restrict__ shift, int E0, int E, TriMap m,
    const int* __restrict__ row_of_node, const int* __restrict__ perm_x, int* __restrict__ csr_perm,
    int* __restrict__ csr_src, int* __restrict__ src_id, int* __restrict__ tgt_id, float* __restrict__ shift_csr,
    int* __restrict__ rt_csr, int* __restrict__ cursor2, int* __restrict__ slots2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E) return;
  const int x = perm_x[k];
  const int e = x / m.T;
  const int s = (int)edge_index[e], t = (int)edge_index[(size_t)E0 + e];
  const int rs = row_of_node[s];
  int vt, k2;
  tri_keys(m, rs, row_of_node[t], x - e * m.T, vt, k2);
  csr_perm[k] = e;
  csr_src[k] = rs;
  src_id[k] = s;
  tgt_id[k] = t;
  rt_csr[k] = vt;
  if (shift != nullptr) {
    shift_csr[3 * k + 0] = shift[3 * e + 0];
    shift_csr[3 * k + 1] = shift[3 * e + 1];
    shift_csr[3 * k + 2] = shift[3 * e + 2];
  }
  slots2[atomicAdd(&cursor2[k2], 1) - E] = k;
}

// target rows: real = the atom exists; active = its relation has at least one edge (hermnet.py:56-57); res_row = the
// atom's own source row (residual, rmnet.py:24-26)
__global__ __launch_bounds__(kBlock) void tri_rows_kernel(TriMap m, int Nt, const int* __restrict__ elem_counts,
                                                         const int* __restrict__ csr_rowptr,
                                                         const unsigned char* __restrict__ rel_active,
                                                         float* __restrict__ row_real, float* __restrict__ row_active,
                                                         int* __restrict__ res_row) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= Nt) return;
  const int rel = r / m.B, loc = r - rel * m.B, el = rel / m.P;
  const float real = loc < elem_counts[el] ? 1.0f : 0.0f;
  row_real[r] = real;
  const bool act = rel_active ? rel_active[rel] != 0 : (csr_rowptr[(rel + 1) * m.B] - csr_rowptr[rel * m.B]) > 0;
  row_active[r] = act ? real : 0.0f;
  res_row[r] = el * m.B + loc;
}

int bits_for(unsigned max_key_exclusive) {
  int b = 1;
  while (b < 32 && (1u << b) < max_key_exclusive) ++b;
  return b;
}

size_t sort_temp_bytes(int n) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr,
                                           (const int*)nullptr, (int*)nullptr, n, 0, 32, (hipStream_t)0);
  return bytes;
}

size_t work_temp_bytes(int n, int nkeys) {
  const size_t a = sort_temp_bytes(n), b = ((size_t)(nkeys + 1 + 1023) / 1024 + 1) * sizeof(int);   // = scan_temp_bytes
  return a > b ? a : b;
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// atoms sorted by (relation, id) into rows: node_order, row_of_node, z_rows, row_real of `out`
int build_rows(const long* atomic_number, int NA, const int* z_list, int T, const int* row_start, int N,
               const hn_relations_out* out, unsigned* keyA, unsigned* keyB, int* valA, void* temp, size_t tbytes,
               hipStream_t s) {
  hipLaunchKernelGGL(zero_i32_kernel, grid_for(N), dim3(kBlock), 0, s, out->z_rows, (long)N);
  hipLaunchKernelGGL(zero_i32_kernel, grid_for(N), dim3(kBlock), 0, s, reinterpret_cast<int*>(out->row_real), (long)N);
  if (NA > 0) {
    hipLaunchKernelGGL(atom_keys_kernel, grid_for(NA), dim3(kBlock), 0, s, atomic_number, NA, z_list, T, keyA, valA);
    if (hipcub::DeviceRadixSort::SortPairs(temp, tbytes, keyA, keyB, valA, out->node_order, NA, 0, bits_for(T + 1), s)
        != hipSuccess) return HN_ERR_LAUNCH;
    hipLaunchKernelGGL(assign_rows_kernel, grid_for(NA), dim3(kBlock), 0, s, keyB, out->node_order, NA, row_start,
                       T, atomic_number, out->row_of_node, out->z_rows, out->row_real);
  }
  return HN_OK;
}

// Per-step flags of an atom-sharded step (hermnet_shard_step_flags): which (target element, source element) pairs are joined
// by an edge of this rank's list, whether the padded list is complete, whether an atom has left the plan's skin.
__global__ __launch_bounds__(kBlock) void shard_flags_kernel(const long* __restrict__ ei, long columns, const long* __restrict__ z,
                                                             int num_atoms, const long* __restrict__ total, long capacity,
                                                             int* __restrict__ has_in, const float* __restrict__ pos,
                                                             const float* __restrict__ pos_ref, long num_pos, float max_dist2) {
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i < columns) {
    const long src = ei[i], tgt = ei[columns + i];
    int slot = 128 * 128;                                   // NULL edges of a padded list
    if (tgt >= 0 && tgt < num_atoms && src >= 0 && src < num_atoms) {
      const long zt = z[tgt], zs = z[src];
      slot = (int)(zt < 0 ? 0 : (zt > 127 ? 127 : zt)) * 128 + (int)(zs < 0 ? 0 : (zs > 127 ? 127 : zs));
    }
    if (has_in[slot] == 0) has_in[slot] = 1;               // (every writer writes 1)
  }
  if (i < num_pos) {
    const float dx = pos[3 * i] - pos_ref[3 * i], dy = pos[3 * i + 1] - pos_ref[3 * i + 1], dz = pos[3 * i + 2] - pos_ref[3 * i + 2];
    if (dx * dx + dy * dy + dz * dz > max_dist2) has_in[128 * 128 + 2] = 1;
  }
  if (i == 0 && total != nullptr) has_in[128 * 128 + 1] = (total[1] != 0 || total[0] > capacity) ? 1 : 0;
}


}  // namespace

extern "C" int hermnet_relation_counts(const long* atomic_number, int num_atoms, const int* z_list, int num_rel,
                                       int* counts, void* stream) {
  if (num_atoms < 0 || num_rel <= 0 || !z_list || !counts) return HN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(zero_i32_kernel, grid_for(num_rel + 1), dim3(kBlock), 0, s, counts, (long)num_rel + 1);
  if (num_atoms == 0) return HN_OK;
  if (!atomic_number) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(count_relations_kernel, grid_for(num_atoms), dim3(kBlock), 0, s, atomic_number, num_atoms,
                     z_list, num_rel, counts);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" size_t hermnet_build_relations_workspace(int num_atoms, int num_rows, int num_edges, int num_rel) {
  (void)num_rows; (void)num_rel;
  const size_t n = (size_t)(num_edges > num_atoms ? num_edges : num_atoms) + 1;
  const size_t nk = (size_t)(num_rel + 1) * (size_t)num_rows + 2;
  // 5 index buffers of n + cursor and full CSC row pointer of (T+1)*N + sort/scan storage
  const size_t nall = nk + (size_t)num_rows + 4;     // concatenated counters of both edge orders
  return align256(work_temp_bytes((int)n, (int)nall)) + 5 * align256(n * sizeof(unsigned)) + 2 * align256(nall * sizeof(int)) + 256;
}

extern "C" int hermnet_build_relations(const long* atomic_number, const long* edge_index, const float* shift,
                                       int num_atoms, int num_edges, const int* z_list, int num_rel,
                                       const int* row_start, int num_rows,
                                       const unsigned char* rel_active, const hn_relations_out* out, int rows_ready,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  const int NA = num_atoms, E = num_edges, T = num_rel, N = num_rows;
  if (NA < 0 || E < 0 || T <= 0 || N < NA || !out || !z_list || !row_start) return HN_ERR_BAD_ARG;
  if ((size_t)(T + 1) * (size_t)N >= 0xFFFFFFFFull) return HN_ERR_BAD_ARG;   // keys are 32-bit
  if (workspace_bytes < hermnet_build_relations_workspace(NA, N, E, T) || !workspace) return HN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)(E > NA ? E : NA) + 1;
  char* w = reinterpret_cast<char*>(workspace);
  const size_t tb = align256(work_temp_bytes((int)n, (int)((size_t)(T + 1) * N + 2 + (size_t)N + 4)));
  void* temp = w; w += tb;
  unsigned* keyA = reinterpret_cast<unsigned*>(w); w += align256(n * sizeof(unsigned));
  unsigned* keyB = reinterpret_cast<unsigned*>(w); w += align256(n * sizeof(unsigned));
  int* valA = reinterpret_cast<int*>(w); w += align256(n * sizeof(unsigned));
  unsigned* key3 = reinterpret_cast<unsigned*>(w); w += align256(n * sizeof(unsigned));
  unsigned* rt_sorted = reinterpret_cast<unsigned*>(w); w += align256(n * sizeof(unsigned));
  size_t tbytes = tb;

  // ---- rows (rows_ready: node_order / row_of_node / z_rows / row_real of `out` were filled by an earlier call with
  // the same atomic numbers and row layout -- they depend on nothing else)
  if (!rows_ready) {
    const int rcr = build_rows(atomic_number, NA, z_list, T, row_start, N, out, keyA, keyB, valA, temp, tbytes, s);
    if (rcr != HN_OK) return rcr;
  }
  // ---- edge orders by counting sort (histogram -> scan -> scatter -> per-group rank sort; see the kernels above)
  int* key1 = reinterpret_cast<int*>(keyA);
  int* key2 = reinterpret_cast<int*>(keyB);
  int* slots = valA;
  int* rt_csr = reinterpret_cast<int*>(rt_sorted);
  const size_t nall = (size_t)(N + 1) + (size_t)(T + 1) * N + 1;            // concatenated counters
  int* hist = reinterpret_cast<int*>(w); w += align256(sizeof(int) * (nall + 2));
  int* rp_all = reinterpret_cast<int*>(w); w += align256(sizeof(int) * (nall + 2));
  int* cursor = hist;                                                       // the counters become the scatter cursors
  const bool want_out = out->out_rowptr != nullptr && out->out_edges != nullptr;
  hipLaunchKernelGGL(zero_i32_kernel, grid_for((long)nall), dim3(kBlock), 0, s, hist, (long)nall);
  if (E > 0)
    hipLaunchKernelGGL(edge_keys_hist_kernel, grid_for(E), dim3(kBlock), 0, s, edge_index, E, out->row_of_node, row_start, T,
                       N, NA, key1, key2, hist);
  {
    const int n = (int)nall, nb = (n + kScanTile - 1) / kScanTile;
    int* sums = reinterpret_cast<int*>(temp);
    if (tb < scan_temp_bytes(n)) return HN_ERR_BAD_ARG;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(kBlock), 0, s, hist, n, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, s, sums, nb);
    hipLaunchKernelGGL(scan_apply_orders_kernel, dim3(nb), dim3(kBlock), 0, s, hist, n, sums, rp_all, N, (long)T * N, E, out->csr_rowptr,
                       out->csc_rowptr, cursor);
  }
  if (E > 0) {
    // (the scan wrote its offsets to rp_all and, as the copy the scatters advance, back over the counters: `cursor` = hist)
    // CSR: edges grouped by row(target), ascending edge id inside a row
    hipLaunchKernelGGL(scatter_kernel, grid_for(E), dim3(kBlock), 0, s, key1, E, cursor, slots, N);
    hipLaunchKernelGGL(group_rank_sort_kernel, dim3((unsigned)((N + 3) / 4)), dim3(kBlock), 0, s, rp_all, N, slots, out->csr_perm);
    // CSC: CSR positions grouped by (relation(target), row(source)); edges to unknown-element targets fall into the extra
    // key range [T*N, (T+1)*N) and are simply not covered by csc_rowptr[0 .. T*N]
    hipLaunchKernelGGL(csr_gather_scatter_kernel, grid_for(E), dim3(kBlock), 0, s, edge_index, shift, E, N, out->row_of_node,
                       out->csr_perm, key2, out->csr_src, out->src_id, out->tgt_id, out->shift_csr, rt_csr, cursor + N + 1,
                       slots, out->csr_rowptr + N);
    const int ng2 = (T + 1) * N;
    hipLaunchKernelGGL(csc_rank_sort_kernel, dim3((unsigned)((ng2 + 3) / 4)), dim3(kBlock), 0, s, rp_all + N + 1, E, ng2, slots,
                       rt_csr, out->csc_pos, out->csc_tgt);
  }
  // out adjacency: CSR positions grouped by row(source) -- optional: hermnet_edge_geometry_bwd_csc reads the same
  // information from the CSC order
  if (want_out) {
    GroupWork gw;
    gw.slots = slots;
    gw.cursor = hist;
    gw.scan_temp = temp;
    gw.scan_bytes = tb;
    int* ikey3 = reinterpret_cast<int*>(key3);
    if (E > 0) hipLaunchKernelGGL(copy_i32_kernel, grid_for(E), dim3(kBlock), 0, s, out->csr_src, E, ikey3);
    int rcg;
    if ((rcg = group_by_key(ikey3, E, N, out->out_rowptr, out->out_edges, gw, s)) != HN_OK) return rcg;
  }
  hipLaunchKernelGGL(row_active_kernel, grid_for(N), dim3(kBlock), 0, s, out->csc_rowptr, row_start, N, T,
                     rel_active, out->row_real, out->row_active);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" size_t hermnet_build_triadic_workspace(int num_atoms, int num_edges, int num_elem, int block) {
  const size_t T = (size_t)num_elem, P = T * (T + 1) / 2, TR = T * P;
  const size_t Ns = T * (size_t)block, Nt = TR * (size_t)block, E = T * (size_t)num_edges;
  const size_t n = (E > (size_t)num_atoms ? E : (size_t)num_atoms) + 1;
  const size_t nall = (Nt + 1) + TR * Ns + 1;
  return align256(work_temp_bytes((int)n, (int)(nall + 4))) + 4 * align256(n * sizeof(unsigned)) + 2 * align256((nall + 2) * sizeof(int)) +
         align256((T + 1) * sizeof(int)) + 256;
}

extern "C" int hermnet_build_triadic(const long* atomic_number, const long* edge_index, const float* shift,
                                     int num_atoms, int num_edges, const int* z_list, int num_elem, int block,
                                     const int* elem_counts, const unsigned char* rel_active,
                                     const hn_relations_out* out, float* tgt_row_real, int* res_row, int rows_ready,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  const int NA = num_atoms, E0 = num_edges, T = num_elem, B = block;
  if (NA < 0 || E0 < 0 || T <= 0 || B < 0 || !out || !z_list || !elem_counts || !tgt_row_real || !res_row) return HN_ERR_BAD_ARG;
  const long P = (long)T * (T + 1) / 2, TR = T * P, Ns = (long)T * B, Nt = TR * B, E = (long)T * E0;
  if (TR * Ns + Nt + 8 >= 0x7FFFFFFFl || E >= 0x7FFFFFFFl || Ns < NA) return HN_ERR_BAD_ARG;      // keys and counters are int32
  if (!workspace || workspace_bytes < hermnet_build_triadic_workspace(NA, E0, T, B)) return HN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)(E > NA ? E : NA) + 1;
  const size_t nall = (size_t)(Nt + 1) + (size_t)(TR * Ns) + 1;
  char* w = reinterpret_cast<char*>(workspace);
  const size_t tb = align256(work_temp_bytes((int)n, (int)(nall + 4)));
  void* temp = w; w += tb;
  unsigned* keyA = reinterpret_cast<unsigned*>(w); w += align256(n * sizeof(unsigned));
  unsigned* keyB = reinterpret_cast<unsigned*>(w); w += align256(n * sizeof(unsigned));
  int* valA = reinterpret_cast<int*>(w); w += align256(n * sizeof(unsigned));
  int* rt_csr = reinterpret_cast<int*>(w); w += align256(n * sizeof(unsigned));
  int* hist = reinterpret_cast<int*>(w); w += align256(sizeof(int) * (nall + 2));
  int* rp_all = reinterpret_cast<int*>(w); w += align256(sizeof(int) * (nall + 2));
  int* row_start = reinterpret_cast<int*>(w); w += align256(sizeof(int) * (T + 1));
  const TriMap m = {T, (int)P, B, (int)Ns};
  if (!rows_ready) {      // source rows: the HVNet layout with every element padded to `block` rows
    // (row_start[t] = t B, written by a kernel: the stream may be capturing)
    hipLaunchKernelGGL(zero_i32_kernel, grid_for(T + 1), dim3(kBlock), 0, s, row_start, (long)T + 1);
    hipLaunchKernelGGL(iota_scaled_kernel, grid_for(T + 1), dim3(kBlock), 0, s, row_start, T + 1, B);
    const int rcr = build_rows(atomic_number, NA, z_list, T, row_start, (int)Ns, out, keyA, keyB, valA, temp, tb, s);
    if (rcr != HN_OK) return rcr;
  }
  int* key1 = reinterpret_cast<int*>(keyA);
  int* perm_x = reinterpret_cast<int*>(keyB);
  int* slots = valA;
  int* cursor = hist;
  hipLaunchKernelGGL(zero_i32_kernel, grid_for((long)nall), dim3(kBlock), 0, s, hist, (long)nall);
  if (E > 0)
    hipLaunchKernelGGL(tri_keys_hist_kernel, grid_for(E), dim3(kBlock), 0, s, edge_index, E0, m, out->row_of_node, (int)Nt, key1, hist);
  {
    const int nn = (int)nall, nb = (nn + kScanTile - 1) / kScanTile;
    int* sums = reinterpret_cast<int*>(temp);
    if (tb < scan_temp_bytes(nn)) return HN_ERR_BAD_ARG;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(kBlock), 0, s, hist, nn, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, s, sums, nb);
    hipLaunchKernelGGL(scan_apply_orders_kernel, dim3(nb), dim3(kBlock), 0, s, hist, nn, sums, rp_all, (int)Nt, TR * Ns, (int)E,
                       out->csr_rowptr, out->csc_rowptr, cursor);
  }
  if (E > 0) {
    hipLaunchKernelGGL(scatter_kernel, grid_for(E), dim3(kBlock), 0, s, key1, (int)E, cursor, slots, -1);
    hipLaunchKernelGGL(group_rank_sort_kernel, dim3((unsigned)((Nt + 3) / 4)), dim3(kBlock), 0, s, rp_all, (int)Nt, slots, perm_x);
    hipLaunchKernelGGL(tri_gather_scatter_kernel, grid_for(E), dim3(kBlock), 0, s, edge_index, shift, E0, (int)E, m,
                       out->row_of_node, perm_x, out->csr_perm, out->csr_src, out->src_id, out->tgt_id, out->shift_csr, rt_csr,
                       cursor + Nt + 1, slots);
    const int ng2 = (int)(TR * Ns);
    hipLaunchKernelGGL(csc_rank_sort_kernel, dim3((unsigned)((ng2 + 3) / 4)), dim3(kBlock), 0, s, rp_all + Nt + 1, (int)E, ng2, slots,
                       rt_csr, out->csc_pos, out->csc_tgt);
  }
  if (Nt > 0)
    hipLaunchKernelGGL(tri_rows_kernel, grid_for(Nt), dim3(kBlock), 0, s, m, (int)Nt, elem_counts, out->csr_rowptr, rel_active,
                       tgt_row_real, out->row_active, res_row);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_shard_step_flags(const long* edge_index, long columns, const long* atomic_number, int num_atoms,
                                        const long* total, long capacity, int* has_in, const float* pos, const float* pos_ref,
                                        long num_pos, float max_dist2, void* stream) {
  if (columns < 0 || num_atoms < 0 || num_pos < 0 || !has_in || (columns > 0 && (!edge_index || !atomic_number)) ||
      (num_pos > 0 && (!pos || !pos_ref)))
    return HN_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  // (a kernel, not hipMemsetAsync: captured memset nodes do not survive eager memsets between two replays, see above)
  hipLaunchKernelGGL(zero_i32_kernel, grid_for(128 * 128 + 3), dim3(kBlock), 0, s, has_in, (long)(128 * 128 + 3));
  const long n = columns > num_pos ? columns : num_pos;
  hipLaunchKernelGGL(shard_flags_kernel, grid_for(n > 0 ? n : 1), dim3(kBlock), 0, s, edge_index, columns, atomic_number, num_atoms,
                     total, capacity, has_in, pos, pos_ref, num_pos, max_dist2);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}
