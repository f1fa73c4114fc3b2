// Internal interface between message_kernels.hip (C-ABI entry points) and message_bwd_cl.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define HN_EDGE_TABLE_FLOATS 32   // floats per edge record of hermnet_edge_radial_table (include/hermnet_hip.h)

struct HnBwdClArgs {
  int N, Nsrc, E, T;         // target rows, source rows, edges, relations
  int identity;              // 1: source and target rows coincide, the residual's identity gradient is added here
  const int* type_rowptr;    // [T+1] device: rows below type_rowptr[T] are targets of a known type
  const int* csc_rowptr;     // [T*Nsrc+1]
  const int* csc_tgt;        // [E]
  const int* csc_pos;        // [E]
  int R, H;
  const float* table;        // [E + 1, 32] per-edge radial record, CSC order (record E repeats E-1)
  const float4* edge;        // [E] (rhat, d), CSR order
  const float* xh;           // [T, Nsrc, 3H]
  const float* xh_bias;      // [T, 3H] or null
  const float* vec;          // [Nsrc, 3, H] or null
  const float* wt;           // [T, R, 3H]
  const float* brbf;         // [T, 3H]
  const float* gx1;          // [N, H]
  const float* gvec1;        // [N, 3, H]
  float* gxh;                // [T, Nsrc, 3H]
  float* gvec;               // workspace [T, Nsrc, 3, H] partial sums per relation (null when vec is null)
  float* gvec_out;           // [Nsrc, 3, H] or null
  float* gx;                 // [Nsrc, H]
  float4* gedge;             // [H/64, E]
  int rows_per_block;
  int xcd_remap;             // 1 = XCD-contiguous workgroup order (message_kernels.hip: xcd_contiguous)
  // atom shards: this launch covers the SOURCE rows of `num_ranges` disjoint ascending ranges [lo, hi) only (the rows
  // whose gradients travel to other ranks first, the rest while they travel); 0 ranges = every row
  const int* src_ranges;     // [num_ranges][2] device
  int num_ranges;
  // tap-row windows (set by hn_bwd_cl_launch when the whole tile does not fit the LDS, num_rbf > 137): this launch stages
  // the padded tile rows [win_base, win_base + win_rows) only and OWNS the edges whose first tap row lies in
  // [win_lo, win_hi); every other edge contributes nothing and its gedge slot is left alone.  win_accumulate = 1: the
  // row sums are added to what the launch over the other window wrote.
  int win_base, win_rows, win_lo, win_hi, win_accumulate;
};

size_t hn_bwd_cl_lds_bytes(int R);
// `ranges_host`: the same [num_ranges][2] values on the host (grid size)
int hn_bwd_cl_launch(HnBwdClArgs a, bool has_vec, int rows_override, const int* ranges_host, hipStream_t s);
