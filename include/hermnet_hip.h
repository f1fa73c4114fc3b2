/*
 * hermnet_hip.h -- C ABI of the MI355X (gfx950) hot-path library `libhermnet_hip.so`.
 *
 * The reference has no FFI: its hot path is reached through Python operators.
 * Each entry point below names the reference operator (file:line under
 * /root/reference) it replaces; INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add at that call site.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - all buffers are allocated and owned by the caller (PyTorch in our host
 *     code); the library never allocates, frees or retains them;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work
 *     on it (no host synchronisation, graph-capturable);
 *   - return value 0 = ok, non-zero = HN_ERR_* (host code raises RuntimeError);
 *   - float = IEEE fp32 (the reference is fp32-only), indices = int32 (int64 where a parameter says so:
 *     the caller's edge_index / atomic_number / halo index lists arrive as torch LongTensors).
 *
 * ABI version 13 (`hermnet_abi_version`): v13 is ADDITIVE over v12 (hermnet_band_product / _grad_a / _grad_b / _grads, hermnet_basis_window,
 * hermnet_edge_unit, hermnet_col_sum: the training path's rbf_proj on the bucketed basis and its neighbours); v12 is ADDITIVE over v11 (hermnet_halo_proj_rows / _accumulate, ranged launches of
 * hermnet_message_scatter_bwd without the finishing launch, hermnet_set_option / _get_option in place of the library's environment
 * variables, hermnet_weight_fragments; no signature, struct or fragment format of v11 changed).  STABLE from v11 on: hn_graph,
 * hn_rbf_desc, hn_pending_grads, the frag(W) / frag16(W) weight streams, and every entry point's argument list -- later versions
 * only add entry points; v11 changes the weight fragment formats of the node chain kernels (three bf16 planes:
 * see "chain kernels on the matrix pipe"); v10 adds hermnet_shard_step_flags; v9 added the fused layer-boundary node kernels (hermnet_node_update_pre_fwd, the `gxh`
 * form of hn_pending_grads, hermnet_node_pre_fwd16 / _bwd16) and hermnet_param_guard; v8 the gradients handed down as partial sums
 * (hn_pending_grads); v7 adds the row windows of the message kernels (interior / boundary launches
 * around the halo exchange), node chain kernels for every width that is a multiple of 64 up to 512, hermnet_stream_copy, and drops
 * the float-atomic mode 3 of hermnet_halo_rows; v6 added the target mask of the neighbour search (lists of an atom shard) and the
 * row windows of the node pre kernels (halo exchange overlap); v5 replaces the stand-alone node GEMM by the node chain kernels
 * (hermnet_node_pre_fwd/_bwd, hermnet_node_update_fwd/_bwd); v4 puts the radial table in CSC order (hermnet_edge_radial_table takes the
 * graph); v3 added the deterministic halo accumulate, separate source / target row
 * spaces (HTNet) and the fused node-chain kernels; v2 added the bias-on-load arguments, LayerNorm, the energy head, the
 * CSC position gradient and the halo packing; the Python side refuses a library of another version.
 *
 * Node order.  All node-level arrays are in "relation order": atoms sorted by
 * (relation index of their element, original id); atoms whose element is not in
 * the model's element list come last.  `type_rowptr[T+1]` delimits the row range
 * of each relation; rows >= type_rowptr[T] are unknown-type atoms (sources only).
 *
 * Edge orders.  CSR: edges sorted by (row(target), original edge id).
 *               CSC: edges sorted by (relation(target), row(source), CSR position).
 */
#ifndef HERMNET_HIP_H
#define HERMNET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HN_OK 0
#define HN_ERR_BAD_ARG 1      /* unsupported shape / null pointer */
#define HN_ERR_LDS 2          /* weight tile does not fit the 160 KiB LDS */
#define HN_ERR_LAUNCH 3       /* hipLaunchKernel / hipGetLastError failed */

#define HN_ENV_POLYNOMIAL 0   /* rmnet.py:175-193 */
#define HN_ENV_EXPONENTIAL 1  /* rmnet.py:196-208 */

/* Library / build identification (also used as the "is the native path loaded" probe). */
int hermnet_abi_version(void);
const char* hermnet_build_info(void);

/* Options (ABI v12): process-wide integers the launchers read at every call -- tuning knobs and the alternative kernel forms
 * that tests and A/Bs compare against.  (They replace the HERMNET_* environment variables that the library read once per process
 * up to ABI v11.)  Defaults are the measured choices; a binder never needs to touch them. */
#define HN_OPT_FWD_VARIANT 0        /* message forward with vec rows: waves * 1000 + VW * 100 + prefetch * 10 + fused (8420) */
#define HN_OPT_FWD_VARIANT_L0 1     /* ... of layer 0, vec == 0 (16420) */
#define HN_OPT_BWD_VARIANT 2        /* the 16-lanes-per-edge backward (8420) */
#define HN_OPT_BWD_VARIANT_L0 3
#define HN_OPT_FWD_ROWS 4           /* target rows per workgroup; 0 = sized by the launcher */
#define HN_OPT_BWD_ROWS 5
#define HN_OPT_BWD_CL_ROWS 6        /* channel-per-lane backward: source rows per workgroup; 0 = whole rounds of one per CU */
#define HN_OPT_BWD_LANES16 7        /* 1: the 16-lanes-per-edge backward even where the channel-per-lane form could run */
#define HN_OPT_NODE_CHAIN_WIDE 8    /* 1: widths 128 / 256 on the panelled chain kernels too */
#define HN_OPT_UPDATE_TILE16 9      /* 16-row update tiles: 0 never, 1 always, 2 (default) where they shorten the launch */
#define HN_OPT_UPDATE_TILE64_MAX 10 /* largest grid (in 64-row tiles) that takes the 64-row update kernels at width 128 */
#define HN_NUM_OPTIONS 11
int hermnet_set_option(int option, int value);
int hermnet_get_option(int option, int* value);

/* The chain kernels' weight stream from a row-major fp32 weight W [out_features][in_features], computed on the HOST (both
 * pointers are host pointers): tile_rows = 32 gives frag(W), 16 gives frag16(W) (see "chain kernels on the matrix pipe"),
 * out_features * in_features * 3 bf16 values.  The format is STABLE from ABI v11 on; this routine exists so that no binder has
 * to re-implement it (hermnet_amd/nodeops.py: weight_fragments is the same map in torch ops, checked bit for bit). */
int hermnet_weight_fragments(const float* w_host, int out_features, int in_features, int tile_rows, unsigned short* frag_host);

/* Radial-basis description shared by the message kernels:
 *   u = d * inv_rc; env(u) per rmnet.py:175-208; Gaussian taps exp(coeff*(u-offset[k])^2)
 *   (PyG GaussianSmearing as used at rmnet.py:156-158: offset = linspace(0,1,R),
 *   coeff = -0.5/(offset[1]-offset[0])^2). */
typedef struct hn_rbf_desc {
  const float* offset;   /* [R] device */
  int   num_rbf;         /* R */
  float inv_rc;          /* 1/rc */
  float coeff;           /* Gaussian coefficient (negative) */
  int   env_kind;        /* HN_ENV_* */
  int   env_p;           /* polynomial exponent p */
} hn_rbf_desc;

/* Graph in relation order (built once per neighbour list by the host code). */
typedef struct hn_graph {
  int num_nodes;              /* N */
  int num_edges;              /* E */
  int num_rel;                /* T */
  const int* type_rowptr;     /* [T+1] */
  const int* csr_rowptr;      /* [N+1]  by target row */
  const int* csr_src;         /* [E]    source row of each CSR edge */
  const int* csc_rowptr;      /* [T*N+1] by (relation(target), source row) */
  const int* csc_tgt;         /* [E]    target row of each CSC edge */
  const int* csc_pos;         /* [E]    CSR position of each CSC edge */
  /* Separate source-row space (HTNet: a target atom appears once per triadic relation, "virtual" target rows,
   * while sources stay one row per atom).  num_src = 0 and res_row = NULL: sources and targets share rows (HVNet). */
  int num_src;                /* rows of xh / vec / gxh / gvec and of csr_src, csc_rowptr ([T*num_src+1]); 0 = num_nodes */
  const int* res_row;         /* [N] source row whose (x, vec) enter the residual of target row r (rmnet.py:24-26), or NULL = r */
} hn_graph;

/* ---- A16 / K18-K19: neighbor_search (data.py:14-24; ase primitive_neighbor_list / torch_cluster
 * radius_graph on the host in the reference).  Device-side cell list, float64 arithmetic on the
 * float32 coordinates; pair (i, j, S) iff |pos_j - pos_i + S cell| < rc (strict), no (i,i,0),
 * output sorted by (i, j, Sx, Sy, Sz).  Two calls around one host read:
 *   hermnet_neighbor_count  -> total_device[0] = E, total_device[1] = flags (bit 0: an |S| component exceeded 8 --
 *                              coordinates many cells away from the cell: wrap them or use the host path; bit 1: an
 *                              atom has more pairs than the per-atom key stash holds); keeps its state in `workspace`
 *   hermnet_neighbor_fill   -> edge_index [2,E] int64, edge_shift [E,3] = shift_sign * S.  stash_ok = 1 (flag bit 1
 *                              clear): the keys stashed by the counting pass are rank-sorted per atom and decoded --
 *                              no second pass over the candidates, no global sort; stash_ok = 0: two-pass form through
 *                              `keys` [E] (may be NULL when stash_ok = 1).
 * cell_host: 9 doubles (rows = lattice vectors) on the HOST, or NULL for an open system, in which
 * case lo_host/hi_host give the bounding box of the coordinates.  source_first = 1 writes rows
 * [j; i] (radius_graph convention: source, target), 0 writes [i; j] (the reference's periodic path).
 * target_ok (ABI v6; NULL = every atom): [num_atoms] bytes, only pairs whose TARGET atom (row 1 of edge_index) is
 * flagged are counted and listed -- the list of an atom shard (owned + halo atoms in, edges into owned atoms out);
 * pass the same pointer to both calls. */
/* hermnet_neighbor_fill_padded (ABI v7; SURVEY 8(f) row 1: no host round-trip inside an MD step): instead of reading E
 * between the two calls, the caller provides `capacity` columns; the pairs found fill the first E of them, the rest
 * become NULL edges (-1, -1; shift 0), and total_device = (E, flags) is written for a read at the END of the step
 * (flags as above, plus bit 2: E > capacity).  With bit 1 or 2 set the list is incomplete: repeat the search in its
 * two-call form with a larger capacity.  hermnet_build_relations files NULL edges behind every row (they are in no CSR /
 * CSC segment), so every kernel of the step runs on the padded arrays with num_edges = capacity -- a fixed launch
 * geometry: search + step can be captured into ONE hipGraph that stays valid across list rebuilds.  Call after
 * hermnet_neighbor_count on the same workspace. */
int hermnet_neighbor_fill_padded(int num_atoms, void* workspace, size_t workspace_bytes, long capacity, float shift_sign,
                                 int source_first, long* edge_index /* [2, capacity] */, float* edge_shift /* [capacity, 3] */,
                                 long* total_device /* [2] */, void* stream);
size_t hermnet_neighbor_workspace(int num_atoms);      /* with the largest stash slot (160 keys per atom: 1.3 kB/atom) */
/* ... or with a smaller one (8 .. 160 keys per atom): every call of a search derives the slot size from the workspace
 * size it is handed, so the workspace a caller allocates decides it.  An atom with more pairs than the slot holds sets
 * flag bit 1: the exact search then takes its two-pass form, the padded one must be repeated with a larger slot. */
size_t hermnet_neighbor_workspace_for(int num_atoms, int stash_per_atom);
int hermnet_neighbor_count(const float* pos, int num_atoms, const double* cell_host, const double* lo_host,
                           const double* hi_host, double rc, void* workspace, size_t workspace_bytes,
                           const unsigned char* target_ok, long* total_device /* [2] */, void* stream);
int hermnet_neighbor_fill(const float* pos, int num_atoms, const double* cell_host, const double* lo_host,
                          const double* hi_host, double rc, void* workspace, size_t workspace_bytes,
                          long num_edges, float shift_sign, int source_first, int stash_ok,
                          unsigned long long* keys, const unsigned char* target_ok, long* edge_index,
                          float* edge_shift, void* stream);

/* ---- A13: in_subgraph (utils.py:11-24) replaced by a one-off device-side build of the relation-ordered
 * graph per neighbour list (three stable radix sorts + binary-searched row pointers, no host sync).
 * Step 1 counts the atoms of each relation (counts[T+1] device ints, last = element not in z_list);
 * the host turns the counts into the row layout `row_start[T+1]` (first row of every relation block,
 * then the first row of the unknown-element atoms) and `num_rows`; step 2 fills `out`. */
typedef struct hn_relations_out {
  int* node_order;    /* [NA]    atoms sorted by (relation, id) */
  int* row_of_node;   /* [NA]    row of every atom */
  int* z_rows;        /* [N]     atomic number per row (0 for padding rows) */
  float* row_real;    /* [N]     1 for rows that hold an atom */
  float* row_active;  /* [N]     1 for real rows of relations that receive edges (hermnet.py:56-57) */
  int* csr_rowptr;    /* [N+1] */
  int* csr_src;       /* [E]     source row, CSR order */
  int* csr_perm;      /* [E]     original edge id of each CSR edge */
  int* src_id;        /* [E]     original source / target atom ids in CSR order (edge geometry) */
  int* tgt_id;        /* [E] */
  float* shift_csr;   /* [E,3]   edge_shift in CSR order (ignored when `shift` is NULL) */
  int* csc_rowptr;    /* [T*N+1] */
  int* csc_tgt;       /* [E] */
  int* csc_pos;       /* [E] */
  int* out_rowptr;    /* [N+1]   edges by source row (position gradient); NULL (with out_edges) = not wanted */
  int* out_edges;     /* [E]     CSR positions */
} hn_relations_out;

int hermnet_relation_counts(const long* atomic_number, int num_atoms, const int* z_list, int num_rel,
                            int* counts, void* stream);
size_t hermnet_build_relations_workspace(int num_atoms, int num_rows, int num_edges, int num_rel);
/* edge_index [2,E] int64 (row 0 = source, row 1 = target, hermnet.py:135); shift [E,3] or NULL;
 * rel_active [T] bytes or NULL (NULL: a relation is active iff it receives at least one edge).
 * rows_ready != 0: `out->node_order / row_of_node / z_rows / row_real` already hold the row layout of these atomic
 * numbers (they depend on atomic_number, z_list and row_start only, not on the edges: a caller that evaluates the same
 * atoms again -- every MD step -- passes the arrays of its previous call); only the edge orders are built. */
int hermnet_build_relations(const long* atomic_number, const long* edge_index, const float* shift,
                            int num_atoms, int num_edges, const int* z_list, int num_rel,
                            const int* row_start, int num_rows, const unsigned char* rel_active,
                            const hn_relations_out* out, int rows_ready, void* workspace, size_t workspace_bytes,
                            void* stream);

/* HTNet (README.md:27 "Heterogeneous Triadic Networks"; build-defined): the relation orders of the
 * triadic graph with the same counting sort (ABI v6).  Relation (c; {p, q}) = centre element c and an unordered pair of
 * neighbour elements, T * P of them (P = T (T + 1) / 2, pairs enumerated p-major); a directed edge j -> i is listed
 * once per pair that contains element(j): E = T * num_edges entries.  SOURCE rows = the atoms in (element, id) order,
 * every element padded to `block` rows (Ns = T * block); TARGET rows = one block per relation (Nt = T * P * block).
 * Every atom must be of a listed element (the host falls back to its torch build otherwise).
 * out: node_order, row_of_node [NA], z_rows, row_real [Ns] (source rows; skipped when rows_ready), row_active [Nt],
 * csr_rowptr [Nt+1], csr_src / csr_perm (ORIGINAL edge id) / src_id / tgt_id [E], shift_csr [E,3], csc_rowptr
 * [T P Ns + 1] (groups = relation * Ns + source row), csc_tgt, csc_pos [E]; out_rowptr / out_edges unused.
 * elem_counts [T] device ints (atoms per element); tgt_row_real [Nt]; res_row [Nt] = the atom's own source row;
 * rel_active [T P] bytes or NULL (NULL: a relation is active iff it receives at least one edge HERE; an atom shard
 * passes the flags of the whole structure). */
size_t hermnet_build_triadic_workspace(int num_atoms, int num_edges, int num_elem, int block);
int hermnet_build_triadic(const long* atomic_number, const long* edge_index, const float* shift, int num_atoms,
                          int num_edges, const int* z_list, int num_elem, int block, const int* elem_counts,
                          const unsigned char* rel_active, const hn_relations_out* out, float* tgt_row_real,
                          int* res_row, int rows_ready,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- A2: HVNet.with_edge (hermnet.py:133-152) -------------------------------------------
 * edge[e] = (rx, ry, rz, d) for CSR edge e, D = pos[src] - pos[tgt] (+ shift @ cell[batch[src]]),
 * d = |D| with d ~ 0 (atol 1e-6) replaced by 1e-6, r = D / d.
 * `src_id`/`tgt_id` are ORIGINAL atom ids in CSR edge order; `shift` is [E,3] in CSR order
 * (NULL for open systems, then cell/batch are ignored); cell [B,3,3]; batch [N] original order. */
int hermnet_edge_geometry_fwd(const float* pos, const int* src_id, const int* tgt_id,
                              const float* shift, const float* cell, const int* batch,
                              int num_edges, float* edge /* [E,4] */, void* stream);

/* Backward of the above w.r.t. pos: gpos[N,3] (ORIGINAL order, overwritten) from
 * gD[E,4] (Cartesian gradient w.r.t. D, CSR order; .w ignored).  Deterministic: per-atom
 * segmented sums over `in_rowptr/in_edges` (edges whose target is the atom, sign -) and
 * `out_rowptr/out_edges` (edges whose source is the atom, sign +); both index CSR positions. */
int hermnet_edge_geometry_bwd(const float* gD, const int* in_rowptr, const int* in_edges,
                              const int* out_rowptr, const int* out_edges,
                              int num_nodes, float* gpos, void* stream);
/* The same sums with the out-edges read from the CSC order (hn_graph: csc_rowptr [T*N+1], csc_pos [E]) instead of
 * a separate out-adjacency: row a's out-edges are its T segments [csc_rowptr[t*N+a], csc_rowptr[t*N+a+1]).  Edges
 * to targets of unknown element are in no segment (they carry no message, so no gradient).  With this entry
 * point `hermnet_build_relations` may be called with out_rowptr = out_edges = NULL (one edge order fewer). */
int hermnet_edge_geometry_bwd_csc(const float* gD, const int* csr_rowptr, const int* csc_rowptr, const int* csc_pos,
                                  int num_rel, int num_nodes, float* gpos, void* stream);

/* ---- A3+A7(rbf_proj)+A8+A9+A10 and the residual of A6 ---------------------------------------
 * Replaces, for ALL relations of one HeteroVertexConv layer at once,
 *   PaiNNMessage.forward's rbf_proj + propagate (rmnet.py:55-73), RadialBasis.forward
 *   (rmnet.py:168-172), in_subgraph's edge slicing (utils.py:11-24) and the residual
 *   `x = (x + dx)/sqrt(2); vec = vec + dvec` (rmnet.py:24-26).
 * xh   [T,N,3H]  xh[t] = x_proj_t(LayerNorm_t(x)) for every row (sources of relation t)
 * xh_bias [T,3H] (NULL = none): added to every row of xh[t] on load, so that the x_proj GEMM can run
 *                without its broadcast bias (which would cost one more write + read of xh)
 * vec  [N,3,H]   (NULL => treated as zero: layer 0, hermnet.py:124)
 * x    [N,H]
 * wt   [T,R,3H]  rbf_proj.weight of relation t, transposed;  brbf [T,3H] its bias
 * edge [E,4]     from hermnet_edge_geometry_fwd (CSR order)
 * out: x1 [N,H], vec1 [N,3,H]; rows >= type_rowptr[T] are written as zero.
 * H must be a multiple of 64.
 * target_ranges (ABI v7; device, [T][2] int32, or NULL = every row; no reference counterpart -- atom shards, SURVEY 8(e)
 *   "run interior edges while the halo is in flight"): this launch computes only the target rows [lo_t, hi_t) of every
 *   relation t; rows outside are left untouched.  Two launches over complementary ranges give bit for bit what one
 *   launch gives: the host runs the targets whose sources are all owned rows while the halo exchange is in flight and
 *   the others behind it.  zero_unknown_rows != 0: this launch also writes the zero rows >= type_rowptr[T] (always
 *   done when target_ranges is NULL).  range_rows: the number of rows the ranges cover (host value; sizes the grid and the
 *   rows per workgroup; <= 0 = unknown, the whole row count is assumed -- correct, but a short launch then lasts as long
 *   as a full one). */
int hermnet_message_scatter_fwd(const hn_graph* g, const hn_rbf_desc* rbf, int hidden,
                                const float* xh, const float* xh_bias, const float* vec, const float* x,
                                const float* wt, const float* brbf, const float* edge,
                                float* x1, float* vec1, const int* target_ranges, int zero_unknown_rows, int range_rows,
                                void* stream);

/* Backward of hermnet_message_scatter_fwd for the force path (first order).
 * in : gx1 [N,H], gvec1 [N,3,H] (gradients w.r.t. x1, vec1) + the forward inputs
 * out: gxh [T,N,3H], gvec [N,3,H] (NULL allowed when vec was NULL), gx [N,H],
 *      gedge [H/64, E, 4]: per 64-channel column block, Cartesian gradient w.r.t. the edge
 *      vector D in CSR order (caller sums over the leading axis; buffer must be zero-filled).
 * split_t = 0: one workgroup walks all relations, gvec is [N,3,H];
 * split_t = 1: one relation per workgroup (3-D grid, better balance), gvec is [T,N,3,H] partial
 *              sums (slice 0 carries the residual's identity term), the caller sums over T.
 * source_ranges (ABI v7; device [num_ranges][2] int32 + the same values in source_ranges_host; num_ranges = 0: every
 *   row): this launch writes gxh / gvec / gx only for the SOURCE rows of the given disjoint ascending ranges (gedge: the
 *   edges leaving those rows).  Two launches over complementary ranges give bit for bit what one gives: the host runs
 *   the halo rows first, sends their gradients home and runs the rest while they travel.  Channel-per-lane form only
 *   (edge_table given): HN_ERR_BAD_ARG otherwise.
 * gx == NULL (ABI v8; channel-per-lane form, HVNet rows, no ranges): the finishing launch is left out -- gvec_partials
 *   [T,N,3,H] keeps the per-relation sums (required then whenever vec != NULL, also for T == 1) and neither gvec nor gx is
 *   written: the consumer forms gvec = sum_t gvec_partials[t] + gvec1 and gx = gx1 / sqrt2 itself
 *   (hermnet_node_update_bwd's `pending`), or needs neither (the first layer, whose inputs carry no gradient). */
int hermnet_message_scatter_bwd(const hn_graph* g, const hn_rbf_desc* rbf, int hidden,
                                const float* xh, const float* xh_bias, const float* vec,
                                const float* wt, const float* brbf, const float* edge,
                                const float* gx1, const float* gvec1,
                                float* gxh, float* gvec, float* gx, float* gedge, int split_t,
                                const float* edge_table, float* gvec_partials,
                                const int* source_ranges, const int* source_ranges_host, int num_ranges, void* stream);

/* Per-edge radial record, computed ONCE per step (geometry and radial basis are shared by every layer):
 * table [E + 1, 32] floats in CSC order -- record q belongs to CSC edge q, i.e. CSR edge csc_pos[q]; `edge` stays in
 * CSR order; record E repeats record E-1, the kernel requests one record ahead without a bounds check --, so the
 * backward, which walks the CSC segments, reads one sequential stream:
 *   [2m], [2m+1]  env(u) g_m  and  (env'(u) g_m + 2 coeff env(u) g_m (u - mu_{lo+m})) / rc   for the 12 taps m of the
 *                 edge's window, g_m = exp(coeff (u - mu_{lo+m})^2): contracted with the rbf_proj rows they give
 *                 rbfh - bias and d rbfh / d d
 *   [24] padded tile row of tap 0 (int bits) | [25] the same of CSC edge q+1 | [26,27] 0 | [28..30] rhat | [31] 1/d
 * (rmnet.py:156-193 evaluated exactly as the message kernels do in registers).  When `edge_table` is handed to
 * hermnet_message_scatter_bwd (NULL = not available) the backward runs in its channel-per-lane form, which reads
 * the record through the scalar path: a wave works on one edge, its taps sit in SGPRs (csrc/message_bwd_cl.hip).
 * That form runs one workgroup per (relation, column block, row chunk) and needs `gvec_partials`, a caller-owned
 * workspace [T, N, 3, H] (per-relation partial sums of gvec, added up in a fixed order by a second small launch;
 * not needed when T = 1 or vec is NULL); without it, or with split_t = 1, the 16-lanes-per-edge form runs. */
int hermnet_edge_radial_table(const hn_graph* g, const hn_rbf_desc* rbf, const float* edge, float* table, void* stream);

/* ---- node-level fused elementwise stages (A11/A12; the GEMMs between them are library calls) ----
 * Bias convention of these stages: the GEMM in front of a stage may run WITHOUT its bias (a GEMM with a
 * broadcast bias writes and re-reads its whole output once more); the stage then adds it on load.
 * `bias` (NULL = none) is [groups, cols] for a [rows, cols] operand, group = row / rows_per_bias
 * (rows_per_bias <= 0: a single bias row) -- one group per relation block of the uniform layout.
 *
 * ScaledSiLU (rmnet.py:110-117): a = silu(h + bias)/0.6; h, a contiguous [rows, cols], cols % 4 == 0. */
int hermnet_ssilu_fwd(const float* h, const float* bias, int rows_per_bias, float* a, long rows, int cols,
                      void* stream);
/* gh[n,t,c] = g[n*g_stride_n + t*g_stride_t + c] * d ssilu(h[n,t,c] + bias); h, gh contiguous [N,T,C];
 * bias for the [N, T*C] view of h. */
int hermnet_ssilu_bwd(const float* g, const float* h, const float* bias, int rows_per_bias, float* gh,
                      int N, int T, int C, long g_stride_n, long g_stride_t, void* stream);
/* LayerNorm without affine over the last axis (`x_layernorm`, rmnet.py:52; gamma/beta are folded into the
 * following Linear by the host): n = (x - mean) * rstd, rstd = 1/sqrt(var + eps) (biased variance).
 * x, n [rows, hidden]; mean, rstd [rows]; hidden % 4 == 0, hidden <= 1024.  `hidden_real` (0 = hidden): the
 * statistics run over the first hidden_real channels only and the remaining outputs are zero -- rows of a model
 * whose hidden_channels is not a multiple of 64 are zero-padded to the next multiple for the message kernels' column
 * blocks (the reference accepts any width, hermnet.py:84-88). */
int hermnet_layernorm_fwd(const float* x, float* n, float* mean, float* rstd, int rows, int hidden, int hidden_real,
                          float eps, void* stream);
/* gx = d(n)/d(x)^T g + add  (add [rows, hidden] may be NULL; gx may alias add). */
int hermnet_layernorm_bwd(const float* g, const float* x, const float* mean, const float* rstd, const float* add,
                          float* gx, int rows, int hidden, int hidden_real, void* stream);
/* PaiNNUpdate middle (rmnet.py:95-100): vp [rows,3,2H] = vec_proj(vec1) ->
 * vdot [rows,H] = sum_d v1 v2 / sqrt(H);  xin [rows,2H] = [x1 | sqrt(sum_d v2^2 + 1e-8)]. */
int hermnet_update_mid(const float* vp, const float* x1, float* vdot, float* xin, int rows, int hidden,
                       void* stream);
/* PaiNNUpdate tail + residual (rmnet.py:101-107,29-31) + zero rows (hermnet.py:51,56-57):
 * x_out = x1 + (q1 + q2 vdot)/sqrt2, vec_out[d] = vec1[d] + q3 v1[d]; rows >= num_known or with
 * row_mask[r] == 0 (row_mask may be NULL) are written as zero without reading the other inputs. */
int hermnet_update_out(const float* q, const float* qbias /* [groups,3H] or NULL */, int rows_per_bias,
                       const float* vdot, const float* vp, const float* x1,
                       const float* vec1, const float* row_mask, float* x_out, float* vec_out,
                       int num_nodes, int num_known, int hidden, void* stream);
/* Backward of hermnet_update_out: gq [N,3H], gvdot [N,H], the v1 half of gvp [N,3,2H], and the
 * identity parts gx1 [N,H] / gvec1 [N,3,H] (masked copies of the incoming gradients). */
int hermnet_update_out_bwd(const float* gx_out, const float* gvec_out, const float* q,
                           const float* qbias, int rows_per_bias, const float* vdot,
                           const float* vp, const float* row_mask, float* gq, float* gvdot, float* gvp,
                           float* gx1, float* gvec1, int num_nodes, int num_known, int hidden, void* stream);
/* Backward of hermnet_update_mid: completes gvp (both halves) from gvdot and gxin [rows,2H] and
 * accumulates gxin[:, :H] into gx1. */
int hermnet_update_mid_bwd(const float* gvdot, const float* gxin, const float* vp, const float* xin,
                           float* gvp, float* gx1, int rows, int hidden, void* stream);

/* Energy read-out head (`out_energy`, hermnet.py:113-117,129) behind its first Linear (a library GEMM):
 * e[n] = (sum_c ScaledSiLU(h[n,c]) * w[c] + b[0]) * row_mask[n];   h [rows, cols], w [cols] = out_energy[2].weight,
 * b [1] its bias (device pointer, may be NULL), row_mask [rows] (may be NULL: all ones) zeroes padding rows. */
int hermnet_energy_head_fwd(const float* h, const float* w, const float* b, const float* row_mask, float* e,
                            int rows, int cols, void* stream);
/* gh[n,c] = ge[n] * row_mask[n] * w[c] * d ScaledSiLU(h[n,c]). */
int hermnet_energy_head_bwd(const float* ge, const float* h, const float* w, const float* row_mask, float* gh,
                            int rows, int cols, void* stream);

/* The whole read-out in one launch each way (no library GEMM): forward  h = x W0^T + b0 [rows, cols] (saved),
 * e = (w2 . ScaledSiLU(h) + b2) row_mask;  backward  gx = (ge row_mask w2 ScaledSiLU'(h)) W0  [rows, hidden].
 * w0t = out_energy[0].weight^T [hidden, cols], w0 = out_energy[0].weight [cols, hidden].  Supported: the output width
 * of each product (cols forward, hidden backward) in {64, 128, 256} and hidden * cols * 4 <= 64 KiB (the weight matrix
 * is staged in LDS); otherwise HN_ERR_BAD_ARG and the caller uses hermnet_energy_head_fwd / _bwd around its own GEMM. */
int hermnet_energy_head_fused_fwd(const float* x, const float* w0t, const float* b0, const float* w2, const float* b2,
                                  const float* row_mask, float* h, float* e, int rows, int hidden, int cols,
                                  void* stream);
int hermnet_energy_head_fused_bwd(const float* ge, const float* h, const float* w0, const float* w2,
                                  const float* row_mask, float* gx, int rows, int hidden, int cols, void* stream);

/* ABI v9: the same read-out with its H -> C product on the fp32 matrix pipe (csrc/node_chain16.hip; hidden 128, cols 64: the
 * width of BASELINE configs[1]): w0_frag16 = frag16(out_energy[0].weight [cols, hidden]), w0t_frag16 = frag16 of its transpose
 * (frag16: hermnet_node_update_fwd, "16-row form").  Same results to rounding (another summation order); ~4 us each way at
 * 10k rows where the staged-weight form above takes 24 / 16 us. */
int hermnet_energy_head16_supported(int hidden, int cols);
int hermnet_energy_head16_fwd(const float* x, const float* w0_frag16, const float* b0, const float* w2, const float* b2,
                              const float* row_mask, float* h, float* e, int rows, int hidden, int cols, void* stream);
int hermnet_energy_head16_bwd(const float* ge, const float* h, const float* w0t_frag16, const float* w2,
                              const float* row_mask, float* gx, int rows, int hidden, int cols, void* stream);

/* ---- A7 (node MLP) + A11 + A12 as chain kernels on the matrix pipe (csrc/node_chain.hip) ---------------------------------
 * One launch per chain instead of library GEMMs joined by elementwise launches; hidden activations never reach HBM.
 * hidden must be a multiple of 64 up to 512 (hermnet_node_chain_supported; other widths: the stage-wise entry points above
 * around the caller's GEMMs).  fp32 in, fp32 out, fp32 accumulation; since ABI v11 every product runs on the BF16 matrix pipe
 * as a three-way split of both operands (x = x0 + x1 + x2 exactly, bf16 planes, round to nearest even of what the planes
 * before leave; the six largest of the nine partial products, the dropped ones <= 2^-24 of a product each: fp32-equivalent at
 * 6/16 of the fp32 MFMA's pipe time).  Weights arrive split and in FRAGMENT ORDER: for an nn.Linear weight W [out, in] (or
 * its transpose for the backward products), per 32-row block cb and 16-deep k-group Q the three planes, smallest first, as
 * v_mfma_f32_32x32x16_bf16 operands -- lane l: 8 bf16 = 16 bytes --
 *     frag(W)[((cb * in/16 + Q) * 3 + s) * 64 + l] = W_(2-s)[32 cb + (l & 31)][16 Q + 8 (l >> 5) .. +7]
 * (out * in * 3 / 2 floats, 6 bytes per weight; hermnet_amd/nodeops.py: weight_fragments), so that a wave loads one coalesced
 * KiB per step straight into the operand.  `[T]` = stacked over the relations.
 *
 * hermnet_node_pre_fwd   (rmnet.py:52): for every relation t and source row,
 *     hb[t] = LayerNorm(x) W1_t^T + b1_t   (LayerNorm without affine: folded into W1 / b1 by the host; statistics over
 *                                           the first hidden_real channels, 0 = all),   saved for the backward
 *     xh[t] = ScaledSiLU(hb[t]) W2_t^T + b2_t      [T, num_src, 3H]   (bias INCLUDED: pass xh_bias = NULL to the
 *                                                                       message kernels)
 *     mean, rstd [num_src]: the LayerNorm statistics.
 *     src_ranges (device, [T][4] int32, or NULL = every row): relation t only ever gathers source rows [r0, r1) and
 *     [r2, r3) (HTNet: relation (c; p, q) gathers atoms of elements p and q); 64-row tiles outside both ranges are
 *     skipped -- their rows of hb / xh are left unwritten, their statistics must be pre-set by the caller (zeros), and
 *     the backward contributes zero for them.
 * hermnet_node_pre_bwd: gx = LayerNorm'(x)^T sum_t ((gxh[t] W2_t) * ScaledSiLU'(hb[t])) W1_t + add
 *     (w2t_frag = frag(W2_t^T [H, 3H]), w1t_frag = frag(W1_t^T [H, H]); gn_parts [T, num_src, H] is workspace; add may
 *     be NULL; gx may alias add).  gx == NULL (ABI v8): only the per-relation partial sums gn_parts are produced -- the
 *     LayerNorm backward over their sum is left to the consumer (hn_pending_grads below); x / mean / rstd / add unused.
 * row_windows (device, [num_windows][2] int32 row ranges) + window_mode (ABI v6; atom shards, no reference counterpart):
 *     0 = every row tile (row_windows may be NULL); 1 = only the tiles (hermnet_node_chain_tile_rows) that touch a window;
 *     2 = only the others.  Two calls with modes 2 and 1 on the same buffers compute what one call with mode 0 does:
 *     the host puts the halo exchange between them (forward: 2, wait + unpack, 1; backward: 1, send, 2). */
int hermnet_node_chain_supported(int hidden);   /* hidden % 64 == 0, 64 <= hidden <= 512 (the reference's default is 512) */
/* Rows per tile of the pre kernels (update = 0) or the update kernels (update != 0) at this width: 64 / 32; 0 when the
 * width is not supported.  The tiles of window modes 1 / 2 are cut at multiples of it. */
int hermnet_node_chain_tile_rows(int hidden, int update);
int hermnet_node_pre_fwd(const float* x, const float* w1_frag, const float* b1, const float* w2_frag, const float* b2,
                         float* hb, float* xh, float* mean, float* rstd, const int* src_ranges, int num_src,
                         int num_rel, int hidden, int hidden_real, float eps, const int* row_windows, int num_windows,
                         int window_mode, void* stream);
int hermnet_node_pre_bwd(const float* gxh, const float* hb, const float* w2t_frag, const float* w1t_frag,
                         float* gn_parts, const float* x, const float* mean, const float* rstd, const float* add,
                         float* gx, const int* src_ranges, int num_src, int num_rel, int hidden, int hidden_real,
                         const int* row_windows, int num_windows, int window_mode, void* stream);
/* hermnet_node_update_fwd (rmnet.py:94-107, 29-31; hermnet.py:51,56-61), target rows in relation order:
 *     vp = vec1 Wv^T  [N,3,2H] = (v1 | v2), saved;   vdot = sum_d v1 v2 / sqrt(H);   n = sqrt(sum_d v2^2 + 1e-8) -> nrm
 *                                                                                    [N,H], saved
 *     h2b = [x1 | n] Wx0^T + bx0  [N,H], saved;      (p | q | r) = ScaledSiLU(h2b) Wx2^T + bx2;   q23 = (q | r) [N,2H], saved
 *     x_out = x1 + (p + q vdot)/sqrt2,  vec_out[d] = vec1[d] + r v1[d];   rows with row_active == 0 (row_active may be
 *     NULL: all active) and rows >= type_rowptr[T] are written as zero.
 * type_rowptr [T+1] on the device and the same values on the host (`type_rowptr_host`: the launch geometry).
 * hermnet_node_update_bwd: (gx_out, gvec_out) -> (gx1, gvec1), the gradients w.r.t. x1 / vec1 (parameters are
 *     constants); wx2t_frag = frag(Wx2^T [H,3H]), wx0t_frag = frag(Wx0^T [2H,H]), wvt_frag = frag(Wv^T [H,2H]).
 * tile_rows (ABI v7): 0 = the width's default row tile (32; 64 at hidden 64); 16 = the 16-row form (hidden 128 only:
 *     v_mfma_f32_16x16x32_bf16, csrc/node_chain16.hip) -- every *_frag argument must then be the frag16 copy of the weight,
 *     frag16(W)[((b * K/32 + Q) * 3 + s) * 64 + l] = 8 bf16 W_(2-s)[16 b + (l & 15)][32 Q + 8 (l >> 4) .. +7]
 *     (nodeops.weight_fragments16).
 *     hermnet_node_update_tile_rows says which form shortens the launch for a row layout (small grids: 32-row tiles
 *     leave most CUs with one workgroup while a few get two; 16-row tiles cost twice the weight bytes from L2).
 * pending (ABI v8; may be NULL): the incoming gradients of this layer's outputs have not been formed yet -- they still sit
 *     in the partial sums of the layer above, which skipped its two small finishing launches (hermnet_message_scatter_bwd
 *     and hermnet_node_pre_bwd, each called with gx == NULL).  The update backward then forms them for the rows of each of
 *     its tiles first, with the same operations in the same order (bit-identical to the separate launches),
 *         gx_out[r]   = LayerNorm'(x[r])^T (sum_p gn_parts[p][r]) + gx1[r] / sqrt2
 *         gvec_out[r] = sum_p gvec_parts[p][r] + gvec1[r]             (the "+" terms on rows < type_rowptr[T] only),
 *     WRITES them to gx_out / gvec_out (buffers of the caller; the `const` is nominal in this mode) and goes on as usual.
 *     Needs num_src == num_nodes (HVNet rows) in the layer above.  Saves two launches per layer boundary. */
typedef struct hn_pending_grads {
  const float* gn_parts;    /* [num_parts][num_nodes][hidden]     from hermnet_node_pre_bwd(gx = NULL) */
  const float* gvec_parts;  /* [num_parts][num_nodes][3][hidden]  from hermnet_message_scatter_bwd(gx = NULL) */
  const float* x;           /* [num_nodes][hidden]  the layer above's input, mean / rstd [num_nodes] its LayerNorm statistics */
  const float* mean;
  const float* rstd;
  const float* gx1;         /* [num_nodes][hidden], [num_nodes][3][hidden]: the layer above's update-backward results */
  const float* gvec1;
  int num_parts;            /* the relations T of the layer above */
  int hidden_real;          /* as in hermnet_node_pre_bwd (0 = hidden) */
  /* ABI v9, fused form (tile_rows == 16 only; gn_parts is then unused and may be NULL): the layer above did not run
   * hermnet_node_pre_bwd at all -- the update backward runs that chain for the rows of each of its tiles first,
   *     gn_t = ((gxh[t] W2_t) * ScaledSiLU'(hb[t])) W1_t,   sum_t in registers in the order (gn_0 + gn_1) + gn_2 ...,
   * then the LayerNorm backward on the tile: no [T, num_nodes, hidden] partial sums in memory, one launch less per layer
   * boundary.  Bit-identical to hermnet_node_pre_bwd16 followed by the `gn_parts` form. */
  const float* gxh;         /* [num_parts][num_nodes][3 hidden]  from hermnet_message_scatter_bwd (NULL: the gn_parts form) */
  const float* hb;          /* [num_parts][num_nodes][hidden]    saved by the layer above's node projection */
  const float* w2t_frag16;  /* [num_parts] frag16(W2_t^T [hidden, 3 hidden]), frag16(W1_t^T [hidden, hidden]) of the layer above */
  const float* w1t_frag16;
} hn_pending_grads;
int hermnet_node_update_tile_rows(const int* type_rowptr_host, int num_nodes, int num_rel, int hidden);
int hermnet_node_update_fwd(const float* x1, const float* vec1, const float* wv_frag, const float* wx0_frag,
                            const float* bx0, const float* wx2_frag, const float* bx2, const float* row_active,
                            const int* type_rowptr, const int* type_rowptr_host, float* vp, float* h2b, float* q23,
                            float* nrm, float* x_out, float* vec_out, int num_nodes, int num_rel, int hidden,
                            int tile_rows, void* stream);
int hermnet_node_update_bwd(const float* gx_out, const float* gvec_out, const float* vp, const float* h2b,
                            const float* q23, const float* nrm, const float* wx2t_frag, const float* wx0t_frag,
                            const float* wvt_frag,
                            const float* row_active, const int* type_rowptr, const int* type_rowptr_host, float* gx1,
                            float* gvec1, int num_nodes, int num_rel, int hidden, int tile_rows,
                            const hn_pending_grads* pending, void* stream);

/* ---- ABI v9: one node launch per layer boundary, each way (csrc/node_chain16.hip; hidden 128, 16-row tiles).
 * A tile's PaiNNUpdate of layer l (rmnet.py:94-107, 29-31) and the node projection of layer l + 1 on the rows it has just
 * produced (rmnet.py:52 for every relation of the NEXT layer) are row-local: hermnet_node_update_pre_fwd runs both in one
 * launch -- the arguments of hermnet_node_update_fwd (16-row form: frag16 weights) followed by those of the next layer's
 * projection (w1_frag16 = frag16 of [next_num_rel] W1 with the LayerNorm affine folded in, b1, w2_frag16, b2; outputs hb, xh,
 * mean, rstd as hermnet_node_pre_fwd writes them, for all num_nodes rows).  The backward mirror is hermnet_node_update_bwd with
 * pending->gxh set.  hermnet_node_pre_fwd16 / hermnet_node_pre_bwd16 run the same projection phases as kernels of their own
 * (16-row tiles, frag16 weights; pre_bwd16 writes the per-relation partial sums gn_parts only): update_fwd(tile_rows = 16) +
 * pre_fwd16 is bit-identical to update_pre_fwd, pre_bwd16 + update_bwd(pending->gn_parts) to update_bwd(pending->gxh). */
int hermnet_node_fused_supported(int hidden);
int hermnet_node_update_pre_fwd(const float* x1, const float* vec1, const float* wv_frag16, const float* wx0_frag16,
                                const float* bx0, const float* wx2_frag16, const float* bx2, const float* row_active,
                                const int* type_rowptr, const int* type_rowptr_host, float* vp, float* h2b, float* q23,
                                float* nrm, float* x_out, float* vec_out, int num_nodes, int num_rel, int hidden,
                                const float* w1_frag16, const float* b1, const float* w2_frag16, const float* b2, float* hb,
                                float* xh, float* mean, float* rstd, int next_num_rel, int hidden_real, float eps,
                                void* stream);
int hermnet_node_pre_fwd16(const float* x, const float* w1_frag16, const float* b1, const float* w2_frag16, const float* b2,
                           float* hb, float* xh, float* mean, float* rstd, int num_src, int num_rel, int hidden,
                           int hidden_real, float eps, void* stream);
int hermnet_node_pre_bwd16(const float* gxh, const float* hb, const float* w2t_frag16, const float* w1t_frag16,
                           float* gn_parts, int num_src, int num_rel, int hidden, void* stream);

/* HTNet (hermnet.py:155-157 is a stub; DESIGN.md "HTNet"): a centre atom's P pair relations are averaged.  Target rows
 * are [num_elem][pairs][block] blocks of `block` rows ("virtual" rows, one per atom and pair relation):
 *   mode 0:  x_out [rows_out,H], vec_out [rows_out,3,H]: row c*block + i = scale * sum_k in[(c*pairs + k)*block + i]
 *            (scale_x for x, scale_vec for vec; the mean: 1/pairs), rows >= num_elem*block are written as zero (atoms
 *            of elements outside `elems`, hermnet.py:51)
 *   mode 1:  x_out / vec_out [num_elem*pairs*block, ...] = the gradient w.r.t. the virtual rows: scale * in[c*block + i]
 *            (x_in / vec_in then hold the [rows_out, ...] gradient)
 *   mode 2:  as mode 0 for the rows < num_elem*block, ACCUMULATED into x_out / vec_out (the residual's gradient summed
 *            over a centre's virtual rows: scale_x = 1/sqrt(2), scale_vec = 1, rmnet.py:24-26).  row_ranges (ABI v7;
 *            device [num_ranges][2] int32, 0 = every row; mode 2 only): only the output rows of these ranges (atom
 *            shards: the message backward runs in two launches over complementary source-row ranges). */
int hermnet_pair_mean(int mode, const float* x_in, const float* vec_in, float* x_out, float* vec_out, int num_elem,
                      int pairs, int block, int rows_out, int hidden, float scale_x, float scale_vec,
                      const int* row_ranges, int num_ranges, void* stream);

/* ---- the training step's per-edge message algebra (rmnet.py:58-66 inside example/dist_train.py:86-99, where the
 * forces are differentiated w.r.t. the parameters: every op needs a second derivative).  ABI v6; csrc/train_kernels.hip.
 *   X [E,3H] = x_proj(LayerNorm(x)) of the edge's source (Xs | Xa | Xb),  R [E,3H] = rbf_proj(rbf(d)) (Rs | Ra | Rb, the
 *   constant factors 1/sqrt(3H), 1/sqrt(H) already on the a / b parts),  V [E,3,H] = vec of the source or NULL (layer 0),
 *   U [E,3] = rhat.   hidden must be a multiple of 4.
 *   fwd :  S = Xs Rs [E,H];  M_d = (Xb Rb) U_d + V_d (Xa Ra) [E,3,H]        (their row sums are dx, dvec)
 *   bwd :  cotangents (GS, GM) -> gX, gR [E,3H], gV [E,3,H] (NULL with V), gU [E,3]
 *   bwd2:  cotangents (cX, cR, cV, cU; each may be NULL = zero) of bwd's outputs -> dGS [E,H], dGM [E,3,H], dX, dR
 *          [E,3H], dV [E,3,H] (NULL with V), dU [E,3]   -- the backward of the backward; the map is multilinear, so all
 *          three are per-edge products and channel sums (deterministic, no atomics).
 *   x_rows / v_rows / t_rows [E] int64 (each may be NULL = row e): X and cX are read at row x_rows[e] of their arrays
 *   (x_j = xh[(relation, source)] without a gathered copy), V and cV at v_rows[e] (vec[source]), GS and GM at t_rows[e]
 *   (the cotangents of dx, dvec of the edge's target); R and cR at r_rows[e], where gR / dR are written as well (R kept
 *   in another edge order; rows no edge points to are left untouched); every other OUTPUT is per edge -- the caller
 *   sums gX / gV / dGS / ... into rows with hermnet_segment_sum. */
int hermnet_edge_message_fwd(const float* X, const float* R, const float* V, const float* U, long num_edges, int hidden,
                             const long* x_rows, const long* v_rows, const long* r_rows, float* S, float* M,
                             void* stream);
int hermnet_edge_message_bwd(const float* GS, const float* GM, const float* X, const float* R, const float* V,
                             const float* U, long num_edges, int hidden, const long* x_rows, const long* v_rows,
                             const long* t_rows, const long* r_rows, float* gX, float* gR, float* gV, float* gU,
                             void* stream);
int hermnet_edge_message_bwd2(const float* cX, const float* cR, const float* cV, const float* cU, const float* GS,
                              const float* GM, const float* X, const float* R, const float* V, const float* U,
                              long num_edges, int hidden, const long* x_rows, const long* v_rows, const long* t_rows,
                              const long* r_rows, float* dGS, float* dGM, float* dX, float* dR, float* dV, float* dU,
                              void* stream);

/* Segmented row sum with an optional gather (the adjoint of a row gather; training path, ABI v6):
 * out[r] = sum over q in [rowptr[r], rowptr[r+1]) of x[perm ? perm[q] : q], rows of `width` floats (a multiple of 4),
 * members added in list order (deterministic).  perm [rowptr[num_rows]] int64 or NULL; rowptr [num_rows + 1] int64. */
/* The same three kernels with the ROW SUMS inside (ABI v8): one lane group per output row of a grouping of the edges --
 * group g holds the edges group_edges[q] (q itself when group_edges is NULL), q in [group_rowptr[g], group_rowptr[g+1]) --
 * and whatever is summed over that grouping never reaches HBM per edge.  fwd: groups = target rows, dx [G,H] and dv [G,3,H]
 * are the sums of S and M.  bwd / bwd2: groups = the (relation, source) rows of xh: gX_rows / dX_rows [G,3H] are the sums,
 * gV_rows / dV_rows [G,3,H] the per-group partial sums (the caller adds a source's T groups); gR, gU, dGS, dGM, dR, dU stay
 * per edge as in the kernels above.  Edges in no group are not visited. */
int hermnet_edge_message_fwd_rows(const float* X, const float* R, const float* V, const float* U, long num_edges, int hidden,
                                  const long* x_rows, const long* v_rows, const long* r_rows, const long* group_rowptr,
                                  const long* group_edges, long num_groups, float* dx, float* dv, void* stream);
int hermnet_edge_message_bwd_rows(const float* GS, const float* GM, const float* X, const float* R, const float* V,
                                  const float* U, long num_edges, int hidden, const long* x_rows, const long* v_rows,
                                  const long* t_rows, const long* r_rows, const long* group_rowptr, const long* group_edges,
                                  long num_groups, float* gX_rows, float* gR, float* gV_rows, float* gU, void* stream);
int hermnet_edge_message_bwd2_rows(const float* cX, const float* cR, const float* cV, const float* cU, const float* GS,
                                   const float* GM, const float* X, const float* R, const float* V, const float* U,
                                   long num_edges, int hidden, const long* x_rows, const long* v_rows, const long* t_rows,
                                   const long* r_rows, const long* group_rowptr, const long* group_edges, long num_groups,
                                   float* dGS, float* dGM, float* dX_rows, float* dR, float* dV_rows, float* dU,
                                   void* stream);
int hermnet_segment_sum(const float* x, const long* perm, const long* rowptr, long num_rows, int width, float* out,
                        void* stream);

/* Node-level stages of the TRAINING step (train() mode; no reference counterpart beyond the formulas: rmnet.py:52, 94-107,
 * 110-117, differentiated twice by autograd there).  ONE entry point, twelve kernels (csrc/train_node_kernels.hip); `in` / `out`
 * are host arrays of device pointers, rows x hidden fp32 row-major unless noted, hidden % 4 == 0, hidden <= 1024.
 * Stages:  SILU y = x sigmoid(x);  LN y = LayerNorm(x) without affine, eps = c0;
 *          MID (vp [R,3,2H] = (v1 | v2), xt) -> vdot = c0 sum_d v1 v2, xin [R,2H] = (xt | sqrt(sum_d v2^2 + c1));
 *          OUT (q [R,3H] = (q1|q2|q3), vdot, vp, xt, vt [R,3,H], m [R] or NULL) -> xo = m (xt + (q1 + q2 vdot) c0), vo_d = m (vt_d + q3 v1_d).
 * `bwd` maps the cotangents of a stage's outputs to those of its inputs, `bwd2` does the same for `bwd` itself.
 *  op  kernel     in                                                                   out
 *   1  silu_bwd   gy, x                 (rows = number of float4 elements)              gx
 *   2  silu_bwd2  u (cot. of gx), gy, x                                                 c_gy, c_x
 *   3  mid_fwd    vp, xt                                                                vdot, xin
 *   4  mid_bwd    g_vdot, g_xin, vp                                                     g_vp, g_xt
 *   5  mid_bwd2   u_vp*, u_xt*, g_vdot, g_xin, vp                                       c_gvdot, c_gxin, c_vp
 *   6  out_fwd    q, vdot, vp, xt, vt, m*                                               xo, vo
 *   7  out_bwd    gx, gv, q, vdot, vp, m*                                               g_q, g_vdot, g_vp, g_xt*, g_vt*
 *   8  out_bwd2   c_gq*, c_gvdot*, c_gvp*, c_gxt*, c_gvt*, gx, gv, q, vdot, vp, m*      d_gx, d_gv, d_q, d_vdot, d_vp
 *   9  ln_bwd     gy, x                                                                 gx
 *  10  ln_bwd2    v (cot. of gx), gy, x                                                 c_gy, c_x
 *  11  res_fwd    x, dx, v* [R,3,H], dv, m*    (rmnet.py:24-26: x1 = m c0 (x + dx), v1 = m (v + dv))   x1, v1
 *  12  mask_scale g1*, gv1*, m*   (its backward, for both summands, and its own backward)         m c0 g1, m gv1
 * (* = may be NULL: a zero cotangent / no mask / an output nobody reads).  num_in / num_out must be the counts above. */
int hermnet_train_node_op(int op, const float* const* in, int num_in, float* const* out, int num_out, long rows, int hidden,
                          float c0, float c1, void* stream);

/* Halo exchange packing for atom-sharded runs (one process per GPU; the exchange itself is an RCCL all-to-all made
 * by the host code, hermnet_amd/sharding.py).  A packed row = [ x (H) | vec (3H) ]; idx [n] (int64) holds rows.
 *   mode 0  buf[k] = rows[idx[k]]                         pack what the neighbours need
 *   mode 1  buf[k] = rows[idx[k]]; rows[idx[k]] = 0       pack the gradients of my halo rows and clear them
 *   mode 2  rows[idx[k]] = buf[k]                         unpack received halo rows (idx unique)
 * Any other mode: HN_ERR_BAD_ARG (the float-atomic accumulate of ABI <= 6 is gone: hermnet_halo_accumulate). */
int hermnet_halo_rows(int mode, float* x, float* vec, const long* idx, int n, int hidden, float* buf, void* stream);

/* Accumulating returned gradients at their owner: the returned packed rows are summed per owner row in
 * a FIXED order, rows[seg_rows[u]] += sum_{q in [seg_ptr[u], seg_ptr[u+1])} buf[seg_pos[q]]  (seg_rows unique, int64
 * lists prepared once per exchange plan by hermnet_amd/sharding.py) -- no atomics, bit-reproducible. */
int hermnet_halo_accumulate(float* x, float* vec, const long* seg_rows, const long* seg_ptr, const long* seg_pos,
                            int num_rows, int hidden, const float* buf, void* stream);

/* Halo exchange of PROJECTED rows (ABI v12; the default form of the per-layer exchange where the node chain kernels run:
 * hermnet_amd/layer.py).  What a neighbour needs of a halo atom is xh[t] = x_proj_t(LayerNorm(x)) for every relation
 * (rmnet.py:52: the source rows the message gathers, rmnet.py:58) and vec: T + 1 blocks of `width` = 3H floats.  The owner
 * sends those -- 12H floats per atom instead of 4H -- and the receiver runs NO node projection on halo rows, forward or
 * backward (no windowed second launch of the chain kernels, no finishing launches in front of the gradient exchange).
 * A packed row = [ a[0][r] | ... | a[S-1][r] | sum_{s < num_sum} b[s][r] ], a[j] = a + j * a_seg_stride, b[s] = b + s *
 * b_slice_stride, rows `width` floats apart.  Forward: a = xh [T, N, 3H], b = vec (num_sum 1).  Backward: a = gxh, b = the
 * per-relation partial sums of gvec [T, N, 3, H] (num_sum T: summed, ascending relation, while they are packed).
 *   mode 0 pack;  mode 1 pack, then clear every source;  mode 2 unpack (a[j][r] = block j, b[0][r] = block S; idx unique) */
int hermnet_halo_proj_rows(int mode, float* a, long a_seg_stride, int num_seg, float* b, long b_slice_stride, int num_sum,
                           const long* idx, int n, int width, float* buf, void* stream);
/* ... and the owner's side of the return path: a[j][seg_rows[u]] += sum of block j of the returned rows of segment u,
 * b[seg_rows[u]] += the sum of their last blocks, in list order (lists as for hermnet_halo_accumulate): no atomics. */
int hermnet_halo_proj_accumulate(float* a, long a_seg_stride, int num_seg, float* b, const long* seg_rows, const long* seg_ptr,
                                 const long* seg_pos, int num_rows, int width, const float* buf, void* stream);

/* float4 stream copy dst[i] = src[i] (16-byte aligned pointers, num_floats % 4 == 0): not part of the path -- the yardstick
 * for SURVEY.md 8(d)'s "measured copy bandwidth on the box" (bench.py: roofline.measured_copy_GBps; the method of
 * MI355X_MICROARCH.md's 6.29 TB/s figure).  `workgroups` <= 0: 8 per CU. */
int hermnet_stream_copy(const float* src, float* dst, size_t num_floats, int workgroups, void* stream);

/* Parameter guard (ABI v9).  The host side caches kernel-ready copies of the module's parameters and rebuilds them when a
 * parameter's identity / version / address changes; a write through `.data` changes none of these (the reference has no such
 * cache: /root/reference/HermNet/hermnet.py:118-131 reads nn.Parameters directly on every call).  `tensor_ptrs` [n] device array
 * of device pointers to CHUNKS of the parameter tensors (at most 4096 32-bit words each: one workgroup per chunk),
 * `word_counts` [n] their sizes in words, `fingerprints` [n] uint32.
 * check == 0: record a position-weighted wrapping sum of every tensor's words.  check != 0: compare; on any difference
 * flag[0] = 1 and, if `poison` is given, poison[0] = NaN (the caller passes a cached value every result depends on, so a step on
 * stale copies yields NaN instead of the old numbers).  One launch, no host read. */
int hermnet_param_guard(const void* const* tensor_ptrs, const long* word_counts, int num_tensors, unsigned* fingerprints,
                        int check, float* poison, int* flag, void* stream);

/* Per-step flags of an atom-sharded step in two launches (clear + mark; ABI v10; hermnet_amd/sharding.py: slab_data,
 * plan_moved -- sixteen elementwise / index launches before).  The reference skips a relation when no atom of its element receives an edge anywhere
 * in the structure (/root/reference/HermNet/hermnet.py:56-57): every rank marks the (target element, source element) pairs its
 * own list joins and the ranks reduce the table.  `edge_index` [2, columns] (rows: source, target; NULL edges of a padded list
 * = -1), `atomic_number` [num_atoms] of the LOCAL atoms, `total` [2] = (pairs found, flags) of a padded list or null,
 * `capacity` its columns.  `has_in` [128 * 128 + 3] int32 is zeroed here, then: [zt * 128 + zs] = 1 per joined pair (elements
 * clamped to 127), [128 * 128] = 1 if the list holds NULL edges, [128 * 128 + 1] = 1 if the padded list is incomplete
 * (flags != 0 or more pairs than columns), [128 * 128 + 2] = 1 if an atom of `pos` [num_pos, 3] is further than
 * sqrt(max_dist2) from its position in `pos_ref` (num_pos = 0: not asked).  No host read, no memset node. */
int hermnet_shard_step_flags(const long* edge_index, long columns, const long* atomic_number, int num_atoms, const long* total,
                             long capacity, int* has_in, const float* pos, const float* pos_ref, long num_pos, float max_dist2,
                             void* stream);

/* rbf_proj of the TRAINING path on the bucketed basis (ABI v13; hermnet_amd/trainops.py: BucketedBasis, BandP / BandQ / BandS;
 * /root/reference/HermNet/rmnet.py:55 on 32-centre windows, differentiated twice by /root/reference/example/dist_train.py:86-99).
 * Edges sorted by (relation, distance bucket) form `num_chunks` chunks of `rows_per_chunk` rows (a multiple of 32; _grad_b: of
 * 16); a chunk has ONE 32 x width weight window (width = 3 * hidden: a multiple of 32, at most 480).  Exact fp32 products on the
 * fp32 matrix pipe, fixed summation order.
 *   hermnet_band_product          out[c] = a[c] b[c] + bias[c]       a [nc,C,32]  b [nc,32,width]  bias [nc,width] | NULL  out [nc,C,width]
 *   hermnet_band_product_grad_a   ga[c]  = (g1[c] + g2[c]) b[c]^T     g1, g2 [nc,C,width] (g2 NULL = absent)               ga  [nc,C,32]
 *   hermnet_band_product_grad_b   gb[c]  = a[c]^T (g1[c] + g2[c]),  gbias[c] = column sums of g1[c] + g2[c] (gbias NULL = not asked)
 *   hermnet_band_product_grads    ga, gb and gbias of the two above from ONE pass over g1 (+ g2) (a, b, g1, ga, gb required)
 * g1 + g2: the radial array has two consumers in the autograd graph (the message algebra and its backward); their gradients are
 * added while they are read.  hermnet_band_product_supported: 1 when (rows_per_chunk, width) is a shape these kernels take. */
int hermnet_band_product_supported(int rows_per_chunk, int width);
int hermnet_band_product(const float* a, const float* b, const float* bias, long num_chunks, int rows_per_chunk, int width,
                         float* out, void* stream);
int hermnet_band_product_grad_a(const float* g1, const float* g2, const float* b, long num_chunks, int rows_per_chunk, int width,
                                float* ga, void* stream);
int hermnet_band_product_grad_b(const float* a, const float* g1, const float* g2, long num_chunks, int rows_per_chunk, int width,
                                float* gb, float* gbias, void* stream);
int hermnet_band_product_grads(const float* a, const float* b, const float* g1, const float* g2, long num_chunks,
                               int rows_per_chunk, int width, float* ga, float* gb, float* gbias, void* stream);
/* Column sums of x [num_slices, rows, width] over the rows (ABI v13; the bias gradients of the training path's node linears,
 * /root/reference/HermNet/rmnet.py:52,94-100): partial [num_slices, ceil(rows / rows_per_block), width] holds one sum per block of
 * rows; the caller adds the partials (fixed order).  width a multiple of 4, at most 1024. */
int hermnet_col_sum(const float* x, long num_slices, long rows, int width, int rows_per_block, float* partial, void* stream);

/* Edge unit vectors and their two derivatives for the training path (ABI v13; /root/reference/HermNet/hermnet.py:144-152 with the
 * distance floor of :146-147): D [E,3] -> U = D / d, d = max(|D|, 1e-6).
 *   order 0: out0 = U [E,3], out1 = d [E];   order 1: out0 = gD [E,3] from the cotangents gU [E,3]*, gd [E]*;
 *   order 2: (cotangent C [E,3] of gD) out0 = c_gU [E,3], out1 = c_gd [E], out2 = c_D [E,3].      (* = may be NULL: zero) */
int hermnet_edge_unit(int order, const float* D, const float* gU, const float* gd, const float* C, long num_edges, float* out0,
                      float* out1, float* out2, void* stream);

/* The basis window of the bucketed rbf_proj and its two derivatives (ABI v13; /root/reference/HermNet/rmnet.py:168-193 on the 32
 * centres of a chunk): phi[r][k] = w[c][k] e(u_r) exp(coeff (u_r - mu[c][k])^2), c = r / rows_per_chunk, e = the polynomial
 * envelope of exponent env_p for u < 1 on rows with src[r] < num_edges (rows that hold an edge), 0 otherwise.
 *   order 0: out0 = phi [rows,32];   order 1: out0 = g_u [rows] = sum_k g[r][k] dphi/du  (g [rows,32]);
 *   order 2: out0 = d_g [rows,32] = cu[r] dphi/du, out1 = d_u [rows] = cu[r] sum_k g[r][k] d2phi/du2  (cu [rows]).
 * mu, w [num_chunks,32]: centres and column weights (0 for centres outside the basis). */
int hermnet_basis_window(int order, const float* u, const long* src, long num_edges, const float* mu, const float* w,
                         long num_chunks, int rows_per_chunk, float coeff, int env_p, const float* g, const float* cu, float* out0,
                         float* out1, void* stream);

/* Host-side (CPU) evaluation of the per-edge radial contraction exactly as the device code
 * computes it (banded 12-tap Gaussian window): rb[c] = b[c] + env(u) * sum_k W[c,k] g_k(u) and
 * its derivative d rb / d d.  Used by the CPU test-suite to check the banded formulation
 * against the dense reference formula without a GPU.  All pointers are HOST pointers. */
int hermnet_host_rbf_row(const float* offset_host, int num_rbf, float inv_rc, float coeff,
                         int env_kind, int env_p, const float* wt_host /* [R,C] */,
                         const float* b_host /* [C] */, int C, float d,
                         float* rb_host /* [C] */, float* drb_host /* [C] */);

#ifdef __cplusplus
}
#endif
#endif /* HERMNET_HIP_H */
