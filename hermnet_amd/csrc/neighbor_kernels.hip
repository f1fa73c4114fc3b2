// gfx950 cutoff neighbour search (cell list) -- the input producer of the hot path.
//
// Replaces /root/reference/HermNet/data.py:14-24 `neighbor_search`, which calls
// ase.neighborlist.primitive_neighbor_list('ijS', ...) (periodic) or torch_cluster.radius_graph
// on the HOST every step (plugin/ase_interface/calculator.py:49, plugin/lmp_interface/lmp_calc.py:224).
// Same conventions as the host implementation in hermnet_amd/neighbor.py (which it must match
// bit for bit on well-separated inputs): pair (i, j, S) listed iff |pos_j - pos_i + S cell| < rc
// (strict), no (i, i, 0), self images allowed, output sorted by (i, j, Sx, Sy, Sz); float64
// arithmetic on the float32 coordinates.
//
// Pipeline: wrap + bin (fractional bins >= rc wide) -> sort atoms by bin -> ONE pass over the candidates that counts
// an atom's pairs and stashes their 64-bit (i, j, S) keys in a per-atom slot of kStash entries -> exclusive scan ->
// (host reads E) -> per-atom rank sort of the stashed keys, decoded straight into edge_index / edge_shift.
// Round 3: the keys of an atom are written by its own wave, so the list is already grouped by i and only needs
// ordering INSIDE an atom (tens of entries): the global radix sort of E 64-bit keys (0.16 ms at 431k pairs) and the
// second pass over the candidates (0.07 ms) are gone.  An atom with more than kStash pairs makes the caller take the
// two-pass form (count, then fill).  Integer/streaming work.
#include <hip/hip_runtime.h>
#include "scan_i32.h"
#include <stdint.h>
#include "../../include/hermnet_hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxImg = 8;                 // |S| per axis after un-wrapping must stay below this
constexpr int kCode = 2 * kMaxImg + 1;     // 17 values per axis
constexpr int kStash = 160;                // keys kept per atom by the counting pass (fcc at rc = 5 A: 43)
inline dim3 grid_for(long n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

struct NbrGeom {
  double cell[9];     // rows = lattice vectors (identity for open systems)
  double inv[9];      // inverse
  double lo[3];       // open systems: lower corner of the bounding box
  int nbins[3];
  int reach[3];       // neighbour bins to visit on each side
  int periodic;       // all three axes periodic, or none
  double rc2;
};

__device__ __forceinline__ void frac_of(const NbrGeom& g, const double* p, double* f) {
  // p @ inv  (row vector times matrix)
  f[0] = p[0] * g.inv[0] + p[1] * g.inv[3] + p[2] * g.inv[6];
  f[1] = p[0] * g.inv[1] + p[1] * g.inv[4] + p[2] * g.inv[7];
  f[2] = p[0] * g.inv[2] + p[1] * g.inv[5] + p[2] * g.inv[8];
}

__device__ __forceinline__ void cart_of(const NbrGeom& g, const double* f, double* p) {
  p[0] = f[0] * g.cell[0] + f[1] * g.cell[3] + f[2] * g.cell[6];
  p[1] = f[0] * g.cell[1] + f[1] * g.cell[4] + f[2] * g.cell[7];
  p[2] = f[0] * g.cell[2] + f[1] * g.cell[5] + f[2] * g.cell[8];
}

// wrapped fractional coordinate, integer wrap, bin id
__global__ __launch_bounds__(kBlock) void nbr_bin_kernel(const float* __restrict__ pos, int N, NbrGeom g,
                                                        double* __restrict__ fw, int* __restrict__ wrap,
                                                        unsigned* __restrict__ bin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double p[3] = {(double)pos[3 * i], (double)pos[3 * i + 1], (double)pos[3 * i + 2]};
  double f[3];
  int b[3];
  if (g.periodic) {
    frac_of(g, p, f);
    for (int k = 0; k < 3; ++k) {
      const double w = floor(f[k]);
      wrap[3 * i + k] = (int)w;
      f[k] -= w;
      int bk = (int)(f[k] * g.nbins[k]);
      b[k] = bk >= g.nbins[k] ? g.nbins[k] - 1 : bk;
    }
  } else {
    for (int k = 0; k < 3; ++k) {
      f[k] = p[k];
      wrap[3 * i + k] = 0;
      int bk = (int)((p[k] - g.lo[k]) * g.inv[4 * k]);     // inv diagonal = 1 / bin width
      b[k] = bk < 0 ? 0 : (bk >= g.nbins[k] ? g.nbins[k] - 1 : bk);
    }
  }
  fw[3 * i] = f[0]; fw[3 * i + 1] = f[1]; fw[3 * i + 2] = f[2];
  bin[i] = (unsigned)((b[0] * g.nbins[1] + b[1]) * g.nbins[2] + b[2]);
}

// Atoms grouped by bin with a counting sort (histogram -> exclusive scan = bin_start -> scatter through per-bin cursors).
// The order INSIDE a bin is whatever the atomics give: it only decides the order in which an atom's keys reach its
// stash slot, and those are rank-sorted before they are decoded -- the list does not depend on it.  (Round 4: replaces a
// library radix sort + scan, whose look-back state does not survive hipGraph replays interleaved with eager runs --
// csrc/relation_kernels.hip found that out for the relation build -- so that search + step can be ONE captured graph.)
__global__ __launch_bounds__(kBlock) void nbr_zero_kernel(int* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = 0;
}
__global__ __launch_bounds__(kBlock) void nbr_bin_hist_kernel(const unsigned* __restrict__ bin, int N, int* __restrict__ hist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) atomicAdd(&hist[bin[i]], 1);
}
__global__ __launch_bounds__(kBlock) void nbr_bin_scatter_kernel(const unsigned* __restrict__ bin, int N,
                                                                const int* __restrict__ start, int* __restrict__ fill,
                                                                int* __restrict__ ids_sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) ids_sorted[start[bin[i]] + atomicAdd(&fill[bin[i]], 1)] = i;
}

// Visit every candidate (j, image) of atom i.  MODE 0 counts and stashes the keys at keys[i * kStash ...] (flags: bit 0
// |S| overflow, bit 1 an atom with more than kStash pairs), MODE 1 writes the keys at keys[offset[i] ...] (two-pass form).
// key = ((i * N + j) * 17^3 + code(S)),  S = image - wrap_j + wrap_i (shift for the caller's coordinates).
template <int MODE>
__global__ __launch_bounds__(kBlock) void nbr_pairs_kernel(const double* __restrict__ fw, const int* __restrict__ wrap,
                                                          const int* __restrict__ sorted_ids,
                                                          const int* __restrict__ bin_start, int N, NbrGeom g,
                                                          const long* __restrict__ offset, int* __restrict__ count,
                                                          unsigned long long* __restrict__ keys,
                                                          int* __restrict__ overflow,
                                                          const unsigned char* __restrict__ target_ok, int target_is_j,
                                                          int stash) {
  // one WAVE per atom: the lanes share the candidates of a bin (one atom per thread left the chip at 40 workgroups
  // for 10k atoms, each thread walking ~200 candidates serially: 0.2 ms per pass)
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= N) return;
  const int lane = threadIdx.x & 63;
  // `target_ok` (atom-sharded lists, sharding.py): only pairs whose TARGET atom is flagged are listed -- the target is j
  // in the periodic convention ([i; j]), i in the open-system one ([j; i])
  if (target_ok != nullptr && !target_is_j && !target_ok[i]) {
    if (MODE == 0 && lane == 0) count[i] = 0;
    return;
  }
  const bool mask_j = target_ok != nullptr && target_is_j;
  const double fi[3] = {fw[3 * i], fw[3 * i + 1], fw[3 * i + 2]};
  double pi[3];
  if (g.periodic) cart_of(g, fi, pi); else { pi[0] = fi[0]; pi[1] = fi[1]; pi[2] = fi[2]; }
  int bi[3];
  for (int k = 0; k < 3; ++k) {
    int bk;
    if (g.periodic) bk = (int)(fi[k] * g.nbins[k]); else bk = (int)((fi[k] - g.lo[k]) * g.inv[4 * k]);
    bi[k] = bk < 0 ? 0 : (bk >= g.nbins[k] ? g.nbins[k] - 1 : bk);
  }
  constexpr bool FILL = MODE == 1;
  long out = FILL ? offset[i] : (long)i * stash;
  int n = 0;
  for (int ox = -g.reach[0]; ox <= g.reach[0]; ++ox)
    for (int oy = -g.reach[1]; oy <= g.reach[1]; ++oy)
      for (int oz = -g.reach[2]; oz <= g.reach[2]; ++oz) {
        int tb[3] = {bi[0] + ox, bi[1] + oy, bi[2] + oz};
        int img[3] = {0, 0, 0};
        bool ok = true;
        for (int k = 0; k < 3; ++k) {
          if (g.periodic) {
            // floor division: image shift carried by leaving the cell through this face
            int q = tb[k] >= 0 ? tb[k] / g.nbins[k] : -((-tb[k] + g.nbins[k] - 1) / g.nbins[k]);
            img[k] = q;
            tb[k] -= q * g.nbins[k];
          } else if (tb[k] < 0 || tb[k] >= g.nbins[k]) {
            ok = false;
          }
        }
        if (!ok) continue;
        const int b = (tb[0] * g.nbins[1] + tb[1]) * g.nbins[2] + tb[2];
        const int s_end = bin_start[b + 1];
        for (int s0 = bin_start[b]; s0 < s_end; s0 += 64) {      // (wave-uniform loop: ballots are well defined)
          const int s = s0 + lane;
          bool hit = false;
          int j = 0;
          if (s < s_end) {
            j = sorted_ids[s];
            double fj[3] = {fw[3 * j] + img[0], fw[3 * j + 1] + img[1], fw[3 * j + 2] + img[2]};
            double pj[3];
            if (g.periodic) cart_of(g, fj, pj); else { pj[0] = fj[0]; pj[1] = fj[1]; pj[2] = fj[2]; }
            const double dx = pj[0] - pi[0], dy = pj[1] - pi[1], dz = pj[2] - pi[2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            hit = (d2 < g.rc2) && !(j == i && img[0] == 0 && img[1] == 0 && img[2] == 0);
            if (mask_j) hit = hit && target_ok[j] != 0;
          }
          const unsigned long long m = __ballot(hit);
          if (hit) {
            const int S[3] = {img[0] - wrap[3 * j] + wrap[3 * i], img[1] - wrap[3 * j + 1] + wrap[3 * i + 1],
                              img[2] - wrap[3 * j + 2] + wrap[3 * i + 2]};
            if (S[0] < -kMaxImg || S[0] > kMaxImg || S[1] < -kMaxImg || S[1] > kMaxImg || S[2] < -kMaxImg ||
                S[2] > kMaxImg)
              atomicOr(overflow, 1);
            const unsigned long long code =
                (unsigned long long)(((S[0] + kMaxImg) * kCode + (S[1] + kMaxImg)) * kCode + (S[2] + kMaxImg));
            // position inside the atom's key range: hits of earlier lanes first (the keys are sorted afterwards,
            // so only "each slot written once" matters)
            const int slot = __popcll(m & ((1ull << lane) - 1ull));
            if (FILL || n + slot < stash)
              keys[out + slot] = ((unsigned long long)i * (unsigned long long)N + (unsigned long long)j) *
                                     (unsigned long long)(kCode * kCode * kCode) + code;
          }
          const int nh = __popcll(m);
          out += nh;
          n += nh;
        }
      }
  if (!FILL && lane == 0) {
    count[i] = n;
    if (n > stash) atomicOr(overflow, 2);
  }
}

// sorted keys -> edge_index [2,E] int64 ([i; j]) and shifts [E,3] float32 (sign * S)
__global__ __launch_bounds__(kBlock) void nbr_decode_kernel(const unsigned long long* __restrict__ keys, long E,
                                                           int N, float sign, int swap_rows,
                                                           long* __restrict__ edge_index, float* __restrict__ shift) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const unsigned long long key = keys[e];
  const unsigned long long c3 = (unsigned long long)(kCode * kCode * kCode);
  const unsigned long long pair = key / c3;
  const int code = (int)(key - pair * c3);
  const long i = (long)(pair / (unsigned long long)N), j = (long)(pair - (unsigned long long)i * N);
  edge_index[e] = swap_rows ? j : i;
  edge_index[E + e] = swap_rows ? i : j;
  if (shift != nullptr) {
    shift[3 * e + 0] = sign * (float)(code / (kCode * kCode) - kMaxImg);
    shift[3 * e + 1] = sign * (float)((code / kCode) % kCode - kMaxImg);
    shift[3 * e + 2] = sign * (float)(code % kCode - kMaxImg);
  }
}

// One wave per atom: rank sort of its keys (unique, so rank = number of smaller keys) and decode into the caller's
// arrays at offset[i] + rank.  `stride` = kStash (stashed keys at src[i * kStash]) or 0 (keys at src[offset[i]]).
// (E = columns of edge_index; in capacity mode the list may hold more pairs than that: positions >= E are dropped and
// nbr_pad_kernel reports it)
__global__ __launch_bounds__(kBlock) void nbr_sort_decode_kernel(const unsigned long long* __restrict__ src, int stride,
                                                                const int* __restrict__ count, const long* __restrict__ offset,
                                                                int N, long E, float sign, int swap_rows,
                                                                long* __restrict__ edge_index, float* __restrict__ shift) {
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= N) return;
  const int lane = threadIdx.x & 63;
  const int cnt = count[i];
  const int n = stride ? min(cnt, stride) : cnt;               // (a stash holds at most `stride` keys of an atom)
  const long base = offset[i];
  const unsigned long long* keys = src + (stride ? (long)i * stride : base);
  const unsigned long long c3 = (unsigned long long)(kCode * kCode * kCode);
  for (int a = lane; a < n; a += 64) {
    const unsigned long long key = keys[a];
    int rank = 0;
    for (int b = 0; b < n; ++b) rank += keys[b] < key ? 1 : 0;
    const long e = base + rank;
    if (e >= E) continue;
    const unsigned long long pair = key / c3;
    const int code = (int)(key - pair * c3);
    const unsigned long long ii = pair / (unsigned long long)N;
    const long j = (long)(pair - ii * (unsigned long long)N);
    // a key whose image shift left the code's range (flag bit 0) has spilled into the pair field: the pair it decodes to is
    // not this atom's -- such a column becomes a NULL edge, never an index that a later kernel would follow out of bounds
    const bool ok = ii == (unsigned long long)i;
    edge_index[e] = ok ? (swap_rows ? j : (long)i) : -1;
    edge_index[E + e] = ok ? (swap_rows ? (long)i : j) : -1;
    if (shift != nullptr) {
      shift[3 * e + 0] = ok ? sign * (float)(code / (kCode * kCode) - kMaxImg) : 0.f;
      shift[3 * e + 1] = ok ? sign * (float)((code / kCode) % kCode - kMaxImg) : 0.f;
      shift[3 * e + 2] = ok ? sign * (float)(code % kCode - kMaxImg) : 0.f;
    }
  }
  // An atom with more pairs than its stash slot (flag bit 1): offset[] counts ALL its pairs, the stash holds the first
  // `stride` -- the columns base + n .. base + cnt - 1 belong to keys that were never kept.  They become NULL edges (the
  // list is incomplete and the caller repeats the search, but whatever runs on it before the flags are read -- the model
  // step of an MD loop does -- must not meet uninitialised indices).
  for (int a = n + lane; a < cnt; a += 64) {
    const long e = base + a;
    if (e >= E) break;
    edge_index[e] = -1;
    edge_index[E + e] = -1;
    if (shift != nullptr) { shift[3 * e] = 0.f; shift[3 * e + 1] = 0.f; shift[3 * e + 2] = 0.f; }
  }
}

// Capacity mode: the columns behind the list's last pair become NULL edges (-1, -1; shift 0) -- the relation build files
// them behind every row, so no row-walking kernel ever meets one -- and total[0] = number of pairs found, total[1] = flags
// (bit 0: image shift overflow, bit 1: an atom with more pairs than its stash slot, bit 2: more pairs than columns).
// With bits 1 or 2 set the list is incomplete: the caller repeats the search in its exact (two-call) form.
__global__ __launch_bounds__(kBlock) void nbr_pad_kernel(const long* __restrict__ offset, const int* __restrict__ overflow,
                                                        int N, long cap, long* __restrict__ edge_index,
                                                        float* __restrict__ shift, long* __restrict__ total) {
  const long found = offset[N];
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) {
    total[0] = found;
    total[1] = (long)(overflow[0] | (found > cap ? 4 : 0));
  }
  if (e < found || e >= cap) return;
  edge_index[e] = -1;
  edge_index[cap + e] = -1;
  if (shift != nullptr) { shift[3 * e] = 0.f; shift[3 * e + 1] = 0.f; shift[3 * e + 2] = 0.f; }
}

// (set / copy as kernels, not hipMemsetAsync / hipMemcpyAsync: captured memset nodes did not survive eager memsets
// between two replays on ROCm 7.2 -- csrc/relation_kernels.hip --, and the search is part of a captured MD step)
__global__ void nbr_clear_kernel(int* __restrict__ count_end, int* __restrict__ overflow) {
  if (threadIdx.x == 0) { count_end[0] = 0; overflow[0] = 0; overflow[1] = 0; }
}
__global__ void nbr_total_kernel(const long* __restrict__ offset_end, const int* __restrict__ overflow, long* __restrict__ total) {
  if (threadIdx.x == 0) { total[0] = offset_end[0]; total[1] = (long)overflow[0]; }
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

int key_bits(int N) {
  double v = (double)N * (double)N * (double)(kCode * kCode * kCode);
  int b = 1;
  while (b < 64 && ldexp(1.0, b) < v) ++b;
  return b;
}

struct NbrWork {
  double* fw; int* wrap; unsigned* bin; int* bin_fill; int* ids_sorted; int* bin_start;
  int* count; long* offset; int* overflow; unsigned long long* stash; void* temp; size_t temp_bytes;
};

size_t temp_bytes_for(int N) {      // scan scratch for the longer of the two scans (bins: <= 8 N + 65 counters; atoms: N + 1)
  return scan_temp_bytes(8 * N + 66);
}

int make_geom(const double* cell_host, const double* lo_host, const double* hi_host, double rc, NbrGeom& g, long& nbins) {
  g.rc2 = rc * rc;
  if (cell_host) {
    g.periodic = 1;
    double c[9];
    for (int k = 0; k < 9; ++k) { c[k] = cell_host[k]; g.cell[k] = c[k]; }
    const double det = c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) + c[2] * (c[3] * c[7] - c[4] * c[6]);
    if (fabs(det) < 1e-12) return HN_ERR_BAD_ARG;
    g.inv[0] = (c[4] * c[8] - c[5] * c[7]) / det; g.inv[1] = (c[2] * c[7] - c[1] * c[8]) / det; g.inv[2] = (c[1] * c[5] - c[2] * c[4]) / det;
    g.inv[3] = (c[5] * c[6] - c[3] * c[8]) / det; g.inv[4] = (c[0] * c[8] - c[2] * c[6]) / det; g.inv[5] = (c[2] * c[3] - c[0] * c[5]) / det;
    g.inv[6] = (c[3] * c[7] - c[4] * c[6]) / det; g.inv[7] = (c[1] * c[6] - c[0] * c[7]) / det; g.inv[8] = (c[0] * c[4] - c[1] * c[3]) / det;
    for (int k = 0; k < 3; ++k) {
      // plane spacing along axis k = 1 / |column k of inv|
      const double h = 1.0 / sqrt(g.inv[k] * g.inv[k] + g.inv[3 + k] * g.inv[3 + k] + g.inv[6 + k] * g.inv[6 + k]);
      int nb = (int)floor(h / rc);
      if (nb < 1) nb = 1;
      if (nb > 1024) nb = 1024;
      g.nbins[k] = nb;
      g.reach[k] = (int)ceil(rc / (h / nb) - 1e-12);
      if (g.reach[k] < 1) g.reach[k] = 1;
      if (g.reach[k] > kMaxImg) return HN_ERR_BAD_ARG;       // cell far smaller than the cutoff
      g.lo[k] = 0.0;
    }
  } else {
    g.periodic = 0;
    for (int k = 0; k < 9; ++k) { g.cell[k] = (k % 4 == 0) ? 1.0 : 0.0; g.inv[k] = 0.0; }
    for (int k = 0; k < 3; ++k) {
      const double span = hi_host[k] - lo_host[k] + 2e-6;
      int nb = (int)floor(span / rc);
      if (nb < 1) nb = 1;
      if (nb > 1024) nb = 1024;
      g.nbins[k] = nb;
      g.lo[k] = lo_host[k] - 1e-6;
      g.inv[4 * k] = nb / span;                               // 1 / bin width (>= rc wide)
      g.reach[k] = 1;
    }
  }
  nbins = (long)g.nbins[0] * g.nbins[1] * g.nbins[2];
  return HN_OK;
}

void carve(void* workspace, int N, long nbins, int stash, NbrWork& w) {
  char* p = reinterpret_cast<char*>(workspace);
  auto take = [&](size_t bytes) { void* r = p; p += align256(bytes); return r; };
  w.fw = (double*)take(sizeof(double) * 3 * (size_t)N);
  w.wrap = (int*)take(sizeof(int) * 3 * (size_t)N);
  w.bin = (unsigned*)take(sizeof(unsigned) * (size_t)N);
  w.ids_sorted = (int*)take(sizeof(int) * (size_t)N);
  w.bin_start = (int*)take(sizeof(int) * (size_t)(nbins + 1));
  w.bin_fill = (int*)take(sizeof(int) * (size_t)(nbins + 1));
  w.count = (int*)take(sizeof(int) * (size_t)(N + 1));
  w.offset = (long*)take(sizeof(long) * (size_t)(N + 1));
  w.overflow = (int*)take(256);
  w.stash = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)N * stash);
  w.temp = p;
}

size_t fixed_bytes(int N, long nbins, int stash) {
  return align256(sizeof(double) * 3 * (size_t)N) + align256(sizeof(int) * 3 * (size_t)N) + 2 * align256(sizeof(int) * (size_t)N) +
         2 * align256(sizeof(int) * (size_t)(nbins + 1)) + align256(sizeof(int) * (size_t)(N + 1)) +
         align256(sizeof(long) * (size_t)(N + 1)) + 256 + align256(sizeof(unsigned long long) * (size_t)N * stash);
}

size_t workspace_for(int num_atoms, int stash) {
  // the bin grid is coarsened to at most 8 bins per atom
  const long nbins = 8l * (num_atoms > 0 ? num_atoms : 1) + 64;
  return fixed_bytes(num_atoms, nbins, stash) + align256(temp_bytes_for(num_atoms > 0 ? num_atoms : 1)) + 512;
}

// the per-atom stash slot a workspace of this size provides: the largest value <= kStash whose workspace fits
// (both calls of a search derive it from the same (num_atoms, workspace_bytes), so they agree)
int stash_of(int num_atoms, size_t workspace_bytes) {
  int lo = 0, hi = kStash;
  while (lo < hi) {
    const int mid = (lo + hi + 1) / 2;
    if (workspace_for(num_atoms, mid) <= workspace_bytes) lo = mid; else hi = mid - 1;
  }
  return lo;
}

}  // namespace

extern "C" size_t hermnet_neighbor_workspace(int num_atoms) { return workspace_for(num_atoms, kStash); }
extern "C" size_t hermnet_neighbor_workspace_for(int num_atoms, int stash_per_atom) {
  const int st = stash_per_atom < 8 ? 8 : (stash_per_atom > kStash ? kStash : stash_per_atom);
  return workspace_for(num_atoms, st);
}

namespace {
int stash_checked(int num_atoms, size_t workspace_bytes) {
  const int st = stash_of(num_atoms, workspace_bytes);
  return st >= 8 ? st : 0;
}
}  // namespace

extern "C" int hermnet_neighbor_count(const float* pos, int num_atoms, const double* cell_host,
                                      const double* lo_host, const double* hi_host, double rc,
                                      void* workspace, size_t workspace_bytes, const unsigned char* target_ok,
                                      long* total_device, void* stream) {
  const int N = num_atoms;
  if (N < 0 || rc <= 0.0 || !workspace || !total_device) return HN_ERR_BAD_ARG;
  if (!cell_host && (!lo_host || !hi_host)) return HN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (N == 0) return hipMemsetAsync(total_device, 0, 2 * sizeof(long), s) == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
  if (!pos) return HN_ERR_BAD_ARG;
  NbrGeom g;
  long nbins = 0;
  int rc_ = make_geom(cell_host, lo_host, hi_host, rc, g, nbins);
  if (rc_) return rc_;
  if (nbins > 8l * N + 64) {   // sparse box: coarsen the grid (bins only get wider, still >= rc)
    while (nbins > 8l * N + 64) {
      int kmax = 0;
      for (int k = 1; k < 3; ++k) if (g.nbins[k] > g.nbins[kmax]) kmax = k;
      if (g.nbins[kmax] <= 1) break;
      const int nb = (g.nbins[kmax] + 1) / 2;
      if (!g.periodic) g.inv[4 * kmax] *= (double)nb / g.nbins[kmax];
      g.nbins[kmax] = nb;
      nbins = (long)g.nbins[0] * g.nbins[1] * g.nbins[2];
    }
  }
  const int stash = stash_checked(N, workspace_bytes);
  if (stash == 0) return HN_ERR_BAD_ARG;
  NbrWork w;
  carve(workspace, N, 8l * N + 64, stash, w);
  w.temp_bytes = workspace_bytes - (size_t)((char*)w.temp - (char*)workspace);
  hipLaunchKernelGGL(nbr_bin_kernel, grid_for(N), dim3(kBlock), 0, s, pos, N, g, w.fw, w.wrap, w.bin);
  hipLaunchKernelGGL(nbr_zero_kernel, grid_for(nbins + 1), dim3(kBlock), 0, s, w.bin_fill, nbins + 1);
  hipLaunchKernelGGL(nbr_bin_hist_kernel, grid_for(N), dim3(kBlock), 0, s, w.bin, N, w.bin_fill);
  if (exclusive_scan_i32(w.bin_fill, w.bin_start, (int)nbins + 1, w.temp, w.temp_bytes, s) != HN_OK) return HN_ERR_BAD_ARG;
  hipLaunchKernelGGL(nbr_zero_kernel, grid_for(nbins + 1), dim3(kBlock), 0, s, w.bin_fill, nbins + 1);
  hipLaunchKernelGGL(nbr_bin_scatter_kernel, grid_for(N), dim3(kBlock), 0, s, w.bin, N, w.bin_start, w.bin_fill, w.ids_sorted);
  hipLaunchKernelGGL(nbr_clear_kernel, dim3(1), dim3(64), 0, s, w.count + N, w.overflow);
  hipLaunchKernelGGL(nbr_pairs_kernel<0>, grid_for((long)N * 64), dim3(kBlock), 0, s, w.fw, w.wrap, w.ids_sorted, w.bin_start, N,
                     g, (const long*)nullptr, w.count, w.stash, w.overflow, target_ok, g.periodic, stash);
  if (exclusive_scan_i32_to_long(w.count, w.offset, N + 1, w.temp, w.temp_bytes, s) != HN_OK) return HN_ERR_BAD_ARG;
  // total_device = (pairs found, flags of the pass)
  hipLaunchKernelGGL(nbr_total_kernel, dim3(1), dim3(64), 0, s, w.offset + N, w.overflow, total_device);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_neighbor_fill_padded(int num_atoms, void* workspace, size_t workspace_bytes, long capacity,
                                            float shift_sign, int source_first, long* edge_index, float* edge_shift,
                                            long* total_device, void* stream) {
  const int N = num_atoms;
  if (N <= 0 || capacity <= 0 || capacity > 0x7fffffffl || !workspace || !edge_index || !total_device) return HN_ERR_BAD_ARG;
  if ((double)N * N * 4913.0 >= 1.8e19) return HN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int stash = stash_checked(N, workspace_bytes);
  if (stash == 0) return HN_ERR_BAD_ARG;
  NbrWork w;
  carve(workspace, N, 8l * N + 64, stash, w);
  // the stashed keys of the counting pass, rank-sorted per atom, into the first `capacity` columns ...
  hipLaunchKernelGGL(nbr_sort_decode_kernel, grid_for((long)N * 64), dim3(kBlock), 0, s, w.stash, stash, w.count, w.offset, N,
                     capacity, shift_sign, source_first, edge_index, edge_shift);
  // ... NULL edges behind them, and the count + flags for whoever reads them (the host: at the END of the step)
  hipLaunchKernelGGL(nbr_pad_kernel, grid_for(capacity), dim3(kBlock), 0, s, w.offset, w.overflow, N, capacity, edge_index,
                     edge_shift, total_device);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}

extern "C" int hermnet_neighbor_fill(const float* pos, int num_atoms, const double* cell_host,
                                     const double* lo_host, const double* hi_host, double rc,
                                     void* workspace, size_t workspace_bytes, long num_edges, float shift_sign,
                                     int source_first, int stash_ok, unsigned long long* keys,
                                     const unsigned char* target_ok, long* edge_index, float* edge_shift,
                                     void* stream) {
  const int N = num_atoms;
  if (N <= 0 || num_edges < 0 || !workspace || !edge_index) return HN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (num_edges == 0) return HN_OK;
  if ((double)N * N * 4913.0 >= 1.8e19 || num_edges > 0x7fffffffl || (!stash_ok && !keys)) return HN_ERR_BAD_ARG;
  NbrGeom g;
  long nbins = 0;
  int rc_ = make_geom(cell_host, lo_host, hi_host, rc, g, nbins);
  if (rc_) return rc_;
  while (nbins > 8l * N + 64) {   // must mirror hermnet_neighbor_count
    int kmax = 0;
    for (int k = 1; k < 3; ++k) if (g.nbins[k] > g.nbins[kmax]) kmax = k;
    if (g.nbins[kmax] <= 1) break;
    const int nb = (g.nbins[kmax] + 1) / 2;
    if (!g.periodic) g.inv[4 * kmax] *= (double)nb / g.nbins[kmax];
    g.nbins[kmax] = nb;
    nbins = (long)g.nbins[0] * g.nbins[1] * g.nbins[2];
  }
  const int stash = stash_checked(N, workspace_bytes);
  if (stash == 0) return HN_ERR_BAD_ARG;
  NbrWork w;
  carve(workspace, N, 8l * N + 64, stash, w);
  w.temp_bytes = workspace_bytes - (size_t)((char*)w.temp - (char*)workspace);
  if (!stash_ok)    // an atom had more pairs than its stash slot: second pass over the candidates into `keys`
    hipLaunchKernelGGL(nbr_pairs_kernel<1>, grid_for((long)N * 64), dim3(kBlock), 0, s, w.fw, w.wrap, w.ids_sorted, w.bin_start, N,
                       g, w.offset, (int*)nullptr, keys, w.overflow, target_ok, g.periodic, 0);
  hipLaunchKernelGGL(nbr_sort_decode_kernel, grid_for((long)N * 64), dim3(kBlock), 0, s,
                     stash_ok ? w.stash : keys, stash_ok ? stash : 0, w.count, w.offset, N, num_edges, shift_sign,
                     source_first, edge_index, edge_shift);
  return hipGetLastError() == hipSuccess ? HN_OK : HN_ERR_LAUNCH;
}
