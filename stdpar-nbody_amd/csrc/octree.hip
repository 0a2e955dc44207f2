// Octree Barnes-Hut for gfx950 — replaces src/octree.h of the reference (its default --algorithm).
//
// The reference builds the tree by concurrent insertion with per-node spin locks and a bump allocator
// (src/octree.h:114-181) and needs parallel forward progress (`par`); node NUMBERS depend on thread timing, but
// everything observable does not: a cell is split iff it holds >= 2 bodies, children are visited in hyperant order
// (src/octree.h:63-71), and a node's monopole is the sum of its 2^D children in child order (src/octree.h:205-216).
// So the tree is rebuilt here WITHOUT locks, deterministically:
//   bounds    scalar min/max over all coordinates (src/octree.h:93-112)
//   keys      every body walks the implicit cell hierarchy on its own, replaying the reference's exact
//             `divide[i] += (2*(pos[i] > divide[i]) - 1) * half_length` chain (src/octree.h:127-138), and records the
//             hyperant taken at each of MAXL levels: a 63/64-bit path key
//   sort      stable LSD radix sort of (key, body) — bodies of a cell are then contiguous
//   build     in ONE pass (default): every cell follows from the key digits neighbouring sorted bodies share, a prefix sum
//             numbers all cells in pre-order and every cell is built on its own — its end and its 2^D child ranges from the
//             sorted keys, the group numbers of its child cells from the same prefix sum (see "one-pass build" below).
//             Also kept, as cross-checks that give the same tree bit for bit (nbody_octree_set_build): breadth first, one
//             launch per level — an internal cell [start, end) finds its 2^D child ranges by binary search on the next key
//             digit; empty / single-body children become leaves, the others are queued for the next level.  Either way
//             children are allocated after their parents, as the traversal requires (src/octree.h:248-249)
//   multipoles  children before parents (src/octree.h:183-224, without the latch): two or three launches over rank chunks
//             (one-pass build), otherwise one launch per level, deepest first
//   force     the reference's walk per body (src/octree.h:226-263) with 2^D lanes per body: the children of an opened node
//             are examined side by side; every opening decision side/(sqrt(d2)+eps) < theta equals the reference's
//             bit for bit (quick bracketing test, IEEE sqrt and divide only inside the guard band), so the per-body
//             visit counters are exact; the accepted term m*(xj-x)/dx^3 uses polished v_rsq/v_rcp seeds.
// Key depth: MAXL = 21 (3D) / 32 (2D) levels; cells that still hold >= 2 bodies there are finished level by level by one
// thread each (ot_build_deep_kernel), so the tree has the reference's shape at any depth.
#include "common.hpp"
#include "radix_sort.hpp"
#include "to_sgpr.hpp"

#include <cstdlib>
#include <cstring>

namespace nbody {

constexpr int kOB            = 256;
constexpr uint32_t kOtEmpty  = 0xffffffffu;  // src/octree.h:37
constexpr uint32_t kOtBody   = 0xfffffffeu;  // src/octree.h:38
constexpr uint32_t kFlagDepth = 1u, kFlagCapacity = 2u, kFlagStack = 4u, kFlagWalk = 8u, kFlagBarrier = 16u;
constexpr int kOtDeepLevels = 128;  // total depth the deep build follows before it calls the bodies coincident
// its DFS stack: the cells on the current path stay for their second visit (<= kOtDeepLevels), plus the waiting siblings — the
// biggest child of a cell is entered LAST, so siblings wait only along the <= log2(n) levels where the path entered a smaller child
constexpr int kOtDeepFrames = kOtDeepLevels + 7 * 28 + 12;

template <int D>
constexpr int kMaxLevels = D == 3 ? 21 : 32;

// Per-level build: an internal cell waiting to be split — its node index, its range of sorted bodies and its rank (= sibling group
// number).  One-pass build: entry r describes the cell of rank r — node index, `start` = its level, `end` = one past the last rank
// of its subtree.
struct ot_cell {
  uint32_t node, start, end, rank;
};

// One tree node: monopole (src/octree.h:55-56 `m`), first child (src/octree.h:52) and depth.  The side of a node's cell
// is root_side * 2^-lvl EXACTLY — the reference halves a running value while it descends (src/octree.h:245), and
// power-of-two scaling is exact — so it is not stored: a visit reads 40 B.  In memory only the root has this shape.
template <typename T>
struct alignas(8 * sizeof(T)) ot_node {
  T p[3];
  T m;
  uint32_t fc, lvl;
};
static_assert(sizeof(ot_node<double>) == 64 && sizeof(ot_node<float>) == 32, "one aligned record per node");

// Storage: the nodes of a sibling group side by side in 16-byte pieces, so that the 2^D lanes that examine a group in the
// walk touch few cache lines per load — f64 3D: (p0,p1) | (p2,m) | (fc,lvl) = 2 + 2 + 1 lines with three loads
// instead of 8 + 8 + 8; f32: (p0,p1,p2,m) | (fc,lvl) = 2 + 1 lines with two loads.  The walk is bound by the
// texture-address unit, whose cost per load instruction is a fixed part plus a part per line touched (measured: fixed
// ~ 48 lines' worth; one 64-byte record per node made the walk 3.9 ms at N = 10^6, this layout 2.7 ms).
// fl[c] = (child group number or kOtEmpty / kOtBody, depth).
template <typename T, int D>
struct ot_group;
template <int D>
struct alignas(64) ot_group<float, D> {
  float pm[1u << D][4];  // (p0, p1, p2 or 0, m)
  uint32_t fl[1u << D][2];
  __device__ void store(uint32_t c, const ot_node<float>& nd, uint32_t fc) {
    pm[c][0] = nd.p[0];
    pm[c][1] = nd.p[1];
    pm[c][2] = nd.p[2];
    pm[c][3] = nd.m;
    fl[c][0] = fc;
    fl[c][1] = nd.lvl;
  }
  __device__ ot_node<float> load(uint32_t c) const {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const f4 v = *reinterpret_cast<const f4*>(pm[c]);
    const u2 f = *reinterpret_cast<const u2*>(fl[c]);
    ot_node<float> nd;
    nd.p[0] = v.x;
    nd.p[1] = v.y;
    nd.p[2] = v.z;
    nd.m    = v.w;
    nd.fc   = f.x;
    nd.lvl  = f.y;
    return nd;
  }
};
template <>
struct alignas(64) ot_group<double, 3> {
  double a[8][2];  // (p0, p1)
  double b[8][2];  // (p2, m)
  uint32_t fl[8][2];
  __device__ void store(uint32_t c, const ot_node<double>& nd, uint32_t fc) {
    a[c][0]  = nd.p[0];
    a[c][1]  = nd.p[1];
    b[c][0]  = nd.p[2];
    b[c][1]  = nd.m;
    fl[c][0] = fc;
    fl[c][1] = nd.lvl;
  }
  __device__ ot_node<double> load(uint32_t c) const {
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const d2 va = *reinterpret_cast<const d2*>(a[c]);
    const d2 vb = *reinterpret_cast<const d2*>(b[c]);
    const u2 f  = *reinterpret_cast<const u2*>(fl[c]);
    ot_node<double> nd;
    nd.p[0] = va.x;
    nd.p[1] = va.y;
    nd.p[2] = vb.x;
    nd.m    = vb.y;
    nd.fc   = f.x;
    nd.lvl  = f.y;
    return nd;
  }
};
template <>
struct alignas(64) ot_group<double, 2> {
  double a[4][2];  // (p0, p1)
  double m[4];
  uint32_t fl[4][2];
  __device__ void store(uint32_t c, const ot_node<double>& nd, uint32_t fc) {
    a[c][0]  = nd.p[0];
    a[c][1]  = nd.p[1];
    m[c]     = nd.m;
    fl[c][0] = fc;
    fl[c][1] = nd.lvl;
  }
  __device__ ot_node<double> load(uint32_t c) const {
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const d2 va = *reinterpret_cast<const d2*>(a[c]);
    const u2 f  = *reinterpret_cast<const u2*>(fl[c]);
    ot_node<double> nd;
    nd.p[0] = va.x;
    nd.p[1] = va.y;
    nd.p[2] = 0.0;
    nd.m    = m[c];
    nd.fc   = f.x;
    nd.lvl  = f.y;
    return nd;
  }
};

// The tree as the kernels see it: node 0 (the root) is a record of its own, node 1 + g * 2^D + c is child slot c of sibling
// group g.  Build-side code thinks in node indices (fc = index of the first child, as in the reference); the stored fc of a
// cell is its child GROUP number, which is what the walk wants.
template <typename T, int D>
struct ot_tree {
  static constexpr uint32_t NCH = 1u << D;
  ot_group<T, D>* groups;
  ot_node<T>* root;
  __device__ ot_node<T> get(uint32_t idx) const {
    ot_node<T> nd = idx == 0 ? *root : groups[(idx - 1u) / NCH].load((idx - 1u) % NCH);
    if (nd.fc < kOtBody) nd.fc = 1u + nd.fc * NCH;
    return nd;
  }
  __device__ void put(uint32_t idx, const ot_node<T>& nd) const {
    const uint32_t stored_fc = nd.fc < kOtBody ? (nd.fc - 1u) / NCH : nd.fc;
    if (idx == 0) {
      ot_node<T> r = nd;
      r.fc         = stored_fc;
      *root        = r;
    } else {
      groups[(idx - 1u) / NCH].store((idx - 1u) % NCH, nd, stored_fc);
    }
  }
};

template <typename T>
__device__ __forceinline__ T ot_fmin(T a, T b) {
  if constexpr (sizeof(T) == 4) return __builtin_fminf(a, b);
  else return __builtin_fmin(a, b);
}
template <typename T>
__device__ __forceinline__ T ot_fmax(T a, T b) {
  if constexpr (sizeof(T) == 4) return __builtin_fmaxf(a, b);
  else return __builtin_fmax(a, b);
}

// ---- bounds (src/octree.h:93-112) -----------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void ot_block_minmax(T& lo, T& hi, T* out2) {
  __shared__ T red[2][kOB / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = ot_fmin(lo, __shfl_xor(lo, off, 64));
    hi = ot_fmax(hi, __shfl_xor(hi, off, 64));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = lo;
    red[1][wave] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    T l = red[0][0], h = red[1][0];
    for (int w = 1; w < kOB / 64; ++w) {
      l = ot_fmin(l, red[0][w]);
      h = ot_fmax(h, red[1][w]);
    }
    out2[0] = l;
    out2[1] = h;
  }
}

template <typename T>
__global__ __launch_bounds__(kOB) void ot_bounds_partial_kernel(const T* __restrict__ x, uint64_t nelem, T* __restrict__ partials) {
  T lo = T(0), hi = T(0);  // the reduction starts from (0, 0): the root cube always contains the origin
  for (uint64_t e = uint64_t(blockIdx.x) * kOB + threadIdx.x; e < nelem; e += uint64_t(gridDim.x) * kOB) {
    const T p = x[e];
    lo        = ot_fmin(lo, p);
    hi        = ot_fmax(hi, p);
  }
  ot_block_minmax(lo, hi, partials + 2 * blockIdx.x);
}

template <typename T, int D>
__global__ __launch_bounds__(kOB) void ot_bounds_final_kernel(const T* __restrict__ partials, uint32_t nblk, T* __restrict__ root) {
#pragma clang fp contract(off)
  __shared__ T res[2];
  T lo = T(0), hi = T(0);
  for (uint32_t b = threadIdx.x; b < nblk; b += kOB) {
    lo = ot_fmin(lo, partials[2 * b]);
    hi = ot_fmax(hi, partials[2 * b + 1]);
  }
  ot_block_minmax(lo, hi, res);
  __syncthreads();
  if (threadIdx.x == 0) {
    T mx = res[1], mn = res[0];
    mx += T(1);  // "adjust boundary"
    mn -= T(1);
    const T divide = (mx + mn) / T(2);
#pragma unroll
    for (int k = 0; k < D; ++k) root[k] = divide;  // root_x = splat(divide)
    root[D] = mx - mn;                             // root_side_length
  }
}

// Small systems (n <= kSmallN: the reference's default run is 1000 bodies): one block does both halves.  1024: the step of the
// one-block kernels below is one CU's work — 0.081 ms at n = 1000 (float) against 0.129 through the launches of the large path, but
// 0.137 at 2048 where the large path (its kernels spread over the chip) takes ~0.12.
constexpr uint32_t kSmallN = 1024;
template <typename T, int D>
__global__ __launch_bounds__(kOB) void ot_bounds_small_kernel(const T* __restrict__ x, uint32_t nelem, T* __restrict__ root) {
#pragma clang fp contract(off)
  __shared__ T res[2];
  T lo = T(0), hi = T(0);
  for (uint32_t e = threadIdx.x; e < nelem; e += kOB) {
    const T p = x[e];
    lo        = ot_fmin(lo, p);
    hi        = ot_fmax(hi, p);
  }
  ot_block_minmax(lo, hi, res);
  __syncthreads();
  if (threadIdx.x == 0) {
    T mx = res[1], mn = res[0];
    mx += T(1);
    mn -= T(1);
    const T divide = (mx + mn) / T(2);
#pragma unroll
    for (int k = 0; k < D; ++k) root[k] = divide;
    root[D] = mx - mn;
  }
}

// ---- path keys (src/octree.h:127-138 replayed per body) ------------------------------------------------------------
template <typename T, int D>
__device__ __forceinline__ uint64_t ot_path_key(const T* __restrict__ x, uint64_t i, const T* __restrict__ root) {
#pragma clang fp contract(off)
  T pos[D], divide[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    pos[k]    = x[i * D + k];
    divide[k] = root[k];
  }
  T side       = root[D];
  uint64_t key = 0;
  for (int l = 0; l < kMaxLevels<D>; ++l) {
    const T half = side / T(4);  // /2 for the child's side, /2 for its half length
    uint32_t cp  = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const int gt = pos[k] > divide[k];
      cp |= uint32_t(gt) << k;
      divide[k] += T(2 * gt - 1) * half;
    }
    side /= T(2);
    key = (key << D) | cp;
  }
  return key;
}

template <typename T, int D>
__global__ __launch_bounds__(kOB) void ot_keys_kernel(const T* __restrict__ x, uint32_t n, const T* __restrict__ root,
                                                      uint64_t* __restrict__ keys) {
  const uint64_t i = uint64_t(blockIdx.x) * kOB + threadIdx.x;
  if (i >= n) return;
  keys[i] = ot_path_key<T, D>(x, i, root);
}

// ---- breadth-first build --------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ void ot_build_init_kernel(uint32_t n, const T* __restrict__ m, const T* __restrict__ x, ot_tree<T, D> tree,
                                     ot_cell* __restrict__ cells, uint32_t* __restrict__ lvl_count) {
  if (threadIdx.x < uint32_t(kMaxLevels<D> + 2)) lvl_count[threadIdx.x] = 0;
  if (threadIdx.x < 2) lvl_count[kMaxLevels<D> + 3 + threadIdx.x] = 0;  // the grid-barrier counters of the two all-level kernels
  if (threadIdx.x == 0) {  // (the overflow flags behind lvl_count are sticky: nbody_octree_info reports and clears them)
    ot_node<T> r;
#pragma unroll
    for (int k = 0; k < 3; ++k) r.p[k] = T(0);
    r.m    = T(0);
    r.lvl  = 0;
    r.fc   = kOtEmpty;
    if (n >= 2) {       // the root holds >= 2 bodies: it is the first cell to split
      cells[0]     = ot_cell{0u, 0u, n, 0u};
      lvl_count[0] = 1;
    } else {  // a single body stays in the root (src/octree.h:140-145)
#pragma unroll
      for (int k = 0; k < D; ++k) r.p[k] = x[k];
      r.m  = m[0];
      r.fc = kOtBody;
    }
    tree.put(0, r);
  }
}

// One thread per (cell, child): the 2^D lanes of a cell search their lower bounds side by side (the top levels are pure
// latency: one cell, ~20 dependent probes per search), take the upper bound from the neighbour lane and write one child
// record each — a cell's sibling group is one contiguous 2^D-record store.  Children that must be split are appended to
// the next level's cell list with ONE counter bump per 1024-thread block: returning same-address atomics complete at
// ~12 ns each on this chip whatever else the kernel does (one per wave made the widest level 402 us, 37k atomics).
constexpr int kOBuild = 1024;

#ifdef NBODY_EXPERIMENTS
#define OT_FORMS_PART 1
#include "experiments/octree_forms.inc"  // the grid barrier of build forms 2 and 4 (measured, not shipped)
#undef OT_FORMS_PART
#endif

// One block's share of one level: cells [vblock * 1024 / 2^D, ...) of `count`.  `cells`, `lvl_count` and the tree are written by
// other blocks during the all-level kernel, so nothing here is __restrict__ and the counters are read with agent-scope loads.
template <typename T, int D>
__device__ __forceinline__ void ot_build_level_body(int level, uint32_t vblock, uint32_t count, uint32_t base,
                                                    const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ sidx,
                                                    const T* __restrict__ m, const T* __restrict__ x, ot_tree<T, D> tree,
                                                    ot_cell* cells, uint32_t* lvl_count, uint32_t* flags, uint32_t capacity,
                                                    uint32_t max_cells) {
  constexpr uint32_t NCH = 1u << D;
  __shared__ uint32_t wave_first[kOBuild / 64];
  __shared__ uint32_t block_first;
  const uint32_t tid = vblock * kOBuild + threadIdx.x;
  const uint32_t k = tid / NCH, c = tid % NCH;
  const uint32_t rank = base + k;
  const uint32_t fc   = 1u + rank * NCH;  // its sibling group (the reference's bump allocator hands out the same shape)
  bool live           = k < count;        // lane groups of a cell stay together
  if (live && fc + NCH > capacity) {
    if (c == 0) atomicOr(flags, kFlagCapacity);
    live = false;
  }
  uint32_t lo = 0, end = 0;
  const uint32_t ci = fc + c;
  if (live) {
    const ot_cell cell = cells[rank];
    // child ranges: bodies are sorted by key, so the bodies of hyperant c are those whose digit at this level is c
    const int shift = D * (kMaxLevels<D> - 1 - level);
    uint32_t hi     = cell.end;
    lo              = cell.start;
    if (c != 0) {  // first position whose digit is >= c
      while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (((skeys[mid] >> shift) & (NCH - 1)) < c) lo = mid + 1;
        else hi = mid;
      }
    }
    const uint32_t up = __shfl_down(lo, 1, NCH);
    end               = c + 1 < NCH ? up : cell.end;
    ot_node<T> r;
#pragma unroll
    for (int q = 0; q < 3; ++q) r.p[q] = T(0);
    r.m    = T(0);  // empty leaf: zero monopole (src/octree.h:77-83); a cell's monopole is filled in by the multipole pass
    r.lvl  = uint32_t(level) + 1u;
    r.fc   = kOtEmpty;
    if (end - lo == 1) {                         // leaf with one body (src/octree.h:140-145, :163-165)
      const uint64_t b = sidx[lo];
#pragma unroll
      for (int q = 0; q < D; ++q) r.p[q] = x[b * D + q];
      r.m  = m[b];
      r.fc = kOtBody;
    }
    tree.put(ci, r);
    if (c == 0) {
      ot_node<T> pn = tree.get(cell.node);
      pn.fc         = fc;
      tree.put(cell.node, pn);
    }
  }
  // children holding >= 2 bodies are split on the next level (which sets their fc)
  const bool split      = live && end - lo >= 2;
  const uint64_t voters = __ballot(split);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  __syncthreads();  // (the previous share of this block is done with the two shared words)
  if (lane == 0) wave_first[wave] = uint32_t(__builtin_popcountll(voters));
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int w = 0; w < kOBuild / 64; ++w) {
      const uint32_t n_w = wave_first[w];
      wave_first[w]      = total;
      total += n_w;
    }
    block_first = total != 0 ? atomicAdd(&lvl_count[level + 1], total) : 0u;
  }
  __syncthreads();
  if (split) {
    const uint32_t pos =
        base + count + block_first + wave_first[wave] + uint32_t(__builtin_popcountll(voters & ((1ull << lane) - 1ull)));
    if (pos < max_cells) cells[pos] = ot_cell{ci, lo, end, pos};
    else atomicOr(flags, kFlagCapacity);
  }
}

// one launch per level (kept as the cross-check of the all-level kernel: nbody_octree_set_build)
template <typename T, int D>
__global__ __launch_bounds__(kOBuild) void ot_build_level_kernel(int level, const uint64_t* __restrict__ skeys,
                                                                 const uint32_t* __restrict__ sidx, const T* __restrict__ m,
                                                                 const T* __restrict__ x, ot_tree<T, D> tree, ot_cell* cells,
                                                                 uint32_t* lvl_count, uint32_t* flags, uint32_t capacity,
                                                                 uint32_t max_cells) {
  constexpr uint32_t NCH = 1u << D;
  const uint32_t count = lvl_count[level];
  if (blockIdx.x * (kOBuild / NCH) >= count) return;  // whole blocks beyond this level's cells
  uint32_t base = 0;  // cells of the shallower levels = rank of this level's first cell
  for (int j = 0; j < level; ++j) base += lvl_count[j];
  ot_build_level_body<T, D>(level, blockIdx.x, count, base, skeys, sidx, m, x, tree, cells, lvl_count, flags, capacity, max_cells);
}

// ---- cells below the key depth ------------------------------------------------------------------------------------------
// A cell at depth kMaxLevels still holding >= 2 bodies is finer than the keys resolve.  That is rare in a fresh system and
// routine in a long run: a few escapers inflate the root cube (side 3*10^4 after 400 steps of the 10^5-body galaxy) until
// root_side / 2^21 exceeds the spacing of the bodies left in the core.  The reference simply keeps splitting
// (src/octree.h:127-176), so these cells are finished the reference's way: one thread per cell follows the same
// `pos > divide` chain level by level, writes the sibling groups and, on the way back up, the monopoles in child order
// (src/octree.h:205-216).  Groups are numbered after all the breadth-first ones.  Bodies that have not separated after
// kOtDeepLevels levels are reported as coincident (the reference would split until its node pool overflows).
template <typename T, int D>
__device__ uint32_t ot_hyperant_at(const T* __restrict__ x, uint32_t body, const T* __restrict__ root, int level) {
#pragma clang fp contract(off)
  T pos[D], divide[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    pos[k]    = x[uint64_t(body) * D + k];
    divide[k] = root[k];
  }
  T side      = root[D];
  uint32_t cp = 0;
  for (int l = 0; l <= level; ++l) {  // the chain of ot_keys_kernel, continued to `level`
    const T half = side / T(4);
    cp           = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const int gt = pos[k] > divide[k];
      cp |= uint32_t(gt) << k;
      divide[k] += T(2 * gt - 1) * half;
    }
    side /= T(2);
  }
  return cp;
}

template <typename T, int D>
__device__ __forceinline__ void ot_build_deep_body(uint32_t k0, uint32_t kstride, uint32_t* sidx, uint32_t* tmp, const T* m, const T* x,
                                                   const T* root, ot_tree<T, D> tree, const ot_cell* cells, const ot_cell* tops,
                                                   uint32_t* lvl_count, uint32_t* flags, uint32_t capacity) {
#pragma clang fp contract(off)
  constexpr uint32_t NCH = 1u << D;
  constexpr int ML       = kMaxLevels<D>;
  uint32_t base = 0;  // cells of the levels above = position of the first cell of depth ML in the per-level build's list
  for (int j = 0; j < ML; ++j) base += lvl_count[j];
  // the cells of depth ML: the per-level build's list continues with them (tops == nullptr); the one-pass build appended them to
  // a list of their own, whose length is its cursor (a top whose parent ran out of node pool is counted but not listed)
  const uint32_t regular = base + lvl_count[ML];
  const uint32_t count   = tops ? lvl_count[ML + 5] : lvl_count[ML];
  if (!tops) tops = cells + base;
  struct frame {
    uint32_t node, s, e;
    uint32_t rank_level;  // level | done << 31 ; rank of a regular (depth ML) cell is implied by its position
  };
  frame stack[kOtDeepFrames];
  for (uint32_t k = k0; k < count; k += kstride) {
    if (tops == cells + base && base + k >= capacity / NCH + 1u) break;  // the per-level list ran out of room (flagged when it did)
    const ot_cell top = tops[k];
    int sp            = 0;
    stack[sp++]       = frame{top.node, top.start, top.end, uint32_t(ML)};
    bool failed       = false;
    while (sp > 0) {
      frame f         = stack[sp - 1];
      const int level = int(f.rank_level & 0x7fffffffu);
      if (f.rank_level >> 31) {  // second visit: every child cell is finished — this cell's monopole, children in order
        --sp;
        ot_node<T> pn = tree.get(f.node);
        T mass = T(0), xx[D];
#pragma unroll
        for (int q = 0; q < D; ++q) xx[q] = T(0);
        for (uint32_t c = 0; c < NCH; ++c) {
          const ot_node<T> ch = tree.get(pn.fc + c);
          mass += ch.m;
#pragma unroll
          for (int q = 0; q < D; ++q) xx[q] += ch.m * ch.p[q];
        }
#pragma unroll
        for (int q = 0; q < D; ++q) pn.p[q] = xx[q] / mass;
        pn.m = mass;
        tree.put(f.node, pn);
        continue;
      }
      // first visit: split [s, e) by the hyperant of this depth
      if (level >= kOtDeepLevels || failed) {
        if (!failed) atomicOr(flags, kFlagDepth);
        failed = true;  // leave the rest of this subtree as (empty) leaves
        --sp;
        continue;
      }
      const uint32_t rank = f.node == top.node ? top.rank : regular + atomicAdd(&lvl_count[ML + 1], 1u);
      const uint32_t fc   = 1u + rank * NCH;
      if (fc + NCH > capacity) {
        atomicOr(flags, kFlagCapacity);
        failed = true;
        --sp;
        continue;
      }
      // counting sort of the segment by hyperant, through the sort's idle index buffer (segments of different cells are disjoint)
      uint32_t bound[NCH + 1];
      for (uint32_t c = 0; c <= NCH; ++c) bound[c] = 0;
      for (uint32_t i = f.s; i < f.e; ++i) ++bound[ot_hyperant_at<T, D>(x, sidx[i], root, level) + 1];
      bound[0] = f.s;
      for (uint32_t c = 0; c < NCH; ++c) bound[c + 1] += bound[c];
      {
        uint32_t next[NCH];
        for (uint32_t c = 0; c < NCH; ++c) next[c] = bound[c];
        for (uint32_t i = f.s; i < f.e; ++i) {
          const uint32_t b = sidx[i];
          tmp[next[ot_hyperant_at<T, D>(x, b, root, level)]++] = b;
        }
        for (uint32_t i = f.s; i < f.e; ++i) sidx[i] = tmp[i];
      }
      {
        ot_node<T> pn = tree.get(f.node);
        pn.fc         = fc;
        tree.put(f.node, pn);
      }
      stack[sp - 1].rank_level = uint32_t(level) | 0x80000000u;  // come back for the monopole
      for (uint32_t c = 0; c < NCH; ++c) {
        const uint32_t cnt = bound[c + 1] - bound[c];
        ot_node<T> r;
#pragma unroll
        for (int q = 0; q < 3; ++q) r.p[q] = T(0);
        r.m   = T(0);
        r.lvl = uint32_t(level) + 1u;
        r.fc  = kOtEmpty;
        if (cnt == 1) {
          const uint64_t b = sidx[bound[c]];
#pragma unroll
          for (int q = 0; q < D; ++q) r.p[q] = x[b * D + q];
          r.m  = m[b];
          r.fc = kOtBody;
        }
        tree.put(fc + c, r);
      }
      // Children that are cells go on the stack; the one with the most bodies FIRST, so that it is entered last: a chain of
      // nested cells with waiting siblings at every level (tests/test_gpu_octree.py::_deep_chain) then holds its siblings
      // only while they are being finished, not all the way down.  The order in which cells are finished changes group
      // NUMBERS only (they come from an atomic counter anyway), never a monopole: those sum the children in child order.
      uint32_t big = NCH, bigcnt = 1;
      for (uint32_t c = 0; c < NCH; ++c) {
        const uint32_t cnt = bound[c + 1] - bound[c];
        if (cnt > bigcnt) {
          bigcnt = cnt;
          big    = c;
        }
      }
      for (uint32_t q = 0; q <= NCH; ++q) {  // q = 0: the biggest; then the others, last first
        const uint32_t c = q == 0 ? big : NCH - q;
        if (c >= NCH || (q > 0 && c == big) || bound[c + 1] - bound[c] < 2) continue;
        if (sp == kOtDeepFrames) {
          atomicOr(flags, kFlagDepth);
          failed = true;
          break;
        }
        stack[sp++] = frame{fc + c, bound[c], bound[c + 1], uint32_t(level) + 1u};
      }
    }
  }
}

template <typename T, int D>
__global__ __launch_bounds__(64) void ot_build_deep_kernel(uint32_t* __restrict__ sidx, uint32_t* __restrict__ tmp,
                                                           const T* __restrict__ m,
                                                           const T* __restrict__ x, const T* __restrict__ root,
                                                           ot_tree<T, D> tree, const ot_cell* __restrict__ cells,
                                                           const ot_cell* __restrict__ tops,
                                                           uint32_t* __restrict__ lvl_count, uint32_t* __restrict__ flags,
                                                           uint32_t capacity) {
  ot_build_deep_body<T, D>(blockIdx.x * 64 + threadIdx.x, gridDim.x * 64, sidx, tmp, m, x, root, tree, cells, tops, lvl_count, flags,
                           capacity);
}

// ---- multipoles (src/octree.h:205-216) ---------------------------------------------------------------------------------
template <typename T, int D>
__device__ __forceinline__ void ot_multipole_cell(ot_tree<T, D> tree, uint32_t node) {
#pragma clang fp contract(off)
  constexpr uint32_t NCH = 1u << D;
  ot_node<T> pn     = tree.get(node);
  const uint32_t fc = pn.fc;
  if (fc == kOtEmpty || fc == kOtBody) return;  // only after an overflow flag
  T mass = T(0), xx[D];
#pragma unroll
  for (int q = 0; q < D; ++q) xx[q] = T(0);
  for (uint32_t c = 0; c < NCH; ++c) {  // child order; empty children add (0, 0)
    const ot_node<T> ch = tree.get(fc + c);
    mass += ch.m;
#pragma unroll
    for (int q = 0; q < D; ++q) xx[q] += ch.m * ch.p[q];
  }
#pragma unroll
  for (int q = 0; q < D; ++q) pn.p[q] = xx[q] / mass;
  pn.m = mass;
  tree.put(node, pn);
}

// one launch per level (the cross-check of the all-level kernel)
template <typename T, int D>
__global__ __launch_bounds__(kOB) void ot_multipole_level_kernel(int level, ot_tree<T, D> tree,
                                                                 const ot_cell* __restrict__ cells,
                                                                 const uint32_t* __restrict__ lvl_count, uint32_t max_cells) {
  const uint32_t count = lvl_count[level];
  const uint32_t k     = blockIdx.x * kOB + threadIdx.x;
  if (k >= count) return;
  uint32_t base = 0;
  for (int j = 0; j < level; ++j) base += lvl_count[j];
  if (base + k >= max_cells) return;  // cells the list had no room for (the build raised kFlagCapacity): counted, never stored
  ot_multipole_cell<T, D>(tree, cells[base + k].node);
}

#ifdef NBODY_EXPERIMENTS
#define OT_FORMS_PART 2
#include "experiments/octree_forms.inc"  // the all-level kernels of build forms 2 and 4
#undef OT_FORMS_PART
#endif

// ---- one-pass build (nbody_octree_set_build 3 / auto) -----------------------------------------------------------------------
// The per-level build above is a chain of dependent launches (a level's cell list is the product of the level before), and a
// dependent launch costs >= 4.7 us on this chip whatever it does: 16 + 16 of them are a third of the octree step at the
// reference's benchmark size (N = 10^5, profiles/r03/small_trees_kernel_stats.txt); a grid barrier is no cheaper (~12 us).
// The sorted keys already hold the whole tree, though.  With l_i = the number of leading key digits bodies i-1 and i share
// (l_0 = l_n = -1), a cell of level d over the sorted range [s, e) exists exactly when s's left boundary has l_s < d and
// the range holds >= 2 bodies, i.e. l_(s+1) >= d: position i starts the cells of levels l_i + 1 ... l_(i+1), and nothing else
// does.  An exclusive prefix sum P of max(0, l_(i+1) - l_i) over the positions therefore NUMBERS every cell — rank(i, d) =
// P[i] + d - l_i - 1, pre-order: a cell's subtree is the rank interval [rank, P[e]) — and every cell can be built on its own:
// its end by a search for the first key with another d-digit prefix, its 2^D child ranges by the digit searches of the per-level
// build, and the group number of a child cell from the same formula — no cell waits for its parent.  Launches: ot_lcp_kernel
// (boundaries, block-local prefix, level histogram), ot_lcp_finish_kernel (one block: block bases, level counts for
// nbody_octree_info), ot_build_lcp_kernel, then ot_build_deep_kernel as before for cells below the key depth.
// The multipoles need children before parents.  In rank order every subtree is an interval, so a block that owns a chunk of ranks
// can finish, level by level with block barriers only, every cell whose subtree ends inside its chunk
// (ot_multipole_chunks_kernel); the cells it cannot — the <= kMaxLevels ancestors of each chunk boundary — are listed and
// finished by ONE block afterwards (ot_multipole_crown_kernel), for large trees after a second round of blocks over the list
// (ot_multipole_round_kernel).  Same children, same order, same arithmetic as
// ot_multipole_cell: the tree is the per-level build's up to the numbering of the sibling groups, and the walk (which pushes
// children in slot order) produces bit-identical forces and counters (tests/test_gpu_octree.py::test_octree_build_forms).
constexpr int kLcpB = 1024;   // positions per block of ot_lcp_kernel
constexpr int kLcpBuildB = 256;  // ot_build_lcp_kernel: lane groups are independent, small blocks give their slots back early
constexpr int kMpChunk = 512;  // ot_multipole_chunks_kernel: ranks (= threads) per block (512: a double 3D cell keeps 2^D x 4 values in registers)
constexpr int kMpCrown = 512;  // ot_multipole_crown_kernel: threads, and cells it finishes out of registers and LDS

template <int D>
__device__ __forceinline__ int ot_common_levels(uint64_t a, uint64_t b) {
  const uint64_t x = a ^ b;
  if (x == 0) return kMaxLevels<D>;
  return (__builtin_clzll(x) - (64 - D * kMaxLevels<D>)) / D;
}

// positions 0 ... n (n + 1 of them: P[n] = the number of cells)
template <int D>
__global__ __launch_bounds__(kLcpB) void ot_lcp_kernel(const uint64_t* __restrict__ skeys, uint32_t n, int8_t* __restrict__ lv,
                                                       uint32_t* __restrict__ plocal, uint32_t* __restrict__ bsum,
                                                       uint32_t* __restrict__ bhist) {
  constexpr int ML = kMaxLevels<D>;
  __shared__ uint32_t hist[ML + 1];
  __shared__ uint32_t wsum[kLcpB / 64];
  if (threadIdx.x <= uint32_t(ML)) hist[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * kLcpB + threadIdx.x;
  int li = -1, ln = -1;
  if (i < n) {
    const uint64_t k = skeys[i];
    if (i > 0) li = ot_common_levels<D>(skeys[i - 1], k);
    if (i + 1 < n) ln = ot_common_levels<D>(k, skeys[i + 1]);
  }
  if (i <= n) lv[i] = int8_t(li);
  const uint32_t cnt = ln > li ? uint32_t(ln - li) : 0u;
  for (int d = li + 1; d <= ln; ++d) atomicAdd(&hist[d], 1u);
  // block-wide exclusive prefix of cnt
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(inc, off, 64);
    if (lane >= uint32_t(off)) inc += up;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kLcpB / 64; ++w) {
    const uint32_t v = wsum[w];
    if (uint32_t(w) < wave) before += v;
    total += v;
  }
  if (i <= n) plocal[i] = before + inc - cnt;
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
  if (threadIdx.x <= uint32_t(ML)) bhist[blockIdx.x * (ML + 1) + threadIdx.x] = hist[threadIdx.x];
}

// one block: exclusive scan of the block sums (in place), the level counts, the counters of the later kernels, the root of n = 1
template <typename T, int D>
__global__ __launch_bounds__(1024) void ot_lcp_finish_kernel(uint32_t n, uint32_t nblk, uint32_t* __restrict__ bsum,
                                                             const uint32_t* __restrict__ bhist, const T* __restrict__ m,
                                                             const T* __restrict__ x, ot_tree<T, D> tree,
                                                             uint32_t* __restrict__ lvl_count) {
  constexpr int ML = kMaxLevels<D>;
  __shared__ uint32_t hist[ML + 1];
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x <= uint32_t(ML)) hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < nblk * uint32_t(ML + 1); k += 1024) {
    const uint32_t v = bhist[k];
    if (v) atomicAdd(&hist[k % uint32_t(ML + 1)], v);
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t b0 = 0; b0 < nblk; b0 += 1024) {
    const uint32_t b = b0 + threadIdx.x;
    const uint32_t v = b < nblk ? bsum[b] : 0u;
    uint32_t inc     = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(inc, off, 64);
      if (lane >= uint32_t(off)) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = carry_s, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const uint32_t q = wsum[w];
      if (uint32_t(w) < wave) before += q;
      total += q;
    }
    if (b < nblk) bsum[b] = before + inc - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += total;
    __syncthreads();
  }
  if (threadIdx.x <= uint32_t(ML)) lvl_count[threadIdx.x] = hist[threadIdx.x];  // [ML]: cells at the key depth (ot_build_deep_kernel)
  if (threadIdx.x == 0) {
    lvl_count[ML + 1] = 0;  // groups of the deep build
    lvl_count[ML + 5] = 0;  // its list of depth-ML cells
    lvl_count[ML + 6] = 0;  // cells left to ot_multipole_crown_kernel
    if (n < 2) {            // a single body stays in the root (src/octree.h:140-145)
      ot_node<T> r;
#pragma unroll
      for (int k = 0; k < 3; ++k) r.p[k] = T(0);
#pragma unroll
      for (int k = 0; k < D; ++k) r.p[k] = x[k];
      r.m   = m[0];
      r.lvl = 0;
      r.fc  = kOtBody;
      tree.put(0, r);
    }
  }
}

// First position in [a, b) of the sorted keys whose key satisfies a monotone predicate (b if none), probing 7 positions per round
// trip: the searches of the big cells are chains of dependent loads, ~1.5 us each.
template <typename P>
__device__ __forceinline__ uint32_t ot_first_true(const uint64_t* __restrict__ skeys, uint32_t a, uint32_t b, P pred) {
  while (b - a > 8) {
    const uint32_t step = (b - a) / 8;
    bool t[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) t[j] = pred(skeys[a + step * uint32_t(j + 1)]);
    uint32_t na = a, nb = b;
    bool found = false;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if (!found) {
        if (t[j]) {
          nb    = a + step * uint32_t(j + 1);
          found = true;
        } else {
          na = a + step * uint32_t(j + 1) + 1;
        }
      }
    }
    a = na;
    b = nb;
  }
  bool t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = a + uint32_t(j) < b ? pred(skeys[a + uint32_t(j)]) : true;
  uint32_t falses = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) falses += !t[j] && falses == uint32_t(j);
  return a + falses;
}

template <typename T, int D>
struct ot_lcp_args {
  const uint64_t* skeys;
  const uint32_t* sidx;
  uint32_t n;
  const int8_t* lv;
  const uint32_t* plocal;
  const uint32_t* bbase;
  const uint32_t* rankpos;
  const T* m;
  const T* x;
  ot_tree<T, D> tree;
  ot_cell* cells;
  ot_cell* tops;
  uint32_t* lvl_count;
  uint32_t* flags;
  uint32_t capacity, max_cells;
};

// One cell — level d, starting at sorted position i, whose left boundary shares li < d key digits — by its 2^D lanes: the end of its
// range (`e`, or 0 = not known yet), its child ranges, its child records, its entry.  false: the node pool is exhausted.
template <typename T, int D>
__device__ __forceinline__ bool ot_lcp_cell(const ot_lcp_args<T, D>& g, uint32_t i, int d, int li, uint32_t c, uint32_t e) {
  constexpr uint32_t NCH = 1u << D;
  constexpr int ML       = kMaxLevels<D>;
  const uint64_t* __restrict__ skeys = g.skeys;
  const uint32_t n = g.n;
  auto rank_at = [&](uint32_t pos, int level, int l_pos) { return g.bbase[pos / kLcpB] + g.plocal[pos] + uint32_t(level - l_pos - 1); };
  auto fits    = [&](uint32_t rank) { return rank < g.max_cells && 1u + rank * NCH + NCH <= g.capacity; };
  const uint32_t r = rank_at(i, d, li);
  if (!fits(r)) {  // its parent left it an empty leaf; reported by nbody_octree_info
    if (c == 0) atomicOr(g.flags, kFlagCapacity);
    return false;
  }
  const uint32_t fc = 1u + r * NCH;
  // The cell's end (the first position whose key leaves the d-digit prefix) and its 2^D child ranges.  Most cells hold a
  // handful of bodies: the 2^D lanes read 2^D keys per round and count, so such a cell costs ONE round trip instead of a
  // chain of ~12 dependent probes.  Cells that have not ended after kScanRounds rounds are searched.
  constexpr int kScanRounds = 4;
  const int psh       = d > 0 ? D * (ML - d) : 0;
  const uint64_t pref = d > 0 ? skeys[i] >> psh : 0;
  const int shift     = D * (ML - 1 - d);
  uint32_t lo = i, end = i;
  bool searched = e != 0;
  if (!searched) {
    uint32_t below = 0, upto = 0, size = 0;
    bool ended = false;
#pragma unroll 1
    for (int round = 0; round < kScanRounds && !ended; ++round) {
      const uint32_t pos = i + uint32_t(round) * NCH + c;
      uint32_t digit     = NCH;  // outside the cell
      if (pos < n) {
        const uint64_t k = skeys[pos];
        if (d == 0 || (k >> psh) == pref) digit = uint32_t(k >> shift) & (NCH - 1);
      }
      uint32_t inside = 0;
#pragma unroll
      for (uint32_t j = 0; j < NCH; ++j) {  // sorted keys: the lanes inside the cell are a prefix of the lane group
        const uint32_t dj = __shfl(digit, int(j), NCH);
        below += dj < c;
        upto += dj <= c;
        inside += dj < NCH;
      }
      size += inside;
      ended = inside < NCH;
    }
    if (ended) {
      e   = i + size;
      lo  = i + below;
      end = i + upto;
    } else {
      searched = true;
      e        = n;
      if (d > 0) {
        uint32_t in_e = i + kScanRounds * NCH - 1, bound = kScanRounds * NCH * 8;  // key[in_e] is inside
        while (i + bound < n && (skeys[i + bound] >> psh) == pref) {
          in_e = i + bound;
          bound *= 8;
        }
        const uint32_t out_e = i + bound < n ? i + bound : n;
        e = ot_first_true(skeys, in_e + 1, out_e, [&](uint64_t k) { return (k >> psh) != pref; });
      }
    }
  }
  if (searched) {
    lo = c == 0 ? i : ot_first_true(skeys, i, e, [&](uint64_t k) { return (uint32_t(k >> shift) & (NCH - 1)) >= c; });
    const uint32_t up = __shfl_down(lo, 1, NCH);
    end               = c + 1 < NCH ? up : e;
  }
  const uint32_t ci = fc + c;
  ot_node<T> rec;
#pragma unroll
  for (int q = 0; q < 3; ++q) rec.p[q] = T(0);
  rec.m   = T(0);
  rec.lvl = uint32_t(d) + 1u;
  rec.fc  = kOtEmpty;
  if (end - lo == 1) {
    const uint64_t b = g.sidx[lo];
#pragma unroll
    for (int q = 0; q < D; ++q) rec.p[q] = g.x[b * D + q];
    rec.m  = g.m[b];
    rec.fc = kOtBody;
  } else if (end - lo >= 2) {  // a cell of level d + 1 starting at lo: its left boundary shares l_i (lo == i) or exactly d digits
    const uint32_t rc = rank_at(lo, d + 1, lo == i ? li : d);
    if (d + 1 == ML) {         // below the key depth: ot_build_deep_kernel splits it (and sets its fc)
      if (fits(rc)) {
        const uint32_t k = atomicAdd(&g.lvl_count[ML + 5], 1u);
        g.tops[k]        = ot_cell{ci, lo, end, rc};
        g.cells[rc]      = ot_cell{ci, uint32_t(ML), rc + 1u, rc};
      } else {
        atomicOr(g.flags, kFlagCapacity);
      }
    } else if (fits(rc)) {
      rec.fc           = 1u + rc * NCH;
      g.cells[rc].node = ci;  // (the cell itself writes the other fields)
    } else {
      atomicOr(g.flags, kFlagCapacity);
    }
  }
  g.tree.put(ci, rec);
  if (c == 0) {
    g.cells[r].start = uint32_t(d);
    g.cells[r].end   = g.bbase[e / kLcpB] + g.plocal[e];  // P[e]: one past its subtree
    g.cells[r].rank  = r;
    if (d == 0) {  // the root cell: node 0 is a record of its own
      ot_node<T> root;
#pragma unroll
      for (int q = 0; q < 3; ++q) root.p[q] = T(0);
      root.m   = T(0);
      root.lvl = 0;
      root.fc  = fc;
      g.tree.put(0, root);
      g.cells[0].node = 0;
    }
  }
  return true;
}

// rank -> the sorted position its cell starts at (the level follows from the rank: d = l_i + 1 + rank - P[i]).  With this map every
// cell has lane groups of its own.  Mapping lane groups to POSITIONS instead left more than half of the waves with nothing to do
// (N = 10^6: 125 000 waves, 185 us) and made one group walk through all the cells that start at its position — position 0 starts one
// nested big cell per level from the root down, 40 us of searches in one lane group while the average wave took 6.
template <int D>
__global__ __launch_bounds__(kOB) void ot_lcp_scatter_kernel(uint32_t n, const int8_t* __restrict__ lv, const uint32_t* __restrict__ plocal,
                                                             const uint32_t* __restrict__ bbase, uint32_t* __restrict__ rankpos,
                                                             uint32_t max_cells) {
  const uint32_t i = blockIdx.x * kOB + threadIdx.x;
  if (i >= n) return;
  const int li = lv[i], ln = lv[i + 1];
  if (ln <= li) return;
  const uint32_t r0 = bbase[i / kLcpB] + plocal[i];
  for (int q = 0; q < ln - li; ++q)
    if (r0 + uint32_t(q) < max_cells) rankpos[r0 + uint32_t(q)] = i;
}

// 2^D lanes per cell, by rank
template <typename T, int D>
__global__ __launch_bounds__(kLcpBuildB) void ot_build_lcp_kernel(ot_lcp_args<T, D> g) {
  constexpr uint32_t NCH = 1u << D;
  constexpr int ML       = kMaxLevels<D>;
  uint32_t total = 0;
#pragma unroll
  for (int l = 0; l <= ML; ++l) total += g.lvl_count[l];
  if (total > g.max_cells) total = g.max_cells;
  const uint32_t r = (blockIdx.x * kLcpBuildB + threadIdx.x) / NCH, c = threadIdx.x % NCH;
  if (r >= total) return;  // (whole lane groups; whole blocks beyond the tree's cells)
  const uint32_t i = g.rankpos[r];
  const int li = g.lv[i];
  const int d  = li + 1 + int(r - (g.bbase[i / kLcpB] + g.plocal[i]));
  if (d >= ML) return;  // a cell at the key depth: its parent hands it to ot_build_deep_kernel
  ot_lcp_cell<T, D>(g, i, d, li, c, 0u);
}

// Small systems (n <= kSmallN): keys, sort, the pass over the sorted keys, the cells and the cells below the key depth in ONE
// launch by one block — at this size the step is its launches (13 of them, 0.129 ms at n = 1000, until round 4) and, inside
// one block, its dependent round trips to memory: the sorted keys and indices, l[], P[] and the rank -> position map stay in
// LDS and ot_lcp_cell reads them there (the build's chains of dependent loads become LDS latencies; positions and masses in LDS as
// well — measured — change nothing: what is left of the cells' time is one CU's instruction issue); global memory gets the sorted
// keys / indices (the walk and nbody_octree_read want them) and what the cells store.  The keys are sorted by the same
// network as the splitter sort's buckets (the order (key, position) is total, so the permutation is the stable sort's); P[] is
// the prefix over all positions at once (bbase = 0: rank_at() and the scatter see the same sums as from the blocks of
// ot_lcp_kernel); the cells are built by rank exactly as ot_build_lcp_kernel does, 2^D lanes each, the deep ones as
// ot_build_deep_kernel does, a thread each, behind a block barrier with a fence (they read what the block stored).
constexpr int kSmallB = 1024;
#ifdef NBODY_EXPERIMENTS
__device__ uint64_t ot_small_stamps[8];  // wall_clock64() at the phase boundaries of the last launch (100 MHz)
#define OT_SMALL_STAMP(k) \
  if (threadIdx.x == 0) ot_small_stamps[k] = wall_clock64()
#else
#define OT_SMALL_STAMP(k)
#endif
template <typename T, int D>
__global__ __launch_bounds__(kSmallB) void ot_insert_small_kernel(ot_lcp_args<T, D> g, const T* root, uint64_t* skeys, uint32_t* sidx,
                                                                 uint32_t* tmp) {
  constexpr uint32_t NCH = 1u << D;
  constexpr int ML       = kMaxLevels<D>;
  __shared__ uint64_t K[kSmallN];
  __shared__ uint32_t V[kSmallN];
  __shared__ uint32_t Pl[kSmallN + 1], Rp[kSmallN + 1];
  __shared__ int8_t Lv[kSmallN + 4];
  __shared__ uint32_t hist[ML + 1];
  __shared__ uint32_t wsum[kSmallB / 64];
  __shared__ uint32_t zero[kSmallN / kLcpB + 1];
  const uint32_t n = g.n, t = threadIdx.x;
  OT_SMALL_STAMP(0);
  static_assert(kSmallN == uint32_t(kSmallB), "one pair per thread");
  uint64_t key[1] = {t < n ? ot_path_key<T, D>(g.x, t, root) : ~0ull};
  uint32_t pos[1] = {t < n ? t : ~0u};
  if (t <= uint32_t(ML)) hist[t] = 0;
  if (t < kSmallN / kLcpB + 1) zero[t] = 0;
  OT_SMALL_STAMP(1);
  uint32_t P = 1;
  while (P < n) P <<= 1;
  block_sort_regs<kSmallB, 1>(key, pos, P, K, V);
  __syncthreads();  // (K / V may have carried a merge round)
  if (t < n) {
    K[t] = key[0], V[t] = pos[0];
    skeys[t] = key[0], sidx[t] = pos[0];
  }
  __syncthreads();
  OT_SMALL_STAMP(2);
  // positions 0 ... n as in ot_lcp_kernel, kSmallB at a time with the running sum carried along
  const uint32_t lane = t & 63u, wave = t >> 6;
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 <= n; b0 += kSmallB) {
    const uint32_t i = b0 + t;
    int li = -1, ln = -1;
    if (i < n) {
      const uint64_t k = K[i];
      if (i > 0) li = ot_common_levels<D>(K[i - 1], k);
      if (i + 1 < n) ln = ot_common_levels<D>(k, K[i + 1]);
    }
    if (i <= n) Lv[i] = int8_t(li);
    const uint32_t cnt = ln > li ? uint32_t(ln - li) : 0u;
    for (int d = li + 1; d <= ln; ++d) atomicAdd(&hist[d], 1u);
    uint32_t inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(inc, off, 64);
      if (lane >= uint32_t(off)) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = carry, total = 0;
#pragma unroll
    for (int w = 0; w < kSmallB / 64; ++w) {
      const uint32_t v = wsum[w];
      if (uint32_t(w) < wave) before += v;
      total += v;
    }
    if (i <= n) Pl[i] = before + inc - cnt;
    carry += total;
    __syncthreads();
  }
  if (t <= uint32_t(ML)) g.lvl_count[t] = hist[t];
  if (t == 0) {  // as ot_lcp_finish_kernel
    g.lvl_count[ML + 1] = 0;
    g.lvl_count[ML + 5] = 0;
    g.lvl_count[ML + 6] = 0;
    if (n < 2) {
      ot_node<T> r;
#pragma unroll
      for (int k = 0; k < 3; ++k) r.p[k] = T(0);
#pragma unroll
      for (int k = 0; k < D; ++k) r.p[k] = g.x[k];
      r.m   = g.m[0];
      r.lvl = 0;
      r.fc  = kOtBody;
      g.tree.put(0, r);
    }
  }
  for (uint32_t i = t; i < n; i += kSmallB) {  // as ot_lcp_scatter_kernel
    const int li = Lv[i], ln = Lv[i + 1];
    if (ln > li) {
      const uint32_t r0 = Pl[i];
      for (int q = 0; q < ln - li; ++q)
        if (r0 + uint32_t(q) < g.max_cells && r0 + uint32_t(q) <= kSmallN) Rp[r0 + uint32_t(q)] = i;
    }
  }
  __threadfence_block();  // (the deep cells' counter, zeroed above, is added to below)
  __syncthreads();
  OT_SMALL_STAMP(3);
  uint32_t total = 0;
#pragma unroll
  for (int l = 0; l <= ML; ++l) total += hist[l];
  if (total > g.max_cells) total = g.max_cells;
  ot_lcp_args<T, D> gl = g;  // the same build, its inputs read from LDS
  gl.skeys   = K;
  gl.sidx    = V;
  gl.lv      = Lv;
  gl.plocal  = Pl;
  gl.bbase   = zero;
  gl.rankpos = Rp;
  for (uint32_t r = t / NCH; r < total; r += kSmallB / NCH) {  // as ot_build_lcp_kernel
    const uint32_t i = Rp[r];
    const int li     = Lv[i];
    const int d      = li + 1 + int(r - Pl[i]);
    if (d < ML) ot_lcp_cell<T, D>(gl, i, d, li, t % NCH, 0u);
  }
  __threadfence_block();
  __syncthreads();
  OT_SMALL_STAMP(4);
  ot_build_deep_body<T, D>(t, kSmallB, sidx, tmp, g.m, g.x, root, g.tree, g.cells, g.tops, g.lvl_count, g.flags, g.capacity);
  OT_SMALL_STAMP(5);
}

template <typename T, int D>
__device__ __forceinline__ uint32_t ot_lcp_total(const uint32_t* lvl_count) {
  uint32_t total = 0;
#pragma unroll
  for (int l = 0; l <= kMaxLevels<D>; ++l) total += lvl_count[l];
  return total;
}

// A cell's monopole from its 2^D child records held in registers, children that are cells of THIS block's work taken from LDS
// instead (`slot[c]` != none).  Same order, same arithmetic as ot_multipole_cell.
constexpr uint32_t kMpNone = 0xffffffffu, kMpLater = 0x80000000u;

template <typename T, int D>
__device__ __forceinline__ void ot_multipole_from_registers(const T (&cm)[1u << D], const T (&cp)[1u << D][D], const uint32_t (&slot)[1u << D],
                                                            const T (*sm)[4], T& mass, T (&com)[D]) {
#pragma clang fp contract(off)
  constexpr uint32_t NCH = 1u << D;
  T xx[D];
  mass = T(0);
#pragma unroll
  for (int q = 0; q < D; ++q) xx[q] = T(0);
#pragma unroll
  for (uint32_t c = 0; c < NCH; ++c) {
    T mc = cm[c], pc[D];
#pragma unroll
    for (int q = 0; q < D; ++q) pc[q] = cp[c][q];
    if (slot[c] != kMpNone) {
      mc = sm[slot[c]][0];
#pragma unroll
      for (int q = 0; q < D; ++q) pc[q] = sm[slot[c]][1 + q];
    }
    mass += mc;
#pragma unroll
    for (int q = 0; q < D; ++q) xx[q] += mc * pc[q];
  }
#pragma unroll
  for (int q = 0; q < D; ++q) com[q] = xx[q] / mass;
}

// Large trees (more chunks than the crown kernel finishes out of LDS): one launch per level over all ranks.  21 launches are then
// a small part of the step, and nothing grows with the number of chunk boundaries.
template <typename T, int D>
__global__ __launch_bounds__(kOB) void ot_multipole_ranks_level_kernel(int level, ot_tree<T, D> tree, const ot_cell* __restrict__ cells,
                                                                       const uint32_t* __restrict__ lvl_count, uint32_t capacity,
                                                                       uint32_t max_cells) {
  constexpr uint32_t NCH = 1u << D;
  if (lvl_count[level] == 0) return;  // (a level the tree does not reach: the launch is all it costs)
  uint32_t total = ot_lcp_total<T, D>(lvl_count);
  if (total > max_cells) total = max_cells;
  const uint32_t r = blockIdx.x * kOB + threadIdx.x;
  if (r >= total || 1u + r * NCH + NCH > capacity) return;
  const ot_cell cl = cells[r];
  if (int(cl.start) == level) ot_multipole_cell<T, D>(tree, cl.node);
}

// A block barrier that waits for the LDS traffic only.  __syncthreads() also waits for every global store in flight (its
// fence covers all address spaces), and a store's round trip is ~2 us here: with one per level that was the whole level loop.
// Nothing in these kernels reads back what it stored to global memory.
__device__ __forceinline__ void ot_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One rank per thread.  Everything a cell needs from memory — its entry and its 2^D child records — is loaded BEFORE the level loop
// and side by side (the loop would otherwise be a chain of dependent round trips per level: 3.7 us each, measured); inside the
// loop the only values that are not there yet, the monopoles of child cells, come from LDS, where the block keeps what it has
// finished.  The cells left to the crown kernel are the ancestors of the next chunk's first cell that lie in this chunk: at
// most one per level, so they go to the fixed slot (block, level) of `later` and a bit of the block's mask — no counter.
template <typename T, int D, int CHUNK>
__device__ __forceinline__ void ot_multipole_chunk_body(uint32_t vblock, T (*sm)[4], ot_tree<T, D> tree, ot_cell* cells,
                                                        const uint32_t* lvl_count, uint32_t* later, uint32_t* later_mask,
                                                        uint32_t capacity, uint32_t max_cells) {
  constexpr uint32_t NCH = 1u << D;
  constexpr int ML       = kMaxLevels<D>;
  __shared__ int lvl_hi, lvl_lo;
  __shared__ uint32_t mask_s;
  uint32_t total = ot_lcp_total<T, D>(lvl_count);
  if (total > max_cells) total = max_cells;
  const uint32_t first = vblock * CHUNK;
  if (first >= total) return;
  const uint32_t last = first + CHUNK < total ? first + CHUNK : total;
  if (threadIdx.x == 0) {
    lvl_hi = -1;
    lvl_lo = ML;
    mask_s = 0;
  }
  ot_lds_barrier();
  const uint32_t r = first + threadIdx.x;
  const bool have  = r < last && 1u + r * NCH + NCH <= capacity;
  int level = -1;  // -1: nothing to do for this rank
  bool wait = false;
  uint32_t node = 0;
  T cm[NCH], cp[NCH][D];
  uint32_t cfc[NCH], slot[NCH];
  if (have) {
    const ot_cell cl = cells[r];
#pragma unroll
    for (uint32_t c = 0; c < NCH; ++c) {  // (does not depend on the entry: all 2^D records and the entry are in flight together)
      const ot_node<T> ch = tree.groups[r].load(c);
      cm[c]  = ch.m;
      cfc[c] = ch.fc;
#pragma unroll
      for (int q = 0; q < D; ++q) cp[c][q] = ch.p[q];
    }
    if (int(cl.start) < ML) {  // (cells at the key depth were finished by ot_build_deep_kernel)
      node  = cl.node;
      level = int(cl.start);
      wait  = cl.end > last;  // part of its subtree belongs to a later block
#pragma unroll
      for (uint32_t c = 0; c < NCH; ++c)  // a child that is a cell above the key depth is finished in this kernel: by this block,
        slot[c] = cfc[c] < kOtBody && level + 1 < ML ? cfc[c] - first : kMpNone;  // if this cell is (stored fc = group = rank)
      if (wait) {
        atomicOr(&mask_s, 1u << level);
        later[vblock * uint32_t(ML) + uint32_t(level)] = r;
        cells[r].rank = kMpLater | (vblock * uint32_t(ML) + uint32_t(level));
      }
    }
  }
  {  // the levels this block has to walk (one LDS atomic per wave: 2 x 512 same-address ones were 7 us)
    int wmax = level >= 0 && !wait ? level : -1, wmin = level >= 0 && !wait ? level : ML;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int a = __shfl_xor(wmax, off, 64), b = __shfl_xor(wmin, off, 64);
      wmax = a > wmax ? a : wmax;
      wmin = b < wmin ? b : wmin;
    }
    if ((threadIdx.x & 63u) == 0 && wmax >= 0) {
      atomicMax(&lvl_hi, wmax);
      atomicMin(&lvl_lo, wmin);
    }
  }
  ot_lds_barrier();
  const int hi = lvl_hi, lo = lvl_lo;
  if (threadIdx.x == 0) later_mask[vblock] = mask_s;
  for (int l = hi; l >= lo; --l) {
    if (level == l && !wait) {
      T mass, com[D];
      ot_multipole_from_registers<T, D>(cm, cp, slot, sm, mass, com);
      ot_node<T> pn;
#pragma unroll
      for (int q = 0; q < 3; ++q) pn.p[q] = T(0);
#pragma unroll
      for (int q = 0; q < D; ++q) pn.p[q] = com[q];
      pn.m   = mass;
      pn.lvl = uint32_t(level);
      pn.fc  = 1u + r * NCH;
      tree.put(node, pn);
      sm[threadIdx.x][0] = mass;
#pragma unroll
      for (int q = 0; q < D; ++q) sm[threadIdx.x][1 + q] = com[q];
    }
    ot_lds_barrier();
  }
}

template <typename T, int D>
__global__ __launch_bounds__(kMpChunk) void ot_multipole_chunks_kernel(ot_tree<T, D> tree, ot_cell* __restrict__ cells,
                                                                      const uint32_t* __restrict__ lvl_count, uint32_t* __restrict__ later,
                                                                      uint32_t* __restrict__ later_mask, uint32_t capacity,
                                                                      uint32_t max_cells) {
  __shared__ T sm[kMpChunk][4];
  ot_multipole_chunk_body<T, D, kMpChunk>(blockIdx.x, sm, tree, cells, lvl_count, later, later_mask, capacity, max_cells);
}

// The waiting cells of a round sit in the slots (block, level) that the blocks' masks name; in (block, level) order they are in
// RANK order, so the compacted list has the property the ranks have: the waiting cells below a waiting cell are an interval of
// it.  Exclusive prefix of the masks' bit counts into LDS (base[nblocks] = the number of waiting cells); returns that number.
constexpr uint32_t kMpMaxBlocks = 8192;  // blocks of the previous round a round can compact (32 KB of LDS): trees of up to 4.2 * 10^6 cells
constexpr uint32_t kMpLater2    = 0x40000000u;

// (written for kMpCrown threads; a larger block passes active = threadIdx.x < kMpCrown: its other threads only keep the barriers)
__device__ __forceinline__ uint32_t ot_compact_masks(const uint32_t* __restrict__ mask, uint32_t nblocks, uint32_t* base,
                                                      uint32_t* wsum, uint32_t* carry_s, bool active = true) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) *carry_s = 0;
  ot_lds_barrier();
  for (uint32_t b0 = 0; b0 < nblocks; b0 += kMpCrown) {
    const uint32_t b = b0 + threadIdx.x;
    const uint32_t v = active && b < nblocks ? uint32_t(__builtin_popcount(mask[b])) : 0u;
    uint32_t inc     = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(inc, off, 64);
      if (lane >= uint32_t(off)) inc += up;
    }
    if (lane == 63 && active) wsum[wave] = inc;
    ot_lds_barrier();
    uint32_t before = *carry_s, total = 0;
#pragma unroll
    for (int w = 0; w < kMpCrown / 64; ++w) {
      const uint32_t q = wsum[w];
      if (uint32_t(w) < wave) before += q;
      total += q;
    }
    if (active && b < nblocks) base[b] = before + inc - v;
    ot_lds_barrier();
    if (threadIdx.x == 0) *carry_s += total;
    ot_lds_barrier();
  }
  if (threadIdx.x == 0) base[nblocks] = *carry_s;
  ot_lds_barrier();
  return base[nblocks];
}

// the rank in compact position k: the block whose cells include it (last b with base[b] <= k), then the (k - base[b] + 1)-th set
// bit of that block's mask names the level slot
template <int ML>
__device__ __forceinline__ uint32_t ot_compact_item(uint32_t k, const uint32_t* base, uint32_t nblocks,
                                                     const uint32_t* __restrict__ mask, const uint32_t* __restrict__ later) {
  uint32_t b = 0, hi_b = nblocks;
  while (hi_b - b > 1) {
    const uint32_t mid = b + (hi_b - b) / 2;
    if (base[mid] <= k) b = mid;
    else hi_b = mid;
  }
  uint32_t m = mask[b];
  for (uint32_t skip = k - base[b]; skip > 0; --skip) m &= m - 1u;
  return later[b * uint32_t(ML) + uint32_t(__builtin_ctz(m))];
}
__device__ __forceinline__ uint32_t ot_compact_slot(uint32_t slot, int ML, const uint32_t* base, const uint32_t* __restrict__ mask) {
  const uint32_t b = slot / uint32_t(ML), bit = slot % uint32_t(ML);
  return base[b] + uint32_t(__builtin_popcount(mask[b] & ((1u << bit) - 1u)));
}

// Second round, for trees of more than kMpCrown chunks: the cells the chunks left are too many for one block (~5 per chunk
// boundary: 2 500 at N = 10^6), so they get the chunks' treatment once more — blocks of kMpCrown consecutive waiting cells finish
// those whose waiting descendants all sit in the block, from registers and LDS, and leave the ancestors of their last boundary,
// one per level at most, to ot_multipole_crown_kernel.  A waiting cell's waiting descendants end at base[subtree end / chunk]: the
// chunk in which its subtree ends holds no waiting cell of that subtree (it would have to span that chunk's end, which the
// subtree does not).
template <typename T, int D>
__global__ __launch_bounds__(kMpCrown) void ot_multipole_round_kernel(ot_tree<T, D> tree, ot_cell* __restrict__ cells,
                                                                  uint32_t* __restrict__ lvl_count,
                                                                  const uint32_t* __restrict__ later0,
                                                                  const uint32_t* __restrict__ mask0, uint32_t* __restrict__ later1,
                                                                  uint32_t* __restrict__ mask1, uint32_t max_cells) {
  constexpr uint32_t NCH = 1u << D;
  constexpr int ML       = kMaxLevels<D>;
  __shared__ T sm[kMpCrown][4];
  __shared__ uint32_t base[kMpMaxBlocks + 1];
  __shared__ uint32_t wsum[kMpCrown / 64];
  __shared__ uint32_t carry_s, mask_s;
  __shared__ int lvl_hi, lvl_lo;
  uint32_t total = ot_lcp_total<T, D>(lvl_count);
  if (total > max_cells) total = max_cells;
  const uint32_t nchunks = (total + kMpChunk - 1) / kMpChunk;
  if (nchunks <= 1 || nchunks > kMpMaxBlocks) {  // nothing is waiting / (not launched for such trees)
    if (blockIdx.x == 0 && threadIdx.x == 0) lvl_count[ML + 6] = 0;
    return;
  }
  const uint32_t count   = ot_compact_masks(mask0, nchunks, base, wsum, &carry_s);
  const uint32_t nblocks = (count + kMpCrown - 1) / kMpCrown;
  if (blockIdx.x == 0 && threadIdx.x == 0) lvl_count[ML + 6] = nblocks;  // what ot_multipole_crown_kernel compacts
  if (blockIdx.x >= nblocks) return;
  const uint32_t k0 = blockIdx.x * kMpCrown, k1 = k0 + kMpCrown < count ? k0 + kMpCrown : count;
  if (threadIdx.x == 0) {
    lvl_hi = -1;
    lvl_lo = ML;
    mask_s = 0;
  }
  ot_lds_barrier();
  const uint32_t k = k0 + threadIdx.x;
  int level = -1;
  bool wait = false;
  uint32_t node = 0, r = 0;
  T cm[NCH], cp[NCH][D];
  uint32_t slot[NCH];
  if (k < k1) {
    r                = ot_compact_item<ML>(k, base, nchunks, mask0, later0);
    const ot_cell cl = cells[r];
    uint32_t crank[NCH];
#pragma unroll
    for (uint32_t c = 0; c < NCH; ++c) {
      const ot_node<T> ch = tree.groups[r].load(c);
      cm[c] = ch.m;
#pragma unroll
      for (int q = 0; q < D; ++q) cp[c][q] = ch.p[q];
      crank[c] = ch.fc < kOtBody && int(cl.start) + 1 < ML ? ch.fc : kMpNone;
    }
    node  = cl.node;
    level = int(cl.start);
    const uint32_t endchunk = cl.end / kMpChunk < nchunks ? cl.end / kMpChunk : nchunks;
    wait = base[endchunk] > k1;  // some of its waiting descendants belong to a later block
    if (wait) {
      atomicOr(&mask_s, 1u << level);
      later1[blockIdx.x * uint32_t(ML) + uint32_t(level)] = r;
      cells[r].rank = kMpLater | kMpLater2 | (blockIdx.x * uint32_t(ML) + uint32_t(level));
    } else {
#pragma unroll
      for (uint32_t c = 0; c < NCH; ++c) {  // a child cell is either finished (by a chunk) or waiting in this block too
        slot[c] = kMpNone;
        if (crank[c] != kMpNone) {
          const uint32_t mark = cells[crank[c]].rank;  // (this block alone may change the marks of this cell's descendants)
          if (mark & kMpLater) slot[c] = ot_compact_slot(mark & ~(kMpLater | kMpLater2), ML, base, mask0) - k0;
        }
      }
    }
  }
  {
    int wmax = level >= 0 && !wait ? level : -1, wmin = level >= 0 && !wait ? level : ML;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int a = __shfl_xor(wmax, off, 64), b = __shfl_xor(wmin, off, 64);
      wmax = a > wmax ? a : wmax;
      wmin = b < wmin ? b : wmin;
    }
    if ((threadIdx.x & 63u) == 0 && wmax >= 0) {
      atomicMax(&lvl_hi, wmax);
      atomicMin(&lvl_lo, wmin);
    }
  }
  ot_lds_barrier();
  const int hi = lvl_hi, lo = lvl_lo;
  if (threadIdx.x == 0) mask1[blockIdx.x] = mask_s;
  for (int l = hi; l >= lo; --l) {
    if (level == l && !wait) {
      T mass, com[D];
      ot_multipole_from_registers<T, D>(cm, cp, slot, sm, mass, com);
      ot_node<T> pn;
#pragma unroll
      for (int q = 0; q < 3; ++q) pn.p[q] = T(0);
#pragma unroll
      for (int q = 0; q < D; ++q) pn.p[q] = com[q];
      pn.m   = mass;
      pn.lvl = uint32_t(level);
      pn.fc  = 1u + r * NCH;
      tree.put(node, pn);
      sm[threadIdx.x][0] = mass;
#pragma unroll
      for (int q = 0; q < D; ++q) sm[threadIdx.x][1 + q] = com[q];
    }
    ot_lds_barrier();
  }
}

// One block finishes what is left: the waiting cells of the chunks (`nprev` < 0: as many blocks as the tree has chunks; pending
// children carry kMpLater) or those of ot_multipole_round_kernel (`nprev` >= 0: lvl_count[ML + 6] blocks; pending children carry
// kMpLater2), deepest level first.  Up to kMpCrown of them: compacted, one per thread, from registers and LDS as above.  More (a
// pathological tree): level by level, children read from memory.
template <typename T, int D>
__device__ __forceinline__ void ot_multipole_crown_body(T (*sm)[4], ot_tree<T, D> tree, const ot_cell* cells, const uint32_t* lvl_count,
                                                        const uint32_t* later, const uint32_t* later_mask, uint32_t max_cells,
                                                        int after_round, bool active = true) {
  constexpr uint32_t NCH = 1u << D;
  constexpr int ML       = kMaxLevels<D>;
  __shared__ uint32_t base[kMpMaxBlocks + 1];
  __shared__ uint32_t wsum[kMpCrown / 64];
  __shared__ uint32_t carry_s;
  __shared__ int lvl_hi;
  uint32_t total = ot_lcp_total<T, D>(lvl_count);
  if (total > max_cells) total = max_cells;
  const uint32_t nblocks  = after_round ? lvl_count[ML + 6] : (total + kMpChunk - 1) / kMpChunk;
  const uint32_t pending  = after_round ? kMpLater2 : kMpLater;
  if (nblocks <= 1 || nblocks > kMpMaxBlocks) return;  // (a single block leaves nothing)
  const uint32_t count = ot_compact_masks(later_mask, nblocks, base, wsum, &carry_s, active);
  if (threadIdx.x == 0) lvl_hi = -1;
  ot_lds_barrier();
  if (count <= uint32_t(kMpCrown)) {
    int level = -1;
    uint32_t node = 0, r = 0;
    T cm[NCH], cp[NCH][D];
    uint32_t slot[NCH];
    if (active && threadIdx.x < count) {
      r                = ot_compact_item<ML>(threadIdx.x, base, nblocks, later_mask, later);
      const ot_cell cl = cells[r];
      node             = cl.node;
      level            = int(cl.start);
      uint32_t crank[NCH];
#pragma unroll
      for (uint32_t c = 0; c < NCH; ++c) {
        const ot_node<T> ch = tree.groups[r].load(c);
        cm[c] = ch.m;
#pragma unroll
        for (int q = 0; q < D; ++q) cp[c][q] = ch.p[q];
        crank[c] = ch.fc < kOtBody && level + 1 < ML ? ch.fc : kMpNone;
      }
#pragma unroll
      for (uint32_t c = 0; c < NCH; ++c) {  // a child cell is either finished already or waiting here too
        slot[c] = kMpNone;
        if (crank[c] != kMpNone) {
          const uint32_t mark = cells[crank[c]].rank;
          if (mark & pending) slot[c] = ot_compact_slot(mark & ~(kMpLater | kMpLater2), ML, base, later_mask);
        }
      }
      atomicMax(&lvl_hi, level);
    }
    ot_lds_barrier();
    for (int l = lvl_hi; l >= 0; --l) {
      if (level == l) {
        T mass, com[D];
        ot_multipole_from_registers<T, D>(cm, cp, slot, sm, mass, com);
        ot_node<T> pn;
#pragma unroll
        for (int q = 0; q < 3; ++q) pn.p[q] = T(0);
#pragma unroll
        for (int q = 0; q < D; ++q) pn.p[q] = com[q];
        pn.m   = mass;
        pn.lvl = uint32_t(level);
        pn.fc  = 1u + r * NCH;
        tree.put(node, pn);
        sm[threadIdx.x][0] = mass;
#pragma unroll
        for (int q = 0; q < D; ++q) sm[threadIdx.x][1 + q] = com[q];
      }
      ot_lds_barrier();
    }
    return;
  }
  // more waiting cells than threads: each level's straight from the slots
  for (uint32_t b = active ? threadIdx.x : nblocks; b < nblocks; b += kMpCrown) {
    const uint32_t m = later_mask[b];
    if (m) atomicMax(&lvl_hi, 31 - __builtin_clz(m));
  }
  __syncthreads();
  for (int l = lvl_hi; l >= 0; --l) {
    for (uint32_t b = active ? threadIdx.x : nblocks; b < nblocks; b += kMpCrown)
      if ((later_mask[b] >> l) & 1u) ot_multipole_cell<T, D>(tree, cells[later[b * uint32_t(ML) + uint32_t(l)]].node);
    __threadfence_block();
    __syncthreads();  // (waits for the stores: the next level reads them back)
  }
}

template <typename T, int D>
__global__ __launch_bounds__(kMpCrown) void ot_multipole_crown_kernel(ot_tree<T, D> tree, const ot_cell* __restrict__ cells,
                                                                  const uint32_t* __restrict__ lvl_count,
                                                                  const uint32_t* __restrict__ later,
                                                                  const uint32_t* __restrict__ later_mask, uint32_t max_cells,
                                                                  int after_round) {
  __shared__ T sm[kMpCrown][4];
  ot_multipole_crown_body<T, D>(sm, tree, cells, lvl_count, later, later_mask, max_cells, after_round);
}

// Small systems (at most kSmallChunks chunks of ranks) in one launch by one block: the chunks one after the other and the crown —
// or, in float, where a cell's 2^D child records fit the registers of 1024 threads, and the tree has at most 1024
// cells, ONE chunk of 1024 ranks and nothing left for a crown.  The crown reads what the chunks stored (records, marks, slots): a fence and a
// full barrier in between.
constexpr uint32_t kSmallChunks = (kSmallN + 1 + kMpChunk - 1) / kMpChunk;
static_assert(kMpChunk == kMpCrown, "the small-tree kernel runs both bodies with one block shape");
template <typename T, int D>
constexpr int kSmallMpThreads = sizeof(T) == 4 ? 1024 : kMpChunk;  // (double: 3D does not fit the registers, 2D not the 64 KB of static LDS)
template <typename T, int D>
__global__ __launch_bounds__((kSmallMpThreads<T, D>)) void ot_multipole_small_kernel(ot_tree<T, D> tree, ot_cell* cells, const uint32_t* lvl_count,
                                                                                  uint32_t* later, uint32_t* later_mask, uint32_t capacity,
                                                                                  uint32_t max_cells, uint32_t max_chunks) {
  constexpr int NT = kSmallMpThreads<T, D>;
  __shared__ T sm[NT][4];
  uint32_t total = ot_lcp_total<T, D>(lvl_count);
  if (total > max_cells) total = max_cells;
  if (NT > kMpChunk && total <= uint32_t(NT)) {
    ot_multipole_chunk_body<T, D, NT>(0, sm, tree, cells, lvl_count, later, later_mask, capacity, max_cells);
    return;
  }
  // The chunks of kMpChunk ranks and the crown are written for kMpChunk threads.  In float this block has 1024: the upper half
  // stays through every barrier and does nothing in between (a chunk gives ranks past its end no cell; the crown is told).
  const bool active     = threadIdx.x < uint32_t(kMpChunk);
  const uint32_t chunks = (total + kMpChunk - 1) / kMpChunk;
  for (uint32_t b = 0; b < chunks && b < max_chunks; ++b) {
    ot_multipole_chunk_body<T, D, kMpChunk>(b, sm, tree, cells, lvl_count, later, later_mask, capacity, max_cells);
    if (chunks > 1) {
      __threadfence_block();
      __syncthreads();
    }
  }
  if (chunks > 1) ot_multipole_crown_body<T, D>(sm, tree, cells, lvl_count, later, later_mask, max_cells, 0, active);
}

// ---- traversal (src/octree.h:226-263) -----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ot_xcd_contiguous_block(uint32_t b, uint32_t nblocks) {
  const uint32_t q = nblocks / 8u, r = nblocks % 8u, xcd = b % 8u, slot = b / 8u;
  return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + slot;
}

template <typename T>
__device__ __forceinline__ T ot_sqrt(T v) {
  if constexpr (sizeof(T) == 4) return __builtin_sqrtf(v);  // correctly rounded (HIP default for f32 sqrt/div)
  else return __builtin_sqrt(v);
}
template <typename T>
__device__ __forceinline__ T ot_ldexp(T v, int e) {
  if constexpr (sizeof(T) == 4) return __builtin_ldexpf(v, e);
  else return __builtin_ldexp(v, e);
}
template <typename T>
__device__ __forceinline__ T ot_rsq(T v) {  // hardware seed: 1 ulp (f32) / ~2^-24 (f64, profiles/r01_valu_rates_microbench.txt)
  if constexpr (sizeof(T) == 4) return __builtin_amdgcn_rsqf(v);
  else return __builtin_amdgcn_rsq(v);
}
template <typename T>
__device__ __forceinline__ T ot_recip(T d) {  // 1/d to ~2e-16 relative (seed + one Newton step)
  if constexpr (sizeof(T) == 4) {
    const float z0 = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(z0, __builtin_fmaf(-d, z0, 1.0f), z0);
  } else {
    const double z0 = __builtin_amdgcn_rcp(d);
    return __builtin_fma(z0, __builtin_fma(-d, z0, 1.0), z0);
  }
}

template <typename T>
struct ot_consts {
  static constexpr T eps  = sizeof(T) == 4 ? T(FLT_EPSILON) : T(DBL_EPSILON);
  static constexpr T tiny = sizeof(T) == 4 ? T(1e-30f) : T(1e-300);  // sqrt(tiny) + eps == eps: coincident points unchanged
};
template <typename T>
struct ot_theta {  // theta and the two guard values of the quick opening test
  T exact, lo, hi;
  __device__ explicit ot_theta(T theta) : exact(theta), lo(theta * (T(1) - T(1) / T(65536))), hi(theta * (T(1) + T(1) / T(65536))) {}
};

// dist2(x, xj) summed exactly as the reference does (src/vec.h:232-241): only the guard-band evaluation needs its bits.
template <typename T, int D>
__device__ __forceinline__ T ot_dist2_exact(const T (&di)[D]) {
#pragma clang fp contract(off)
  T d2 = T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) d2 = d2 + di[k] * di[k];
  return d2;
}
// The same sum as one FMA chain started at `tiny` (differs from the exact one by ~1e-16 relative; a coincident point
// gives `tiny`, whose square root vanishes against eps): feeds the quick test and the accepted term.
template <typename T, int D>
__device__ __forceinline__ T ot_dist2_fused(const T (&di)[D]) {
  T d2 = ot_consts<T>::tiny;
#pragma unroll
  for (int k = 0; k < D; ++k) d2 = __builtin_elementwise_fma(di[k], di[k], d2);
  return d2;
}

// Opening test `side / (sqrt(d2) + eps) < theta` (src/octree.h:243, src/vec.h:243-246).  The reference's decision is
// reproduced bit-for-bit: q = side * rsq(d2) brackets the exact quotient to ~2^-23, so a lane whose q is outside
// theta*(1 -+ 2^-16) is decided by it; if any lane that `need`s a decision is inside the band (or holds a NaN/inf), the
// wave evaluates the reference's expression with IEEE sqrt and divide.  y0 = rsq(ot_dist2_fused(dj)), dj = xj - x (the
// exact sum uses the squares only, so the sign convention does not matter to it).
template <typename T, int D>
__device__ __forceinline__ bool ot_accept(bool need, T side, const T (&dj)[D], T y0, const ot_theta<T>& th) {
  const T q = side * y0;  // >= side / (sqrt(d2) + eps), up to the seed error
  bool sure_take = q < th.lo, sure_open;
  if constexpr (sizeof(T) == 8) {
    // the quotient is also >= q * (1 - eps * y0): with y0 < 2^35 that factor differs from 1 by < 2^-17, inside the band
    sure_open = q > th.hi && y0 < T(0x1p35);
  } else {
    const T ql = __builtin_elementwise_fma(-q, ot_consts<T>::eps * y0, q);  // 1/(1 + eps/s) >= 1 - eps/s
    sure_open  = ql > th.hi;
  }
  bool take = sure_take;
  if (__ballot(need && !sure_take && !sure_open) != 0ull) {
#pragma clang fp contract(off)
    take = side / (ot_sqrt(ot_dist2_exact<T, D>(dj)) + ot_consts<T>::eps) < th.exact;
  }
  return take;
}

// a += mj * (xj - x) / dx^3, dx = sqrt(d2) + eps (src/octree.h:240-241), for the lanes in `on` (tolerance parity).
// dj = xj - x, d2f = ot_dist2_fused(dj), y0 = rsq(d2f) — the seed the opening test already paid for.
//   far  (d2 >= 2^-46 in f64, 2^-18 in f32): no square root, no reciprocal.  With s = sqrt(d2):
//        1/(s + eps)^3 = s^-3 * (1 + eps/s)^-3 = s^-3 * (1 - 3 eps/s + 6 (eps/s)^2 - ...), and eps/s <= 2^-29 (2^-14) there, so
//        the dropped 6 (eps/s)^2 is below half an ulp.  f64: s^-3 from the 2^-24 seed by the third-order step on the cube
//        K1 uses (pair_math<double>::weight_far), both corrections in one FMA: w = (mj*y^3)(1 + e(3/2 + 15/8 e) - 3 eps y):
//        8 full-rate ops.  f32: the 1-ulp seed cubed, 5 ops.  (Round 1: polished sqrt, cube, polished v_rcp: 12 ops + a
//        quarter-rate transcendental.)
//   near (anything closer — the body's own leaf, coincident or very close bodies; found with one compare on the high word
//        and one wave-uniform branch): the reference's expression from a polished sqrt and reciprocal, kept per lane.
// The body's own leaf and empty leaves add exactly 0 (dj == 0 or mj == 0): dx^3 >= eps^3 keeps the near weight finite.
template <typename T>
struct ot_near {
  static constexpr uint32_t bits = sizeof(T) == 8 ? 0x3D100000u : 0x36800000u;  // high word of 2^-46 / bits of 2^-18
  __device__ static __forceinline__ uint32_t of(T d2) {
    if constexpr (sizeof(T) == 8) return uint32_t(__builtin_bit_cast(unsigned long long, d2) >> 32);
    else return __builtin_bit_cast(uint32_t, d2);
  }
};

template <typename T, int D>
__device__ __forceinline__ void ot_accumulate(bool on, uint64_t on_mask, T (&acc)[D], const T (&dj)[D], T mj, T d2f, T y0,
                                              const pair_consts<T>& pc) {
  T w;
  if constexpr (sizeof(T) == 8) {
    const T a  = y0 * y0;
    const T e  = __builtin_elementwise_fma(-d2f, a, T(1));
    const T y3 = a * y0;
    const T p  = __builtin_elementwise_fma(e, pc.k1875, pc.k15);
    const T g  = __builtin_elementwise_fma(p, e, -(T(3) * ot_consts<T>::eps) * y0);
    const T my = mj * y3;
    w          = __builtin_elementwise_fma(my, g, my);
  } else {
    (void)pc;
    const T my = mj * ((y0 * y0) * y0);
    w          = __builtin_elementwise_fma(my, -(T(3) * ot_consts<T>::eps) * y0, my);
  }
  const bool close = ot_near<T>::of(d2f) < ot_near<T>::bits;
  if (__builtin_expect((__builtin_amdgcn_ballot_w64(close) & on_mask) != 0ull, 0)) {
#pragma clang fp contract(off)
    const T t  = d2f * y0;
    const T e  = __builtin_elementwise_fma(-t, y0, T(1));
    const T sq = __builtin_elementwise_fma(T(0.5) * t, e, t);
    const T dx = sq + ot_consts<T>::eps;
    const T wn = mj * ot_recip((dx * dx) * dx);
    w          = close ? wn : w;
  }
  w = on ? w : T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) acc[k] = __builtin_elementwise_fma(w, dj[k], acc[k]);
}

// Shard windows.  The walk takes bodies in key order, the window [first, first + count) is a range of BODY indices, so the
// owned bodies are scattered over the key order: they are compacted (order kept) into a dense list first, otherwise every
// wave would carry one owned body among 2^(6-D) and a 1/G shard would cost as much as the whole system.
constexpr int kOC = 1024;
__global__ __launch_bounds__(kOC) void ot_owned_count_kernel(const uint32_t* __restrict__ sidx, uint32_t sz, uint32_t first,
                                                             uint32_t count, uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wsum[kOC / 64];
  const uint32_t t = blockIdx.x * kOC + threadIdx.x;
  const bool own   = t < sz && sidx[t] - first < count;  // unsigned: also false for sidx[t] < first
  const uint64_t b = __ballot(own);
  if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = uint32_t(__builtin_popcountll(b));
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int w = 0; w < kOC / 64; ++w) tot += wsum[w];
    block_counts[blockIdx.x] = tot;
  }
}
__global__ __launch_bounds__(kOC) void ot_owned_scan_kernel(uint32_t* __restrict__ block_counts, uint32_t nblocks) {
  __shared__ uint32_t buf[kOC];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nblocks; base += kOC) {  // exclusive scan, one chunk of 1024 block counts at a time
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nblocks ? block_counts[i] : 0u;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1; off < kOC; off <<= 1) {
      const uint32_t add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0u;
      __syncthreads();
      buf[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nblocks) block_counts[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == kOC - 1) carry += buf[kOC - 1];
    __syncthreads();
  }
}
__global__ __launch_bounds__(kOC) void ot_owned_scatter_kernel(const uint32_t* __restrict__ sidx, uint32_t sz, uint32_t first,
                                                               uint32_t count, const uint32_t* __restrict__ block_offsets,
                                                               uint32_t* __restrict__ owned) {
  __shared__ uint32_t wfirst[kOC / 64];
  const uint32_t t    = blockIdx.x * kOC + threadIdx.x;
  const uint32_t body = t < sz ? sidx[t] : 0u;
  const bool own      = t < sz && body - first < count;
  const uint64_t b    = __ballot(own);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) wfirst[wave] = uint32_t(__builtin_popcountll(b));
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int w = 0; w < kOC / 64; ++w) {
      const uint32_t n_w = wfirst[w];
      wfirst[w]          = run;
      run += n_w;
    }
  }
  __syncthreads();
  if (own) owned[block_offsets[blockIdx.x] + wfirst[wave] + uint32_t(__builtin_popcountll(b & ((1ull << lane) - 1ull)))] = body;
}

// One body per 2^D lanes.  When a node is opened its 2^D children are examined side by side, one per lane (their records
// are contiguous), the ones to open are pushed on the body's stack in LDS in reverse child order and the walk continues
// with the popped one — the order of the reference's walk.  The dependent chain of a body is the number of nodes it OPENS
// (~180 at N = 10^6, theta 0.5), not the number it visits (~1430).  Each lane sums the terms of its own child slots and
// the 2^D partial sums are combined at the end in a fixed order: the summation order differs from the reference's
// (tolerance parity), the set of tests, accepted terms and therefore the counters do not, and the result of a body
// depends on nothing but the tree and that body (so it is independent of the shard window).
template <typename T, int D, bool COUNT>
__global__ __launch_bounds__(64) void ot_force_kernel(const ot_node<T>* __restrict__ rootrec, const ot_group<T, D>* __restrict__ groups,
                                                      const uint32_t* __restrict__ list,
                                                      uint32_t nlist, const T* __restrict__ x, T* __restrict__ a, T c,
                                                      uint32_t first, T theta, uint32_t capacity, const T* __restrict__ root,
                                                      uint32_t* __restrict__ flags, uint32_t* __restrict__ counters) {
  constexpr uint32_t NCH   = 1u << D;
  constexpr uint32_t GPW   = 64u / NCH;                         // bodies per wave
  constexpr uint32_t DEPTH = (NCH - 1u) * kMaxLevels<D> + NCH;  // a pop frees one slot, an open adds <= 2^D
  __shared__ uint32_t stack[GPW][DEPTH];
  const uint32_t g = threadIdx.x / NCH, cc = threadIdx.x % NCH;
  // `list`: the owned bodies in key order (neighbours share most of their walk: cache); XCD-contiguous blocks
  const uint32_t t    = ot_xcd_contiguous_block(blockIdx.x, gridDim.x) * GPW + g;
  const bool valid    = t < nlist;
  const uint32_t body = valid ? list[t] : first;
  const ot_theta<T> th(theta);
  const pair_consts<T> pc;
  const T root_side = root[D];
  T xi[D], acc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    xi[k]  = valid ? x[uint64_t(body) * D + k] : T(0);
    acc[k] = T(0);
  }
  uint32_t c_nodes = 0, c_terms = 0;
  uint32_t cur = 0, sp = 0;
  bool more = false;
  if (valid) {  // the root is examined alone (by every lane of the group; lane 0 keeps the result)
    const ot_node<T> nd = *rootrec;
    T di[D];
#pragma unroll
    for (int k = 0; k < D; ++k) di[k] = nd.p[k] - xi[k];
    const T d2f     = ot_dist2_fused<T, D>(di);
    const T y0      = ot_rsq(d2f);
    const bool leaf = nd.fc >= kOtBody;  // kOtBody or kOtEmpty
    const bool take = leaf || ot_accept<T, D>(!leaf, root_side, di, y0, th);
    {
      const bool on0 = take && cc == 0;  // the root is examined by every lane of the group; lane 0 keeps the result
      const uint64_t m0 = __builtin_amdgcn_ballot_w64(on0);
      if (m0 != 0ull) ot_accumulate<T, D>(on0, m0, acc, di, nd.m, d2f, y0, pc);
    }
    if (COUNT && cc == 0) {
      c_nodes = 1;
      c_terms = take;
    }
    more = !take;
    cur  = nd.fc;  // the root's stored fc is already a sibling-group number, like every fl[][0] below
  }
  uint32_t guard = capacity;  // a well-formed tree is left after < capacity steps; never spin on a damaged one
  while (more && guard-- != 0u) {  // the lanes of a group leave together
    const ot_node<T> nd = groups[cur].load(cc);  // this lane's child: two or three 16-/8-byte loads
    T di[D];
#pragma unroll
    for (int k = 0; k < D; ++k) di[k] = nd.p[k] - xi[k];
    const T d2f     = ot_dist2_fused<T, D>(di);
    const T y0      = ot_rsq(d2f);
    const bool leaf = nd.fc >= kOtBody;
    const bool take = leaf || ot_accept<T, D>(!leaf, ot_ldexp(root_side, -int(nd.lvl)), di, y0, th);
    if (COUNT) {
      ++c_nodes;
      c_terms += take;
    }
    const uint64_t take_mask = __builtin_amdgcn_ballot_w64(take);
    if (take_mask != 0ull) ot_accumulate<T, D>(take, take_mask, acc, di, nd.m, d2f, y0, pc);
    const uint32_t open_mask = uint32_t((__ballot(!take) >> (g * NCH)) & ((1ull << NCH) - 1ull));
    if (sp + uint32_t(__builtin_popcount(open_mask)) > DEPTH) {  // only below the key depth can a walk hold this many
      if (cc == 0) atomicOr(flags, kFlagStack);                  // pending nodes; reported by nbody_octree_info
      break;
    }
    if (!take) stack[g][sp + uint32_t(__builtin_popcount(open_mask >> (cc + 1u)))] = nd.fc;  // reverse child order
    sp += uint32_t(__builtin_popcount(open_mask));
    if (sp == 0u) break;
    __builtin_amdgcn_wave_barrier();  // one lane pushed, all lanes of the group pop: keep the LDS write before the read
    cur = stack[g][--sp];
    __builtin_amdgcn_wave_barrier();  // ... and this read before the next round's push into the same slot
  }
  if (more && guard == 0xffffffffu && cc == 0) atomicOr(flags, kFlagWalk);  // step budget spent: the tree is damaged
  // combine the 2^D partial sums of a body (fixed order)
#pragma unroll
  for (uint32_t off = NCH / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < D; ++k) acc[k] += __shfl_xor(acc[k], int(off), 64);
    if (COUNT) {
      c_nodes += __shfl_xor(c_nodes, int(off), 64);
      c_terms += __shfl_xor(c_terms, int(off), 64);
    }
  }
  if (valid && cc == 0) {
#pragma unroll
    for (int k = 0; k < D; ++k) a[uint64_t(body - first) * D + k] = c * acc[k];
    if (COUNT) {
      counters[uint64_t(body) * 2 + 0] = c_nodes;
      counters[uint64_t(body) * 2 + 1] = c_terms;
    }
  }
}

// the accepted term's weight for d2 >= 2^-46 (ot_accumulate's far form, same operations in the same order); v[60:61] = mass
#define OT_FAR                                                                                                            \
  "v_mul_f64 %[y2], %[y], %[y]\n\t"                                                                                       \
  "v_fma_f64 %[e], -%[r2], %[y2], 1.0\n\t"                                                                                \
  "v_mul_f64 %[y2], %[y], %[y2]\n\t"                                                                                      \
  "v_fma_f64 %[p], %[k1875], %[e], %[k15]\n\t"                                                                            \
  "v_mul_f64 %[g], %[y], %[m3eps]\n\t"                                                                                    \
  "v_fmac_f64_e32 %[g], %[p], %[e]\n\t"                                                                                   \
  "v_mul_f64 %[w], v[60:61], %[y2]\n\t"                                                                                   \
  "v_fmac_f64_e32 %[w], %[w], %[g]\n\t"

// One visit round of ot_force_kernel written out as ISA (double, 3D): see ot_force_isa_kernel.
#define OT_KEEP(...) __VA_ARGS__
#define OT_DROP(...) ""
// record loads of a round: 3D (p0,p1) | (p2,m) | (fc,depth) from a 320-byte group; 2D (p0,p1) | m | (fc,depth) from a 128-byte one
#define OT_LOADS_F64_D3                                                                                                   \
  "v_mad_u32_u24 %[oa], %[cur], %[gsz], %[cc16]\n\t"                                                                      \
  "v_mad_u32_u24 %[of], %[cur], %[gsz], %[cc8]\n\t"                                                                       \
  "global_load_dwordx4 v[54:57], %[oa], %[groups]\n\t"                                                                    \
  "global_load_dwordx4 v[58:61], %[oa], %[groups] offset:128\n\t"                                                         \
  "global_load_dwordx2 v[62:63], %[of], %[groups]\n\t"
#define OT_LOADS_F64_D2                                                                                                   \
  "v_mad_u32_u24 %[oa], %[cur], %[gsz], %[cc16]\n\t"                                                                      \
  "v_mad_u32_u24 %[of], %[cur], %[gsz], %[cc8]\n\t"                                                                       \
  "global_load_dwordx4 v[54:57], %[oa], %[groups]\n\t"                                                                    \
  "global_load_dwordx2 v[60:61], %[of], %[groups] offset:-32\n\t"                                                         \
  "global_load_dwordx2 v[62:63], %[of], %[groups]\n\t"
#define OT_ISA_TEXT(Z, LOADS, MASK, SSH, CNT_N, CNT_T)                                                                                      \
  "s_mov_b64 %[sv], exec\n\t"                                                                                             \
  "v_cmp_ne_u32_e32 vcc, 0, %[more]\n\t"                                                                                  \
  "s_and_b64 exec, exec, vcc\n\t"                                                                                         \
  "s_cbranch_execz .LOTend%=\n"                                                                                           \
  ".LOTtop%=:\n\t"                                                                                                        \
  LOADS                                                                                                                   \
  CNT_N                                                                                                                   \
  "s_waitcnt vmcnt(2)\n\t"                                                                                                \
  "v_add_f64 %[d0], v[54:55], -%[xi0]\n\t"                                                                                \
  "v_add_f64 %[d1], v[56:57], -%[xi1]\n\t"                                                                                \
  "v_fma_f64 %[r2], %[d0], %[d0], %[tiny]\n\t"                                                                            \
  Z("s_waitcnt vmcnt(1)\n\t"                                                                                              \
    "v_add_f64 %[d2], v[58:59], -%[xi2]\n\t")                                                                             \
  "v_fmac_f64_e32 %[r2], %[d1], %[d1]\n\t"                                                                                \
  Z("v_fmac_f64_e32 %[r2], %[d2], %[d2]\n\t")                                                                             \
  "v_rsq_f64_e32 %[y], %[r2]\n\t"                                                                                         \
  "s_waitcnt vmcnt(0)\n\t"                                                                                                \
  "v_sub_u32_e32 %[t], 0, v63\n\t"                                                                                        \
  "v_ldexp_f64 %[g], %[rootside], %[t]\n\t"                                                                               \
  "v_cmp_gt_u32_e64 %[nonleaf], -2, v62\n\t"                                                                              \
  "v_mul_f64 %[w], %[g], %[y]\n\t"                                                                                        \
  "v_cmp_lt_f64_e64 %[st], %[w], %[lo]\n\t"                                                                               \
  "v_cmp_gt_f64_e64 %[so], %[w], %[hi]\n\t"                                                                               \
  "v_cmp_lt_f64_e64 %[so2], %[y], %[c35]\n\t"                                                                             \
  "s_and_b64 %[so], %[so], %[so2]\n\t"                                                                                    \
  "s_or_b64 %[so], %[so], %[st]\n\t"                                                                                      \
  "s_andn2_b64 %[so], %[nonleaf], %[so]\n\t"                                                                              \
  "s_cbranch_scc1 .LOTexact%=\n"                                                                                          \
  ".LOTdecided%=:\n\t"                                                                                                    \
  "s_andn2_b64 %[take], exec, %[nonleaf]\n\t"                                                                             \
  "s_or_b64 %[take], %[take], %[st]\n\t"                                                                                  \
  "s_mov_b64 %[act], exec\n\t"                                                                                            \
  "s_andn2_b64 %[open], exec, %[take]\n\t"                                                                                \
  "s_and_b64 exec, %[take], %[take]\n\t"                                                                                  \
  "s_cbranch_scc0 .LOTnotake%=\n\t"                                                                                       \
  OT_FAR                                                                                                                  \
  "v_cmp_gt_u64_e64 %[near], %[nearhi], %[r2]\n\t"                                                                        \
  CNT_T                                                                                                                   \
  "s_cmp_lg_u64 %[near], 0\n\t"                                                                                           \
  "s_cbranch_scc1 .LOTnear%=\n"                                                                                           \
  ".LOTacc%=:\n\t"                                                                                                        \
  "v_fmac_f64_e32 %[acc0], %[w], %[d0]\n\t"                                                                               \
  "v_fmac_f64_e32 %[acc1], %[w], %[d1]\n\t"                                                                               \
  Z("v_fmac_f64_e32 %[acc2], %[w], %[d2]\n\t")                                                                            \
  ".LOTnotake%=:\n\t"                                                                                                     \
  /* the children to open go on the body's stack in reverse child order; the 2^D lanes of a body pop the same entry */    \
  "s_mov_b64 exec, %[act]\n\t"                                                                                            \
  "v_lshrrev_b64 v[52:53], %[gshift], %[open]\n\t"                                                                        \
  "v_and_b32_e32 v52, " MASK ", v52\n\t"                                                                                    \
  "v_bcnt_u32_b32 %[nsp], v52, %[sp]\n\t"                                                                                 \
  "s_and_b64 exec, %[open], %[open]\n\t"                                                                                  \
  "v_lshrrev_b32_e32 v53, %[ccp1], v52\n\t"                                                                               \
  "v_bcnt_u32_b32 v53, v53, %[sp]\n\t"                                                                                    \
  "v_lshl_add_u32 v53, v53, " SSH ", %[stk]\n\t"                                                                            \
  "ds_write_b32 v53, v62\n\t"                                                                                             \
  "s_mov_b64 exec, %[act]\n\t"                                                                                            \
  "v_add_u32_e32 %[sp], -1, %[nsp]\n\t"                                                                                   \
  "v_cmpx_gt_u32_e64 %[act], %[depth], %[sp]\n\t"                                                                         \
  "s_cbranch_execz .LOTend%=\n\t"                                                                                         \
  "v_lshl_add_u32 v53, %[sp], " SSH ", %[stk]\n\t"                                                                            \
  "ds_read_b32 %[cur], v53\n\t"                                                                                           \
  "s_add_i32 %[guard], %[guard], -1\n\t"                                                                                  \
  "s_cmp_lg_u32 %[guard], 0\n\t"                                                                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                              \
  "s_cbranch_scc1 .LOTtop%=\n\t"                                                                                          \
  "s_mov_b32 %[gflag], 1\n\t"                                                                                             \
  "s_branch .LOTend%=\n"                                                                                                  \
  ".LOTexact%=:\n\t" /* some lane is inside the guard band: the reference's expression with IEEE sqrt and divide, for every lane (ot_accept) */\
  "v_mul_f64 v[52:53], %[d0], %[d0]\n\t"                                                                                  \
  "v_mul_f64 v[54:55], %[d1], %[d1]\n\t"                                                                                  \
  "v_add_f64 v[52:53], v[52:53], v[54:55]\n\t"                                                                            \
  Z("v_mul_f64 v[54:55], %[d2], %[d2]\n\t"                                                                                \
    "v_add_f64 v[52:53], v[54:55], v[52:53]\n\t")                                                                         \
  "v_cmp_gt_f64_e32 vcc, %[sqmin], v[52:53]\n\t"                                                                          \
  "s_nop 1\n\t"                                                                                                           \
  "v_mov_b32_e32 %[oa], 0x100\n\t"                                                                                          \
  "v_cndmask_b32_e32 %[t], 0, %[oa], vcc\n\t"                                                                           \
  "v_ldexp_f64 v[52:53], v[52:53], %[t]\n\t"                                                                              \
  "v_rsq_f64_e32 v[54:55], v[52:53]\n\t"                                                                                  \
  "v_mov_b32_e32 %[oa], 0xffffff80\n\t"                                                                                     \
  "v_cndmask_b32_e32 %[t], 0, %[oa], vcc\n\t"                                                                           \
  "v_cmp_class_f64_e64 vcc, v[52:53], %[c260]\n\t"                                                                        \
  "v_mul_f64 v[56:57], v[52:53], v[54:55]\n\t"                                                                            \
  "v_mul_f64 v[54:55], v[54:55], 0.5\n\t"                                                                                 \
  "v_fma_f64 v[58:59], -v[54:55], v[56:57], 0.5\n\t"                                                                      \
  "v_fmac_f64_e32 v[56:57], v[56:57], v[58:59]\n\t"                                                                       \
  "v_fma_f64 %[w], -v[56:57], v[56:57], v[52:53]\n\t"                                                                     \
  "v_fmac_f64_e32 v[54:55], v[54:55], v[58:59]\n\t"                                                                       \
  "v_fmac_f64_e32 v[56:57], %[w], v[54:55]\n\t"                                                                           \
  "v_fma_f64 v[58:59], -v[56:57], v[56:57], v[52:53]\n\t"                                                                 \
  "v_fmac_f64_e32 v[56:57], v[58:59], v[54:55]\n\t"                                                                       \
  "v_ldexp_f64 v[54:55], v[56:57], %[t]\n\t"                                                                              \
  "v_cndmask_b32_e32 v53, v55, v53, vcc\n\t"                                                                              \
  "v_cndmask_b32_e32 v52, v54, v52, vcc\n\t"                                                                              \
  "v_add_f64 v[52:53], v[52:53], %[eps]\n\t"                                                                              \
  "v_div_scale_f64 v[54:55], %[so2], v[52:53], v[52:53], %[g]\n\t"                                                        \
  "v_rcp_f64_e32 v[56:57], v[54:55]\n\t"                                                                                  \
  "s_nop 0\n\t"                                                                                                           \
  "v_fma_f64 v[58:59], -v[54:55], v[56:57], 1.0\n\t"                                                                      \
  "v_fmac_f64_e32 v[56:57], v[56:57], v[58:59]\n\t"                                                                       \
  "v_fma_f64 v[58:59], -v[54:55], v[56:57], 1.0\n\t"                                                                      \
  "v_fmac_f64_e32 v[56:57], v[56:57], v[58:59]\n\t"                                                                       \
  "v_div_scale_f64 v[58:59], vcc, %[g], v[52:53], %[g]\n\t"                                                               \
  "v_mul_f64 %[w], v[58:59], v[56:57]\n\t"                                                                                \
  "v_fma_f64 v[54:55], -v[54:55], %[w], v[58:59]\n\t"                                                                     \
  "s_nop 1\n\t"                                                                                                           \
  "v_div_fmas_f64 v[54:55], v[54:55], v[56:57], %[w]\n\t"                                                                 \
  "v_div_fixup_f64 v[52:53], v[54:55], v[52:53], %[g]\n\t"                                                                \
  "v_cmp_gt_f64_e64 %[st], %[theta], v[52:53]\n\t"                                                                        \
  "s_branch .LOTdecided%=\n"                                                                                              \
  ".LOTnear%=:\n\t" /* an accepted child closer than 2^-23 (the body's own leaf, coincident bodies): the guarded form, per lane */\
  "s_mov_b64 %[so], exec\n\t"                                                                                             \
  "s_mov_b64 exec, %[near]\n\t"                                                                                           \
  "v_mul_f64 %[y2], %[r2], %[y]\n\t"                                                                                      \
  "v_fma_f64 %[e], -%[y2], %[y], 1.0\n\t"                                                                                 \
  "v_mul_f64 %[p], %[y2], 0.5\n\t"                                                                                        \
  "v_fmac_f64_e32 %[y2], %[p], %[e]\n\t"                                                                                  \
  "v_add_f64 %[y2], %[y2], %[eps]\n\t"                                                                                    \
  "v_mul_f64 %[e], %[y2], %[y2]\n\t"                                                                                      \
  "v_mul_f64 %[y2], %[y2], %[e]\n\t"                                                                                      \
  "v_rcp_f64_e32 %[e], %[y2]\n\t"                                                                                         \
  "s_nop 0\n\t"                                                                                                           \
  "v_fma_f64 %[y2], -%[y2], %[e], 1.0\n\t"                                                                                \
  "v_fmac_f64_e32 %[e], %[e], %[y2]\n\t"                                                                                  \
  "v_mul_f64 %[w], v[60:61], %[e]\n\t"                                                                                    \
  "s_mov_b64 exec, %[so]\n\t"                                                                                             \
  "s_branch .LOTacc%=\n"                                                                                                  \
  ".LOTend%=:\n\t"                                                                                                        \
  "s_mov_b64 exec, %[sv]"

// The walk of ot_force_kernel for double precision in 3D with its visit round written as ISA — the same tests, the same
// arithmetic in the same order, bitwise the same results, counters and flags.  hipcc's schedule of the C++ round is ~96
// instructions (50 VALU): predicated weights, mask round trips through v_cndmask/v_cmp, one saveexec/branch pair per `if`.
// Here a round is ~60 (36 VALU): record addresses are 32-bit offsets from an SGPR base (two v_mad_u32_u24 instead of 64-bit
// multiply-adds), the conditions live in scalar masks and EXEC (accepted term under EXEC = take, push under EXEC = open), the
// step budget is one scalar counter for the wave (its lanes make the same rounds), and "stack empty" and "stack full" are ONE
// unsigned compare on sp - 1 (the stacks are entry-major in LDS, so a round that overflows writes past the END of the block's
// LDS, where the hardware drops it; the lane then leaves and reports kFlagStack).  The records of the round live in v[52:63] (their halves are addressed
// separately, which an asm operand cannot express).  Group offsets are 32 bits: the host uses this kernel while the group
// array is below 4 GiB (N <= 1.3e7).
template <int D, bool COUNT>
__global__ __launch_bounds__(64) void ot_force_isa_kernel(const ot_node<double>* __restrict__ rootrec,
                                                          const ot_group<double, D>* __restrict__ groups,
                                                          const uint32_t* __restrict__ list, uint32_t nlist,
                                                          const double* __restrict__ x, double* __restrict__ a, double c,
                                                          uint32_t first, double theta, uint32_t capacity,
                                                          const double* __restrict__ root, uint32_t* __restrict__ flags,
                                                          uint32_t* __restrict__ counters) {
  using T = double;
  constexpr uint32_t NCH = 1u << D, GPW = 64u / NCH;
  constexpr uint32_t DEPTH = (NCH - 1u) * kMaxLevels<D> + NCH;
  static_assert(sizeof(ot_group<double, 3>) == 320 && sizeof(ot_group<double, 2>) == 128, "the round addresses 320- / 128-byte sibling groups");
  __shared__ uint32_t stack[DEPTH][GPW];  // entry-major: slot i of body g at (i * GPW + g) * 4
  const uint32_t g = threadIdx.x / NCH, cc = threadIdx.x % NCH;
  const uint32_t t    = ot_xcd_contiguous_block(blockIdx.x, gridDim.x) * GPW + g;
  const bool valid    = t < nlist;
  const uint32_t body = valid ? list[t] : first;
  const ot_theta<T> th(theta);
  const pair_consts<T> pc;
  const T root_side = root[D];
  T xi[3] = {T(0), T(0), T(0)}, acc[3] = {T(0), T(0), T(0)};  // (the third pair is an unused operand in 2D)
#pragma unroll
  for (int k = 0; k < D; ++k) xi[k] = valid ? x[uint64_t(body) * D + k] : T(0);
  uint32_t c_nodes = 0, c_terms = 0;
  uint32_t cur = 0;
  bool more = false;
  if (valid) {  // the root is examined alone, exactly as in ot_force_kernel
    const ot_node<T> nd = *rootrec;
    T di[D], a0[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      di[k] = nd.p[k] - xi[k];
      a0[k] = T(0);
    }
    const T d2f     = ot_dist2_fused<T, D>(di);
    const T y0      = ot_rsq(d2f);
    const bool leaf = nd.fc >= kOtBody;
    const bool take = leaf || ot_accept<T, D>(!leaf, root_side, di, y0, th);
    {
      const bool on0    = take && cc == 0;
      const uint64_t m0 = __builtin_amdgcn_ballot_w64(on0);
      if (m0 != 0ull) ot_accumulate<T, D>(on0, m0, a0, di, nd.m, d2f, y0, pc);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) acc[k] = a0[k];
    if (COUNT && cc == 0) {
      c_nodes = 1;
      c_terms = take;
    }
    more = !take;
    cur  = nd.fc;
  }
  // the rounds
  uint32_t more_v = more ? 1u : 0u, sp = 0, nsp = 0, gflag = 0, guard = capacity;
  const uint32_t cc16 = cc * 16u, cc8 = (D == 3 ? 256u : 96u) + cc * 8u, gshift = threadIdx.x & (64u - NCH), ccp1 = cc + 1u;
  const uint32_t stk = uint32_t(reinterpret_cast<uintptr_t>(&stack[0][g]));  // LDS byte address (low word of the flat one)
  double lo = to_sgpr(th.lo), hi = to_sgpr(th.hi), thx = to_sgpr(th.exact), c35 = 0x1p35, tiny = ot_consts<T>::tiny,
         eps = ot_consts<T>::eps, m3eps = -(3.0 * ot_consts<T>::eps), sqmin = 0x1p-767, rs = to_sgpr(root_side);
  uint64_t nearhi = uint64_t(ot_near<T>::bits) << 32;
  uint32_t gsz = uint32_t(sizeof(ot_group<double, D>)), depth = DEPTH, c260 = 0x260u;
  asm volatile("" : "+s"(c35), "+s"(tiny), "+s"(eps), "+s"(m3eps), "+s"(sqmin), "+s"(nearhi), "+s"(gsz), "+s"(depth), "+s"(c260));
  double d0, d1, d2, r2, y, y2, e, p, gq, w;
  uint32_t oa, of, tt;
  uint64_t nonleaf, st, so, so2, take, open, act, sv, near;
#define OT_OPERANDS                                                                                                        \
  : [acc0] "+v"(acc[0]), [acc1] "+v"(acc[1]), [acc2] "+v"(acc[2]), [cur] "+v"(cur), [sp] "+v"(sp), [nsp] "+v"(nsp),          \
    OT_CNT_OPERANDS [guard] "+s"(guard), [gflag] "+s"(gflag), [d0] "=&v"(d0), [d1] "=&v"(d1),       \
    [d2] "=&v"(d2), [r2] "=&v"(r2), [y] "=&v"(y), [y2] "=&v"(y2), [e] "=&v"(e), [p] "=&v"(p), [g] "=&v"(gq), [w] "=&v"(w),  \
    [oa] "=&v"(oa), [of] "=&v"(of), [t] "=&v"(tt), [nonleaf] "=&s"(nonleaf), [st] "=&s"(st), [so] "=&s"(so),               \
    [so2] "=&s"(so2), [take] "=&s"(take), [open] "=&s"(open), [act] "=&s"(act), [sv] "=&s"(sv), [near] "=&s"(near)          \
  : [more] "v"(more_v), [groups] "s"(groups), [xi0] "v"(xi[0]), [xi1] "v"(xi[1]), [xi2] "v"(xi[2]), [cc16] "v"(cc16),     \
    [cc8] "v"(cc8), [gshift] "v"(gshift), [ccp1] "v"(ccp1), [stk] "v"(stk), [k15] "v"(pc.k15), [k1875] "s"(pc.k1875),       \
    [lo] "s"(lo), [hi] "s"(hi), [theta] "s"(thx), [c35] "s"(c35), [tiny] "s"(tiny), [eps] "s"(eps), [m3eps] "s"(m3eps),    \
    [sqmin] "s"(sqmin), [nearhi] "s"(nearhi), [rootside] "s"(rs), [gsz] "s"(gsz), [depth] "s"(depth), [c260] "s"(c260)                                                                                     \
  : "vcc", "scc", "memory", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
#define OT_CNT_OPERANDS [cn] "+v"(c_nodes), [ct] "+v"(c_terms),
#define OT_CN "v_add_u32_e32 %[cn], 1, %[cn]\n\t"
#define OT_CT "v_add_u32_e32 %[ct], 1, %[ct]\n\t"
  if constexpr (COUNT && D == 3) asm volatile(OT_ISA_TEXT(OT_KEEP, OT_LOADS_F64_D3, "0xff", "5", OT_CN, OT_CT) OT_OPERANDS);
  if constexpr (COUNT && D == 2) asm volatile(OT_ISA_TEXT(OT_DROP, OT_LOADS_F64_D2, "0xf", "6", OT_CN, OT_CT) OT_OPERANDS);
#undef OT_CNT_OPERANDS
#define OT_CNT_OPERANDS
  if constexpr (!COUNT && D == 3) asm volatile(OT_ISA_TEXT(OT_KEEP, OT_LOADS_F64_D3, "0xff", "5", "", "") OT_OPERANDS);
  if constexpr (!COUNT && D == 2) asm volatile(OT_ISA_TEXT(OT_DROP, OT_LOADS_F64_D2, "0xf", "6", "", "") OT_OPERANDS);
#undef OT_CNT_OPERANDS
#undef OT_OPERANDS
  if (more && nsp > DEPTH && cc == 0) atomicOr(flags, kFlagStack);  // a stack ran full: reported by nbody_octree_info
  if (gflag != 0u && threadIdx.x == 0) atomicOr(flags, kFlagWalk);             // step budget spent: the tree is damaged
  // combine the 2^D partial sums of a body (fixed order)
#pragma unroll
  for (uint32_t off = NCH / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < D; ++k) acc[k] += __shfl_xor(acc[k], int(off), 64);
    if (COUNT) {
      c_nodes += __shfl_xor(c_nodes, int(off), 64);
      c_terms += __shfl_xor(c_terms, int(off), 64);
    }
  }
  if (valid && cc == 0) {
#pragma unroll
    for (int k = 0; k < D; ++k) a[uint64_t(body - first) * D + k] = c * acc[k];
    if (COUNT) {
      counters[uint64_t(body) * 2 + 0] = c_nodes;
      counters[uint64_t(body) * 2 + 1] = c_terms;
    }
  }
}


// ---- the same visit round for single precision in 3D (the reference's DEFAULT run: float, octree) ------------------------------
// 192-byte sibling groups: (p0, p1, p2, m) x 8 | (fc, depth) x 8 — two loads per round.  Same tests, same arithmetic in the same
// order as ot_force_kernel<float, 3>: the quick test q = side * rsq(d2), ql = q - q (eps y) against theta (1 -+ 2^-16); if a
// non-leaf lane is inside the band the wave evaluates the reference's side / (sqrt(d2) + eps) < theta with the correctly rounded
// f32 square root and quotient (v_sqrt_f32 / v_rcp_f32 seeds + the fix-up sequences hipcc emits for them, instruction for
// instruction); the accepted term m y^3 (1 - 3 eps y), the guarded form below d2 = 2^-18 per lane.  The record lives in v[54:57]
// and v[62:63], temporaries in v52, v53, v58-v61.
#define OT_FAR_F32                                                                                                        \
  "v_mul_f32_e32 %[t], %[y], %[y]\n\t"                                                                                    \
  "v_mul_f32_e32 %[t], %[y], %[t]\n\t"                                                                                    \
  "v_mul_f32_e32 %[w], v57, %[t]\n\t"                                                                                     \
  "v_mul_f32_e32 %[t], %[m3eps], %[y]\n\t"                                                                                \
  "v_fmac_f32_e32 %[w], %[w], %[t]\n\t"

#define OT_ISA_TEXT_F32(Z, MASK, SSH, CNT_N, CNT_T)                                                                                  \
  "s_mov_b64 %[sv], exec\n\t"                                                                                             \
  "v_cmp_ne_u32_e32 vcc, 0, %[more]\n\t"                                                                                  \
  "s_and_b64 exec, exec, vcc\n\t"                                                                                         \
  "s_cbranch_execz .LOFend%=\n"                                                                                           \
  ".LOFtop%=:\n\t"                                                                                                        \
  "v_mad_u32_u24 %[oa], %[cur], %[gsz], %[cc16]\n\t"                                                                      \
  "v_mad_u32_u24 %[of], %[cur], %[gsz], %[cc8]\n\t"                                                                       \
  "global_load_dwordx4 v[54:57], %[oa], %[groups]\n\t"                                                                    \
  "global_load_dwordx2 v[62:63], %[of], %[groups]\n\t"                                                                    \
  CNT_N                                                                                                                   \
  "s_waitcnt vmcnt(1)\n\t"                                                                                                \
  "v_sub_f32_e32 %[d0], v54, %[xi0]\n\t"                                                                                  \
  "v_sub_f32_e32 %[d1], v55, %[xi1]\n\t"                                                                                  \
  Z("v_sub_f32_e32 %[d2], v56, %[xi2]\n\t")                                                                               \
  "v_fma_f32 %[r2], %[d0], %[d0], %[tiny]\n\t"                                                                            \
  "v_fmac_f32_e32 %[r2], %[d1], %[d1]\n\t"                                                                                \
  Z("v_fmac_f32_e32 %[r2], %[d2], %[d2]\n\t")                                                                             \
  "v_rsq_f32_e32 %[y], %[r2]\n\t"                                                                                         \
  "s_waitcnt vmcnt(0)\n\t"                                                                                                \
  "v_sub_u32_e32 %[t], 0, v63\n\t"                                                                                        \
  "v_ldexp_f32 %[g], %[rootside], %[t]\n\t"                                                                               \
  "v_cmp_gt_u32_e64 %[nonleaf], -2, v62\n\t"                                                                              \
  "v_mul_f32_e32 %[w], %[g], %[y]\n\t"                                                                                    \
  "v_mul_f32_e32 %[t], %[meps], %[y]\n\t"                                                                                 \
  "v_cmp_lt_f32_e64 %[st], %[w], %[lo]\n\t"                                                                               \
  "v_fmac_f32_e32 %[w], %[w], %[t]\n\t"                                                                                   \
  "v_cmp_gt_f32_e64 %[so], %[w], %[hi]\n\t"                                                                               \
  "s_or_b64 %[so], %[so], %[st]\n\t"                                                                                      \
  "s_andn2_b64 %[so], %[nonleaf], %[so]\n\t"                                                                              \
  "s_cbranch_scc1 .LOFexact%=\n"                                                                                          \
  ".LOFdecided%=:\n\t"                                                                                                    \
  "s_andn2_b64 %[take], exec, %[nonleaf]\n\t"                                                                             \
  "s_or_b64 %[take], %[take], %[st]\n\t"                                                                                  \
  "s_mov_b64 %[act], exec\n\t"                                                                                            \
  "s_andn2_b64 %[open], exec, %[take]\n\t"                                                                                \
  "s_and_b64 exec, %[take], %[take]\n\t"                                                                                  \
  "s_cbranch_scc0 .LOFnotake%=\n\t"                                                                                       \
  OT_FAR_F32                                                                                                              \
  "v_cmp_gt_u32_e64 %[near], %[nearbits], %[r2]\n\t"                                                                      \
  CNT_T                                                                                                                   \
  "s_cmp_lg_u64 %[near], 0\n\t"                                                                                           \
  "s_cbranch_scc1 .LOFnear%=\n"                                                                                           \
  ".LOFacc%=:\n\t"                                                                                                        \
  "v_fmac_f32_e32 %[acc0], %[w], %[d0]\n\t"                                                                               \
  "v_fmac_f32_e32 %[acc1], %[w], %[d1]\n\t"                                                                               \
  Z("v_fmac_f32_e32 %[acc2], %[w], %[d2]\n\t")                                                                            \
  ".LOFnotake%=:\n\t"                                                                                                     \
  "s_mov_b64 exec, %[act]\n\t"                                                                                            \
  "v_lshrrev_b64 v[52:53], %[gshift], %[open]\n\t"                                                                        \
  "v_and_b32_e32 v52, " MASK ", v52\n\t"                                                                                    \
  "v_bcnt_u32_b32 %[nsp], v52, %[sp]\n\t"                                                                                 \
  "s_and_b64 exec, %[open], %[open]\n\t"                                                                                  \
  "v_lshrrev_b32_e32 v53, %[ccp1], v52\n\t"                                                                               \
  "v_bcnt_u32_b32 v53, v53, %[sp]\n\t"                                                                                    \
  "v_lshl_add_u32 v53, v53, " SSH ", %[stk]\n\t"                                                                            \
  "ds_write_b32 v53, v62\n\t"                                                                                             \
  "s_mov_b64 exec, %[act]\n\t"                                                                                            \
  "v_add_u32_e32 %[sp], -1, %[nsp]\n\t"                                                                                   \
  "v_cmpx_gt_u32_e64 %[act], %[depth], %[sp]\n\t"                                                                         \
  "s_cbranch_execz .LOFend%=\n\t"                                                                                         \
  "v_lshl_add_u32 v53, %[sp], " SSH ", %[stk]\n\t"                                                                            \
  "ds_read_b32 %[cur], v53\n\t"                                                                                           \
  "s_add_i32 %[guard], %[guard], -1\n\t"                                                                                  \
  "s_cmp_lg_u32 %[guard], 0\n\t"                                                                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                              \
  "s_cbranch_scc1 .LOFtop%=\n\t"                                                                                          \
  "s_mov_b32 %[gflag], 1\n\t"                                                                                             \
  "s_branch .LOFend%=\n"                                                                                                  \
  ".LOFexact%=:\n\t" /* the reference's expression with correctly rounded sqrt and quotient, for every lane (ot_accept) */  \
  "v_mul_f32_e32 v52, %[d0], %[d0]\n\t"                                                                                   \
  "v_mul_f32_e32 v53, %[d1], %[d1]\n\t"                                                                                   \
  "v_add_f32_e32 v52, v52, v53\n\t"                                                                                       \
  Z("v_mul_f32_e32 v53, %[d2], %[d2]\n\t"                                                                                 \
    "v_add_f32_e32 v52, v53, v52\n\t")                                                                                    \
  "v_mul_f32_e32 v53, 0x4f800000, v52\n\t"                                                                                \
  "v_cmp_gt_f32_e32 vcc, %[sqmin], v52\n\t"                                                                               \
  "s_nop 1\n\t"                                                                                                           \
  "v_cndmask_b32_e32 v52, v52, v53, vcc\n\t"                                                                              \
  "v_sqrt_f32_e32 v53, v52\n\t"                                                                                           \
  "s_nop 0\n\t"                                                                                                           \
  "v_add_u32_e32 v58, -1, v53\n\t"                                                                                        \
  "v_fma_f32 v59, -v58, v53, v52\n\t"                                                                                     \
  "v_cmp_ge_f32_e64 %[so2], 0, v59\n\t"                                                                                   \
  "v_add_u32_e32 v59, 1, v53\n\t"                                                                                         \
  "s_nop 0\n\t"                                                                                                           \
  "v_cndmask_b32_e64 v58, v53, v58, %[so2]\n\t"                                                                           \
  "v_fma_f32 v53, -v59, v53, v52\n\t"                                                                                     \
  "v_cmp_lt_f32_e64 %[so2], 0, v53\n\t"                                                                                   \
  "s_nop 1\n\t"                                                                                                           \
  "v_cndmask_b32_e64 v53, v58, v59, %[so2]\n\t"                                                                           \
  "v_mul_f32_e32 v58, 0x37800000, v53\n\t"                                                                                \
  "v_cndmask_b32_e32 v53, v53, v58, vcc\n\t"                                                                              \
  "v_mov_b32_e32 v58, 0x260\n\t"                                                                                          \
  "v_cmp_class_f32_e32 vcc, v52, v58\n\t"                                                                                 \
  "s_nop 1\n\t"                                                                                                           \
  "v_cndmask_b32_e32 v52, v53, v52, vcc\n\t"                                                                              \
  "v_add_f32_e32 v52, %[eps], v52\n\t"                                                                                    \
  "v_div_scale_f32 v53, %[so2], v52, v52, %[g]\n\t"                                                                       \
  "v_rcp_f32_e32 v58, v53\n\t"                                                                                            \
  "s_nop 0\n\t"                                                                                                           \
  "v_fma_f32 v59, -v53, v58, 1.0\n\t"                                                                                     \
  "v_fmac_f32_e32 v58, v59, v58\n\t"                                                                                      \
  "v_div_scale_f32 v59, vcc, %[g], v52, %[g]\n\t"                                                                         \
  "v_mul_f32_e32 v60, v59, v58\n\t"                                                                                       \
  "v_fma_f32 v61, -v53, v60, v59\n\t"                                                                                     \
  "v_fmac_f32_e32 v60, v61, v58\n\t"                                                                                      \
  "v_fma_f32 v53, -v53, v60, v59\n\t"                                                                                     \
  "s_nop 0\n\t"                                                                                                           \
  "v_div_fmas_f32 v53, v53, v58, v60\n\t"                                                                                 \
  "v_div_fixup_f32 v52, v53, v52, %[g]\n\t"                                                                               \
  "v_cmp_gt_f32_e64 %[st], %[theta], v52\n\t"                                                                             \
  "s_branch .LOFdecided%=\n"                                                                                              \
  ".LOFnear%=:\n\t" /* an accepted child closer than 2^-9 (the body's own leaf, coincident bodies): the guarded form, per lane */\
  "s_mov_b64 %[so], exec\n\t"                                                                                             \
  "s_mov_b64 exec, %[near]\n\t"                                                                                           \
  "v_mul_f32_e32 %[t], %[r2], %[y]\n\t"                                                                                   \
  "v_fma_f32 v52, -%[t], %[y], 1.0\n\t"                                                                                   \
  "v_mul_f32_e32 v53, 0.5, %[t]\n\t"                                                                                      \
  "v_fmac_f32_e32 %[t], v53, v52\n\t"                                                                                     \
  "v_add_f32_e32 %[t], %[eps], %[t]\n\t"                                                                                  \
  "v_mul_f32_e32 v52, %[t], %[t]\n\t"                                                                                     \
  "v_mul_f32_e32 %[t], %[t], v52\n\t"                                                                                     \
  "v_rcp_f32_e32 v52, %[t]\n\t"                                                                                           \
  "s_nop 0\n\t"                                                                                                           \
  "v_fma_f32 %[t], -%[t], v52, 1.0\n\t"                                                                                   \
  "v_fmac_f32_e32 v52, v52, %[t]\n\t"                                                                                     \
  "v_mul_f32_e32 %[w], v57, v52\n\t"                                                                                      \
  "s_mov_b64 exec, %[so]\n\t"                                                                                             \
  "s_branch .LOFacc%=\n"                                                                                                  \
  ".LOFend%=:\n\t"                                                                                                        \
  "s_mov_b64 exec, %[sv]"

template <int D, bool COUNT>
__global__ __launch_bounds__(64) void ot_force_isa_f32_kernel(const ot_node<float>* __restrict__ rootrec,
                                                              const ot_group<float, D>* __restrict__ groups,
                                                              const uint32_t* __restrict__ list, uint32_t nlist,
                                                              const float* __restrict__ x, float* __restrict__ a, float c,
                                                              uint32_t first, float theta, uint32_t capacity,
                                                              const float* __restrict__ root, uint32_t* __restrict__ flags,
                                                              uint32_t* __restrict__ counters) {
  using T = float;
  constexpr uint32_t NCH = 1u << D, GPW = 64u / NCH;
  constexpr uint32_t DEPTH = (NCH - 1u) * kMaxLevels<D> + NCH;
  static_assert(sizeof(ot_group<float, 3>) == 192 && sizeof(ot_group<float, 2>) == 128, "the round addresses 192- / 128-byte sibling groups");
  __shared__ uint32_t stack[DEPTH][GPW];  // entry-major: slot i of body g at (i * GPW + g) * 4
  const uint32_t g = threadIdx.x / NCH, cc = threadIdx.x % NCH;
  const uint32_t t    = ot_xcd_contiguous_block(blockIdx.x, gridDim.x) * GPW + g;
  const bool valid    = t < nlist;
  const uint32_t body = valid ? list[t] : first;
  const ot_theta<T> th(theta);
  const pair_consts<T> pc;
  const T root_side = root[D];
  T xi[3] = {T(0), T(0), T(0)}, acc[3] = {T(0), T(0), T(0)};  // (the third pair is an unused operand in 2D)
#pragma unroll
  for (int k = 0; k < D; ++k) xi[k] = valid ? x[uint64_t(body) * D + k] : T(0);
  uint32_t c_nodes = 0, c_terms = 0;
  uint32_t cur = 0;
  bool more = false;
  if (valid) {  // the root is examined alone, exactly as in ot_force_kernel
    const ot_node<T> nd = *rootrec;
    T di[D], a0[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      di[k] = nd.p[k] - xi[k];
      a0[k] = T(0);
    }
    const T d2f     = ot_dist2_fused<T, D>(di);
    const T y0      = ot_rsq(d2f);
    const bool leaf = nd.fc >= kOtBody;
    const bool take = leaf || ot_accept<T, D>(!leaf, root_side, di, y0, th);
    {
      const bool on0    = take && cc == 0;
      const uint64_t m0 = __builtin_amdgcn_ballot_w64(on0);
      if (m0 != 0ull) ot_accumulate<T, D>(on0, m0, a0, di, nd.m, d2f, y0, pc);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) acc[k] = a0[k];
    if (COUNT && cc == 0) {
      c_nodes = 1;
      c_terms = take;
    }
    more = !take;
    cur  = nd.fc;
  }
  uint32_t more_v = more ? 1u : 0u, sp = 0, nsp = 0, gflag = 0, guard = capacity;
  const uint32_t cc16 = cc * 16u, cc8 = (D == 3 ? 128u : 64u) + cc * 8u, gshift = threadIdx.x & (64u - NCH), ccp1 = cc + 1u;
  const uint32_t stk = uint32_t(reinterpret_cast<uintptr_t>(&stack[0][g]));
  float lo = to_sgpr(th.lo), hi = to_sgpr(th.hi), thx = to_sgpr(th.exact), tiny = ot_consts<T>::tiny, eps = ot_consts<T>::eps,
        meps = -ot_consts<T>::eps, m3eps = -(3.0f * ot_consts<T>::eps), sqmin = 0x1p-96f, rs = to_sgpr(root_side);
  uint32_t nearbits = ot_near<T>::bits, gsz = uint32_t(sizeof(ot_group<float, D>)), depth = DEPTH;
  asm volatile("" : "+s"(tiny), "+s"(eps), "+s"(meps), "+s"(m3eps), "+s"(sqmin), "+s"(nearbits), "+s"(gsz), "+s"(depth));
  float d0, d1, d2, r2, y, gq, w, tt;
  uint32_t oa, of;
  uint64_t nonleaf, st, so, so2, take, open, act, sv, near;
#define OT_OPERANDS                                                                                                        \
  : [acc0] "+v"(acc[0]), [acc1] "+v"(acc[1]), [acc2] "+v"(acc[2]), [cur] "+v"(cur), [sp] "+v"(sp), [nsp] "+v"(nsp),          \
    OT_CNT_OPERANDS [guard] "+s"(guard), [gflag] "+s"(gflag), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2),              \
    [r2] "=&v"(r2), [y] "=&v"(y), [g] "=&v"(gq), [w] "=&v"(w), [t] "=&v"(tt), [oa] "=&v"(oa), [of] "=&v"(of),             \
    [nonleaf] "=&s"(nonleaf), [st] "=&s"(st), [so] "=&s"(so), [so2] "=&s"(so2), [take] "=&s"(take), [open] "=&s"(open),     \
    [act] "=&s"(act), [sv] "=&s"(sv), [near] "=&s"(near)                                                                   \
  : [more] "v"(more_v), [groups] "s"(groups), [xi0] "v"(xi[0]), [xi1] "v"(xi[1]), [xi2] "v"(xi[2]), [cc16] "v"(cc16),     \
    [cc8] "v"(cc8), [gshift] "v"(gshift), [ccp1] "v"(ccp1), [stk] "v"(stk), [lo] "s"(lo), [hi] "s"(hi), [theta] "s"(thx),  \
    [tiny] "s"(tiny), [eps] "s"(eps), [meps] "s"(meps), [m3eps] "s"(m3eps), [sqmin] "s"(sqmin), [nearbits] "s"(nearbits),  \
    [rootside] "s"(rs), [gsz] "s"(gsz), [depth] "s"(depth)                                                              \
  : "vcc", "scc", "memory", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
#define OT_CNT_OPERANDS [cn] "+v"(c_nodes), [ct] "+v"(c_terms),
  if constexpr (COUNT && D == 3) asm volatile(OT_ISA_TEXT_F32(OT_KEEP, "0xff", "5", OT_CN, OT_CT) OT_OPERANDS);
  if constexpr (COUNT && D == 2) asm volatile(OT_ISA_TEXT_F32(OT_DROP, "0xf", "6", OT_CN, OT_CT) OT_OPERANDS);
#undef OT_CNT_OPERANDS
#define OT_CNT_OPERANDS
  if constexpr (!COUNT && D == 3) asm volatile(OT_ISA_TEXT_F32(OT_KEEP, "0xff", "5", "", "") OT_OPERANDS);
  if constexpr (!COUNT && D == 2) asm volatile(OT_ISA_TEXT_F32(OT_DROP, "0xf", "6", "", "") OT_OPERANDS);
#undef OT_CNT_OPERANDS
#undef OT_OPERANDS
  if (more && nsp > DEPTH && cc == 0) atomicOr(flags, kFlagStack);
  if (gflag != 0u && threadIdx.x == 0) atomicOr(flags, kFlagWalk);
#pragma unroll
  for (uint32_t off = NCH / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < D; ++k) acc[k] += __shfl_xor(acc[k], int(off), 64);
    if (COUNT) {
      c_nodes += __shfl_xor(c_nodes, int(off), 64);
      c_terms += __shfl_xor(c_terms, int(off), 64);
    }
  }
  if (valid && cc == 0) {
#pragma unroll
    for (int k = 0; k < D; ++k) a[uint64_t(body - first) * D + k] = c * acc[k];
    if (COUNT) {
      counters[uint64_t(body) * 2 + 0] = c_nodes;
      counters[uint64_t(body) * 2 + 1] = c_terms;
    }
  }
}

}  // namespace nbody

// ---- host side / C ABI -----------------------------------------------------------------------------------------------
struct nbody_octree {
  int dtype = 0, dim = 0, device = 0;  // device: nbody_octree_create_on's (nbody_octree_create: the current one); every call runs there
  int walk = 0;                        // nbody_octree_set_walk: 0 auto, 1 compiler-scheduled kernel, 2 the visit round as ISA
  int build = 0;                       // nbody_octree_set_build: 0 auto (= 3), 1 one launch per level, 2 all levels in one launch
                                       // (grid barrier), 3 one pass over the sorted keys (ot_build_lcp_kernel)
  int depth_hint = 64;                 // levels launched one by one (the rest share one launch); from the last nbody_octree_info
  int ncu = 0;                         // compute units of the device (the all-level kernels launch at most one block per CU)
  uint32_t step_budget = 0;            // nbody_octree_set_step_budget: visit rounds a body may make; 0 = the node pool size
  uint32_t n = 0, capacity = 0, max_cells = 0, bounds_blocks = 0;
  size_t tsz = 0;
  void* root       = nullptr;  // T[D+1]: root_x (D), root_side_length
  void* partials   = nullptr;
  uint64_t* keys[2] = {nullptr, nullptr};
  uint32_t* idx[2]  = {nullptr, nullptr};
  uint32_t* hist   = nullptr;
  void* rootrec    = nullptr;  // ot_node<T>: node 0
  void* groups     = nullptr;  // ot_group<T,D>[max_cells]: sibling group g = nodes 1 + g * 2^D ...
  size_t group_bytes = 0;
  nbody::ot_cell* cells = nullptr;
  nbody::ot_cell* tops  = nullptr;  // one-pass build: the cells at the key depth, for ot_build_deep_kernel
  int8_t* lcp           = nullptr;  // one-pass build: l_i for positions 0 ... n
  uint32_t* plocal      = nullptr;  //   block-local exclusive prefix of the cells starting at each position
  uint32_t* bsum        = nullptr;  //   per-block sums -> bases
  uint32_t* bhist       = nullptr;  //   per-block cells per level
  uint32_t* rankpos     = nullptr;  //   rank -> sorted position its cell starts at
  uint32_t* later       = nullptr;  //   ranks left to ot_multipole_crown_kernel: slot (chunk, level)
  uint32_t* later_mask  = nullptr;  //   per chunk: the levels whose slot is in use
  uint32_t* later2 = nullptr, *later2_mask = nullptr;  //   the same for the blocks of ot_multipole_round_kernel
  uint32_t* lvl_count = nullptr;  // [MAXL + 2] level counts and deep groups, flags, two barrier counters, deep-list cursor, crown count
  uint32_t* counters = nullptr;
  int sorted_buf   = 0;
  bool counters_on = false, have_bounds = false, inserted = false, have_tree = false;
};

using namespace nbody;

namespace nbody {

static int ot_check(const nbody_octree* t, const nbody_state* s, void* stream) {
  NB_ARG(t != nullptr, "nbody_octree is NULL");
  if (int r = check_state(s)) return r;
  if (int r = check_same_device(t->device, as_stream(stream), "nbody_octree")) return r;
  NB_ARG(t->dtype == s->dtype && t->dim == s->dim && t->n == s->sz,
         "octree was created for (dtype=%d, dim=%d, n=%u), state is (%d, %d, %u)", t->dtype, t->dim, t->n, s->dtype, s->dim, s->sz);
  return NBODY_OK;
}

template <typename T, int D>
static int ot_bounds_run(nbody_octree* t, const nbody_state* s, hipStream_t st) {
  if (s->sz <= kSmallN) {
    hipLaunchKernelGGL((ot_bounds_small_kernel<T, D>), dim3(1), dim3(kOB), 0, st, static_cast<const T*>(s->x), s->sz * uint32_t(D),
                       static_cast<T*>(t->root));
    NB_HIP(hipGetLastError());
    return NBODY_OK;
  }
  hipLaunchKernelGGL((ot_bounds_partial_kernel<T>), dim3(t->bounds_blocks), dim3(kOB), 0, st, static_cast<const T*>(s->x),
                     uint64_t(s->sz) * D, static_cast<T*>(t->partials));
  NB_HIP(hipGetLastError());
  hipLaunchKernelGGL((ot_bounds_final_kernel<T, D>), dim3(1), dim3(kOB), 0, st, static_cast<const T*>(t->partials),
                     t->bounds_blocks, static_cast<T*>(t->root));
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

template <typename T, int D>
static int ot_insert_run(nbody_octree* t, const nbody_state* s, hipStream_t st) {
  constexpr uint32_t NCH = 1u << D;
  const uint32_t n       = s->sz;
  const ot_tree<T, D> tree{static_cast<ot_group<T, D>*>(t->groups), static_cast<ot_node<T>*>(t->rootrec)};
  if (n <= kSmallN && (t->build == 0 || t->build == 3)) {  // one block does everything up to the cells
    uint32_t* flags = t->lvl_count + (kMaxLevels<D> + 2);
    const ot_lcp_args<T, D> args{t->keys[0], t->idx[0], n, t->lcp, t->plocal, t->bsum, t->rankpos, static_cast<const T*>(s->m),
                                 static_cast<const T*>(s->x), tree, t->cells, t->tops, t->lvl_count, flags, t->capacity, t->max_cells};
    hipLaunchKernelGGL((ot_insert_small_kernel<T, D>), dim3(1), dim3(kSmallB), 0, st, args, static_cast<const T*>(t->root), t->keys[0],
                       t->idx[0], t->idx[1]);
    NB_HIP(hipGetLastError());
    t->sorted_buf = 0;
    return NBODY_OK;
  }
  hipLaunchKernelGGL((ot_keys_kernel<T, D>), dim3((n + kOB - 1) / kOB), dim3(kOB), 0, st, static_cast<const T*>(s->x), n,
                     static_cast<const T*>(t->root), t->keys[0]);
  NB_HIP(hipGetLastError());
  int fin = 0;
  if (int r = radix_sort_pairs(t->keys, t->idx, n, D == 3 ? 63 : 64, t->hist, st, &fin)) return r;
  t->sorted_buf = fin;
  uint32_t* flags = t->lvl_count + (kMaxLevels<D> + 2);
  if (t->build == 0 || t->build == 3) {  // one pass over the sorted keys: 4 launches whatever the depth
    const uint32_t nblk = n / kLcpB + 1;  // positions 0 ... n
    hipLaunchKernelGGL((ot_lcp_kernel<D>), dim3(nblk), dim3(kLcpB), 0, st, t->keys[fin], n, t->lcp, t->plocal, t->bsum, t->bhist);
    NB_HIP(hipGetLastError());
    hipLaunchKernelGGL((ot_lcp_finish_kernel<T, D>), dim3(1), dim3(1024), 0, st, n, nblk, t->bsum, t->bhist,
                       static_cast<const T*>(s->m), static_cast<const T*>(s->x), tree, t->lvl_count);
    NB_HIP(hipGetLastError());
    hipLaunchKernelGGL((ot_lcp_scatter_kernel<D>), dim3((n + kOB - 1) / kOB), dim3(kOB), 0, st, n, t->lcp, t->plocal, t->bsum, t->rankpos,
                       t->max_cells);
    NB_HIP(hipGetLastError());
    {
      const ot_lcp_args<T, D> args{t->keys[fin], t->idx[fin], n, t->lcp, t->plocal, t->bsum, t->rankpos, static_cast<const T*>(s->m),
                                   static_cast<const T*>(s->x), tree, t->cells, t->tops, t->lvl_count, flags, t->capacity, t->max_cells};
      const uint32_t blocks = uint32_t((uint64_t(t->max_cells) * NCH + kLcpBuildB - 1) / kLcpBuildB);
      hipLaunchKernelGGL((ot_build_lcp_kernel<T, D>), dim3(blocks), dim3(kLcpBuildB), 0, st, args);
      NB_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL((ot_build_deep_kernel<T, D>), dim3(64), dim3(64), 0, st, t->idx[fin], t->idx[1 - fin],
                       static_cast<const T*>(s->m), static_cast<const T*>(s->x), static_cast<const T*>(t->root), tree, t->cells,
                       t->tops, t->lvl_count, flags, t->capacity);
    NB_HIP(hipGetLastError());
    return NBODY_OK;
  }
  hipLaunchKernelGGL((ot_build_init_kernel<T, D>), dim3(1), dim3(64), 0, st, n, static_cast<const T*>(s->m),
                     static_cast<const T*>(s->x), tree, t->cells, t->lvl_count);
  NB_HIP(hipGetLastError());
  // Levels [0, own) get a launch each; the levels behind them — empty unless the tree has grown deeper than the last
  // nbody_octree_info saw it (+ 2) — share ONE launch that walks them with a grid barrier and normally returns at once.  (All levels
  // behind grid barriers, nbody_octree_set_build(t, 2), is slower on this chip: an agent-scope barrier has to write back and
  // invalidate the per-XCD L2s and costs ~12 us, a dependent launch 4.6 us — N = 10^5: 0.87 against 0.55 ms per step.)
#ifdef NBODY_EXPERIMENTS
  const int own = t->build == 2 ? 0 : (t->depth_hint < kMaxLevels<D> ? t->depth_hint : kMaxLevels<D>);
#else
  constexpr int own = kMaxLevels<D>;  // build form 1: every level its own launch
#endif
  {
    uint64_t width = 1;  // a level has at most min(n/2, 2^(D*level)) cells to split
    for (int l = 0; l < own; ++l) {
      const uint64_t cap_l = width < uint64_t(n / 2 + 1) ? width : uint64_t(n / 2 + 1);
      hipLaunchKernelGGL((ot_build_level_kernel<T, D>), dim3(uint32_t((cap_l * NCH + kOBuild - 1) / kOBuild)), dim3(kOBuild), 0, st, l,
                         t->keys[fin], t->idx[fin], static_cast<const T*>(s->m), static_cast<const T*>(s->x), tree, t->cells,
                         t->lvl_count, flags, t->capacity, t->max_cells);
      NB_HIP(hipGetLastError());
      if (width < (uint64_t(1) << 40)) width *= NCH;
    }
  }
#ifdef NBODY_EXPERIMENTS
  if (own < kMaxLevels<D>) {  // at most one block per CU, and no more blocks than the widest level can use
    const uint64_t widest = uint64_t(n / 2 + 1) * NCH;
    uint32_t grid         = uint32_t((widest + kOBuild - 1) / kOBuild);
    if (grid > uint32_t(t->ncu)) grid = uint32_t(t->ncu);
    hipLaunchKernelGGL((ot_build_all_levels_kernel<T, D>), dim3(grid), dim3(kOBuild), 0, st, t->keys[fin], t->idx[fin],
                       static_cast<const T*>(s->m), static_cast<const T*>(s->x), tree, t->cells, t->lvl_count, flags, t->capacity,
                       t->max_cells, own);
    NB_HIP(hipGetLastError());
  }
#endif
  // cells still holding >= 2 bodies at the key depth (none in a typical step: the kernel then returns at once)
  hipLaunchKernelGGL((ot_build_deep_kernel<T, D>), dim3(64), dim3(64), 0, st, t->idx[fin], t->idx[1 - fin],
                     static_cast<const T*>(s->m), static_cast<const T*>(s->x), static_cast<const T*>(t->root), tree, t->cells,
                     static_cast<const ot_cell*>(nullptr), t->lvl_count, flags, t->capacity);
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

template <typename T, int D>
static int ot_tree_run(nbody_octree* t, hipStream_t st) {
  constexpr uint32_t NCH = 1u << D;
  const ot_tree<T, D> tree{static_cast<ot_group<T, D>*>(t->groups), static_cast<ot_node<T>*>(t->rootrec)};
  const uint32_t max_chunks = (t->max_cells + kMpChunk - 1) / kMpChunk;
  if ((t->build == 0 || t->build == 3) && max_chunks > kMpMaxBlocks) {  // beyond 4.2 * 10^6 bodies: one launch per level over all ranks
    for (int l = kMaxLevels<D> - 1; l >= 0; --l) {
      hipLaunchKernelGGL((ot_multipole_ranks_level_kernel<T, D>), dim3((t->max_cells + kOB - 1) / kOB), dim3(kOB), 0, st, l, tree,
                         t->cells, t->lvl_count, t->capacity, t->max_cells);
      NB_HIP(hipGetLastError());
    }
    return NBODY_OK;
  }
  // one block for the whole tree where that is one chunk's work: a single chunk of kMpChunk ranks, or — in float — of 1024 (more cells
  // than that, rare: chunk after chunk, then the crown).  Double trees of two or three chunks take the two launches below: their
  // chunks side by side on two CUs + the crown are 22 us, one after the other in one block 30 (n = 1000)
  if ((t->build == 0 || t->build == 3) && (max_chunks == 1 || (sizeof(T) == 4 && max_chunks <= kSmallChunks))) {
    hipLaunchKernelGGL((ot_multipole_small_kernel<T, D>), dim3(1), dim3((kSmallMpThreads<T, D>)), 0, st, tree, t->cells, t->lvl_count, t->later,
                       t->later_mask, t->capacity, t->max_cells, max_chunks);
    NB_HIP(hipGetLastError());
    return NBODY_OK;
  }
  if (t->build == 0 || t->build == 3) {  // rank chunks, then the cells that span chunk boundaries: in one block, or in two rounds
    hipLaunchKernelGGL((ot_multipole_chunks_kernel<T, D>), dim3(max_chunks), dim3(kMpChunk), 0, st, tree, t->cells, t->lvl_count,
                       t->later, t->later_mask, t->capacity, t->max_cells);
    NB_HIP(hipGetLastError());
    const bool rounds = max_chunks > uint32_t(kMpCrown);  // (decided by the size the tree was created for: the launches are recorded)
    if (rounds) {
      const uint32_t blocks = (max_chunks * uint32_t(kMaxLevels<D>) + kMpCrown - 1) / kMpCrown;
      hipLaunchKernelGGL((ot_multipole_round_kernel<T, D>), dim3(blocks), dim3(kMpCrown), 0, st, tree, t->cells, t->lvl_count, t->later,
                         t->later_mask, t->later2, t->later2_mask, t->max_cells);
      NB_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL((ot_multipole_crown_kernel<T, D>), dim3(1), dim3(kMpCrown), 0, st, tree, t->cells, t->lvl_count,
                       rounds ? t->later2 : t->later, rounds ? t->later2_mask : t->later_mask, t->max_cells, rounds ? 1 : 0);
    NB_HIP(hipGetLastError());
    return NBODY_OK;
  }
#ifdef NBODY_EXPERIMENTS
  const int own = t->build == 2 ? 0 : (t->depth_hint < kMaxLevels<D> ? t->depth_hint : kMaxLevels<D>);
  if (own < kMaxLevels<D>) {  // the levels below `own`, deepest first, in one launch (see ot_insert_run)
    uint32_t grid = (t->n / 2 + 1 + kOB - 1) / kOB;
    if (grid > uint32_t(t->ncu)) grid = uint32_t(t->ncu);
    hipLaunchKernelGGL((ot_multipole_all_levels_kernel<T, D>), dim3(grid), dim3(kOB), 0, st, tree, t->cells, t->lvl_count,
                       t->lvl_count + (kMaxLevels<D> + 4), t->lvl_count + (kMaxLevels<D> + 2), own, t->max_cells);
    NB_HIP(hipGetLastError());
  }
#else
  constexpr int own = kMaxLevels<D>;
#endif
  for (int l = own - 1; l >= 0; --l) {
    uint64_t width = 1;
    for (int j = 0; j < l && width < (uint64_t(1) << 40); ++j) width *= NCH;
    const uint64_t cap_l = width < uint64_t(t->n / 2 + 1) ? width : uint64_t(t->n / 2 + 1);
    hipLaunchKernelGGL((ot_multipole_level_kernel<T, D>), dim3(uint32_t((cap_l + kOB - 1) / kOB)), dim3(kOB), 0, st, l,
                       tree, t->cells, t->lvl_count, t->max_cells);
    NB_HIP(hipGetLastError());
  }
  return NBODY_OK;
}

template <typename T, int D>
static int ot_force_run(nbody_octree* t, const nbody_state* s, double theta, hipStream_t st) {
  if (s->count == 0) return NBODY_OK;
  const uint32_t* list = t->idx[t->sorted_buf];  // whole system: the sorted body indices themselves
  if (s->count < s->sz) {                        // shard window: compact the owned ones (scratch: the sort's other buffers)
    uint32_t* owned       = t->idx[1 - t->sorted_buf];
    const uint32_t nblk   = (s->sz + kOC - 1) / kOC;
    hipLaunchKernelGGL(ot_owned_count_kernel, dim3(nblk), dim3(kOC), 0, st, list, s->sz, s->first, s->count, t->hist);
    NB_HIP(hipGetLastError());
    hipLaunchKernelGGL(ot_owned_scan_kernel, dim3(1), dim3(kOC), 0, st, t->hist, nblk);
    NB_HIP(hipGetLastError());
    hipLaunchKernelGGL(ot_owned_scatter_kernel, dim3(nblk), dim3(kOC), 0, st, list, s->sz, s->first, s->count, t->hist, owned);
    NB_HIP(hipGetLastError());
    list = owned;
  }
  const uint32_t per_wave = 64u >> D;
  const uint32_t blocks   = (s->count + per_wave - 1) / per_wave;
  const uint32_t budget   = t->step_budget ? t->step_budget : t->capacity;  // a well-formed tree is left after < capacity rounds
  auto* rootrec           = static_cast<const ot_node<T>*>(t->rootrec);
#define NB_OT_LAUNCH(CNT)                                                                                                    \
  hipLaunchKernelGGL((ot_force_kernel<T, D, CNT>), dim3(blocks), dim3(64), 0, st, rootrec,                                   \
                     static_cast<const ot_group<T, D>*>(t->groups), list, s->count,                                          \
                     static_cast<const T*>(s->x), static_cast<T*>(s->a), static_cast<T>(s->c), s->first,                     \
                     static_cast<T>(theta), budget, static_cast<const T*>(t->root),                                        \
                     t->lvl_count + ((D == 3 ? kMaxLevels<3> : kMaxLevels<2>) + 2), t->counters)
  // the visit round written as ISA (ot_force_isa_kernel, ot_force_isa_f32_kernel; 2D and 3D), while 24-bit group numbers times the group
  // size stay inside 32-bit offsets; nbody_octree_set_walk(t, 1) keeps the compiler-scheduled kernel (tests compare the two bitwise)
  bool isa = false;
  {
    const char* fe = experiment_env("NBODY_OT_FORM");  // -DNBODY_EXPERIMENTS builds only
    isa = t->walk != 1 && !(fe && fe[0] == '1') &&
          uint64_t(t->max_cells) * sizeof(ot_group<T, D>) < (1ull << 32) && t->max_cells < (1u << 24);
  }
  if (t->walk == 2 && !isa) {
    set_error("octree walk: the ISA visit round needs the tree within 24-bit group numbers and 32-bit group offsets");
    return NBODY_ERR_ARG;
  }
  if (isa) {
#define NB_OT_ISA(KERN, CNT)                                                                                                  \
  hipLaunchKernelGGL((KERN<D, CNT>), dim3(blocks), dim3(64), 0, st, rootrec, static_cast<const ot_group<T, D>*>(t->groups), list, \
                     s->count, static_cast<const T*>(s->x), static_cast<T*>(s->a), static_cast<T>(s->c), s->first,            \
                     static_cast<T>(theta), budget, static_cast<const T*>(t->root),                                           \
                     t->lvl_count + ((D == 3 ? kMaxLevels<3> : kMaxLevels<2>) + 2), t->counters)
    if constexpr (sizeof(T) == 8) {
      if (t->counters_on) NB_OT_ISA(ot_force_isa_kernel, true);
      else NB_OT_ISA(ot_force_isa_kernel, false);
    } else {
      if (t->counters_on) NB_OT_ISA(ot_force_isa_f32_kernel, true);
      else NB_OT_ISA(ot_force_isa_f32_kernel, false);
    }
#undef NB_OT_ISA
  } else if (t->counters_on) NB_OT_LAUNCH(true);
  else NB_OT_LAUNCH(false);
#undef NB_OT_LAUNCH
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

}  // namespace nbody

extern "C" int nbody_octree_create(nbody_octree** out, int dtype, int dim, uint32_t n) {
  return nbody_octree_create_on(out, dtype, dim, n, -1);
}

extern "C" int nbody_octree_set_walk(nbody_octree* t, int mode) {
  NB_ARG(t != nullptr, "nbody_octree is NULL");
  NB_ARG(mode >= 0 && mode <= 2, "walk form must be 0 (auto), 1 (compiler-scheduled) or 2 (visit round as ISA), got %d", mode);
  t->walk = mode;
  return NBODY_OK;
}

extern "C" int nbody_octree_set_build(nbody_octree* t, int mode) {
  NB_ARG(t != nullptr, "nbody_octree is NULL");
#ifdef NBODY_EXPERIMENTS
  NB_ARG(mode >= 0 && mode <= 4,
         "build form must be 0 (auto), 1 (one launch per level), 2 (all levels in one launch), 3 (one pass over the sorted keys) or 4 "
         "(one launch per level the last tree used, one for the rest), got %d", mode);
#else
  NB_ARG(mode == 0 || mode == 1 || mode == 3, "build form must be 0 (auto), 1 (breadth-first, one launch per level) or 3 (one pass over the sorted keys), got %d%s",
         mode, mode == 2 || mode == 4 ? " — that form exists only in the -DNBODY_EXPERIMENTS build (make experiments)" : "");
#endif
  if (mode != t->build) t->inserted = t->have_tree = false;  // the forms lay the cell list out differently: a tree inserted by one
                                                             // is not the other's to finish (insert again after a change)
  t->build = mode;
  if (mode != 4) t->depth_hint = 64;  // 1: every level its own launch, whatever earlier trees looked like
  return NBODY_OK;
}

#ifdef NBODY_EXPERIMENTS
// wall_clock64() stamps (100 MHz) of the last ot_insert_small_kernel launch on the current device: entry, keys, sort, numbering,
// cells, deep cells (tools/time_small_insert_phases.py)
extern "C" int nbody_exp_octree_small_stamps(uint64_t* out8) {
  NB_HIP(hipDeviceSynchronize());
  NB_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(nbody::ot_small_stamps), sizeof(uint64_t) * 8));
  return NBODY_OK;
}
#endif

extern "C" int nbody_octree_set_step_budget(nbody_octree* t, uint32_t steps) {
  NB_ARG(t != nullptr, "nbody_octree is NULL");
  t->step_budget = steps;
  return NBODY_OK;
}

extern "C" int nbody_octree_create_on(nbody_octree** out, int dtype, int dim, uint32_t n, int device) {
  NB_ARG(out != nullptr, "out is NULL");
  *out = nullptr;
  NB_ARG(dtype == NBODY_F32 || dtype == NBODY_F64, "bad dtype %d", dtype);
  NB_ARG(dim == 2 || dim == 3, "bad dim %d", dim);
  NB_ARG(n >= 1 && n <= (1u << 28), "octree needs 1 <= n <= 2^28 (got %u)", n);
  int ndev = 0;
  NB_HIP(hipGetDeviceCount(&ndev));
  if (device < 0) device = current_device();
  NB_ARG(device >= 0 && device < ndev, "device %d out of range (%d HIP devices visible)", device, ndev);
  device_guard guard(device);
  auto* t  = new nbody_octree;
  t->device = device;
  t->dtype = dtype;
  t->dim   = dim;
  t->n     = n;
  t->tsz   = dtype == NBODY_F32 ? 4 : 8;
  const uint32_t nch = 1u << dim;
  const uint64_t cap = uint64_t(nch) * n < 1000 ? 1000 : uint64_t(nch) * n;  // System::max_tree_node_size (src/system.h:30)
  t->capacity        = uint32_t(cap);
  t->max_cells       = t->capacity / nch + 1;
  t->bounds_blocks   = uint32_t((uint64_t(n) * dim + kOB * 8 - 1) / (kOB * 8));
  if (t->bounds_blocks > 1024) t->bounds_blocks = 1024;
  const int maxl = dim == 3 ? kMaxLevels<3> : kMaxLevels<2>;
  auto fail = [&](hipError_t e, const char* what) {
    int r = hip_fail(e, what, __FILE__, __LINE__);
    nbody_octree_destroy(t);
    return r;
  };
#define NB_ALLOC(ptr, bytes)                                              \
  do {                                                                    \
    hipError_t e_ = hipMalloc(reinterpret_cast<void**>(&(ptr)), (bytes)); \
    if (e_ != hipSuccess) return fail(e_, "hipMalloc(" #ptr ")");         \
  } while (0)
  NB_ALLOC(t->root, t->tsz * (dim + 1));
  NB_ALLOC(t->partials, t->tsz * 2 * t->bounds_blocks);
  NB_ALLOC(t->keys[0], sizeof(uint64_t) * size_t(n));
  NB_ALLOC(t->keys[1], sizeof(uint64_t) * size_t(n));
  NB_ALLOC(t->idx[0], sizeof(uint32_t) * size_t(n));
  NB_ALLOC(t->idx[1], sizeof(uint32_t) * size_t(n));
  NB_ALLOC(t->hist, sizeof(uint32_t) * radix_sort_scratch_words(n));
  NB_ALLOC(t->rootrec, 64);
  t->group_bytes = dtype == NBODY_F32 ? (dim == 3 ? sizeof(ot_group<float, 3>) : sizeof(ot_group<float, 2>))
                                      : (dim == 3 ? sizeof(ot_group<double, 3>) : sizeof(ot_group<double, 2>));
  NB_ALLOC(t->groups, t->group_bytes * size_t(t->max_cells));
  NB_ALLOC(t->cells, sizeof(ot_cell) * size_t(t->max_cells));
  NB_ALLOC(t->lvl_count, sizeof(uint32_t) * size_t(maxl + 8));  // level counts, deep groups, flags, two grid-barrier counters, ...
  {
    const size_t nblk = size_t(n) / kLcpB + 1;
    NB_ALLOC(t->tops, sizeof(ot_cell) * (size_t(n) / 2 + 2));
    NB_ALLOC(t->lcp, size_t(n) + 2);
    NB_ALLOC(t->plocal, sizeof(uint32_t) * (size_t(n) + 1));
    NB_ALLOC(t->bsum, sizeof(uint32_t) * nblk);
    NB_ALLOC(t->bhist, sizeof(uint32_t) * nblk * size_t(maxl + 1));
    NB_ALLOC(t->rankpos, sizeof(uint32_t) * size_t(t->max_cells));
    NB_ALLOC(t->later, sizeof(uint32_t) * (size_t(t->max_cells) / kMpChunk + 2) * size_t(maxl));
    NB_ALLOC(t->later_mask, sizeof(uint32_t) * (size_t(t->max_cells) / kMpChunk + 2));
    {
      const size_t blocks2 = ((size_t(t->max_cells) / kMpChunk + 2) * size_t(maxl) + kMpCrown - 1) / kMpCrown + 1;
      NB_ALLOC(t->later2, sizeof(uint32_t) * blocks2 * size_t(maxl));
      NB_ALLOC(t->later2_mask, sizeof(uint32_t) * blocks2);
    }
  }
#undef NB_ALLOC
  if (hipError_t e = hipMemset(t->lvl_count, 0, sizeof(uint32_t) * size_t(maxl + 8)); e != hipSuccess) return fail(e, "hipMemset");
  {
    hipDeviceProp_t prop;
    if (hipError_t e = hipGetDeviceProperties(&prop, device); e != hipSuccess) return fail(e, "hipGetDeviceProperties");
    t->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
  }
  *out = t;
  return NBODY_OK;
}

extern "C" void nbody_octree_destroy(nbody_octree* t) {
  if (!t) return;
  device_guard guard(t->device);
  (void)hipFree(t->root);
  (void)hipFree(t->partials);
  (void)hipFree(t->keys[0]);
  (void)hipFree(t->keys[1]);
  (void)hipFree(t->idx[0]);
  (void)hipFree(t->idx[1]);
  (void)hipFree(t->hist);
  (void)hipFree(t->rootrec);
  (void)hipFree(t->groups);
  (void)hipFree(t->cells);
  (void)hipFree(t->tops);
  (void)hipFree(t->lcp);
  (void)hipFree(t->plocal);
  (void)hipFree(t->bsum);
  (void)hipFree(t->bhist);
  (void)hipFree(t->rankpos);
  (void)hipFree(t->later);
  (void)hipFree(t->later_mask);
  (void)hipFree(t->later2);
  (void)hipFree(t->later2_mask);
  (void)hipFree(t->lvl_count);
  (void)hipFree(t->counters);
  delete t;
}

extern "C" int nbody_octree_clear(nbody_octree* t, void* stream) {
  NB_ARG(t != nullptr, "nbody_octree is NULL");
  if (int r = check_same_device(t->device, as_stream(stream), "nbody_octree")) return r;
  // every node the build allocates is fully rewritten by it: nothing to reset but the phase flags
  t->inserted  = false;
  t->have_tree = false;
  return NBODY_OK;
}

extern "C" int nbody_octree_compute_bounds(nbody_octree* t, const nbody_state* s, void* stream) {
  if (int r = ot_check(t, s, stream)) return r;
  device_guard guard(t->device);
  int r = dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return ot_bounds_run<typename TG::type, TG::dim>(t, s, as_stream(stream));
  });
  if (r == NBODY_OK) t->have_bounds = true;
  return r;
}

extern "C" int nbody_octree_insert(nbody_octree* t, const nbody_state* s, void* stream) {
  if (int r = ot_check(t, s, stream)) return r;
  device_guard guard(t->device);
  if (!t->have_bounds) {
    set_error("nbody_octree_insert before nbody_octree_compute_bounds");
    return NBODY_ERR_STATE;
  }
  int r = dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return ot_insert_run<typename TG::type, TG::dim>(t, s, as_stream(stream));
  });
  if (r == NBODY_OK) t->inserted = true;
  return r;
}

extern "C" int nbody_octree_compute_tree(nbody_octree* t, void* stream) {
  NB_ARG(t != nullptr, "nbody_octree is NULL");
  if (int r = check_same_device(t->device, as_stream(stream), "nbody_octree")) return r;
  device_guard guard(t->device);
  if (!t->inserted) {
    set_error("nbody_octree_compute_tree before nbody_octree_insert");
    return NBODY_ERR_STATE;
  }
  int r = dispatch(t->dtype, t->dim, [&](auto tg) {
    using TG = decltype(tg);
    return ot_tree_run<typename TG::type, TG::dim>(t, as_stream(stream));
  });
  if (r == NBODY_OK) t->have_tree = true;
  return r;
}

extern "C" int nbody_octree_compute_force(nbody_octree* t, const nbody_state* s, double theta, void* stream) {
  if (int r = ot_check(t, s, stream)) return r;
  device_guard guard(t->device);
  if (!t->have_tree) {
    set_error("nbody_octree_compute_force before nbody_octree_compute_tree");
    return NBODY_ERR_STATE;
  }
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return ot_force_run<typename TG::type, TG::dim>(t, s, theta, as_stream(stream));
  });
}

extern "C" int nbody_octree_enable_counters(nbody_octree* t, int on) {
  NB_ARG(t != nullptr, "nbody_octree is NULL");
  device_guard guard(t->device);
  if (on && !t->counters) NB_HIP(hipMalloc(reinterpret_cast<void**>(&t->counters), sizeof(uint32_t) * 2 * size_t(t->n)));
  t->counters_on = on != 0;
  return NBODY_OK;
}

extern "C" int nbody_octree_info(nbody_octree* t, uint32_t* tree_size, void* root_mass, void* stream) {
  NB_ARG(t != nullptr, "nbody_octree is NULL");
  if (int r = check_same_device(t->device, as_stream(stream), "nbody_octree")) return r;
  device_guard guard(t->device);
  if (!t->inserted) {
    set_error("nbody_octree_info before nbody_octree_insert");
    return NBODY_ERR_STATE;
  }
  hipStream_t st = as_stream(stream);
  const int maxl = t->dim == 3 ? kMaxLevels<3> : kMaxLevels<2>;
  uint32_t lv[40];
  NB_HIP(hipMemcpyAsync(lv, t->lvl_count, sizeof(uint32_t) * (maxl + 3), hipMemcpyDeviceToHost, st));
  char rec[32];
  NB_HIP(hipMemcpyAsync(rec, t->rootrec, t->tsz * 4, hipMemcpyDeviceToHost, st));
  NB_HIP(hipStreamSynchronize(st));
  const uint32_t flags = lv[maxl + 2];  // set by ANY build since the last call
  if (flags != 0) NB_HIP(hipMemsetAsync(t->lvl_count + (maxl + 2), 0, sizeof(uint32_t), st));
  if (flags & kFlagDepth) {
    set_error("octree depth limit: bodies not separated after %d levels (coincident positions, or closer than root_side/2^%d)",
              kOtDeepLevels, kOtDeepLevels);
    return NBODY_ERR_STATE;
  }
  if (flags & kFlagCapacity) {
    set_error("octree node pool exhausted (capacity %u nodes = System::max_tree_node_size)", t->capacity);
    return NBODY_ERR_STATE;
  }
  if (flags & kFlagBarrier) {
    set_error("octree build: a grid barrier of the all-level kernels timed out (the blocks of one launch were not all running "
              "within seconds: is the GPU shared?); nbody_octree_set_build(t, 1) builds with one launch per level");
    return NBODY_ERR_STATE;
  }
  if (flags & kFlagStack) {
    set_error("octree walk: more pending nodes than the per-body stack holds (a tree far deeper than %d levels)", maxl);
    return NBODY_ERR_STATE;
  }
  if (flags & kFlagWalk) {
    set_error("octree walk: a body was still walking after %u visit rounds (%s)", t->step_budget ? t->step_budget : t->capacity,
              t->step_budget ? "the budget set with nbody_octree_set_step_budget" : "more than the tree has nodes: the tree is damaged");
    return NBODY_ERR_STATE;
  }
  int deepest = -1;
  for (int l = 0; l < maxl; ++l)
    if (lv[l] != 0) deepest = l;
  if (t->build == 4) t->depth_hint = deepest + 3 < maxl ? deepest + 3 : maxl;  // the levels the next builds launch one by one
  uint64_t cells = 0;
  for (int l = 0; l <= maxl + 1; ++l) cells += lv[l];  // breadth-first levels, then the groups of the deep build
  if (tree_size) *tree_size = uint32_t(1 + cells * (1u << t->dim));  // next_free_child_group (src/octree.h:152)
  if (root_mass) memcpy(root_mass, rec + 3 * t->tsz, t->tsz);         // m[0].mass()
  return NBODY_OK;
}

extern "C" int nbody_octree_read_counters(nbody_octree* t, uint32_t* host_out, size_t bytes, void* stream) {
  NB_ARG(t != nullptr && host_out != nullptr, "NULL argument");
  NB_ARG(t->counters != nullptr, "counters were never enabled");
  if (int r = check_same_device(t->device, as_stream(stream), "nbody_octree")) return r;
  device_guard guard(t->device);
  NB_ARG(bytes == sizeof(uint32_t) * 2 * size_t(t->n), "expected %zu bytes", sizeof(uint32_t) * 2 * size_t(t->n));
  NB_HIP(hipMemcpyAsync(host_out, t->counters, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  NB_HIP(hipStreamSynchronize(as_stream(stream)));
  return NBODY_OK;
}
