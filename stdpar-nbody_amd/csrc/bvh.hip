// Hilbert-sorted BVH for gfx950 — K4 bounding box, K5 Hilbert keys, K6 radix sort + gather,
// K7/K8 tree build, K9 stackless traversal.  Replaces src/bvh.h:17-325 of the reference.
//
// Data layout in HBM (per tree, sized by nbody_bvh_create):
//   bbox      T[3*D]            xmin, xmax (K4) and the grid cell size (K5's divisor)
//   keys[2]   u64[n] x2         ping-pong radix-sort key buffers
//   idx[2]    u32[n] x2         ping-pong payload (original body index); the final one is the permutation
//   hist      u32[256*(nblk+1)] per-(digit, block) counts -> exclusive offsets per digit row, + 256 digit totals
//   tmp       T[n*(4D+1)]       gather scratch for the in-place permutation of m,x,v,a,ao
//   node      rec[nnodes+nleafs]  one 8-scalar record per internal node (COM, mass, width, width^2) followed by one
//                               per body slot (position, mass), so the traversal touches exactly one aligned
//                               record per step (the reference keeps m/bw/b in three arrays and reads bodies
//                               from x and m, src/bvh.h:103-106,294-295).
//   box       T[nnodes*2D]      node AABBs, only used by the build
// Everything integer (keys, sort, permutation, traversal decisions, node visit counts) is bit-exact
// against the oracle; tree COMs/widths are bit-exact too (FP contraction is off in the build and in
// the opening test), only the accumulated force uses the fast pair math of common.hpp.
#include "common.hpp"
#include "radix_sort.hpp"
#include "to_sgpr.hpp"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace nbody {

constexpr int kB = 256;

// One 8-scalar record (32 B f32 / 64 B f64) per tree entry, so every traversal step is ONE aligned load:
//   internal node i (level l < nlevels, index (2^l - 1) + k): v[0..D-1] centre of mass, v[D] mass, v[D+1] width,
//                                                               v[D+2] width^2
//   body b (level nlevels, index (2^nlevels - 1) + b):          v[0..D-1] x[b], v[D] m[b]; absent bodies b >= N
//                                                               are zero records (mass 0 contributes exactly 0)
// i.e. the implicit complete binary tree of the reference (src/bvh.h:108-138) continued by one level: the bodies.
// A body entry is "always accepted", and the reference's ascend rules (left child -> sibling, right child ->
// parent + 1) then visit a leaf pair exactly as src/bvh.h:288-303 does — body 2k, body 2k+1, then the
// leaf-parent's right neighbour — so the traversal kernels need no separate leaf path.
template <typename T>
struct alignas(sizeof(T) * 8) tree_rec {
  T v[8];
};

}  // namespace nbody

struct nbody_bvh {
  int dtype = 0, dim = 0, device = 0;  // device: nbody_bvh_create_on's (nbody_bvh_create: the current one); every call runs there
  uint32_t n = 0, nlevels = 0, nnodes = 0;
  size_t tsz = 0, rec_bytes = 0;
  uint32_t sort_blocks = 0, bbox_blocks = 0;
  void* bbox      = nullptr;
  void* partials  = nullptr;
  uint64_t* keys[2] = {nullptr, nullptr};
  uint32_t* idx[2]  = {nullptr, nullptr};
  uint32_t* hist  = nullptr;
  void* tmp       = nullptr;
  void* node      = nullptr;
  void* box       = nullptr;
  uint32_t* counters = nullptr;
  uint32_t* order   = nullptr;  // K9 sweep: 8 lists of work items (one per XCD), see bvh_items_kernel
  uint32_t* order_n = nullptr;  // items per list
  int final_buf   = 0;  // which idx[] holds the permutation after the sort
  int traversal   = 0;  // 0 = auto (wave-cooperative when nlevels <= 26), 1 = per-lane, 2 = wave-cooperative
  int launch_order = 0; // sweep: 0 = work items (cut groups first), 1 = one block per group in index order
  double theta      = 0.5;   // the opening angle the next build writes thresholds for (the last one a traversal was asked for)
  double th2_built  = -1.0;  // theta^2 (in T) of the thresholds the records hold; -1: none.  What the HOST believes after its
                             // own eager calls; a recorded step changes the records whenever it is replayed, so:
  bool ever_recorded = false;            // a build or traversal of this tree has been recorded: eager traversals always rethreshold
  unsigned long long rec_id = 0;         // the capture whose recorded build / traversal last wrote the thresholds ...
  double rec_th2            = -1.0;      // ... and the theta^2 it wrote them for (valid inside that capture only)
  bool counters_on = false, have_bbox = false, sorted = false, built = false;
};

namespace nbody {

// ... and back in ONE launch (five hipMemcpyAsync were five dependent launches, 5 us each, per step)
template <typename T, int D>
__global__ __launch_bounds__(kB) void ungather_kernel(const T* __restrict__ tmp, uint32_t n, T* __restrict__ m, T* __restrict__ x,
                                                      T* __restrict__ v, T* __restrict__ a, T* __restrict__ ao) {
  const uint64_t e = uint64_t(blockIdx.x) * kB + threadIdx.x, nd = uint64_t(n) * D;
  if (e >= nd) return;
  x[e]  = tmp[e];
  v[e]  = tmp[nd + e];
  a[e]  = tmp[nd * 2 + e];
  ao[e] = tmp[nd * 3 + e];
  if (e < n) m[e] = tmp[nd * 4 + e];
}

// ------------------------------------------------------------------------------------------------
// K4 bounding box  (src/bvh.h:17-22, src/vec.h:382-405)
// min over i of fl(p_i - tol) == fl(min_i p_i - tol) (rounding is monotone), so the raw coordinates
// are reduced and the +-10 eps pad is applied once; the origin is part of the reduction because the
// reference's init value is the AABB of the origin.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T fmin_(T a, T b) {
  if constexpr (sizeof(T) == 4) return __builtin_fminf(a, b);
  else return __builtin_fmin(a, b);
}
template <typename T>
__device__ __forceinline__ T fmax_(T a, T b) {
  if constexpr (sizeof(T) == 4) return __builtin_fmaxf(a, b);
  else return __builtin_fmax(a, b);
}

template <typename T, int D>
__device__ __forceinline__ void block_minmax(T (&lo)[D], T (&hi)[D], T* out /* [2D] */) {
  __shared__ T red[2 * D][kB / 64];
#pragma unroll
  for (int k = 0; k < D; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = fmin_(lo[k], __shfl_xor(lo[k], off, 64));
      hi[k] = fmax_(hi[k], __shfl_xor(hi[k], off, 64));
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      red[k][wave]     = lo[k];
      red[D + k][wave] = hi[k];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      T l = red[k][0], h = red[D + k][0];
      for (int w = 1; w < kB / 64; ++w) {
        l = fmin_(l, red[k][w]);
        h = fmax_(h, red[D + k][w]);
      }
      out[k]     = l;
      out[D + k] = h;
    }
  }
}

template <typename T, int D>
__global__ __launch_bounds__(kB) void bbox_partial_kernel(const T* __restrict__ x, uint32_t n, T* __restrict__ partials) {
  T lo[D], hi[D];
#pragma unroll
  for (int k = 0; k < D; ++k) lo[k] = hi[k] = T(0);
  for (uint64_t i = uint64_t(blockIdx.x) * kB + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kB) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      T p   = x[i * D + k];
      lo[k] = fmin_(lo[k], p);
      hi[k] = fmax_(hi[k], p);
    }
  }
  block_minmax<T, D>(lo, hi, partials + uint64_t(blockIdx.x) * 2 * D);
}

template <typename T, int D>
__global__ __launch_bounds__(kB) void bbox_final_kernel(const T* __restrict__ partials, uint32_t nblk, T* __restrict__ bbox) {
#pragma clang fp contract(off)
  __shared__ T res[2 * D];
  T lo[D], hi[D];
#pragma unroll
  for (int k = 0; k < D; ++k) lo[k] = hi[k] = T(0);
  for (uint32_t b = threadIdx.x; b < nblk; b += kB) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      lo[k] = fmin_(lo[k], partials[uint64_t(b) * 2 * D + k]);
      hi[k] = fmax_(hi[k], partials[uint64_t(b) * 2 * D + D + k]);
    }
  }
  block_minmax<T, D>(lo, hi, res);
  __syncthreads();
  if (threadIdx.x == 0) {
    // tol = epsilon * 10. evaluated in double then converted (src/vec.h:389)
    const T tol          = T(double(sizeof(T) == 4 ? double(FLT_EPSILON) : DBL_EPSILON) * 10.);
    const uint32_t cells = (D == 2) ? 0xffffffffu : 0x1fffffu;  // src/bvh.h:33
#pragma unroll
    for (int k = 0; k < D; ++k) {
      T mn            = res[k] - tol;
      T mx            = res[D + k] + tol;
      bbox[k]         = mn;
      bbox[D + k]     = mx;
      bbox[2 * D + k] = (mx - mn) / T(cells);  // grid_cell_size (src/bvh.h:34), IEEE divide
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K5 Hilbert keys  (src/bvh.h:33-45, src/vec.h:266-356)
// ------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ uint64_t interleave_bits(const uint32_t (&c)[D]) {
  if constexpr (D == 2) {
    auto split = [](uint64_t q) {
      q = (q | q << 16) & 0xffff0000ffffull;
      q = (q | q << 8) & 0xff00ff00ff00ffull;
      q = (q | q << 4) & 0xf0f0f0f0f0f0f0full;
      q = (q | q << 2) & 0x3333333333333333ull;
      q = (q | q << 1) & 0x5555555555555555ull;
      return q;
    };
    return split(c[1]) | (split(c[0]) << 1);
  } else {
    auto split = [](uint64_t q) {
      q &= 0x1fffffull;
      q = (q | q << 32) & 0x1f00000000ffffull;
      q = (q | q << 16) & 0x1f0000ff0000ffull;
      q = (q | q << 8) & 0x100f00f00f00f00full;
      q = (q | q << 4) & 0x10c30c30c30c30c3ull;
      q = (q | q << 2) & 0x1249249249249249ull;
      return q;
    };
    return split(c[2]) | (split(c[1]) << 1) | (split(c[0]) << 2);
  }
}

// Skilling's transform over dims 0 and 1 only — in 3D too (the reference sets n = 2 there,
// src/vec.h:328) — with 32 (2D) or 21 (3D) bits; dim 2 is interleaved untransformed.
template <int D>
__device__ __forceinline__ uint64_t hilbert_key(uint32_t (&xx)[D]) {
  constexpr uint32_t M = (D == 2) ? (1u << 31) : (1u << 20);
  for (uint32_t Q = M; Q > 1; Q >>= 1) {
    const uint32_t P = Q - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (xx[i] & Q) {
        xx[0] ^= P;
      } else {
        uint32_t t = (xx[0] ^ xx[i]) & P;
        xx[0] ^= t;
        xx[i] ^= t;
      }
    }
  }
  xx[1] ^= xx[0];
  uint32_t t = 0;
  for (uint32_t Q = M; Q > 1; Q >>= 1)
    if (xx[1] & Q) t ^= Q - 1;
  xx[0] ^= t;
  xx[1] ^= t;
  return interleave_bits<D>(xx);
}

template <typename T, int D>
__global__ __launch_bounds__(kB) void hilbert_keys_kernel(const T* __restrict__ x, uint32_t n, const T* __restrict__ bbox,
                                                          uint64_t* __restrict__ keys, uint64_t* __restrict__ sort_copy) {
#pragma clang fp contract(off)
  uint64_t i = uint64_t(blockIdx.x) * kB + threadIdx.x;
  if (i >= n) return;
  uint32_t cell[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    T q = (x[i * D + k] - bbox[k]) / bbox[2 * D + k];  // IEEE divide, as the reference
    // cast<uint32_t> of an out-of-range value: x86-64 converts through 64 bits and keeps the low half
    cell[k] = uint32_t(static_cast<long long>(q));
  }
  const uint64_t key = hilbert_key<D>(cell);
  keys[i]      = key;  // kept for nbody_bvh_read(what = 0)
  sort_copy[i] = key;  // the buffer the sort starts from (a separate copy was one more dependent launch per step)
}

// gather all five state arrays through the permutation into tmp (then copied back)
template <typename T, int D>
__global__ __launch_bounds__(kB) void gather_kernel(const uint32_t* __restrict__ perm, uint32_t n, const T* __restrict__ m,
                                                    const T* __restrict__ x, const T* __restrict__ v, const T* __restrict__ a,
                                                    const T* __restrict__ ao, T* __restrict__ tmp) {
  uint64_t i = uint64_t(blockIdx.x) * kB + threadIdx.x;
  if (i >= n) return;
  const uint64_t o = perm[i];
  T* tx  = tmp;
  T* tv  = tmp + uint64_t(n) * D;
  T* ta  = tmp + uint64_t(n) * D * 2;
  T* tao = tmp + uint64_t(n) * D * 3;
  T* tm  = tmp + uint64_t(n) * D * 4;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    tx[i * D + k]  = x[o * D + k];
    tv[i * D + k]  = v[o * D + k];
    ta[i * D + k]  = a[o * D + k];
    tao[i * D + k] = ao[o * D + k];
  }
  tm[i] = m[o];
}

// ------------------------------------------------------------------------------------------------
// K7 / K8 tree build  (src/bvh.h:175-244)
// ------------------------------------------------------------------------------------------------
template <typename T, int D>
__device__ __forceinline__ T node_width(const T* b) {  // src/bvh.h:140-144, std::max
  T w = b[D] - b[0];
#pragma unroll
  for (int k = 1; k < D; ++k) {
    T l = b[D + k] - b[k];
    w   = (w < l) ? l : w;
  }
  return w;
}

// The opening test as ONE compare (K9's sweep).  The reference accepts a node iff  w2 < fl(theta^2 * d2)  (src/bvh.h:246-248;
// w2 = width * width rounded once, the product rounded once).  fl(theta^2 * d2) is monotone non-decreasing in d2 >= 0, so the
// set of distances that accept is an up-set of the doubles: with dmin its smallest member (found below by bisection on the bit
// pattern, which orders non-negative floating-point numbers) and v the number just below dmin,
//     w2 < fl(theta^2 * d2)   <=>   v < d2        for every d2 >= 0,
// bit for bit and for every theta — the record carries v (slot D + 3), and the sweep tests !(v >= d2), which like the product
// form accepts a NaN distance.  No distance accepts (theta = 0, w2 = inf): v = +inf.  Body entries carry -1 (always accepted),
// a NaN width stays NaN (always accepted, as !(NaN >= x) was).  The product is evaluated here exactly as the per-lane form and
// the reference evaluate it: one multiply in T, no contraction.
template <typename T>
__host__ __device__ __forceinline__ T open_threshold(T w2, T th2) {
#pragma clang fp contract(off)
  using U = typename std::conditional<sizeof(T) == 8, unsigned long long, uint32_t>::type;
  constexpr U kInf = sizeof(T) == 8 ? U(0x7ff0000000000000ull) : U(0x7f800000u);
  if (w2 < T(0)) return T(-1);
  if (!(w2 == w2)) return w2;
  auto accepts = [&](U bits) { return w2 < th2 * __builtin_bit_cast(T, bits); };
  if (!accepts(kInf)) return __builtin_bit_cast(T, kInf);
  U lo = 0, hi = kInf;  // !accepts(lo) (theta^2 * 0 = 0 <= w2), accepts(hi)
  const T q = w2 / th2;  // dmin is within a few ulps of the quotient unless the product leaves the normal range
  if (q == q && q < __builtin_bit_cast(T, kInf)) {
    const U qb = __builtin_bit_cast(U, q);
    const U a = qb > U(8) ? qb - U(8) : U(0), b = qb + U(8) < kInf ? qb + U(8) : kInf;
    if (accepts(a)) hi = a;
    else lo = a;
    if (b < hi) {
      if (accepts(b)) hi = b;
      else lo = b;
    }
  }
  while (hi - lo > U(1)) {
    const U mid = lo + (hi - lo) / U(2);
    if (accepts(mid)) hi = mid;
    else lo = mid;
  }
  return __builtin_bit_cast(T, lo);
}

template <typename T, int D>
__global__ __launch_bounds__(kB) void build_leaf_level_kernel(const T* __restrict__ m, const T* __restrict__ x, uint32_t nbodies,
                                                              uint32_t first, uint32_t count, uint32_t nnodes,
                                                              tree_rec<T>* __restrict__ node, T* __restrict__ box, T th2) {
#pragma clang fp contract(off)
  uint32_t li = blockIdx.x * kB + threadIdx.x;
  if (li >= count) return;
  const uint32_t i  = first + li;
  const uint64_t bl = uint64_t(li) * 2, br = bl + 1;
  const T tol = T(double(sizeof(T) == 4 ? double(FLT_EPSILON) : DBL_EPSILON) * 10.);
  tree_rec<T> r, ba, bb;  // this node and its two body entries
#pragma unroll
  for (int k = 0; k < 8; ++k) r.v[k] = ba.v[k] = bb.v[k] = T(0);
  T* b = box + uint64_t(i) * 2 * D;
  if (bl >= nbodies) {  // dead node (src/bvh.h:185-188): mass 0; box/width are don't-care in the reference, 0 here
#pragma unroll
    for (int k = 0; k < 2 * D; ++k) b[k] = T(0);
  } else if (br >= nbodies) {  // single body (src/bvh.h:190-194)
#pragma unroll
    for (int k = 0; k < D; ++k) {
      T p      = x[bl * D + k];
      r.v[k]   = p;
      ba.v[k]  = p;
      b[k]     = p - tol;
      b[D + k] = p + tol;
    }
    r.v[D]     = m[bl];
    ba.v[D]    = r.v[D];
    r.v[D + 1] = node_width<T, D>(b);
  } else {  // two bodies (src/bvh.h:195-205)
    const T ml = m[bl], mr = m[br];
    const T mass = ml + mr;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      T p0 = x[bl * D + k], p1 = x[br * D + k];
      T com    = ml * p0 + mr * p1;
      r.v[k]   = com / mass;
      ba.v[k]  = p0;
      bb.v[k]  = p1;
      b[k]     = fmin_(p0, p1) - tol;
      b[D + k] = fmax_(p0, p1) + tol;
    }
    r.v[D]     = mass;
    ba.v[D]    = ml;
    bb.v[D]    = mr;
    r.v[D + 1] = node_width<T, D>(b);
  }
  r.v[D + 2]                          = r.v[D + 1] * r.v[D + 1];  // the product the opening test needs, rounded once
  r.v[D + 3]                          = open_threshold<T>(r.v[D + 2], th2);
  ba.v[D + 2] = bb.v[D + 2] = T(-1);  // a body entry is always accepted: -1 < theta^2 * d2 holds for every d2 >= 0
  ba.v[D + 3] = bb.v[D + 3] = T(-1);  // ... and so does -1 < d2
  node[i]                             = r;
  node[uint64_t(nnodes) + 2 * li]     = ba;  // body level continues the level-order numbering: nnodes = 2^nlevels - 1
  node[uint64_t(nnodes) + 2 * li + 1] = bb;
}

// Levels [l_hi ... l_lo] (descending), one launch.  Levels with more than one block's worth of nodes
// are launched one per kernel (l_hi == l_lo); the top of the tree (<= kB nodes per level) is built
// by a single block looping over levels with a barrier in between.
template <typename T, int D>
__global__ __launch_bounds__(kB) void build_upper_levels_kernel(int l_hi, int l_lo, tree_rec<T>* __restrict__ node,
                                                                T* __restrict__ box, T th2) {
#pragma clang fp contract(off)
  for (int l = l_hi; l >= l_lo; --l) {
    // block b owns nodes [b, b + 1) * count / blocks of every level it walks: the subtree chunk under its nodes of level l_lo
    const uint32_t first = (1u << l) - 1u, count = 1u << l, share = count / gridDim.x;
    for (uint32_t q = threadIdx.x; q < share; q += kB) {
      const uint32_t li = blockIdx.x * share + q;
      const uint32_t i  = first + li;
      const uint32_t bl = li * 2 + first + count, br = bl + 1;
      const tree_rec<T> ml = node[bl];
      const tree_rec<T> mr = node[br];
      tree_rec<T> r;
#pragma unroll
      for (int k = 0; k < 8; ++k) r.v[k] = T(0);
      T* b = box + uint64_t(i) * 2 * D;
      if (!(ml.v[D] != T(0))) {  // left dead (src/bvh.h:225-228): copy left monopole (all zeros)
#pragma unroll
        for (int k = 0; k <= D; ++k) r.v[k] = ml.v[k];
#pragma unroll
        for (int k = 0; k < 2 * D; ++k) b[k] = T(0);
      } else if (!(mr.v[D] != T(0))) {  // right dead (src/bvh.h:230-233): copy left node entirely
        r = ml;
#pragma unroll
        for (int k = 0; k < 2 * D; ++k) b[k] = box[uint64_t(bl) * 2 * D + k];
      } else {  // src/bvh.h:234-241
        const T mass = ml.v[D] + mr.v[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          r.v[k]   = (ml.v[D] * ml.v[k] + mr.v[D] * mr.v[k]) / mass;
          b[k]     = fmin_(box[uint64_t(bl) * 2 * D + k], box[uint64_t(br) * 2 * D + k]);
          b[D + k] = fmax_(box[uint64_t(bl) * 2 * D + D + k], box[uint64_t(br) * 2 * D + D + k]);
        }
        r.v[D]     = mass;
        r.v[D + 1] = node_width<T, D>(b);
        r.v[D + 2] = r.v[D + 1] * r.v[D + 1];
      }
      r.v[D + 3] = open_threshold<T>(r.v[D + 2], th2);
      node[i] = r;
    }
    if (l > l_lo) {
      __threadfence_block();
      __syncthreads();  // the next level up reads what this block wrote (nothing of any other block's)
    }
  }
}

// a traversal with another opening angle than the tree was built for: the thresholds alone are rewritten
template <typename T, int D>
__global__ __launch_bounds__(kB) void rethreshold_kernel(tree_rec<T>* __restrict__ node, uint32_t nnodes, T th2) {
  const uint32_t i = blockIdx.x * kB + threadIdx.x;
  if (i < nnodes) node[i].v[D + 3] = open_threshold<T>(node[i].v[D + 2], th2);
}

// ------------------------------------------------------------------------------------------------
// K9 traversal  (src/bvh.h:246-324)
// ------------------------------------------------------------------------------------------------
// One entry against one body, shared by both scheduling forms (so they stay bitwise identical):
//   d = xs - p          the reference's operand order in can_approximate / dist2 (src/vec.h:232-240);
//   d2                  dist2 summed exactly as the reference does — separate multiply and add, ascending k — because the
//                       opening test  bw*bw < theta^2 * d2  (src/bvh.h:246-248) must decide bit for bit like it;
//   term                m * (p - xs) / dist3 (src/bvh.h:297,308) = -w * d.  In f64 the weight comes from the test's own d2
//                       (one rounding per product instead of fused: 1e-16 relative, the force is tolerance parity) through
//                       K1's reciprocal-free weight_far; only if some lane of the wave holds an accepted entry closer than
//                       2^-8 (the body's own leaf, coincident bodies) does the wave also evaluate the guarded form from the
//                       fused r2 (+tiny, so that d = 0 gives exactly 0) and each lane keeps what its own d2 asks for.
//                       f32 keeps weight() on the fused r2 (1-ulp hardware seeds, nothing to gain).
template <typename T, int D>
__device__ __forceinline__ T dist2_ref(const T (&d)[D]) {
#pragma clang fp contract(off)
  T d2 = d[0] * d[0];  // 0 + d0*d0 of the reference: exact
#pragma unroll
  for (int k = 1; k < D; ++k) d2 = d2 + d[k] * d[k];
  return d2;
}

// `take_mask` = ballot(take), handed in by the caller: wave-uniform conditions are kept as scalar masks built from
// ballots of plain compares (a ballot of a derived boolean costs hipcc a v_cndmask + v_cmp round trip).
template <typename T, int D>
__device__ __forceinline__ void tree_accumulate(bool take, uint64_t take_mask, T (&acc)[D], const T (&d)[D], T d2, T m,
                                                const pair_consts<T>& pc) {
  T w;
  if constexpr (sizeof(T) == 8) {
    w                 = pair_math<T>::weight_far(d2, m, pc.k15, pc.k1875);
    const bool close  = uint32_t(__builtin_bit_cast(unsigned long long, d2) >> 32) < pair_math<T>::near_hi;
    if (__builtin_expect((__builtin_amdgcn_ballot_w64(close) & take_mask) != 0ull, 0)) {
      T r2 = pair_math<T>::tiny;
#pragma unroll
      for (int k = 0; k < D; ++k) r2 = __builtin_elementwise_fma(d[k], d[k], r2);
      const T wn = pair_math<T>::template weight<3>(r2, m);
      w          = close ? wn : w;
    }
  } else {
    T r2 = pair_math<T>::tiny;
#pragma unroll
    for (int k = 0; k < D; ++k) r2 = __builtin_elementwise_fma(d[k], d[k], r2);
    w = pair_math<T>::weight(r2, m);
  }
  w = take ? w : T(0);  // predicated weight, not control flow: the accumulators stay in place (see pair_accumulate_if)
#pragma unroll
  for (int k = 0; k < D; ++k) acc[k] = __builtin_elementwise_fma(-w, d[k], acc[k]);
}

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2), while
// Hilbert-neighbour waves walk nearly the same near-field records.  Remapping the block id so that each XCD works
// on one contiguous range of bodies lets neighbours share their XCD's 4 MiB L2.  Bijective for any grid size
// (MI355X_MICROARCH: placement is a speed matter only, never correctness).
__device__ __forceinline__ uint32_t xcd_contiguous_block(uint32_t b, uint32_t nblocks) {
  const uint32_t q = nblocks / 8u, r = nblocks % 8u, xcd = b % 8u, slot = b / 8u;
  return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + slot;
}

// Per-lane form: the reference's loop, one independent stackless walk per lane.  With bodies as tree entries every
// iteration is the same short program — fetch, (test), accumulate if accepted, move — so a lane standing on a body
// no longer makes the whole wave run a separate two-body path.  Software-pipelined: a step's DECISION (opening
// test -> next entry) is the serial chain, the accepted term's arithmetic is not, so each iteration decides,
// requests the next record, and only then accumulates the current one.
template <typename T, int D, bool COUNT>
__device__ __forceinline__ void bvh_walk_lane(const tree_rec<T>* node, uint32_t local, T* __restrict__ a, const T* __restrict__ x, T c,
                                              uint32_t sz, uint32_t first, T theta2, uint32_t nlevels, uint32_t* __restrict__ counters) {
  const uint32_t i = first + local;
  const pair_consts<T> pc;
  T xs[D], acc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    xs[k]  = x[uint64_t(i) * D + k];
    acc[k] = T(0);
  }
  uint32_t tree_index = 0, level = 0, covered = 0;
  uint32_t c_nodes = 0, c_leaf = 0, c_mono = 0, c_body = 0;
  tree_rec<T> rec = node[0];

  while (covered < sz) {
    const bool body = level == nlevels;
    T d[D];
#pragma unroll
    for (int k = 0; k < D; ++k) d[k] = xs[k] - rec.v[k];
    const T d2 = dist2_ref<T, D>(d);
    // decision: a body is always taken (src/bvh.h:288-300), a node if it passes the opening test (src/bvh.h:306)
    bool take;
    {
#pragma clang fp contract(off)
      // the reference's  bw*bw < theta^2*d2  for numbers; a NaN distance accepts, so that no walk can descend below the
      // body level (records carry width^2 = -1: always taken) whatever the state holds
      take = !(rec.v[D + 2] >= theta2 * d2);
    }
    uint32_t n_index, n_level = level, n_cov = covered;
    if (take) {
      n_cov = covered + (1u << (nlevels - level));
      if ((tree_index - 1u) & 1u) {  // right child -> parent + 1 (src/bvh.h:272-281, :115-120); the root counts as one
        n_index = (level == 0) ? 1u : ((1u << (level - 1)) - 1u) + (tree_index - ((1u << level) - 1u)) / 2u + 1u;
        n_level = level - 1u;
      } else {  // left child -> sibling
        n_index = tree_index + 1u;
      }
    } else {  // descend to the left child (src/bvh.h:126-130,283-286)
      const uint32_t f = (1u << level) - 1u;
      n_index          = (tree_index - f) * 2u + f + (1u << level);
      n_level          = level + 1u;
    }
    // next record, requested before the current one is accumulated
    tree_rec<T> nrec = rec;
    if (n_cov < sz) nrec = node[n_index];
    if (COUNT) {
      if (body) {
        c_body += (covered != i);
        c_leaf += !(covered & 1u);  // one leaf visit per body pair, counted at its first body
      } else {
        ++c_nodes;
        c_mono += take;
      }
    }
    const uint64_t take_mask = __builtin_amdgcn_ballot_w64(take);
    if (take_mask != 0ull) tree_accumulate<T, D>(take, take_mask, acc, d, d2, rec.v[D], pc);  // the self pair adds exactly 0
    rec        = nrec;
    tree_index = n_index;
    level      = n_level;
    covered    = n_cov;
  }
#pragma unroll
  for (int k = 0; k < D; ++k) a[uint64_t(local) * D + k] = c * acc[k];
  if (COUNT) {
    counters[uint64_t(i) * 4 + 0] = c_nodes;
    counters[uint64_t(i) * 4 + 1] = c_leaf;
    counters[uint64_t(i) * 4 + 2] = c_mono;
    counters[uint64_t(i) * 4 + 3] = c_body;
  }
}

template <typename T, int D, bool COUNT>
__global__ __launch_bounds__(64) void bvh_force_kernel(const tree_rec<T>* __restrict__ node, T* __restrict__ a, const T* __restrict__ x,
                                                       T c, uint32_t sz, uint32_t first, uint32_t count, T theta2,
                                                       uint32_t nlevels, uint32_t* __restrict__ counters) {
  const uint32_t local = xcd_contiguous_block(blockIdx.x, gridDim.x) * 64 + threadIdx.x;
  if (local >= count) return;
  bvh_walk_lane<T, D, COUNT>(node, local, a, x, c, sz, first, theta2, nlevels, counters);
}

// Float trees that fit LDS (all records, bodies included, 32 bytes each: up to 2048 bodies — the reference's default run is 1000):
// every block copies the tree into its LDS and its lanes walk it there; same arithmetic, same bits.  Measured at n = 1000
// (profiles/r04/tiny_trees_kernel_stats.txt): 75 -> 53 us in float — the walk is a chain of dependent record fetches, but the
// decision's own arithmetic is most of an iteration — and 128 -> 180 us in double (64-byte records: the lanes' four 16-byte
// reads land on 4 of the 16 bank groups), so double keeps the walk over global memory.
constexpr int kTreeLdsThreads    = 256;
constexpr uint32_t kTreeLdsBytes = 144u << 10;  // of the CU's 160 KB
template <typename T, int D, bool COUNT>
__global__ __launch_bounds__(kTreeLdsThreads) void bvh_force_lds_kernel(const tree_rec<T>* __restrict__ node, uint32_t nrec, T* __restrict__ a,
                                                                        const T* __restrict__ x, T c, uint32_t sz, uint32_t first,
                                                                        uint32_t count, T theta2, uint32_t nlevels,
                                                                        uint32_t* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bvh_tree_lds[];
  {
    const uint4* src   = reinterpret_cast<const uint4*>(node);
    uint4* dst         = reinterpret_cast<uint4*>(bvh_tree_lds);
    const uint32_t n16 = nrec * uint32_t(sizeof(tree_rec<T>) / 16);
    for (uint32_t q = threadIdx.x; q < n16; q += kTreeLdsThreads) dst[q] = src[q];
  }
  __syncthreads();
  const uint32_t local = blockIdx.x * kTreeLdsThreads + threadIdx.x;
  if (local >= count) return;
  bvh_walk_lane<T, D, COUNT>(reinterpret_cast<const tree_rec<T>*>(bvh_tree_lds), local, a, x, c, sz, first, theta2, nlevels, counters);
}

// ------------------------------------------------------------------------------------------------
// K9, wave-cooperative sweep (the idea; bvh_force_sweep_isa_kernel below is the shipped form of it).
//
// Every lane still performs exactly the reference's own sequence of node tests / body terms, in its own order,
// with its own decisions (results and counters are bitwise those of bvh_force_kernel).  What changes is WHEN: a
// lane's state is the key (covered, level) of the entry it must visit next — `covered` = index of the entry's
// first body — and along any lane's walk that key only grows in (covered, level) lexicographic order, which is
// DFS pre-order.  The wave therefore sweeps the UNION of its 64 lanes' entries once, in key order: at each step
// the record is wave-uniform (one scalar load instead of 64 divergent gathers), lanes whose key equals the
// current one take part, the others wait.  Lanes cannot drift apart.  Union per wave at config 4: ~7.0k entries
// vs ~4.3k per lane, so with few waves in flight (small N) the per-lane form is faster.
// This form is bound by its serial chain and by issue slots — of the FP64 pipe and of the CU's one scalar unit,
// which serves its four SIMDs in turn — so the step is written to be short in both:
//   * the whole 64-byte record arrives with ONE s_load_dwordx16 whose address is base + a byte offset kept in an
//     SGPR (off = 64 * level-order index: child = 2*off + 64, sibling = off + 64, parent + 1 = off / 2);
//   * the sweep position is the packed key cur = covered << 5 | level plus span = 32 << (levels below the entry); both
//     successors are one or two scalar operations away (descend: cur + 1; ascend: cur + span - right);
//   * body records carry width^2 = -1, so "a body is always taken" falls out of the opening test itself (no level check);
//   * the opening test's differences are reused by the accepted term, whose weight needs no reciprocal (tree_accumulate).
// Measured and rejected earlier: fetching both possible successors speculatively (scalar loads return out of order,
// so every step waits for the not-taken, often cold, one: 26 vs 14 ms); a wave-private LDS window over a
// pre-order copy of the tree filled by LDS-DMA (15.6 ms: refills cost more than the misses they replace);
// pipelining the next record's load ahead of the accumulation as in the per-lane kernel.
// Packed key = covered << 5 | level, so this form needs nlevels <= 26.
// ------------------------------------------------------------------------------------------------
// Work items of the sweep.  A wave's sweep is 2.9k-14.2k steps long (config 4: mean 7.4k, p90 9.0k); the longest belong to
// the ~1-2 % of groups whose 64 consecutive bodies straddle a jump of the reference's key order (bounding-box diagonals of
// 20-70 length units against a median of 5): their unions are 2-3.5x a body's walk.  A SIMD slot runs only about two waves
// per launch and blocks start in index order, so a long sweep that starts late is the tail of the kernel, and the longest
// one is its critical path (14k steps at ~1300 cycles each is the whole 8 ms).  The jump shows in the sorted keys: the
// highest bit in which two keys differ.  Per XCD range of groups (xcd_contiguous_block's ranges, so the L2 neighbourhood of
// everything else is kept) bvh_items_kernel
//   * takes the len / 16 groups whose first and last key differ in the highest bit,
//   * cuts each of them at its largest internal jump into two work items (lanes [0, cut) and [cut, 64) of the same group:
//     two compact half-groups sweep two short unions side by side instead of one long one), and
//   * lists those items first, highest jump first; every other group follows as one item in index order.
// item = group | first lane << 20 | last lane << 26.  Nothing but grouping and start order changes: a body's result depends
// on the tree and that body alone.
constexpr uint32_t kOrderMax = 8192;  // groups per XCD range one block handles (N <= 4.2M bodies); more: plain index order
constexpr uint32_t kSplitMax = 2048;  // groups per XCD range that may be cut
__host__ __device__ __forceinline__ void xcd_range(uint32_t xcd, uint32_t nblocks, uint32_t* start, uint32_t* len) {
  const uint32_t q = nblocks / 8u, r = nblocks % 8u;
  *start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  *len   = q + (xcd < r ? 1u : 0u);
}
__host__ __device__ __forceinline__ uint32_t split_budget(uint32_t len, uint32_t den = 16u) {
  const uint32_t b = len / den;
  return len > kOrderMax ? 0u : (b < kSplitMax ? b : kSplitMax);
}
__device__ __forceinline__ uint32_t pack_item(uint32_t group, uint32_t lo, uint32_t hi) { return group | (lo << 20) | (hi << 26); }

// items of XCD x live at items[x * stride ...], nitems[x] of them; stride = longest range + its budget
__global__ __launch_bounds__(1024) void bvh_items_kernel(const uint64_t* __restrict__ sorted_keys, uint32_t first, uint32_t count,
                                                          uint32_t* __restrict__ items, uint32_t* __restrict__ nitems,
                                                          uint32_t nblocks, uint32_t stride, uint32_t den) {
  __shared__ uint8_t lvl[kOrderMax];
  __shared__ uint32_t hist[65], tsum[1024], outl[kSplitMax];
  __shared__ uint32_t thr, nout;
  uint32_t start, len;
  xcd_range(blockIdx.x, nblocks, &start, &len);
  uint32_t* out    = items + blockIdx.x * stride;
  const uint32_t t = threadIdx.x;
  if (len > kOrderMax || len == 0) {
    for (uint32_t i = t; i < len; i += blockDim.x) out[i] = pack_item(start + i, 0, 63);
    if (t == 0) nitems[blockIdx.x] = len;
    return;
  }
  if (t < 65) hist[t] = 0;
  __syncthreads();
  auto jump = [](uint64_t a, uint64_t b) { return a == b ? 0u : 64u - uint32_t(__builtin_clzll(a ^ b)); };
  for (uint32_t i = t; i < len; i += blockDim.x) {
    const uint32_t b0 = (start + i) * 64u, b1 = min(b0 + 63u, count - 1u);
    const uint32_t l  = jump(sorted_keys[first + b0], sorted_keys[first + b1]);
    lvl[i]            = uint8_t(l);
    atomicAdd(&hist[l], 1u);
  }
  __syncthreads();
  if (t == 0) {  // the smallest level such that at most split_budget(len) groups lie at or above it
    const uint32_t budget = split_budget(len, den);
    uint32_t acc = 0, l = 64;
    while (l > 0 && acc + hist[l] <= budget) acc += hist[l--];
    thr = l + 1;
  }
  __syncthreads();
  const uint32_t th = thr;
  // stable partition: the groups to cut are collected, everything else goes behind their 2 * nout items in index order
  const uint32_t per = (len + blockDim.x - 1) / blockDim.x, lo = min(len, t * per), hi = min(len, lo + per);
  uint32_t mine = 0;
  for (uint32_t i = lo; i < hi; ++i) mine += lvl[i] >= th;
  tsum[t] = mine;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (uint32_t k = 0; k < blockDim.x; ++k) {
      const uint32_t v = tsum[k];
      tsum[k]          = run;
      run += v;
    }
    nout                = run;
    nitems[blockIdx.x]  = len + run;
  }
  __syncthreads();
  uint32_t before          = tsum[t];
  const uint32_t total_out = nout;
  for (uint32_t i = lo; i < hi; ++i) {
    if (lvl[i] >= th) outl[before++] = i;
    else out[2 * total_out + (i - before)] = pack_item(start + i, 0, 63);
  }
  __syncthreads();
  for (uint32_t k = t; k < total_out; k += blockDim.x) {
    const uint32_t i = outl[k];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < total_out; ++j) rank += lvl[outl[j]] > lvl[i] || (lvl[outl[j]] == lvl[i] && j < k);
    // the cut: behind the largest jump between consecutive keys of the group (first one wins)
    const uint32_t b0 = (start + i) * 64u, nb = min(64u, count - b0);
    uint32_t cut = nb / 2u, best = 0;
    uint64_t prev = sorted_keys[first + b0];
    for (uint32_t j = 1; j < nb; ++j) {
      const uint64_t cur = sorted_keys[first + b0 + j];
      const uint32_t l   = jump(prev, cur);
      if (l > best) {
        best = l;
        cut  = j;
      }
      prev = cur;
    }
    if (cut == 0 || cut >= nb) cut = nb > 1 ? nb / 2u : 1u;
    out[2 * rank]     = pack_item(start + i, 0, cut - 1u);
    out[2 * rank + 1] = nb > cut ? pack_item(start + i, cut, 63) : pack_item(start + i, 63, 0);  // empty range: no lane
  }
}

#ifdef NBODY_EXPERIMENTS
#include "experiments/bvh_forms.inc"  // traversal modes 3, 4, 6: measured forms that are not shipped
#endif

// ------------------------------------------------------------------------------------------------
// K9, the sweep's step program written out as ISA (f64 described; the f32 text follows the same skeleton).
//
// The sweep above is bound by the number of instructions a step issues — vector AND scalar: the CU's one scalar unit serves
// its four SIMDs in turn — so the same step (same tests in the same order per body, bitwise the same results and counters) is
// written for the machine.  Round 4 form: 40.3 instructions per step measured at config 4 (23.0 VALU + 12.1 SALU + 4.1 branch +
// 1.15 SMEM; round 2/3: 48.0 = 23.4 + 18.1 + 5.5 + 1; hipcc's schedule of the C++: 61):
//   * execution masks instead of selects: v_cmpx puts the lanes standing on the current entry in EXEC, they are moved to its
//     left child (key + 1), differences / d2 run for them alone, a second v_cmpx narrows EXEC to the lanes that accept, and
//     `key = skip key` plus the accepted term run under it;
//   * the opening test is ONE compare against a threshold the tree build stores in the record (open_threshold above): for every
//     width^2 and theta^2 there is a largest v with !(w2 < fl(theta^2 * d2)) for all d2 <= v (the product is monotone in d2), so
//     `w2 < fl(theta^2 d2)` <=> `v < d2` bit for bit.  Bodies carry v = -1 (always accepted); a NaN distance accepts, as
//     before, so no walk can descend below the body level whatever the state holds.  The per-lane form keeps the reference's
//     product; the tests hold the two forms equal decision by decision;
//   * two record blocks (s[40:55], s[56:71]) and two sets of position registers used in turn by two copies of the step: the
//     next record is requested into the other block as soon as the decision is made and the accepted term reads the mass
//     straight from its own block (no copies), every successor is computed directly into the other set (no moves), the skip key
//     `ka = cur + (span - 1) + "is a left child"` (one s_addc_u32) IS the other copy's `cur`;
//   * "somebody accepted" is s_cbranch_execz on the v_cmpx result (no scalar compare), on a skip everybody did (no test);
//     the end of the walk is tested on the skip path only (a descent cannot end it: cur + 1 < (sz << 5) - 1);
//   * a skip requests the record of the skip key BEFORE it evaluates the current one; if some lane waits behind that key (18 %
//     of the skips at config 4) the request is repeated for the right record once the first has landed;
//   * a scalar compare on the threshold's high word decides whether the wave looks for near pairs at all.
// Measured and not kept (round 4): tracking the lanes of a descent in EXEC alone (match &= ~take, no key compare and no key
// update for the lanes that open an entry) is WRONG as it stands — lanes that came up from the subtree before wait exactly at
// left children (the reference's parent + 1 ascent lands there) and must join the sweep when it descends to them; with those
// waiters found by one compare after each descent it is bitwise right, one VALU instruction shorter per step and exactly as
// fast (6.61-6.66 against 6.60-6.62 ms): what is left is not the instruction count (see DESIGN).
// Hazards are handled by hand inside the block (gfx940 rules: a transcendental's result needs one instruction before its
// first use, an SGPR written by a VALU instruction two before a VALU instruction reads it; SALU readers interlock) and
// checked statically in the built code object (tools/check_isa_hazards.py, tools/check_smem_pipeline.py).
// ------------------------------------------------------------------------------------------------
#define K9_APPLY(M, ...) M(__VA_ARGS__)
#define K9_KEEP(...) __VA_ARGS__
#define K9_DROP(...) ""

// per-body counters (tests): under EXEC = lanes on the entry.  X = "A" / "B": the copy's position registers.
#define K9_COUNT_A(X)                                                                                                     \
  "s_cmp_eq_u32 %[sp" X "], 31\n\t"                                                                                       \
  "s_cbranch_scc1 .LK9cb" X "%=\n\t"                                                                                      \
  "v_add_u32_e32 %[cn], 1, %[cn]\n\t"                                                                                     \
  "s_mov_b32 %[cinc], 1\n\t"                                                                                              \
  "s_branch .LK9cc" X "%=\n"                                                                                              \
  ".LK9cb" X "%=:\n\t"                                                                                                    \
  "s_lshr_b32 %[t1], %[cur" X "], 5\n\t"                                                                                  \
  "v_cmp_ne_u32_e32 vcc, %[t1], %[bi]\n\t"                                                                                \
  "s_not_b32 %[t2], %[t1]\n\t"                                                                                            \
  "s_and_b32 %[t2], %[t2], 1\n\t"                                                                                         \
  "v_addc_co_u32_e32 %[cb], vcc, 0, %[cb], vcc\n\t"                                                                       \
  "v_add_u32_e32 %[cl], %[t2], %[cl]\n\t"                                                                                 \
  "s_mov_b32 %[cinc], 0\n"                                                                                                \
  ".LK9cc" X "%=:\n\t"
#ifdef NBODY_EXPERIMENTS
// experiments library: the sweep counts its steps (one scalar add; SCC is free here, K9_COUNT_A compares at the same place) for
// the timeline tools (tools/k9_timeline.py, tools/k9_duration_carryover.py: is a union's LENGTH a usable start-order predictor?)
#define K9_NOCOUNT_A(X) "s_add_u32 %[cinc], %[cinc], 1\n\t"
#define K9_CINC "+s"
#else
#define K9_NOCOUNT_A(X) ""
#define K9_CINC "=&s"
#endif
#define K9_COUNT_B "v_add_u32_e32 %[cm], %[cinc], %[cm]\n\t"

// ---- f64 pieces.  R = first SGPR of the record block as a number token pasted by the callers below.
// the accepted term's weight for r2 >= 2^-16 (pair_math<double>::weight_far, same operations in the same order); RM = mass
#define K9_FAR(RM)                                                                                                        \
  "v_mul_f64 %[y2], %[y], %[y]\n\t"                                                                                       \
  "v_fma_f64 %[e], -%[r2], %[y2], 1.0\n\t"                                                                                \
  "v_mul_f64 %[y], %[y], %[y2]\n\t"                                                                                       \
  "v_fma_f64 %[p], %[k1875], %[e], %[k15]\n\t"                                                                            \
  "v_ldexp_f64 %[q], -%[y], %[m52]\n\t"                                                                                   \
  "v_mul_f64 %[y], %[y], " RM "\n\t"                                                                                      \
  "v_fmac_f64_e32 %[q], %[p], %[e]\n\t"                                                                                   \
  "v_fmac_f64_e32 %[y], %[y], %[q]\n\t"
// differences, d2 summed as the reference does, the opening test: EXEC = lanes that accept (also in %[take])
#define K9_TEST_F64(Z, RX, RY, RZ, RV)                                                                                    \
  "v_add_f64 %[d0], %[xs0], -" RX "\n\t"                                                                                  \
  "v_add_f64 %[d1], %[xs1], -" RY "\n\t"                                                                                  \
  Z("v_add_f64 %[d2], %[xs2], -" RZ "\n\t")                                                                               \
  "v_mul_f64 %[r2], %[d0], %[d0]\n\t"                                                                                     \
  "v_mul_f64 %[t], %[d1], %[d1]\n\t"                                                                                      \
  "v_add_f64 %[r2], %[r2], %[t]\n\t"                                                                                      \
  Z("v_mul_f64 %[t], %[d2], %[d2]\n\t"                                                                                    \
    "v_add_f64 %[r2], %[r2], %[t]\n\t")                                                                                   \
  "v_cmpx_nge_f64_e64 %[take], " RV ", %[r2]\n\t"
// the accepted term under EXEC = take.  L: unique label suffix of this instance; RVHI: high word of the threshold — a body
// record (-1) or a node so small that an accepted d2 can be below 2^-16 sends the wave to look for near pairs (K9_NEAR_F64)
#define K9_EVAL_F64(L, Z, RM, RVHI, CNT_B)                                                                                \
  "v_rsq_f64_e32 %[y], %[r2]\n\t"                                                                                         \
  CNT_B                                                                                                                   \
  "s_cmp_lt_i32 " RVHI ", 0x3ef00000\n\t"                                                                                 \
  "s_cbranch_scc1 .LK9maybe" L "%=\n"                                                                                     \
  ".LK9far" L "%=:\n\t"                                                                                                   \
  K9_FAR(RM)                                                                                                              \
  ".LK9acc" L "%=:\n\t"                                                                                                   \
  "v_fma_f64 %[acc0], -%[y], %[d0], %[acc0]\n\t"                                                                          \
  "v_fma_f64 %[acc1], -%[y], %[d1], %[acc1]\n\t"                                                                          \
  Z("v_fma_f64 %[acc2], -%[y], %[d2], %[acc2]\n\t")
// out of line: some accepted entry is closer than 2^-8: those lanes take the guarded form (pair_math::weight<3>)
#define K9_NEAR_F64(L, Z, RM)                                                                                             \
  ".LK9maybe" L "%=:\n\t"                                                                                                 \
  "v_cmp_gt_u64_e64 %[near], %[nearhi], %[r2]\n\t"                                                                        \
  "s_cmp_eq_u64 %[near], 0\n\t"                                                                                           \
  "s_cbranch_scc1 .LK9far" L "%=\n\t"                                                                                     \
  K9_FAR(RM)                                                                                                              \
  "s_mov_b64 exec, %[near]\n\t"                                                                                           \
  "v_mov_b64_e32 %[r2], %[tiny]\n\t"                                                                                      \
  "v_fmac_f64_e32 %[r2], %[d0], %[d0]\n\t"                                                                                \
  "v_fmac_f64_e32 %[r2], %[d1], %[d1]\n\t"                                                                                \
  Z("v_fmac_f64_e32 %[r2], %[d2], %[d2]\n\t")                                                                             \
  "v_rsq_f64_e32 %[y2], %[r2]\n\t"                                                                                        \
  "s_nop 0\n\t"                                                                                                           \
  "v_mul_f64 %[e], %[r2], %[y2]\n\t"                                                                                      \
  "v_fma_f64 %[y2], -%[e], %[y2], 1.0\n\t"                                                                                \
  "v_fma_f64 %[p], %[y2], %[k0375], 0.5\n\t"                                                                              \
  "v_mul_f64 %[y2], %[e], %[y2]\n\t"                                                                                      \
  "v_fmac_f64_e32 %[e], %[y2], %[p]\n\t"                                                                                  \
  "v_fma_f64 %[r2], %[r2], %[e], %[eps]\n\t"                                                                              \
  "v_rcp_f64_e32 %[y2], %[r2]\n\t"                                                                                        \
  "s_nop 0\n\t"                                                                                                           \
  "v_fma_f64 %[r2], -%[r2], %[y2], 1.0\n\t"                                                                               \
  "v_mul_f64 %[y2], %[y2], " RM "\n\t"                                                                                    \
  "v_fmac_f64_e32 %[r2], %[r2], %[r2]\n\t"                                                                                \
  "v_fma_f64 %[y], %[y2], %[r2], %[y2]\n\t"                                                                               \
  "s_mov_b64 exec, %[take]\n\t"                                                                                           \
  "s_branch .LK9acc" L "%=\n"

// ---- f32 pieces: 32-byte records, the opening test on the unfused d2 as in dist2_ref, the accepted term as
// pair_math<float>::weight on the fused r2 (no near path: 1-ulp seeds)
#define K9_TEST_F32(Z, RX, RY, RZ, RV)                                                                                    \
  "v_subrev_f32_e32 %[d0], " RX ", %[xs0]\n\t"                                                                            \
  "v_subrev_f32_e32 %[d1], " RY ", %[xs1]\n\t"                                                                            \
  Z("v_subrev_f32_e32 %[d2], " RZ ", %[xs2]\n\t")                                                                         \
  "v_mul_f32_e32 %[r2], %[d0], %[d0]\n\t"                                                                                 \
  "v_mul_f32_e32 %[t], %[d1], %[d1]\n\t"                                                                                  \
  "v_add_f32_e32 %[r2], %[r2], %[t]\n\t"                                                                                  \
  Z("v_mul_f32_e32 %[t], %[d2], %[d2]\n\t"                                                                                \
    "v_add_f32_e32 %[r2], %[r2], %[t]\n\t")                                                                               \
  "v_cmpx_nge_f32_e64 %[take], " RV ", %[r2]\n\t"
#define K9_EVAL_F32(L, Z, RM, RVHI, CNT_B)                                                                                \
  "v_fma_f32 %[r2], %[d0], %[d0], %[tiny]\n\t"                                                                            \
  "v_fmac_f32_e32 %[r2], %[d1], %[d1]\n\t"                                                                                \
  Z("v_fmac_f32_e32 %[r2], %[d2], %[d2]\n\t")                                                                             \
  "v_rsq_f32_e32 %[y], %[r2]\n\t"                                                                                         \
  CNT_B                                                                                                                   \
  "s_nop 0\n\t"                                                                                                           \
  "v_mul_f32_e32 %[y], %[r2], %[y]\n\t"                                                                                   \
  "v_fma_f32 %[y], %[r2], %[y], %[eps]\n\t"                                                                               \
  "v_rcp_f32_e32 %[y], %[y]\n\t"                                                                                          \
  "s_nop 0\n\t"                                                                                                           \
  "v_mul_f32_e32 %[y], " RM ", %[y]\n\t"                                                                                  \
  "v_fma_f32 %[acc0], -%[y], %[d0], %[acc0]\n\t"                                                                          \
  "v_fma_f32 %[acc1], -%[y], %[d1], %[acc1]\n\t"                                                                          \
  Z("v_fma_f32 %[acc2], -%[y], %[d2], %[acc2]\n\t")
#define K9_NEAR_F32(L, Z, RM) ""

// ---- the skeleton.  One copy of the step: X = this copy ("A" / "B"), Y = the other; BIT / RBS = log2 / value of the record
// size; LOADY requests the record at off<Y> into Y's block; TEST / EVALD / EVALS are the pieces above with X's registers
// (two EVAL instances per copy: labels must differ); TAIL closes the main line of the copy.
#define K9_STEP(X, Y, BIT, RBS, LOADY, CNT_A, TEST, EVALD, TAIL)                                                          \
  ".LK9top" X "%=:\n\t"                                                                                                   \
  "v_cmpx_eq_u32_e64 %[match], %[cur" X "], %[key]\n\t" /* EXEC = the lanes standing on the entry */                      \
  "s_bitcmp1_b32 %[off" X "], " BIT "\n\t"                                                                                \
  "s_addc_u32 %[cur" Y "], %[cur" X "], %[sp" X "]\n\t" /* the skip key: cur + (span - 1) + (left child) */               \
  CNT_A(X)                                                                                                                \
  "v_add_u32_e32 %[key], 1, %[key]\n\t"        /* they move to its left child ... */                                      \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                                              \
  TEST                                                                                                                    \
  "v_mov_b32_e32 %[key], %[cur" Y "]\n\t"      /* ... except those that accept it: the skip key */                        \
  "s_andn2_b64 %[op], %[match], %[take]\n\t"   /* SCC: somebody opens the entry */                                        \
  "s_cbranch_scc0 .LK9skip" X "%=\n\t"                                                                                    \
  "s_add_i32 %[cur" Y "], %[cur" X "], 1\n\t"                                                                             \
  "s_lshl1_add_u32 %[off" Y "], %[off" X "], " RBS "\n\t"                                                                 \
  "s_lshr_b32 %[sp" Y "], %[sp" X "], 1\n\t"                                                                              \
  LOADY                                                                                                                   \
  "s_cbranch_execz .LK9dn" X "%=\n\t"        /* nobody accepted */                                                        \
  EVALD                                                                                                                   \
  ".LK9dn" X "%=:\n\t"                                                                                                    \
  "s_mov_b64 exec, %[sv]\n\t"                                                                                             \
  TAIL
// out of line: nobody opens the entry (every lane on it accepted; EXEC = those lanes, cur<Y> = their skip key)
#define K9_SKIP(X, Y, BIT, RBS, RBSH, LOADY, EVALS)                                                                       \
  ".LK9skip" X "%=:\n\t"                                                                                                  \
  "s_lshr_b32 %[t1], %[off" X "], 1\n\t"        /* right child -> parent + 1 */                                           \
  "s_add_i32 %[t2], %[off" X "], " RBS "\n\t"   /* left child -> sibling */                                               \
  "s_lshl1_add_u32 %[t3], %[sp" X "], 1\n\t"                                                                              \
  "s_bitcmp1_b32 %[off" X "], " BIT "\n\t"                                                                                \
  "s_cselect_b32 %[off" Y "], %[t2], %[t1]\n\t"                                                                           \
  "s_cselect_b32 %[sp" Y "], %[sp" X "], %[t3]\n\t"                                                                       \
  LOADY                                                                                                                   \
  EVALS                                                                                                                   \
  "s_mov_b64 exec, %[sv]\n\t"                                                                                             \
  "v_cmp_gt_u32_e32 vcc, %[cur" Y "], %[key]\n\t"          /* lanes waiting below the entry just left */                  \
  "s_cbranch_vccnz .LK9jump" X "%=\n\t"                                                                                   \
  "s_cmp_lt_u32 %[cur" Y "], %[endk]\n\t"                                                                                 \
  "s_cbranch_scc1 .LK9top" Y "%=\n\t"                                                                                     \
  "s_branch .LK9end%=\n"                                                                                                  \
  ".LK9jump" X "%=:\n\t" /* continue at the smallest key any lane holds; the record requested above is not the one */     \
  "s_ff1_i32_b64 %[t1], vcc\n\t"                                                                                          \
  "v_readlane_b32 %[cur" Y "], %[key], %[t1]\n\t"                                                                         \
  "s_nop 1\n\t"                                                                                                           \
  "v_cmp_gt_u32_e32 vcc, %[cur" Y "], %[key]\n\t"                                                                         \
  "s_cbranch_vccnz .LK9jump" X "%=\n\t"                                                                                   \
  "s_cmp_lt_u32 %[cur" Y "], %[endk]\n\t" /* only finished lanes were behind: their keys are positions past the tree */   \
  "s_cbranch_scc0 .LK9end%=\n\t"                                                                                          \
  "s_and_b32 %[t1], %[cur" Y "], 31\n\t"                                                                                  \
  "s_sub_i32 %[t1], %[nlev], %[t1]\n\t"                                                                                   \
  "s_lshl_b32 %[t2], -1, %[cur" Y "]\n\t"                                                                                 \
  "s_not_b32 %[t2], %[t2]\n\t"                                                                                            \
  "s_lshr_b32 %[t3], %[cur" Y "], 5\n\t"                                                                                  \
  "s_lshr_b32 %[t3], %[t3], %[t1]\n\t"                                                                                    \
  "s_add_i32 %[t3], %[t3], %[t2]\n\t"                                                                                     \
  "s_lshl_b32 %[off" Y "], %[t3], " RBSH "\n\t"                                                                           \
  "s_lshl_b32 %[sp" Y "], 32, %[t1]\n\t"                                                                                  \
  "s_add_i32 %[sp" Y "], %[sp" Y "], -1\n\t"                                                                              \
  "s_waitcnt lgkmcnt(0)\n\t" /* the first request must have landed before its registers are requested again */            \
  LOADY                                                                                                                   \
  "s_branch .LK9top" Y "%=\n"

#define K9_LOAD16(BLK, O) "s_load_dwordx16 " BLK ", %[node], %[off" O "]\n\t"
#define K9_LOAD8(BLK, O) "s_load_dwordx8 " BLK ", %[node], %[off" O "]\n\t"

// f64: A = s[40:55], B = s[56:71] (the highest SGPR the kernel names decides its waves per SIMD: see below).  D = 3: x s[0:1] y s[2:3] z s[4:5] m s[6:7] w s[8:9] w2 s[10:11] v s[12:13] of the block;
// D = 2: x s[0:1] y s[2:3] m s[4:5] w s[6:7] w2 s[8:9] v s[10:11].
#define K9_PROGRAM_F64(Z, AX, AY, AZ, AM, AV, AVHI, BX, BY, BZ, BM, BV, BVHI, CNT_A, CNT_B)                               \
  "s_mov_b64 %[sv], exec\n\t"                                                                                             \
  K9_LOAD16("s[40:55]", "A")                                                                                              \
  K9_STEP("A", "B", "6", "64", K9_LOAD16("s[56:71]", "B"), CNT_A, K9_TEST_F64(Z, AX, AY, AZ, AV),                         \
          K9_EVAL_F64("Ad", Z, AM, AVHI, CNT_B), "")                                                                    \
  K9_STEP("B", "A", "6", "64", K9_LOAD16("s[40:55]", "A"), CNT_A, K9_TEST_F64(Z, BX, BY, BZ, BV),                         \
          K9_EVAL_F64("Bd", Z, BM, BVHI, CNT_B), "s_branch .LK9topA%=\n")                                               \
  K9_SKIP("A", "B", "6", "64", "6", K9_LOAD16("s[56:71]", "B"), K9_EVAL_F64("As", Z, AM, AVHI, CNT_B))                    \
  K9_SKIP("B", "A", "6", "64", "6", K9_LOAD16("s[40:55]", "A"), K9_EVAL_F64("Bs", Z, BM, BVHI, CNT_B))                    \
  K9_NEAR_F64("Ad", Z, AM) K9_NEAR_F64("As", Z, AM) K9_NEAR_F64("Bd", Z, BM) K9_NEAR_F64("Bs", Z, BM)                     \
  ".LK9end%=:\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(0)\n\t" /* a record requested past the end of the walk: it must land before the registers are reused */ \
  "s_mov_b64 exec, %[sv]"
// f32: A = s[56:63], B = s[64:71].  D = 3: x y z m w w2 v = s0..s6 of the block; D = 2: x y m w w2 v = s0..s5.
#define K9_PROGRAM_F32(Z, AX, AY, AZ, AM, AV, BX, BY, BZ, BM, BV, CNT_A, CNT_B)                                           \
  "s_mov_b64 %[sv], exec\n\t"                                                                                             \
  K9_LOAD8("s[56:63]", "A")                                                                                               \
  K9_STEP("A", "B", "5", "32", K9_LOAD8("s[64:71]", "B"), CNT_A, K9_TEST_F32(Z, AX, AY, AZ, AV),                          \
          K9_EVAL_F32("Ad", Z, AM, "", CNT_B), "")                                                                      \
  K9_STEP("B", "A", "5", "32", K9_LOAD8("s[56:63]", "A"), CNT_A, K9_TEST_F32(Z, BX, BY, BZ, BV),                          \
          K9_EVAL_F32("Bd", Z, BM, "", CNT_B), "s_branch .LK9topA%=\n")                                                 \
  K9_SKIP("A", "B", "5", "32", "5", K9_LOAD8("s[64:71]", "B"), K9_EVAL_F32("As", Z, AM, "", CNT_B))                       \
  K9_SKIP("B", "A", "5", "32", "5", K9_LOAD8("s[56:63]", "A"), K9_EVAL_F32("Bs", Z, BM, "", CNT_B))                       \
  ".LK9end%=:\n\t"                                                                                                        \
  "s_waitcnt lgkmcnt(0)\n\t" /* a record requested past the end of the walk: it must land before the registers are reused */ \
  "s_mov_b64 exec, %[sv]"

#ifdef NBODY_EXPERIMENTS
// tools/k9_timeline.py: per work item of a sweep launch (start, end) in 100 MHz ticks and the item, three words per launch block
__device__ unsigned long long* g_k9_timeline = nullptr;
static unsigned long long* g_k9_timeline_host = nullptr;  // the same buffer, for nbody_exp_k9_timeline
static size_t g_k9_timeline_words = 0;
#endif

template <typename T, int D, bool COUNT>
__global__ __launch_bounds__(64) void bvh_force_sweep_isa_kernel(const tree_rec<T>* __restrict__ node, T* __restrict__ a,
                                                                 const T* __restrict__ x, T c, uint32_t sz, uint32_t first,
                                                                 uint32_t count, uint32_t nlevels,
                                                                 uint32_t* __restrict__ counters, const uint32_t* __restrict__ items,
                                                                 const uint32_t* __restrict__ nitems, uint32_t stride, uint32_t parts) {
  static_assert(sizeof(tree_rec<T>) == 8 * sizeof(T), "the step program addresses records of 8 scalars");
  // work item of this block: with an item list, the XCD it runs on (block index mod 8) owns one list (bvh_items_kernel)
  uint32_t group = xcd_contiguous_block(blockIdx.x, gridDim.x), lane_lo = 0, lane_hi = 63;
  if (items) {
    const uint32_t xcd = blockIdx.x % 8u, slot = blockIdx.x / 8u;
    if (slot >= nitems[xcd]) return;
    const uint32_t it = items[xcd * stride + slot];
    group   = it & 0xfffffu;
    lane_lo = (it >> 20) & 63u;
    lane_hi = it >> 26;
  } else if (parts > 1u) {  // every group as `parts` equal lane ranges (small systems: see force_run)
    lane_lo = (group % parts) * (64u / parts);
    lane_hi = lane_lo + 64u / parts - 1u;
    group /= parts;
  }
#ifdef NBODY_EXPERIMENTS
  const unsigned long long tl_start = g_k9_timeline ? wall_clock64() : 0ull;
#endif
  const uint32_t local = group * 64u + threadIdx.x;
  const bool valid     = local < count && threadIdx.x >= lane_lo && threadIdx.x <= lane_hi;
  const uint32_t bi    = first + (valid ? local : 0u);
  // A lane's key is the packed position (covered << 5 | level) of the entry it visits next; it keeps counting past the end
  // (covered >= sz: finished), so only lanes outside the item need a sentinel.
  uint32_t key         = valid ? 0u : 0xffffffffu;
  T xs[3] = {T(0), T(0), T(0)}, acc[3] = {T(0), T(0), T(0)};
#pragma unroll
  for (int k = 0; k < D; ++k) xs[k] = x[uint64_t(bi) * D + k];
  uint32_t cn = 0, cl = 0, cm = 0, cb = 0;
  // the sweep's position: packed key, byte offset of the record (level-order index * record size), (32 << levels below) - 1
  uint32_t curA = 0, offA = 0, spA = (32u << nlevels) - 1u, curB, offB, spB;
  // cur >= (sz << 5) - 1: every remaining key is >= cur and covered >= sz: all lanes are finished.  One less than sz << 5 because
  // an accepted ROOT is left by the ascend rule with covered + 2^nlevels and level - 1 = 31 after the borrow, which is
  // (sz << 5) - 1 when sz is a power of two; no live key has level 31.
  const uint32_t endk = (sz << 5) - 1u;
  uint32_t t1, t2, t3, cinc = 0;
  uint64_t match, take, op, sv;
#define K9_CLOBBER8                                                                                                        \
  "vcc", "scc", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71"
  if constexpr (sizeof(T) == 8) {
    const pair_consts<double> pc;
    double k0375 = 0.375, tiny = pair_math<double>::tiny, eps = DBL_EPSILON;
    uint64_t nearhi = uint64_t(pair_math<double>::near_hi) << 32;
    int m52 = -52;
    asm volatile("" : "+s"(k0375), "+s"(nearhi), "+s"(m52), "+v"(tiny), "+v"(eps));
    double d0, d1, d2, r2, t, y, y2, e, p, q;
    uint64_t near;
#define K9_OPERANDS                                                                                                        \
  : [acc0] "+v"(acc[0]), [acc1] "+v"(acc[1]), [acc2] "+v"(acc[2]), [key] "+v"(key), [curA] "+s"(curA), [offA] "+s"(offA),    \
    [spA] "+s"(spA), [curB] "=&s"(curB), [offB] "=&s"(offB), [spB] "=&s"(spB), [cn] "+v"(cn), [cl] "+v"(cl), [cm] "+v"(cm),  \
    [cb] "+v"(cb), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [r2] "=&v"(r2), [t] "=&v"(t), [y] "=&v"(y),              \
    [y2] "=&v"(y2), [e] "=&v"(e), [p] "=&v"(p), [q] "=&v"(q), [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3),               \
    [match] "=&s"(match), [take] "=&s"(take), [op] "=&s"(op), [near] "=&s"(near), [sv] "=&s"(sv), [cinc] K9_CINC(cinc)                       \
  : [node] "s"(node), [nlev] "s"(nlevels), [endk] "s"(endk), [k1875] "s"(pc.k1875), [nearhi] "s"(nearhi),                  \
    [k0375] "s"(k0375), [m52] "s"(m52), [xs0] "v"(xs[0]), [xs1] "v"(xs[1]), [xs2] "v"(xs[2]), [k15] "v"(pc.k15),           \
    [tiny] "v"(tiny), [eps] "v"(eps), [bi] "v"(bi)                                                                         \
  : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56",  \
    "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71"
    // The record blocks sit at s[40:71]: the highest SGPR a wave names decides how many waves a SIMD holds, and the steps are NOT
    // where the arithmetic puts them (800 SGPRs, granules of 16): measured with long one-wave blocks that name one high register
    // (tools/microbench/cu_map.hip, blocks alive at once per SIMD): up to s73 EIGHT, s74 ... s88 seven, s95 and above six (round 6
    // measured s72 / s73 / s74: 8 / 8 / 7, profiles/r06/cu_map_microbench.txt) — the granule is 16 SGPRs of a count that includes VCC and
    // the two reserved pairs (s73 = count 80), and 16 more are held back per wave: floor(800 / (ceil16(count) + 16)).  Round 4's first form had the blocks at s[64:95] (six waves, believed seven), its second at
    // s[48:79] (SEVEN waves — 7168 items alive at once in tools/k9_timeline.py — believed eight); at s[40:71] the kernel's count is
    // 78 with VCC and the reserved pairs and 8192 items are alive at once: config 4 6.38 -> 6.21 ms per traversal, float 4.49 -> 4.31.
#define K9_REGS3                                                                                                           \
  "s[40:41]", "s[42:43]", "s[44:45]", "s[46:47]", "s[52:53]", "s53", "s[56:57]", "s[58:59]", "s[60:61]", "s[62:63]", "s[68:69]", "s69"
#define K9_REGS2                                                                                                           \
  "s[40:41]", "s[42:43]", "", "s[44:45]", "s[50:51]", "s51", "s[56:57]", "s[58:59]", "", "s[60:61]", "s[66:67]", "s67"
    if constexpr (D == 3) {
      if constexpr (COUNT) asm volatile(K9_APPLY(K9_PROGRAM_F64, K9_KEEP, K9_REGS3, K9_COUNT_A, K9_COUNT_B) K9_OPERANDS);
      else asm volatile(K9_APPLY(K9_PROGRAM_F64, K9_KEEP, K9_REGS3, K9_NOCOUNT_A, "") K9_OPERANDS);
    } else {
      if constexpr (COUNT) asm volatile(K9_APPLY(K9_PROGRAM_F64, K9_DROP, K9_REGS2, K9_COUNT_A, K9_COUNT_B) K9_OPERANDS);
      else asm volatile(K9_APPLY(K9_PROGRAM_F64, K9_DROP, K9_REGS2, K9_NOCOUNT_A, "") K9_OPERANDS);
    }
#undef K9_REGS3
#undef K9_REGS2
#undef K9_OPERANDS
  } else {
    float tiny = pair_math<float>::tiny, eps = FLT_EPSILON;
    asm volatile("" : "+s"(tiny), "+s"(eps));
    float d0, d1, d2, r2, t, y;
#define K9_OPERANDS                                                                                                        \
  : [acc0] "+v"(acc[0]), [acc1] "+v"(acc[1]), [acc2] "+v"(acc[2]), [key] "+v"(key), [curA] "+s"(curA), [offA] "+s"(offA),    \
    [spA] "+s"(spA), [curB] "=&s"(curB), [offB] "=&s"(offB), [spB] "=&s"(spB), [cn] "+v"(cn), [cl] "+v"(cl), [cm] "+v"(cm),  \
    [cb] "+v"(cb), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [r2] "=&v"(r2), [t] "=&v"(t), [y] "=&v"(y),              \
    [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3), [match] "=&s"(match), [take] "=&s"(take), [op] "=&s"(op),              \
    [sv] "=&s"(sv), [cinc] K9_CINC(cinc)                                                                                                     \
  : [node] "s"(node), [nlev] "s"(nlevels), [endk] "s"(endk), [xs0] "v"(xs[0]), [xs1] "v"(xs[1]), [xs2] "v"(xs[2]),         \
    [tiny] "s"(tiny), [eps] "s"(eps), [bi] "v"(bi)                                                                         \
  : K9_CLOBBER8
#define K9_REGS3 "s56", "s57", "s58", "s59", "s62", "s64", "s65", "s66", "s67", "s70"
#define K9_REGS2 "s56", "s57", "", "s58", "s61", "s64", "s65", "", "s66", "s69"
    if constexpr (D == 3) {
      if constexpr (COUNT) asm volatile(K9_APPLY(K9_PROGRAM_F32, K9_KEEP, K9_REGS3, K9_COUNT_A, K9_COUNT_B) K9_OPERANDS);
      else asm volatile(K9_APPLY(K9_PROGRAM_F32, K9_KEEP, K9_REGS3, K9_NOCOUNT_A, "") K9_OPERANDS);
    } else {
      if constexpr (COUNT) asm volatile(K9_APPLY(K9_PROGRAM_F32, K9_DROP, K9_REGS2, K9_COUNT_A, K9_COUNT_B) K9_OPERANDS);
      else asm volatile(K9_APPLY(K9_PROGRAM_F32, K9_DROP, K9_REGS2, K9_NOCOUNT_A, "") K9_OPERANDS);
    }
#undef K9_REGS3
#undef K9_REGS2
#undef K9_OPERANDS
  }
#undef K9_CLOBBER8
  if (valid) {
#pragma unroll
    for (int k = 0; k < D; ++k) a[uint64_t(local) * D + k] = c * acc[k];
    if (COUNT) {
      counters[uint64_t(bi) * 4 + 0] = cn;
      counters[uint64_t(bi) * 4 + 1] = cl;
      counters[uint64_t(bi) * 4 + 2] = cm;
      counters[uint64_t(bi) * 4 + 3] = cb;
    }
  }
#ifdef NBODY_EXPERIMENTS
  if (g_k9_timeline && threadIdx.x == 0) {
    g_k9_timeline[3 * size_t(blockIdx.x) + 0] = tl_start;
    g_k9_timeline[3 * size_t(blockIdx.x) + 1] = wall_clock64();
    g_k9_timeline[3 * size_t(blockIdx.x) + 2] = (1ull << 63) | (unsigned long long)(group) | ((unsigned long long)lane_lo << 32) | ((unsigned long long)lane_hi << 40) |
                                                 ((unsigned long long)(cinc & 0x1ffffu) << 46);  // the sweep's steps (not with counters on)
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int check_tree(const nbody_bvh* t, const nbody_state* s, bool need_full, void* stream) {
  NB_ARG(t != nullptr, "nbody_bvh is NULL");
  if (int r = check_state(s)) return r;
  if (int r = check_same_device(t->device, as_stream(stream), "nbody_bvh")) return r;
  NB_ARG(t->dtype == s->dtype && t->dim == s->dim && t->n == s->sz, "bvh was created for (dtype=%d, dim=%d, n=%u), state is (%d, %d, %u)",
         t->dtype, t->dim, t->n, s->dtype, s->dim, s->sz);
  if (need_full) NB_ARG(s->first == 0 && s->count == s->sz, "this bvh phase needs the whole system (first=0, count=sz)");
  return NBODY_OK;
}

template <typename T, int D>
static int bbox_run(nbody_bvh* t, const nbody_state* s, hipStream_t st) {
  hipLaunchKernelGGL((bbox_partial_kernel<T, D>), dim3(t->bbox_blocks), dim3(kB), 0, st, static_cast<const T*>(s->x), s->sz,
                     static_cast<T*>(t->partials));
  NB_HIP(hipGetLastError());
  hipLaunchKernelGGL((bbox_final_kernel<T, D>), dim3(1), dim3(kB), 0, st, static_cast<const T*>(t->partials), t->bbox_blocks,
                     static_cast<T*>(t->bbox));
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

template <typename T, int D>
static int sort_run(nbody_bvh* t, const nbody_state* s, hipStream_t st) {
  const uint32_t n = s->sz;
  hipLaunchKernelGGL((hilbert_keys_kernel<T, D>), dim3((n + kB - 1) / kB), dim3(kB), 0, st, static_cast<const T*>(s->x), n,
                     static_cast<const T*>(t->bbox), t->keys[0], t->keys[1]);
  NB_HIP(hipGetLastError());
  // the sort ping-pongs between keys[1] and a key buffer carved from tmp (keys[0] is kept for nbody_bvh_read)
  uint64_t* kbuf[2]   = {t->keys[1], reinterpret_cast<uint64_t*>(t->tmp)};
  const int key_bits  = (D == 2) ? 64 : 63;
  int cur             = 0;
  if (int r = radix_sort_pairs(kbuf, t->idx, n, key_bits, t->hist, st, &cur)) return r;
  t->final_buf = cur;
  // permute the state in place: gather into tmp, copy back
  T* tmp = static_cast<T*>(t->tmp);
  hipLaunchKernelGGL((gather_kernel<T, D>), dim3((n + kB - 1) / kB), dim3(kB), 0, st, t->idx[cur], n, static_cast<const T*>(s->m),
                     static_cast<const T*>(s->x), static_cast<const T*>(s->v), static_cast<const T*>(s->a),
                     static_cast<const T*>(s->ao), tmp);
  NB_HIP(hipGetLastError());
  if (n <= (1u << 18)) {  // launch-bound sizes: one kernel; beyond, the runtime's copy kernels are the faster way to move 5 arrays
    hipLaunchKernelGGL((ungather_kernel<T, D>), dim3(uint32_t((uint64_t(n) * D + kB - 1) / kB)), dim3(kB), 0, st, tmp, n,
                       static_cast<T*>(s->m), static_cast<T*>(s->x), static_cast<T*>(s->v), static_cast<T*>(s->a),
                       static_cast<T*>(s->ao));
    NB_HIP(hipGetLastError());
  } else {
    const size_t vb = sizeof(T) * size_t(n) * D;
    NB_HIP(hipMemcpyAsync(s->x, tmp, vb, hipMemcpyDeviceToDevice, st));
    NB_HIP(hipMemcpyAsync(s->v, tmp + size_t(n) * D, vb, hipMemcpyDeviceToDevice, st));
    NB_HIP(hipMemcpyAsync(s->a, tmp + size_t(n) * D * 2, vb, hipMemcpyDeviceToDevice, st));
    NB_HIP(hipMemcpyAsync(s->ao, tmp + size_t(n) * D * 3, vb, hipMemcpyDeviceToDevice, st));
    NB_HIP(hipMemcpyAsync(s->m, tmp + size_t(n) * D * 4, sizeof(T) * size_t(n), hipMemcpyDeviceToDevice, st));
  }
  return NBODY_OK;
}

template <typename T, int D>
static int build_run(nbody_bvh* t, const nbody_state* s, hipStream_t st) {
  auto* node         = static_cast<tree_rec<T>*>(t->node);
  T* box             = static_cast<T*>(t->box);
  const int last     = int(t->nlevels) - 1;
  const uint32_t cnt = 1u << last;
  const T th         = static_cast<T>(t->theta);
  const T th2        = th * th;  // src/bvh.h:252, in T: the opening thresholds of the records are written for it
  hipLaunchKernelGGL((build_leaf_level_kernel<T, D>), dim3((cnt + kB - 1) / kB), dim3(kB), 0, st, static_cast<const T*>(s->m),
                     static_cast<const T*>(s->x), s->sz, cnt - 1u, cnt, t->nnodes, node, box, th2);
  NB_HIP(hipGetLastError());
  // Nine levels per launch: a block takes 256 nodes of the deepest one and follows their subtree chunk up to its single node nine
  // levels higher, with block barriers between the levels (one launch per level was 8 dependent launches at N = 10^5, 11 at 10^6);
  // the top levels, from 256 nodes on, are one block's.
  for (int l = last - 1; l >= 0;) {
    if ((1u << l) > uint32_t(kB)) {
      const int lo = l - 8 > 0 ? l - 8 : 0;
      hipLaunchKernelGGL((build_upper_levels_kernel<T, D>), dim3((1u << l) / kB), dim3(kB), 0, st, l, lo, node, box, th2);
      NB_HIP(hipGetLastError());
      l = lo - 1;
    } else {
      hipLaunchKernelGGL((build_upper_levels_kernel<T, D>), dim3(1), dim3(kB), 0, st, l, 0, node, box, th2);
      NB_HIP(hipGetLastError());
      l = -1;
    }
  }
  if (const unsigned long long cap = capture_id(st)) {  // replayed at times the host does not see: th2_built says nothing any more
    t->ever_recorded = true;
    t->rec_id        = cap;
    t->rec_th2       = double(th2);
  }
  t->th2_built = double(th2);
  return NBODY_OK;
}

template <typename T, int D>
static int force_run(nbody_bvh* t, const nbody_state* s, double theta, hipStream_t st) {
  if (s->count == 0) return NBODY_OK;
  const T th  = static_cast<T>(theta);
  const T th2 = th * th;  // src/bvh.h:252, in T
  auto* node = static_cast<const tree_rec<T>*>(t->node);
  t->theta   = theta;  // the next build writes the records' opening thresholds for this angle
  // The sweep trusts the thresholds in the records.  The host knows what they hold only along its own eager calls: a recorded
  // step rewrites them whenever it is replayed.  So a recorded traversal skips the rewrite only behind a build (or traversal) of
  // the SAME capture for the same angle, and once anything of this tree has been recorded an eager traversal always rewrites.
  const unsigned long long cap = capture_id(st);
  const bool fresh = cap ? (t->rec_id == cap && t->rec_th2 == double(th2)) : (!t->ever_recorded && t->th2_built == double(th2));
  if (!fresh) {
    hipLaunchKernelGGL((rethreshold_kernel<T, D>), dim3((t->nnodes + kB - 1) / kB), dim3(kB), 0, st,
                       static_cast<tree_rec<T>*>(t->node), t->nnodes, th2);
    NB_HIP(hipGetLastError());
  }
  if (cap) {
    t->ever_recorded = true;
    t->rec_id        = cap;
    t->rec_th2       = double(th2);
  }
  t->th2_built = double(th2);
  // auto: the wave-cooperative sweep needs enough waves in flight to hide its serial chain.  Measured in the CLI's step loop on
  // 256 CUs (ms per whole bvh step over the first 200 steps of the galaxy, sweep / per-lane): f64 (hand-scheduled sweep) 0.90 / 0.85
  // at 4*10^4, 1.0 / 1.0 at 6*10^4, 1.1 / 1.15 at 8*10^4, 1.2 / 1.25 at 10^5, 1.3 / 1.5 at 1.3*10^5, 2.05 / 3.45 at 2.5*10^5, 3.45 / 7.1
  // at 5*10^5; f32 0.90 / 0.65 at 6*10^4, 1.05 / 0.80 at 10^5, 1.25 / 1.15 at 1.6*10^5, 1.5 / 1.9 at 2.5*10^5, 4.45 / 10.6 at 10^6.  A system that has
  // evolved favours the sweep further: escapers inflate the box, walks get 3x longer and the per-lane form's divergent
  // gathers pay for every entry (10^5 bodies after 400-1000 steps: sweep 1.7-2.1 ms, per-lane 3.0-3.7 ms per traversal).
  // f64 below 5*10^4: the first steps of the galaxy favour the per-lane form by 7 % (4*10^4) to 20 % (10^4), its evolved states the
  // sweep by 25-40 % (whole 1000-step runs, sweep / per-lane: 0.86 / 0.91 s at 10^4, 0.99 / 1.07 at 2*10^4, 1.16 / 1.47 at 3*10^4)
  // f64 with the finer work items of small systems (below): the sweep wins from the smallest sizes measured (10^4: 0.27 against 0.33 ms)
  const uint32_t crossover = sizeof(T) == 8 ? 4096u : 180000u;
  int traversal = t->traversal;
  if (const char* e = experiment_env("NBODY_K9_MODE"); e && traversal == 0) traversal = atoi(e);  // -DNBODY_EXPERIMENTS builds only
#ifdef NBODY_EXPERIMENTS
  if (traversal == 6) {  // four 16-lane row sweeps per wave (experiment; no counters)
    if (t->counters_on || t->nlevels > 26) {
      set_error("traversal mode 6 (row sweeps) has no counters and needs nlevels <= 26");
      return NBODY_ERR_ARG;
    }
    const uint32_t rblocks = (s->count + 63u) / 64u;
    hipLaunchKernelGGL((bvh_force_row_kernel<T, D>), dim3(rblocks), dim3(64), 0, st, node, static_cast<T*>(s->a),
                       static_cast<const T*>(s->x), static_cast<T>(s->c), s->sz, s->first, s->count, th2, t->nlevels);
    NB_HIP(hipGetLastError());
    return NBODY_OK;
  }
#endif
  const bool wave = traversal >= 2 || (traversal == 0 && t->nlevels <= 26 && s->count >= crossover);
  if (wave && t->nlevels > 26) {
    set_error("wave-cooperative traversal needs nlevels <= 26 (n <= 2^26), tree has %u levels", t->nlevels);
    return NBODY_ERR_ARG;
  }
  // The sweep is the step program written out as ISA (bvh_force_sweep_isa_kernel: auto, 2, 5).  The -DNBODY_EXPERIMENTS build also
  // has the compiler-scheduled step (3; with 2 bodies per lane: 4 — 128 bodies per wave, config 4 12.2 ms against 9.55) and the
  // row sweeps (6).  All forms are bitwise identical.
  const int bpl  = traversal == 4 ? 2 : 1;
  [[maybe_unused]] const bool isa = wave && (traversal == 0 || traversal == 2 || traversal == 5);
  const uint32_t per_block = wave ? 64u * uint32_t(bpl) : 64u;
  const uint32_t blocks    = (s->count + per_block - 1) / per_block;
#define NB_ARGS                                                                                                   \
  dim3(blocks), dim3(64), 0, st, node, static_cast<T*>(s->a), static_cast<const T*>(s->x), static_cast<T>(s->c), s->sz, s->first, \
   s->count, th2, t->nlevels, t->counters
  const uint32_t *items = nullptr, *nitems = nullptr;
  uint32_t stride = 0, wave_blocks = blocks, parts = 1;
  const char* oe = experiment_env("NBODY_K9_ORDER");  // -DNBODY_EXPERIMENTS builds: 0 = one block per group in index order
  const bool plain = t->launch_order != 0 || (oe && oe[0] == '0');
  // Finer work items (profiles/r03/k9_parts.txt, ms per traversal of the initial galaxy, f64): every group swept as `parts` equal
  // lane ranges side by side has shorter unions (16 Hilbert-adjacent walks: 5.7k steps against 7.4k for 64 at config 4) but
  // more of them.  That pays only while the chip is nearly empty — N = 10^4: 0.40 (1 part) / 0.36 (4) / 0.27 (8), per-lane form
  // 0.33; N = 3*10^4: 0.64 / 0.56 / 0.56, per-lane 0.58 — and costs from 6*10^4 bodies on (10^5: 1.06 / 1.62 / 1.86; 10^6: 6.96 /
  // 16.6 / 28.2).  The key-jump cutter below behaves the same way at N = 10^6: 1/16 of the groups cut 6.97 ms, 1/8 7.11, 1/4 7.39,
  // 1/2 7.84, all 10.3 — the filling and draining of a launch cannot be bought back with more, shorter items.
  if (wave && bpl == 1 && !plain) {
    parts = blocks <= 320u ? 8u : (blocks <= 640u ? 4u : 1u);
    if (const char* pe = experiment_env("NBODY_K9_PARTS")) parts = uint32_t(atoi(pe));  // -DNBODY_EXPERIMENTS builds only
    if (parts != 1u && parts != 2u && parts != 4u && parts != 8u && parts != 16u) parts = 1u;
  }
  if (parts > 1u) {
    wave_blocks = blocks * parts;
  } else if (wave && bpl == 1 && t->sorted && t->final_buf == 0 && !plain) {
    // Work items (bvh_items_kernel): groups that straddle a jump of the key order are cut in two and started first.  Measured
    // in the CLI's step loop (ms per whole bvh step; index order / start order only / start order + cut): N = 10^6 8.05 / 7.4 /
    // 7.4, 5*10^5 5.3 / 4.6 / 4.3.  The sorted keys are in keys[1] when the sort ends in the buffer it started from
    // (final_buf == 0: the splitter sort, the radix sort's 8 passes; not the one-block sort, whose small trees take finer work items instead).
    uint32_t den = 16;
    if (const char* de = experiment_env("NBODY_K9_SPLIT")) den = uint32_t(atoi(de)) ? uint32_t(atoi(de)) : 16u;  // experiments only
    uint32_t s0, l0;
    xcd_range(0, blocks, &s0, &l0);  // XCD 0 has the longest range
    stride      = l0 + split_budget(l0, den);
    wave_blocks = 8u * stride;       // blocks are dealt round-robin over the XCDs: 8 lists of up to `stride` items
    hipLaunchKernelGGL(bvh_items_kernel, dim3(8), dim3(1024), 0, st, t->keys[1], s->first, s->count, t->order, t->order_n, blocks,
                       stride, den);
    NB_HIP(hipGetLastError());
    items  = t->order;
    nitems = t->order_n;
  }
  const char* le = experiment_env("NBODY_K9_LDS");  // -DNBODY_EXPERIMENTS builds: dynamic LDS bytes per block, to cap the waves per SIMD
  const uint32_t lds = le ? uint32_t(atoi(le)) : 0u;
#define NB_WARGS                                                                                                             \
  dim3(wave_blocks), dim3(64), lds, st, node, static_cast<T*>(s->a), static_cast<const T*>(s->x), static_cast<T>(s->c), s->sz,     \
   s->first, s->count, th2, t->nlevels, t->counters, items, nitems, stride, parts
#ifdef NBODY_EXPERIMENTS
  if (wave && bpl == 2) {
    if (t->counters_on) hipLaunchKernelGGL((bvh_force_wave_kernel<T, D, 2, true>), NB_WARGS);
    else hipLaunchKernelGGL((bvh_force_wave_kernel<T, D, 2, false>), NB_WARGS);
  } else if (wave && !isa) {
    if (t->counters_on) hipLaunchKernelGGL((bvh_force_wave_kernel<T, D, 1, true>), NB_WARGS);
    else hipLaunchKernelGGL((bvh_force_wave_kernel<T, D, 1, false>), NB_WARGS);
  } else
#endif
  if (wave) {
#define NB_IARGS                                                                                                             \
  dim3(wave_blocks), dim3(64), lds, st, node, static_cast<T*>(s->a), static_cast<const T*>(s->x), static_cast<T>(s->c), s->sz,     \
   s->first, s->count, t->nlevels, t->counters, items, nitems, stride, parts
#ifdef NBODY_EXPERIMENTS
    if (const char* e = getenv("NBODY_K9_TIMELINE"); e && e[0] == '1') {  // tools/k9_timeline.py
      static unsigned long long* buf = nullptr;
      static size_t cap = 0;
      if (cap < 3 * size_t(wave_blocks)) {
        if (buf) (void)hipFree(buf);
        cap = 3 * size_t(wave_blocks);
        NB_HIP(hipMalloc(&buf, cap * sizeof(unsigned long long)));
        NB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_k9_timeline), &buf, sizeof buf));
      }
      NB_HIP(hipMemsetAsync(buf, 0, cap * sizeof(unsigned long long), st));
      g_k9_timeline_host = buf;
      g_k9_timeline_words = 3 * size_t(wave_blocks);
    }
#endif
    if (t->counters_on) hipLaunchKernelGGL((bvh_force_sweep_isa_kernel<T, D, true>), NB_IARGS);
    else hipLaunchKernelGGL((bvh_force_sweep_isa_kernel<T, D, false>), NB_IARGS);
#undef NB_IARGS
  } else {
    bool in_lds = false;
    if constexpr (sizeof(T) == 4) {  // (nbody_bvh_set_traversal(t, 1) keeps the walk over global memory: the tests compare the two bit for bit)
      const uint32_t nrec = t->nnodes + (1u << t->nlevels);
      if (traversal == 0 && uint64_t(nrec) * sizeof(tree_rec<T>) <= kTreeLdsBytes) {
        const uint32_t lblocks = (s->count + kTreeLdsThreads - 1) / kTreeLdsThreads;
        const uint32_t bytes   = nrec * uint32_t(sizeof(tree_rec<T>));
#define NB_LARGS                                                                                                              \
  dim3(lblocks), dim3(kTreeLdsThreads), bytes, st, node, nrec, static_cast<T*>(s->a), static_cast<const T*>(s->x), static_cast<T>(s->c), \
   s->sz, s->first, s->count, th2, t->nlevels, t->counters
        if (t->counters_on) hipLaunchKernelGGL((bvh_force_lds_kernel<T, D, true>), NB_LARGS);
        else hipLaunchKernelGGL((bvh_force_lds_kernel<T, D, false>), NB_LARGS);
#undef NB_LARGS
        in_lds = true;
      }
    }
    if (!in_lds) {
      if (t->counters_on) hipLaunchKernelGGL((bvh_force_kernel<T, D, true>), NB_ARGS);
      else hipLaunchKernelGGL((bvh_force_kernel<T, D, false>), NB_ARGS);
    }
  }
#undef NB_ARGS
#undef NB_WARGS
  NB_HIP(hipGetLastError());
  return NBODY_OK;
}

}  // namespace nbody

// ---- C ABI ------------------------------------------------------------------------------------------
using namespace nbody;

extern "C" int nbody_bvh_create(nbody_bvh** out, int dtype, int dim, uint32_t n) {
  return nbody_bvh_create_on(out, dtype, dim, n, -1);
}

extern "C" int nbody_bvh_create_on(nbody_bvh** out, int dtype, int dim, uint32_t n, int device) {
  NB_ARG(out != nullptr, "out is NULL");
  *out = nullptr;
  NB_ARG(dtype == NBODY_F32 || dtype == NBODY_F64, "bad dtype %d", dtype);
  NB_ARG(dim == 2 || dim == 3, "bad dim %d", dim);
  NB_ARG(n >= 2 && n <= (1u << 30), "bvh needs 2 <= n <= 2^30 (got %u)", n);
  int ndev = 0;
  NB_HIP(hipGetDeviceCount(&ndev));
  if (device < 0) device = current_device();
  NB_ARG(device >= 0 && device < ndev, "device %d out of range (%d HIP devices visible)", device, ndev);
  device_guard guard(device);
  auto* t  = new nbody_bvh;
  t->device = device;
  t->dtype = dtype;
  t->dim   = dim;
  t->n     = n;
  t->tsz   = dtype == NBODY_F32 ? 4 : 8;
  uint32_t nleafs = 1, nl = 0;  // bit_ceil / countr_zero (src/bvh.h:151-157)
  while (nleafs < n) {
    nleafs <<= 1;
    ++nl;
  }
  t->nlevels     = nl;
  t->nnodes      = (1u << nl) - 1u;
  t->sort_blocks = (n + kSortTile - 1) / kSortTile;
  t->bbox_blocks = (n + kB * 8 - 1) / (kB * 8);
  if (t->bbox_blocks > 1024) t->bbox_blocks = 1024;
  const size_t D = size_t(dim);
  t->rec_bytes   = 8 * t->tsz;  // tree_rec<T>
  // tmp doubles as a radix key buffer (u64[n]) and as the gather scratch (T[n*(4D+1)])
  size_t tmp_bytes = t->tsz * size_t(n) * (4 * D + 1);
  if (tmp_bytes < sizeof(uint64_t) * size_t(n)) tmp_bytes = sizeof(uint64_t) * size_t(n);
  auto fail = [&](hipError_t e, const char* what) {
    int r = hip_fail(e, what, __FILE__, __LINE__);
    nbody_bvh_destroy(t);
    return r;
  };
#define NB_ALLOC(ptr, bytes)                                         \
  do {                                                               \
    hipError_t e_ = hipMalloc(reinterpret_cast<void**>(&(ptr)), (bytes)); \
    if (e_ != hipSuccess) return fail(e_, "hipMalloc(" #ptr ")");    \
  } while (0)
  NB_ALLOC(t->bbox, t->tsz * 3 * D);
  NB_ALLOC(t->partials, t->tsz * 2 * D * t->bbox_blocks);
  NB_ALLOC(t->keys[0], sizeof(uint64_t) * size_t(n));
  NB_ALLOC(t->keys[1], sizeof(uint64_t) * size_t(n));
  NB_ALLOC(t->idx[0], sizeof(uint32_t) * size_t(n));
  NB_ALLOC(t->idx[1], sizeof(uint32_t) * size_t(n));
  NB_ALLOC(t->hist, sizeof(uint32_t) * radix_sort_scratch_words(n));
  NB_ALLOC(t->tmp, tmp_bytes);
  // internal nodes + body slots = level-order indices 0 .. 2 nleafs - 2, one record (one s_load_dwordx16 / x8) each.  No padding
  // is needed for the sweep's speculative requests: a descent asks for 2 i + 1 only when some lane OPENS entry i, and a body
  // record (threshold -1, NaN accepts) is never opened; a skip asks for the sibling i + 1 of a left child (i odd, so i + 1 is the
  // even index on the same level) or for parent + 1 = i / 2 <= i of a right child (the root, i = 0, asks for itself); the jump path
  // converts a live key (covered < sz, level <= nlevels) and leaves first if the key is past the tree.  Every request stays in
  // [0, 2 nleafs - 2] (tests: the sweep at theta = 3 accepting the root, n = 2, 3, powers of two — test_gpu_bvh.py).
  NB_ALLOC(t->node, t->rec_bytes * (size_t(t->nnodes) + size_t(nleafs)));
  NB_ALLOC(t->box, t->tsz * 2 * D * size_t(t->nnodes));
  NB_ALLOC(t->order, sizeof(uint32_t) * (((size_t(n) + 63) / 64 / 8 + 1) * 2 + 8) * 8);  // 8 x (longest range + the largest budget)
  NB_ALLOC(t->order_n, sizeof(uint32_t) * 8);
#undef NB_ALLOC
  if (dtype == NBODY_F32 && (uint64_t(t->nnodes) + nleafs) * t->rec_bytes <= kTreeLdsBytes) {  // bvh_force_lds_kernel asks for more dynamic LDS than a kernel gets unasked
    hipError_t e = hipSuccess;
    auto allow   = [&](const void* f) {
      if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(kTreeLdsBytes));
    };
#define NB_ALLOW(T, DD)                                                                  \
  allow(reinterpret_cast<const void*>(&bvh_force_lds_kernel<T, DD, false>)); \
  allow(reinterpret_cast<const void*>(&bvh_force_lds_kernel<T, DD, true>))
    if (dim == 2) { NB_ALLOW(float, 2); }
    if (dim == 3) { NB_ALLOW(float, 3); }
#undef NB_ALLOW
    if (e != hipSuccess) return fail(e, "hipFuncSetAttribute(bvh_force_lds_kernel)");
  }
  *out = t;
  return NBODY_OK;
}

extern "C" void nbody_bvh_destroy(nbody_bvh* t) {
  if (!t) return;
  device_guard guard(t->device);
  (void)hipFree(t->bbox);
  (void)hipFree(t->partials);
  (void)hipFree(t->keys[0]);
  (void)hipFree(t->keys[1]);
  (void)hipFree(t->idx[0]);
  (void)hipFree(t->idx[1]);
  (void)hipFree(t->hist);
  (void)hipFree(t->tmp);
  (void)hipFree(t->node);
  (void)hipFree(t->box);
  (void)hipFree(t->counters);
  (void)hipFree(t->order);
  (void)hipFree(t->order_n);
  delete t;
}

extern "C" uint32_t nbody_bvh_nnodes(const nbody_bvh* t) { return t ? t->nnodes : 0; }

extern "C" int nbody_bvh_enable_counters(nbody_bvh* t, int on) {
  NB_ARG(t != nullptr, "nbody_bvh is NULL");
  device_guard guard(t->device);
  if (on && !t->counters) NB_HIP(hipMalloc(reinterpret_cast<void**>(&t->counters), sizeof(uint32_t) * 4 * size_t(t->n)));
  t->counters_on = on != 0;
  return NBODY_OK;
}

extern "C" int nbody_bvh_set_traversal(nbody_bvh* t, int mode) {
  NB_ARG(t != nullptr, "nbody_bvh is NULL");
#ifdef NBODY_EXPERIMENTS
  NB_ARG(mode >= 0 && mode <= 6, "traversal mode must be 0 (auto), 1 (per-lane), 2 / 5 (wave-cooperative sweep), 3 / 4 (compiler-scheduled sweep with 1 / 2 bodies per lane), 6 (four 16-lane row sweeps per wave), got %d", mode);
#else
  NB_ARG(mode == 0 || mode == 1 || mode == 2 || mode == 5, "traversal mode must be 0 (auto), 1 (per-lane walks) or 2 (wave-cooperative sweep; 5 is the same), got %d%s", mode,
         mode == 3 || mode == 4 || mode == 6 ? " — that form exists only in the -DNBODY_EXPERIMENTS build (make experiments)" : "");
#endif
  t->traversal = mode;
  return NBODY_OK;
}

extern "C" int nbody_bvh_set_launch_order(nbody_bvh* t, int mode) {
  NB_ARG(t != nullptr, "nbody_bvh is NULL");
  NB_ARG(mode == 0 || mode == 1, "launch order must be 0 (work items) or 1 (index order), got %d", mode);
  t->launch_order = mode;
  return NBODY_OK;
}

extern "C" int nbody_bvh_bounding_box(nbody_bvh* t, const nbody_state* s, void* stream) {
  if (int r = check_tree(t, s, false, stream)) return r;
  device_guard guard(t->device);
  int r = dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return bbox_run<typename TG::type, TG::dim>(t, s, as_stream(stream));
  });
  if (r == NBODY_OK) t->have_bbox = true;
  return r;
}

extern "C" int nbody_bvh_get_bounding_box(nbody_bvh* t, void* xmin_out, void* xmax_out, void* stream) {
  NB_ARG(t != nullptr && xmin_out && xmax_out, "NULL argument");
  if (int r = check_same_device(t->device, as_stream(stream), "nbody_bvh")) return r;
  device_guard guard(t->device);
  if (!t->have_bbox) {
    set_error("nbody_bvh_get_bounding_box before nbody_bvh_bounding_box");
    return NBODY_ERR_STATE;
  }
  char buf[2 * 3 * 8];
  NB_HIP(hipMemcpyAsync(buf, t->bbox, t->tsz * 2 * t->dim, hipMemcpyDeviceToHost, as_stream(stream)));
  NB_HIP(hipStreamSynchronize(as_stream(stream)));
  memcpy(xmin_out, buf, t->tsz * t->dim);
  memcpy(xmax_out, buf + t->tsz * t->dim, t->tsz * t->dim);
  return NBODY_OK;
}

extern "C" int nbody_bvh_hilbert_sort(nbody_bvh* t, const nbody_state* s, void* stream) {
  if (int r = check_tree(t, s, true, stream)) return r;
  device_guard guard(t->device);
  if (!t->have_bbox) {
    set_error("nbody_bvh_hilbert_sort before nbody_bvh_bounding_box");
    return NBODY_ERR_STATE;
  }
  int r = dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return sort_run<typename TG::type, TG::dim>(t, s, as_stream(stream));
  });
  if (r == NBODY_OK) t->sorted = true;
  return r;
}

extern "C" int nbody_bvh_build_tree(nbody_bvh* t, const nbody_state* s, void* stream) {
  if (int r = check_tree(t, s, false, stream)) return r;
  device_guard guard(t->device);
  int r = dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return build_run<typename TG::type, TG::dim>(t, s, as_stream(stream));
  });
  if (r == NBODY_OK) t->built = true;
  return r;
}

extern "C" int nbody_bvh_compute_force(nbody_bvh* t, const nbody_state* s, double theta, void* stream) {
  if (int r = check_tree(t, s, false, stream)) return r;
  device_guard guard(t->device);
  if (!t->built) {
    set_error("nbody_bvh_compute_force before nbody_bvh_build_tree");
    return NBODY_ERR_STATE;
  }
  return dispatch(s->dtype, s->dim, [&](auto tg) {
    using TG = decltype(tg);
    return force_run<typename TG::type, TG::dim>(t, s, theta, as_stream(stream));
  });
}

extern "C" int nbody_bvh_opening_thresholds(int dtype, const void* width2, double theta, size_t n, void* out) {
  NB_ARG(dtype == NBODY_F32 || dtype == NBODY_F64, "bad dtype %d", dtype);
  NB_ARG((width2 != nullptr && out != nullptr) || n == 0, "NULL argument");
  if (dtype == NBODY_F32) {
    const float th = static_cast<float>(theta), th2 = th * th;
    for (size_t i = 0; i < n; ++i) static_cast<float*>(out)[i] = open_threshold<float>(static_cast<const float*>(width2)[i], th2);
  } else {
    const double th2 = theta * theta;
    for (size_t i = 0; i < n; ++i) static_cast<double*>(out)[i] = open_threshold<double>(static_cast<const double*>(width2)[i], th2);
  }
  return NBODY_OK;
}

extern "C" int nbody_bvh_read(nbody_bvh* t, int what, void* host_out, size_t bytes, void* stream) {
  NB_ARG(t != nullptr && host_out != nullptr, "NULL argument");
  if (int r = check_same_device(t->device, as_stream(stream), "nbody_bvh")) return r;
  device_guard guard(t->device);
  const size_t D = size_t(t->dim);
  hipStream_t st = as_stream(stream);
  auto copy_out = [&](const void* dev, size_t need) -> int {
    NB_ARG(bytes == need, "nbody_bvh_read(what=%d): expected %zu bytes, got %zu", what, need, bytes);
    NB_HIP(hipMemcpyAsync(host_out, dev, need, hipMemcpyDeviceToHost, st));
    NB_HIP(hipStreamSynchronize(st));
    return NBODY_OK;
  };
  switch (what) {
    case 0: return copy_out(t->keys[0], sizeof(uint64_t) * size_t(t->n));
    case 1: return copy_out(t->idx[t->final_buf], sizeof(uint32_t) * size_t(t->n));
    case 6:
      if (t->th2_built < 0.0) {
        set_error("nbody_bvh_read(what=6) before nbody_bvh_build_tree");
        return NBODY_ERR_STATE;
      }
      [[fallthrough]];
    case 3:
    case 2: {
      // unpack node records -> T[nnodes][D+1] (what=2), T[nnodes] widths (what=3) or opening thresholds (what=6)
      const size_t need = what == 2 ? t->tsz * (D + 1) * t->nnodes : t->tsz * t->nnodes;
      NB_ARG(bytes == need, "nbody_bvh_read(what=%d): expected %zu bytes, got %zu", what, need, bytes);
      char* raw = static_cast<char*>(malloc(t->rec_bytes * size_t(t->nnodes)));
      NB_ARG(raw != nullptr, "out of host memory");
      hipError_t e = hipMemcpyAsync(raw, t->node, t->rec_bytes * size_t(t->nnodes), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) {
        free(raw);
        return hip_fail(e, "read node records", __FILE__, __LINE__);
      }
      char* o = static_cast<char*>(host_out);
      for (size_t i = 0; i < t->nnodes; ++i) {
        const char* r = raw + i * t->rec_bytes;
        if (what == 2) memcpy(o + i * t->tsz * (D + 1), r, t->tsz * (D + 1));
        else memcpy(o + i * t->tsz, r + t->tsz * (D + (what == 3 ? 1 : 3)), t->tsz);
      }
      free(raw);
      return NBODY_OK;
    }
    case 4: return copy_out(t->box, t->tsz * 2 * D * size_t(t->nnodes));
    case 5:
      NB_ARG(t->counters != nullptr, "counters were never enabled");
      return copy_out(t->counters, sizeof(uint32_t) * 4 * size_t(t->n));
    default: set_error("nbody_bvh_read: unknown what=%d", what); return NBODY_ERR_ARG;
  }
}

#ifdef NBODY_EXPERIMENTS
// experiments library only (not in nbody_hip.h): the timeline of the last sweep launch made with NBODY_K9_TIMELINE=1
extern "C" long long nbody_exp_k9_timeline(unsigned long long* out, size_t max_words) {
  if (!nbody::g_k9_timeline_host || max_words < nbody::g_k9_timeline_words) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpy(out, nbody::g_k9_timeline_host, nbody::g_k9_timeline_words * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -3;
  return (long long)nbody::g_k9_timeline_words;
}
#endif
