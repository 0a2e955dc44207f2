/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU oracle for the stdpar-nbody hot path: a plain-C restatement of the reference's all-pairs,
 * all-pairs-collapsed, leapfrog, Hilbert-BVH (bbox, keys, sort, build, traversal), energy and
 * workload-generator algorithms.  The algorithms live in nbody_oracle_impl.inc (one instantiation
 * per T in {float,double} x D in {2,3}); this file adds the RNG restatement, the key sort and a
 * runtime-dispatched C entry per function so tests can drive it through ctypes.
 *
 * Parity pinning: tests/test_oracle_vs_ref.py checks this oracle BIT-FOR-BIT against outputs of the
 * real reference compiled from /root/reference (oracle/_ref, recipe in oracle/Makefile) and against
 * the committed fixtures under tests/golden/ generated from that same build.
 *
 * Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
 * (stdpar-nbody_amd/) never links or calls anything in this directory.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- std::mt19937 + libstdc++ uniform_real_distribution<double> ---------------------------- */
/* system.h:22-25: std::mt19937 gen{42}; uniform_real_distribution<> (double) for angle/unit/sym.
 * mt19937 is the published MT19937 (Matsumoto & Nishimura 1998) with the C++11 seeding
 * x[i] = 1812433253 * (x[i-1] ^ (x[i-1] >> 30)) + i.  libstdc++'s distribution draws
 * generate_canonical<double,53>: two 32-bit outputs, sum = g1 + g2 * 2^32 (in double),
 * ret = sum / 2^64, clamped below 1; result = ret * (b - a) + a  (bits/random.tcc, random.h). */
typedef struct {
  uint32_t mt[624];
  int idx;
} oracle_rng;

static void oracle_rng_seed(oracle_rng* g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

static uint32_t oracle_rng_next(oracle_rng* g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
      if (y & 1u) v ^= 0x9908b0dfu;
      g->mt[i] = v;
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

static double oracle_uniform(oracle_rng* g, double a, double b) {
  double sum = 0.0, tmp = 1.0;
  for (int k = 0; k < 2; ++k) {
    sum += (double)oracle_rng_next(g) * tmp;
    tmp *= 4294967296.0;
  }
  double ret = sum / tmp;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret * (b - a) + a;
}

/* ---- instantiate the algorithms ----------------------------------------------------------- */
#define T float
#define IS_F32 1
#define D 2
#define SFX f2
#include "nbody_oracle_impl.inc"
#undef D
#undef SFX
#define D 3
#define SFX f3
#include "nbody_oracle_impl.inc"
#undef D
#undef SFX
#undef T
#undef IS_F32

#define T double
#define IS_F32 0
#define D 2
#define SFX d2
#include "nbody_oracle_impl.inc"
#undef D
#undef SFX
#define D 3
#define SFX d3
#include "nbody_oracle_impl.inc"
#undef D
#undef SFX
#undef T
#undef IS_F32

/* ---- key sort: ascending by key, ties by original index (see impl .inc apply_perm note) ----- */
typedef struct {
  uint64_t key;
  uint32_t idx;
} key_idx;

static int key_idx_cmp(const void* pa, const void* pb) {
  const key_idx* a = (const key_idx*)pa;
  const key_idx* b = (const key_idx*)pb;
  if (a->key != b->key) return a->key < b->key ? -1 : 1;
  return a->idx < b->idx ? -1 : (a->idx > b->idx ? 1 : 0);
}

void oracle_sort_keys(const uint64_t* keys, uint32_t sz, uint32_t* perm) {
  key_idx* t = (key_idx*)malloc(sizeof(key_idx) * sz);
  for (uint32_t i = 0; i < sz; ++i) { t[i].key = keys[i]; t[i].idx = i; }
  qsort(t, sz, sizeof(key_idx), key_idx_cmp);
  for (uint32_t i = 0; i < sz; ++i) perm[i] = t[i].idx;
  free(t);
}

/* ---- runtime dispatch (dtype: 0 = f32, 1 = f64; dim: 2 | 3) ---------------------------------- */
#define DISPATCH(dtype, dim, CALL)                           \
  do {                                                       \
    if ((dtype) == 0 && (dim) == 2) { CALL(float, f2); }     \
    else if ((dtype) == 0 && (dim) == 3) { CALL(float, f3); } \
    else if ((dtype) == 1 && (dim) == 2) { CALL(double, d2); } \
    else if ((dtype) == 1 && (dim) == 3) { CALL(double, d3); } \
    else return -1;                                          \
  } while (0)

int oracle_all_pairs_force(int dtype, int dim, const void* m, const void* x, void* a, double c, uint32_t sz, uint32_t first,
                           uint32_t count) {
#define CALL(TT, S) all_pairs_force_##S((const TT*)m, (const TT*)x, (TT*)a, (TT)c, sz, first, count)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return 0;
}

int oracle_all_pairs_collapsed_force(int dtype, int dim, const void* m, const void* x, void* a, const void* ao, double c,
                                     uint32_t sz, int mode) {
#define CALL(TT, S) all_pairs_collapsed_force_##S((const TT*)m, (const TT*)x, (TT*)a, (const TT*)ao, (TT)c, sz, mode)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return 0;
}

int oracle_accelerate_step(int dtype, int dim, void* x, void* v, const void* a, void* ao, double dt, uint32_t count) {
#define CALL(TT, S) accelerate_step_##S((TT*)x, (TT*)v, (const TT*)a, (TT*)ao, (TT)dt, count)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return 0;
}

int oracle_calc_energies(int dtype, int dim, const void* m, const void* x, const void* v, double c, uint32_t sz, void* ke,
                         void* pe) {
#define CALL(TT, S) calc_energies_##S((const TT*)m, (const TT*)x, (const TT*)v, (TT)c, sz, (TT*)ke, (TT*)pe)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return 0;
}

int oracle_calc_energies_wide(int dtype, int dim, const void* m, const void* x, const void* v, double c, uint32_t sz, double* ke,
                              double* pe) {
#define CALL(TT, S) calc_energies_wide_##S((const TT*)m, (const TT*)x, (const TT*)v, (TT)c, sz, ke, pe)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return 0;
}

int oracle_bounding_box(int dtype, int dim, const void* x, uint32_t sz, void* xmin, void* xmax) {
#define CALL(TT, S) bounding_box_##S((const TT*)x, sz, (TT*)xmin, (TT*)xmax)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return 0;
}

int oracle_hilbert_keys(int dtype, int dim, const void* x, uint32_t sz, const void* xmin, const void* xmax, uint64_t* keys) {
#define CALL(TT, S) hilbert_keys_##S((const TT*)x, sz, (const TT*)xmin, (const TT*)xmax, keys)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return 0;
}

uint64_t oracle_hilbert_cell(int dim, const uint32_t* cell) { return dim == 2 ? hilbert_d2(cell) : hilbert_d3(cell); }
uint64_t oracle_interleave_bits(int dim, const uint32_t* cell) {
  return dim == 2 ? interleave_bits_d2(cell) : interleave_bits_d3(cell);
}

int oracle_apply_perm(int dtype, int dim, void* m, void* x, void* v, void* a, void* ao, uint32_t sz, const uint32_t* perm) {
#define CALL(TT, S) apply_perm_##S((TT*)m, (TT*)x, (TT*)v, (TT*)a, (TT*)ao, sz, perm)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return 0;
}

/* bvh.h:147-164 alloc: nleafs = bit_ceil(N); nlevels = countr_zero(nleafs); last_level = nlevels-1;
 * nnodes = 2^nlevels - 1 */
uint32_t oracle_bvh_nlevels(uint32_t sz) {
  uint32_t nleafs = 1, nl = 0;
  while (nleafs < sz) { nleafs <<= 1; ++nl; }
  return nl;
}

/* node arrays are caller-allocated: nm[nnodes*(D+1)], nb[nnodes*2D], nbw[nnodes] */
int oracle_bvh_build(int dtype, int dim, const void* m, const void* x, uint32_t sz, void* nm, void* nb, void* nbw) {
  uint32_t nl = oracle_bvh_nlevels(sz);
  if (nl == 0) return -2;
#define CALL(TT, S)                                                          \
  bvh_t_##S t = {nl - 1, (1u << nl) - 1u, (TT*)nm, (TT*)nb, (TT*)nbw};      \
  bvh_build_##S(&t, (const TT*)m, (const TT*)x, sz)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return 0;
}

int oracle_bvh_force(int dtype, int dim, const void* m, const void* x, void* a, double c, uint32_t sz, double theta,
                     const void* nm, const void* nbw, uint32_t* counts) {
  uint32_t nl = oracle_bvh_nlevels(sz);
  if (nl == 0) return -2;
#define CALL(TT, S)                                                              \
  bvh_t_##S t = {nl - 1, (1u << nl) - 1u, (TT*)nm, (TT*)0, (TT*)nbw};           \
  bvh_force_##S(&t, (const TT*)m, (const TT*)x, (TT*)a, (TT)c, sz, (TT)theta, counts)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return 0;
}

/* bvh.h:246-248 can_approximate on arrays: out[i] = bw[i] * bw[i] < theta^2 * d2[i], in T, with theta^2 = theta * theta in T as
 * compute_force forms it (bvh.h:252).  Used by the tests of the product's one-compare form of this test. */
int oracle_can_approximate(int dtype, const void* bw, double theta, const void* d2, uint64_t n, uint8_t* out) {
  if (dtype == 0) {
    const float *w = (const float*)bw, *d = (const float*)d2;
    const float th = (float)theta, th2 = th * th;
    for (uint64_t i = 0; i < n; ++i) out[i] = w[i] * w[i] < th2 * d[i];
  } else if (dtype == 1) {
    const double *w = (const double*)bw, *d = (const double*)d2;
    const double th2 = theta * theta;
    for (uint64_t i = 0; i < n; ++i) out[i] = w[i] * w[i] < th2 * d[i];
  } else {
    return -1;
  }
  return 0;
}

/* One full bvh force phase as run_bvh does it per step (bvh.h:382-393): bbox, keys+sort (permutes
 * m,x,v,a,ao in place), build, traversal.  Scratch is allocated here. */
int oracle_bvh_step_force(int dtype, int dim, void* m, void* x, void* v, void* a, void* ao, double c, uint32_t sz,
                          double theta) {
  size_t ts   = dtype == 0 ? 4 : 8;
  uint32_t nl = oracle_bvh_nlevels(sz);
  if (nl == 0) return -2;
  uint32_t nn = (1u << nl) - 1u;
  char xmin[24], xmax[24];
  uint64_t* keys = (uint64_t*)malloc(8 * (size_t)sz);
  uint32_t* perm = (uint32_t*)malloc(4 * (size_t)sz);
  void* nm       = malloc(ts * (size_t)nn * (dim + 1));
  void* nb       = malloc(ts * (size_t)nn * 2 * dim);
  void* nbw      = malloc(ts * (size_t)nn);
  oracle_bounding_box(dtype, dim, x, sz, xmin, xmax);
  oracle_hilbert_keys(dtype, dim, x, sz, xmin, xmax, keys);
  oracle_sort_keys(keys, sz, perm);
  oracle_apply_perm(dtype, dim, m, x, v, a, ao, sz, perm);
  oracle_bvh_build(dtype, dim, m, x, sz, nm, nb, nbw);
  oracle_bvh_force(dtype, dim, m, x, a, c, sz, theta, nm, nbw, NULL);
  free(keys); free(perm); free(nm); free(nb); free(nbw);
  return 0;
}

/* workload: 0 = uniform, 1 = plummer (3D only), 2 = galaxy.  Arrays sized for n bodies.
 * Returns the system size (galaxy: 2*(n/2.0) truncated), or <0 on error. */
int64_t oracle_build_model(int dtype, int dim, int workload, uint32_t n, void* m, void* x, void* v, double* dt, double* c) {
  if (workload == 0) {
#define CALL(TT, S) model_uniform_##S(n, (TT*)m, (TT*)x, (TT*)v, dt, c)
    DISPATCH(dtype, dim, CALL);
#undef CALL
    return n;
  }
  if (workload == 2) {
    if (n < 2) return -4; /* models.h:112-136 writes the second central mass at index 1 whatever the size */
    uint32_t sz = 0;
#define CALL(TT, S) sz = model_galaxy_##S(n, (TT*)m, (TT*)x, (TT*)v, dt, c)
    DISPATCH(dtype, dim, CALL);
#undef CALL
    return sz;
  }
  if (workload == 1) {
    if (dim != 3) return -3; /* models.h:68-71: throws for D != 3 */
    if (dtype == 0) model_plummer_f3(n, (float*)m, (float*)x, (float*)v, dt, c);
    else model_plummer_d3(n, (double*)m, (double*)x, (double*)v, dt, c);
    return n;
  }
  return -1;
}

/* One octree force phase (octree.h:321-326).  counts: u32[sz][2] {nodes examined, terms} or NULL; tree_size and
 * root_mass (T) may be NULL.  Returns -2 when the tree overflows its capacity (coincident bodies). */
int oracle_octree_step_force(int dtype, int dim, const void* m, const void* x, void* a, double c, uint32_t sz, double theta,
                             uint32_t* counts, uint32_t* tree_size, void* root_mass) {
  int rc = 0;
#define CALL(TT, S) rc = octree_step_force_##S((const TT*)m, (const TT*)x, (TT*)a, (TT)c, sz, (TT)theta, counts, tree_size, (TT*)root_mass)
  DISPATCH(dtype, dim, CALL);
#undef CALL
  return rc < 0 ? -2 : 0;
}
