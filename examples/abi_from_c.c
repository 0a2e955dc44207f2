/* The boundary used from plain C99: a 3D double system of n bodies in a cube, one all-pairs force pass, one octree force
 * pass and one leapfrog step through include/nbody_hip.h, nothing else.  Build (tests/test_gpu_cli.py does this):
 *   gcc -std=c99 -Iinclude examples/abi_from_c.c -Lstdpar-nbody_amd -lnbody_hip -Wl,-rpath,$PWD/stdpar-nbody_amd -lm -o abi_from_c
 * Prints the two maximum accelerations and their relative difference (theta = 0: the tree never approximates). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "nbody_hip.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    int rc_ = (call);                                                        \
    if (rc_ != NBODY_OK) {                                                   \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, nbody_last_error()); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

int main(void) {
  const uint32_t n = 1000;
  double *m = malloc(sizeof(double) * n), *x = malloc(sizeof(double) * 3 * n), *v = calloc(3 * n, sizeof(double)),
         *a = calloc(3 * n, sizeof(double)), *ao = calloc(3 * n, sizeof(double)), *a2 = calloc(3 * n, sizeof(double));
  uint32_t lcg = 12345u;  /* bodies spread over a volume: the node pool is the reference's 2^dim * n (src/system.h:30),
                             which points on a curve would exhaust */
  for (uint32_t i = 0; i < n; ++i) {
    m[i] = 1.0 + 0.001 * i;
    for (int k = 0; k < 3; ++k) {
      lcg          = lcg * 1664525u + 1013904223u;
      x[3 * i + k] = (double)(lcg >> 8) / 16777216.0 - 0.5;
    }
  }
  if (nbody_abi_version() / 1000 != NBODY_HIP_ABI_VERSION / 1000) {
    fprintf(stderr, "ABI mismatch: library %d, header %d\n", nbody_abi_version(), NBODY_HIP_ABI_VERSION);
    return 1;
  }
  nbody_ctx* ctx = NULL;
  CHECK(nbody_create(&ctx, NBODY_F64, 3, n, 0));
  CHECK(nbody_upload(ctx, m, x, v, a, ao, 0.01, 1.0));
  nbody_state st;
  CHECK(nbody_ctx_state(ctx, &st));
  void* stream = nbody_ctx_stream(ctx);

  CHECK(nbody_all_pairs_force(&st, stream));
  CHECK(nbody_download(ctx, NULL, NULL, NULL, a, NULL)); /* NBODY_ERR_STATE here if K1's chunk hand-off had failed (ABI 2.3) */
  uint64_t k1[6];
  CHECK(nbody_all_pairs_status(stream, k1, 0)); /* {failed, block, group, chunk, polls, waves that waited} of this stream's K1 launches */

  nbody_octree* tree = NULL;
  CHECK(nbody_octree_create(&tree, NBODY_F64, 3, n));
  CHECK(nbody_octree_clear(tree, stream));
  CHECK(nbody_octree_compute_bounds(tree, &st, stream));
  CHECK(nbody_octree_insert(tree, &st, stream));
  CHECK(nbody_octree_compute_tree(tree, stream));
  CHECK(nbody_octree_compute_force(tree, &st, 0.0, stream));
  uint32_t tree_size = 0;
  double root_mass   = 0.0;
  CHECK(nbody_octree_info(tree, &tree_size, &root_mass, stream));
  CHECK(nbody_download(ctx, NULL, NULL, NULL, a2, NULL));
  CHECK(nbody_accelerate_step(&st, stream));
  CHECK(nbody_stream_sync(stream));

  double amax = 0.0, a2max = 0.0, dmax = 0.0;
  for (uint32_t i = 0; i < 3 * n; ++i) {
    if (fabs(a[i]) > amax) amax = fabs(a[i]);
    if (fabs(a2[i]) > a2max) a2max = fabs(a2[i]);
    if (fabs(a[i] - a2[i]) > dmax) dmax = fabs(a[i] - a2[i]);
  }
  printf("all-pairs max|a| %.12e\noctree    max|a| %.12e\nrelative difference %.3e\ntree size %u root mass %.6f\n", amax, a2max,
         dmax / amax, tree_size, root_mass);
  printf("K1 hand-off: failed %llu, waves that waited %llu\n", (unsigned long long)k1[0], (unsigned long long)k1[5]);
  nbody_octree_destroy(tree);
  nbody_destroy(ctx);
  free(m); free(x); free(v); free(a); free(ao); free(a2);
  return dmax / amax < 1e-9 ? 0 : 2;
}
