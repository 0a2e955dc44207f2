/* Multi-GPU all-pairs through the C ABI from plain C99: one process, `ngpus` devices (default: argv[1] or 1), one context per
 * device owning the ABI's shard of the targets, K1 + K3 on every device and the one collective of the path — the all-gather
 * of positions (RCCL over xGMI behind nbody_allgather_positions) — every step.  The result is compared with the same steps
 * on a single whole-system context: any device count gives bitwise the same trajectory.  Build:
 *   gcc -std=c99 -Iinclude examples/abi_multi_gpu.c -Lstdpar-nbody_amd -lnbody_hip -Wl,-rpath,$PWD/stdpar-nbody_amd -o abi_multi_gpu
 * tests/test_gpu_cli.py builds it and runs it with 1 device (all a one-GPU box has). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nbody_hip.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    int rc_ = (call);                                                        \
    if (rc_ != NBODY_OK) {                                                   \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, nbody_last_error()); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

#define MAXG 16

int main(int argc, char** argv) {
  const int ngpus  = argc > 1 ? atoi(argv[1]) : 1;
  /* default 5003: not a multiple of 2, 3, 4 or 8, so uneven shards take the grouped send/recv path; argv[2] overrides */
  const uint32_t n = argc > 2 ? (uint32_t)atoi(argv[2]) : 5003u;
  const int steps  = 3;
  if (ngpus < 1 || ngpus > MAXG) return 2;
  double *m = malloc(sizeof(double) * n), *x = malloc(sizeof(double) * 3 * n), *v = calloc(3 * n, sizeof(double)),
         *a = calloc(3 * n, sizeof(double)), *ao = calloc(3 * n, sizeof(double));
  double *rx = malloc(sizeof(double) * 3 * n), *rv = malloc(sizeof(double) * 3 * n), *ra = malloc(sizeof(double) * 3 * n);
  double *gx = malloc(sizeof(double) * 3 * n), *gv = malloc(sizeof(double) * 3 * n), *ga = malloc(sizeof(double) * 3 * n);
  uint32_t lcg = 2024u;
  for (uint32_t i = 0; i < n; ++i) {
    m[i] = 1.0 + 0.001 * i;
    for (int k = 0; k < 3; ++k) {
      lcg          = lcg * 1664525u + 1013904223u;
      x[3 * i + k] = (double)(lcg >> 8) / 16777216.0 - 0.5;
    }
  }

  /* reference: one context, whole system */
  nbody_ctx* whole = NULL;
  nbody_state ws;
  CHECK(nbody_create(&whole, NBODY_F64, 3, n, 0));
  CHECK(nbody_upload(whole, m, x, v, a, ao, 0.01, 1.0));
  CHECK(nbody_ctx_state(whole, &ws));
  for (int s = 0; s < steps; ++s) {
    CHECK(nbody_all_pairs_force(&ws, nbody_ctx_stream(whole)));
    CHECK(nbody_accelerate_step(&ws, nbody_ctx_stream(whole)));
  }
  CHECK(nbody_download(whole, NULL, rx, rv, ra, NULL));

  /* sharded: one context and one communicator per device */
  nbody_ctx* ctx[MAXG];
  nbody_comm* comm[MAXG];
  nbody_state st[MAXG];
  CHECK(nbody_comm_create_all(comm, ngpus, NULL));
  for (int g = 0; g < ngpus; ++g) {
    uint32_t first, count;
    nbody_shard_range(n, ngpus, g, &first, &count);
    CHECK(nbody_create(&ctx[g], NBODY_F64, 3, n, g));
    CHECK(nbody_ctx_set_shard(ctx[g], first, count));
    CHECK(nbody_upload(ctx[g], m, x, v, a, ao, 0.01, 1.0));
    CHECK(nbody_ctx_state(ctx[g], &st[g]));
  }
  for (int s = 0; s < steps; ++s) {
    for (int g = 0; g < ngpus; ++g) CHECK(nbody_all_pairs_force(&st[g], nbody_ctx_stream(ctx[g])));
    for (int g = 0; g < ngpus; ++g) CHECK(nbody_accelerate_step(&st[g], nbody_ctx_stream(ctx[g])));
    if (ngpus > 1) CHECK(nbody_comm_group_begin());
    for (int g = 0; g < ngpus; ++g) CHECK(nbody_allgather_positions(comm[g], &st[g], nbody_ctx_stream(ctx[g])));
    if (ngpus > 1) CHECK(nbody_comm_group_end());
  }
  for (int g = 0; g < ngpus; ++g) CHECK(nbody_download(ctx[g], NULL, gx, gv, ga, NULL)); /* every context writes its own rows */

  const int same = !memcmp(rx, gx, sizeof(double) * 3 * n) && !memcmp(rv, gv, sizeof(double) * 3 * n) &&
                   !memcmp(ra, ga, sizeof(double) * 3 * n);
  printf("devices %d, rccl %d, world %d: %s\n", ngpus, nbody_comm_rccl_version(), nbody_comm_world(comm[0]),
         same ? "bitwise equal to the single-context run" : "MISMATCH");
  for (int g = 0; g < ngpus; ++g) {
    nbody_comm_destroy(comm[g]);
    nbody_destroy(ctx[g]);
  }
  nbody_destroy(whole);
  return same ? 0 : 1;
}
