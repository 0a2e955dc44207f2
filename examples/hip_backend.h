// src/hip_backend.h  (new file in the reference tree)
//
// The reference-side binding of include/nbody_hip.h: what a maintainer of UoB-HPC/stdpar-nbody adds to call the gfx950
// backend in place of the stdpar algorithms.  Interface being bound: System<T,N> and its state_t view (src/system.h:13-50);
// call sites replaced: src/all_pairs.h:17, src/system.h:55, src/bvh.h:382-393, src/octree.h:321-326, src/main.cpp:19-40.
// Needs only ISO C++20 and -lnbody_hip (no HIP headers on the host side).  INTEGRATION.md quotes this file verbatim and
// tests/test_integration_stub.py compiles it against the reference's own headers and links it.
#pragma once
#include <cstdlib>
#include <iostream>
#include <memory>
#include <vector>

#include <nbody_hip.h>

#include "system.h"

template <typename T, dim_t N>
struct hip_mirror {                       // one per System (and per device); device-resident state between steps
  static constexpr int dtype = sizeof(T) == 4 ? NBODY_F32 : NBODY_F64;
  nbody_ctx* ctx       = nullptr;
  nbody_bvh* tree      = nullptr;
  nbody_octree* octree = nullptr;
  nbody_state st{};
  int device;

  // rank / world select the shard of targets this device owns (multi-GPU all-pairs); 0 / 1 = the whole system
  explicit hip_mirror(System<T, N>& s, int device = 0, int rank = 0, int world = 1) : device(device) {
    check(nbody_create(&ctx, dtype, N, s.size, device));
    if (world > 1) {
      uint32_t first, count;
      nbody_shard_range(s.size, world, rank, &first, &count);
      check(nbody_ctx_set_shard(ctx, first, count));
    }
    upload(s);
  }
  hip_mirror(hip_mirror const&)            = delete;
  hip_mirror& operator=(hip_mirror const&) = delete;
  ~hip_mirror() {
    if (tree) nbody_bvh_destroy(tree);
    if (octree) nbody_octree_destroy(octree);
    nbody_destroy(ctx);
  }

  void upload(System<T, N>& s) {          // vec<T,N> is N contiguous T (src/vec.h:17-19): plain copies of the vectors
    check(nbody_upload(ctx, s.m.data(), s.x.data(), s.v.data(), s.a.data(), s.ao.data(), s.dt, s.constant));
    check(nbody_ctx_state(ctx, &st));     // st = this context's view (the shard's window when sharded)
  }
  void download(System<T, N>& s, bool masses = true) {   // before print() / Saver::save_all / end of run; blocking
    check(nbody_download(ctx, masses ? s.m.data() : nullptr, s.x.data(), s.v.data(), s.a.data(), s.ao.data()));
  }
  void* stream() { return nbody_ctx_stream(ctx); }
  void sync() { check(nbody_stream_sync(stream())); }    // inside each time([&]{...}) of the --csv-detailed loops
  static void check(int rc) {             // reference convention: message + exit(EXIT_FAILURE)
    if (rc) {
      std::cerr << nbody_last_error() << std::endl;
      std::exit(EXIT_FAILURE);
    }
  }
};

// src/all_pairs.h:14-27  becomes
template <typename T, dim_t N>
void all_pairs_force(hip_mirror<T, N>& d) {
  d.check(nbody_all_pairs_force(&d.st, d.stream()));
}

// src/all_pairs.h:29-50  becomes (the intended semantics: all components, 64-bit pair space)
template <typename T, dim_t N>
void all_pairs_collapsed_force(hip_mirror<T, N>& d) {
  d.check(nbody_all_pairs_collapsed_force(&d.st, d.stream()));
}

// src/system.h:52-60  becomes
template <typename T, dim_t N>
void accelerate_step(hip_mirror<T, N>& d) {
  d.check(nbody_accelerate_step(&d.st, d.stream()));
}

// src/system.h:62-79  becomes (blocking; writes one T each)
template <typename T, dim_t N>
auto calc_energies(hip_mirror<T, N>& d) -> std::tuple<T, T> {
  T kinetic, potential;
  d.check(nbody_calc_energies(&d.st, &kinetic, &potential, d.stream()));
  return {kinetic, potential};
}

// src/bvh.h:382-393 (the `kernels` lambda of run_bvh) becomes
template <typename T, dim_t N>
void bvh_force(hip_mirror<T, N>& d, T theta) {
  if (!d.tree) d.check(nbody_bvh_create_on(&d.tree, d.dtype, N, d.st.sz, d.device));
  d.check(nbody_bvh_bounding_box(d.tree, &d.st, d.stream()));
  d.check(nbody_bvh_hilbert_sort(d.tree, &d.st, d.stream()));   // permutes m, x, v, a, ao in place, as src/bvh.h:47-95
  d.check(nbody_bvh_build_tree(d.tree, &d.st, d.stream()));
  d.check(nbody_bvh_compute_force(d.tree, &d.st, theta, d.stream()));
}

// src/octree.h:321-326 (the force phase of run_octree's step) becomes
template <typename T, dim_t N>
void octree_force(hip_mirror<T, N>& d, T theta) {
  if (!d.octree) d.check(nbody_octree_create_on(&d.octree, d.dtype, N, d.st.sz, d.device));
  d.check(nbody_octree_clear(d.octree, d.stream()));
  d.check(nbody_octree_compute_bounds(d.octree, &d.st, d.stream()));
  d.check(nbody_octree_insert(d.octree, &d.st, d.stream()));
  d.check(nbody_octree_compute_tree(d.octree, d.stream()));
  d.check(nbody_octree_compute_force(d.octree, &d.st, theta, d.stream()));
}

// Multi-GPU all-pairs in one process (no reference counterpart: the reference is one process, one device).  Bodies shard by
// target; every device keeps all of m and x; the one exchange per step is the all-gather of x (RCCL over xGMI, in place).
template <typename T, dim_t N>
struct hip_multi {
  std::vector<std::unique_ptr<hip_mirror<T, N>>> d;   // one mirror per device, each with its shard window
  std::vector<nbody_comm*> comm;

  hip_multi(System<T, N>& s, int ngpus) : comm(ngpus, nullptr) {
    hip_mirror<T, N>::check(nbody_comm_create_all(comm.data(), ngpus, /*devices*/ nullptr));   // ncclCommInitAll
    for (int g = 0; g < ngpus; ++g) d.push_back(std::make_unique<hip_mirror<T, N>>(s, g, g, ngpus));
  }
  ~hip_multi() {
    d.clear();
    for (auto* c : comm)
      if (c) nbody_comm_destroy(c);
  }

  // the `kernels` lambda of run_all_pairs (src/all_pairs.h:86-91) becomes
  void step() {
    auto check = hip_mirror<T, N>::check;
    for (auto& m : d) check(nbody_all_pairs_force(&m->st, m->stream()));
    for (auto& m : d) check(nbody_accelerate_step(&m->st, m->stream()));
    check(nbody_comm_group_begin());                    // one thread, several devices: group the calls
    for (std::size_t g = 0; g < d.size(); ++g) check(nbody_allgather_positions(comm[g], &d[g]->st, d[g]->stream()));
    check(nbody_comm_group_end());
  }
  // before print() / save: every context downloads its own rows into the one host System
  void download(System<T, N>& s) {
    for (std::size_t g = 0; g < d.size(); ++g) d[g]->download(s, /*masses*/ g == 0);
  }
};

// The reference's own seam for the force phase is the generic callable of run_all_pairs (`Force&& f`, invoked as f(system):
// src/all_pairs.h:52-53,77,88).  Passing this callable runs K1 on the GPU under the reference's UNMODIFIED driver, integrator,
// Saver and CLI (state crosses PCIe every step: the thinnest possible drop-in; the device-resident loop uses the functions above).
template <typename T, dim_t N>
auto hip_all_pairs_callable(hip_mirror<T, N>& d) {
  return [&d](System<T, N>& s) {
    d.upload(s);
    all_pairs_force(d);
    d.download(s);
  };
}
