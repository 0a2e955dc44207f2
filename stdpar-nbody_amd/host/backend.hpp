// RAII view of the HIP backend (include/nbody_hip.h) for one host System.  Every failure is fatal in
// the reference's style: message on stderr, exit(EXIT_FAILURE).  There is no CPU path behind this.
#pragma once
#include <cstdlib>
#include <iostream>
#include <type_traits>
#include <utility>
#include <vector>

#include "nbody_hip.h"
#include "system.hpp"

namespace nb {

inline void backend_check(int rc, char const* what) {
  if (rc != NBODY_OK) {
    std::cerr << "HIP backend failure in " << what << ": " << nbody_last_error() << std::endl;
    std::exit(EXIT_FAILURE);
  }
}

template <typename T, int D>
class Device {
 public:
  static constexpr int dtype = std::is_same_v<T, float> ? NBODY_F32 : NBODY_F64;

  explicit Device(System<T, D>& host, int device = 0) : host_(host) {
    backend_check(nbody_create(&ctx_, dtype, D, host.n, device), "nbody_create");
    push();
  }
  ~Device() {
    if (octree_) nbody_octree_destroy(octree_);
    if (tree_) nbody_bvh_destroy(tree_);
    nbody_destroy(ctx_);
  }
  Device(Device const&)            = delete;
  Device& operator=(Device const&) = delete;

  // host System -> device mirrors
  void push() {
    backend_check(nbody_upload(ctx_, host_.m.data(), host_.x.data(), host_.v.data(), host_.a.data(), host_.ao.data(), host_.dt,
                               host_.G),
                  "nbody_upload");
    backend_check(nbody_ctx_state(ctx_, &view_), "nbody_ctx_state");
  }
  // device mirrors -> host System (all five arrays: bvh permutes m too)
  void pull() {
    backend_check(nbody_download(ctx_, host_.m.data(), host_.x.data(), host_.v.data(), host_.a.data(), host_.ao.data()),
                  "nbody_download");
  }
  void pull_positions() { backend_check(nbody_download(ctx_, nullptr, host_.x.data(), nullptr, nullptr, nullptr), "nbody_download"); }
  void sync() { backend_check(nbody_stream_sync(stream()), "nbody_stream_sync"); }
  void* stream() { return nbody_ctx_stream(ctx_); }

  void all_pairs_force() { backend_check(nbody_all_pairs_force(&view_, stream()), "nbody_all_pairs_force"); }
  void all_pairs_collapsed_force() {
    backend_check(nbody_all_pairs_collapsed_force(&view_, stream()), "nbody_all_pairs_collapsed_force");
  }
  void accelerate_step() { backend_check(nbody_accelerate_step(&view_, stream()), "nbody_accelerate_step"); }

  // System::calc_energies (src/system.h:62-79) on the device: {kinetic, potential}
  std::pair<T, T> calc_energies() {
    T ke{}, pe{};
    backend_check(nbody_calc_energies(&view_, &ke, &pe, stream()), "nbody_calc_energies");
    return {ke, pe};
  }

  // Record the phase calls issued by `step` once and return a replayable graph of them (one submission per step).
  template <typename F>
  nbody_graph* record(F&& step) {
    nbody_graph* g = nullptr;
    backend_check(nbody_graph_begin(stream()), "nbody_graph_begin");
    step();
    backend_check(nbody_graph_end(stream(), &g), "nbody_graph_end");
    return g;
  }
  void replay(nbody_graph* g) { backend_check(nbody_graph_launch(g, stream()), "nbody_graph_launch"); }

  void bvh_alloc() {
    if (!tree_) backend_check(nbody_bvh_create(&tree_, dtype, D, host_.n), "nbody_bvh_create");
  }
  void bvh_bounding_box() { backend_check(nbody_bvh_bounding_box(tree_, &view_, stream()), "nbody_bvh_bounding_box"); }
  void bvh_hilbert_sort() { backend_check(nbody_bvh_hilbert_sort(tree_, &view_, stream()), "nbody_bvh_hilbert_sort"); }
  void bvh_build_tree() { backend_check(nbody_bvh_build_tree(tree_, &view_, stream()), "nbody_bvh_build_tree"); }
  void bvh_compute_force(double theta) {
    backend_check(nbody_bvh_compute_force(tree_, &view_, theta, stream()), "nbody_bvh_compute_force");
  }
  // octree phases (src/octree.h)
  void octree_alloc() {
    if (!octree_) backend_check(nbody_octree_create(&octree_, dtype, D, host_.n), "nbody_octree_create");
  }
  void octree_clear() { backend_check(nbody_octree_clear(octree_, stream()), "nbody_octree_clear"); }
  void octree_compute_bounds() { backend_check(nbody_octree_compute_bounds(octree_, &view_, stream()), "nbody_octree_compute_bounds"); }
  void octree_insert() { backend_check(nbody_octree_insert(octree_, &view_, stream()), "nbody_octree_insert"); }
  void octree_compute_tree() { backend_check(nbody_octree_compute_tree(octree_, stream()), "nbody_octree_compute_tree"); }
  void octree_compute_force(double theta) {
    backend_check(nbody_octree_compute_force(octree_, &view_, theta, stream()), "nbody_octree_compute_force");
  }
  // {tree size, total mass}; also where device-side build errors (depth limit, node pool) surface
  std::pair<std::uint32_t, T> octree_info() {
    std::uint32_t size = 0;
    T mass{};
    backend_check(nbody_octree_info(octree_, &size, &mass, stream()), "nbody_octree_info");
    return {size, mass};
  }

  // mass of the root monopole, for --print-info (src/bvh.h:377)
  T bvh_total_mass() {
    std::vector<T> nodes(std::size_t(nbody_bvh_nnodes(tree_)) * (D + 1));
    backend_check(nbody_bvh_read(tree_, 2, nodes.data(), nodes.size() * sizeof(T), stream()), "nbody_bvh_read");
    return nodes[D];
  }

 private:
  System<T, D>& host_;
  nbody_ctx* ctx_  = nullptr;
  nbody_bvh* tree_ = nullptr;
  nbody_octree* octree_ = nullptr;
  nbody_state view_{};
};

}  // namespace nb
