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

// One host System on `ngpus` devices.  ngpus == 1: the reference's single device, whole system.  ngpus > 1 (all-pairs only;
// the reference has no counterpart): one context per device, each owning the ABI's contiguous shard of the targets
// (nbody_shard_range) and all N sources; one host thread drives them all, asynchronously, and after every
// accelerate_step the devices exchange their position shards (nbody_allgather_positions, RCCL over xGMI).
template <typename T, int D>
class Device {
 public:
  static constexpr int dtype = std::is_same_v<T, float> ? NBODY_F32 : NBODY_F64;

  // exchange: run the collective even with one device (an explicit `--gpus 1`: the multi-GPU path on a one-GPU box)
  explicit Device(System<T, D>& host, int ngpus = 1, bool exchange = false) : host_(host), sharded_(ngpus > 1 || exchange) {
    ctx_.resize(std::size_t(ngpus), nullptr);
    view_.resize(std::size_t(ngpus));
    for (int g = 0; g < ngpus; ++g) backend_check(nbody_create(&ctx_[std::size_t(g)], dtype, D, host.n, g), "nbody_create");
    if (sharded_) {
      comm_.resize(std::size_t(ngpus), nullptr);
      backend_check(nbody_comm_create_all(comm_.data(), ngpus, nullptr), "nbody_comm_create_all");
      for (int g = 0; g < ngpus; ++g) {
        std::uint32_t first = 0, count = 0;
        nbody_shard_range(host.n, ngpus, g, &first, &count);
        backend_check(nbody_ctx_set_shard(ctx_[std::size_t(g)], first, count), "nbody_ctx_set_shard");
      }
    }
    push();
  }
  ~Device() {
    if (octree_) nbody_octree_destroy(octree_);
    if (tree_) nbody_bvh_destroy(tree_);
    for (auto* c : comm_) nbody_comm_destroy(c);
    for (auto* c : ctx_) nbody_destroy(c);
  }
  Device(Device const&)            = delete;
  Device& operator=(Device const&) = delete;

  int ngpus() const { return int(ctx_.size()); }
  bool sharded() const { return sharded_; }
  bool multi() const { return ctx_.size() > 1; }

  // host System -> device mirrors (every device gets everything; it uses all of m and x and its own rows of v, a, ao)
  void push() {
    for (std::size_t g = 0; g < ctx_.size(); ++g) {
      backend_check(nbody_upload(ctx_[g], host_.m.data(), host_.x.data(), host_.v.data(), host_.a.data(), host_.ao.data(),
                                 host_.dt, host_.G),
                    "nbody_upload");
      backend_check(nbody_ctx_state(ctx_[g], &view_[g]), "nbody_ctx_state");
    }
  }
  // device mirrors -> host System (all five arrays: bvh permutes m too); a sharded context writes its own rows
  void pull() {
    for (std::size_t g = 0; g < ctx_.size(); ++g)
      backend_check(nbody_download(ctx_[g], g == 0 ? host_.m.data() : nullptr, host_.x.data(), host_.v.data(), host_.a.data(),
                                   host_.ao.data()),
                    "nbody_download");
  }
  void pull_positions() {
    for (auto* c : ctx_) backend_check(nbody_download(c, nullptr, host_.x.data(), nullptr, nullptr, nullptr), "nbody_download");
  }
  void sync() {
    for (auto* c : ctx_) backend_check(nbody_stream_sync(nbody_ctx_stream(c)), "nbody_stream_sync");
  }
  void* stream(std::size_t g = 0) { return nbody_ctx_stream(ctx_[g]); }

  void all_pairs_force() {
    for (std::size_t g = 0; g < ctx_.size(); ++g) backend_check(nbody_all_pairs_force(&view_[g], stream(g)), "nbody_all_pairs_force");
  }
  void all_pairs_collapsed_force() {
    single("all-pairs-collapsed");
    backend_check(nbody_all_pairs_collapsed_force(&view_[0], stream()), "nbody_all_pairs_collapsed_force");
  }
  // K3 on every device's shard, then the one exchange of the path
  void accelerate_step() {
    for (std::size_t g = 0; g < ctx_.size(); ++g) backend_check(nbody_accelerate_step(&view_[g], stream(g)), "nbody_accelerate_step");
    if (!sharded_) return;
    if (multi()) backend_check(nbody_comm_group_begin(), "nbody_comm_group_begin");
    for (std::size_t g = 0; g < ctx_.size(); ++g)
      backend_check(nbody_allgather_positions(comm_[g], &view_[g], stream(g)), "nbody_allgather_positions");
    if (multi()) backend_check(nbody_comm_group_end(), "nbody_comm_group_end");
  }

  // System::calc_energies (src/system.h:62-79) on the device: {kinetic, potential}
  std::pair<T, T> calc_energies() {
    single("--save energy");
    T ke{}, pe{};
    backend_check(nbody_calc_energies(&view_[0], &ke, &pe, stream()), "nbody_calc_energies");
    return {ke, pe};
  }

  // Record the phase calls issued by `step` once and return a replayable graph of them (one submission per step).
  template <typename F>
  nbody_graph* record(F&& step) {
    single("step graphs");
    nbody_graph* g = nullptr;
    backend_check(nbody_graph_begin(stream()), "nbody_graph_begin");
    step();
    backend_check(nbody_graph_end(stream(), &g), "nbody_graph_end");
    return g;
  }
  void replay(nbody_graph* g) { backend_check(nbody_graph_launch(g, stream()), "nbody_graph_launch"); }

  void bvh_alloc() {
    single("bvh");
    if (!tree_) backend_check(nbody_bvh_create_on(&tree_, dtype, D, host_.n, 0), "nbody_bvh_create_on");
  }
  void bvh_bounding_box() { backend_check(nbody_bvh_bounding_box(tree_, &view_[0], stream()), "nbody_bvh_bounding_box"); }
  void bvh_hilbert_sort() { backend_check(nbody_bvh_hilbert_sort(tree_, &view_[0], stream()), "nbody_bvh_hilbert_sort"); }
  void bvh_build_tree() { backend_check(nbody_bvh_build_tree(tree_, &view_[0], stream()), "nbody_bvh_build_tree"); }
  void bvh_compute_force(double theta) {
    backend_check(nbody_bvh_compute_force(tree_, &view_[0], theta, stream()), "nbody_bvh_compute_force");
  }
  // octree phases (src/octree.h)
  void octree_alloc() {
    single("octree");
    if (!octree_) backend_check(nbody_octree_create_on(&octree_, dtype, D, host_.n, 0), "nbody_octree_create_on");
  }
  void octree_clear() { backend_check(nbody_octree_clear(octree_, stream()), "nbody_octree_clear"); }
  void octree_compute_bounds() { backend_check(nbody_octree_compute_bounds(octree_, &view_[0], stream()), "nbody_octree_compute_bounds"); }
  void octree_insert() { backend_check(nbody_octree_insert(octree_, &view_[0], stream()), "nbody_octree_insert"); }
  void octree_compute_tree() { backend_check(nbody_octree_compute_tree(octree_, stream()), "nbody_octree_compute_tree"); }
  void octree_compute_force(double theta) {
    backend_check(nbody_octree_compute_force(octree_, &view_[0], theta, stream()), "nbody_octree_compute_force");
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
  // bodies shard over GPUs for all-pairs only (north_star); everything else runs on one device
  void single(char const* what) const {
    if (sharded_) {
      std::cerr << what << " runs on one GPU only (bodies shard over --gpus N for --algorithm all-pairs)" << std::endl;
      std::exit(EXIT_FAILURE);
    }
  }
  System<T, D>& host_;
  bool sharded_ = false;
  std::vector<nbody_ctx*> ctx_;
  std::vector<nbody_comm*> comm_;
  std::vector<nbody_state> view_;
  nbody_bvh* tree_ = nullptr;
  nbody_octree* octree_ = nullptr;
};

}  // namespace nb
