// Step drivers: the L3 layer of the reference (run_all_pairs src/all_pairs.h:52-106, run_bvh
// src/bvh.h:327-418, run_simulation src/main.cpp:19-40) re-stated over the HIP backend.  They keep
// the reference's observable behaviour: which steps are warm-up and which are timed, the CSV text,
// when frames are saved.  Device work is asynchronous, so every wall-clock bracket ends with a
// stream sync.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "backend.hpp"
#include "options.hpp"
#include "saver.hpp"

namespace nb {

using wall_clock = std::chrono::steady_clock;
using seconds_t  = std::chrono::duration<double>;

template <typename F>
seconds_t timed(F&& f) {
  auto t0 = wall_clock::now();
  f();
  return seconds_t(wall_clock::now() - t0);
}

inline std::string fixed2(double v) {
  char b[64];
  std::snprintf(b, sizeof b, "%.2f", v);
  return b;
}

inline void csv_total_guard(Options const& o) {  // src/all_pairs.h:58-62, src/bvh.h:334-339
  if (o.csv_total && (o.print_state || o.print_info || o.save_pos || o.save_energy)) std::abort();
}

template <typename T, int D>
void run_all_pairs(System<T, D>& sys, Device<T, D>& dev, Options o, char const* name, bool collapsed) {
  Saver<T, D> saver(o);
  saver.save_all(sys, dev);
  csv_total_guard(o);
  if (o.csv_total) std::cout << "algorithm,dim,precision,nsteps,nbodies,total [s]\n";  // detailed prints no header here

  auto force = [&] { collapsed ? dev.all_pairs_collapsed_force() : dev.all_pairs_force(); };
  seconds_t t_force(0), t_accel(0), t_total(0);
  if (o.csv_detailed) {
    t_total = timed([&] {
      for (std::size_t step = 0; step < o.steps; ++step) {
        t_force += timed([&] { force(); dev.sync(); });
        t_accel += timed([&] { dev.accelerate_step(); dev.sync(); });
        saver.save_all(sys, dev);
      }
    });
  } else if (dev.sharded()) {
    // several devices driven by this one thread: plain asynchronous launches (5 per device and step against >= 80 ms of force)
    auto one_step = [&] { force(); dev.accelerate_step(); };
    for (std::size_t step = 0; step < o.warmup_steps; ++step) one_step();
    dev.sync();
    t_total = timed([&] {
      for (std::size_t step = o.warmup_steps; step < o.steps; ++step) one_step();
      dev.sync();
    });
    o.steps -= o.warmup_steps;
  } else {
    // the step is a fixed launch sequence: record it once, replay it (hipGraph)
    nbody_graph* g = dev.record([&] { force(); dev.accelerate_step(); });
    for (std::size_t step = 0; step < o.warmup_steps; ++step) dev.replay(g);
    dev.sync();
    t_total = timed([&] {
      for (std::size_t step = o.warmup_steps; step < o.steps; ++step) dev.replay(g);
      dev.sync();
    });
    nbody_graph_destroy(g);
    o.steps -= o.warmup_steps;  // wraps for steps < 10, as printed by the reference
  }
  if (o.csv_detailed || o.csv_total) {
    std::cout << name << ',' << D << ',' << sizeof(T) * 8 << ',' << o.steps << ',' << sys.n << ',' << fixed2(t_total.count());
    if (o.csv_detailed) std::cout << ',' << fixed2(t_force.count()) << ',' << fixed2(t_accel.count());
    std::cout << "\n";
  }
}

template <typename T, int D>
void run_bvh(System<T, D>& sys, Device<T, D>& dev, Options o) {
  Saver<T, D> saver(o);
  saver.save_all(sys, dev);
  csv_total_guard(o);
  if (o.csv_total || o.csv_detailed) {
    std::cout << "algorithm,dim,precision,nsteps,nbodies,total [s]";
    if (o.csv_detailed) std::cout << ",force [s],accel [s],bbox [s],sort [s],multipoles [s],force approx [s]";
    std::cout << "\n";
  }
  dev.bvh_alloc();
  T const theta = T(o.theta);
  seconds_t t_force(0), t_accel(0), t_bbox(0), t_sort(0), t_tree(0), t_walk(0), t_total(0);
  if (o.csv_detailed) {
    t_total = timed([&] {
      for (std::size_t step = 0; step < o.steps; ++step) {
        t_force += timed([&] {
          t_bbox += timed([&] { dev.bvh_bounding_box(); dev.sync(); });
          t_sort += timed([&] { dev.bvh_hilbert_sort(); dev.sync(); });
          t_tree += timed([&] { dev.bvh_build_tree(); dev.sync(); });
          t_walk += timed([&] { dev.bvh_compute_force(theta); dev.sync(); });
        });
        t_accel += timed([&] { dev.accelerate_step(); dev.sync(); });
        if (o.print_info) {
          char b[64];
          std::snprintf(b, sizeof b, "Total mass: % .5f\n", double(dev.bvh_total_mass()));
          std::cout << b;
        }
        saver.save_all(sys, dev);
      }
    });
  } else {
    auto one_step = [&] {
      dev.bvh_bounding_box();
      dev.bvh_hilbert_sort();
      dev.bvh_build_tree();
      dev.bvh_compute_force(theta);
      dev.accelerate_step();
    };
    nbody_graph* g = dev.record(one_step);  // ~40 launches per step -> one graph submission
    for (std::size_t step = 0; step < o.warmup_steps; ++step) dev.replay(g);
    dev.sync();
    t_total = timed([&] {
      for (std::size_t step = o.warmup_steps; step < o.steps; ++step) dev.replay(g);
      dev.sync();
    });
    nbody_graph_destroy(g);
    o.steps -= o.warmup_steps;
  }
  if (o.csv_detailed || o.csv_total) {
    std::cout << "bvh," << D << ',' << sizeof(T) * 8 << ',' << o.steps << ',' << sys.n << ',' << fixed2(t_total.count());
    if (o.csv_detailed)
      std::cout << ',' << fixed2(t_force.count()) << ',' << fixed2(t_accel.count()) << ',' << fixed2(t_bbox.count()) << ','
                << fixed2(t_sort.count()) << ',' << fixed2(t_tree.count()) << ',' << fixed2(t_walk.count());
    std::cout << "\n";
  }
}

// src/octree.h:266-347
template <typename T, int D>
void run_octree(System<T, D>& sys, Device<T, D>& dev, Options o) {
  Saver<T, D> saver(o);
  saver.save_all(sys, dev);
  csv_total_guard(o);
  if (o.csv_total || o.csv_detailed) {
    std::cout << "algorithm,dim,precision,nsteps,nbodies,total [s]";
    if (o.csv_detailed) std::cout << ",force [s],accel [s],clear [s],bbox [s],insert [s],multipoles [s],force approx [s]";
    std::cout << "\n";
  }
  dev.octree_alloc();
  if (o.print_info) std::cout << "Tree init complete\n";
  T const theta = T(o.theta);
  seconds_t t_force(0), t_accel(0), t_clear(0), t_bbox(0), t_insert(0), t_tree(0), t_walk(0), t_total(0);
  if (o.csv_detailed) {
    t_total = timed([&] {
      for (std::size_t step = 0; step < o.steps; ++step) {
        t_force += timed([&] {
          t_clear += timed([&] { dev.octree_clear(); dev.sync(); });
          t_bbox += timed([&] { dev.octree_compute_bounds(); dev.sync(); });
          t_insert += timed([&] { dev.octree_insert(); dev.sync(); });
          t_tree += timed([&] { dev.octree_compute_tree(); dev.sync(); });
          t_walk += timed([&] { dev.octree_compute_force(theta); dev.sync(); });
        });
        t_accel += timed([&] { dev.accelerate_step(); dev.sync(); });
        auto [size, mass] = dev.octree_info();  // also reports a build that hit the depth limit / node pool
        if (o.print_info) {
          char b[96];
          std::snprintf(b, sizeof b, "Tree size: %u\nTotal mass: % .5f\n", size, double(mass));
          std::cout << b;
        }
        saver.save_all(sys, dev);
      }
    });
  } else {
    auto one_step = [&] {
      dev.octree_clear();
      dev.octree_compute_bounds();
      dev.octree_insert();
      dev.octree_compute_tree();
      dev.octree_compute_force(theta);
      dev.accelerate_step();
    };
    one_step();  // the phase-order checks of the ABI need one direct pass before the sequence is recorded
    (void)dev.octree_info();  // ... and tells the library how deep this system's tree is: the recorded step launches that many levels
    nbody_graph* g = dev.record(one_step);
    for (std::size_t step = 1; step < o.warmup_steps; ++step) dev.replay(g);
    dev.sync();
    (void)dev.octree_info();  // a build flagged on the device (bodies that never separate, node pool) stops the run here
    t_total = timed([&] {
      for (std::size_t step = o.warmup_steps; step < o.steps; ++step) dev.replay(g);
      dev.sync();
    });
    nbody_graph_destroy(g);
    (void)dev.octree_info();
    o.steps -= o.warmup_steps;
  }
  if (o.csv_detailed || o.csv_total) {
    std::cout << "octree," << D << ',' << sizeof(T) * 8 << ',' << o.steps << ',' << sys.n << ',' << fixed2(t_total.count());
    if (o.csv_detailed)
      std::cout << ',' << fixed2(t_force.count()) << ',' << fixed2(t_accel.count()) << ',' << fixed2(t_clear.count()) << ','
                << fixed2(t_bbox.count()) << ',' << fixed2(t_insert.count()) << ',' << fixed2(t_tree.count()) << ','
                << fixed2(t_walk.count());
    std::cout << "\n";
  }
}

// src/main.cpp:19-40
template <typename T, int D>
void run_simulation(Options const& o, System<T, D>& sys) {
  bool const quiet = o.csv_total || o.csv_detailed;
  if (o.print_state) {
    std::cout << "Starting state:" << std::endl;
    sys.print();
  }
  if (!quiet) std::cout << "Starting simulation" << std::endl;
  auto t0 = wall_clock::now();
  {
    if (o.gpus > 1 && o.algorithm != Algorithm::AllPairs) {
      std::cerr << "--gpus " << o.gpus << ": bodies shard over GPUs for --algorithm all-pairs only." << std::endl;
      std::exit(EXIT_FAILURE);
    }
    // an explicit --gpus N with all-pairs goes through the communicator, the shard windows and the exchange for ANY N (N = 1
    // is how a one-GPU box exercises that path); without the flag: the reference's single device, no communicator
    bool const exchange = o.algorithm == Algorithm::AllPairs && o.gpus_given;
    Device<T, D> dev(sys, o.gpus, exchange);
    switch (o.algorithm) {
      case Algorithm::AllPairs: run_all_pairs(sys, dev, o, "all-pairs", false); break;
      case Algorithm::AllPairsCollapsed: run_all_pairs(sys, dev, o, "all-pairs-collapsed", true); break;
      case Algorithm::Bvh: run_bvh(sys, dev, o); break;
      case Algorithm::Octree: run_octree(sys, dev, o); break;
    }
    dev.pull();
  }
  auto t1 = wall_clock::now();
  if (o.print_state) {
    std::cout << "Final state:" << std::endl;
    sys.print();
  }
  if (!quiet) {
    char b[96];
    std::snprintf(b, sizeof b, "Done simulation\nTotal time: %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    std::cout << b;
  }
}

}  // namespace nb
