// The reference's own program with the gfx950 backend dropped in — the proof that examples/hip_backend.h (the stub of
// INTEGRATION.md) binds the interface it claims to bind.  This translation unit includes the UNMODIFIED reference sources where
// they lie (-I /root/reference/src; nothing is copied): its CLI (src/arguments.h), generators (src/models.h), System / print()
// (src/system.h), Saver (src/saving.h), run_simulation (src/main.cpp:19-40) and step driver run_all_pairs (src/all_pairs.h:52-106)
// all stay the reference's; only the force phase — and, in the resident loop, the integrator — goes through the C ABI.
//
// Built by `make -C oracle ref_hip` into oracle/_ref/nbody_ref_hip_d{2,3} (it embeds reference code, so it lives with the other
// reference binaries: git-ignored, travels to the GPU box).  TEST INFRASTRUCTURE: tests/test_integration_stub.py builds it here
// (no GPU needed to compile and link) and tests/test_gpu_cli.py runs it on the GPU box against the reference's golden text.
//
//   all-pairs, default:              the reference's run_all_pairs, UNMODIFIED, with hip_all_pairs_callable as its Force&& f
//   NBODY_HIP_RESIDENT=1, any algo:  device-resident loop (all_pairs_force / bvh_force / octree_force + accelerate_step on the
//                                    GPU), the reference's step-count semantics of the default mode (max(steps, warmup) steps)
#define main reference_main   // the reference's main() is compiled too; this file provides the entry point
#include "main.cpp"
#undef main

#include "hip_backend.h"

template <typename T, dim_t N>
void run_all_pairs_hip(System<T, N>& system, Arguments arguments) {   // a sim_func_t<T, N> (src/main.cpp:16-17)
  hip_mirror<T, N> d(system);
  run_all_pairs(system, arguments, "all-pairs", hip_all_pairs_callable(d));
}

template <typename T, dim_t N>
void run_resident_hip(System<T, N>& system, Arguments arguments) {    // a sim_func_t<T, N>; default output mode only
  if (arguments.csv_total || arguments.csv_detailed || arguments.save_pos || arguments.save_energy) {
    std::cerr << "the resident demo loop prints no CSV and saves nothing" << std::endl;
    std::exit(EXIT_FAILURE);
  }
  hip_mirror<T, N> d(system);
  auto const theta = static_cast<T>(arguments.theta);
  auto const steps = std::max(arguments.steps, arguments.warmup_steps);   // src/all_pairs.h:93-97, src/bvh.h:395-401
  for (std::size_t step = 0; step < steps; step++) {
    switch (arguments.simulation_algo) {
      case SimulationAlgo::AllPairs: all_pairs_force(d); break;
      case SimulationAlgo::AllPairsCollapsed: all_pairs_collapsed_force(d); break;
      case SimulationAlgo::BVH: bvh_force(d, theta); break;
      case SimulationAlgo::Octree: octree_force(d, theta); break;
    }
    accelerate_step(d);
  }
  d.download(system);   // before run_simulation prints the final state (src/main.cpp:32-35)
}

template <typename T, dim_t N>
void run_precision_hip(Arguments arguments) {
  auto system = [&arguments] {
    switch (arguments.simulation_type) {
      case SimulationType::Plummer: return build_plummer_model<T, N>(arguments);
      case SimulationType::Galaxy: return build_galaxy_model<T, N>(arguments);
      case SimulationType::Load: return Saver<T, N>::load_system(arguments.load_input.value());
      default: return build_uniform_model<T, N>(arguments);
    }
  }();
  arguments.size = system.size;
  bool const resident = std::getenv("NBODY_HIP_RESIDENT") != nullptr;
  if (!resident && arguments.simulation_algo != SimulationAlgo::AllPairs) {
    std::cerr << "without NBODY_HIP_RESIDENT=1 only --algorithm all-pairs goes through the backend" << std::endl;
    std::exit(EXIT_FAILURE);
  }
  sim_func_t<T, N> algo = resident ? run_resident_hip<T, N> : run_all_pairs_hip<T, N>;
  run_simulation<T, N>(arguments, system, algo);
}

// every template of the stub is instantiated, used above or not: a drifted signature in nbody_hip.h fails this build
template struct hip_mirror<double, DIM_SIZE>;
template struct hip_mirror<float, DIM_SIZE>;
template struct hip_multi<double, DIM_SIZE>;
template auto calc_energies<double, DIM_SIZE>(hip_mirror<double, DIM_SIZE>&) -> std::tuple<double, double>;
template auto calc_energies<float, DIM_SIZE>(hip_mirror<float, DIM_SIZE>&) -> std::tuple<float, float>;

int main(int argc, char* argv[]) {
  auto arguments = parse_args(std::vector<std::string>(argv + 1, argv + argc));
  if (arguments.single_precision) run_precision_hip<float, DIM_SIZE>(arguments);
  else run_precision_hip<double, DIM_SIZE>(arguments);
  return EXIT_SUCCESS;
}
