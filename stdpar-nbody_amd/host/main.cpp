// nbody_hip_d2 / nbody_hip_d3 — CLI with the reference's flags (src/main.cpp:42-74); the spatial
// dimension is a compile-time constant exactly as in the reference (-DDIM_SIZE=2|3).
#ifndef DIM_SIZE
  #error Must specify spatial dimensions by compiling with -DDIM_SIZE=2 or -DDIM_SIZE=3 .
#endif

#include <string>
#include <vector>

#include "drivers.hpp"
#include "models.hpp"
#include "options.hpp"

template <typename T, int D>
static void run_precision(nb::Options o) {
  auto sys = [&o]() -> nb::System<T, D> {
    switch (o.workload) {
      case nb::Workload::Plummer: return nb::make_plummer<T, D>(o.size);
      case nb::Workload::Uniform: return nb::make_uniform<T, D>(o.size);
      case nb::Workload::Galaxy: return nb::make_galaxy<T, D>(o.size);
      case nb::Workload::Load: {
        auto s = nb::load_bin<T, D>(o.load_input.value());
        o.size = s.n;
        return s;
      }
    }
    throw std::runtime_error("Unknown simulation type");
  }();
  nb::run_simulation<T, D>(o, sys);
}

int main(int argc, char* argv[]) {
  auto o = nb::parse_options(std::vector<std::string>(argv + 1, argv + argc));
  if (o.single_precision) run_precision<float, DIM_SIZE>(o);
  else run_precision<double, DIM_SIZE>(o);
  return EXIT_SUCCESS;
}
