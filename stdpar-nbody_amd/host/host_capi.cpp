// C entry to the host workload generators so the Python bench/test harness builds its inputs with the
// PRODUCT's generators (host/models.hpp), not with the test oracle.
#include <cstdint>
#include <cstring>

#include "models.hpp"

namespace {
template <typename T, int D>
std::int64_t fill(int workload, std::uint32_t n, void* m, void* x, void* v, double* dt, double* G) {
  auto emit = [&](nb::System<T, D> const& s) -> std::int64_t {
    std::memcpy(m, s.m.data(), sizeof(T) * s.n);
    std::memcpy(x, s.x.data(), sizeof(T) * D * s.n);
    std::memcpy(v, s.v.data(), sizeof(T) * D * s.n);
    *dt = s.dt;
    *G  = s.G;
    return s.n;
  };
  try {
    if (workload == 0) return emit(nb::make_uniform<T, D>(n));
    if (workload == 1) return emit(nb::make_plummer<T, D>(n));
    if (workload == 2) return emit(nb::make_galaxy<T, D>(n));
  } catch (...) {
    return -3;
  }
  return -1;
}
}  // namespace

// dtype 0 = f32, 1 = f64; workload 0 uniform, 1 plummer, 2 galaxy; arrays sized for n bodies.
// Returns the system size (galaxy may be n-1 for odd n) or < 0 on error.
extern "C" std::int64_t nbody_host_build_model(int dtype, int dim, int workload, std::uint32_t n, void* m, void* x, void* v,
                                               double* dt, double* G) {
  if (dtype == 0 && dim == 2) return fill<float, 2>(workload, n, m, x, v, dt, G);
  if (dtype == 0 && dim == 3) return fill<float, 3>(workload, n, m, x, v, dt, G);
  if (dtype == 1 && dim == 2) return fill<double, 2>(workload, n, m, x, v, dt, G);
  if (dtype == 1 && dim == 3) return fill<double, 3>(workload, n, m, x, v, dt, G);
  return -2;
}
