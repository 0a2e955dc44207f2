// Workload generators: uniform cube, Plummer sphere (3D), two colliding disc galaxies — the same
// initial conditions, RNG (std::mt19937{42}) and draw order as src/models.h of the reference, since
// "synthetic galaxy init" is the benchmark input.  Host-serial; stays on the CPU.
#pragma once
#include <cmath>
#include <cstdint>
#include <fstream>
#include <limits>
#include <stdexcept>
#include <string>

#include "system.hpp"

namespace nb {

struct ModelRequest {
  std::size_t size;
};

// uniform: m = 1/N, position and velocity components alternate draws from [-1,1); dt = 0.1, G = 1
// (src/models.h:12-28)
template <typename T, int D>
System<T, D> make_uniform(std::size_t count) {
  System<T, D> sys(static_cast<std::uint32_t>(count), T(1e-1), T(1));
  for (std::size_t b = 0; b < count; ++b) {
    T const mass = 1.0 / static_cast<T>(count);
    vecn<T, D> p{}, w{};
    for (int k = 0; k < D; ++k) {
      p[k] = sys.sym_dis(sys.gen);
      w[k] = sys.sym_dis(sys.gen);
    }
    sys.add_body(mass, p, w);
  }
  return sys;
}

// Plummer sphere, 3D only; dt = 1, G = 6.674e-11 (src/models.h:30-71)
template <typename T, int D>
System<T, D> make_plummer(std::size_t count) {
  if constexpr (D != 3) {
    throw std::runtime_error("Cannot build Plummer model for D=" + std::to_string(D));
  } else {
    System<T, D> sys(static_cast<std::uint32_t>(count), T(1), static_cast<T>(6.674e-11));
    auto on_sphere = [](T r, T th, T ph) {
      return vecn<T, 3>{r * (std::sin(th) * std::cos(ph)), r * (std::sin(th) * std::sin(ph)), r * std::cos(th)};
    };
    for (std::size_t b = 0; b < count; ++b) {
      T const mass = 1.0 / static_cast<T>(count);
      T const r    = 1.0 / std::sqrt(std::pow(sys.unit_dis(sys.gen), -2.0 / 3.0) - 1);
      T const th   = std::acos(sys.sym_dis(sys.gen));
      T const ph   = sys.angle_dis(sys.gen);
      auto pos     = on_sphere(r, th, ph);
      T q = 0.0, g = 0.1;  // rejection sampling of the speed
      while (g > q * q * std::pow(1.0 - q * q, 3.5)) {
        q = sys.unit_dis(sys.gen);
        g = 0.1 * sys.unit_dis(sys.gen);
      }
      T const speed = q * 1.41421356237309504880168872420969808 * std::pow(r * r + 1, -0.25);
      T const vth   = std::acos(sys.sym_dis(sys.gen));
      T const vph   = sys.angle_dis(sys.gen);
      sys.add_body(mass, pos, on_sphere(speed, vth, vph));
    }
    return sys;
  }
}

namespace detail {
// One disc of `count` light bodies on circular orbits around `centre` (src/models.h:81-110).
template <typename T, int D>
void add_disc(System<T, D>& sys, std::size_t count, T central_mass, T disc_mass, vecn<T, D> const& centre) {
  constexpr T eps = std::numeric_limits<T>::epsilon();
  for (std::size_t b = 0; b < count; ++b) {
    T const mass   = disc_mass / static_cast<T>(count);
    T const radius = 30 + 20 * sys.unit_dis(sys.gen);
    T const angle  = sys.angle_dis(sys.gen);
    vecn<T, D> pos{}, vel{};
    pos[0] = radius * std::sin(angle);
    pos[1] = radius * std::cos(angle);
    T const speed = std::sqrt(sys.G * central_mass / (radius + eps));
    T norm2       = T(0.);
    for (int k = 0; k < D; ++k) norm2 += pos[k] * pos[k];
    T const scale = speed / (std::sqrt(norm2) + eps);
    vel[0]        = scale * (-pos[1]);
    vel[1]        = scale * pos[0];
    if constexpr (D == 3) {
      pos[2] = 10 * sys.sym_dis(sys.gen);
      vel[2] = 0.00001 * sys.sym_dis(sys.gen);
      // tilt the disc (the matrix is not orthogonal; it is what the reference uses)
      T const tilt[3][3] = {{T(0.0), T(-1.0), T(0.0)}, {T(0.9), T(0.0), T(0.5)}, {T(0.5), T(0.0), T(0.9)}};
      auto apply = [&](vecn<T, 3> const& u) {
        vecn<T, 3> r{};
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) r[i] += tilt[i][j] * u[j];
        return r;
      };
      pos = apply(pos);
      vel = apply(vel);
    }
    for (int k = 0; k < D; ++k) pos[k] = pos[k] + centre[k];
    sys.add_body(mass, pos, vel);
  }
}
}  // namespace detail

// Two discs, central masses 1e4 and 1e3 at -/+(100, -50, 0); dt = 10, G = 1e-4 (src/models.h:112-136).
// The system holds 2*(size/2.0) bodies (truncated), each disc has size/2 - 1 orbiters.
template <typename T, int D>
System<T, D> make_galaxy(std::size_t count) {
  // (the reference writes its second central mass past the end of a one-body system; refused here)
  if (count < 2) throw std::invalid_argument("the galaxy workload needs at least 2 bodies (one central mass per disc)");
  double const half = count / 2.0;
  System<T, D> sys(static_cast<std::uint32_t>(2 * half), T(1e1), T(1e-4));
  T central      = 1e4;
  T const offset = 100.0;
  vecn<T, D> c{};
  c[0] = offset * T(-1);
  c[1] = offset * T(1 / 2.0);
  sys.add_body(central, c, vecn<T, D>{});
  detail::add_disc<T, D>(sys, static_cast<std::size_t>(half - 1), central + 1, T(1), c);
  central /= 10;
  c[0] = offset * T(1);
  c[1] = offset * T(-1 / 2.0);
  sys.add_body(central, c, vecn<T, D>{});
  detail::add_disc<T, D>(sys, static_cast<std::size_t>(half - 1), central + 1, T(1), c);
  return sys;
}

// `--workload load f.bin`: u32 n, u32 dim, f32 dt, f32 G, then n x (m, pos[D], vel[D]) as f32
// (src/saving.h:25-68; writer scripts/thuering_nbody/conv_csv.py:62-80).
template <typename T, int D>
System<T, D> load_bin(std::string const& path) {
  std::ifstream in(path, std::ios::binary);
  std::uint32_t n = 0, dim = 0;
  float dt = 0, G = 0;
  in.read(reinterpret_cast<char*>(&n), 4);
  in.read(reinterpret_cast<char*>(&dim), 4);
  in.read(reinterpret_cast<char*>(&dt), 4);
  in.read(reinterpret_cast<char*>(&G), 4);
  if (dim != D)
    throw std::runtime_error("This version is compiled with D=" + std::to_string(D) + ", but the file provided is D=" +
                             std::to_string(dim));
  std::size_t const stride = 1 + 2 * std::size_t(dim);
  std::vector<float> raw(std::size_t(n) * stride);
  in.read(reinterpret_cast<char*>(raw.data()), std::streamsize(raw.size() * sizeof(float)));
  System<T, D> sys(n, dt, G);
  for (std::size_t b = 0; b < n; ++b) {
    vecn<T, D> p{}, w{};
    for (int k = 0; k < D; ++k) {
      p[k] = raw[b * stride + 1 + k];
      w[k] = raw[b * stride + 1 + D + k];
    }
    sys.add_body(raw[b * stride], p, w);
  }
  return sys;
}

}  // namespace nb
