// Host-side System<T,D>: owns the body arrays in the reference's layout (m[N]; x,v,a,ao[N][D],
// src/system.h:13-19) so they can be handed to the backend unchanged.  ISO C++20, no HIP headers.
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <random>
#include <vector>

namespace nb {

template <typename T, int D>
using vecn = std::array<T, D>;

template <typename T, int D>
struct System {
  using index_t = std::uint32_t;  // src/system.h:13
  index_t n;
  T dt;
  T G;  // "constant" in the reference
  std::vector<T> m;
  std::vector<vecn<T, D>> x, v, a, ao;  // a, ao start at 0 (value-initialised, src/system.h:35-36)

  // RNG state lives with the system so generators draw in the reference's order (src/system.h:22-25)
  std::mt19937 gen{42};
  std::uniform_real_distribution<> angle_dis{0, 2 * 3.141592653589793238462643383279502884};
  std::uniform_real_distribution<> unit_dis{0, 1};
  std::uniform_real_distribution<> sym_dis{-1, 1};

  System(index_t n_, T dt_, T G_) : n(n_), dt(dt_), G(G_), m(n_), x(n_), v(n_), a(n_), ao(n_) {}

  void add_body(T mass, vecn<T, D> const& pos, vecn<T, D> const& vel) {
    m[next_] = mass;
    x[next_] = pos;
    v[next_] = vel;
    ++next_;
  }

  // One row per body, components 0 and 1 only, 4 significant digits (src/system.h:90-97):
  //   "{:02}: m={: .3e}, p=({: .3e}, {: .3e}), v=(...), f=(...)"
  void print(std::ostream& os = std::cout) const {
    char row[256];
    for (std::size_t i = 0; i < n; ++i) {
      std::snprintf(row, sizeof row, "%02zu: m=% .3e, p=(% .3e, % .3e), v=(% .3e, % .3e), f=(% .3e, % .3e)", i, double(m[i]),
                    double(x[i][0]), double(x[i][1]), double(v[i][0]), double(v[i][1]), double(a[i][0]), double(a[i][1]));
      os << row << std::endl;
    }
  }

 private:
  std::size_t next_ = 0;
};

}  // namespace nb
