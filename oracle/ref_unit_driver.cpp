// TEST INFRASTRUCTURE. Tiny driver around the reference's own vec.h (included from where it lies under
// /root/reference/src — nothing is copied): prints hilbert<2>, hilbert<3> and interleave_bits for the
// cells given on stdin so tests/golden/generate_golden.py can commit them as known-answer vectors.
// Build (oracle/Makefile target `ref`): g++ -std=c++20 -O2 -I/root/reference/src ref_unit_driver.cpp
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <type_traits>
#include "vec.h"

int main() {
  unsigned dim;
  unsigned long long c0, c1, c2;
  while (std::scanf("%u %llu %llu %llu", &dim, &c0, &c1, &c2) == 4) {
    if (dim == 2) {
      vec<uint32_t, 2> c{{uint32_t(c0), uint32_t(c1)}};
      std::printf("2 %llu %llu 0 %llu %llu\n", c0, c1, (unsigned long long)hilbert<2>(c), (unsigned long long)interleave_bits<2>(c));
    } else {
      vec<uint32_t, 3> c{{uint32_t(c0), uint32_t(c1), uint32_t(c2)}};
      std::printf("3 %llu %llu %llu %llu %llu\n", c0, c1, c2, (unsigned long long)hilbert<3>(c), (unsigned long long)interleave_bits<3>(c));
    }
  }
  return 0;
}
