// positions.bin / energy.bin writers in the reference's on-disk format (src/saving.h:85-122):
//   positions.bin: u32 nbodies, u32 steps, u32 sizeof(T), u32 dim, then one frame of x per save_all
//   energy.bin:    u32 steps, u32 sizeof(T), then (kinetic, potential) per save_all
#pragma once
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "backend.hpp"
#include "options.hpp"

namespace nb {

template <typename T, int D>
class Saver {
 public:
  explicit Saver(Options const& o)
   : pos_(o.save_pos), energy_(o.save_energy), nbodies_(std::uint32_t(o.size)), nsteps_(std::uint32_t(o.steps)) {
    std::uint32_t const tsz = sizeof(T), dim = D;
    if (pos_) {
      pos_file_.open("positions.bin", std::ios::out | std::ios::binary);
      put(pos_file_, nbodies_);
      put(pos_file_, nsteps_);
      put(pos_file_, tsz);
      put(pos_file_, dim);
    }
    if (energy_) {
      energy_file_.open("energy.bin", std::ios::out | std::ios::binary);
      put(energy_file_, nsteps_);
      put(energy_file_, tsz);
    }
  }

  bool active() const { return pos_ || energy_; }

  // one frame; the device copy is authoritative, so positions are pulled first
  void save_all(System<T, D>& sys, Device<T, D>& dev) {
    if (pos_) {
      dev.pull_positions();
      pos_file_.write(reinterpret_cast<char const*>(sys.x.data()), std::streamsize(std::size_t(nbodies_) * sizeof(T) * D));
    }
    if (energy_) {
      auto [kinetic, potential] = dev.calc_energies();
      put(energy_file_, kinetic);
      put(energy_file_, potential);
    }
  }

 private:
  template <typename U>
  static void put(std::ofstream& f, U const& v) {
    f.write(reinterpret_cast<char const*>(&v), sizeof v);
  }
  bool pos_, energy_;
  std::uint32_t nbodies_, nsteps_;
  std::ofstream pos_file_, energy_file_;
};

}  // namespace nb
