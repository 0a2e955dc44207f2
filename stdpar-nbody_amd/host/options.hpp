// Command line of the reference binary (src/arguments.h): same flags, defaults, help and error text.
#pragma once
#include <cstdlib>
#include <iostream>
#include <optional>
#include <string>
#include <vector>

namespace nb {

enum class Workload { Uniform, Plummer, Galaxy, Load };
enum class Algorithm { AllPairs, AllPairsCollapsed, Octree, Bvh };

struct Options {
  std::size_t size         = 1000;
  std::size_t steps        = 1;
  std::size_t warmup_steps = 10;  // no flag sets it (src/arguments.h:26)
  bool single_precision    = true;
  Workload workload        = Workload::Uniform;
  Algorithm algorithm      = Algorithm::Octree;
  bool print_state         = false;
  bool print_info          = false;
  double theta             = 0.5;
  bool save_pos            = false;
  bool save_energy         = false;
  bool csv_detailed        = false;
  bool csv_total           = false;
  std::optional<std::string> load_input;
  // not in the reference (and therefore not in the --help text, which stays the reference's word for word): --gpus N shards
  // the bodies of --algorithm all-pairs over N devices of this node, one RCCL all-gather of positions per step
  int gpus = 1;
  bool gpus_given = false;  // an explicit --gpus N (any N, also 1) takes the sharded path: communicator, shard windows, exchange
};

namespace detail {
inline constexpr char const* kHelp =
 "Help:\n"
 "-n size\t\tNumber of particles to simulate\n"
 "-s steps\t\tNumber of steps to run simulation for\n"
 "--theta t\t\tTheta threshold parameter to use in Octree\n"
 "--precision double|float(default)\t\tSelects floating-point precision\n"
 "--algorithm all-pairs|all-pairs-collapsed|bvh|octree(default)<algo>\t\tSelects simulation algorithm\n"
 "--workload plummer|galaxy|uniform(default)|load <file.bin>\t\tSelects workload\n"
 "--print-state\t\tPrint the initial and final state of the simulation\n"
 "--print-info\t\tPrint info every timestep\n"
 "--save pos|energy|all|none(default) \t\tSelects what data to save every timestep\n"
 "--help\t\tDisplay this help message and quit\n";

[[noreturn]] inline void reject(char const* what, std::string const& got, char const* choices) {
  std::cerr << "Unknown " << what << ": \"" << got << "\"." << std::endl;
  std::cerr << "Options are: " << choices << "." << std::endl;
  std::exit(EXIT_FAILURE);
}
}  // namespace detail

inline Options parse_options(std::vector<std::string> const& argv) {
  Options o;
  for (std::size_t k = 0; k < argv.size(); ++k) {
    std::string const& f = argv[k];
    auto value           = [&]() -> std::string const& { return argv.at(++k); };  // missing value -> std::out_of_range
    if (f == "-n") {
      o.size = std::stoi(value());
    } else if (f == "-s") {
      o.steps = std::stoi(value());
    } else if (f == "--theta") {
      o.theta = std::stod(value());
    } else if (f == "--csv-detailed") {
      o.csv_detailed = true;
    } else if (f == "--csv-total") {
      o.csv_total = true;
    } else if (f == "--precision") {
      auto const& p = value();
      if (p == "float") o.single_precision = true;
      else if (p == "double") o.single_precision = false;
      else detail::reject("precision", p, "double, float (default)");
    } else if (f == "--algorithm") {
      auto const& a = value();
      if (a == "all-pairs") o.algorithm = Algorithm::AllPairs;
      else if (a == "all-pairs-collapsed") o.algorithm = Algorithm::AllPairsCollapsed;
      else if (a == "octree") o.algorithm = Algorithm::Octree;
      else if (a == "bvh") o.algorithm = Algorithm::Bvh;
      else detail::reject("algorithm", a, "all-pairs, all-pairs-collapsed, octree (default)");
    } else if (f == "--workload") {
      auto const& w = value();
      if (w == "plummer") o.workload = Workload::Plummer;
      else if (w == "galaxy") o.workload = Workload::Galaxy;
      else if (w == "uniform") o.workload = Workload::Uniform;
      else if (w == "load") {
        o.load_input = value();
        o.workload   = Workload::Load;
      } else detail::reject("workload", w, "plummer, galaxy, uniform (default)");
    } else if (f == "--gpus") {
      o.gpus = std::stoi(value());
      o.gpus_given = true;
      if (o.gpus < 1) {
        std::cerr << "--gpus needs a positive device count." << std::endl;
        std::exit(EXIT_FAILURE);
      }
    } else if (f == "--print-state") {
      o.print_state = true;
    } else if (f == "--print-info") {
      o.print_info = true;
    } else if (f == "--save") {
      auto const& s = value();
      if (s == "pos") o.save_pos = true;
      else if (s == "energy") o.save_energy = true;
      else if (s == "all") o.save_pos = o.save_energy = true;
      else if (s == "none") o.save_pos = o.save_energy = false;
      else detail::reject("save options", s, "pos, energy, all, none (default)");
    } else if (f == "--help" || f == "-h") {
      std::cout << detail::kHelp;
      std::exit(EXIT_SUCCESS);
    } else {
      std::cout << "Unknown argument: '" << f << "'\n";
      std::exit(EXIT_FAILURE);
    }
  }
  if (o.csv_detailed && o.csv_total) {
    std::cerr << "Cannot capture a CSV detailed and coarse trace in the same run. Specify one or the other." << std::endl;
    std::exit(EXIT_FAILURE);
  }
  return o;
}

}  // namespace nb
