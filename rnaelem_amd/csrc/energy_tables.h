// energy_tables.h -- Turner-style nearest-neighbour parameters as flat log-Boltzmann tables.
//
// Host side of row (a3) of SURVEY.md §8: reads ViennaRNA-2.0 parameter text and produces the same
// numbers the reference keeps in EnergyParam (RNAelem/energy_param.hpp:61-84): every entry is
// log weight = -dG[dcal/mol]*10/kT, kT = (37+273.15)*1.98717 (energy_param.hpp:20-27, 108-114).
// The struct is a POD that is copied to the GPU verbatim (kernels index it by the same formulas).
#pragma once
#include <cstdint>
#include <string>

namespace elemdp {

constexpr int kMaxLoop = 30;
constexpr int kNumSpecial = 40;  // capacity of the tri/tetra/hexa-loop lists (energy_param.hpp:67-69)

struct EnergyTables {
  double stack[7 * 7];
  double hairpin[kMaxLoop + 1];
  double bulge[kMaxLoop + 1];
  double interior[kMaxLoop + 1];
  double ninio[kMaxLoop + 1];
  double mismatch_h[7 * 25];
  double mismatch_i[7 * 25];
  double mismatch_m[7 * 25];
  double mismatch_1ni[7 * 25];
  double mismatch_23i[7 * 25];
  double mismatch_ext[7 * 25];
  double dangle5[8 * 5];
  double dangle3[8 * 5];
  double int11[8 * 8 * 25];
  double int21[8 * 8 * 125];
  double int22[8 * 8 * 625];
  double triloop[kNumSpecial];
  double tetraloop[kNumSpecial];
  double hexaloop[kNumSpecial];
  // special loops as packed base codes (2 bits per base, A=0..U=3; first base in the top bits)
  uint32_t tri_key[kNumSpecial];
  uint32_t tetra_key[kNumSpecial];
  uint32_t hexa_key[kNumSpecial];
  int32_t n_tri, n_tetra, n_hexa;
  int32_t pad_;
  double term_au;
  double ml_intern;
  double ml_closing;
  double ml_base;
  double lxc37;
};

// x = the tables of e exponentiated entry by entry (Boltzmann weights instead of their logarithms; exp(log 0) = 0); the
// special-loop keys and lxc37 are copied.  For loop_weight (energy_rules.h).
void exp_tables(const EnergyTables& e, EnergyTables* x);

// no entry loop_energy can reach for canonical pair types (1 .. 6) in a sequence WITHOUT N is log 0 (the 2x2 table has no N
// entries): a candidate loop of such a sequence is then dropped by its size alone (count_interior_by_end, plan_rules.h)
bool loop_tables_finite(const EnergyTables& e);

// Parses ViennaRNA-2.0 format text (the shipped *.elempar files are a comment-free subset of it).
// Throws std::runtime_error on malformed input.
void parse_energy_text(const std::string& text, EnergyTables* out);

// -dG*10/kT and its "smoothed" variant used for dangles / multi / exterior mismatches
// (energy_param.hpp:94-114).
double log_boltzmann(int dcal, bool smooth);

}  // namespace elemdp
