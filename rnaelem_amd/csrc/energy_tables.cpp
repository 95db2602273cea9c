// energy_tables.cpp -- ViennaRNA-2.0 parameter text -> flat log-Boltzmann tables (host only).
//
// Behavioural spec = the reference's parser, RNAelem/energy_param.hpp:116-640: which section
// names are recognised (:116-157), how a block of numbers is consumed line by line (:159-183:
// whole lines, a word containing "/*" ends the line, surplus words on the last line are dropped,
// INF -> log 0, DEF -> -50 dcal/mol), and which index ranges each section fills (:519-640).
// Two deliberate deviations, both outside what a real base pair can index and documented in
// (1) the 7th ("NS") block of mismatch_multi / mismatch_exterior, which the reference
// writes past the end of its [7][5][5] arrays, is read and discarded; (2) int22 entries that
// involve an 'N' base are log 0 here (the reference leaves them uninitialised).
#include "energy_tables.h"

#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>
#include <vector>

namespace elemdp {
namespace {

const double kNegInf = -std::numeric_limits<double>::infinity();
const double kKT = (37 + 273.15) * 1.98717;

std::vector<std::string> split_words(const std::string& line) {
  std::vector<std::string> w;
  std::istringstream iss(line);
  for (std::string t; iss >> t;) w.push_back(t);
  return w;
}

// One "# name" section: the data lines that follow the header up to the first blank line.
class Section {
 public:
  void add_line(const std::string& l) { lines_.push_back(l); }
  bool present = false;
  // Fills dst[0..n) the way one get_array(dst, n, smooth) call does; returns #values stored.
  int take(double* dst, int n, bool smooth) {
    int i = 0;
    while (i < n && cursor_ < lines_.size()) {
      std::vector<std::string> w = split_words(lines_[cursor_++]);
      for (size_t k = 0; k < w.size() && i < n; ++k) {
        if (w[k].find("/*") != std::string::npos) break;
        if (w[k] == "INF") dst[i++] = kNegInf;
        else if (w[k] == "DEF") dst[i++] = log_boltzmann(-50, smooth);
        else dst[i++] = log_boltzmann(std::atoi(w[k].c_str()), smooth);
      }
    }
    return i;
  }
  // first data line that is not a comment, split into words
  std::vector<std::string> first_record() const {
    for (auto const& l : lines_) if (l.find('*') == std::string::npos) return split_words(l);
    return {};
  }
  std::vector<std::vector<std::string>> records() const {
    std::vector<std::vector<std::string>> r;
    for (auto const& l : lines_) if (l.find('*') == std::string::npos) r.push_back(split_words(l));
    return r;
  }

 private:
  std::vector<std::string> lines_;
  size_t cursor_ = 0;
};

void fill_neg_inf(double* a, int n) { for (int i = 0; i < n; ++i) a[i] = kNegInf; }

int base_code2(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'U': case 'u': case 'T': case 't': return 3;
  }
  return -1;
}

void load_special(Section& sec, int len, double* energy, uint32_t* key, int32_t* count) {
  fill_neg_inf(energy, kNumSpecial);
  *count = 0;
  for (auto const& w : sec.records()) {
    if (w.size() < 2) throw std::runtime_error("energy parameters: malformed special-loop line");
    if (*count >= kNumSpecial) throw std::runtime_error("energy parameters: more than 40 special loops");
    if ((int)w[0].size() != len) throw std::runtime_error("energy parameters: special loop of wrong length: " + w[0]);
    uint32_t k = 0;
    for (char c : w[0]) {
      int b = base_code2(c);
      if (b < 0) throw std::runtime_error("energy parameters: bad base in special loop: " + w[0]);
      k = (k << 2) | (uint32_t)b;
    }
    key[*count] = k;
    energy[*count] = log_boltzmann(std::atoi(w[1].c_str()), false);
    ++*count;
  }
}

}  // namespace

void exp_tables(const EnergyTables& e, EnergyTables* x) {
  *x = e;
  auto ex = [](const double* src, double* dst, size_t n) { for (size_t k = 0; k < n; ++k) dst[k] = std::exp(src[k]); };
#define ELEMDP_EXP_TABLE(f) ex(e.f, x->f, sizeof(e.f) / sizeof(double))
  ELEMDP_EXP_TABLE(stack); ELEMDP_EXP_TABLE(hairpin); ELEMDP_EXP_TABLE(bulge); ELEMDP_EXP_TABLE(interior); ELEMDP_EXP_TABLE(ninio);
  ELEMDP_EXP_TABLE(mismatch_h); ELEMDP_EXP_TABLE(mismatch_i); ELEMDP_EXP_TABLE(mismatch_m); ELEMDP_EXP_TABLE(mismatch_1ni);
  ELEMDP_EXP_TABLE(mismatch_23i); ELEMDP_EXP_TABLE(mismatch_ext); ELEMDP_EXP_TABLE(dangle5); ELEMDP_EXP_TABLE(dangle3);
  ELEMDP_EXP_TABLE(int11); ELEMDP_EXP_TABLE(int21); ELEMDP_EXP_TABLE(int22);
  ELEMDP_EXP_TABLE(triloop); ELEMDP_EXP_TABLE(tetraloop); ELEMDP_EXP_TABLE(hexaloop);
#undef ELEMDP_EXP_TABLE
  x->term_au = std::exp(e.term_au);
  x->ml_intern = std::exp(e.ml_intern);
  x->ml_closing = std::exp(e.ml_closing);
  x->ml_base = std::exp(e.ml_base);
}

bool loop_tables_finite(const EnergyTables& e) {
  auto fin = [](double v) { return std::isfinite(v); };
  if (!fin(e.term_au)) return false;
  for (int u = 1; u <= kMaxLoop; ++u) if (!fin(e.bulge[u])) return false;
  for (int u = 4; u <= kMaxLoop; ++u) if (!fin(e.interior[u])) return false;
  for (int u = 0; u <= kMaxLoop; ++u) if (!fin(e.ninio[u])) return false;
  for (int t = 1; t <= 6; ++t) {
    for (int b = 0; b < 25; ++b)
      if (!fin(e.mismatch_i[t * 25 + b]) || !fin(e.mismatch_1ni[t * 25 + b]) || !fin(e.mismatch_23i[t * 25 + b])) return false;
    for (int t2 = 1; t2 <= 6; ++t2) {
      if (!fin(e.stack[t * 7 + t2])) return false;
      for (int b = 0; b < 25; ++b) if (!fin(e.int11[(t * 8 + t2) * 25 + b])) return false;
      for (int b = 0; b < 125; ++b) if (!fin(e.int21[(t * 8 + t2) * 125 + b])) return false;
      for (int b = 0; b < 625; ++b) {       // (the 2x2 table has no entries for N: bases A .. U only)
        if (b % 5 == 0 || (b / 5) % 5 == 0 || (b / 25) % 5 == 0 || b / 125 == 0) continue;
        if (!fin(e.int22[(t * 8 + t2) * 625 + b])) return false;
      }
    }
  }
  return true;
}

double log_boltzmann(int dcal, bool smooth) {
  if (!smooth) return -dcal * 10. / kKT;
  // smooth(-z) (energy_param.hpp:94-106): a C1 clamp of stabilising energies at 0
  double z = double(-dcal);
  double s;
  if (z / 10. < -1.2283697) s = 0.;
  else if (0.8660254 < z / 10.) s = z;
  else s = 10. * 0.38490018 * (1. + std::sin(z / 10. - 0.34242663)) * (1. + std::sin(z / 10. - 0.34242663));
  return s * 10. / kKT;
}

void parse_energy_text(const std::string& text, EnergyTables* t) {
  // ---- split into sections
  std::map<std::string, Section> sec;
  {
    std::istringstream iss(text);
    Section* cur = nullptr;
    bool open = false;
    for (std::string line; std::getline(iss, line);) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (!line.empty() && line[0] == '#') {
        std::vector<std::string> w = split_words(line);
        cur = nullptr;
        open = false;
        if (w.size() >= 2 && w[0] == "#") {
          Section& s = sec[w[1]];
          if (!s.present) { s.present = true; cur = &s; open = true; }  // first occurrence wins
        }
        continue;
      }
      if (!open || !cur) continue;
      if (line.size() < 2) { open = false; continue; }  // a blank line ends the block
      cur->add_line(line);
    }
  }
  auto has = [&](const char* n) { auto it = sec.find(n); return it != sec.end() && it->second.present; };

  // ---- defaults: everything log 0 until a section says otherwise
  fill_neg_inf(t->stack, 49);
  fill_neg_inf(t->hairpin, 31); fill_neg_inf(t->bulge, 31); fill_neg_inf(t->interior, 31); fill_neg_inf(t->ninio, 31);
  fill_neg_inf(t->mismatch_h, 175); fill_neg_inf(t->mismatch_i, 175); fill_neg_inf(t->mismatch_m, 175);
  fill_neg_inf(t->mismatch_1ni, 175); fill_neg_inf(t->mismatch_23i, 175); fill_neg_inf(t->mismatch_ext, 175);
  fill_neg_inf(t->dangle5, 40); fill_neg_inf(t->dangle3, 40);
  fill_neg_inf(t->int11, 1600); fill_neg_inf(t->int21, 8000); fill_neg_inf(t->int22, 40000);
  fill_neg_inf(t->triloop, kNumSpecial); fill_neg_inf(t->tetraloop, kNumSpecial); fill_neg_inf(t->hexaloop, kNumSpecial);
  t->n_tri = t->n_tetra = t->n_hexa = 0;
  t->pad_ = 0;
  t->term_au = t->ml_intern = t->ml_closing = t->ml_base = 0.;
  t->lxc37 = 107.856;

  // ---- pair-type x pair-type / mismatch blocks
  if (has("stack")) {  // rows CG..UA, first six columns of each row (the NN column is skipped)
    Section& s = sec["stack"];
    for (int a = 1; a <= 6; ++a) s.take(&t->stack[a * 7 + 1], 6, false);
  }
  struct MM { const char* name; double* dst; int blocks; bool smooth; };
  MM mms[] = {{"mismatch_hairpin", t->mismatch_h, 6, false},     {"mismatch_interior", t->mismatch_i, 6, false},
              {"mismatch_interior_1n", t->mismatch_1ni, 6, false}, {"mismatch_interior_23", t->mismatch_23i, 6, false},
              {"mismatch_multi", t->mismatch_m, 7, true},          {"mismatch_exterior", t->mismatch_ext, 7, true}};
  for (auto& m : mms) {
    if (!has(m.name)) continue;
    Section& s = sec[m.name];
    double discard[25];
    for (int a = 1; a <= m.blocks; ++a) s.take(a <= 6 ? &m.dst[a * 25] : discard, 25, m.smooth);
  }
  if (has("dangle5")) for (int a = 1; a <= 7; ++a) sec["dangle5"].take(&t->dangle5[a * 5], 5, true);
  if (has("dangle3")) for (int a = 1; a <= 7; ++a) sec["dangle3"].take(&t->dangle3[a * 5], 5, true);
  if (has("int11"))
    for (int a = 1; a <= 7; ++a) for (int b = 1; b <= 7; ++b) sec["int11"].take(&t->int11[(a * 8 + b) * 25], 25, false);
  if (has("int21"))
    for (int a = 1; a <= 7; ++a) for (int b = 1; b <= 7; ++b) sec["int21"].take(&t->int21[(a * 8 + b) * 125], 125, false);
  if (has("int22"))  // six real pair types only, unpaired bases A..U only
    for (int a = 1; a <= 6; ++a) for (int b = 1; b <= 6; ++b)
      for (int c = 1; c <= 4; ++c) for (int d = 1; d <= 4; ++d) for (int e = 1; e <= 4; ++e)
        sec["int22"].take(&t->int22[((((a * 8 + b) * 5 + c) * 5 + d) * 5 + e) * 5 + 1], 4, false);
  if (has("hairpin")) sec["hairpin"].take(t->hairpin, 31, false);
  if (has("bulge")) sec["bulge"].take(t->bulge, 31, false);
  if (has("interior")) sec["interior"].take(t->interior, 31, false);

  // ---- scalar records
  if (has("NINIO")) {  // "m  m_dH  max": ninio[u] = min(max, u*m)  (energy_param.hpp:399-420)
    auto w = sec["NINIO"].first_record();
    if (w.size() <= 2) throw std::runtime_error("energy parameters: malformed NINIO record");
    int m = std::atoi(w[0].c_str()), mx = std::atoi(w[2].c_str());
    for (int u = 0; u <= kMaxLoop; ++u) t->ninio[u] = log_boltzmann(mx < u * m ? mx : u * m, false);
  }
  if (has("ML_params")) {  // "cu cu_dH cc cc_dH ci ci_dH"  (:422-442)
    auto w = sec["ML_params"].first_record();
    if (w.size() <= 4) throw std::runtime_error("energy parameters: malformed ML_params record");
    t->ml_base = log_boltzmann(std::atoi(w[0].c_str()), false);
    t->ml_closing = log_boltzmann(std::atoi(w[2].c_str()), false);
    t->ml_intern = log_boltzmann(std::atoi(w[4].c_str()), false);
  }
  if (has("Misc")) {  // "DuplexInit dH TerminalAU dH [LXC ...]"  (:444-456, 478-492)
    for (auto const& w : sec["Misc"].records()) {
      if (w.size() <= 2) throw std::runtime_error("energy parameters: malformed Misc record");
      t->term_au = log_boltzmann(std::atoi(w[2].c_str()), false);
      if (w.size() > 4) t->lxc37 = std::atof(w[4].c_str());
    }
  }
  if (has("Triloops")) load_special(sec["Triloops"], 5, t->triloop, t->tri_key, &t->n_tri);
  if (has("Tetraloops")) load_special(sec["Tetraloops"], 6, t->tetraloop, t->tetra_key, &t->n_tetra);
  if (has("Hexaloops")) load_special(sec["Hexaloops"], 8, t->hexaloop, t->hexa_key, &t->n_hexa);
}

}  // namespace elemdp
