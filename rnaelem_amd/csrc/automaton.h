// automaton.h -- search pattern -> interval-state automaton -> flat tables for the kernels (host).
//
// Behavioural spec: ProfileHMM::build and the helpers it calls, RNAelem/profile_hmm.hpp:188-463
// (SURVEY.md §3.3).  Nodes are 'z' + regularised pattern + 'o'; an interval state (l,r) exists for
// every reachable node pair; the DP iterates four transition families (right / left loop
// emissions, pair emissions, bifurcation splits) and the interior-loop quadruples.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "device_layout.h"

namespace elemdp {

struct IntervalState { int id, l, r; };

class Automaton {
 public:
  // Throws std::runtime_error for malformed patterns (empty, bad character, unmatched bracket).
  explicit Automaton(const std::string& pattern);

  int M() const { return (int)node_.size(); }
  int S() const { return (int)states_.size(); }
  int n_rows() const { return (int)row_width_.size(); }
  int n_theta() const { return row_off_.back(); }
  const std::string& pattern() const { return pattern_; }
  const std::string& reg_pattern() const { return reg_; }
  char node(int h) const { return node_[h]; }
  int theta_row(int h) const { return theta_row_[h]; }  // -1 for '('
  int row_width(int r) const { return row_width_[r]; }
  int row_offset(int r) const { return row_off_[r]; }
  const IntervalState& state(int s) const { return states_[s]; }
  int state_id(int l, int r) const { return n2s_[l * M() + r]; }
  bool reachable(int a, int b) const { return reach_[a * M() + b]; }
  bool is_loop_state(int s) const { return loop_flag_[s]; }
  const std::vector<int>& right(int s) const { return right_[s]; }
  const std::vector<int>& left(int s) const { return left_[s]; }
  const std::vector<int>& pair(int s) const { return pair_[s]; }
  const std::vector<std::array<int, 4>>& quads() const { return quads_; }
  // h-splits of s: (s1=(l,h), s2=(h,r)) in ascending h
  std::vector<std::array<int, 2>> splits(int s) const;

  // JSON with the same keys as the oracle / reference dumps (tests compare them)
  std::string to_json() const;

  // Flattened tables for the kernels
  // only_state0: keep only transitions among state 0 = (0,0), the background state 'z'.  Outside values of an
  // evaluation whose only terminal is O(L,(0,0)) vanish on every other state (no transition leads from (0,0) to
  // another state), so that pass can be swept on this one-state automaton over the same tables.
  // prune: drop every list entry that cannot carry weight in ANY sequence (see liveness()): the lists of the inside direction
  // keep a transition only when its parent is useful and all its children are inside-live, the lists of the outside direction are
  // the same entries regrouped by child.  Table entries of useless (plane, state) pairs then stay 0 / log 0; partition functions,
  // posteriors, expected counts and the Viterbi parse are unchanged (they only see parses that reach a terminal).
  // shadow: append a copy of state 0 = (0,0) as state S, closed under the same transitions but isolated from every other
  // state (requires that state 0 is closed, Engine::linear_ok_).  One outside sweep with the "has motif" terminals on the
  // states of the pattern and the "no motif" terminal on the shadow then yields both outside passes of the train schedule.
  // row_pad: the rows of the compact tables of the scaled-linear pipeline (AutomatonLayout::tab_*) are padded to a multiple of
  // row_pad doubles (8 = every row starts on a 64-byte line; 1, the default since round 4 = no padding: 70 instead of 104 doubles
  // per cell for ((.*.)), and 4.7 % off the train evaluation -- the memory system is bound by requests).  cell_major: the seven rows of a cell lie side by side (one record of
  // tab_row doubles per cell, padded to a multiple of 8) instead of one plane after the other (AutomatonLayout::tab_cell).
  void flatten(AutomatonLayout* lay, std::vector<int32_t>* ints, bool only_state0 = false, bool prune = false,
               bool shadow = false, int row_pad = 1, bool cell_major = false) const;

  // Static liveness of the (structural state, interval state) pairs, from the rule table alone in the
  // boolean semiring: inside_live[e][s] = some sequence gives inside(., ., e, s) a non-zero weight; useful[e][s] = inside-live
  // and reachable from a terminal O(L, (0,0) | (0,M-1) | (0,M-2)) through transitions whose siblings are inside-live, i.e. the
  // entry can occur in a complete parse.  e = ST_P .. ST_L, ST_O (8 planes).
  struct Liveness { std::vector<char> inside_live[8], useful[8]; };
  Liveness liveness() const;

 private:
  std::string pattern_, reg_;
  std::vector<char> node_;
  std::vector<int> mate_;
  std::vector<std::vector<int>> edge_to_, edge_from_;
  std::vector<int> theta_row_, row_width_, row_off_;
  std::vector<char> reach_, reach_loop_;
  std::vector<IntervalState> states_;
  std::vector<int> n2s_;
  std::vector<char> loop_flag_;
  std::vector<std::vector<int>> right_, left_, pair_;
  std::vector<std::array<int, 4>> quads_;
};

// A degenerate one-state automaton (S = 1, no emissions) under which the motif DP reduces to the
// plain McCaskill partition function used by the BPP filter (energy_model.hpp:559-661).
void flatten_trivial(AutomatonLayout* lay, std::vector<int32_t>* ints);

}  // namespace elemdp
