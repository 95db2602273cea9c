// automaton.cpp -- see automaton.h.  Host only.
#include "automaton.h"

#include <algorithm>
#include <sstream>
#include <stdexcept>

namespace elemdp {
namespace {

bool is_background(char c) { return c == 'z' || c == '*' || c == 'o'; }
bool emits_single(char c) { return c == 'z' || c == '.' || c == '*' || c == 'o'; }
bool is_weighted(char c) { return c == '.' || c == '(' || c == ')'; }  // motif_model.hpp:131-134

void transitive_closure(std::vector<char>& m, int n) {  // Warshall, profile_hmm.hpp:357-366
  for (int k = 0; k < n; ++k)
    for (int i = 0; i < n; ++i)
      if (m[i * n + k])
        for (int j = 0; j < n; ++j)
          if (m[k * n + j]) m[i * n + j] = 1;
}

// CSR builder over S keys
struct Csr {
  std::vector<std::vector<int32_t>> rows;
  int width;
  Csr(int S, int w) : rows(S), width(w) {}
  void add(int key, std::initializer_list<int32_t> v) { rows[key].insert(rows[key].end(), v); }
  // appends offsets (S+1) then entries to the blob; returns {off_pos, ent_pos}
  // appends the grouping key of every entry; returns its position
  int32_t emit_targets(std::vector<int32_t>* blob) const {
    int32_t pos = (int32_t)blob->size();
    for (size_t k = 0; k < rows.size(); ++k)
      for (size_t e = 0; e < rows[k].size() / width; ++e) blob->push_back((int32_t)k);
    return pos;
  }
  int32_t count() const { int32_t n = 0; for (auto const& r : rows) n += (int32_t)r.size() / width; return n; }
  std::pair<int32_t, int32_t> emit(std::vector<int32_t>* blob) const {
    int32_t off_pos = (int32_t)blob->size();
    int32_t n = 0;
    for (auto const& r : rows) { blob->push_back(n); n += (int32_t)r.size() / width; }
    blob->push_back(n);
    int32_t ent_pos = (int32_t)blob->size();
    for (auto const& r : rows) blob->insert(blob->end(), r.begin(), r.end());
    return {off_pos, ent_pos};
  }
};

}  // namespace

Automaton::Automaton(const std::string& pattern) : pattern_(pattern) {
  if (pattern.empty()) throw std::runtime_error("empty motif");
  // regularise: runs of '*' collapse to one, leading / trailing '*' are dropped (profile_hmm.hpp:188-204)
  for (char c : pattern) {
    if (c == '*' && !reg_.empty() && reg_.back() == '*') continue;
    reg_.push_back(c);
  }
  size_t b = reg_.find_first_not_of('*');
  reg_ = (b == std::string::npos) ? std::string() : reg_.substr(b);
  size_t e = reg_.find_last_not_of('*');
  if (e != std::string::npos) reg_.erase(e + 1);

  node_.push_back('z');
  for (char c : reg_) node_.push_back(c);
  node_.push_back('o');
  const int m = M();

  // bracket mates
  mate_.assign(m, -1);
  {
    std::vector<int> open;
    for (int h = 0; h < m; ++h) {
      if (node_[h] == '(') open.push_back(h);
      else if (node_[h] == ')') {
        if (open.empty()) throw std::runtime_error("unmatched brackets");
        mate_[h] = open.back();
        mate_[open.back()] = h;
        open.pop_back();
      }
    }
    if (!open.empty()) throw std::runtime_error("unmatched brackets");
  }

  // node edges: h -> h-1, h -> h (self), and h -> h-2 across a '*' (profile_hmm.hpp:257-283)
  edge_to_.assign(m, {});
  edge_from_.assign(m, {});
  for (int h = 0; h < m; ++h) {
    if (h > 0) {
      if (node_[h - 1] == '*') { edge_to_[h].push_back(h - 2); edge_from_[h - 2].push_back(h); }
      edge_to_[h].push_back(h - 1);
      edge_from_[h - 1].push_back(h);
    }
    edge_to_[h].push_back(h);
    edge_from_[h].push_back(h);
  }

  // theta rows: row 0 = shared background (z, *, o); one row per '.', one 6-wide row per ')'
  theta_row_.assign(m, -1);
  row_width_.assign(1, 4);
  for (int h = 0; h < m; ++h) {
    switch (node_[h]) {
      case ')': theta_row_[h] = (int)row_width_.size(); row_width_.push_back(6); break;
      case '.': theta_row_[h] = (int)row_width_.size(); row_width_.push_back(4); break;
      case '*': case 'z': case 'o': theta_row_[h] = 0; break;
      case '(': break;
      default: throw std::runtime_error(std::string("bad motif char: ") + node_[h]);
    }
  }
  row_off_.assign(1, 0);
  for (int w : row_width_) row_off_.push_back(row_off_.back() + w);

  // reachability between nodes (profile_hmm.hpp:316-354)
  reach_.assign(m * m, 0);
  reach_loop_.assign(m * m, 0);
  for (int h = 0; h < m; ++h) {
    if (node_[h] == ')') {
      for (int h1 : edge_to_[mate_[h]]) reach_[h1 * m + h] = 1;
    } else if (node_[h] != '(') {
      for (int h1 : edge_to_[h]) { reach_[h1 * m + h] = 1; reach_loop_[h1 * m + h] = 1; }
    }
    reach_[h * m + h] = 1;
    reach_loop_[h * m + h] = 1;
  }
  transitive_closure(reach_, m);
  transitive_closure(reach_loop_, m);

  // interval states ordered by r ascending, l descending (profile_hmm.hpp:369-384)
  n2s_.assign(m * m, -1);
  for (int r = 0; r < m; ++r)
    for (int l = r; l >= 0; --l)
      if (reach_[l * m + r]) {
        n2s_[l * m + r] = (int)states_.size();
        states_.push_back(IntervalState{(int)states_.size(), l, r});
      }
  const int S_ = S();
  loop_flag_.assign(S_, 0);
  for (auto const& s : states_) loop_flag_[s.id] = reach_loop_[s.l * m + s.r];

  // transitions (profile_hmm.hpp:387-449)
  right_.assign(S_, {});
  left_.assign(S_, {});
  pair_.assign(S_, {});
  for (auto const& s : states_) {
    if (emits_single(node_[s.r]))
      for (int h : edge_to_[s.r])
        if (s.l <= h && reach_[s.l * m + h]) right_[s.id].push_back(state_id(s.l, h));
  }
  for (auto const& s : states_) {
    if (emits_single(node_[s.l]))
      for (int h : edge_to_[s.l])
        if (h <= s.r && reach_[h * m + s.r]) left_[state_id(h, s.r)].push_back(s.id);
  }
  for (int hr = 0; hr < m; ++hr) {
    if (node_[hr] != ')') continue;
    const int kl = mate_[hr];
    for (int hl : edge_to_[kl]) {
      const int parent = state_id(hl, hr);
      for (int kr : edge_to_[hr])
        if (reach_[kl * m + kr]) pair_[parent].push_back(state_id(kl, kr));
    }
  }
  for (auto const& s : states_) {  // background "pairs": both ends emitted by z / * / o
    if (!is_background(node_[s.r])) continue;
    for (int hl : edge_from_[s.l]) {
      if (!is_background(node_[hl])) continue;
      for (int hr : edge_to_[s.r])
        if (reach_[hl * m + hr]) pair_[s.id].push_back(state_id(hl, hr));
    }
  }

  // interior-loop quadruples {s, s1, s2, s3} (profile_hmm.hpp:451-463)
  for (auto const& s2 : states_) {
    if (!loop_flag_[s2.id]) continue;
    for (auto const& s3 : states_) {
      if (!loop_flag_[s3.id]) continue;
      if (s3.r < s2.l || !reach_[s2.r * m + s3.l] || !reach_[s2.l * m + s3.r]) continue;
      quads_.push_back({state_id(s2.l, s3.r), state_id(s2.r, s3.l), s2.id, s3.id});
    }
  }
}

std::vector<std::array<int, 2>> Automaton::splits(int s) const {
  std::vector<std::array<int, 2>> out;
  const IntervalState& st = states_[s];
  const int m = M();
  for (int h = st.l; h <= st.r; ++h)
    if (reach_[st.l * m + h] && reach_[h * m + st.r]) out.push_back({state_id(st.l, h), state_id(h, st.r)});
  return out;
}

std::string Automaton::to_json() const {
  std::ostringstream o;
  auto list = [&](const std::vector<int>& v) {
    o << "[";
    for (size_t i = 0; i < v.size(); ++i) o << (i ? "," : "") << v[i];
    o << "]";
  };
  const int m = M();
  o << "{\"reg_pattern\":\"" << reg_ << "\",\"M\":" << m << ",\"S\":" << S() << ",\"node\":\"";
  for (char c : node_) o << c;
  o << "\",\"theta_id\":";
  list(theta_row_);
  o << ",\"theta_sizes\":";
  list(row_width_);
  o << ",\"state\":[";
  for (int s = 0; s < S(); ++s) o << (s ? "," : "") << "[" << states_[s].l << "," << states_[s].r << "]";
  o << "],\"loop_state\":[";
  {
    bool first = true;
    for (int s = 0; s < S(); ++s)
      if (loop_flag_[s]) { o << (first ? "" : ",") << s; first = false; }
  }
  o << "],\"reachable\":[";
  for (int a = 0; a < m; ++a) {
    o << (a ? "," : "") << "[";
    for (int b = 0; b < m; ++b) o << (b ? "," : "") << int(reach_[a * m + b]);
    o << "]";
  }
  o << "],\"right\":[";
  for (int s = 0; s < S(); ++s) { if (s) o << ","; list(right_[s]); }
  o << "],\"left\":[";
  for (int s = 0; s < S(); ++s) { if (s) o << ","; list(left_[s]); }
  o << "],\"pair\":[";
  for (int s = 0; s < S(); ++s) { if (s) o << ","; list(pair_[s]); }
  o << "],\"loop_loop\":[";
  for (size_t i = 0; i < quads_.size(); ++i)
    o << (i ? "," : "") << "[" << quads_[i][0] << "," << quads_[i][1] << "," << quads_[i][2] << "," << quads_[i][3] << "]";
  o << "]";
  // static liveness per plane P,E,M,B,1,2,L,O (see liveness()): lists of state ids
  const Liveness lv = liveness();
  for (int which = 0; which < 2; ++which) {
    o << (which ? ",\"useful\":[" : ",\"inside_live\":[");
    for (int e = 0; e < 8; ++e) {
      std::vector<int> ids;
      for (int s = 0; s < S(); ++s) if ((which ? lv.useful : lv.inside_live)[e][s]) ids.push_back(s);
      if (e) o << ",";
      list(ids);
    }
    o << "]";
  }
  o << "}";
  return o.str();
}

Automaton::Liveness Automaton::liveness() const {
  const int S_ = S(), m = M();
  Liveness lv;
  for (auto& v : lv.inside_live) v.assign(S_, 0);
  for (auto& v : lv.useful) v.assign(S_, 0);
  std::vector<std::array<int, 3>> sp;   // (parent, (l,h), (h,r))
  for (int s = 0; s < S_; ++s)
    for (auto const& p : splits(s)) sp.push_back({s, p[0], p[1]});
  auto& I = lv.inside_live;
  bool changed = true;
  auto set = [&](std::vector<char>& v, int s) { if (!v[s]) { v[s] = 1; changed = true; } };
  // inside: initial values L(i,i,(k,k)) = 1, O(0,(0,0)) = 1 (motif_trainer.hpp:89-98), then the rules to a fixed point
  for (int s = 0; s < S_; ++s) if (loop_flag_[s] && states_[s].l == states_[s].r) I[ST_L][s] = 1;
  I[ST_O][state_id(0, 0)] = 1;
  while (changed) {
    changed = false;
    for (int s = 0; s < S_; ++s) {
      for (int c : pair_[s]) if (I[ST_E][c] || I[ST_P][c]) set(I[ST_P], s);           // rules 1a, 1b
      for (int c : right_[s]) {
        if (I[ST_2][c]) set(I[ST_2], s);                                              // 3a
        if (loop_flag_[s] && I[ST_L][c]) set(I[ST_L], s);                             // L <- L
        if (I[ST_O][c]) set(I[ST_O], s);                                              // 8
      }
      if (I[ST_P][s]) set(I[ST_2], s);                                                // 3b
      if (I[ST_2][s] || I[ST_B][s]) set(I[ST_1], s);                                  // 4a, 4b
      for (int c : left_[s]) if (I[ST_M][c]) set(I[ST_M], s);                         // 5a
      if (I[ST_B][s]) set(I[ST_M], s);                                                // 5b
      if (I[ST_M][s] || (loop_flag_[s] && I[ST_L][s])) set(I[ST_E], s);               // 6a, 6b
    }
    for (auto const& t : sp) {
      if (I[ST_1][t[1]] && I[ST_2][t[2]]) set(I[ST_B], t[0]);                         // 2
      if (I[ST_O][t[1]] && I[ST_P][t[2]]) set(I[ST_O], t[0]);                         // 7
    }
    for (auto const& q : quads_)
      if (I[ST_P][q[1]] && I[ST_L][q[2]] && I[ST_L][q[3]]) set(I[ST_E], q[0]);        // 6c
  }
  // outside: from the terminals down; a child is reached when it and its siblings are inside-live
  auto& U = lv.useful;
  auto reach = [&](int e, int s) { if (I[e][s]) set(U[e], s); };
  for (int t : {state_id(0, 0), state_id(0, m - 1), state_id(0, m - 2)}) if (t >= 0) reach(ST_O, t);
  changed = true;
  while (changed) {
    changed = false;
    for (int s = 0; s < S_; ++s) {
      if (U[ST_O][s]) for (int c : right_[s]) reach(ST_O, c);
      if (U[ST_P][s]) for (int c : pair_[s]) { reach(ST_E, c); reach(ST_P, c); }
      if (U[ST_2][s]) { for (int c : right_[s]) reach(ST_2, c); reach(ST_P, s); }
      if (U[ST_1][s]) { reach(ST_2, s); reach(ST_B, s); }
      if (U[ST_M][s]) { for (int c : left_[s]) reach(ST_M, c); reach(ST_B, s); }
      if (U[ST_E][s]) { reach(ST_M, s); if (loop_flag_[s]) reach(ST_L, s); }
      if (U[ST_L][s]) for (int c : right_[s]) reach(ST_L, c);
    }
    for (auto const& t : sp) {
      if (U[ST_B][t[0]] && I[ST_1][t[1]] && I[ST_2][t[2]]) { reach(ST_1, t[1]); reach(ST_2, t[2]); }
      if (U[ST_O][t[0]] && I[ST_O][t[1]] && I[ST_P][t[2]]) { reach(ST_O, t[1]); reach(ST_P, t[2]); }
    }
    for (auto const& q : quads_)
      if (U[ST_E][q[0]] && I[ST_P][q[1]] && I[ST_L][q[2]] && I[ST_L][q[3]]) { reach(ST_P, q[1]); reach(ST_L, q[2]); reach(ST_L, q[3]); }
  }
  return lv;
}

void Automaton::flatten(AutomatonLayout* lay, std::vector<int32_t>* ints, bool only_state0, bool prune, bool shadow,
                        int row_pad, bool cell_major) const {
  const int S_ = S(), m = M();
  const int ST = S_ + (shadow ? 1 : 0);   // states of the flattened automaton (the shadow of (0,0) is the last one)
  Liveness lv;
  if (prune) lv = liveness();
  const auto& I = lv.inside_live;
  const auto& U = lv.useful;
  // a transition is kept when its parent is useful and its children are inside-live in at least one of the rules that
  // walk the list (the children are then useful, too); arguments are reference state ids
  auto live_right = [&](int s, int c) {
    return !prune || (U[ST_2][s] && I[ST_2][c]) || (U[ST_L][s] && I[ST_L][c]) || (U[ST_O][s] && I[ST_O][c]);
  };
  auto live_left = [&](int s, int c) { return !prune || (U[ST_M][s] && I[ST_M][c]); };
  auto live_pair = [&](int s, int c) { return !prune || (U[ST_P][s] && (I[ST_E][c] || I[ST_P][c])); };
  auto live_split = [&](int s, int a, int b) {
    return !prune || (U[ST_B][s] && I[ST_1][a] && I[ST_2][b]) || (U[ST_O][s] && I[ST_O][a] && I[ST_P][b]);
  };
  auto live_quad = [&](const std::array<int, 4>& q) {
    return !prune || (U[ST_E][q[0]] && I[ST_P][q[1]] && I[ST_L][q[2]] && I[ST_L][q[3]]);
  };
  // Internal state order: with the pruned lists, the states that can take part in a bifurcation (useful in plane B, 1 or 2)
  // come first -- the kernels stage only that prefix of the rows of those planes.  (0,0) stays state 0; the relative order
  // of the others is the reference's, so every list keeps its order (the Viterbi tie rule depends on it).  ref_of[k] =
  // reference id of internal state k, id_of = the inverse.
  std::vector<int> ref_of, id_of(S_, -1);
  int n_front = S_;
  if (prune) {
    auto front = [&](int s) { return s == 0 || U[ST_B][s] || U[ST_1][s] || U[ST_2][s]; };
    for (int s = 0; s < S_; ++s) if (front(s)) ref_of.push_back(s);
    n_front = (int)ref_of.size();
    for (int s = 0; s < S_; ++s) if (!front(s)) ref_of.push_back(s);
  } else {
    for (int s = 0; s < S_; ++s) ref_of.push_back(s);
  }
  for (int k = 0; k < S_; ++k) id_of[ref_of[k]] = k;

  AutomatonLayout& A = *lay;
  ints->clear();
  A.S = ST;
  A.n_active = only_state0 ? 1 : ST;
  A.n_front = only_state0 ? 1 : (shadow ? ST : n_front);
  A.shadow = shadow ? S_ : -1;
  A.M = m;
  A.n_theta = n_theta();
  A.n_rows = n_rows();
  A.s00 = id_of[state_id(0, 0)];
  A.s0m1 = id_of[state_id(0, m - 1)];
  A.s0m2 = id_of[state_id(0, m - 2)];
  auto per_state = [&](auto fn) {
    int32_t pos = (int32_t)ints->size();
    for (int k = 0; k < S_; ++k) ints->push_back(fn(states_[ref_of[k]]));
    if (shadow) ints->push_back(fn(states_[0]));
    return pos;
  };
  A.st_l = per_state([&](const IntervalState& s) { return s.l; });
  A.st_r = per_state([&](const IntervalState& s) { return s.r; });
  A.st_is_loop = per_state([&](const IntervalState& s) { return (int)loop_flag_[s.id]; });
  A.st_row_r = per_state([&](const IntervalState& s) { return theta_row_[s.r]; });
  A.st_row_l = per_state([&](const IntervalState& s) { return theta_row_[s.l]; });
  A.st_pair_r = per_state([&](const IntervalState& s) { return (int)(node_[s.r] == ')'); });
  A.st_w_r = per_state([&](const IntervalState& s) { return (int)is_weighted(node_[s.r]); });
  A.st_w_l = per_state([&](const IntervalState& s) { return (int)is_weighted(node_[s.l]); });
  A.st_lam = per_state([&](const IntervalState& s) { return s.l == s.r ? 0 : 1; });
  A.st_ref = per_state([&](const IntervalState& s) { return s.id; });
  A.row_off = (int32_t)ints->size();
  for (int v : row_off_) ints->push_back(v);

  // tau applies to a self-loop on the emitting node (motif_model.hpp:250-251, 278-279, 352-353); reference ids
  auto tau_right = [&](int par, int ch) { return (int)(states_[par].r == states_[ch].r && node_[states_[par].r] == '.'); };
  auto tau_left = [&](int par, int ch) { return (int)(states_[par].l == states_[ch].l && node_[states_[par].l] == '.'); };
  auto tau_pair = [&](int par, int ch) { return (int)(states_[par].r == states_[ch].r && node_[states_[ch].r] == ')'); };

  Csr right(ST, 2), left(ST, 2), pair(ST, 2), rright(ST, 2), rleft(ST, 2), rpair(ST, 2);
  auto keep = [&](std::initializer_list<int> ids) {
    if (!only_state0) return true;
    for (int v : ids) if (v != 0) return false;
    return true;
  };
  // (parents in internal order, the children of a parent in the reference's list order; s, c = reference ids)
  for (int k = 0; k < S_; ++k) {
    const int s = ref_of[k];
    for (int c : right_[s]) if (keep({s, c}) && live_right(s, c)) { right.add(k, {id_of[c], tau_right(s, c)}); }
    for (int c : left_[s]) if (keep({s, c}) && live_left(s, c)) { left.add(k, {id_of[c], tau_left(s, c)}); }
    for (int c : pair_[s]) if (keep({s, c}) && live_pair(s, c)) { pair.add(k, {id_of[c], tau_pair(s, c)}); }
  }
  // the reverse lists enumerate the parents of a child in the REFERENCE's parent order (as before the renumbering)
  for (int s = 0; s < S_; ++s) {
    for (int c : right_[s]) if (keep({s, c}) && live_right(s, c)) rright.add(id_of[c], {id_of[s], tau_right(s, c)});
    for (int c : left_[s]) if (keep({s, c}) && live_left(s, c)) rleft.add(id_of[c], {id_of[s], tau_left(s, c)});
    for (int c : pair_[s]) if (keep({s, c}) && live_pair(s, c)) rpair.add(id_of[c], {id_of[s], tau_pair(s, c)});
  }
  Csr split(ST, 2), split1(ST, 2), split2(ST, 2);
  for (int s = 0; s < S_; ++s)
    for (auto const& p : splits(s)) {
      if (!keep({s, p[0], p[1]}) || !live_split(s, p[0], p[1])) continue;
      split.add(id_of[s], {id_of[p[0]], id_of[p[1]]});
      split1.add(id_of[p[0]], {id_of[s], id_of[p[1]]});
      split2.add(id_of[p[1]], {id_of[s], id_of[p[0]]});
    }
  Csr quad(ST, 3), quad1(ST, 3), quad2(ST, 3), quad3(ST, 3);
  for (auto const& q : quads_) {
    if (!keep({q[0], q[1], q[2], q[3]}) || !live_quad(q)) continue;
    const int q0 = id_of[q[0]], q1 = id_of[q[1]], q2 = id_of[q[2]], q3 = id_of[q[3]];
    quad.add(q0, {q1, q2, q3});
    quad1.add(q1, {q0, q2, q3});
    quad2.add(q2, {q0, q1, q3});
    quad3.add(q3, {q0, q1, q2});
  }
  if (shadow) {
    // the transitions among state 0 once more for the shadow state.  Forward lists (by parent): row 0 holds children 0 only
    // ((0,0) is closed).  Reverse lists (by child): row 0 also names the other parents of (0,0) -- the shadow keeps the
    // entries whose states are all 0, i.e. it is reachable from the shadow alone.
    auto copy_row = [&](Csr& c, int width_states, bool reverse) {
      std::vector<int32_t> row;
      for (size_t e = 0; e + c.width <= c.rows[0].size(); e += c.width) {
        bool all0 = true;
        for (int k = 0; k < width_states; ++k) all0 = all0 && c.rows[0][e + k] == 0;
        if (!all0) {
          if (!reverse) throw std::runtime_error("shadow state: (0,0) is not closed");
          continue;
        }
        for (int k = 0; k < c.width; ++k) row.push_back(k < width_states ? S_ : c.rows[0][e + k]);
      }
      c.rows[S_] = row;
    };
    for (Csr* c : {&right, &left, &pair}) copy_row(*c, 1, false);   // (state, tau flag)
    for (Csr* c : {&rright, &rleft, &rpair}) copy_row(*c, 1, true);
    copy_row(split, 2, false);
    for (Csr* c : {&split1, &split2}) copy_row(*c, 2, true);
    copy_row(quad, 3, false);
    for (Csr* c : {&quad1, &quad2, &quad3}) copy_row(*c, 3, true);
  }
  auto put = [&](const Csr& c, int32_t* off, int32_t* ent) { auto p = c.emit(ints); *off = p.first; *ent = p.second; };
  // small part: unary transition lists (staged in LDS together with the per-state attributes)
  put(right, &A.right_off, &A.right_ent);
  put(left, &A.left_off, &A.left_ent);
  put(pair, &A.pair_off, &A.pair_ent);
  put(rright, &A.rright_off, &A.rright_ent);
  put(rleft, &A.rleft_off, &A.rleft_ent);
  put(rpair, &A.rpair_off, &A.rpair_ent);
  // pairs (s1, t) of the factorised rule 2: the kept splits (s; s1, t), closed under t' in right(t) (the tail of unpaired
  // bases behind the last stem of a multiloop part grows by right emissions); internal ids, split order
  std::vector<std::array<int, 3>> ap_keep;
  Csr chain_keep(1, 2), rchain_keep(1, 2);
  {
    std::vector<std::array<int, 3>> ap;   // (s1, t, tgt)
    auto find = [&](int s1, int t) { for (size_t k = 0; k < ap.size(); ++k) if (ap[k][0] == s1 && ap[k][1] == t) return (int)k; return -1; };
    auto ref = [&](int k) { return k < S_ ? ref_of[k] : 0; };   // (the shadow state has the liveness of (0,0))
    for (int k = 0; k < ST; ++k)
      for (size_t e = 0; e + 1 < split.rows[k].size(); e += 2) {
        const int s1 = split.rows[k][e], t = split.rows[k][e + 1];
        // (entries kept only for rule 7 pair an O state with a P state: they never carry weight in rule 2)
        if (prune && !(U[ST_B][ref(k)] && I[ST_1][ref(s1)] && I[ST_2][ref(t)])) continue;
        if (find(s1, t) < 0) ap.push_back({s1, t, k});
      }
    for (size_t k = 0; k < ap.size(); ++k) {   // closure (the list grows while it is walked)
      const int s1 = ap[k][0], t = ap[k][1];
      for (size_t e = 0; e + 1 < right.rows[t].size(); e += 2) {
        const int tc = right.rows[t][e];
        if (prune && !I[ST_2][ref(tc)]) continue;
        if (find(s1, tc) < 0) ap.push_back({s1, tc, -1});
      }
    }
    const int n_ap = (int)ap.size();
    A.n_ap = n_ap;
    A.ap_s1 = (int32_t)ints->size(); for (auto const& x : ap) ints->push_back(x[0]);
    A.ap_t = (int32_t)ints->size(); for (auto const& x : ap) ints->push_back(x[1]);
    A.ap_tgt = (int32_t)ints->size(); for (auto const& x : ap) ints->push_back(x[2]);
    Csr chain(std::max(n_ap, 1), 2), rchain(std::max(n_ap, 1), 2), by_s1(ST, 1), by_t(ST, 1);
    for (int k = 0; k < n_ap; ++k) {
      const int s1 = ap[k][0], t = ap[k][1];
      by_s1.add(s1, {k});
      by_t.add(t, {k});
      for (size_t e = 0; e + 1 < right.rows[t].size(); e += 2) {
        const int kc = find(s1, right.rows[t][e]);
        if (kc < 0) continue;
        chain.add(k, {kc, right.rows[t][e + 1]});
        rchain.add(kc, {k, right.rows[t][e + 1]});
      }
    }
    put(chain, &A.ap_chain_off, &A.ap_chain_ent);
    put(rchain, &A.ap_rchain_off, &A.ap_rchain_ent);
    put(by_s1, &A.ap_by_s1_off, &A.ap_by_s1_ent);
    put(by_t, &A.ap_by_t_off, &A.ap_by_t_ent);
    ap_keep = ap; chain_keep = chain; rchain_keep = rchain;
  }
  // compact tables of the scaled-linear pipeline: one column per state that is useful in the plane (all states without pruning;
  // the shadow state has the liveness of (0,0)); strides padded to a multiple of `row_pad` doubles
  {
    const int pad = row_pad > 0 ? row_pad : 1;
    A.tab_cmap = (int32_t)ints->size();
    int cs = 0;
    for (int e = 0; e < 7; ++e) {
      int n = 0;
      for (int k = 0; k < ST; ++k) {
        const int r = k < S_ ? ref_of[k] : 0;
        const bool live = !prune || only_state0 || U[e][r];
        ints->push_back(live ? n++ : -1);
      }
      A.tab_rs[e] = std::max(pad, ((n + pad - 1) / pad) * pad);
    }
    // (the table-driven kernels keep X = P * exp(lambda e_ml) in the rows of the B plane under P's columns: those rows hold at
    // least as many columns as P's)
    A.tab_rs[ST_B] = std::max(A.tab_rs[ST_B], A.tab_rs[ST_P]);
    for (int e = 0; e < 7; ++e) {
      A.tab_cs[e] = cs;
      cs += A.tab_rs[e];
    }
    A.tab_row = cell_major ? ((cs + 7) / 8) * 8 : cs;     // (cell records start on 64-byte lines)
    A.tab_cell = cell_major ? 1 : 0;
    A.ap_rs = std::max(pad, ((A.n_ap + pad - 1) / pad) * pad);
  }
  // table-driven unary phases (device_layout.h: fp_*): programs per state, static attributes per forward transition.  They
  // travel with the tuple lists of their direction (the "big" runs below), not with the small part every kernel stages.
  std::vector<int32_t> prog_in, prog_out, attr_r, attr_p, pair_rec, scan_fl;
  // the states that have a column in some plane, in state order: the unary phases of the table-driven kernels give them a lane each
  // per cell, and the heavy sums of a cell are indexed by a state's position in this list (li; -1 for the others)
  std::vector<int32_t> live_states;
  std::vector<int> li(ST, -1);
  for (int k = 0; k < ST; ++k) {
    bool any = false;
    for (int e = 0; e < 7; ++e) any = any || (*ints)[A.tab_cmap + e * ST + k] >= 0;
    if (any) { li[k] = (int)live_states.size(); live_states.push_back(k); }
  }
  A.st_live = (int32_t)ints->size(); ints->insert(ints->end(), live_states.begin(), live_states.end());
  A.st_li = (int32_t)ints->size(); ints->insert(ints->end(), li.begin(), li.end());
  A.n_lane = (int32_t)live_states.size();
  {
    auto colof = [&](int e, int k) { return (*ints)[A.tab_cmap + e * ST + k] & 0xff; };   // (-1 -> 0xff)
    auto base_of = [&](const Csr& c) {   // first transition id of every row
      std::vector<int> b(ST + 1, 0);
      for (int k = 0; k < ST; ++k) b[k + 1] = b[k] + (int)c.rows[k].size() / c.width;
      return b;
    };
    const std::vector<int> br = base_of(right), bp = base_of(pair), bl = base_of(left);
    A.n_wr = br[ST]; A.n_wp = bp[ST]; A.n_wl = bl[ST];
    auto fwd_id = [&](const Csr& c, const std::vector<int>& b, int par, int child) {   // id of the transition par -> child
      for (size_t e = 0; e + 1 < c.rows[par].size(); e += 2) if (c.rows[par][e] == child) return b[par] + (int)e / 2;
      throw std::runtime_error("flatten: reverse transition without its forward entry");
    };
    auto attr = [&](int32_t pos, int k) { return (*ints)[pos + k]; };
    bool ok = A.n_wr < 32768 && A.n_wp < 32768 && A.n_wl < 32768 && A.tab_row < 250;
    {   // the train kernels keep X = P * exp(lambda e_ml) in the rows of the (otherwise unused) B plane under P's columns
      int n_p = 0;
      for (int k = 0; k < ST; ++k) n_p = std::max(n_p, (*ints)[A.tab_cmap + ST_P * ST + k] + 1);
      ok = ok && n_p <= A.tab_rs[ST_B];
    }
    A.fp_max_p = 0;
    for (int k = 0; k < ST; ++k)
      for (const Csr* c : {&pair, &rpair}) A.fp_max_p = std::max(A.fp_max_p, (int32_t)c->rows[k].size() / 2);
    for (int k = 0; k < ST; ++k) {
      for (const Csr* c : {&right, &rright}) ok = ok && (int)c->rows[k].size() / 2 <= kFastR;
      for (const Csr* c : {&pair, &rpair}) ok = ok && (int)c->rows[k].size() / 2 <= kFastP;
      for (const Csr* c : {&left, &rleft}) ok = ok && (int)c->rows[k].size() / 2 <= kFastL;
    }
    A.fp_ok = ok ? 1 : 0;
    for (int k = 0; k < ST; ++k) {
      int32_t w[kFastW] = {0};
      const int nR = ok ? (int)right.rows[k].size() / 2 : 0, nP = ok ? (int)pair.rows[k].size() / 2 : 0, nL = ok ? (int)left.rows[k].size() / 2 : 0;
      w[0] = (attr(A.st_is_loop, k) ? 1 : 0) | (attr(A.st_l, k) == attr(A.st_r, k) ? 2 : 0) | (attr(A.st_lam, k) ? 4 : 0) |
             (attr(A.st_w_r, k) ? 8 : 0) | (nR << 8) | (nP << 12) | (nL << 16);
      w[1] = colof(ST_P, k) | (colof(ST_E, k) << 8) | (colof(ST_M, k) << 16) | (colof(ST_B, k) << 24);
      w[2] = colof(ST_1, k) | (colof(ST_2, k) << 8) | (colof(ST_L, k) << 16);
      for (int u = 0; u < nR; ++u) {
        const int ch = right.rows[k][2 * u];
        w[4 + u] = colof(ST_L, ch) | (colof(ST_2, ch) << 8) | ((br[k] + u) << 16);
      }
      for (int u = 0; u < nP; ++u) {
        const int ch = pair.rows[k][2 * u];
        w[8 + u] = colof(ST_E, ch) | (colof(ST_P, ch) << 8) | ((bp[k] + u) << 16) | (attr(A.st_w_l, ch) ? (int32_t)0x80000000 : 0);
      }
      for (int u = 0; u < nL; ++u) {
        const int ch = left.rows[k][2 * u];
        w[12 + u] = colof(ST_M, ch) | ((bl[k] + u) << 16) | (attr(A.st_w_l, ch) ? (int32_t)0x80000000 : 0);
      }
      prog_in.insert(prog_in.end(), w, w + kFastW);
    }
    auto rowoff1 = [&](int row) { return row >= 0 ? (*ints)[A.row_off + row] - 1 : -1000000; };   // index of base b: + b
    for (int k = 0; k < ST; ++k) {
      int32_t w[kFastW] = {0};
      const int nR = ok ? (int)rright.rows[k].size() / 2 : 0, nP = ok ? (int)rpair.rows[k].size() / 2 : 0, nL = ok ? (int)rleft.rows[k].size() / 2 : 0;
      w[0] = (attr(A.st_is_loop, k) ? 1 : 0) | (attr(A.st_lam, k) ? 4 : 0) | (attr(A.st_w_l, k) ? 16 : 0) | (k == A.shadow ? 32 : 0) |
             (nR << 8) | (nP << 12) | (nL << 16);
      w[1] = colof(ST_P, k) | (colof(ST_E, k) << 8) | (colof(ST_M, k) << 16) | (colof(ST_B, k) << 24);
      w[2] = colof(ST_1, k) | (colof(ST_2, k) << 8) | (colof(ST_L, k) << 16);
      w[3] = rowoff1(attr(A.st_row_l, k));
      for (int u = 0; u < nR; ++u) {
        const int par = rright.rows[k][2 * u];
        w[4 + u] = colof(ST_2, par) | (colof(ST_L, par) << 8) | (fwd_id(right, br, par, k) << 16) | (attr(A.st_is_loop, par) ? (int32_t)0x80000000 : 0);
      }
      for (int u = 0; u < nP; ++u) {
        const int par = rpair.rows[k][2 * u];
        w[8 + u] = colof(ST_P, par) | (fwd_id(pair, bp, par, k) << 16);
      }
      for (int u = 0; u < nL; ++u) {
        const int par = rleft.rows[k][2 * u];
        w[12 + u] = colof(ST_M, par) | (fwd_id(left, bl, par, k) << 16);
      }
      prog_out.insert(prog_out.end(), w, w + kFastW);
    }
    // static attributes per forward transition.  right: {index of base b of the parent's r-row minus b, flags: bit 0 the
    // position weight applies (parent's r-node), bit 1 lambda class of the parent}; pair: {flags: bit 0 the parent's r-node
    // emits the pair as a pair type, 1 position weight of the child's l-node, 2 of the parent's r-node, 3 lambda class of the
    // parent; r-row of the parent; l-row of the child (same convention)}
    for (int k = 0; k < ST; ++k)
      for (size_t e = 0; e + 1 < right.rows[k].size(); e += 2) {
        attr_r.push_back(rowoff1(attr(A.st_row_r, k)));
        attr_r.push_back((attr(A.st_w_r, k) ? 1 : 0) | (attr(A.st_lam, k) ? 2 : 0));
      }
    for (int k = 0; k < ST; ++k)
      for (size_t e = 0; e + 1 < pair.rows[k].size(); e += 2) {
        const int ch = pair.rows[k][e];
        attr_p.push_back((attr(A.st_pair_r, k) ? 1 : 0) | (attr(A.st_w_l, ch) ? 2 : 0) | (attr(A.st_w_r, k) ? 4 : 0) | (attr(A.st_lam, k) ? 8 : 0));
        attr_p.push_back(rowoff1(attr(A.st_row_r, k)));
        attr_p.push_back(rowoff1(attr(A.st_row_l, ch)));
      }
    // scan flags per forward transition (device_layout.h: ScanFlag), right | left | pair
    {
      auto sf = [&](int par, int ch) {
        const int pl = attr(A.st_l, par), pr = attr(A.st_r, par), cl = attr(A.st_l, ch), cr = attr(A.st_r, ch), M = A.M;
        return ((pl == 0 && cl == 1) ? SF_SL : 0) | ((cr == 0 && pr == 1) ? SF_SR : 0) | ((cl != 0 && cl != M - 1) ? SF_IL : 0) |
               ((pr != 0 && pr != M - 1) ? SF_IR : 0) | ((pl == M - 2 && cl == M - 1) ? SF_EL : 0) |
               ((cr == M - 2 && pr == M - 1) ? SF_ER : 0) | ((pr == M - 2) ? SF_PM2 : 0);
      };
      for (const Csr* c : {&right, &left, &pair})
        for (int k = 0; k < ST; ++k)
          for (size_t e = 0; e + 1 < c->rows[k].size(); e += 2) scan_fl.push_back(sf(k, c->rows[k][e]));
    }
    // pair records of the factorised rule 2 (device_layout.h)
    for (int k = 0; k < A.n_ap && ok; ++k) {
      const int s1 = ap_keep[k][0], t = ap_keep[k][1], tgt = ap_keep[k][2];
      const int nch = (int)chain_keep.rows[k].size() / 2, nrch = (int)rchain_keep.rows[k].size() / 2;
      if (nch > kFastR || nrch > kFastR || A.n_ap > 254 || ST > 254) { ok = false; break; }
      int32_t w[8] = {0};
      // (target, s1 and t as live indices: they address the heavy sums)
      w[0] = colof(ST_1, s1) | (colof(ST_P, t) << 8) | (((tgt >= 0 && li[tgt] >= 0) ? li[tgt] : 0xff) << 16) |
             (((attr(A.st_lam, t) ? 1 : 0) | (attr(A.st_w_r, t) ? 2 : 0) | (t == A.shadow ? 4 : 0)) << 24);
      if (li[s1] < 0 || li[t] < 0) { ok = false; break; }
      w[1] = li[s1] | (li[t] << 8) | (nch << 16) | (nrch << 20);
      for (int u = 0; u < nch; ++u) {
        const int pc = chain_keep.rows[k][2 * u];
        w[2 + u] = pc | (fwd_id(right, br, t, ap_keep[pc][1]) << 8);
      }
      for (int u = 0; u < nrch; ++u) {
        const int pp = rchain_keep.rows[k][2 * u];
        w[5 + u] = pp | (fwd_id(right, br, ap_keep[pp][1], t) << 8);
      }
      pair_rec.insert(pair_rec.end(), w, w + 8);
    }
    A.fp_ok = ok ? 1 : 0;
    A.lin_wr = 11 + A.n_theta;            // (kLinEth + n_theta, lin_params.h)
    A.lin_wl = A.lin_wr + 5 * A.n_wr;
    A.lin_wp = A.lin_wl + 5 * A.n_wl;
    A.lin_total = A.lin_wp + 8 * A.n_wp;
  }
  A.n_small = (int32_t)ints->size();
  // big part: tuple lists of the bifurcation and interior-loop rules -- first the ones the inside direction reads (by
  // parent), then the ones of the outside direction (by child): a kernel stages the small part and its own run
  put(split, &A.split_off, &A.split_ent);
  put(quad, &A.quad_off, &A.quad_ent);
  A.split_tgt = split.emit_targets(ints);
  A.quad_tgt = quad.emit_targets(ints);
  // column records of the interior-loop tuples (device_layout.h: qc_*)
  auto col_of = [&](int e, int k) { return (*ints)[A.tab_cmap + e * ST + k]; };
  auto emit_qc = [&](const Csr& c, int which) {
    const int32_t pos = (int32_t)ints->size();
    for (int k = 0; k < ST; ++k)
      for (size_t e = 0; e + 2 < c.rows[k].size(); e += 3) {
        const int a = c.rows[k][e], b = c.rows[k][e + 1], d3 = c.rows[k][e + 2];
        int c0, c1, c2, ca, par;
        if (which == 0) { c0 = col_of(ST_P, a); c1 = col_of(ST_L, b); c2 = col_of(ST_L, d3); ca = 0; par = k; }
        else if (which == 1) { c0 = col_of(ST_E, a); c1 = col_of(ST_L, b); c2 = col_of(ST_L, d3); ca = col_of(ST_P, k); par = a; }
        else { c0 = col_of(ST_E, a); c1 = col_of(ST_P, b); c2 = col_of(ST_L, d3); ca = col_of(ST_L, k); par = a; }
        const bool dead = c0 < 0 || c1 < 0 || c2 < 0 || ca < 0 || c0 > 254 || c1 > 254 || c2 > 254 || ca > 254;
        const int fl = ((*ints)[A.st_lam + par] ? 1 : 0) | (par == A.shadow ? 2 : 0) | (dead ? 4 : 0);
        ints->push_back(dead ? 0 : (c0 | (c1 << 8) | (c2 << 16) | (ca << 24)));
        ints->push_back(k | (fl << 16));
      }
    return pos;
  };
  A.qc_in = emit_qc(quad, 0);
  // deterministic mode: the tuples of a column-record list dealt to four waves by target (device_layout.h: qd_*)
  auto emit_det = [&](const std::vector<int32_t>& q, size_t first, int n, std::vector<int32_t>* out) {
    std::vector<int32_t> ids[4];
    for (int t = 0; t < n; ++t) {
      const int w = q[2 * (first + t) + 1];
      if ((w >> 16) & 4) continue;           // dead tuple
      ids[(w & 0xffff) & 3].push_back(t);
    }
    int off = 0;
    for (int k = 0; k < 4; ++k) { out->push_back(off); off += (int)ids[k].size(); }
    out->push_back(off);
    const size_t at = out->size();
    for (int k = 0; k < 4; ++k) out->insert(out->end(), ids[k].begin(), ids[k].end());
    out->resize(at + n, 0);
  };
  {
    const std::vector<int32_t> q(ints->begin() + A.qc_in, ints->begin() + A.qc_in + 2 * quad.count());
    A.qd_in = (int32_t)ints->size();
    emit_det(q, 0, quad.count(), ints);
  }
  A.big_in_end = (int32_t)ints->size();
  put(split1, &A.split1_off, &A.split1_ent);
  put(split2, &A.split2_off, &A.split2_ent);
  put(quad1, &A.quad1_off, &A.quad1_ent);
  put(quad2, &A.quad2_off, &A.quad2_ent);
  put(quad3, &A.quad3_off, &A.quad3_ent);
  A.split1_tgt = split1.emit_targets(ints);
  A.split2_tgt = split2.emit_targets(ints);
  A.quad1_tgt = quad1.emit_targets(ints);
  A.quad2_tgt = quad2.emit_targets(ints);
  A.quad3_tgt = quad3.emit_targets(ints);
  A.qc_out1 = emit_qc(quad1, 1);
  A.qc_out2 = emit_qc(quad2, 2);
  A.qc_out3 = emit_qc(quad3, 2);
  A.n_split = split.count();
  A.n_quad = quad.count();
  {
    const std::vector<int32_t> q(ints->begin() + A.qc_out1, ints->begin() + A.qc_out1 + 6 * A.n_quad);
    A.qd_out = (int32_t)ints->size();
    for (int role = 0; role < 3; ++role) emit_det(q, (size_t)role * A.n_quad, A.n_quad, ints);
  }
  A.n_ints = (int32_t)ints->size();
  // fast blobs of the table-driven train kernels (behind n_ints: the generic kernels never stage them)
  auto append = [&](const std::vector<int32_t>& v) { const int32_t pos = (int32_t)ints->size(); ints->insert(ints->end(), v.begin(), v.end()); return pos; };
  auto copy_of = [&](int32_t from, int n) { return std::vector<int32_t>(ints->begin() + from, ints->begin() + from + n); };
  std::vector<int32_t> qci = copy_of(A.qc_in, 2 * A.n_quad), qco = copy_of(A.qc_out1, 6 * A.n_quad);
  for (std::vector<int32_t>* q : {&qci, &qco})   // (the fast copies name the target by its live index; dead without one)
    for (size_t r = 1; r < q->size(); r += 2) {
      const int k = (*q)[r] & 0xffff, fl = (*q)[r] >> 16;
      (*q)[r] = (li[k] >= 0 ? li[k] : 0) | ((fl | (li[k] >= 0 ? 0 : 4)) << 16);
    }
  A.n_lane = (int32_t)live_states.size();
  A.fb_in = (int32_t)ints->size();
  A.f_live_in = append(live_states);
  A.fp_in = append(prog_in);
  A.fqc_in = append(qci);
  { std::vector<int32_t> dl; emit_det(qci, 0, A.n_quad, &dl); A.fqd_in = append(dl); }
  A.fpr_in = append(pair_rec);
  A.fs_in = append(scan_fl);
  A.fb_in_n = (int32_t)ints->size() - A.fb_in;
  A.fb_out = (int32_t)ints->size();
  A.f_live_out = append(live_states);
  A.fp_out = append(prog_out);
  A.fe_r = append(attr_r);
  A.fe_p = append(attr_p);
  A.fqc_out = append(qco);
  { std::vector<int32_t> dl; for (int role = 0; role < 3; ++role) emit_det(qco, (size_t)role * A.n_quad, A.n_quad, &dl); A.fqd_out = append(dl); }
  A.fpr_out = append(pair_rec);
  A.fs_out = append(scan_fl);
  A.fb_out_n = (int32_t)ints->size() - A.fb_out;
}

void flatten_trivial(AutomatonLayout* lay, std::vector<int32_t>* ints) {
  AutomatonLayout& A = *lay;
  ints->clear();
  A.S = 1; A.n_active = 1; A.n_front = 1; A.shadow = -1; A.M = 1; A.n_theta = 0; A.n_rows = 0;
  A.s00 = A.s0m1 = A.s0m2 = 0;
  auto one = [&](int32_t v) { int32_t p = (int32_t)ints->size(); ints->push_back(v); return p; };
  A.st_l = one(0); A.st_r = one(0); A.st_is_loop = one(1);
  A.st_row_r = one(-1); A.st_row_l = one(-1); A.st_pair_r = one(0);
  A.st_w_r = one(0); A.st_w_l = one(0); A.st_lam = one(0); A.st_ref = one(0);
  A.row_off = one(0);
  auto csr = [&](int width, int32_t* off, int32_t* ent) {
    *off = (int32_t)ints->size();
    ints->push_back(0); ints->push_back(1);
    *ent = (int32_t)ints->size();
    for (int k = 0; k < width; ++k) ints->push_back(0);  // (state 0, flag 0) / (0,0) / (0,0,0)
  };
  csr(2, &A.right_off, &A.right_ent); csr(2, &A.left_off, &A.left_ent); csr(2, &A.pair_off, &A.pair_ent);
  csr(2, &A.rright_off, &A.rright_ent); csr(2, &A.rleft_off, &A.rleft_ent); csr(2, &A.rpair_off, &A.rpair_ent);
  A.n_ap = 1;                                          // the pair ((0,0), (0,0)), target (0,0), its own chain predecessor
  A.ap_s1 = one(0); A.ap_t = one(0); A.ap_tgt = one(0);
  csr(2, &A.ap_chain_off, &A.ap_chain_ent); csr(2, &A.ap_rchain_off, &A.ap_rchain_ent);
  csr(1, &A.ap_by_s1_off, &A.ap_by_s1_ent); csr(1, &A.ap_by_t_off, &A.ap_by_t_ent);
  A.tab_cmap = (int32_t)ints->size();
  for (int e = 0; e < 7; ++e) { ints->push_back(0); A.tab_rs[e] = 1; A.tab_cs[e] = e; }
  A.tab_row = 7; A.ap_rs = 1; A.tab_cell = 0;
  A.fp_ok = 0; A.fp_in = A.fp_out = A.fe_r = A.fe_p = 0; A.n_wr = A.n_wp = A.n_wl = 0;
  A.fb_in = A.fb_in_n = A.fb_out = A.fb_out_n = A.fqc_in = A.fpr_in = A.fqc_out = A.fpr_out = A.fs_in = A.fs_out = 0;
  A.qd_in = A.qd_out = A.fqd_in = A.fqd_out = 0;
  A.st_live = one(0); A.st_li = one(0);
  A.fp_max_p = kFastP; A.n_lane = 1; A.f_live_in = A.f_live_out = 0;
  A.lin_wr = A.lin_wl = A.lin_wp = A.lin_total = 11;
  A.qc_in = A.qc_out1 = A.qc_out2 = A.qc_out3 = 0;
  A.n_small = (int32_t)ints->size();
  csr(2, &A.split_off, &A.split_ent); csr(3, &A.quad_off, &A.quad_ent);
  A.split_tgt = A.quad_tgt = one(0);
  A.big_in_end = (int32_t)ints->size();
  csr(2, &A.split1_off, &A.split1_ent); csr(2, &A.split2_off, &A.split2_ent);
  csr(3, &A.quad1_off, &A.quad1_ent); csr(3, &A.quad2_off, &A.quad2_ent); csr(3, &A.quad3_off, &A.quad3_ent);
  A.split1_tgt = A.split2_tgt = A.quad1_tgt = A.quad2_tgt = A.quad3_tgt = one(0);
  A.n_split = 1; A.n_quad = 1;
  A.n_ints = (int32_t)ints->size();
}

}  // namespace elemdp
