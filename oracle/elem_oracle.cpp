// elem_oracle.cpp -- CPU oracle for the RNAelem inside/outside/CYK hot path.
//
// TEST INFRASTRUCTURE ONLY (see elem_oracle.h).  Parity status: PINNED against the reference's
// known-answer tests and against golden vectors produced by the compiled reference
// (tests/golden/, generator tests/golden/make_golden.py).
//
// This is a restatement, in our own flat-array C++, of the reference's CPU algorithm.  It keeps
// the reference's loop order and scatter-style outside pass on purpose, so that single-threaded
// results agree with the reference to rounding.  Map (reference paths relative to /root/reference):
//   log-semiring                         RNAelem/util.hpp:191-229
//   base / pair encoding                 RNAelem/bio_sequence.hpp:17-39
//   EnergyTables::parse                  RNAelem/energy_param.hpp:159-183, 399-640
//   hairpin / loop / sum_ext_m           RNAelem/energy_param.hpp:686-795
//   Hmm::build                           RNAelem/profile_hmm.hpp:188-463
//   Seq::prepare (BPP filter)            RNAelem/energy_model.hpp:188-276
//   parsable()                           RNAelem/energy_model.hpp:289-338
//   sweep_inside / sweep_outside         RNAelem/energy_model.hpp:340-547
//   PlainInside / PlainOutside           RNAelem/energy_model.hpp:559-661
//   MotifInside / MotifOutside           RNAelem/motif_model.hpp:230-613 (+ no-rss :170-206)
//   TrainIn / TrainOut                   RNAelem/motif_trainer.hpp:274-458
//   train_seq schedule, gradient         RNAelem/motif_trainer.hpp:89-116, 204-271
//   ScanOut / EndIn / EndOut / Cyk       RNAelem/motif_scanner.hpp:364-913
//   scan_seq schedule, traceback         RNAelem/motif_scanner.hpp:172-362
#include "elem_oracle.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

using std::string;
using std::vector;
typedef vector<double> V;
typedef vector<V> VV;
const double NINF = -std::numeric_limits<double>::infinity();
const int BIG = std::numeric_limits<int>::max();

[[noreturn]] void die(const string& m) { throw std::runtime_error(m); }

// ---------------------------------------------------------------- log semiring (util.hpp:195-229)
inline double lse2(double x, double y) {
  return (NINF == y) ? x : (NINF == x) ? y : x < y ? y + log1p(exp(x - y)) : x + log1p(exp(y - x));
}
inline void addL(double& x, double y) { x = lse2(x, y); }
inline double mul3(double a, double b, double c) { return a + (b + c); }
inline double mul4(double a, double b, double c, double d) { return a + (b + (c + d)); }
inline double mul5(double a, double b, double c, double d, double e) { return a + (b + (c + (d + e))); }

// ---------------------------------------------------------------- bases (bio_sequence.hpp:20-26)
const int BP[5][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 5}, {0, 0, 0, 1, 0}, {0, 0, 2, 0, 3}, {0, 6, 0, 4, 0}};
const char NACGU[] = "NACGU";

// structural states / transitions (energy_model.hpp:58-91)
enum { ST_P = 0, ST_E, ST_M, ST_B, ST_1, ST_2, ST_L, ST_O, NST };
enum { TT_E_H = 0, TT_P_E, TT_P_P, TT_O_O, TT_O_OP, TT_E_P, TT_E_M, TT_M_M, TT_M_B, TT_B_12, TT_1_B,
       TT_1_2, TT_2_2, TT_2_P, TT_L_L, NTT };
const int TT_PARENT[NTT] = {ST_E, ST_P, ST_P, ST_O, ST_O, ST_E, ST_E, ST_M, ST_M, ST_B, ST_1, ST_1, ST_2, ST_2, ST_L};
const int TT_CHILD[NTT] = {ST_L, ST_E, ST_P, ST_O, ST_P, ST_P, ST_M, ST_M, ST_B, ST_1, ST_B, ST_2, ST_2, ST_P, ST_L};
int states_to_trans(int e, int e1) {
  for (int t = 0; t < NTT; ++t) if (TT_PARENT[t] == e && TT_CHILD[t] == e1) return t;
  return -1;
}

// ================================================================= energy tables
struct EnergyTables {
  static constexpr int maxloop = 30;
  static constexpr double kT = (37 + 273.15) * 1.98717;
  double hairpin[31], bulge[31], internal_[31], ninio[31];
  double mm_h[7][5][5], mm_i[7][5][5], mm_m[7][5][5], mm_1ni[7][5][5], mm_23i[7][5][5], mm_ext[7][5][5];
  double stack[7][7];
  double int11[8][8][5][5];
  double int21[8][8][5][5][5];
  double int22[8][8][5][5][5][5];
  double dangle5[8][5], dangle3[8][5];
  double tri[40], tetra[40], hexa[40];
  string tris, tetras, hexas;
  double term_au, mlintern, mlclosing, ml_base, lxc37;

  static double smooth(int a) { /* energy_param.hpp:94-106 */
    double z = double(a);
    if (z / 10. < -1.2283697) return 0.;
    else if (0.8660254 < z / 10.) return z;
    else return 10. * 0.38490018 * (1. + sin(z / 10. - 0.34242663)) * (1. + sin(z / 10. - 0.34242663));
  }
  static double log_energy(int z, bool smo = false) { /* :108-114 */
    if (smo) return smooth(-z) * 10. / kT;
    return -z * 10. / kT;
  }

  // --- text reader with the semantics of get_array (:159-183)
  vector<string> lines;
  size_t pos = 0;
  bool getline_(string& s) { if (pos >= lines.size()) return false; s = lines[pos++]; return true; }
  static vector<string> words_of(const string& s) {
    vector<string> w; std::istringstream iss(s); for (string t; iss >> t;) w.push_back(t); return w;
  }
  void get_array(double* a, int size, bool smo = false) {
    for (int i = 0; i < size;) {
      string str;
      if (!getline_(str)) break;
      else if (str.length() < 2) break;
      vector<string> w = words_of(str);
      int prev = i;
      for (; i < size && i - prev < (int)w.size(); ++i) {
        const string& t = w[i - prev];
        if (t.find("/*") != string::npos) break;
        if (t == "INF") a[i] = NINF;
        else if (t == "DEF") a[i] = log_energy(-50, smo);
        else a[i] = log_energy(atoi(t.c_str()), smo);
      }
    }
  }
  static void fill(double* a, int n) { for (int i = 0; i < n; ++i) a[i] = NINF; }

  void read_triple(const char* what, int idx[3], double* dst[3]) {
    string str;
    while (getline_(str)) {
      if (str == "") break;
      if (str.find("*") != string::npos) continue;
      vector<string> w = words_of(str);
      for (int k = 0; k < 3; ++k) {
        if (!dst[k]) continue;
        if ((int)w.size() <= idx[k]) die(string("bad line in ") + what);
        *dst[k] = log_energy(atoi(w[idx[k]].c_str()));
      }
      break;
    }
  }
  void read_loops(double* arr, string& names) { /* read_string :458-476 */
    string str;
    for (int i = 0; getline_(str); ++i) {
      if (str == "") break;
      if (str.find("*") != string::npos) { --i; continue; }
      vector<string> w = words_of(str);
      if (w.size() < 2) die("bad special-loop line");
      names += w[0] + " ";
      if (i < 40) arr[i] = log_energy(atoi(w[1].c_str()));
    }
  }

  void parse(const string& text) {
    lines.clear();
    {
      std::istringstream iss(text);
      for (string l; std::getline(iss, l);) {
        if (!l.empty() && l.back() == '\r') l.pop_back();
        lines.push_back(l);
      }
    }
    lxc37 = 107.856;
    term_au = mlintern = mlclosing = ml_base = 0;
    fill(hairpin, 31); fill(bulge, 31); fill(internal_, 31); fill(ninio, 31);
    fill(&mm_h[0][0][0], 175); fill(&mm_i[0][0][0], 175); fill(&mm_m[0][0][0], 175);
    fill(&mm_1ni[0][0][0], 175); fill(&mm_23i[0][0][0], 175); fill(&mm_ext[0][0][0], 175);
    fill(&stack[0][0], 49); fill(&int11[0][0][0][0], 1600); fill(&int21[0][0][0][0][0], 8000);
    fill(&int22[0][0][0][0][0][0], 40000); /* reference leaves part of this uninitialised */
    fill(&dangle5[0][0], 40); fill(&dangle3[0][0], 40);
    fill(tri, 40); fill(tetra, 40); fill(hexa, 40);
    tris = tetras = hexas = "";
    /* first pass: only LXC from Misc (read_only_misc :478-492) */
    for (size_t p = 0; p < lines.size(); ++p) {
      vector<string> w = words_of(lines[p]);
      if (lines[p].size() && lines[p][0] == '#' && w.size() > 1 && w[1] == "Misc") {
        for (size_t q = p + 1; q < lines.size(); ++q) {
          if (lines[q] == "") break;
          if (lines[q].find("*") != string::npos) continue;
          vector<string> ww = words_of(lines[q]);
          if (ww.size() > 4) lxc37 = atof(ww[4].c_str());
        }
        break;
      }
    }
    pos = 0;
    string str;
    while (getline_(str)) {
      if (str.empty() || str[0] != '#') continue;
      vector<string> w = words_of(str);
      if (w.size() <= 1) continue;
      const string& id = w[1];
      if (id == "stack") {
        for (int i = 1; i < 7; ++i) get_array(&stack[i][1], 6);
      } else if (id == "mismatch_hairpin") { for (int i = 1; i < 7; ++i) get_array(&mm_h[i][0][0], 25);
      } else if (id == "mismatch_interior") { for (int i = 1; i < 7; ++i) get_array(&mm_i[i][0][0], 25);
      } else if (id == "mismatch_interior_1n") { for (int i = 1; i < 7; ++i) get_array(&mm_1ni[i][0][0], 25);
      } else if (id == "mismatch_interior_23") { for (int i = 1; i < 7; ++i) get_array(&mm_23i[i][0][0], 25);
      } else if (id == "mismatch_multi") {
        double dump[25]; /* the reference reads a 7th (NS) block out of bounds; unused for real pairs */
        for (int i = 1; i < 8; ++i) get_array(i < 7 ? &mm_m[i][0][0] : dump, 25, true);
      } else if (id == "mismatch_exterior") {
        double dump[25];
        for (int i = 1; i < 8; ++i) get_array(i < 7 ? &mm_ext[i][0][0] : dump, 25, true);
      } else if (id == "dangle5") { for (int i = 1; i < 8; ++i) get_array(&dangle5[i][0], 5, true);
      } else if (id == "dangle3") { for (int i = 1; i < 8; ++i) get_array(&dangle3[i][0], 5, true);
      } else if (id == "int11") {
        for (int i = 1; i < 8; ++i) for (int j = 1; j < 8; ++j) get_array(&int11[i][j][0][0], 25);
      } else if (id == "int21") {
        for (int i = 1; i < 8; ++i) for (int j = 1; j < 8; ++j) get_array(&int21[i][j][0][0][0], 125);
      } else if (id == "int22") {
        for (int i = 1; i < 7; ++i) for (int j = 1; j < 7; ++j)
          for (int k = 1; k < 5; ++k) for (int l = 1; l < 5; ++l) for (int m = 1; m < 5; ++m)
            get_array(&int22[i][j][k][l][m][1], 4);
      } else if (id == "hairpin") { get_array(hairpin, 31);
      } else if (id == "bulge") { get_array(bulge, 31);
      } else if (id == "interior") { get_array(internal_, 31);
      } else if (id == "NINIO") { /* read_ninio :399-420 */
        string s2;
        while (getline_(s2)) {
          if (s2 == "") break;
          if (s2.find("*") != string::npos) continue;
          vector<string> ww = words_of(s2);
          if (ww.size() <= 2) die("read_ninio");
          int f = atoi(ww[0].c_str()), mx = atoi(ww[2].c_str());
          for (int i = 0; i <= maxloop; ++i) ninio[i] = log_energy(std::min(mx, i * f));
          break;
        }
      } else if (id == "ML_params") {
        int idx[3] = {0, 2, 4}; double* dst[3] = {&ml_base, &mlclosing, &mlintern};
        read_triple("ML_params", idx, dst);
      } else if (id == "Misc") { /* read_misc(false) :444-456 : every data line until blank */
        string s2;
        while (getline_(s2)) {
          if (s2 == "") break;
          if (s2.find("*") != string::npos) continue;
          vector<string> ww = words_of(s2);
          if (ww.size() <= 2) die("read_misc");
          term_au = log_energy(atoi(ww[2].c_str()));
        }
      } else if (id == "Triloops") { fill(tri, 40); read_loops(tri, tris);
      } else if (id == "Tetraloops") { fill(tetra, 40); read_loops(tetra, tetras);
      } else if (id == "Hexaloops") { fill(hexa, 40); read_loops(hexa, hexas);
      }
    }
  }

  static bool is_au(int type) { return 2 < type; }

  double sum_ext_m(int i, int j, bool ext, const vector<int>& s) const { /* :686-708 */
    int type = BP[s[i]][s[j]];
    int n = (int)s.size();
    int five = 0 <= i - 1 ? s[i - 1] : -1;
    int three = j + 1 < n ? s[j + 1] : -1;
    double z = 0;
    if (0 <= i - 1 && j + 1 < n) {
      z = z + (ext ? mm_ext[type][five][three] : mm_m[type][five][three]);
      if (is_au(type)) z = z + term_au;
    } else {
      if (0 <= i - 1) z = z + dangle5[type][five];
      if (j + 1 < n) z = z + dangle3[type][three];
      if (is_au(type)) z = z + term_au;
    }
    return z;
  }

  static string slice(const vector<int>& s, int i, int j) {
    string t; for (int k = i; k < j; ++k) t += NACGU[s[k]]; return t;
  }

  double hairpin_energy(int i, int j, const vector<int>& s) const { /* :710-742 */
    int d = j - i - 1;
    if (d < 1) return NINF;
    int type = BP[s[i]][s[j]];
    double z = (d <= maxloop) ? hairpin[d]
                              : hairpin[maxloop] - (lxc37 * log(double(d) * (1. / maxloop)) * 10. * (1. / kT));
    if (d < 3) {
    } else if (3 == d) {
      size_t tel = tris.find(slice(s, i, j + 1));
      if (tel != string::npos) return tri[tel / 6];
      else if (is_au(type)) z = z + term_au;
    } else if (4 == d) {
      size_t tel = tetras.find(slice(s, i, j + 1));
      if (tel != string::npos) {
        if (7 != type) return tetra[tel / 7];
        else z = z + tetra[tel / 7];
      }
    } else if (6 == d) {
      size_t tel = hexas.find(slice(s, i, j + 1));
      if (tel != string::npos) return hexa[tel / 9];
    }
    if (3 < d) z = z + mm_h[type][s[i + 1]][s[j - 1]];
    return z;
  }

  double loop_energy(int i, int j, int p, int q, const vector<int>& s) const { /* :744-795 */
    int type = BP[s[i]][s[j]];
    int type2 = BP[s[q]][s[p]];
    int u1 = p - i - 1, u2 = j - q - 1;
    int u = std::max(u1, u2);
    double z;
    if (u1 < 0 || u2 < 0 || maxloop < u1 + u2) {
      z = NINF;
    } else if (0 == u1 && 0 == u2) {
      z = stack[type][type2];
    } else if (0 == u1 || 0 == u2) {
      z = bulge[u];
      if (1 == u) z = z + stack[type][type2];
      else {
        if (is_au(type)) z = z + term_au;
        if (is_au(type2)) z = z + term_au;
      }
    } else if (u <= 2) {
      if (2 == u1 + u2) z = int11[type][type2][s[i + 1]][s[j - 1]];
      else if (1 == u1 && 2 == u2) z = int21[type][type2][s[i + 1]][s[q + 1]][s[j - 1]];
      else if (2 == u1 && 1 == u2) z = int21[type2][type][s[q + 1]][s[i + 1]][s[p - 1]];
      else z = int22[type][type2][s[i + 1]][s[p - 1]][s[q + 1]][s[j - 1]];
    } else {
      z = internal_[u1 + u2] + ninio[std::abs(u1 - u2)];
      if (1 == u1 || 1 == u2)
        z = mul3(z, mm_1ni[type][s[i + 1]][s[j - 1]], mm_1ni[type2][s[q + 1]][s[p - 1]]);
      else if (5 == u1 + u2)
        z = mul3(z, mm_23i[type][s[i + 1]][s[j - 1]], mm_23i[type2][s[q + 1]][s[p - 1]]);
      else
        z = mul3(z, mm_i[type][s[i + 1]][s[j - 1]], mm_i[type2][s[q + 1]][s[p - 1]]);
    }
    return z;
  }
};

// ================================================================= pattern automaton
struct IS { int id, l, r; };

struct Hmm {
  int M = 0;
  string pattern, reg;
  vector<int> node, pair, theta_id;
  vector<vector<int>> edge_to, edge_from;
  vector<vector<char>> reach, reach_loop;
  vector<IS> state;
  vector<vector<int>> n2s_;
  vector<int> loop_state;
  vector<vector<int>> right, left, pairt;
  vector<std::array<int, 4>> lls;
  VV theta, s;

  const IS& st(int id) const { return state[id]; }
  int n2s(int h, int h1) const { return n2s_[h][h1]; }
  int S() const { return (int)state.size(); }

  void calc_theta() { /* :103-111 */
    theta.clear();
    for (size_t i = 0; i < s.size(); ++i) {
      theta.push_back(V(s[i].size(), -1));
      double tot = NINF;
      for (double e : s[i]) addL(tot, e);
      for (size_t j = 0; j < s[i].size(); ++j) theta[i][j] = s[i][j] - tot;
    }
  }

  static void closure(vector<vector<char>>& m) {
    int n = (int)m.size();
    for (int k = 0; k < n; ++k) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
      if (m[i][k] && m[k][j]) m[i][j] = 1;
  }

  void build(const string& str) { /* :206-226 */
    if (str.empty()) die("empty motif");
    pattern = str;
    /* set_reg_pattern :188-204 : collapse runs of '*', trim leading/trailing '*' */
    reg = str;
    reg.erase(std::unique(reg.begin(), reg.end(), [](char a, char b) { return a == '*' && b == '*'; }), reg.end());
    reg.erase(0, reg.find_first_not_of('*'));
    size_t last = reg.find_last_not_of('*');
    if (last != string::npos) reg.erase(last + 1);
    node.clear();
    node.push_back('z');
    for (char c : reg) node.push_back(c);
    node.push_back('o');
    M = (int)node.size();
    /* set_pair */
    pair.assign(M, -1);
    {
      vector<int> stk;
      for (int h = 0; h < M; ++h) {
        if (node[h] == '(') stk.push_back(h);
        else if (node[h] == ')') {
          if (stk.empty()) die("unmatched brackets");
          int hl = stk.back(); stk.pop_back();
          pair[hl] = h; pair[h] = hl;
        }
      }
      if (!stk.empty()) die("unmatched brackets");
    }
    /* set_edge :257-283 */
    edge_to.assign(M, vector<int>());
    edge_from.assign(M, vector<int>());
    for (int h = 0; h < M; ++h) {
      int c = node[h];
      if (0 < h) {
        if ('*' == node[h - 1]) { edge_to[h].push_back(h - 2); edge_from[h - 2].push_back(h); }
        edge_to[h].push_back(h - 1); edge_from[h - 1].push_back(h);
      }
      if ('<' != c && '>' != c) { edge_to[h].push_back(h); edge_from[h].push_back(h); }
    }
    /* set_s_theta :286-313 */
    theta_id.assign(M, -1);
    s.assign(1, V(4, 0));
    for (int h = 0; h < M; ++h) {
      switch (node[h]) {
        case ')': theta_id[h] = (int)s.size(); s.push_back(V(6, 0)); break;
        case '.': theta_id[h] = (int)s.size(); s.push_back(V(4, 0)); break;
        case '*': case 'z': case 'o': theta_id[h] = 0; break;
        case '<': case '>': case '(': break;
        default: die(string("bad motif char: ") + char(node[h]));
      }
    }
    calc_theta();
    /* set_reachable :316-354 */
    reach.assign(M, vector<char>(M, 0));
    reach_loop.assign(M, vector<char>(M, 0));
    for (int h = 0; h < M; ++h) {
      int c = node[h];
      if (c == ')') { for (int h1 : edge_to[pair[h]]) reach[h1][h] = 1; }
      else if (c == '(') {}
      else if (c == '>') { for (int h1 : edge_to[pair[h]]) { reach[h1][h] = 1; reach_loop[h1][h] = 1; } }
      else if (c == '<') {}
      else { for (int h1 : edge_to[h]) { reach[h1][h] = 1; reach_loop[h1][h] = 1; } }
      reach[h][h] = 1; reach_loop[h][h] = 1;
    }
    closure(reach);
    closure(reach_loop);
    /* set_interval_state :369-384 */
    state.clear();
    for (int hr = 0; hr < M; ++hr) for (int hl = hr; 0 <= hl; --hl)
      if (reach[hl][hr]) state.push_back(IS{(int)state.size(), hl, hr});
    n2s_.assign(M, vector<int>(M, -1));
    for (auto const& x : state) n2s_[x.l][x.r] = x.id;
    loop_state.clear();
    for (auto const& x : state) if (reach_loop[x.l][x.r]) loop_state.push_back(x.id);
    /* set_interval_state_trans :387-449 */
    int S_ = (int)state.size();
    right.assign(S_, vector<int>());
    for (auto const& x : state) {
      int cr = node[x.r];
      if (cr == 'z' || cr == '.' || cr == '*' || cr == 'o')
        for (int h : edge_to[x.r]) if (x.l <= h && reach[x.l][h]) right[x.id].push_back(n2s(x.l, h));
    }
    left.assign(S_, vector<int>());
    for (auto const& x : state) {
      int cl = node[x.l];
      if (cl == 'z' || cl == '.' || cl == '*' || cl == 'o')
        for (int h : edge_to[x.l]) if (h <= x.r && reach[h][x.r]) left[n2s(h, x.r)].push_back(x.id);
    }
    pairt.assign(S_, vector<int>());
    for (int hr = 0; hr < M; ++hr) if (')' == node[hr]) {
      int kl = pair[hr];
      for (int hl : edge_to[kl]) {
        int sid = n2s(hl, hr);
        for (int kr : edge_to[hr]) if (reach[kl][kr]) pairt[sid].push_back(n2s(kl, kr));
      }
    }
    for (auto const& x : state) {
      if ('z' == node[x.r] || 'o' == node[x.r] || '*' == node[x.r])
        for (int hl : edge_from[x.l]) if ('z' == node[hl] || 'o' == node[hl] || '*' == node[hl])
          for (int hr : edge_to[x.r]) if (reach[hl][hr]) pairt[x.id].push_back(n2s(hl, hr));
    }
    /* set_states_pairs :451-463 */
    lls.clear();
    for (int a : loop_state) for (int b : loop_state) {
      const IS& s2 = state[a]; const IS& s3 = state[b];
      if (s3.r < s2.l || !reach[s2.r][s3.l] || !reach[s2.l][s3.r]) continue;
      lls.push_back({n2s(s2.l, s3.r), n2s(s2.r, s3.l), a, b});
    }
  }
};

// ================================================================= model
struct Model {
  Hmm mm;
  EnergyTables ep;
  int flags = 0;
  int max_pair = BIG, max_iloop = BIG;
  double min_bpp = 0, min_lnbpp = NINF;
  double tau = 1, log_tau = 0;
  double lambda[2] = {1., 1.};
  bool no_rss() const { return flags & ORC_NO_RSS; }
  bool no_prf() const { return flags & ORC_NO_PRF; }
  bool no_ene() const { return flags & ORC_NO_ENE; }
  bool softmax() const { return flags & ORC_THETA_SOFTMAX; }
  bool lik_ratio() const { return flags & ORC_LIK_RATIO; }
  bool no_theta() const { return flags & ORC_DBG_NO_THETA; }
  bool fix_rss() const { return flags & ORC_DBG_FIX_RSS; }
  bool no_turn() const { return flags & ORC_DBG_NO_TURN; }
  int S() const { return mm.S(); }
  int M() const { return mm.M; }
  int n_theta() const { int n = 0; for (auto const& r : mm.theta) n += (int)r.size(); return n; }

  double lam(const IS& s) const { return s.l == s.r ? lambda[0] : lambda[1]; } /* motif_model.hpp:117 */

  double theta2(int h, int h1, int i, int j) const { /* profile_hmm.hpp:113-135 */
    if (')' == mm.node[h1]) return (0 == BP[i][j] || no_theta()) ? 0. : mm.theta[mm.theta_id[h1]][BP[i][j] - 1];
    return no_theta() ? 0. : (0 == i ? 0. : mm.theta[mm.theta_id[h]][i - 1]) + (0 == j ? 0. : mm.theta[mm.theta_id[h1]][j - 1]);
  }
  double theta1(int h, int j) const { /* :137-141 */
    return (0 == j || no_theta()) ? 0. : mm.theta[mm.theta_id[h]][j - 1];
  }
  void emit2(VV& e, int h, int h1, int i, int j, double w) const { /* :144-179 */
    if (')' == mm.node[h1]) { if (0 < BP[i][j]) e[mm.theta_id[h1]][BP[i][j] - 1] += w; }
    else {
      if (0 != i) e[mm.theta_id[h]][i - 1] += w;
      if (0 != j) e[mm.theta_id[h1]][j - 1] += w;
    }
  }
  void emit1(VV& e, int h, int j, double w) const { if (0 != j) e[mm.theta_id[h]][j - 1] += w; }
  void clear_counts(VV& e) const { e.assign(mm.theta.size(), V()); for (size_t i = 0; i < e.size(); ++i) e[i].assign(mm.theta[i].size(), 0.); }

  void pack(V& x) const { /* motif_model.hpp:147-157 */
    x.clear();
    for (auto const& r : (softmax() ? mm.s : mm.theta)) x.insert(x.end(), r.begin(), r.end());
    x.push_back(lambda[0]); x.push_back(lambda[1]);
  }
  void unpack(const double* x) { /* :159-168 */
    int i = 0;
    for (auto& r : (softmax() ? mm.s : mm.theta)) for (auto& v : r) v = x[i++];
    if (softmax()) mm.calc_theta();
    lambda[0] = x[i++]; lambda[1] = x[i++];
  }
};

// ================================================================= per-sequence context
struct Seq {
  const Model* m = nullptr;
  vector<int> seq;
  V ws;
  int L = 0, W = 0, C = 0;
  vector<char> bp_ok, left_ok;
  double bpp_eff = 0;
  string fix_s;
  /* plain McCaskill tables for the BPP filter */
  V pin_o, pout_o, pin, pout;

  bool ok(int i, int d) const { return bp_ok[i * (W + 1) + d]; }
  bool lok(int i, int d) const { return left_ok[i * (W + 1) + d]; }
  double& PI(int i, int j, int e) { return pin[(i * (W + 1) + (j - i)) * 7 + e]; }
  double& PO(int i, int j, int e) { return pout[(i * (W + 1) + (j - i)) * 7 + e]; }

  bool parsable(int e, int i, int j) const { /* energy_model.hpp:289-338 */
    switch (e) {
      case ST_P: return 0 <= i && j - i <= W && bp_ok[i * (W + 1) + (j - i)];
      case ST_E: return 0 < i && j - i + 2 <= W && bp_ok[(i - 1) * (W + 1) + (j - i + 2)];
      case ST_M: return 0 < i && j < L && j - i <= W && (m->no_turn() ? 4 <= j - i : 10 <= j - i);
      case ST_B: case ST_1: case ST_2: return j - i <= W && left_ok[i * (W + 1) + (j - i)];
    }
    return false;
  }

  void set_ws(const vector<int>& q) { /* motif_model.hpp:62-70 */
    vector<int> cnt(127 - 33, 0);
    for (int v : q) cnt.at(v) += 1;
    int mode = 0; { int mx = std::numeric_limits<int>::lowest(); for (int i = 0; i < (int)cnt.size(); ++i) if (mx <= cnt[i]) { mode = i; mx = cnt[i]; } }
    ws.clear();
    for (size_t i = 0; i + 1 < q.size(); ++i) ws.push_back(log((0.01 + double(q[i])) / (0.01 + mode)));
    ws.push_back(0 == q.back() ? NINF : 0.);
  }
  double weight(int h, int i) const { /* motif_model.hpp:131-134 */
    int c = m->mm.node[h];
    return ('.' == c || '(' == c || ')' == c) ? ws[i] : 0.;
  }

  void fill_left() { /* energy_model.hpp:203-209 */
    left_ok.assign((L + 1) * (W + 1), 0);
    for (int i = 0; i <= L; ++i) for (int j = i + 1; j <= std::min(L, i + W); ++j)
      if (left_ok[i * (W + 1) + j - i - 1] || bp_ok[i * (W + 1) + j - i]) left_ok[i * (W + 1) + j - i] = 1;
  }

  template <class F> void sweep_inside(F& f);
  template <class F> void sweep_outside(F& f);
  void calc_bpp();
  double lnBPP(int i, int j) { /* :195-201 */
    if (0 <= i && j <= L && j - i <= W && (m->no_turn() ? true : (5 <= j - i)))
      return (PI(i, j, ST_P) + PO(i, j, ST_P)) - pin_o[L];
    return NINF;
  }

  void prepare(const Model& model, const vector<int>& z, const string& fix) { /* set_seq :268-276 + fill_bpp_tables :211-266 */
    m = &model; seq = z; fix_s = fix;
    L = (int)z.size();
    W = std::min(L, model.max_pair);
    C = std::min(W - 2 - (model.no_turn() ? 2 : 5), model.max_iloop);
    if (model.no_rss()) return;
    bp_ok.assign((L + 1) * (W + 1), 0);
    int total = 0, nbp = 0;
    for (int i = 0; i <= L; ++i) for (int j = model.no_turn() ? i + 1 : i + 5; j <= std::min(L, i + W); ++j)
      if ((bp_ok[i * (W + 1) + j - i] = 0 < BP[seq[i]][seq[j - 1]])) ++total;
    if (model.fix_rss()) {
      vector<int> stk;
      if ((int)fix_s.size() != L) die("fix_rss length");
      bp_ok.assign((L + 1) * (W + 1), 0);
      for (int i = 0; i < L; ++i) {
        if (fix_s[i] == '(') stk.push_back(i);
        else if (fix_s[i] == ')') { int j = stk.back(); bp_ok.at(j * (W + 1) + (i + 1 - j)) = 1; ++nbp; stk.pop_back(); }
        else if (fix_s[i] != '.') die("bad rss char");
      }
    } else if (0 == model.min_bpp) {
      nbp = total;
    } else {
      fill_left();
      calc_bpp();
      vector<char> keep((L + 1) * (W + 1), 0);
      for (int i = 0; i <= L; ++i) for (int j = model.no_turn() ? i + 1 : i + 5; j <= std::min(L, i + W); ++j)
        if ((keep[i * (W + 1) + j - i] = model.min_lnbpp <= lnBPP(i, j))) ++nbp;
      bp_ok.swap(keep);
    }
    fill_left();
    bpp_eff = (double)nbp / (double)total;
  }
};

// ---- structural sweeps (energy_model.hpp:340-547)
template <class F> void Seq::sweep_inside(F& f) {
  const EnergyTables& ep = m->ep;
  const bool ne = m->no_ene(), fx = m->fix_rss();
  for (int j = 0; j <= L; ++j) {
    int i0 = std::max(0, j - W);
    f.before(i0, j);
    for (int i = j; i0 <= i; --i) {
      double tsc;
      if (parsable(ST_P, i, j)) {
        if (parsable(ST_E, i + 1, j - 1)) f.on(TT_P_E, i, j, i + 1, j - 1, 0.);
        if (parsable(ST_P, i + 1, j - 1)) {
          tsc = ne ? 0. : ep.loop_energy(i, j - 1, i + 1, j - 2, seq);
          if (NINF != tsc) f.on(TT_P_P, i, j, i + 1, j - 1, tsc);
        }
      }
      if (parsable(ST_B, i, j))
        for (int k = i; k <= j; ++k)
          if (parsable(ST_1, i, k) && parsable(ST_2, k, j)) f.on(TT_B_12, i, j, i, k, 0.);
      if (parsable(ST_2, i, j)) {
        if (parsable(ST_2, i, j - 1)) { if (fx && '.' != fix_s[j - 1]) {} else f.on(TT_2_2, i, j, i, j - 1, 0.); }
        if (parsable(ST_P, i, j)) {
          tsc = ne ? 0. : ep.sum_ext_m(i, j - 1, false, seq) + ep.mlintern;
          if (NINF != tsc) f.on(TT_2_P, i, j, i, j, tsc);
        }
      }
      if (parsable(ST_1, i, j)) {
        if (parsable(ST_2, i, j)) f.on(TT_1_2, i, j, i, j, 0.);
        if (parsable(ST_B, i, j)) f.on(TT_1_B, i, j, i, j, 0.);
      }
      if (parsable(ST_M, i, j)) {
        if (parsable(ST_M, i + 1, j)) { if (fx && '.' != fix_s[i]) {} else f.on(TT_M_M, i, j, i + 1, j, 0.); }
        if (parsable(ST_B, i, j)) f.on(TT_M_B, i, j, i, j, 0.);
      }
      if (parsable(ST_E, i, j)) {
        if (parsable(ST_M, i, j)) {
          tsc = ne ? 0. : mul3(ep.sum_ext_m(j, i - 1, false, seq), ep.mlclosing, ep.mlintern);
          if (NINF != tsc) f.on(TT_E_M, i, j, i, j, tsc);
        }
        tsc = ne ? 0. : ep.hairpin_energy(i - 1, j, seq);
        if (fx && string(j - i, '.') != fix_s.substr(i, j - i)) {}
        else if (NINF != tsc) f.on(TT_E_H, i, j, i, j, tsc);
        for (int l = j; l >= std::max(i, j - C); --l)
          for (int k = i; k <= std::min(l, i + C - (j - l)); ++k) {
            if (i == k && l == j) continue;
            if (parsable(ST_P, k, l)) {
              tsc = ne ? 0. : ep.loop_energy(i - 1, j, k, l - 1, seq);
              if (fx && (string(k - i, '.') != fix_s.substr(i, k - i) || string(j - l, '.') != fix_s.substr(l, j - l))) {}
              else if (NINF != tsc) f.on(TT_E_P, i, j, k, l, tsc);
            }
          }
      }
      if (parsable(ST_P, i, j)) {
        tsc = ne ? 0. : ep.sum_ext_m(i, j - 1, true, seq);
        if (NINF != tsc) f.on(TT_O_OP, 0, j, 0, i, tsc);
      }
      if (i0 == i && 0 < j) { if (fx && '.' != fix_s[j - 1]) {} else f.on(TT_O_O, 0, j, 0, j - 1, 0.); }
    }
    f.after(i0, j);
  }
}

template <class F> void Seq::sweep_outside(F& f) {
  const EnergyTables& ep = m->ep;
  const bool ne = m->no_ene(), fx = m->fix_rss();
  for (int j = L; 0 <= j; --j) {
    int i0 = std::max(0, j - W);
    if (1 <= j) f.before(i0, j - 1);
    for (int i = i0; i <= j; ++i) {
      double tsc;
      if (i0 == i && j < L) { if (fx && '.' != fix_s[j]) {} else f.on(TT_O_O, 0, j, 0, j + 1, 0.); }
      if (parsable(ST_2, i, j)) {
        if (parsable(ST_2, i, j + 1)) { if (fx && '.' != fix_s[j]) {} else f.on(TT_2_2, i, j, i, j + 1, 0.); }
        if (parsable(ST_1, i, j)) f.on(TT_1_2, i, j, i, j, 0.);
      }
      if (parsable(ST_P, i, j)) {
        tsc = ne ? 0. : ep.sum_ext_m(i, j - 1, true, seq);
        if (NINF != tsc) f.on(TT_O_OP, 0, i, 0, j, tsc);
        if (parsable(ST_P, i - 1, j + 1)) {
          tsc = ne ? 0. : ep.loop_energy(i - 1, j, i, j - 1, seq);
          if (NINF != tsc) f.on(TT_P_P, i, j, i - 1, j + 1, tsc);
        }
        if (parsable(ST_2, i, j)) {
          tsc = ne ? 0. : ep.sum_ext_m(i, j - 1, false, seq) + ep.mlintern;
          if (NINF != tsc) f.on(TT_2_P, i, j, i, j, tsc);
        }
      }
      if (parsable(ST_E, i, j)) {
        if (parsable(ST_P, i - 1, j + 1)) f.on(TT_P_E, i, j, i - 1, j + 1, 0.);
        tsc = ne ? 0. : ep.hairpin_energy(i - 1, j, seq);
        if (fx && string(j - i, '.') != fix_s.substr(i, j - i)) {}
        else if (NINF != tsc) f.on(TT_E_H, i, j, i, j, tsc);
        if (parsable(ST_M, i, j)) {
          tsc = ne ? 0. : mul3(ep.sum_ext_m(j, i - 1, false, seq), ep.mlclosing, ep.mlintern);
          if (NINF != tsc) f.on(TT_E_M, i, j, i, j, tsc);
        }
      }
      if (parsable(ST_M, i, j) && parsable(ST_M, i - 1, j)) { if (fx && '.' != fix_s[i - 1]) {} else f.on(TT_M_M, i, j, i - 1, j, 0.); }
      if (parsable(ST_B, i, j)) {
        if (parsable(ST_1, i, j)) f.on(TT_1_B, i, j, i, j, 0.);
        if (parsable(ST_M, i, j)) f.on(TT_M_B, i, j, i, j, 0.);
        for (int k = j; k >= i; --k)
          if (parsable(ST_1, i, k) && parsable(ST_2, k, j)) f.on(TT_B_12, i, k, i, j, 0.);
      }
      if (parsable(ST_E, i, j)) {
        for (int k = i; k <= std::min(j - 2, i + C); ++k)
          /* the reference's bound max(k+2, l-(C-(k-i))) is self-referential in l (:529): since
             C-(k-i) >= 0 inside this loop it never terminates the scan early, i.e. l >= k+2 */
          for (int l = j; l >= k + 2; --l) {
            if (i == k && l == j) continue;
            if (parsable(ST_P, k, l)) {
              tsc = ne ? 0. : ep.loop_energy(i - 1, j, k, l - 1, seq);
              if (fx && (string(k - i, '.') != fix_s.substr(i, k - i) || string(j - l, '.') != fix_s.substr(l, j - l))) {}
              else if (NINF != tsc) f.on(TT_E_P, k, l, i, j, tsc);
            }
          }
      }
    }
    if (1 <= j) f.after(i0, j - 1);
  }
}

// ---- plain McCaskill (energy_model.hpp:559-661)
struct PlainInside {
  Seq& q;
  void before(int, int) {}
  void after(int, int) {}
  void on(int t, int i, int j, int k, int l, double tsc) {
    switch (t) {
      case TT_O_OP: addL(q.pin_o[j], mul3(q.pin_o[l], q.PI(l, j, ST_P), tsc)); break;
      case TT_O_O: addL(q.pin_o[j], q.pin_o[l] + tsc); break;
      case TT_E_H: addL(q.PI(i, j, ST_E), tsc); break;
      case TT_B_12: addL(q.PI(i, j, ST_B), mul3(q.PI(k, l, ST_1), q.PI(l, j, ST_2), tsc)); break;
      default: addL(q.PI(i, j, TT_PARENT[t]), q.PI(k, l, TT_CHILD[t]) + tsc);
    }
  }
};
struct PlainOutside {
  Seq& q;
  void before(int, int) {}
  void after(int, int) {}
  void on(int t, int i, int j, int k, int l, double tsc) {
    switch (t) {
      case TT_O_OP:
        addL(q.pout_o[j], mul3(q.PI(j, l, ST_P), q.pout_o[l], tsc));
        addL(q.PO(j, l, ST_P), mul3(q.pin_o[j], q.pout_o[l], tsc));
        break;
      case TT_O_O: addL(q.pout_o[j], q.pout_o[l] + tsc); break;
      case TT_E_H: break;
      case TT_B_12:
        addL(q.PO(i, j, ST_1), mul3(q.PI(j, l, ST_2), q.PO(k, l, ST_B), tsc));
        addL(q.PO(j, l, ST_2), mul3(q.PI(i, j, ST_1), q.PO(k, l, ST_B), tsc));
        break;
      default: addL(q.PO(i, j, TT_CHILD[t]), q.PO(k, l, TT_PARENT[t]) + tsc);
    }
  }
};
void Seq::calc_bpp() { /* :180-193 */
  pin_o.assign(L + 1, NINF); pout_o.assign(L + 1, NINF);
  pin.assign((size_t)(L + 1) * (W + 1) * 7, NINF);
  pout.assign((size_t)(L + 1) * (W + 1) * 7, NINF);
  pin_o[0] = 0; pout_o[L] = 0;
  PlainInside fi{*this}; sweep_inside(fi);
  PlainOutside fo{*this}; sweep_outside(fo);
}

// ================================================================= motif DP
struct Trace { int k, l, t, e1, s1; };

struct DP {
  const Model& m;
  Seq& q;
  int S, M, L, W;
  V in_, out_, in_o, out_o;
  V cyk_, cyk_o;
  vector<Trace> tr_, tr_o;
  DP(const Model& mm, Seq& qq) : m(mm), q(qq), S(mm.S()), M(mm.M()), L(qq.L), W(qq.W) {}
  size_t idx(int i, int j, int e, int s) const { return (((size_t)i * (W + 1) + (j - i)) * 7 + e) * S + s; }
  double& I(int i, int j, int e, int s) { return in_[idx(i, j, e, s)]; }
  double& O(int i, int j, int e, int s) { return out_[idx(i, j, e, s)]; }
  double& IO(int j, int s) { return in_o[(size_t)j * S + s]; }
  double& OO(int j, int s) { return out_o[(size_t)j * S + s]; }
  size_t tsize() const { return m.no_rss() ? 0 : (size_t)(L + 1) * (W + 1) * 7 * S; }

  void init_inside(V& t, V& to) { /* motif_trainer.hpp:89-98 */
    t.assign(tsize(), NINF);
    if (!m.no_rss())
      for (int i = 0; i < L + 1; ++i) for (int k = 0; k < M; ++k) t[idx(i, i, ST_L, m.mm.n2s(k, k))] = 0.;
    to.assign((size_t)(L + 1) * S, NINF);
    to[m.mm.n2s(0, 0)] = 0.;
  }
  void init_outside(bool ari = true, bool nasi = true) { /* :100-106 */
    out_.assign(tsize(), NINF);
    out_o.assign((size_t)(L + 1) * S, NINF);
    OO(L, m.mm.n2s(0, 0)) = nasi ? 0. : NINF;
    OO(L, m.mm.n2s(0, M - 1)) = ari ? 0. : NINF;
    OO(L, m.mm.n2s(0, M - 2)) = ari ? 0. : NINF;
  }
  double part_func(bool ari, bool nasi, V& to) { /* :108-112 : sumL(a, sumL(b, c)) */
    double a = nasi ? to[(size_t)L * S + m.mm.n2s(0, 0)] : NINF;
    double b = ari ? to[(size_t)L * S + m.mm.n2s(0, M - 2)] : NINF;
    double c = ari ? to[(size_t)L * S + m.mm.n2s(0, M - 1)] : NINF;
    return lse2(a, lse2(b, c));
  }
  double part_func(bool ari = true, bool nasi = true) { return part_func(ari, nasi, in_o); }
};

// ---- grammar expansion, inside direction (motif_model.hpp:230-423)
template <class G> struct MotifInside {
  DP& d; G& g;
  const Model& m; const Hmm& mm; Seq& q;
  MotifInside(DP& dd, G& gg) : d(dd), g(gg), m(dd.m), mm(dd.m.mm), q(dd.q) {}
  double tauL() const { return m.log_tau; }
  void after(int, int) {}
  void before(int i0, int j) {
    for (int i = j - 1; i0 <= i; --i)
      for (int sid : mm.loop_state) {
        const IS& s = mm.st(sid);
        double lam = m.lam(s);
        for (int s1id : mm.right[sid]) {
          const IS& s1 = mm.st(s1id);
          double w = m.no_prf() ? 0. : m.theta1(s.r, q.seq[j - 1]);
          double ws = q.weight(s.r, j - 1);
          double t = (s.r == s1.r && '.' == mm.node[s.r]) ? tauL() : 0.;
          g(ST_L, ST_L, i, j, i, j - 1, s, s1, s, s, 0., mul3(w, t, ws), lam);
        }
      }
  }
  void on(int tt, int i, int j, int k, int l, double tsc) {
    const vector<int>& seq = q.seq;
    switch (tt) {
      case TT_E_H:
        for (int sid : mm.loop_state) { const IS& s = mm.st(sid); g(ST_E, ST_L, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); }
        break;
      case TT_P_E: case TT_P_P:
        for (const IS& s : mm.state) {
          double lam = m.lam(s);
          for (int s1id : mm.pairt[s.id]) {
            const IS& s1 = mm.st(s1id);
            int rb = (tt == TT_P_E) ? seq[j - 1] : seq[l];
            int rp = (tt == TT_P_E) ? j - 1 : l;
            double w = m.no_prf() ? 0. : m.theta2(s1.l, s.r, seq[i], rb);
            double ws = q.weight(s1.l, i) + q.weight(s.r, rp);
            double t = (s.r == s1.r && ')' == mm.node[s1.r]) ? tauL() : 0.;
            g(ST_P, tt == TT_P_E ? ST_E : ST_P, i, j, k, l, s, s1, s, s, tsc, mul3(w, t, ws), lam);
          }
        }
        break;
      case TT_O_O: case TT_2_2:
        for (const IS& s : mm.state) {
          double lam = m.lam(s);
          for (int s1id : mm.right[s.id]) {
            const IS& s1 = mm.st(s1id);
            double w = m.no_prf() ? 0. : m.theta1(s.r, seq[l]);
            double ws = q.weight(s.r, l);
            double t = (s.r == s1.r && '.' == mm.node[s.r]) ? tauL() : 0.;
            int e = tt == TT_O_O ? ST_O : ST_2;
            g(e, e, i, j, k, l, s, s1, s, s, tsc, mul3(w, t, ws), lam);
          }
        }
        break;
      case TT_O_OP:
        for (const IS& s : mm.state) {
          double lam = m.lam(s);
          for (int h = s.l; h <= s.r; ++h)
            if (mm.reach[s.l][h] && mm.reach[h][s.r]) {
              const IS& s1 = mm.st(mm.n2s(h, s.r));
              const IS& s2 = mm.st(mm.n2s(s.l, h));
              g(ST_O, ST_P, i, j, l, j, s, s1, s2, s, tsc, 0., lam);
            }
        }
        break;
      case TT_E_P:
        for (auto const& ss : mm.lls) {
          const IS& s = mm.st(ss[0]);
          g(ST_E, ST_P, i, j, k, l, s, mm.st(ss[1]), mm.st(ss[2]), mm.st(ss[3]), tsc, 0., m.lam(s));
        }
        break;
      case TT_E_M: for (const IS& s : mm.state) g(ST_E, ST_M, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); break;
      case TT_M_M:
        for (const IS& s : mm.state) {
          double lam = m.lam(s);
          for (int s1id : mm.left[s.id]) {
            const IS& s1 = mm.st(s1id);
            double w = m.no_prf() ? 0. : m.theta1(s1.l, seq[i]);
            double ws = q.weight(s1.l, i);
            double t = (s.l == s1.l && '.' == mm.node[s.l]) ? tauL() : 0.;
            g(ST_M, ST_M, i, j, k, l, s, s1, s, s, tsc, mul3(w, t, ws), lam);
          }
        }
        break;
      case TT_M_B: for (const IS& s : mm.state) g(ST_M, ST_B, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); break;
      case TT_B_12:
        for (const IS& s : mm.state) {
          double lam = m.lam(s);
          for (int h = s.l; h <= s.r; ++h) {
            if (!mm.reach[s.l][h] || !mm.reach[h][s.r]) continue;
            g(ST_B, ST_1, i, j, k, l, s, mm.st(mm.n2s(s.l, h)), mm.st(mm.n2s(h, s.r)), s, tsc, 0., lam);
          }
        }
        break;
      case TT_2_P: for (const IS& s : mm.state) g(ST_2, ST_P, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); break;
      case TT_1_2: for (const IS& s : mm.state) g(ST_1, ST_2, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); break;
      case TT_1_B: for (const IS& s : mm.state) g(ST_1, ST_B, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); break;
    }
  }
};

// ---- grammar expansion, outside direction (motif_model.hpp:425-613)
// g(e, e1, i,j, k,l, s, s1, s2, s3, tsc, wt, lam): (e,s,i,j) = child, (e1,s1,k,l) = parent
template <class G> struct MotifOutside {
  DP& d; G& g;
  const Model& m; const Hmm& mm; Seq& q;
  MotifOutside(DP& dd, G& gg) : d(dd), g(gg), m(dd.m), mm(dd.m.mm), q(dd.q) {}
  double tauL() const { return m.log_tau; }
  void before(int, int) {}
  void after(int j0, int j) {
    for (int i = j0; i <= j; ++i)
      for (int s1id : mm.loop_state) {
        const IS& s1 = mm.st(s1id);
        double lam = m.lam(s1);
        for (int sid : mm.right[s1id]) {
          const IS& s = mm.st(sid);
          double w = m.no_prf() ? 0. : m.theta1(s1.r, q.seq[j]);
          double ws = q.weight(s1.r, j);
          double t = (s.r == s1.r && '.' == mm.node[s1.r]) ? tauL() : 0.;
          g(ST_L, ST_L, i, j, i, j + 1, s, s1, s, s, 0., mul3(w, t, ws), lam);
        }
      }
  }
  void on(int tt, int i, int j, int k, int l, double tsc) {
    const vector<int>& seq = q.seq;
    switch (tt) {
      case TT_E_H:
        for (int sid : mm.loop_state) { const IS& s = mm.st(sid); g(ST_L, ST_E, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); }
        break;
      case TT_P_E: case TT_P_P:
        for (const IS& s1 : mm.state) {
          double lam = m.lam(s1);
          for (int sid : mm.pairt[s1.id]) {
            const IS& s = mm.st(sid);
            double w = m.no_prf() ? 0. : m.theta2(s.l, s1.r, seq[k], seq[j]);
            double ws = q.weight(s.l, k) + q.weight(s1.r, j);
            double t = (s.r == s1.r && ')' == mm.node[s1.r]) ? tauL() : 0.;
            g(tt == TT_P_E ? ST_E : ST_P, ST_P, i, j, k, l, s, s1, s, s, tsc, mul3(w, t, ws), lam);
          }
        }
        break;
      case TT_O_O: case TT_2_2:
        for (const IS& s1 : mm.state) {
          double lam = m.lam(s1);
          for (int sid : mm.right[s1.id]) {
            const IS& s = mm.st(sid);
            double w = m.no_prf() ? 0. : m.theta1(s1.r, seq[j]);
            double ws = q.weight(s1.r, j);
            double t = (s.r == s1.r && '.' == mm.node[s1.r]) ? tauL() : 0.;
            int e = tt == TT_O_O ? ST_O : ST_2;
            g(e, e, i, j, k, l, s, s1, s, s, tsc, mul3(w, t, ws), lam);
          }
        }
        break;
      case TT_O_OP:
        for (const IS& s1 : mm.state) {
          double lam = m.lam(s1);
          for (int h = s1.l; h <= s1.r; ++h)
            if (mm.reach[s1.l][h] && mm.reach[h][s1.r]) {
              const IS& s = mm.st(mm.n2s(h, s1.r));
              const IS& s2 = mm.st(mm.n2s(s1.l, h));
              g(ST_P, ST_O, j, l, i, l, s, s1, s2, s, tsc, 0., lam);
            }
        }
        break;
      case TT_E_P:
        for (auto const& ss : mm.lls) {
          const IS& s0 = mm.st(ss[0]);
          g(ST_P, ST_E, i, j, k, l, mm.st(ss[1]), s0, mm.st(ss[2]), mm.st(ss[3]), tsc, 0., m.lam(s0));
        }
        break;
      case TT_E_M: for (const IS& s : mm.state) g(ST_M, ST_E, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); break;
      case TT_M_M:
        for (const IS& s1 : mm.state) {
          double lam = m.lam(s1);
          for (int sid : mm.left[s1.id]) {
            const IS& s = mm.st(sid);
            double w = m.no_prf() ? 0. : m.theta1(s.l, seq[k]);
            double ws = q.weight(s.l, k);
            double t = (s.l == s1.l && '.' == mm.node[s1.l]) ? tauL() : 0.;
            g(ST_M, ST_M, i, j, k, l, s, s1, s, s, tsc, mul3(w, t, ws), lam);
          }
        }
        break;
      case TT_2_P: for (const IS& s : mm.state) g(ST_P, ST_2, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); break;
      case TT_1_2: for (const IS& s : mm.state) g(ST_2, ST_1, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); break;
      case TT_1_B: for (const IS& s : mm.state) g(ST_B, ST_1, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); break;
      case TT_M_B: for (const IS& s : mm.state) g(ST_B, ST_M, i, j, k, l, s, s, s, s, tsc, 0., m.lam(s)); break;
      case TT_B_12:
        for (const IS& s1 : mm.state) {
          double lam = m.lam(s1);
          for (int h = s1.l; h <= s1.r; ++h) {
            if (!mm.reach[s1.l][h] || !mm.reach[h][s1.r]) continue;
            g(ST_1, ST_B, i, j, k, l, mm.st(mm.n2s(s1.l, h)), s1, mm.st(mm.n2s(h, s1.r)), s1, tsc, 0., lam);
          }
        }
        break;
    }
  }
};

// ---- no-rss forward / backward (motif_model.hpp:171-185, 193-206)
template <class G> void norss_forward(DP& d, G& g) {
  const Model& m = d.m; const Hmm& mm = m.mm; Seq& q = d.q;
  for (int i = 1; i <= d.L; ++i)
    for (const IS& s : mm.state) for (int s1id : mm.right[s.id]) {
      const IS& s1 = mm.st(s1id);
      double w = m.theta1(s.r, q.seq[i - 1]);
      double ws = q.weight(s.r, i - 1);
      double t = (s.r == s1.r && '.' == mm.node[s.r]) ? m.log_tau : 0.;
      g(ST_O, ST_O, 0, i, 0, i - 1, s, s1, s, s, 0., mul3(w, t, ws), 0.);
    }
}
template <class G> void norss_backward(DP& d, G& g) {
  const Model& m = d.m; const Hmm& mm = m.mm; Seq& q = d.q;
  for (int i = d.L; 1 <= i; --i)
    for (const IS& s : mm.state) for (int s1id : mm.right[s.id]) {
      const IS& s1 = mm.st(s1id);
      double w = m.theta1(s.r, q.seq[i - 1]);
      double ws = q.weight(s.r, i - 1);
      double t = (s.r == s1.r && '.' == mm.node[s.r]) ? m.log_tau : 0.;
      g(ST_O, ST_O, 0, i - 1, 0, i, s1, s, s, s, 0., mul3(w, t, ws), 0.);
    }
}

template <class G> void run_inside(DP& d, G& g) {
  if (d.m.no_rss()) norss_forward(d, g);
  else { MotifInside<G> f(d, g); d.q.sweep_inside(f); }
}
template <class G> void run_outside(DP& d, G& g) {
  if (d.m.no_rss()) norss_backward(d, g);
  else { MotifOutside<G> f(d, g); d.q.sweep_outside(f); }
}

// ---- semiring functors
inline void inside_update(DP& d, V& T, V& TO, int e, int e1, int i, int j, int k, int l, const IS& s, const IS& s1,
                          const IS& s2, const IS& s3, double diff) { /* motif_trainer.hpp:295-326 */
  if (ST_E == e && ST_P == e1)
    addL(T[d.idx(i, j, e, s.id)], mul4(T[d.idx(k, l, e1, s1.id)], T[d.idx(i, k, ST_L, s2.id)], T[d.idx(l, j, ST_L, s3.id)], diff));
  else if (ST_O == e && ST_P == e1)
    addL(TO[(size_t)j * d.S + s.id], mul3(TO[(size_t)k * d.S + s2.id], T[d.idx(k, l, e1, s1.id)], diff));
  else if (ST_B == e && ST_1 == e1)
    addL(T[d.idx(i, j, e, s.id)], mul3(T[d.idx(k, l, ST_1, s1.id)], T[d.idx(l, j, ST_2, s2.id)], diff));
  else if (ST_O == e && ST_O == e1)
    addL(TO[(size_t)j * d.S + s.id], TO[(size_t)l * d.S + s1.id] + diff);
  else
    addL(T[d.idx(i, j, e, s.id)], T[d.idx(k, l, e1, s1.id)] + diff);
}

struct TrainIn {
  DP& d;
  void operator()(int e, int e1, int i, int j, int k, int l, const IS& s, const IS& s1, const IS& s2, const IS& s3,
                  double tsc, double wt, double lam) {
    inside_update(d, d.in_, d.in_o, e, e1, i, j, k, l, s, s1, s2, s3, wt + lam * tsc);
  }
};

inline double outside_z(DP& d, int e, int e1, int i, int j, int k, int l, const IS& s, const IS& s1, const IS& s2,
                        const IS& s3, double diff, double Z) { /* motif_trainer.hpp:357-377 */
  double a = (ST_O == e) ? d.IO(j, s.id) : d.I(i, j, e, s.id);
  double b = (ST_E == e1 && ST_P == e) ? mul3(d.O(k, l, e1, s1.id), d.I(k, i, ST_L, s2.id), d.I(j, l, ST_L, s3.id))
             : (ST_O == e1 && ST_P == e) ? d.OO(l, s1.id) + d.IO(i, s2.id)
             : (ST_B == e1 && ST_1 == e) ? d.O(k, l, e1, s1.id) + d.I(j, l, ST_2, s2.id)
             : (ST_O == e1 && ST_O == e) ? d.OO(l, s1.id)
                                         : d.O(k, l, e1, s1.id);
  return mul3(diff, a, b) - Z;
}
inline void outside_update(DP& d, int e, int e1, int i, int j, int k, int l, const IS& s, const IS& s1, const IS& s2,
                           const IS& s3, double diff) { /* :408-456 */
  if (ST_E == e1 && ST_P == e) {
    addL(d.O(i, j, e, s.id), mul4(d.O(k, l, e1, s1.id), d.I(k, i, ST_L, s2.id), d.I(j, l, ST_L, s3.id), diff));
    addL(d.O(k, i, ST_L, s2.id), mul4(d.O(k, l, e1, s1.id), d.I(i, j, e, s.id), d.I(j, l, ST_L, s3.id), diff));
    addL(d.O(j, l, ST_L, s3.id), mul4(d.O(k, l, e1, s1.id), d.I(i, j, e, s.id), d.I(k, i, ST_L, s2.id), diff));
  } else if (ST_O == e1 && ST_P == e) {
    addL(d.O(i, j, e, s.id), mul3(d.OO(l, s1.id), d.IO(i, s2.id), diff));
    addL(d.OO(i, s2.id), mul3(d.OO(l, s1.id), d.I(i, j, e, s.id), diff));
  } else if (ST_B == e1 && ST_1 == e) {
    addL(d.O(i, j, e, s.id), mul3(d.O(k, l, e1, s1.id), d.I(j, l, ST_2, s2.id), diff));
    addL(d.O(j, l, ST_2, s2.id), mul3(d.I(i, j, e, s.id), d.O(k, l, e1, s1.id), diff));
  } else if (ST_O == e1 && ST_O == e) {
    addL(d.OO(j, s.id), d.OO(l, s1.id) + diff);
  } else {
    addL(d.O(i, j, e, s.id), d.O(k, l, e1, s1.id) + diff);
  }
}
inline void count_emissions(DP& d, VV& EN, int e1, int i, int j, int k, int l, const IS& s, const IS& s1, double ez) {
  /* motif_trainer.hpp:383-406 */
  const Model& m = d.m; const vector<int>& seq = d.q.seq;
  switch (e1) {
    case ST_P: if (k == i - 1 && j == l - 1 && !m.no_prf()) m.emit2(EN, s.l, s1.r, seq[k], seq[j], ez); break;
    case ST_2: case ST_O: case ST_L: if (k == i && j == l - 1 && !m.no_prf()) m.emit1(EN, s1.r, seq[j], ez); break;
    case ST_M: if (k == i - 1 && j == l && !m.no_prf()) m.emit1(EN, s.l, seq[k], ez); break;
    default: break;
  }
}

struct TrainOut {
  DP& d; double Z; V& EH; VV& EN;
  void operator()(int e, int e1, int i, int j, int k, int l, const IS& s, const IS& s1, const IS& s2, const IS& s3,
                  double tsc, double wt, double lam) {
    double diff = wt + lam * tsc;
    double z = outside_z(d, e, e1, i, j, k, l, s, s1, s2, s3, diff, Z);
    if (NINF == z) return;
    if (lam == d.m.lambda[0]) EH[0] += tsc * exp(z); else EH[1] += tsc * exp(z); /* :380-381 */
    count_emissions(d, EN, e1, i, j, k, l, s, s1, exp(z));
    outside_update(d, e, e1, i, j, k, l, s, s1, s2, s3, diff);
  }
};

struct ScanOut { /* motif_scanner.hpp:438-578 */
  DP& d; double Z; V& Pys; V& Pyi; VV& EN;
  void operator()(int e, int e1, int i, int j, int k, int l, const IS& s, const IS& s1, const IS& s2, const IS& s3,
                  double tsc, double wt, double lam) {
    double diff = wt + lam * tsc;
    double z = outside_z(d, e, e1, i, j, k, l, s, s1, s2, s3, diff, Z);
    if (NINF == z) return;
    count_emissions(d, EN, e1, i, j, k, l, s, s1, exp(z));
    outside_update(d, e, e1, i, j, k, l, s, s1, s2, s3, diff);
    const int M = d.M;
    switch (e1) {
      case ST_P:
        if (k == i - 1 && j == l - 1) {
          if (0 == s1.l && 1 == s.l) addL(Pys[k], z);
          if (0 == s.r && 1 == s1.r) addL(Pys[j], z);
          if (0 != s.l && M - 1 != s.l) addL(Pyi[k], z);
          if (0 != s1.r && M - 1 != s1.r) addL(Pyi[j], z);
        }
        break;
      case ST_2: case ST_O: case ST_L:
        if (i == k && j == l - 1) {
          if (0 == s.r && 1 == s1.r) addL(Pys[j], z);
          if (0 != s1.r && M - 1 != s1.r) addL(Pyi[j], z);
        }
        break;
      case ST_M:
        if (k == i - 1 && j == l) {
          if (0 == s1.l && 1 == s.l) addL(Pys[k], z);
          if (0 != s.l && M - 1 != s.l) addL(Pyi[k], z);
        }
        break;
      default: break;
    }
  }
};

struct EndIn { /* motif_scanner.hpp:594-664 */
  DP& d; int Ys;
  void operator()(int e, int e1, int i, int j, int k, int l, const IS& s, const IS& s1, const IS& s2, const IS& s3,
                  double tsc, double wt, double lam) {
    switch (e) {
      case ST_P:
        if (i == k - 1 && l == j - 1) {
          if (i == Ys) if (0 != s.l || 1 != s1.l) return;
          if (l == Ys) if (0 != s1.r || 1 != s.r) return;
        }
        break;
      case ST_O: case ST_2: case ST_L:
        if (i == k && l == j - 1) { if (l == Ys) if (0 != s1.r || 1 != s.r) return; }
        break;
      case ST_M:
        if (i == k - 1 && l == j) { if (i == Ys) if (0 != s.l || 1 != s1.l) return; }
        break;
      default: break;
    }
    inside_update(d, d.in_, d.in_o, e, e1, i, j, k, l, s, s1, s2, s3, wt + lam * tsc);
  }
};

struct EndOut { /* motif_scanner.hpp:682-799 */
  DP& d; int Ys; double Z; V& Pye;
  void operator()(int e, int e1, int i, int j, int k, int l, const IS& s, const IS& s1, const IS& s2, const IS& s3,
                  double tsc, double wt, double lam) {
    double diff = wt + lam * tsc;
    double z = outside_z(d, e, e1, i, j, k, l, s, s1, s2, s3, diff, Z);
    if (NINF == z) return;
    const int M = d.M, L = d.L;
    switch (e1) {
      case ST_P:
        if (k == i - 1 && j == l - 1) {
          if (Ys == k) if (0 != s1.l || 1 != s.l) return;
          if (Ys == j) if (0 != s.r || 1 != s1.r) return;
          if (M - 2 == s1.l && M - 1 == s.l) addL(Pye[k], z);
          if (M - 2 == s.r && M - 1 == s1.r) addL(Pye[j], z);
          if (M - 2 == s1.r && L == l) addL(Pye[L], z);
        }
        break;
      case ST_O: case ST_2: case ST_L:
        if (i == k && j == l - 1) {
          if (Ys == j) if (0 != s.r || 1 != s1.r) return;
          if (M - 2 == s.r && M - 1 == s1.r) addL(Pye[j], z);
          if (M - 2 == s1.r && L == l) addL(Pye[L], z);
        }
        break;
      case ST_M:
        if (k == i - 1 && j == l) {
          if (Ys == k) if (0 != s1.l || 1 != s.l) return;
          if (M - 2 == s1.l && M - 1 == s.l) addL(Pye[k], z);
        }
        break;
      default: break;
    }
    outside_update(d, e, e1, i, j, k, l, s, s1, s2, s3, diff);
  }
};

struct Cyk { /* motif_scanner.hpp:802-913 */
  DP& d; int ys, ye;
  void compare(int e, int e1, int i, int j, int k, int l, const IS& s, const IS& s1, double& x, double y) {
    if (x < y) {
      x = y;
      Trace t{k, l, states_to_trans(e, e1), e1, s1.id};
      if (ST_O == e) d.tr_o[(size_t)j * d.S + s.id] = t; else d.tr_[d.idx(i, j, e, s.id)] = t;
    }
  }
  void operator()(int e, int e1, int i, int j, int k, int l, const IS& s, const IS& s1, const IS& s2, const IS& s3,
                  double tsc, double wt, double lam) {
    const int M = d.M, L = d.L;
    switch (e) {
      case ST_P:
        if (i == k - 1 && l == j - 1) {
          if (i == ys && !(0 == s.l && 1 == s1.l)) return;
          if (l == ys && !(0 == s1.r && 1 == s.r)) return;
          if (i == ye && !(M - 2 == s.l && M - 1 == s1.l)) return;
          if (l == ye && !(M - 2 == s1.r && M - 1 == s.r)) return;
          if ((j == ye && L == j) && M - 2 != s.r) return;
        }
        break;
      case ST_O: case ST_2: case ST_L:
        if (i == k && l == j - 1) {
          if (l == ys && !(0 == s1.r && 1 == s.r)) return;
          if (l == ye && !(M - 2 == s1.r && M - 1 == s.r)) return;
          if ((j == ye && L == j) && M - 2 != s.r) return;
        }
        break;
      case ST_M:
        if (i == k - 1 && l == j) {
          if (i == ys && !(0 == s.l && 1 == s1.l)) return;
          if (i == ye && !(M - 2 == s.l && M - 1 == s1.l)) return;
        }
        break;
      default: break;
    }
    double diff = wt + lam * tsc;
    V& T = d.cyk_; V& TO = d.cyk_o;
    if (ST_E == e && ST_P == e1)
      compare(e, e1, i, j, k, l, s, s1, T[d.idx(i, j, e, s.id)],
              mul4(T[d.idx(k, l, e1, s1.id)], T[d.idx(i, k, ST_L, s2.id)], T[d.idx(l, j, ST_L, s3.id)], diff));
    else if (ST_O == e && ST_P == e1)
      compare(e, e1, i, j, k, l, s, s1, TO[(size_t)j * d.S + s.id],
              mul3(TO[(size_t)k * d.S + d.m.mm.n2s(s.l, s1.l)], T[d.idx(k, l, e1, s1.id)], diff));
    else if (ST_B == e && ST_1 == e1)
      compare(e, e1, i, j, k, l, s, s1, T[d.idx(i, j, e, s.id)], mul3(T[d.idx(k, l, e1, s1.id)], T[d.idx(l, j, ST_2, s2.id)], diff));
    else if (ST_O == e && ST_O == e1)
      compare(e, e1, i, j, k, l, s, s1, TO[(size_t)j * d.S + s.id], TO[(size_t)l * d.S + s1.id] + diff);
    else
      compare(e, e1, i, j, k, l, s, s1, T[d.idx(i, j, e, s.id)], T[d.idx(k, l, e1, s1.id)] + diff);
  }
};

// ---- traceback (motif_scanner.hpp:262-362)
void trace_back(DP& d, int i0, int j0, int e0, int s0, string& rss, vector<int>& path) {
  const Hmm& mm = d.m.mm;
  struct T2 { int i, j, e, s; };
  vector<T2> st{{i0, j0, e0, s0}};
  while (!st.empty()) {
    T2 t2 = st.back(); st.pop_back();
    const Trace& t = (ST_O == t2.e) ? d.tr_o[(size_t)t2.j * d.S + t2.s] : d.tr_[d.idx(t2.i, t2.j, t2.e, t2.s)];
    if (t.t < 0) continue; /* leaf (the reference reads state()[-1] here and falls through the switch) */
    const IS& s1 = mm.st(t.s1);
    const IS& s2s = mm.st(t2.s);
    switch (t.t) {
      case TT_L_L: path[t.l] = s2s.r; st.push_back({t.k, t.l, t.e1, s1.id}); break;
      case TT_O_O: path[t.l] = s2s.r; rss[t.l] = 'O'; st.push_back({t.k, t.l, t.e1, s1.id}); break;
      case TT_2_2: path[t.l] = s2s.r; rss[t.l] = 'M'; st.push_back({t.k, t.l, t.e1, s1.id}); break;
      case TT_E_H: { int n = t2.j - t2.i; rss.replace(t2.i, n, n, 'H'); st.push_back({t.k, t.l, t.e1, t2.s}); break; }
      case TT_E_M: case TT_M_B: case TT_2_P: case TT_1_2: case TT_1_B: st.push_back({t.k, t.l, t.e1, t2.s}); break;
      case TT_P_E: case TT_P_P:
        path[t2.i] = s1.l; rss[t2.i] = 'L'; path[t.l] = s2s.r; rss[t.l] = 'R';
        st.push_back({t.k, t.l, t.e1, s1.id});
        break;
      case TT_O_OP: {
        int s2 = mm.n2s(s2s.l, s1.l);
        st.push_back({t.k, t.l, t.e1, s1.id});
        st.push_back({s2s.l, t.k, ST_O, s2});
        break;
      }
      case TT_E_P: {
        int s2 = mm.n2s(s2s.l, s1.l), s3 = mm.n2s(s1.r, s2s.r);
        int n1 = t2.j - t.l, n2 = t.k - t2.i;
        if (0 == n1) rss.replace(t2.i, n2, n2, 'B');
        else if (0 == n2) rss.replace(t.l, n1, n1, 'B');
        else { rss.replace(t2.i, n2, n2, 'I'); rss.replace(t.l, n1, n1, 'I'); }
        st.push_back({t.l, t2.j, ST_L, s3});
        st.push_back({t2.i, t.k, ST_L, s2});
        st.push_back({t.k, t.l, t.e1, s1.id});
        break;
      }
      case TT_B_12: {
        int s2 = mm.n2s(s1.r, s2s.r);
        st.push_back({t.l, t2.j, ST_2, s2});
        st.push_back({t.k, t.l, t.e1, s1.id});
        break;
      }
      case TT_M_M: path[t2.i] = s1.l; rss[t2.i] = 'M'; st.push_back({t.k, t.l, ST_M, s1.id}); break;
    }
  }
}

thread_local string g_err;

}  // namespace

struct orc_model { Model m; };

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

orc_model* orc_create(const char* pattern, const char* par_text, int max_span, int max_iloop, double min_bpp,
                      double tau, int flags) {
  try {
    orc_model* h = new orc_model();
    Model& m = h->m;
    m.flags = flags;
    if ((flags & ORC_NO_RSS) && (flags & ORC_NO_PRF)) die("no-rss, no-profile are exclusive.");
    string pat(pattern);
    if (flags & ORC_NO_RSS) for (auto& c : pat) if (c == '_') c = '.';
    m.mm.build(pat);
    if ((flags & ORC_NO_RSS) && pat.find(')') != string::npos) die("search pattern must not include pair when no-rss mode");
    m.ep.parse(par_text ? par_text : "");
    m.max_pair = max_span; m.max_iloop = max_iloop;
    m.min_bpp = min_bpp; m.min_lnbpp = log(min_bpp);
    m.tau = tau; m.log_tau = log(tau);
    return h;
  } catch (std::exception& e) { g_err = e.what(); return nullptr; }
}
void orc_destroy(orc_model* h) { delete h; }
int orc_n_param(orc_model* h) { return h->m.n_theta() + 2; }
int orc_n_state(orc_model* h) { return h->m.S(); }
int orc_n_node(orc_model* h) { return h->m.M(); }
void orc_get_params(orc_model* h, double* x) { V v; h->m.pack(v); std::copy(v.begin(), v.end(), x); }
void orc_set_params(orc_model* h, const double* x) { h->m.unpack(x); }

int orc_hmm_json(orc_model* h, char* buf, int cap) {
  const Hmm& mm = h->m.mm;
  std::ostringstream o;
  auto ids = [&](const vector<int>& v) { o << "["; for (size_t i = 0; i < v.size(); ++i) o << (i ? "," : "") << v[i]; o << "]"; };
  o << "{\"reg_pattern\":\"" << mm.reg << "\",\"M\":" << mm.M << ",\"S\":" << mm.S() << ",\"node\":\"";
  for (int c : mm.node) o << char(c);
  o << "\",\"theta_id\":"; ids(mm.theta_id);
  o << ",\"theta_sizes\":["; for (size_t i = 0; i < mm.theta.size(); ++i) o << (i ? "," : "") << mm.theta[i].size(); o << "]";
  o << ",\"state\":["; for (int s = 0; s < mm.S(); ++s) o << (s ? "," : "") << "[" << mm.state[s].l << "," << mm.state[s].r << "]"; o << "]";
  o << ",\"loop_state\":"; ids(mm.loop_state);
  o << ",\"reachable\":["; for (int a = 0; a < mm.M; ++a) { o << (a ? "," : "") << "["; for (int b = 0; b < mm.M; ++b) o << (b ? "," : "") << int(mm.reach[a][b]); o << "]"; } o << "]";
  o << ",\"right\":["; for (int s = 0; s < mm.S(); ++s) { if (s) o << ","; ids(mm.right[s]); } o << "]";
  o << ",\"left\":["; for (int s = 0; s < mm.S(); ++s) { if (s) o << ","; ids(mm.left[s]); } o << "]";
  o << ",\"pair\":["; for (int s = 0; s < mm.S(); ++s) { if (s) o << ","; ids(mm.pairt[s]); } o << "]";
  o << ",\"loop_loop\":["; for (size_t i = 0; i < mm.lls.size(); ++i) o << (i ? "," : "") << "[" << mm.lls[i][0] << "," << mm.lls[i][1] << "," << mm.lls[i][2] << "," << mm.lls[i][3] << "]"; o << "]}";
  string s = o.str();
  if ((int)s.size() + 1 > cap) return -(int)s.size() - 1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}

int orc_energy_table(orc_model* h, const char* name, double* out, int cap) {
  EnergyTables& e = h->m.ep;
  struct Ent { const char* n; double* p; int c; };
  Ent tab[] = {{"stack", &e.stack[0][0], 49}, {"hairpin", e.hairpin, 31}, {"bulge", e.bulge, 31}, {"internal", e.internal_, 31},
               {"ninio", e.ninio, 31}, {"mismatch_h", &e.mm_h[0][0][0], 175}, {"mismatch_i", &e.mm_i[0][0][0], 175},
               {"mismatch_m", &e.mm_m[0][0][0], 175}, {"mismatch_1ni", &e.mm_1ni[0][0][0], 175},
               {"mismatch_23i", &e.mm_23i[0][0][0], 175}, {"mismatch_ext", &e.mm_ext[0][0][0], 175},
               {"dangle5", &e.dangle5[0][0], 40}, {"dangle3", &e.dangle3[0][0], 40}, {"int_11", &e.int11[0][0][0][0], 1600},
               {"int_21", &e.int21[0][0][0][0][0], 8000}, {"int_22", &e.int22[0][0][0][0][0][0], 40000},
               {"triloop", e.tri, 40}, {"tetraloop", e.tetra, 40}, {"hexaloop", e.hexa, 40}, {"term_au", &e.term_au, 1},
               {"mlintern", &e.mlintern, 1}, {"mlclosing", &e.mlclosing, 1}, {"ml_base", &e.ml_base, 1}, {"lxc37", &e.lxc37, 1}};
  for (auto& t : tab) if (!strcmp(t.n, name)) {
    if (cap < t.c) return -t.c;
    std::copy(t.p, t.p + t.c, out);
    return t.c;
  }
  return 0;
}
static vector<int> to_vec(const uint8_t* s, int L) { return vector<int>(s, s + L); }
double orc_hairpin_energy(orc_model* h, const uint8_t* seq, int L, int i, int j) { return h->m.ep.hairpin_energy(i, j, to_vec(seq, L)); }
double orc_loop_energy(orc_model* h, const uint8_t* seq, int L, int i, int j, int p, int q) { return h->m.ep.loop_energy(i, j, p, q, to_vec(seq, L)); }
double orc_sum_ext_m(orc_model* h, const uint8_t* seq, int L, int i, int j, int ext) { return h->m.ep.sum_ext_m(i, j, ext, to_vec(seq, L)); }

int orc_bpp(orc_model* h, const uint8_t* seq, int L, double* lnbpp, uint8_t* kept, double* bpp_eff, double* lnZ) {
  try {
    Model& m = h->m;
    Seq q;
    q.prepare(m, to_vec(seq, L), "");
    int W = q.W;
    if (kept) for (int i = 0; i < (L + 1) * (W + 1); ++i) kept[i] = q.bp_ok[i];
    if (bpp_eff) *bpp_eff = q.bpp_eff;
    if (lnbpp || lnZ) {
      Model m0 = m; m0.min_bpp = 0; m0.min_lnbpp = NINF;
      Seq q0; q0.prepare(m0, to_vec(seq, L), "");
      q0.fill_left(); q0.calc_bpp();
      if (lnZ) *lnZ = q0.pin_o[L];
      if (lnbpp) for (int i = 0; i <= L; ++i) for (int d = 0; d <= W; ++d)
        lnbpp[i * (W + 1) + d] = (i + d <= L && q0.ok(i, d)) ? q0.lnBPP(i, i + d) : NINF;
    }
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 1; }
}

static void flatten(const VV& e, double* out) { int k = 0; for (auto const& r : e) for (double v : r) out[k++] = v; }

static int train_seq_impl(Model& m, const vector<int>& seq, const vector<int>& qual, const string& fix,
                          orc_seq_result* res, VV& ENo, V& EHo, VV& ENx, V& EHx, double* inside_o, double* inside,
                          double* outside, double* outside_o) {
  /* motif_trainer.hpp:204-227 */
  Seq q;
  q.prepare(m, seq, fix);
  q.set_ws(qual);
  DP d(m, q);
  d.init_inside(d.in_, d.in_o);
  d.init_outside(true, true);
  TrainIn fi{d};
  run_inside(d, fi);
  double Zo = d.part_func(true, true), Za = d.part_func(true, false), Zn = d.part_func(false, true);
  res->Zo = Zo; res->Zari = Za; res->Znasi = Zn; res->L = q.L; res->W = q.W;
  res->bpp_eff = m.no_rss() ? 0. : q.bpp_eff; res->f = 0; res->skipped = 0;
  if (inside_o) std::copy(d.in_o.begin(), d.in_o.end(), inside_o);
  if (inside) std::copy(d.in_.begin(), d.in_.end(), inside);
  if (!(std::isfinite(Zo) && std::isfinite(Za))) { res->skipped = 1; return 0; }
  if (m.lik_ratio() && NINF < q.ws.back()) {
    /* --lik-ratio, sequence without motif (motif_trainer.hpp:163-171): the roles are swapped,
       "x" = Z(ari,nasi) with the full terminals, "o" = Z(ari only) */
    TrainOut fx{d, Zo, EHx, ENx};
    run_outside(d, fx);
    if (outside) std::copy(d.out_.begin(), d.out_.end(), outside);
    if (outside_o) std::copy(d.out_o.begin(), d.out_o.end(), outside_o);
    d.init_outside(true, false);
    TrainOut fo{d, Za, EHo, ENo};
    run_outside(d, fo);
    res->f = Za - Zo;
    return 0;
  }
  TrainOut fo{d, Zo, EHo, ENo};
  run_outside(d, fo);
  if (outside) std::copy(d.out_.begin(), d.out_.end(), outside);
  if (outside_o) std::copy(d.out_o.begin(), d.out_o.end(), outside_o);
  double Zx;
  if (NINF < q.ws.back()) { d.init_outside(false, true); Zx = Zn; }
  else { d.init_outside(true, false); Zx = Za; }   /* (the same with --lik-ratio, motif_trainer.hpp:172-180) */
  TrainOut fx{d, Zx, EHx, ENx};
  run_outside(d, fx);
  res->f = Zo - Zx;
  return 0;
}

int orc_train_seq(orc_model* h, const uint8_t* seq, int L, const uint8_t* qual, const char* fix_rss, orc_seq_result* res,
                  double* ENo, double* EHo, double* ENx, double* EHx, double* inside_o, double* inside, double* outside,
                  double* outside_o) {
  try {
    Model& m = h->m;
    VV eno, enx; V eho{0., 0.}, ehx{0., 0.};
    m.clear_counts(eno); m.clear_counts(enx);
    vector<int> q(qual, qual + L + 1);
    train_seq_impl(m, to_vec(seq, L), q, fix_rss ? fix_rss : "", res, eno, eho, enx, ehx, inside_o, inside, outside, outside_o);
    if (ENo) flatten(eno, ENo);
    if (ENx) flatten(enx, ENx);
    if (EHo) { EHo[0] = eho[0]; EHo[1] = eho[1]; }
    if (EHx) { EHx[0] = ehx[0]; EHx[1] = ehx[1]; }
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 1; }
}

int orc_train_eval(orc_model* h, const double* x, const uint8_t* seqs, const int32_t* off, const uint8_t* quals,
                   const int32_t* qoff, int n_seq, int n_threads, double* fn, double* gr, double* sum_eff,
                   int32_t* n_skipped) {
  try {
    h->m.unpack(x);
    const int np = h->m.n_theta() + 2;
    std::fill(gr, gr + np, 0.);
    *fn = 0; *sum_eff = 0; if (n_skipped) *n_skipped = 0;
    std::atomic<int> next(0);
    std::mutex mx;
    string err;
    auto worker = [&]() { /* one RNAelemTrainDP copy per thread (motif_trainer.hpp:124-272) */
      try {
        Model m = h->m;
        VV ENo, ENx; V EHo{0., 0.}, EHx{0., 0.};
        m.clear_counts(ENo); m.clear_counts(ENx);
        double f = 0, eff = 0; int skipped = 0;
        for (;;) {
          int n = next.fetch_add(1);
          if (n >= n_seq) break;
          int L = off[n + 1] - off[n];
          if (qoff[n + 1] - qoff[n] != L + 1) die("bad seq format.");
          vector<int> sq(seqs + off[n], seqs + off[n + 1]);
          vector<int> ql(quals + qoff[n], quals + qoff[n + 1]);
          orc_seq_result r;
          train_seq_impl(m, sq, ql, "", &r, ENo, EHo, ENx, EHx, nullptr, nullptr, nullptr, nullptr);
          if (r.skipped) { ++skipped; continue; }
          f += r.f; eff += r.bpp_eff;
        }
        std::lock_guard<std::mutex> lk(mx);
        int k = 0;
        if (m.softmax()) { /* :251-261 */
          for (size_t i = 0; i < ENo.size(); ++i) {
            double tot = 0.;
            for (size_t j = 0; j < ENo[i].size(); ++j) tot += ENo[i][j] - ENx[i][j];
            for (size_t j = 0; j < ENo[i].size(); ++j) {
              double tmp = ENo[i][j] - ENx[i][j];
              double p = exp(m.mm.theta[i][j]);
              gr[k++] += (1 - p) * tmp - p * (tot - tmp);
            }
          }
        } else {
          for (size_t i = 0; i < ENo.size(); ++i) for (size_t j = 0; j < ENo[i].size(); ++j) gr[k++] += ENo[i][j] - ENx[i][j];
        }
        for (int i = 0; i < 2; ++i) gr[k++] += EHo[i] - EHx[i];
        *fn += f; *sum_eff += eff; if (n_skipped) *n_skipped += skipped;
      } catch (std::exception& e) { std::lock_guard<std::mutex> lk(mx); err = e.what(); }
    };
    if (n_threads <= 1) worker();
    else {
      vector<std::thread> th;
      for (int t = 0; t < n_threads; ++t) th.emplace_back(worker);
      for (auto& t : th) t.join();
    }
    if (!err.empty()) die(err);
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 1; }
}

int orc_scan_seq(orc_model* h, const uint8_t* seq, int L, const uint8_t* qual, orc_scan_result* res, double* start,
                 double* end, double* inner, int32_t* psihat, char* rss_out, double* EN_out) {
  try { /* motif_scanner.hpp:215-260 */
    Model& m = h->m;
    Seq q;
    q.prepare(m, to_vec(seq, L), "");
    q.set_ws(vector<int>(qual, qual + L + 1));
    DP d(m, q);
    V Pys(L, NINF), Pye(L + 1, NINF), Pyi(L, NINF);
    VV EN; m.clear_counts(EN);
    /* calc_motif_start_position :186-193 */
    d.init_inside(d.in_, d.in_o);
    d.init_outside();
    TrainIn fi{d}; run_inside(d, fi);
    double ZL = d.part_func();
    ScanOut so{d, ZL, Pys, Pyi, EN}; run_outside(d, so);
    int Ys = 0; { double mx = std::numeric_limits<double>::lowest(); for (int i = 0; i < L; ++i) if (mx <= Pys[i]) { Ys = i; mx = Pys[i]; } }
    double PyNL = d.IO(L, m.mm.n2s(0, 0)) - ZL;
    /* calc_motif_end_position :195-202 */
    d.init_inside(d.in_, d.in_o);
    d.init_outside();
    EndIn ei{d, Ys}; run_inside(d, ei);
    double ZeL = d.part_func();
    EndOut eo{d, Ys, ZeL, Pye}; run_outside(d, eo);
    int Ye = 0; { double mx = std::numeric_limits<double>::lowest(); for (int i = 0; i <= L; ++i) if (mx <= Pye[i]) { Ye = i; mx = Pye[i]; } }
    /* calc_viterbi_alignment :172-184 */
    d.init_inside(d.cyk_, d.cyk_o);
    d.tr_.assign(d.tsize(), Trace{-1, -1, -1, -1, -1});
    d.tr_o.assign((size_t)(L + 1) * d.S, Trace{-1, -1, -1, -1, -1});
    string rss(L, ' '); vector<int> path(L, 0);
    Cyk cf{d, Ys, Ye}; run_inside(d, cf);
    int sa = m.mm.n2s(0, d.M - 2), sb = m.mm.n2s(0, d.M - 1);
    int s0 = d.cyk_o[(size_t)L * d.S + sa] < d.cyk_o[(size_t)L * d.S + sb] ? sb : sa;
    trace_back(d, 0, L, ST_O, s0, rss, path);
    double tot = NINF; for (double v : Pys) addL(tot, v);
    res->Ys = Ys; res->Ye = Ye; res->exist_prob = exp(tot); res->ZL = ZL; res->ZeL = ZeL; res->PyNL = PyNL;
    if (start) std::copy(Pys.begin(), Pys.end(), start);
    if (end) std::copy(Pye.begin(), Pye.end(), end);
    if (inner) std::copy(Pyi.begin(), Pyi.end(), inner);
    if (psihat) for (int i = 0; i < L; ++i) psihat[i] = path[i];
    if (rss_out) memcpy(rss_out, rss.data(), L);
    if (EN_out) { int k = 0; for (auto const& r : EN) for (double v : r) EN_out[k++] += v; }
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 1; }
}

}  // extern "C"
