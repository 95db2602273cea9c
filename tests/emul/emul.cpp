// emul.cpp -- TEST-ONLY serial driver around the product's host/device-agnostic rule headers.
//
// Built by tests (tests/emul/build.py) into tests/emul/libelemdp_emul.so; it is NOT part of
// libelemdp.so and nothing in rnaelem_amd/ can reach it.  It runs the very same per-target
// functions the HIP kernels run (rnaelem_amd/csrc/dp_rules.h, plan_rules.h, energy_rules.h),
// one target after the other on the CPU, so that the gather formulation, the plan builder, the
// automaton flattening and the energy parser can be checked against the oracle without a GPU
// (pytest -m "not gpu").  The thread mapping, LDS staging, barriers and reductions of the real
// kernels are covered by the -m gpu tests.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../rnaelem_amd/csrc/automaton.h"
#include "../../rnaelem_amd/csrc/dp_rules.h"
#include "../../rnaelem_amd/csrc/energy_rules.h"
#include "../../rnaelem_amd/csrc/energy_tables.h"
#include "../../rnaelem_amd/csrc/host_prep.h"
#include <limits>
#include <memory>
#include "../../rnaelem_amd/csrc/lin_rules.h"
#include "../../rnaelem_amd/csrc/lin_fast.h"
#include "../../rnaelem_amd/csrc/plan_rules.h"
#include "../../rnaelem_amd/csrc/scan_rules.h"
#include "../../rnaelem_amd/csrc/bpp_cand.h"

using namespace elemdp;
static long g_enum_checked = 0, g_enum_mismatch = 0;
static long g_fast_cells = 0;   // cells evaluated through the table-driven forms (tests assert the path was taken)

namespace {

const double NEG = -HUGE_VAL;

enum { F_NO_RSS = 1, F_NO_PRF = 2, F_NO_ENE = 4, F_SOFTMAX = 8, F_FIX_RSS = 1 << 9, F_NO_TURN = 1 << 10 };

struct Emu {
  Automaton* au = nullptr;
  EnergyTables et;
  AutomatonLayout lay, lay0, lay_r, lay_s;
  std::vector<int32_t> ints, ints0, ints_r, ints_s;   // (_s: with the shadow copy of state (0,0): the merged outside sweep)
  bool fast = false;   // band targets through the table-driven forms (lin_fast.h + the records of the fast blobs)
  int max_span, max_iloop, flags;
  double min_bpp, tau;
  ~Emu() { delete au; }
};

// CPU-side plan of one sequence (what plan kernels build on the GPU)
struct HostPlan {
  int L, W, C;
  std::vector<uint8_t> seq, unp, item_in;
  std::vector<double> ws;
  std::vector<uint32_t> okbits, okbits_end;   // okbits_end: the same pairs indexed by (end, span), filled by view()
  std::vector<int16_t> dmin;
  std::vector<double> e_stack, e_ext, e_ml, e_close, e_hp;
  std::vector<LoopItem> items;
  std::vector<int32_t> by_outer_off, by_inner_off, by_inner_idx, by_left_off, by_left_idx, by_right_off, by_right_idx;
  std::vector<int32_t> ndot;
  bool have_fix = false;
  int n_pairs = 0;

  bool ok(int i, int d) const {
    if (i < 0 || d < 0 || d > W || i + d > L) return false;
    int c = i * (W + 1) + d;
    return (okbits[c >> 5] >> (c & 31)) & 1u;
  }
  void set_ok(int i, int d) { int c = i * (W + 1) + d; okbits[c >> 5] |= 1u << (c & 31); }

  SeqView view() {
    SeqView q;
    q.L = L; q.W = W; q.C = C;
    okbits_end.assign(okbits.size() + 2, 0u);   // (k_mask_by_end on the GPU)
    for (int i = 0; i <= L; ++i)
      for (int d = 1; d <= W && i + d <= L; ++d)
        if (ok(i, d)) { const int c = (i + d) * (W + 1) + d; okbits_end[c >> 5] |= 1u << (c & 31); }
    q.okbits_end = okbits_end.data();
    q.seq = seq.data(); q.ws = ws.data(); q.okbits = okbits.data(); q.dmin = dmin.data(); q.unp = unp.data();
    q.e_stack = e_stack.data(); q.e_ext = e_ext.data(); q.e_ml = e_ml.data(); q.e_close = e_close.data(); q.e_hp = e_hp.data();
    q.items = items.data(); q.by_outer_off = by_outer_off.data();
    q.by_inner_off = by_inner_off.data(); q.by_inner_idx = by_inner_idx.data();
    q.by_left_off = by_left_off.data(); q.by_left_idx = by_left_idx.data();
    q.by_right_off = by_right_off.data(); q.by_right_idx = by_right_idx.data();
    q.item_in = item_in.data();
    return q;
  }
};

void plan_init(const Emu& E, HostPlan& P, const uint8_t* seq, int L, const uint8_t* qual, const char* fix) {
  P.L = L;
  P.W = std::min(L, E.max_span);
  P.C = std::min(P.W - 2 - ((E.flags & F_NO_TURN) ? 2 : 5), E.max_iloop);  // energy_model.hpp:271-273
  P.seq.assign(seq, seq + L);
  P.ws.assign(L + 1, 0.);
  if (qual) position_weights(qual, L + 1, P.ws.data());
  P.unp.assign(L + 1, 1);
  P.have_fix = (E.flags & F_FIX_RSS) && fix;
  if (P.have_fix) {
    P.ndot.assign(L + 1, 0);
    nondot_prefix(fix, L, P.ndot.data());
    for (int p = 0; p < L; ++p) P.unp[p] = fix[p] == '.';
  }
  P.okbits.assign(((size_t)(L + 1) * (P.W + 1) + 31) / 32, 0u);
}

// canonical mask / fixed structure (energy_model.hpp:213-247)
int plan_mask(const Emu& E, HostPlan& P, const char* fix) {
  const int min_span = (E.flags & F_NO_TURN) ? 1 : 5;
  int total = 0;
  for (int i = 0; i <= P.L; ++i)
    for (int d = min_span; d <= P.W && i + d <= P.L; ++d)
      if (canonical_pair(P.seq.data(), P.L, P.W, min_span, i, d)) { ++total; if (!P.have_fix) P.set_ok(i, d); }
  if (P.have_fix) {
    std::vector<int> st;
    for (int p = 0; p < P.L; ++p) {
      if (fix[p] == '(') st.push_back(p);
      else if (fix[p] == ')') {
        if (st.empty()) throw std::runtime_error("bad rss");
        int o = st.back(); st.pop_back();
        if (p + 1 - o > P.W) throw std::runtime_error("fixed pair exceeds max span");
        P.set_ok(o, p + 1 - o);
      } else if (fix[p] != '.') throw std::runtime_error("bad rss char");
    }
  }
  return total;
}

// everything derived from the pair mask
void plan_finish(const Emu& E, HostPlan& P) {
  const int L = P.L, W = P.W;
  const size_t nc = (size_t)(L + 1) * (W + 1);
  PlanCfg cfg{(E.flags & F_NO_ENE) ? 1 : 0, (E.flags & F_NO_TURN) ? 1 : 5, P.have_fix ? 1 : 0};
  const int32_t* ndot = P.have_fix ? P.ndot.data() : nullptr;
  P.dmin.assign(L + 1, 0);
  P.n_pairs = 0;
  for (int i = 0; i <= L; ++i)
    for (int d = 1; d <= W && i + d <= L; ++d)
      if (P.ok(i, d)) { if (!P.dmin[i]) P.dmin[i] = (int16_t)d; ++P.n_pairs; }
  P.e_stack.assign(nc, NEG); P.e_ext.assign(nc, NEG); P.e_ml.assign(nc, NEG); P.e_close.assign(nc, NEG); P.e_hp.assign(nc, NEG);
  for (int i = 0; i <= L; ++i)
    for (int d = 1; d <= W && i + d <= L; ++d)
      if (P.ok(i, d)) {
        PairTerms t = pair_terms(E.et, cfg, P.seq.data(), L, ndot, i, d, P.ok(i + 1, d - 2));
        size_t c = (size_t)i * (W + 1) + d;
        P.e_stack[c] = t.stack; P.e_ext[c] = t.ext; P.e_ml[c] = t.ml; P.e_close[c] = t.close; P.e_hp[c] = t.hp;
      }
  // interior-loop items, CSR by outer E cell
  P.items.clear(); P.item_in.clear();
  P.by_outer_off.assign(nc + 1, 0);
  auto okfn = [&](int i, int d) { return P.ok(i, d); };
  for (int i = 0; i <= L; ++i)
    for (int d = 0; d <= W; ++d) {
      size_t c = (size_t)i * (W + 1) + d;
      P.by_outer_off[c] = (int32_t)P.items.size();
      if (i + d > L) continue;
      if (!(i > 0 && d + 2 <= W && P.ok(i - 1, d + 2))) continue;
      enum_interior(E.et, cfg, P.seq.data(), L, W, P.C, ndot, okfn, i, d, [&](int k, int l, double tsc, bool in) {
        LoopItem it; it.tsc = tsc; it.i = (int16_t)i; it.j = (int16_t)(i + d); it.k = (int16_t)k; it.l = (int16_t)l;
        P.items.push_back(it); P.item_in.push_back(in ? 1 : 0);
      });
    }
  P.by_outer_off[nc] = (int32_t)P.items.size();
  {  // the product's plan builder enumerates from the END-major mask (enum_interior_by_end): same items, same order
    std::vector<uint32_t> endbits((nc + 31) / 32 + 2, 0u);
    for (int l = 0; l <= L; ++l)
      for (int d = 0; d <= W && d <= l; ++d)
        if (P.ok(l - d, d)) { size_t c = (size_t)l * (W + 1) + d; endbits[c >> 5] |= 1u << (c & 31); }
    auto word = [&](int n) { return (size_t)n < endbits.size() ? endbits[n] : 0u; };
    size_t n = 0;
    for (int i = 0; i <= L; ++i)
      for (int d = 0; d <= W && i + d <= L; ++d) {
        if (!(i > 0 && d + 2 <= W && P.ok(i - 1, d + 2))) continue;
        if (!ndot && loop_tables_finite(E.et) && std::find(P.seq.begin(), P.seq.end(), (uint8_t)0) == P.seq.end()) {   // the count pass of the product (popcounts): as many as the enumeration visits
          size_t m = 0;
          enum_interior_by_end(E.et, cfg, P.seq.data(), L, W, P.C, ndot, word, i, d, [&](int, int, double, bool) { ++m; });
          if ((size_t)count_interior_by_end(cfg.no_ene != 0, L, W, P.C, word, i, d) != m) ++g_enum_mismatch;
        }
        enum_interior_by_end(E.et, cfg, P.seq.data(), L, W, P.C, ndot, word, i, d, [&](int k, int l, double tsc, bool in) {
          if (n >= P.items.size()) { ++g_enum_mismatch; return; }
          const LoopItem& it = P.items[n];
          if (it.i != i || it.j != i + d || it.k != k || it.l != l || it.tsc != tsc || (P.item_in[n] != 0) != in) ++g_enum_mismatch;
          ++n;
        });
      }
    if (n != P.items.size()) ++g_enum_mismatch;
    ++g_enum_checked;
  }
  // secondary orderings (ascending item index inside a key)
  auto build = [&](std::vector<int32_t>& off, std::vector<int32_t>& idx, auto key) {
    off.assign(nc + 1, 0);
    for (auto const& it : P.items) off[key(it) + 1]++;
    for (size_t c = 0; c < nc; ++c) off[c + 1] += off[c];
    idx.assign(P.items.size(), 0);
    std::vector<int32_t> cur(off.begin(), off.end() - 1);
    for (size_t n = 0; n < P.items.size(); ++n) idx[cur[key(P.items[n])]++] = (int32_t)n;
  };
  build(P.by_inner_off, P.by_inner_idx, [&](const LoopItem& it) { return (size_t)it.k * (W + 1) + (it.l - it.k); });
  build(P.by_left_off, P.by_left_idx, [&](const LoopItem& it) { return (size_t)it.i * (W + 1) + (it.k - it.i); });
  build(P.by_right_off, P.by_right_idx, [&](const LoopItem& it) { return (size_t)it.l * (W + 1) + (it.j - it.l); });
}

ModelView make_view(const AutomatonLayout& lay, const std::vector<int32_t>& ints, const double* theta, double l0, double l1,
                    double log_tau, bool no_prf, bool no_turn) {
  ModelView m(lay);
  m.ints = ints.data(); m.big = ints.data(); m.theta = theta;
  m.lambda[0] = l0; m.lambda[1] = l1; m.log_tau = log_tau;
  m.lam_same = (l0 == l1);
  m.no_prf = no_prf;
  m.m_min = no_turn ? 4 : 10;
  return m;
}

struct Tab {
  std::vector<double> band, ext, ap;
  TableView v;
  Tab(int L, int W, int S, int nA = 0) : band((size_t)7 * (W + 1) * (L + 1) * S, NEG), ext((size_t)(L + 1) * S, NEG),
                                         ap((size_t)(W + 1) * (L + 1) * nA + 1, 0.) {
    v.band = band.data(); v.ext = ext.data(); v.L = L; v.W = W; v.S = S;
    v.ap = ap.data(); v.nA = nA;
  }
};

// compact tables of the scaled-linear rules (TableView::ld / st): every entry starts as NaN, so that a read of an entry
// nobody stored -- a dead cell or a state without a column -- poisons the result instead of passing as a structural zero
struct LinTab {
  std::vector<double> band, ext, ap;
  TableView v;
  LinTab(int L, int W, const AutomatonLayout& A, const int32_t* ints)
      : band((size_t)(W + 1) * (L + 1) * A.tab_row + 1), ext((size_t)(L + 1) * A.S, 0.), ap((size_t)(W + 1) * (L + 1) * A.ap_rs + 1) {
    v.band = band.data(); v.ext = ext.data(); v.L = L; v.W = W; v.S = A.S;
    v.ap = ap.data(); v.nA = A.n_ap;
    v.set_compact(A, ints);
    poison();
  }
  void poison() {
    std::fill(band.begin(), band.end(), std::numeric_limits<double>::quiet_NaN());
    std::fill(ap.begin(), ap.end(), std::numeric_limits<double>::quiet_NaN());
    std::fill(ext.begin(), ext.end(), 0.);
  }
};

template <bool CON> void run_inside(const ModelView& m, const SeqView& q, Tab& T, const Constraint& c) {
  const int S = m.lay.S;
  for (int d = 0; d <= q.W; ++d)
    for (int i = 0; i + d <= q.L; ++i)
      for (int s = 0; s < S; ++s) inside_target<CON>(m, q, T.v, c, d, i, s);
  for (int s = 0; s < S; ++s) T.v.o(0, s) = (s == m.lay.s00) ? 0. : NEG;
  for (int j = 1; j <= q.L; ++j)
    for (int s = 0; s < S; ++s) inside_ext_target<CON>(m, q, T.v, c, j, s);
}

struct CpuSink {
  double* EN; double* EH; double* post[3];
  void en(int idx, double w) { EN[idx] += w; }
  void eh(int k, double w) { EH[k] += w; }
  void pos(int which, int p, double z) { if (post[which]) post[which][p] = lse2(post[which][p], z); }
};


// ---- Table-driven band targets, serial: the FAST branches of k4_in / k4_out (lin_kernels.hip) restated per cell on top of the
// same records of the fast blobs -- pair records (fpr_*), tuple column records (fqc_*), live-state lists, unary programs,
// weight tables, cell records, ScanFlag words -- and the same lin_fast.h functions the kernels call.  What differs from the GPU is
// only the order of the sums.  m.ints = the whole blob (the kernels index their staged copy with the same offsets).
template <bool CON>
void fast_inside_cell(const ModelView& m, const SeqView& q, const TableView& T, int d, int i, const Constraint& con) {
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.ints;
  const double* lin = m.lin;
  const int NL = A.n_lane, j = i + d, W1 = q.W + 1, nq = A.n_quad;
  ++g_fast_cells;
  std::vector<double> hb(NL, 0.), he(NL, 0.);
  const int dmi = q.dmin[i];
  // pair phase (rule 2 factorised): k4_in, FAST branch
  for (int p = 0; p < A.n_ap; ++p) {
    const int32_t* PR = G + A.fpr_in + 8 * p;
    const int r0 = PR[0], r1 = PR[1];
    const int c1 = fcol(r0, 0), cP = fcol(r0, 1), tg = (r0 >> 16) & 0xff;
    double av = 0.;
    if (dmi > 0 && dmi < d) {
      const int nch = (dmi < d - 1 && q.unp[j - 1]) ? (r1 >> 16) & 15 : 0;
      const double wt = (r0 & (2 << 24)) ? q.ews[j - 1] : 1.;
      const int bj = q.seq[j - 1];
      for (int u = 0; u < nch; ++u) {
        const int ce = PR[2 + u], id = (ce >> 8) & 0x7fff;
        if (CON && j - 1 == con.ys && !(G[A.fs_in + id] & SF_SR)) continue;   // allow_right
        av = fma(T.lda(d - 1, i, ce & 0xff, true), lin[A.lin_wr + 5 * id + bj] * wt, av);
      }
      for_mask_bits(q.okbits_end, j * W1, 1, d - dmi, [&](int sp) {
        av = fma(T.ldc(ST_1, d - sp, i, c1, true), T.ldc(ST_B, sp, j - sp, cP, true), av);   // (X = P exp(lambda e_ml) in the B rows)
      });
      T.a(d, i, p) = av;
    }
    if (tg != 0xff && av != 0.) hb[tg] += av;
  }
  // item sums (rule 6c): one record, all tuples through their column records
  if (q.e_ok(i, d)) {
    const int cell = q.cell(i, d);
    for (int n = q.by_outer_off[cell]; n < q.by_outer_off[cell + 1]; ++n) {
      if (!q.item_in[n]) continue;
      const LoopItem it = q.items[n];
      const uint32_t rP = T.cidx(ST_P, it.l - it.k, it.k, 0), rL1 = T.cidx(ST_L, it.k - i, i, 0), rL2 = T.cidx(ST_L, j - it.l, it.l, 0);
      const double xw0 = lin_weight(m.lambda[0], it.tsc), xw1 = lin_weight(m.lambda[1], it.tsc);
      for (int t = 0; t < nq; ++t) {
        const int qa = G[A.fqc_in + 2 * t], qb = G[A.fqc_in + 2 * t + 1];
        if (qb & (4 << 16)) continue;
        const double term = T.band[rP + (qa & 0xff)] * (T.band[rL1 + ((qa >> 8) & 0xff)] * T.band[rL2 + ((qa >> 16) & 0xff)]) * ((qb & (1 << 16)) ? xw1 : xw0);
        if (term != 0.) he[qb & 0xffff] += term;
      }
    }
  }
  // cell record + unary programs
  double crec[kCellInD];
  for (int k = 0; k < 8; ++k) {
    const bool on = k < 4 ? q.pair_ok(i, d) : q.e_ok(i, d);
    crec[2 + k] = on ? cell_in_fetch(q, d, i, k) : 0.;
  }
  crec[0] = q.ews[i];
  crec[1] = q.ews[j > 0 ? j - 1 : 0];
  int fl = cell_in_flags(m, q, d, i);
  if (CON) fl |= (i == con.ys ? CF_YL : 0) | (j - 1 == con.ys ? CF_YR : 0);
  for (int l = 0; l < NL; ++l) {
    const int s = G[A.f_live_in + l];
    fast_inside_unary<kFastR, kFastP, kFastL, CON>(A, G + A.fp_in + s * kFastW, lin, T, crec, fl, d, i, &hb[l], &he[l], 1, 0, G + A.fs_in);
  }
}

// rule-7 term of every pair cell, written into its outside P entry before the sweep (k4_r7)
void fast_rule7(const ModelView& m, const SeqView& q, const TableView& in, const TableView& out) {
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.big;
  for (int i = 0; i <= q.L; ++i)
    for (int d = 0; d <= q.W && i + d <= q.L; ++d) {
      if (!q.pair_ok(i, d)) continue;
      const double x0 = xw_cell(q, 0, XT_EXT, q.cell(i, d)), x1 = xw_cell(q, 1, XT_EXT, q.cell(i, d));
      for (int s = 0; s < A.S; ++s) {
        double acc = 0.;
        for (int u = G[A.split2_off + s]; u < G[A.split2_off + s + 1]; ++u) {
          const int par = G[A.split2_ent + 2 * u], s2i = G[A.split2_ent + 2 * u + 1];
          acc = fma(out.o(i + d, par), in.o(i, s2i) * (lamk(m, par) ? x1 : x0), acc);
        }
        out.st(ST_P, d, i, s, acc);
      }
    }
}

// One cell of an outside sweep: k4_out, FAST branch.  x0: world 0 (its 1/Z and statistics); x1: world 1 = the shadow copy of
// (0,0) in the merged sweep (same tables, its own 1/Z and statistics), or null.
template <int MODE, class Sink>
void fast_outside_cell(LinOutCtx<Sink>& x0, LinOutCtx<Sink>* x1, int d, int i, int ys) {
  const ModelView& m = x0.m;
  const SeqView& q = x0.q;
  const TableView& in = x0.in;
  const TableView& out = x0.out;
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.ints;
  const double* lin = m.lin;
  const bool merged = x1 != nullptr;
  const int NL = A.n_lane, L = q.L, W = q.W, j = i + d, W1 = W + 1, nq = A.n_quad;
  if (MODE == OUT_END && j <= ys) return;      // (cells that end at or before Ys take no part: see k4_out)
  ++g_fast_cells;
  std::vector<double> h(4 * NL, 0.);
  double* h1 = h.data();
  double* h2 = h1 + NL;
  double* hp = h2 + NL;
  double* hl = hp + NL;
  const int dmi = q.dmin[i];
  // H1: stems (j, l) that start at the cell's end
  for (int p = 0; p < A.n_ap; ++p) {
    const int32_t* PR = G + A.fpr_out + 8 * p;
    const int s1 = PR[1] & 0xff, cP = fcol(PR[0], 1);
    if (dmi > 0 && dmi <= d) {
      const int hi = (W - d < L - j) ? W - d : L - j;
      double acc = 0.;
      for_mask_bits(q.okbits, j * W1, 1, hi, [&](int sp) { acc = fma(out.lda(d + sp, i, p, true), in.ldc(ST_B, sp, j, cP, true), acc); });
      if (acc != 0.) h1[s1] += acc;
    }
  }
  // HA where the cell itself is a stem
  if (q.pair_ok(i, d))
    for (int b = 1; b <= W - d; ++b) {
      const int ii = i - b;
      const int dmii = ii >= 0 ? (int)q.dmin[ii] : 0;
      if (!(dmii > 0 && b >= dmii)) continue;
      for (int p = 0; p < A.n_ap; ++p) {
        const int32_t* PR = G + A.fpr_out + 8 * p;
        const double term = out.lda(d + b, ii, p, true) * in.ldc(ST_1, b, ii, fcol(PR[0], 0), true);
        if (term != 0.) h2[(PR[1] >> 8) & 0xff] += term;
      }
    }
  // item sums of the three roles
  for (int role = 0; role < 3; ++role) {
    if (role == 0 && !q.pair_ok(i, d)) continue;
    if (role != 0 && d == 0) continue;
    const int cell = q.cell(i, d);
    const int32_t* off = role == 0 ? q.by_inner_off : role == 1 ? q.by_left_off : q.by_right_off;
    const int32_t* idx = role == 0 ? q.by_inner_idx : role == 1 ? q.by_left_idx : q.by_right_idx;
    for (int n = off[cell]; n < off[cell + 1]; ++n) {
      const LoopItem it = q.items[idx[n]];
      const uint32_t rE = out.cidx(ST_E, it.j - it.i, it.i, 0);
      const uint32_t rPi = in.cidx(ST_P, it.l - it.k, it.k, 0);
      const uint32_t r1 = role == 0 ? in.cidx(ST_L, i - it.i, it.i, 0) : rPi;
      const uint32_t r2 = role == 0 ? in.cidx(ST_L, it.j - j, j, 0) : role == 1 ? in.cidx(ST_L, it.j - it.l, it.l, 0) : in.cidx(ST_L, it.k - it.i, it.i, 0);
      const uint32_t rA = role == 0 ? in.cidx(ST_P, d, i, 0) : in.cidx(ST_L, d, i, 0);
      const double xw0 = lin_weight(m.lambda[0], it.tsc), xw1 = lin_weight(m.lambda[1], it.tsc);
      double* hrow = role == 0 ? hp : hl;
      const int qc0 = A.fqc_out + role * 2 * nq;
      for (int t = 0; t < nq; ++t) {
        const int qa = G[qc0 + 2 * t], qb = G[qc0 + 2 * t + 1];
        if (qb & (4 << 16)) continue;
        const double a0 = out.band[rE + (qa & 0xff)], a1 = in.band[r1 + ((qa >> 8) & 0xff)], a2 = in.band[r2 + ((qa >> 16) & 0xff)];
        const double aux = role == 0 ? in.band[rA + ((qa >> 24) & 0xff)] : 1.;
        const double term = a0 * (a1 * a2) * ((qb & (1 << 16)) ? xw1 : xw0);
        if (aux == 0. || term == 0.) continue;
        hrow[qb & 0xffff] += term;
        if (MODE == OUT_TRAIN && role == 0) {
          const bool w1 = merged && (qb & (2 << 16));
          const bool k1 = !m.lam_same && (qb & (1 << 16));
          LinOutCtx<Sink>& xw = w1 ? *x1 : x0;
          xw.sink.eh(k1 ? 1 : 0, it.tsc * term * (aux * xw.invZ));
        }
      }
    }
  }
  // cell record + unary programs; out B of the targets for the pair entries
  double crec[kCellOutD];
  for (int k = 0; k < 12; ++k) {
    const bool on = k < 4 ? q.e_ok(i, d) : k < 6 ? q.pair_ok(i, d) : k < 8 ? (q.pair_ok(i - 1, d + 2) && q.pair_ok(i, d)) : true;
    crec[2 + k] = on ? cell_out_fetch(q, d, i, k) : 0.;
  }
  crec[0] = q.ews[i > 0 ? i - 1 : 0];
  crec[1] = q.ews[j < L ? j : L];
  int fl = cell_out_flags(m, q, d, i);
  if (MODE == OUT_END) fl |= (i - 1 == ys ? CF_YL : 0) | (j == ys ? CF_YR : 0) | (L == j + 1 ? CF_JLAST : 0);
  std::vector<double> oB(NL, 0.);
  for (int l = 0; l < NL; ++l) {
    const int s = G[A.f_live_out + l];
    LinOutCtx<Sink>& xw = (merged && s == A.shadow) ? *x1 : x0;
    oB[l] = fast_outside_unary<kFastR, kFastP, kFastL, MODE>(A, G + A.fp_out + s * kFastW, G, lin, in, out, crec, fl, d, i, xw.invZ, m.lam_same != 0,
                                                             m.no_prf != 0, xw.sink, h1 + l, NL, 1, 0, G + A.fs_out);
  }
  // pair entries (lin_outside_apair from the pair record)
  for (int p = 0; p < A.n_ap; ++p) {
    const int32_t* PR = G + A.fpr_out + 8 * p;
    const int r0 = PR[0], r1 = PR[1];
    const int tg = (r0 >> 16) & 0xff;
    if (!(dmi > 0 && dmi < d)) continue;
    LinOutCtx<Sink>& xw = (merged && (r0 & (4 << 24))) ? *x1 : x0;
    const bool step = d + 1 <= W && j < L && q.unp[j];
    const int nr = step ? (r1 >> 20) & 15 : 0;
    const double a_in = in.a(d, i, p);
    double acc = 0.;
    if (a_in != 0.) {
      acc = tg != 0xff ? oB[tg] : 0.;
      const double inz = a_in * xw.invZ;
      const int bj = step ? (int)q.seq[j] : 0;
      const double ewj = step ? q.ews[j] : 1.;
      for (int u = 0; u < nr; ++u) {
        const int ce = PR[5 + u], id = (ce >> 8) & 0x7fff;
        const double term = out.lda(d + 1, i, ce & 0xff, true) * (lin[A.lin_wr + 5 * id + bj] * ((G[A.fe_r + 2 * id + 1] & 1) ? ewj : 1.));
        const double z = term * inz;
        if ((MODE == OUT_SCAN || MODE == OUT_END) &&
            !fast_scan_stat<MODE, false, true>(xw.sink, G[A.fs_out + id], i - 1, j, false, MODE == OUT_END && j == ys, L == j + 1, z))
          continue;
        if (MODE != OUT_END && !m.no_prf && z != 0. && bj) xw.sink.en(G[A.fe_r + 2 * id] + bj, z);
        acc += term;
      }
    }
    out.a(d, i, p) = acc;
  }
}

template <int MODE>
void run_outside(const ModelView& m, const SeqView& q, Tab& in, Tab& out, double Z, const Constraint& c, CpuSink& sink,
                 bool ari, bool nasi) {
  const int S = m.lay.S;
  OutCtx<CpuSink> x{m, q, in.v, out.v, Z, c, sink};
  for (int s = 0; s < S; ++s) out.v.o(q.L, s) = NEG;
  if (nasi) out.v.o(q.L, m.lay.s00) = 0.;
  if (ari) { out.v.o(q.L, m.lay.s0m1) = 0.; out.v.o(q.L, m.lay.s0m2) = 0.; }
  for (int i = q.L - 1; i >= 0; --i)
    for (int s = 0; s < S; ++s) outside_ext_target<MODE>(x, i, s);
  for (int d = q.W; d >= 0; --d)
    for (int i = 0; i + d <= q.L; ++i)
      for (int s = 0; s < S; ++s) outside_target<MODE>(x, d, i, s);
}

// BPP filter through the one-state automaton (energy_model.hpp:188-266)
void bpp_filter(const Emu& E, HostPlan& P, int total, std::vector<double>* lnbpp_out, double* lnZ, double* eff) {
  plan_finish(E, P);
  ModelView m0 = make_view(E.lay0, E.ints0, nullptr, 1., 1., 0., true, E.flags & F_NO_TURN);
  SeqView q = P.view();
  std::vector<double> ws0(P.L + 1, 0.);
  q.ws = ws0.data();
  Tab in(P.L, P.W, 1), out(P.L, P.W, 1);
  Constraint c{-1, -1, 0};
  run_inside<false>(m0, q, in, c);
  double Z = in.v.o(P.L, 0);
  double en[1] = {0}, eh[2] = {0, 0};
  CpuSink sink{en, eh, {nullptr, nullptr, nullptr}};
  run_outside<OUT_NONE>(m0, q, in, out, Z, Constraint{-1, -1, 0}, sink, false, true);
  const double thr = std::log(E.min_bpp);
  std::vector<uint32_t> keep(P.okbits.size(), 0u);
  int nbp = 0;
  if (lnbpp_out) lnbpp_out->assign((size_t)(P.L + 1) * (P.W + 1), NEG);
  for (int i = 0; i <= P.L; ++i)
    for (int d = 1; d <= P.W && i + d <= P.L; ++d)
      if (P.ok(i, d)) {
        double ln = (in.v.at(ST_P, d, i, 0) + out.v.at(ST_P, d, i, 0)) - Z;  // energy_model.hpp:195-201
        if (lnbpp_out) (*lnbpp_out)[(size_t)i * (P.W + 1) + d] = ln;
        if (thr <= ln) { int c2 = i * (P.W + 1) + d; keep[c2 >> 5] |= 1u << (c2 & 31); ++nbp; }
      }
  if (lnZ) *lnZ = Z;
  if (E.min_bpp > 0) { P.okbits.swap(keep); if (eff) *eff = (double)nbp / (double)total; }
  else if (eff) *eff = 1.0;
}

// full preparation of one sequence as elemdp_load_batch does it
double prepare(const Emu& E, HostPlan& P, const uint8_t* seq, int L, const uint8_t* qual, const char* fix) {
  plan_init(E, P, seq, L, qual, fix);
  int total = plan_mask(E, P, fix);
  double eff = 1.0;
  if (E.flags & F_NO_RSS) {   // --no-rss: no pair is ever parsable, as Engine::load_batch clears the mask (motif_model.hpp:57, 171-206)
    std::fill(P.okbits.begin(), P.okbits.end(), 0u);
    eff = 0.;
  } else if (P.have_fix) {
    int nbp = 0;
    for (int i = 0; i <= L; ++i) for (int d = 1; d <= P.W && i + d <= L; ++d) nbp += P.ok(i, d);
    eff = (double)nbp / (double)total;
  } else if (E.min_bpp > 0) {
    bpp_filter(E, P, total, nullptr, nullptr, &eff);
  }
  plan_finish(E, P);
  return eff;
}

thread_local std::string g_err;

}  // namespace

extern "C" {
// plans built so far / differences between the two interior-loop enumerators (must stay 0)
// The candidate table of the BPP filter (bpp_cand.h) against loop_weight, shape by shape and context by context: every (u1, u2) with
// u1 + u2 <= kMaxLoop except (0, 0) is either one of the eight special shapes or exactly one table entry whose coefficient times the
// closing pair's factor times the inner pair's factor is loop_weight (to a few ulp: the same products in another order); the runs
// of the generic class repeat the entries' coefficients.  Returns the number of violations; *max_rel = largest relative deviation.
long emu_check_bpp_cand(void* h, double* max_rel) {
  Emu& E = *static_cast<Emu*>(h);
  std::unique_ptr<EnergyTables> xp(new EnergyTables);
  exp_tables(E.et, xp.get());
  const EnergyTables& x = *xp;
  std::unique_ptr<BppCandTable> tp(new BppCandTable);
  build_bpp_cand(x, tp.get());
  const BppCandTable& t = *tp;
  long bad = 0;
  double worst = 0.;
  static const char pl[7][2] = {{0, 0}, {2, 3}, {3, 2}, {3, 4}, {4, 3}, {1, 4}, {4, 1}};   // bases of pair type 1 .. 6 (bp_type)
  for (int ty = 1; ty <= 6; ++ty)
    if (bp_type(pl[ty][0], pl[ty][1]) != ty) ++bad;
  int seen[kMaxLoop + 1][kMaxLoop + 1] = {};
  for (int c = 0; c < BC_CLASSES; ++c) {
    const int n = t.upto[c][kMaxLoop];
    for (int k = 0; k < n; ++k) {
      const BppCand& e = t.e[t.base[c] + k];
      if (e.T < 0 || e.T > kMaxLoop || e.u1 < 0 || e.u1 > e.T) { ++bad; continue; }
      seen[e.u1][e.T - e.u1] += 1 + 10 * c;
      if (k > 0 && t.e[t.base[c] + k - 1].T > e.T) ++bad;                       // sorted by T
      if (t.upto[c][e.T] <= k || (e.T > 0 && t.upto[c][e.T - 1] > k)) ++bad;    // the prefix counts
    }
  }
  for (int u1 = 0; u1 <= kMaxLoop; ++u1)
    for (int u2 = 0; u1 + u2 <= kMaxLoop; ++u2) {
      bool special = (u1 == 0 && u2 == 0);
      for (int k = 0; k < kBppSpecial; ++k) special = special || (kBppSpecialU1[k] == u1 && kBppSpecialU2[k] == u2);
      if (special) { if (seen[u1][u2] != 0) ++bad; continue; }
      const int c = (seen[u1][u2] - 1) / 10;
      if (seen[u1][u2] != 1 + 10 * c || c < 0 || c >= BC_CLASSES) { ++bad; continue; }
      double coef = 0.;
      for (int k = 0; k < t.upto[c][kMaxLoop]; ++k) { const BppCand& e = t.e[t.base[c] + k]; if (e.u1 == u1 && e.T == u1 + u2) coef = e.coef; }
      if (c == BC_I) {
        const int T = u1 + u2;
        if (T < kBppRunMin || t.run_coef[t.run_off[T] + (u1 - 2)] != coef || t.quad_T[(t.run_off[T] + (u1 - 2)) / 4] != T) ++bad;
      }
      // closing pair (i, j), inner pair (p, q); the four neighbours get every base incl. N
      uint8_t s[2 * kMaxLoop + 16];
      const int i = 0, pp = u1 + 1, qq = pp + 3, j = qq + u2 + 1;
      for (int ty = 1; ty <= 6; ++ty)
        for (int ty2 = 1; ty2 <= 6; ++ty2)
          for (int nb = 0; nb < 625; ++nb) {
            for (int z = 0; z <= j; ++z) s[z] = 1;
            s[i + 1] = nb % 5; s[j - 1] = (nb / 5) % 5; s[pp - 1] = (nb / 25) % 5; s[qq + 1] = nb / 125;
            s[i] = pl[ty][0]; s[j] = pl[ty][1]; s[qq] = pl[ty2][0]; s[pp] = pl[ty2][1];     // type2 = bp_type(s[q], s[p])
            const double ref = loop_weight(x, s, i, j, pp, qq);
            const int mo = ty * 25 + s[i + 1] * 5 + s[j - 1], mi = ty2 * 25 + s[qq + 1] * 5 + s[pp - 1];
            const double fo = c == BC_I ? x.mismatch_i[mo] : c == BC_N ? x.mismatch_1ni[mo] : (is_au(ty) ? x.term_au : 1.);
            const double fi = c == BC_I ? x.mismatch_i[mi] : c == BC_N ? x.mismatch_1ni[mi] : (is_au(ty2) ? x.term_au : 1.);
            const double got = coef * fo * fi;
            const double rel = ref == got ? 0. : std::fabs(got - ref) / std::max(std::fabs(ref), 1e-300);
            worst = std::max(worst, rel);
            if (!(rel <= 1e-14)) ++bad;
          }
    }
  if (max_rel) *max_rel = worst;
  return bad;
}
long emu_enum_checked() { return g_enum_checked; }
long emu_enum_mismatches() { return g_enum_mismatch; }
long emu_fast_cells() { return g_fast_cells; }


const char* emu_last_error() { return g_err.c_str(); }

void* emu_create(const char* pattern, const char* par_text, int max_span, int max_iloop, double min_bpp, double tau,
                 int flags) {
  try {
    Emu* E = new Emu();
    E->au = new Automaton(pattern);
    parse_energy_text(par_text, &E->et);
    E->au->flatten(&E->lay, &E->ints);
    E->au->flatten(&E->lay_r, &E->ints_r, true);
    E->au->flatten(&E->lay_s, &E->ints_s, false, false, true);
    flatten_trivial(&E->lay0, &E->ints0);
    E->max_span = max_span; E->max_iloop = max_iloop; E->min_bpp = min_bpp; E->tau = tau; E->flags = flags;
    return E;
  } catch (std::exception& e) { g_err = e.what(); return nullptr; }
}
// re-flattens the automaton with / without the static pruning of the transition lists (Automaton::flatten)
void emu_set_prune(void* h, int prune) {
  Emu* E = (Emu*)h;
  E->au->flatten(&E->lay, &E->ints, false, prune != 0);
  E->au->flatten(&E->lay_r, &E->ints_r, true, prune != 0);
  E->au->flatten(&E->lay_s, &E->ints_s, false, prune != 0, true);
}
// 1: the table-driven forms of the band targets (what k4_in / k4_out run with FAST); returns whether the automaton's lists fit
// the programs (AutomatonLayout::fp_ok) -- otherwise the generic rule code runs, as on the GPU
int emu_set_fast(void* h, int fast) {
  Emu* E = (Emu*)h;
  E->fast = fast != 0;
  return (E->lay.fp_ok ? 1 : 0) | (E->lay_r.fp_ok ? 2 : 0) | (E->lay_s.fp_ok ? 4 : 0) | (E->lay_s.shadow >= 0 ? 8 : 0);
}
void emu_destroy(void* h) { delete (Emu*)h; }
int emu_n_param(void* h) { return ((Emu*)h)->au->n_theta() + 2; }
int emu_n_state(void* h) { return ((Emu*)h)->au->S(); }
int emu_describe(void* h, char* buf, int cap) {
  std::string s = ((Emu*)h)->au->to_json();
  if ((int)s.size() + 1 > cap) return -1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}
int emu_energy_table(void* h, const char* name, double* out, int cap) {
  EnergyTables& e = ((Emu*)h)->et;
  struct Ent { const char* n; double* p; int c; };
  Ent tab[] = {{"stack", e.stack, 49}, {"hairpin", e.hairpin, 31}, {"bulge", e.bulge, 31}, {"internal", e.interior, 31},
               {"ninio", e.ninio, 31}, {"mismatch_h", e.mismatch_h, 175}, {"mismatch_i", e.mismatch_i, 175},
               {"mismatch_m", e.mismatch_m, 175}, {"mismatch_1ni", e.mismatch_1ni, 175}, {"mismatch_23i", e.mismatch_23i, 175},
               {"mismatch_ext", e.mismatch_ext, 175}, {"dangle5", e.dangle5, 40}, {"dangle3", e.dangle3, 40},
               {"int_11", e.int11, 1600}, {"int_21", e.int21, 8000}, {"int_22", e.int22, 40000}, {"triloop", e.triloop, 40},
               {"tetraloop", e.tetraloop, 40}, {"hexaloop", e.hexaloop, 40}, {"term_au", &e.term_au, 1},
               {"mlintern", &e.ml_intern, 1}, {"mlclosing", &e.ml_closing, 1}, {"ml_base", &e.ml_base, 1}, {"lxc37", &e.lxc37, 1}};
  for (auto& t : tab) if (!strcmp(t.n, name)) { if (cap < t.c) return -t.c; std::copy(t.p, t.p + t.c, out); return t.c; }
  return 0;
}
double emu_hairpin_energy(void* h, const uint8_t* seq, int L, int i, int j) { return hairpin_energy(((Emu*)h)->et, seq, i, j); }
double emu_loop_energy(void* h, const uint8_t* seq, int L, int i, int j, int p, int q) { return loop_energy(((Emu*)h)->et, seq, i, j, p, q); }
// exp(loop_energy) from the exponentiated tables (what the BPP filter kernels use; energy_rules.h: loop_weight)
double emu_loop_weight(void* h, const uint8_t* seq, int L, int i, int j, int p, int q) {
  static thread_local EnergyTables x;
  exp_tables(((Emu*)h)->et, &x);
  return loop_weight(x, seq, i, j, p, q);
}
// ScanFlag word of every forward transition (right | left | pair) with the nodes of its parent and child state: for the test
// that re-derives the flags from the scanner's node conditions.  out: rows of 6 ints {kind, pl, pr, cl, cr, flags}.
int emu_scan_flags(void* h, int32_t* out, int cap) {
  Emu& E = *(Emu*)h;
  const AutomatonLayout& A = E.lay;
  const int32_t* I = E.ints.data();
  int n = 0;
  const int32_t offs[3] = {A.right_off, A.left_off, A.pair_off}, ents[3] = {A.right_ent, A.left_ent, A.pair_ent};
  for (int kind = 0; kind < 3; ++kind)
    for (int k = 0; k < A.S; ++k)
      for (int t = I[offs[kind] + k]; t < I[offs[kind] + k + 1]; ++t) {
        const int ch = I[ents[kind] + 2 * t];
        if (n >= cap) return -1;
        int32_t* r = out + 6 * n;
        r[0] = kind; r[1] = I[A.st_l + k]; r[2] = I[A.st_r + k]; r[3] = I[A.st_l + ch]; r[4] = I[A.st_r + ch];
        r[5] = I[A.fs_in + n];
        if (I[A.fs_out + n] != r[5]) return -2;
        ++n;
      }
  return (n == A.n_wr + A.n_wl + A.n_wp) ? n : -3;
}
// Deterministic-mode tuple lists (AutomatonLayout::qd_* / fqd_*): every live tuple of a column-record list appears exactly once, in
// the share of the wave (target & 3); dead tuples in none.  Returns 0, or the number of the first violated rule.
static int check_det(const std::vector<int32_t>& I, int qc, int qd, int n) {
  std::vector<int> seen(n, 0);
  if (I[qd] != 0) return 1;
  for (int w = 0; w < 4; ++w) {
    if (I[qd + w + 1] < I[qd + w] || I[qd + w + 1] > n) return 2;
    for (int k = I[qd + w]; k < I[qd + w + 1]; ++k) {
      const int t = I[qd + 5 + k];
      if (t < 0 || t >= n) return 3;
      const int word = I[qc + 2 * t + 1];
      if ((word >> 16) & 4) return 4;                 // a dead tuple was dealt
      if (((word & 0xffff) & 3) != w) return 5;       // to the wrong wave
      if (seen[t]++) return 6;
    }
  }
  for (int t = 0; t < n; ++t) if (!seen[t] && !((I[qc + 2 * t + 1] >> 16) & 4)) return 7;   // a live tuple was left out
  return 0;
}
int emu_check_det_lists(void* h) {
  Emu& E = *(Emu*)h;
  const AutomatonLayout* lays[3] = {&E.lay, &E.lay_r, &E.lay_s};
  const std::vector<int32_t>* ints[3] = {&E.ints, &E.ints_r, &E.ints_s};
  for (int k = 0; k < 3; ++k) {
    const AutomatonLayout& A = *lays[k];
    const std::vector<int32_t>& I = *ints[k];
    const int nq = A.n_quad;
    if (int r = check_det(I, A.qc_in, A.qd_in, nq)) return 100 * (k + 1) + r;
    if (int r = check_det(I, A.fqc_in, A.fqd_in, nq)) return 100 * (k + 1) + 10 + r;
    for (int role = 0; role < 3; ++role) {
      if (int r = check_det(I, A.qc_out1 + role * 2 * nq, A.qd_out + role * (5 + nq), nq)) return 100 * (k + 1) + 20 + 10 * role + r;
      if (int r = check_det(I, A.fqc_out + role * 2 * nq, A.fqd_out + role * (5 + nq), nq)) return 100 * (k + 1) + 50 + 10 * role + r;
    }
    if (!(A.qd_in >= A.n_small && A.qd_in + 5 + nq <= A.big_in_end)) return 100 * (k + 1) + 91;     // staged with the inside run
    if (!(A.qd_out >= A.big_in_end && A.qd_out + 3 * (5 + nq) <= A.n_ints)) return 100 * (k + 1) + 92;
    if (!(A.fqd_in >= A.fb_in && A.fqd_in + 5 + nq <= A.fb_in + A.fb_in_n)) return 100 * (k + 1) + 93;
    if (!(A.fqd_out >= A.fb_out && A.fqd_out + 3 * (5 + nq) <= A.fb_out + A.fb_out_n)) return 100 * (k + 1) + 94;
  }
  return 0;
}
int emu_pattern_nodes(void* h) { return ((Emu*)h)->lay.M; }
double emu_sum_ext_m(void* h, const uint8_t* seq, int L, int i, int j, int ext) { return sum_ext_m(((Emu*)h)->et, seq, L, i, j, ext); }

int emu_bpp(void* h, const uint8_t* seq, int L, double* lnbpp, uint8_t* kept, double* bpp_eff, double* lnZ) {
  try {
    Emu& E = *(Emu*)h;
    HostPlan P;
    plan_init(E, P, seq, L, nullptr, nullptr);
    int total = plan_mask(E, P, nullptr);
    std::vector<double> ln;
    double eff = 1.0;
    bpp_filter(E, P, total, &ln, lnZ, &eff);
    if (lnbpp) std::copy(ln.begin(), ln.end(), lnbpp);
    if (kept) for (int i = 0; i <= L; ++i) for (int d = 0; d <= P.W; ++d) kept[i * (P.W + 1) + d] = P.ok(i, d);
    if (bpp_eff) *bpp_eff = eff;
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// out9: Zo, Zari, Znasi, f, bpp_eff, skipped, L, W, n_items
int emu_train_seq(void* h, const double* x, const uint8_t* seq, int L, const uint8_t* qual, const char* fix, double* out9,
                  double* ENo, double* EHo, double* ENx, double* EHx, double* inside_o, double* inside, double* outside,
                  double* outside_o) {
  try {
    Emu& E = *(Emu*)h;
    const int nt = E.au->n_theta();
    std::vector<double> theta(x, x + nt);
    if (E.flags & F_SOFTMAX)  // theta = log-softmax of the rows (profile_hmm.hpp:103-111)
      for (int r = 0; r < E.au->n_rows(); ++r) {
        double tot = NEG;
        for (int c = 0; c < E.au->row_width(r); ++c) tot = lse2(tot, x[E.au->row_offset(r) + c]);
        for (int c = 0; c < E.au->row_width(r); ++c) theta[E.au->row_offset(r) + c] = x[E.au->row_offset(r) + c] - tot;
      }
    ModelView m = make_view(E.lay, E.ints, theta.data(), x[nt], x[nt + 1], std::log(E.tau), E.flags & F_NO_PRF,
                            E.flags & F_NO_TURN);
    HostPlan P;
    double eff = prepare(E, P, seq, L, qual, fix);
    SeqView q = P.view();
    const int S = m.lay.S;
    Tab in(L, P.W, S), out(L, P.W, S);
    Constraint c{-1, -1, 0};
    run_inside<false>(m, q, in, c);
    double Zo = part_func(m, in.v, true, true), Za = part_func(m, in.v, true, false), Zn = part_func(m, in.v, false, true);
    out9[0] = Zo; out9[1] = Za; out9[2] = Zn; out9[3] = 0; out9[4] = eff; out9[5] = 0; out9[6] = L; out9[7] = P.W;
    out9[8] = (double)P.items.size();
    auto copy_tab = [&](Tab& T, double* dst) {  // -> reference index order [i][d][e][s]
      for (int i = 0; i <= L; ++i) for (int d = 0; d <= P.W; ++d) for (int e = 0; e < 7; ++e) for (int s = 0; s < S; ++s)
        dst[(((size_t)i * (P.W + 1) + d) * 7 + e) * S + E.ints[E.lay.st_ref + s]] = (i + d <= L) ? T.v.at(e, d, i, s) : NEG;
    };
    auto copy_ext = [&](Tab& T, double* dst) {   // (states in the reference's order, AutomatonLayout::st_ref)
      for (int j = 0; j <= L; ++j) for (int s = 0; s < S; ++s) dst[(size_t)j * S + E.ints[E.lay.st_ref + s]] = T.v.o(j, s);
    };
    if (inside_o) copy_ext(in, inside_o);
    if (inside) copy_tab(in, inside);
    if (!(std::isfinite(Zo) && std::isfinite(Za))) { out9[5] = 1; return 0; }
    std::vector<double> eno(nt + 1, 0.), enx(nt + 1, 0.);
    double eho[2] = {0, 0}, ehx[2] = {0, 0};
    CpuSink s1{eno.data(), eho, {nullptr, nullptr, nullptr}};
    run_outside<OUT_TRAIN>(m, q, in, out, Zo, c, s1, true, true);
    if (outside) copy_tab(out, outside);
    if (outside_o) copy_ext(out, outside_o);
    const bool positive = !(P.ws[L] > NEG);
    CpuSink s2{enx.data(), ehx, {nullptr, nullptr, nullptr}};
    double Zx = positive ? Za : Zn;
    run_outside<OUT_TRAIN>(m, q, in, out, Zx, c, s2, positive, !positive);
    out9[3] = Zo - Zx;
    if (ENo) std::copy(eno.begin(), eno.begin() + nt, ENo);
    if (ENx) std::copy(enx.begin(), enx.begin() + nt, ENx);
    if (EHo) { EHo[0] = eho[0]; EHo[1] = eho[1]; }
    if (EHx) { EHx[0] = ehx[0]; EHx[1] = ehx[1]; }
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// ---- scaled-linear train evaluation (lin_rules.h), same outputs as emu_train_seq (tables converted to logs) ----
// schedule 0: the reference's two outside passes; 1: ari-only pass + nasi-only pass on the one-state automaton,
// combined as k3_combine / k4_combine do.  out9[5] = 2 when the linear range was left (sequence flagged).
int emu_train_seq_lin(void* h, const double* x, const uint8_t* seq, int L, const uint8_t* qual, const char* fix, int schedule,
                      double* out9, double* ENo, double* EHo, double* ENx, double* EHx, double* inside_o, double* inside,
                      double* outside, double* outside_o) {
  try {
    Emu& E = *(Emu*)h;
    const int nt = E.au->n_theta();
    std::vector<double> theta(x, x + nt);
    if (E.flags & F_SOFTMAX)
      for (int r = 0; r < E.au->n_rows(); ++r) {
        double tot = NEG;
        for (int c = 0; c < E.au->row_width(r); ++c) tot = lse2(tot, x[E.au->row_offset(r) + c]);
        for (int c = 0; c < E.au->row_width(r); ++c) theta[E.au->row_offset(r) + c] = x[E.au->row_offset(r) + c] - tot;
      }
    const bool no_prf = E.flags & F_NO_PRF;
    // schedule 2: the train schedule of the GPU's default -- ONE outside sweep on the automaton with the shadow copy of (0,0)
    // (DESIGN.md 2.3): "has motif" terminals on the pattern's states (world 0: Z(ari), statistics A), the "no motif" terminal
    // on the shadow (world 1: Z(nasi), statistics B); combined like schedule 1.  No table export.
    const bool merged = schedule == 2;
    if (merged && (E.lay_s.shadow < 0 || inside || outside)) throw std::runtime_error("schedule 2: no shadow state / no table export");
    if (merged) inside_o = outside_o = nullptr;   // (exports are in the numbering of the plain automaton)
    const AutomatonLayout& LY = merged ? E.lay_s : E.lay;
    const std::vector<int32_t>& IY = merged ? E.ints_s : E.ints;
    std::vector<double> lin;
    make_lin_params(LY, IY.data(), theta.data(), E.tau, no_prf, &lin);
    ModelView m = make_view(LY, IY, theta.data(), x[nt], x[nt + 1], std::log(E.tau), no_prf, E.flags & F_NO_TURN);
    m.lin = lin.data();
    ModelView mr = make_view(E.lay_r, E.ints_r, theta.data(), x[nt], x[nt + 1], std::log(E.tau), no_prf, E.flags & F_NO_TURN);
    mr.lin = lin.data();
    HostPlan P;
    double eff = prepare(E, P, seq, L, qual, fix);
    SeqView q = P.view();
    const int S = m.lay.S;
    const size_t nc = (size_t)(L + 1) * (P.W + 1), ni = P.items.size();
    std::vector<double> ews(L + 1), xwc(10 * nc), xwi(2 * ni + 1);
    for (int p = 0; p <= L; ++p) ews[p] = std::exp(P.ws[p]);
    const double* terms[5] = {P.e_stack.data(), P.e_ext.data(), P.e_ml.data(), P.e_close.data(), P.e_hp.data()};
    for (int k = 0; k < 2; ++k) {
      for (int t = 0; t < 5; ++t)
        for (size_t c = 0; c < nc; ++c) xwc[(size_t)(k * 5 + t) * nc + c] = lin_weight(m.lambda[k], terms[t][c]);
      for (size_t n = 0; n < ni; ++n) xwi[(size_t)k * ni + n] = lin_weight(m.lambda[k], P.items[n].tsc);
    }
    q.ews = ews.data(); q.xwc = xwc.data(); q.xwc_stride = nc; q.xwi = xwi.data(); q.xwi_stride = ni;
    std::vector<double> cum(L + 1, 0.);   // log2 of prod_{p<j} psb
    for (int p = 0; p < L; ++p) cum[p + 1] = cum[p] + lin[kLinPl2 + seq[p]];
    const double ln2 = 0.69314718055994530942;
    LinTab in(L, P.W, m.lay, IY.data()), out(L, P.W, m.lay, IY.data());
    const bool fast = E.fast && m.lay.fp_ok;     // (table-driven forms: fast_inside_cell / fast_outside_cell)
    const Constraint c0{-1, -1, 0};
    for (int d = 0; d <= q.W; ++d)
      for (int i = 0; i + d <= q.L; ++i) {
        if (fast) { fast_inside_cell<false>(m, q, in.v, d, i, c0); continue; }
        lin_inside_cell_pairs(m, q, in.v, d, i);     // rule 2, factorised: the pair table of the cell first
        for (int s = 0; s < S; ++s) lin_inside_target(m, q, in.v, d, i, s);
      }
    for (int s = 0; s < S; ++s) in.v.o(0, s) = (s == m.lay.s00 || s == m.lay.shadow) ? 1. : 0.;   // (the shadow of (0,0) starts like it)
    for (int j = 1; j <= L; ++j)
      for (int s = 0; s < S; ++s) lin_inside_ext_target(m, q, in.v, j, s);
    const double Zo = lin_part(m, in.v, true, true), Za = lin_part(m, in.v, true, false), Zn = lin_part(m, in.v, false, true);
    auto tolog = [&](double v, double sc) { return v > 0. ? std::log(v) - sc * ln2 : NEG; };
    out9[0] = tolog(Zo, cum[L]); out9[1] = tolog(Za, cum[L]); out9[2] = tolog(Zn, cum[L]);
    out9[3] = 0; out9[4] = eff; out9[5] = 0; out9[6] = L; out9[7] = P.W; out9[8] = (double)P.items.size();
    auto copy_tab = [&](LinTab& T, double* dst, bool outside_tab) {
      for (int i = 0; i <= L; ++i) for (int d = 0; d <= P.W; ++d) for (int e = 0; e < 7; ++e) for (int s = 0; s < S; ++s) {
        double sc = (i + d <= L) ? cum[i + d] - cum[i] : 0.;
        if (outside_tab) sc = cum[L] - sc;
        dst[(((size_t)i * (P.W + 1) + d) * 7 + e) * S + E.ints[E.lay.st_ref + s]] = (i + d <= L) ? tolog(lin_get(m, q, T.v, e, d, i, s), sc) : NEG;
      }
    };
    if (inside_o) for (int j = 0; j <= L; ++j) for (int s = 0; s < S; ++s) inside_o[(size_t)j * S + E.ints[E.lay.st_ref + s]] = tolog(in.v.o(j, s), cum[j]);
    if (inside) copy_tab(in, inside, false);
    auto bad = [](double z) { return !(z > 0.) || !std::isfinite(z); };
    if (bad(Zo) || bad(Za) || (schedule >= 1 && bad(Zn))) { out9[5] = 2; return 0; }
    const bool positive = !(P.ws[L] > NEG);
    std::vector<double> enA(nt + 1, 0.), enB(nt + 1, 0.);
    double ehA[2] = {0, 0}, ehB[2] = {0, 0};
    auto run_out = [&](const ModelView& mm, double Z, bool ari, bool nasi, std::vector<double>& en, double* eh) {
      CpuSink sink{en.data(), eh, {nullptr, nullptr, nullptr}};
      LinOutCtx<CpuSink> xo{mm, q, in.v, out.v, 1. / Z, sink};
      const int NA = mm.lay.n_active;
      for (int s = 0; s < S; ++s) out.v.o(L, s) = 0.;
      if (nasi) out.v.o(L, mm.lay.s00) = 1.;
      if (ari) { out.v.o(L, mm.lay.s0m1) = 1.; out.v.o(L, mm.lay.s0m2) = 1.; }
      for (int i = L - 1; i >= 0; --i)
        for (int s = 0; s < NA; ++s) lin_outside_ext_target<OUT_TRAIN>(xo, i, s);
      const bool f = fast && &mm == &m;     // (the one-state automaton's programs index its own columns, not these tables')
      if (f) fast_rule7(mm, q, in.v, out.v);
      for (int d = q.W; d >= 0; --d)
        for (int i = 0; i + d <= L; ++i) {
          if (f) { fast_outside_cell<OUT_TRAIN>(xo, (LinOutCtx<CpuSink>*)nullptr, d, i, -1); continue; }
          for (int s = 0; s < NA; ++s) lin_outside_target<OUT_TRAIN>(xo, d, i, s);
          lin_outside_cell_pairs<OUT_TRAIN>(xo, d, i);
        }
    };
    // export only: the plane-2 outside table holds the direct part (rules 4a, 3a); the reference's value adds what arrives
    // through rule 2, HA(k,l,t) = sum_i sum_{p=(s1,t)} outA(i,l,p) 1(i,k,s1), wherever the inside value is non-zero
    auto complete_plane2 = [&]() {
      LinOutCtx<CpuSink>* none = nullptr; (void)none;
      double en0[1] = {0}, eh0[2] = {0, 0};
      CpuSink sink{en0, eh0, {nullptr, nullptr, nullptr}};
      LinOutCtx<CpuSink> xo{m, q, in.v, out.v, 1., sink};
      for (int d = 0; d <= q.W; ++d)
        for (int i = 0; i + d <= L; ++i)
          for (int t = 0; t < S; ++t)
            if (q.left_ok(i, d) && in.v.ld(ST_2, d, i, t) != 0.) out.v.st(ST_2, d, i, t, out.v.ld(ST_2, d, i, t) + lheavy_o2(xo, d, i, t));
    };
    if (merged) {
      CpuSink sA{enA.data(), ehA, {nullptr, nullptr, nullptr}}, sB{enB.data(), ehB, {nullptr, nullptr, nullptr}};
      LinOutCtx<CpuSink> x0{m, q, in.v, out.v, 1. / Za, sA}, x1{m, q, in.v, out.v, 1. / Zn, sB};
      const int NA = m.lay.n_active, sh = m.lay.shadow;
      for (int s = 0; s < S; ++s) out.v.o(L, s) = 0.;
      out.v.o(L, m.lay.s0m1) = 1.; out.v.o(L, m.lay.s0m2) = 1.; out.v.o(L, sh) = 1.;
      for (int i = L - 1; i >= 0; --i)
        for (int s = 0; s < NA; ++s) lin_outside_ext_target<OUT_TRAIN>(s == sh ? x1 : x0, i, s);
      if (fast) fast_rule7(m, q, in.v, out.v);
      for (int d = q.W; d >= 0; --d)
        for (int i = 0; i + d <= L; ++i) {
          if (fast) { fast_outside_cell<OUT_TRAIN>(x0, &x1, d, i, -1); continue; }
          for (int s = 0; s < NA; ++s) lin_outside_target<OUT_TRAIN>(s == sh ? x1 : x0, d, i, s);
          for (int p = 0; p < m.lay.n_ap; ++p) {
            const int tgt = m.ints[m.lay.ap_tgt + p];
            lin_outside_apair<OUT_TRAIN>(m.ints[m.lay.ap_t + p] == sh ? x1 : x0, d, i, p, tgt >= 0 ? out.v.ld(ST_B, d, i, tgt, q.left_ok(i, d)) : 0.);
          }
        }
      const double pa = Za / Zo, pn = Zn / Zo;
      for (int t = 0; t < nt; ++t) { const double a = enA[t], b = enB[t]; enA[t] = pa * a + pn * b; enB[t] = positive ? a : b; }
      for (int t = 0; t < 2; ++t) { const double a = ehA[t], b = ehB[t]; ehA[t] = pa * a + pn * b; ehB[t] = positive ? a : b; }
    } else if (schedule == 0) {
      run_out(m, Zo, true, true, enA, ehA);
      if (outside) { complete_plane2(); copy_tab(out, outside, true); }
      if (outside_o) for (int j = 0; j <= L; ++j) for (int s = 0; s < S; ++s) outside_o[(size_t)j * S + E.ints[E.lay.st_ref + s]] = tolog(out.v.o(j, s), cum[L] - cum[j]);
      run_out(m, positive ? Za : Zn, positive, !positive, enB, ehB);
    } else {
      run_out(m, Za, true, false, enA, ehA);
      if (outside) { complete_plane2(); copy_tab(out, outside, true); }
      out.poison();
      run_out(mr, Zn, false, true, enB, ehB);
      const double pa = Za / Zo, pn = Zn / Zo;
      for (int t = 0; t < nt; ++t) { const double a = enA[t], b = enB[t]; enA[t] = pa * a + pn * b; enB[t] = positive ? a : b; }
      for (int t = 0; t < 2; ++t) { const double a = ehA[t], b = ehB[t]; ehA[t] = pa * a + pn * b; ehB[t] = positive ? a : b; }
    }
    out9[3] = positive ? std::log1p(Zn / Za) : std::log1p(Za / Zn);
    if (ENo) std::copy(enA.begin(), enA.begin() + nt, ENo);
    if (ENx) std::copy(enB.begin(), enB.begin() + nt, ENx);
    if (EHo) { EHo[0] = ehA[0]; EHo[1] = ehA[1]; }
    if (EHx) { EHx[0] = ehB[0]; EHx[1] = ehB[1]; }
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// scan of one sequence: out6 = Ys, Ye, exist_prob, ZL, ZeL, PyNL
int emu_scan_seq(void* h, const double* x, const uint8_t* seq, int L, const uint8_t* qual, double* out6, double* start,
                 double* end, double* inner, int32_t* psihat, char* rss, double* EN) {
  try {
    Emu& E = *(Emu*)h;
    const int nt = E.au->n_theta();
    std::vector<double> theta(x, x + nt);
    if (E.flags & F_SOFTMAX)
      for (int r = 0; r < E.au->n_rows(); ++r) {
        double tot = NEG;
        for (int c = 0; c < E.au->row_width(r); ++c) tot = lse2(tot, x[E.au->row_offset(r) + c]);
        for (int c = 0; c < E.au->row_width(r); ++c) theta[E.au->row_offset(r) + c] = x[E.au->row_offset(r) + c] - tot;
      }
    ModelView m = make_view(E.lay, E.ints, theta.data(), x[nt], x[nt + 1], std::log(E.tau), E.flags & F_NO_PRF,
                            E.flags & F_NO_TURN);
    HostPlan P;
    prepare(E, P, seq, L, qual, nullptr);
    SeqView q = P.view();
    const int S = m.lay.S;
    Tab in(L, P.W, S), out(L, P.W, S);
    std::vector<double> Pys(L, NEG), Pyi(L, NEG), Pye(L + 1, NEG), en(nt + 1, 0.);
    double eh[2] = {0, 0};
    // start / inner posteriors (motif_scanner.hpp:186-193)
    Constraint c0{-1, -1, 0};
    run_inside<false>(m, q, in, c0);
    double ZL = part_func(m, in.v, true, true);
    CpuSink s1{en.data(), eh, {Pys.data(), Pyi.data(), nullptr}};
    run_outside<OUT_SCAN>(m, q, in, out, ZL, c0, s1, true, true);
    int Ys = last_argmax(Pys.data(), L);
    double PyNL = in.v.o(L, m.lay.s00) - ZL;
    // end posterior given the start (:195-202)
    Constraint c1{Ys, -1, 0};
    run_inside<true>(m, q, in, c1);
    double ZeL = part_func(m, in.v, true, true);
    CpuSink s2{en.data(), eh, {nullptr, nullptr, Pye.data()}};
    run_outside<OUT_END>(m, q, in, out, ZeL, c1, s2, true, true);
    int Ye = last_argmax(Pye.data(), L + 1);
    // Viterbi parse (:172-184, 262-362)
    std::vector<TraceRec> tro((size_t)(L + 1) * S);
    TraceView tv{tro.data()};
    Constraint c2{Ys, Ye, 1};
    for (int d = 0; d <= q.W; ++d)
      for (int i = 0; i + d <= q.L; ++i)
        for (int s = 0; s < S; ++s) cyk_target(m, q, in.v, tv, c2, d, i, s);
    for (int s = 0; s < S; ++s) { in.v.o(0, s) = (s == m.lay.s00) ? 0. : NEG; tro[s] = TraceRec{-1, -1, -1, -1, -1}; }
    for (int j = 1; j <= L; ++j)
      for (int s = 0; s < S; ++s) cyk_ext_target(m, q, in.v, tv, c2, j, s);
    std::vector<int32_t> path(L, 0);
    std::string r(L, ' ');
    std::vector<TraceFrame> stack((size_t)4 * (L + 2));
    int s0 = in.v.o(L, m.lay.s0m2) < in.v.o(L, m.lay.s0m1) ? m.lay.s0m1 : m.lay.s0m2;
    trace_back(m, q, in.v, tv, c2, L, s0, path.data(), &r[0], stack.data(), (int)stack.size());
    double tot = NEG;
    for (double v : Pys) tot = lse2(tot, v);
    out6[0] = Ys; out6[1] = Ye; out6[2] = std::exp(tot); out6[3] = ZL; out6[4] = ZeL; out6[5] = PyNL;
    if (start) std::copy(Pys.begin(), Pys.end(), start);
    if (end) std::copy(Pye.begin(), Pye.end(), end);
    if (inner) std::copy(Pyi.begin(), Pyi.end(), inner);
    if (psihat) std::copy(path.begin(), path.end(), psihat);
    if (rss) memcpy(rss, r.data(), L);
    if (EN) for (int k = 0; k < nt; ++k) EN[k] += en[k];
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 1; }
}


// ---- scan of one sequence with the sum passes (K4, K5) in the scaled-linear semiring (lin_rules.h) and the CYK pass
// in log space, as elemdp_scan runs them on the GPU.  Same outputs as emu_scan_seq.
struct CpuLinSink {
  double* EN; double* EH; double* post[3];
  void en(int idx, double w) { EN[idx] += w; }
  void eh(int k, double w) { EH[k] += w; }
  void pos(int which, int p, double w) { if (post[which]) post[which][p] += w; }
};
int emu_scan_seq_lin(void* h, const double* x, const uint8_t* seq, int L, const uint8_t* qual, double* out6, double* start,
                     double* end, double* inner, int32_t* psihat, char* rss, double* EN) {
  try {
    Emu& E = *(Emu*)h;
    const int nt = E.au->n_theta();
    std::vector<double> theta(x, x + nt);
    if (E.flags & F_SOFTMAX)
      for (int r = 0; r < E.au->n_rows(); ++r) {
        double tot = NEG;
        for (int c = 0; c < E.au->row_width(r); ++c) tot = lse2(tot, x[E.au->row_offset(r) + c]);
        for (int c = 0; c < E.au->row_width(r); ++c) theta[E.au->row_offset(r) + c] = x[E.au->row_offset(r) + c] - tot;
      }
    const bool no_prf = E.flags & F_NO_PRF;
    std::vector<double> lin;
    make_lin_params(E.lay, E.ints.data(), theta.data(), E.tau, no_prf, &lin);
    ModelView m = make_view(E.lay, E.ints, theta.data(), x[nt], x[nt + 1], std::log(E.tau), no_prf, E.flags & F_NO_TURN);
    m.lin = lin.data();
    HostPlan P;
    prepare(E, P, seq, L, qual, nullptr);
    SeqView q = P.view();
    const int S = m.lay.S;
    const size_t nc = (size_t)(L + 1) * (P.W + 1), ni = P.items.size();
    std::vector<double> ews(L + 1), xwc(10 * nc), xwi(2 * ni + 1);
    for (int p = 0; p <= L; ++p) ews[p] = std::exp(P.ws[p]);
    const double* terms[5] = {P.e_stack.data(), P.e_ext.data(), P.e_ml.data(), P.e_close.data(), P.e_hp.data()};
    for (int k = 0; k < 2; ++k) {
      for (int t = 0; t < 5; ++t)
        for (size_t c = 0; c < nc; ++c) xwc[(size_t)(k * 5 + t) * nc + c] = lin_weight(m.lambda[k], terms[t][c]);
      for (size_t n = 0; n < ni; ++n) xwi[(size_t)k * ni + n] = lin_weight(m.lambda[k], P.items[n].tsc);
    }
    q.ews = ews.data(); q.xwc = xwc.data(); q.xwc_stride = nc; q.xwi = xwi.data(); q.xwi_stride = ni;
    LinTab in(L, P.W, m.lay, E.ints.data()), out(L, P.W, m.lay, E.ints.data());
    auto zero = [](LinTab& T) { T.poison(); };
    const bool fast = E.fast && m.lay.fp_ok;
    auto run_in = [&](const Constraint& c, bool con) {
      zero(in);
      for (int d = 0; d <= q.W; ++d)
        for (int i = 0; i + d <= q.L; ++i) {
          if (fast) { if (con) fast_inside_cell<true>(m, q, in.v, d, i, c); else fast_inside_cell<false>(m, q, in.v, d, i, c); continue; }
          if (con) lin_inside_cell_pairs<true>(m, q, in.v, d, i, c); else lin_inside_cell_pairs<false>(m, q, in.v, d, i, c);
          for (int s = 0; s < S; ++s) { if (con) lin_inside_target<true>(m, q, in.v, d, i, s, c); else lin_inside_target<false>(m, q, in.v, d, i, s, c); }
        }
      for (int s = 0; s < S; ++s) in.v.o(0, s) = (s == m.lay.s00) ? 1. : 0.;
      for (int j = 1; j <= L; ++j)
        for (int s = 0; s < S; ++s) { if (con) lin_inside_ext_target<true>(m, q, in.v, j, s, c); else lin_inside_ext_target<false>(m, q, in.v, j, s, c); }
    };
    std::vector<double> Pys(L, 0.), Pyi(L, 0.), Pye(L + 1, 0.), en(nt + 1, 0.);
    double eh[2] = {0, 0};
    // K4: start / inner posteriors
    Constraint c0{-1, -1, 0};
    run_in(c0, false);
    const double ZLm = lin_part(m, in.v, true, true);
    const double PyNL = std::log(in.v.o(L, m.lay.s00) / ZLm);
    double sl = 0.;
    for (int p = 0; p < L; ++p) sl += lin[kLinPl2 + seq[p]];
    const double ln2 = 0.69314718055994530942;
    {
      zero(out);
      CpuLinSink s1{en.data(), eh, {Pys.data(), Pyi.data(), nullptr}};
      LinOutCtx<CpuLinSink> xo{m, q, in.v, out.v, 1. / ZLm, s1, c0};
      out.v.o(L, m.lay.s00) = 1.; out.v.o(L, m.lay.s0m1) = 1.; out.v.o(L, m.lay.s0m2) = 1.;
      for (int i = L - 1; i >= 0; --i) for (int s = 0; s < S; ++s) lin_outside_ext_target<OUT_SCAN>(xo, i, s);
      if (fast) fast_rule7(m, q, in.v, out.v);
      for (int d = q.W; d >= 0; --d) for (int i = 0; i + d <= L; ++i) {
        if (fast) { fast_outside_cell<OUT_SCAN>(xo, (LinOutCtx<CpuLinSink>*)nullptr, d, i, -1); continue; }
        for (int s = 0; s < S; ++s) lin_outside_target<OUT_SCAN>(xo, d, i, s);
        lin_outside_cell_pairs<OUT_SCAN>(xo, d, i);
      }
    }
    auto tolog = [](double v) { return v > 0. ? std::log(v) : NEG; };
    std::vector<double> lPys(L), lPyi(L), lPye(L + 1);
    for (int p = 0; p < L; ++p) { lPys[p] = tolog(Pys[p]); lPyi[p] = tolog(Pyi[p]); }
    const int Ys = last_argmax(lPys.data(), L);
    // K5: end posterior given the start
    Constraint c1{Ys, -1, 0};
    run_in(c1, true);
    const double ZeLm = lin_part(m, in.v, true, true);
    {
      zero(out);
      CpuLinSink s2{en.data(), eh, {nullptr, nullptr, Pye.data()}};
      LinOutCtx<CpuLinSink> xo{m, q, in.v, out.v, 1. / ZeLm, s2, c1};
      out.v.o(L, m.lay.s00) = 1.; out.v.o(L, m.lay.s0m1) = 1.; out.v.o(L, m.lay.s0m2) = 1.;
      for (int i = L - 1; i >= 0; --i) for (int s = 0; s < S; ++s) lin_outside_ext_target<OUT_END>(xo, i, s);
      if (fast) fast_rule7(m, q, in.v, out.v);
      for (int d = q.W; d >= 0; --d) for (int i = 0; i + d <= L; ++i) {
        if (fast) { fast_outside_cell<OUT_END>(xo, (LinOutCtx<CpuLinSink>*)nullptr, d, i, Ys); continue; }
        for (int s = 0; s < S; ++s) lin_outside_target<OUT_END>(xo, d, i, s);
        lin_outside_cell_pairs<OUT_END>(xo, d, i);
      }
    }
    for (int p = 0; p <= L; ++p) lPye[p] = tolog(Pye[p]);
    const int Ye = last_argmax(lPye.data(), L + 1);
    // K6: Viterbi parse in log space (scan_rules.h)
    // (with the table-driven forms: the Viterbi pass on the COMPACT table, as launch_cyk_group runs it -- every entry starts as NaN,
    // so a read of an entry nobody stored would show in the parse)
    Tab cyk(L, P.W, S);
    LinTab cykc(L, P.W, m.lay, E.ints.data());
    if (E.fast) { cyk.v = cykc.v; cyk.v.cyk_compact = 1; }
    std::vector<TraceRec> tro((size_t)(L + 1) * S);
    TraceView tv{tro.data()};
    Constraint c2{Ys, Ye, 1};
    for (int d = 0; d <= q.W; ++d)
      for (int i = 0; i + d <= q.L; ++i)
        for (int s = 0; s < S; ++s) cyk_target(m, q, cyk.v, tv, c2, d, i, s);
    for (int s = 0; s < S; ++s) { cyk.v.o(0, s) = (s == m.lay.s00) ? 0. : NEG; tro[s] = TraceRec{-1, -1, -1, -1, -1}; }
    for (int j = 1; j <= L; ++j)
      for (int s = 0; s < S; ++s) cyk_ext_target(m, q, cyk.v, tv, c2, j, s);
    std::vector<int32_t> path(L, 0);
    std::string r(L, ' ');
    std::vector<TraceFrame> stack((size_t)4 * (L + 2));
    int s0 = cyk.v.o(L, m.lay.s0m2) < cyk.v.o(L, m.lay.s0m1) ? m.lay.s0m1 : m.lay.s0m2;
    trace_back(m, q, cyk.v, tv, c2, L, s0, path.data(), &r[0], stack.data(), (int)stack.size());
    double tot = 0.;
    for (double v : Pys) tot += v;
    out6[0] = Ys; out6[1] = Ye; out6[2] = tot; out6[3] = tolog(ZLm) - sl * ln2; out6[4] = tolog(ZeLm) - sl * ln2; out6[5] = PyNL;
    if (start) std::copy(lPys.begin(), lPys.end(), start);
    if (end) std::copy(lPye.begin(), lPye.end(), end);
    if (inner) std::copy(lPyi.begin(), lPyi.end(), inner);
    if (psihat) std::copy(path.begin(), path.end(), psihat);
    if (rss) memcpy(rss, r.data(), L);
    if (EN) for (int k = 0; k < nt; ++k) EN[k] += en[k];
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 1; }
}

}  // extern "C"
