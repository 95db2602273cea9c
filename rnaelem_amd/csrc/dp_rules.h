// dp_rules.h -- the per-target ("gather") form of the RNAelem inside / outside / CYK recurrences.
//
// One *target* = one (cell (i,d), interval state s): the function computes every structural state
// P,E,M,B,1,2,L of that target from cells of strictly smaller span (inside) or strictly larger
// span (outside), so all targets of one anti-diagonal d are independent -- that is the unit the
// HIP kernels (dp_kernels.hip) distribute over the lanes of a workgroup, one diagonal per barrier.
// The exterior chain O(j) is a second, sequential phase with S independent targets per step.
//
// The reference evaluates the same rules as a *scatter* driven by the structural sweep
// (RNAelem/energy_model.hpp:340-547 x RNAelem/motif_model.hpp:230-613 x the functors in
// RNAelem/motif_trainer.hpp:274-458 and RNAelem/motif_scanner.hpp:364-913).  Rule numbers in the
// comments are those of SURVEY.md Appendix A.  Summation order differs from the reference, so
// parity is to tolerance (1e-9 rel on log Z), not bitwise; CYK keeps the reference's candidate
// order per target because ties are broken by "first strictly greater" (motif_scanner.hpp:821).
//
// The code is plain C++ (no HIP intrinsics) so that tests/emul can run the *same* functions on
// the CPU against the oracle; the product only ever calls them from device code.
#pragma once
#include <cmath>
#include <cstdint>

#include "device_layout.h"

// Under hipcc (the product build) the rules exist as DEVICE code only: the shipped library has no
// host instantiation of the recurrences, hence no CPU path.  Plain g++ (tests/emul) sees inline functions.
#if defined(__HIPCC__)
#define ELEMDP_HD __device__ __forceinline__
#else
#define ELEMDP_HD inline
#endif

namespace elemdp {

#if defined(__HIP_DEVICE_COMPILE__)
#define ELEMDP_NEG_INF (-__builtin_huge_val())
#else
#define ELEMDP_NEG_INF (-HUGE_VAL)
#endif

// ---------------------------------------------------------------------------------------------
// log-semiring accumulators
// ---------------------------------------------------------------------------------------------
// exp(x) for x <= 0 (any x below -746 gives 0): Cody-Waite reduction to |r| <= ln2/2, degree-12 Taylor
// polynomial (truncation 1.7e-16 relative), one ldexp.  About a third of the instructions of the
// generic library exp, which matters because one exp per term is the inner loop of the whole DP.
ELEMDP_HD double exp_neg(double x) {
  x = fmax(x, -746.);
  const double kf = rint(x * 1.4426950408889634074);
  double r = fma(-kf, 6.93147180369123816490e-01, x);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double p = 2.08767569878680989792e-09;            // 1/12!
  p = fma(p, r, 2.50521083854417187751e-08);        // 1/11!
  p = fma(p, r, 2.75573192239858906526e-07);        // 1/10!
  p = fma(p, r, 2.75573192239858906526e-06);        // 1/9!
  p = fma(p, r, 2.48015873015873015873e-05);        // 1/8!
  p = fma(p, r, 1.98412698412698412698e-04);        // 1/7!
  p = fma(p, r, 1.38888888888888888889e-03);        // 1/6!
  p = fma(p, r, 8.33333333333333333333e-03);        // 1/5!
  p = fma(p, r, 4.16666666666666666667e-02);        // 1/4!
  p = fma(p, r, 1.66666666666666666667e-01);        // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)kf);
}

// streaming log-sum-exp: value = m + log(s); one exp per term, one log per target; branch-free
struct LseAcc {
  double m, s;
  ELEMDP_HD LseAcc() : m(ELEMDP_NEG_INF), s(0.) {}
  ELEMDP_HD void add(double x) {
    const bool dead = (x == ELEMDP_NEG_INF);
    const double t = dead ? 0. : exp_neg(-fabs(x - m));  // m = -inf: exp_neg(-inf) = 0
    const bool up = x > m;
    s = up ? fma(s, t, 1.) : s + t;
    m = up ? x : m;
  }
  // merge another partial sum (m2 + log s2)
  ELEMDP_HD void merge(double m2, double s2) {
    const bool dead = (m2 == ELEMDP_NEG_INF);
    const double t = dead ? 0. : exp_neg(-fabs(m2 - m));
    const bool up = m2 > m;
    s = up ? fma(s, t, s2) : fma(s2, t, s);
    m = up ? m2 : m;
  }
  ELEMDP_HD double value() const { return (m == ELEMDP_NEG_INF) ? ELEMDP_NEG_INF : m + log(s); }
};

// Serial heavy sums with ONE state tuple (the BPP filter's one-state automaton: one lane per cell walks up to ~100 split
// points / items): the terms of kRun consecutive steps are fetched together, then added in the original order -- the
// chain of dependent loads per step (index -> item record -> table values) is paid once per kRun steps.
constexpr int kRun = 8;
template <class OkFn, class TermFn>
ELEMDP_HD void lse_run4(LseAcc& a, int n, OkFn ok, TermFn term) {
  for (int k0 = 0; k0 < n; k0 += kRun) {
    double v[kRun];
    bool g[kRun];
#pragma unroll
    for (int u = 0; u < kRun; ++u) {
      const int k = (k0 + u < n) ? k0 + u : n - 1;
      g[u] = k0 + u < n && ok(k);
      v[u] = term(k);
    }
#pragma unroll
    for (int u = 0; u < kRun; ++u)
      if (g[u]) a.add(v[u]);
  }
}

// ---------------------------------------------------------------------------------------------
// views
// ---------------------------------------------------------------------------------------------
struct ModelView {
  const AutomatonLayout& lay;  // (a reference: on the GPU it aliases the kernel argument, no per-lane copy)
  ELEMDP_HD explicit ModelView(const AutomatonLayout& l) : lay(l) {}
  const int32_t* ints;   // automaton blob: per-state attributes + unary lists (first lay.n_small ints; LDS on the GPU)
  const int32_t* big;    // the whole blob incl. the tuple lists of rules 2 / 6c / 7 (global memory on the GPU)
  const double* theta;   // n_theta log-probabilities
  const double* lin = nullptr;  // linear parameter block of the scaled-linear train pipeline (lin_rules.h) or null
  double lambda[2];
  double log_tau;
  int32_t lam_same;      // lambda[0] == lambda[1]
  int32_t no_prf;        // --no-profile: theta terms are 0 and no emission counts
  int32_t m_min;         // minimal span of a multiloop cell M: 2*(2+turn)=10, or 4 with NO_TURN
  int32_t dbg = 0;       // timing experiments only (LinArgs::dbg): 32 skips the rule-7 gather of the outside unary phase

  ELEMDP_HD int st_l(int s) const { return ints[lay.st_l + s]; }
  ELEMDP_HD int st_r(int s) const { return ints[lay.st_r + s]; }
  // (a select, not an indexed read: dynamic indexing would force the whole view into scratch memory on the GPU)
  ELEMDP_HD double lam(int s) const { return ints[lay.st_lam + s] ? lambda[1] : lambda[0]; }
  // index of the EH accumulator a transition with parent s feeds (motif_trainer.hpp:380-381)
  ELEMDP_HD int eh_index(int s) const { return lam_same ? 0 : ints[lay.st_lam + s]; }
  ELEMDP_HD double theta_at(int row, int col) const { return theta[ints[lay.row_off + row] + col]; }
  ELEMDP_HD int param_index(int row, int col) const { return ints[lay.row_off + row] + col; }
};

struct SeqView {
  int32_t L, W, C;
  const uint8_t* seq;        // L base codes 0..4
  const double* ws;          // L+1 position weights (motif_model.hpp:62-70)
  const uint32_t* okbits;    // kept pairs, bit (i*(W+1)+d)
  const int16_t* dmin;       // L+1: smallest kept span starting at i, 0 = none  (left_bp_ok, energy_model.hpp:203-209)
  const uint8_t* unp;        // L: position may be emitted unpaired (all 1 unless FIX_RSS)
  // structural terms keyed by the PAIR cell (i,d): index i*(W+1)+d
  const double* e_stack;     // rule 1b: loop_energy(i,j-1,i+1,j-2); -inf if inner pair not kept
  const double* e_ext;       // rule 7 : sum_ext_m(i,j-1,true)
  const double* e_ml;        // rule 3b: sum_ext_m(i,j-1,false) + mlintern
  const double* e_close;     // rule 6a for E(i+1,j-1): sum_ext_m(j-1,i,false)+mlclosing+mlintern
  const double* e_hp;        // rule 6b for E(i+1,j-1): hairpin_energy(i,j-1)
  // rule 6c items; CSR by outer E cell, by inner P cell, by left loop cell (i,k), by right loop cell (l,j)
  const LoopItem* items;
  const int32_t* by_outer_off;
  const int32_t* by_inner_off; const int32_t* by_inner_idx;
  const int32_t* by_left_off;  const int32_t* by_left_idx;
  const int32_t* by_right_off; const int32_t* by_right_idx;
  const uint8_t* item_in;    // item belongs to the inside enumeration (u1+u2 <= C; see SURVEY §7 quirk ii)
  // scaled-linear train pipeline (lin_rules.h): exp of the position weights; exp(lambda_k * term) of the structural
  // terms [k*5+term][cell] and of the interior-loop items [k][item], recomputed per evaluation
  const double* ews = nullptr;
  const double* xwc = nullptr; size_t xwc_stride = 0;
  const double* xwi = nullptr; size_t xwi_stride = 0;
  // items in the by_inner / by_left / by_right orders and their weights: order o (1..3) at xwi + o * 2 * xwi_stride
  const LoopItem* items_inner = nullptr; const LoopItem* items_left = nullptr; const LoopItem* items_right = nullptr;
  // the pair mask indexed by (end j, span): bit j * (W+1) + span <=> pair cell (j - span, span)  (k_mask_by_end)
  const uint32_t* okbits_end = nullptr;

  ELEMDP_HD int cell(int i, int d) const { return i * (W + 1) + d; }
  ELEMDP_HD bool pair_ok(int i, int d) const {  // is_parsable<ST_P>
    if (i < 0 || d < 0 || d > W || i + d > L) return false;
    int c = i * (W + 1) + d;
    return (okbits[c >> 5] >> (c & 31)) & 1u;
  }
  ELEMDP_HD bool left_ok(int i, int d) const {  // is_parsable<ST_B/1/2>
    if (d > W || d < 0 || i + d > L) return false;
    int dm = dmin[i];
    return dm > 0 && d >= dm;
  }
  ELEMDP_HD bool e_ok(int i, int d) const { return i > 0 && d + 2 <= W && pair_ok(i - 1, d + 2); }  // is_parsable<ST_E>
};

// banded tables [e][d][i][s] and exterior table [j][s]
struct TableView {
  double* band;
  double* ext;
  int32_t L, W, S;
  // (the engine rejects batches whose band table would exceed 2^31 entries)
  ELEMDP_HD uint32_t idx(int e, int d, int i, int s) const {
    return (((uint32_t)e * (uint32_t)(W + 1) + (uint32_t)d) * (uint32_t)(L + 1) + (uint32_t)i) * (uint32_t)S + (uint32_t)s;
  }
  ELEMDP_HD double& at(int e, int d, int i, int s) const { return band[idx(e, d, i, s)]; }
  // (omask: the exterior-chain kernels keep the last rows of the chain in an LDS ring of a power-of-two number of rows)
  uint32_t omask = 0xffffffffu;
  ELEMDP_HD double& o(int j, int s) const { return ext[((uint32_t)j & omask) * (uint32_t)S + (uint32_t)s]; }
  // pair table of the factorised rule 2 (scaled-linear pipeline, lin_rules.h): [d][i][p], p < nA pairs (s1, t), rows of
  // nAs >= nA doubles
  double* ap = nullptr;
  int32_t nA = 0, nAs = 0;
  ELEMDP_HD uint32_t aidx(int d, int i, int p) const { return ((uint32_t)d * (uint32_t)(L + 1) + (uint32_t)i) * (uint32_t)nAs + (uint32_t)p; }
  ELEMDP_HD double& a(int d, int i, int p) const { return ap[aidx(d, i, p)]; }
  // ---- compact layout (scaled-linear pipeline only; the accessors above are the dense layout of the log-space pipelines and
  // the Viterbi pass).  Plane e keeps one row of rs[e] doubles per cell, a column per interval state that is useful in the
  // plane (AutomatonLayout::tab_cmap); entries that are structurally 0 -- a state without a column, or a cell that is not
  // parsable in that plane (is_parsable, energy_model.hpp:289-338) -- are neither stored nor read: the table holds garbage
  // there, so every reader decides liveness from the pair mask / dmin first (`live`) and never multiplies a loaded value by 0.
  uint32_t pb[7] = {0, 0, 0, 0, 0, 0, 0};   // first entry of plane e: tab_cs[e] * (W+1) * (L+1)
  int32_t rs[7] = {0, 0, 0, 0, 0, 0, 0};    // row stride of plane e
  const int32_t* cm = nullptr;              // [7][S] column of (plane, state) or -1 (LDS on the GPU)
  ELEMDP_HD void set_compact(const AutomatonLayout& A, const int32_t* ints) {
    const uint32_t cells = (uint32_t)(W + 1) * (uint32_t)(L + 1);
#pragma unroll
    for (int e = 0; e < 7; ++e) {
      pb[e] = A.tab_cell ? (uint32_t)A.tab_cs[e] : (uint32_t)A.tab_cs[e] * cells;
      rs[e] = A.tab_cell ? A.tab_row : A.tab_rs[e];
    }
    cm = ints + A.tab_cmap;
    nAs = A.ap_rs;
  }
  ELEMDP_HD int col(int e, int s) const { return cm[e * S + s]; }
  ELEMDP_HD uint32_t cidx(int e, int d, int i, int c) const {
    return pb[e] + ((uint32_t)d * (uint32_t)(L + 1) + (uint32_t)i) * (uint32_t)rs[e] + (uint32_t)c;
  }
  // load by column (c < 0 or !live: 0, from an address that is valid anyway -- a select, not a branch)
  ELEMDP_HD double ldc(int e, int d, int i, int c, bool live = true) const {
    const bool ok = live && c >= 0;
    const double v = band[ok ? cidx(e, d, i, c) : 0u];
    return ok ? v : 0.;
  }
  ELEMDP_HD double ld(int e, int d, int i, int s, bool live = true) const { return ldc(e, d, i, col(e, s), live); }
  ELEMDP_HD void st(int e, int d, int i, int s, double v, bool live = true) const {
    const int c = col(e, s);
    if (live && c >= 0) band[cidx(e, d, i, c)] = v;
  }
  // ---- the Viterbi pass (max-plus, scan_rules.h) on either layout: dense [e][d][i][s] (the fused scan kernel, the CPU driver), or
  // -- cyk_compact, the batch pipeline -- the compact rows above with EVERY cell stored (log 0 where a cell is not parsable) and a
  // missing column read as log 0 / never written: an entry without a column cannot occur in a complete parse, so no kept transition
  // reads it (Automaton::flatten keeps a transition only under a useful parent, and its children are then useful themselves).
  // The dense table of (.....) is 203 doubles per cell, the compact one 64; the split points of rule 2 read rows of 5 front
  // states out of 232-byte rows there, out of 64-byte rows here.
  int32_t cyk_compact = 0;
  ELEMDP_HD double ldm(int e, int d, int i, int s) const {
    if (!cyk_compact) return band[idx(e, d, i, s)];
    const int c = col(e, s);
    const double v = band[c >= 0 ? cidx(e, d, i, c) : 0u];
    return c >= 0 ? v : ELEMDP_NEG_INF;
  }
  ELEMDP_HD void stm(int e, int d, int i, int s, double v) const {
    if (!cyk_compact) { band[idx(e, d, i, s)] = v; return; }
    const int c = col(e, s);
    if (c >= 0) band[cidx(e, d, i, c)] = v;
  }
  // pair table: an entry (d, i, .) exists iff 0 < dmin[i] < d
  ELEMDP_HD double lda(int d, int i, int p, bool live = true) const {
    const double v = ap[live ? aidx(d, i, p) : 0u];
    return live ? v : 0.;
  }
};

// calls fn(n) for every set bit bit0 + n of a pair mask with n in [lo, hi], ascending (walks the 32-bit words)
template <class F> ELEMDP_HD void for_mask_bits(const uint32_t* m, int bit0, int lo, int hi, F fn) {
  if (hi < lo) return;
  const int b = bit0 + lo, e = bit0 + hi;
  int w = b >> 5;
  uint32_t word = m[w] & (~0u << (b & 31));
  for (;;) {
    while (word) {
      const int bit = (w << 5) + __builtin_ctz(word);
      if (bit > e) return;
      fn(bit - bit0);
      word &= word - 1;
    }
    if (((w + 1) << 5) > e) return;
    word = m[++w];
  }
}

ELEMDP_HD bool m_ok(const ModelView& m, const SeqView& q, int i, int d) {  // is_parsable<ST_M>
  return 0 < i && i + d < q.L && d <= q.W && m.m_min <= d;
}

// base-pair type (bio_sequence.hpp:20-26): CG=1 GC=2 GU=3 UG=4 AU=5 UA=6
ELEMDP_HD int bp_type(int a, int b) {
  // rows N,A,C,G,U
  const int t = a * 5 + b;
  return t == 9 ? 5 : t == 13 ? 1 : t == 17 ? 2 : t == 19 ? 3 : t == 21 ? 6 : t == 23 ? 4 : 0;
}

// ---------------------------------------------------------------------------------------------
// emission weights  wt = theta + (tau + position weight)   (motif_model.hpp:243-421)
// ---------------------------------------------------------------------------------------------
// right emission of base at `pos` by the r-node of `par` (rules 3a, 8, L<-L)
ELEMDP_HD double w_right(const ModelView& m, const SeqView& q, int par, int tau_flag, int pos) {
  const int b = q.seq[pos];
  double w = (m.no_prf || b == 0) ? 0. : m.theta_at(m.ints[m.lay.st_row_r + par], b - 1);
  double ws = m.ints[m.lay.st_w_r + par] ? q.ws[pos] : 0.;
  double t = tau_flag ? m.log_tau : 0.;
  return w + (t + ws);
}
// left emission of base at `pos` by the l-node of `child` (rule 5a)
ELEMDP_HD double w_left(const ModelView& m, const SeqView& q, int child, int tau_flag, int pos) {
  const int b = q.seq[pos];
  double w = (m.no_prf || b == 0) ? 0. : m.theta_at(m.ints[m.lay.st_row_l + child], b - 1);
  double ws = m.ints[m.lay.st_w_l + child] ? q.ws[pos] : 0.;
  double t = tau_flag ? m.log_tau : 0.;
  return w + (t + ws);
}
// pair emission: l-node of `child` at pos pi, r-node of `par` at pos pj (rules 1a, 1b; profile_hmm.hpp:113-135)
ELEMDP_HD double w_pair(const ModelView& m, const SeqView& q, int par, int child, int tau_flag, int pi, int pj) {
  const int bi = q.seq[pi], bj = q.seq[pj];
  double w = 0.;
  if (!m.no_prf) {
    if (m.ints[m.lay.st_pair_r + par]) {
      int t = bp_type(bi, bj);
      w = t ? m.theta_at(m.ints[m.lay.st_row_r + par], t - 1) : 0.;
    } else {
      w = (bi ? m.theta_at(m.ints[m.lay.st_row_l + child], bi - 1) : 0.) +
          (bj ? m.theta_at(m.ints[m.lay.st_row_r + par], bj - 1) : 0.);
    }
  }
  double ws = (m.ints[m.lay.st_w_l + child] ? q.ws[pi] : 0.) + (m.ints[m.lay.st_w_r + par] ? q.ws[pj] : 0.);
  double t = tau_flag ? m.log_tau : 0.;
  return w + (t + ws);
}

// ---------------------------------------------------------------------------------------------
// constraints of the scan passes (motif_scanner.hpp:594-622, 839-873)
// ---------------------------------------------------------------------------------------------
struct Constraint {
  int32_t ys;  // motif must start at ys (-1 = off)
  int32_t ye;  // motif must end at ye   (-1 = off; CYK only)
  int32_t use_end;
};
// pair emission: parent P(i,j,par), child (i+1,j-1,ch); emits positions i and j-1
ELEMDP_HD bool allow_pair(const ModelView& m, const Constraint& c, int L, int i, int j, int par, int ch) {
  const int pl = m.st_l(par), pr = m.st_r(par), cl = m.st_l(ch), cr = m.st_r(ch);
  if (i == c.ys && !(pl == 0 && cl == 1)) return false;
  if (j - 1 == c.ys && !(cr == 0 && pr == 1)) return false;
  if (c.use_end) {
    const int M = m.lay.M;
    if (i == c.ye && !(pl == M - 2 && cl == M - 1)) return false;
    if (j - 1 == c.ye && !(cr == M - 2 && pr == M - 1)) return false;
    if (j == c.ye && L == j && pr != M - 2) return false;
  }
  return true;
}
// right emission: parent (.,j,par), child (.,j-1,ch); emits position j-1
ELEMDP_HD bool allow_right(const ModelView& m, const Constraint& c, int L, int j, int par, int ch) {
  const int pr = m.st_r(par), cr = m.st_r(ch);
  if (j - 1 == c.ys && !(cr == 0 && pr == 1)) return false;
  if (c.use_end) {
    const int M = m.lay.M;
    if (j - 1 == c.ye && !(cr == M - 2 && pr == M - 1)) return false;
    if (j == c.ye && L == j && pr != M - 2) return false;
  }
  return true;
}
// left emission: parent M(i,j,par), child M(i+1,j,ch); emits position i
ELEMDP_HD bool allow_left(const ModelView& m, const Constraint& c, int i, int par, int ch) {
  const int pl = m.st_l(par), cl = m.st_l(ch);
  if (i == c.ys && !(pl == 0 && cl == 1)) return false;
  if (c.use_end) {
    const int M = m.lay.M;
    if (i == c.ye && !(pl == M - 2 && cl == M - 1)) return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// INSIDE, sum semiring, optional start constraint (train K2, scan K4/K5 inside halves)
// ---------------------------------------------------------------------------------------------
// The two long-range rules -- bifurcation (2) and interior loops (6c), > 90 % of all terms -- are
// kept apart as "heavy" sums: on the GPU a whole wave evaluates them for one cell with one lane per
// (state tuple) and the k / item loop unrolled for memory-level parallelism (kernels.hip); the
// serial forms below define the same sums for the CPU emulation.
ELEMDP_HD bool bif_valid(const SeqView& q, int j, int k) {  // is_parsable<ST_2>(k, j), energy_model.hpp:360-361
  const int dk = q.dmin[k];
  return dk != 0 && j - k >= dk;
}
ELEMDP_HD double bif_term(const TableView& T, int i, int j, int k, int s1, int s2) {
  return T.at(ST_1, k - i, i, s1) + T.at(ST_2, j - k, k, s2);
}
ELEMDP_HD double loop_term(const TableView& T, int i, int j, const LoopItem& x, int s1, int s2, int s3, double lt) {
  return T.at(ST_P, x.l - x.k, x.k, s1) + (T.at(ST_L, x.k - i, i, s2) + (T.at(ST_L, j - x.l, x.l, s3) + lt));
}
// rule 2: B(i,j,s); requires left_ok(i,d)
ELEMDP_HD double heavy_bif(const ModelView& m, const SeqView& q, const TableView& T, int d, int i, int s) {
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.big;
  const int j = i + d;
  LseAcc a;
  const int t0 = G[A.split_off + s];
  if (G[A.split_off + s + 1] - t0 == 1) {
    const int s1 = G[A.split_ent + 2 * t0], s2 = G[A.split_ent + 2 * t0 + 1], k_lo = i + q.dmin[i];
    lse_run4(a, j - k_lo, [&](int n) { return bif_valid(q, j, k_lo + n); }, [&](int n) { return bif_term(T, i, j, k_lo + n, s1, s2); });
    return a.value();
  }
  for (int k = i + q.dmin[i]; k < j; ++k) {
    if (!bif_valid(q, j, k)) continue;
    for (int t = G[A.split_off + s]; t < G[A.split_off + s + 1]; ++t)
      a.add(bif_term(T, i, j, k, G[A.split_ent + 2 * t], G[A.split_ent + 2 * t + 1]));
  }
  return a.value();
}
// rule 6c: interior-loop part of E(i,j,s); requires e_ok(i,d)
ELEMDP_HD double heavy_loop(const ModelView& m, const SeqView& q, const TableView& T, int d, int i, int s) {
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.big;
  const int j = i + d;
  const double lam = m.lam(s);
  LseAcc a;
  const int c0 = q.by_outer_off[q.cell(i, d)], c1 = q.by_outer_off[q.cell(i, d) + 1];
  const int t0 = G[A.quad_off + s];
  if (G[A.quad_off + s + 1] - t0 == 1) {
    const int s1 = G[A.quad_ent + 3 * t0], s2 = G[A.quad_ent + 3 * t0 + 1], s3 = G[A.quad_ent + 3 * t0 + 2];
    lse_run4(a, c1 - c0, [&](int n) { return q.item_in[c0 + n] != 0; },
             [&](int n) { const LoopItem x = q.items[c0 + n]; return loop_term(T, i, j, x, s1, s2, s3, lam * x.tsc); });
    return a.value();
  }
  for (int it = c0; it < c1; ++it) {
    if (!q.item_in[it]) continue;
    const LoopItem x = q.items[it];
    const double lt = lam * x.tsc;
    for (int t = G[A.quad_off + s]; t < G[A.quad_off + s + 1]; ++t)
      a.add(loop_term(T, i, j, x, G[A.quad_ent + 3 * t], G[A.quad_ent + 3 * t + 1], G[A.quad_ent + 3 * t + 2], lt));
  }
  return a.value();
}

#ifndef ELEMDP_KUNARY
#define ELEMDP_KUNARY 2
#endif
// Parent values fetched ahead per unary transition list (complete lists hold up to 3 entries: edges h, h-1, h-2; the lists
// pruned to complete parses rarely more than 2); longer lists continue in a loop.  2 instead of 3: k4_in needs 78 instead
// of 94 vector registers = six workgroups per CU instead of five (+6 % on the whole evaluation).
constexpr int kUnary = ELEMDP_KUNARY;
#ifndef ELEMDP_KUNARYR
#define ELEMDP_KUNARYR ELEMDP_KUNARY
#endif
// (the right-unpaired lists are the ones that keep a third entry after pruning: their own prefetch depth, scaled-linear rules)
constexpr int kUnaryR = ELEMDP_KUNARYR;

// Computes and stores P,E,M,B,1,2,L of target (i, d, s) given the heavy sums HB (= B) and HE.
// All short-range operands (diagonals d-1, d-2) are fetched first, with fixed unrolling, so that the
// loads of one target are in flight together.  CONSTRAINED=false compiles the scan checks away.
template <bool CONSTRAINED>
ELEMDP_HD void inside_target_u(const ModelView& m, const SeqView& q, const TableView& T, const Constraint& c, int d,
                               int i, int s, double HB, double HE) {
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int j = i + d;
  const double NEG = ELEMDP_NEG_INF;
  const double lam = m.lam(s);
  const bool isloop = I[A.st_is_loop + s] != 0;
  const bool pok = q.pair_ok(i, d);
  const bool lok = q.left_ok(i, d);
  const bool mok = m_ok(m, q, i, d);
  const bool eok = q.e_ok(i, d);
  const bool doL = isloop && d > 0;
  const bool do2 = lok && q.left_ok(i, d - 1) && q.unp[j - 1];
  const bool doM = mok && m_ok(m, q, i + 1, d - 1) && q.unp[i];

  // ---- operand fetch
  const int r0 = I[A.right_off + s], nR = I[A.right_off + s + 1] - r0;
  const int p0 = I[A.pair_off + s], nP = I[A.pair_off + s + 1] - p0;
  const int l0 = I[A.left_off + s], nL = I[A.left_off + s + 1] - l0;
  double xL[kUnary], x2[kUnary], xE[kUnary], xP[kUnary], xM[kUnary];
  double est = NEG, eml = NEG, ecl = NEG, ehp = NEG;
  if (pok) { est = q.e_stack[q.cell(i, d)]; eml = q.e_ml[q.cell(i, d)]; }
  if (eok) { const int pc = q.cell(i - 1, d + 2); ecl = q.e_close[pc]; ehp = q.e_hp[pc]; }
  // The loads are unconditional (from clamped, always valid cells) and masked afterwards: a predicated load
  // becomes a branch with its own wait, which would serialise the fetches of one target.
  const int d1 = d > 0 ? d - 1 : 0, d2 = d > 1 ? d - 2 : 0, i1 = i < q.L ? i + 1 : i;
#pragma unroll
  for (int u = 0; u < kUnary; ++u) {
    const bool vr = u < nR;
    const int s1 = vr ? I[A.right_ent + 2 * (r0 + u)] : 0;
    const bool okr = vr && (!CONSTRAINED || allow_right(m, c, q.L, j, s, s1));
    const bool vp = u < nP;
    const int sp = vp ? I[A.pair_ent + 2 * (p0 + u)] : 0;
    const bool okp = pok && vp && (!CONSTRAINED || allow_pair(m, c, q.L, i, j, s, sp));
    const bool vl = u < nL;
    const int sl = vl ? I[A.left_ent + 2 * (l0 + u)] : 0;
    const bool okl = doM && vl && (!CONSTRAINED || allow_left(m, c, i, s, sl));
    const double tL = T.at(ST_L, d1, i, s1), t2 = T.at(ST_2, d1, i, s1);
    const double tE = T.at(ST_E, d2, i1, sp), tP = T.at(ST_P, d2, i1, sp);
    const double tM = T.at(ST_M, d1, i1, sl);
    xL[u] = (doL && okr) ? tL : NEG;
    x2[u] = (do2 && okr) ? t2 : NEG;
    xE[u] = okp ? tE : NEG;
    xP[u] = (okp && est != NEG) ? tP : NEG;
    xM[u] = okl ? tM : NEG;
  }

  // ---- L(i,j,s): loop emission chain (motif_model.hpp:243-257; init motif_trainer.hpp:89-95)
  double vL = NEG;
  if (isloop) {
    if (d == 0) vL = (m.st_l(s) == m.st_r(s)) ? 0. : NEG;
    else {
      LseAcc a;
#pragma unroll
      for (int u = 0; u < kUnary; ++u)
        if (u < nR) a.add(xL[u] + w_right(m, q, s, I[A.right_ent + 2 * (r0 + u) + 1], j - 1));
      for (int u = kUnary; u < nR; ++u) {
        const int s1 = I[A.right_ent + 2 * (r0 + u)], tf = I[A.right_ent + 2 * (r0 + u) + 1];
        if (CONSTRAINED && !allow_right(m, c, q.L, j, s, s1)) continue;
        a.add(T.at(ST_L, d - 1, i, s1) + w_right(m, q, s, tf, j - 1));
      }
      vL = a.value();
    }
  }
  T.at(ST_L, d, i, s) = vL;

  // ---- P(i,j,s): rules 1a, 1b
  double vP = NEG;
  if (pok) {
    LseAcc a;
    const double lest = (est != NEG) ? lam * est : 0.;  // (0 * -inf would be NaN when lambda = 0)
#pragma unroll
    for (int u = 0; u < kUnary; ++u)
      if (u < nP) {
        const int s1 = I[A.pair_ent + 2 * (p0 + u)], tf = I[A.pair_ent + 2 * (p0 + u) + 1];
        const double w = w_pair(m, q, s, s1, tf, i, j - 1);
        a.add(xE[u] + w);                       // 1a (tsc = 0)
        a.add(xP[u] + (w + lest));              // 1b (xP is log 0 when the stack term is)
      }
    for (int u = kUnary; u < nP; ++u) {
      const int s1 = I[A.pair_ent + 2 * (p0 + u)], tf = I[A.pair_ent + 2 * (p0 + u) + 1];
      if (CONSTRAINED && !allow_pair(m, c, q.L, i, j, s, s1)) continue;
      const double w = w_pair(m, q, s, s1, tf, i, j - 1);
      a.add(T.at(ST_E, d - 2, i + 1, s1) + w);
      if (est != NEG) a.add(T.at(ST_P, d - 2, i + 1, s1) + (w + lam * est));
    }
    vP = a.value();
  }
  T.at(ST_P, d, i, s) = vP;

  // ---- B(i,j,s): rule 2 (heavy sum)
  const double vB = lok ? HB : NEG;
  T.at(ST_B, d, i, s) = vB;

  // ---- 2(i,j,s): rules 3a, 3b ; 1(i,j,s): rules 4a, 4b
  double v2 = NEG, v1 = NEG;
  if (lok) {
    LseAcc a;
    if (do2) {
#pragma unroll
      for (int u = 0; u < kUnary; ++u)
        if (u < nR) a.add(x2[u] + w_right(m, q, s, I[A.right_ent + 2 * (r0 + u) + 1], j - 1));
      for (int u = kUnary; u < nR; ++u) {
        const int s1 = I[A.right_ent + 2 * (r0 + u)], tf = I[A.right_ent + 2 * (r0 + u) + 1];
        if (CONSTRAINED && !allow_right(m, c, q.L, j, s, s1)) continue;
        a.add(T.at(ST_2, d - 1, i, s1) + w_right(m, q, s, tf, j - 1));
      }
    }
    if (pok && eml != NEG) a.add(vP + lam * eml);
    v2 = a.value();
    LseAcc b;
    b.add(v2);
    b.add(vB);
    v1 = b.value();
  }
  T.at(ST_2, d, i, s) = v2;
  T.at(ST_1, d, i, s) = v1;

  // ---- M(i,j,s): rules 5a, 5b
  double vM = NEG;
  if (mok) {
    LseAcc a;
    if (doM) {
#pragma unroll
      for (int u = 0; u < kUnary; ++u)
        if (u < nL) {
          const int s1 = I[A.left_ent + 2 * (l0 + u)], tf = I[A.left_ent + 2 * (l0 + u) + 1];
          a.add(xM[u] + w_left(m, q, s1, tf, i));
        }
      for (int u = kUnary; u < nL; ++u) {
        const int s1 = I[A.left_ent + 2 * (l0 + u)], tf = I[A.left_ent + 2 * (l0 + u) + 1];
        if (CONSTRAINED && !allow_left(m, c, i, s, s1)) continue;
        a.add(T.at(ST_M, d - 1, i + 1, s1) + w_left(m, q, s1, tf, i));
      }
    }
    if (lok) a.add(vB);
    vM = a.value();
  }
  T.at(ST_M, d, i, s) = vM;

  // ---- E(i,j,s): rules 6a, 6b, 6c (heavy sum) ; closing pair is the cell (i-1, d+2)
  double vE = NEG;
  if (eok) {
    LseAcc a;
    if (mok && ecl != NEG) a.add(vM + lam * ecl);
    if (isloop && ehp != NEG) a.add(vL + lam * ehp);
    a.add(HE);
    vE = a.value();
  }
  T.at(ST_E, d, i, s) = vE;
}

// serial form: heavy sums evaluated in place (CPU emulation, S = 1 paths)
template <bool CONSTRAINED>
ELEMDP_HD void inside_target(const ModelView& m, const SeqView& q, const TableView& T, const Constraint& c, int d,
                             int i, int s) {
  const double HB = q.left_ok(i, d) ? heavy_bif(m, q, T, d, i, s) : ELEMDP_NEG_INF;
  const double HE = q.e_ok(i, d) ? heavy_loop(m, q, T, d, i, s) : ELEMDP_NEG_INF;
  inside_target_u<CONSTRAINED>(m, q, T, c, d, i, s, HB, HE);
}

// exterior chain, one step: O(j,s) from O(<j,.) and P(.,j,.)  (rules 7, 8).  j >= 1.
template <bool CONSTRAINED>
ELEMDP_HD void inside_ext_target(const ModelView& m, const SeqView& q, const TableView& T, const Constraint& c, int j,
                                 int s) {
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int32_t* G = m.big;
  const double NEG = ELEMDP_NEG_INF;
  const double lam = m.lam(s);
  LseAcc a;
  const int i0 = (j - q.W > 0) ? j - q.W : 0;
  for (int i = j - 1; i >= i0; --i) {  // rule 7
    const int d = j - i;
    if (!q.pair_ok(i, d)) continue;
    const double t = q.e_ext[q.cell(i, d)];
    if (t == NEG) continue;
    const double lt = lam * t;
    for (int u = G[A.split_off + s]; u < G[A.split_off + s + 1]; ++u) {
      const int s2 = G[A.split_ent + 2 * u];      // prefix part (l,h)
      const int s1 = G[A.split_ent + 2 * u + 1];  // pair part   (h,r)
      a.add(T.o(i, s2) + (T.at(ST_P, d, i, s1) + lt));
    }
  }
  if (q.unp[j - 1]) {  // rule 8
    for (int t = I[A.right_off + s]; t < I[A.right_off + s + 1]; ++t) {
      const int s1 = I[A.right_ent + 2 * t], tf = I[A.right_ent + 2 * t + 1];
      if (CONSTRAINED && !allow_right(m, c, q.L, j, s, s1)) continue;
      a.add(T.o(j - 1, s1) + w_right(m, q, s, tf, j - 1));
    }
  }
  T.o(j, s) = a.value();
}

// Z(ari, nasi) (motif_trainer.hpp:108-112)
ELEMDP_HD double lse2(double x, double y) {
  if (y == ELEMDP_NEG_INF) return x;
  if (x == ELEMDP_NEG_INF) return y;
  return x < y ? y + log1p(exp(x - y)) : x + log1p(exp(y - x));
}
ELEMDP_HD double part_func(const ModelView& m, const TableView& T, bool ari, bool nasi) {
  const double a = nasi ? T.o(T.L, m.lay.s00) : ELEMDP_NEG_INF;
  const double b = ari ? T.o(T.L, m.lay.s0m2) : ELEMDP_NEG_INF;
  const double c = ari ? T.o(T.L, m.lay.s0m1) : ELEMDP_NEG_INF;
  return lse2(a, lse2(b, c));
}

// ---------------------------------------------------------------------------------------------
// OUTSIDE (gather form) with expected counts / posteriors
// ---------------------------------------------------------------------------------------------
enum OutsideMode : int { OUT_TRAIN = 0, OUT_SCAN = 1, OUT_END = 2, OUT_NONE = 3 };  // NONE: tables only (BPP filter)

// Sink for the statistics of one transition; implementations: LDS atomics on the GPU, plain adds in
// the CPU emulation.  en(idx, w): EN[idx] += w ; eh(k, w): EH[k] += w ;
// pos(which, p, z): log-space position posteriors, which: 0 = start (PysL), 1 = inner (PyiL), 2 = end (PyeL)
template <class Sink> struct OutCtx {
  const ModelView& m;
  const SeqView& q;
  const TableView& in;
  const TableView& out;
  double Z;
  Constraint c;
  Sink& sink;
};

// statistics of a pair-emission transition: parent P(i-1,j+1,par), child (i,j,ch) -- positions i-1 and j
template <int MODE, class Sink>
ELEMDP_HD bool stat_pair(OutCtx<Sink>& x, int i, int j, int par, int ch, double z) {
  if (MODE == OUT_NONE) return true;
  const ModelView& m = x.m;
  const int k = i - 1;
  const int pl = m.st_l(par), pr = m.st_r(par), cl = m.st_l(ch), cr = m.st_r(ch);
  const int M = m.lay.M;
  if (MODE == OUT_END) {  // motif_scanner.hpp:715-724
    if (x.c.ys == k && !(pl == 0 && cl == 1)) return false;
    if (x.c.ys == j && !(cr == 0 && pr == 1)) return false;
    if (pl == M - 2 && cl == M - 1) x.sink.pos(2, k, z);
    if (cr == M - 2 && pr == M - 1) x.sink.pos(2, j, z);
    if (pr == M - 2 && x.q.L == j + 1) x.sink.pos(2, x.q.L, z);
    return true;
  }
  if (!m.no_prf) {  // motif_trainer.hpp:384-388 / profile_hmm.hpp:144-179
    const double w = exp(z);
    const int bi = x.q.seq[k], bj = x.q.seq[j];
    if (m.ints[m.lay.st_pair_r + par]) {
      const int t = bp_type(bi, bj);
      if (t) x.sink.en(m.param_index(m.ints[m.lay.st_row_r + par], t - 1), w);
    } else {
      if (bi) x.sink.en(m.param_index(m.ints[m.lay.st_row_l + ch], bi - 1), w);
      if (bj) x.sink.en(m.param_index(m.ints[m.lay.st_row_r + par], bj - 1), w);
    }
  }
  if (MODE == OUT_SCAN) {  // motif_scanner.hpp:546-553
    if (pl == 0 && cl == 1) x.sink.pos(0, k, z);
    if (cr == 0 && pr == 1) x.sink.pos(0, j, z);
    if (cl != 0 && cl != M - 1) x.sink.pos(1, k, z);
    if (pr != 0 && pr != M - 1) x.sink.pos(1, j, z);
  }
  return true;
}
// right-emission transition: parent (.,j+1,par), child (.,j,ch) -- position j
template <int MODE, class Sink>
ELEMDP_HD bool stat_right(OutCtx<Sink>& x, int j, int par, int ch, double z) {
  if (MODE == OUT_NONE) return true;
  const ModelView& m = x.m;
  const int pr = m.st_r(par), cr = m.st_r(ch);
  const int M = m.lay.M;
  if (MODE == OUT_END) {  // motif_scanner.hpp:730-738
    if (x.c.ys == j && !(cr == 0 && pr == 1)) return false;
    if (cr == M - 2 && pr == M - 1) x.sink.pos(2, j, z);
    if (pr == M - 2 && x.q.L == j + 1) x.sink.pos(2, x.q.L, z);
    return true;
  }
  if (!m.no_prf) {
    const int b = x.q.seq[j];
    if (b) x.sink.en(m.param_index(m.ints[m.lay.st_row_r + par], b - 1), exp(z));
  }
  if (MODE == OUT_SCAN) {  // :559-564
    if (cr == 0 && pr == 1) x.sink.pos(0, j, z);
    if (pr != 0 && pr != M - 1) x.sink.pos(1, j, z);
  }
  return true;
}
// left-emission transition: parent M(i-1,j,par), child M(i,j,ch) -- position i-1
template <int MODE, class Sink>
ELEMDP_HD bool stat_left(OutCtx<Sink>& x, int i, int par, int ch, double z) {
  if (MODE == OUT_NONE) return true;
  const ModelView& m = x.m;
  const int k = i - 1;
  const int pl = m.st_l(par), cl = m.st_l(ch);
  const int M = m.lay.M;
  if (MODE == OUT_END) {  // :741-747
    if (x.c.ys == k && !(pl == 0 && cl == 1)) return false;
    if (pl == M - 2 && cl == M - 1) x.sink.pos(2, k, z);
    return true;
  }
  if (!m.no_prf) {
    const int b = x.q.seq[k];
    if (b) x.sink.en(m.param_index(m.ints[m.lay.st_row_l + ch], b - 1), exp(z));
  }
  if (MODE == OUT_SCAN) {  // :568-573
    if (pl == 0 && cl == 1) x.sink.pos(0, k, z);
    if (cl != 0 && cl != M - 1) x.sink.pos(1, k, z);
  }
  return true;
}
// energy-gradient statistic: EH[idx(parent)] += tsc * e^z (motif_trainer.hpp:380-381)
template <int MODE, class Sink> ELEMDP_HD void stat_energy(OutCtx<Sink>& x, int par, double tsc, double z) {
  if (MODE == OUT_TRAIN) x.sink.eh(x.m.eh_index(par), tsc * exp(z));
}

// exterior chain backwards, one step: outside_o(i, s) for i < L from outside_o(>i,.) (rules 8, 7 reversed).
// Also accounts the statistics of those transitions.
template <int MODE, class Sink> ELEMDP_HD void outside_ext_target(OutCtx<Sink>& x, int i, int s) {
  const ModelView& m = x.m;
  const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int32_t* G = m.big;
  const double NEG = ELEMDP_NEG_INF;
  const double in_c = x.in.o(i, s);
  if (in_c == NEG) { x.out.o(i, s) = NEG; return; }
  LseAcc a;
  // rule 8: parent O(i+1, par) with s in right(par); emits position i
  if (q.unp[i]) {
    for (int t = I[A.rright_off + s]; t < I[A.rright_off + s + 1]; ++t) {
      const int par = I[A.rright_ent + 2 * t], tf = I[A.rright_ent + 2 * t + 1];
      const double term = x.out.o(i + 1, par) + w_right(m, q, par, tf, i);
      const double z = term + in_c - x.Z;
      if (z == NEG) continue;
      if (!stat_right<MODE>(x, i, par, s, z)) continue;
      a.add(term);
    }
  }
  // rule 7: parent O(j, par), this = prefix part s2=(l,h), sibling P(i,j,(h,r))
  const int jmax = (i + q.W < q.L) ? i + q.W : q.L;
  for (int j = i + 1; j <= jmax; ++j) {
    const int d = j - i;
    if (!q.pair_ok(i, d)) continue;
    const double t = q.e_ext[q.cell(i, d)];
    if (t == NEG) continue;
    for (int u = G[A.split1_off + s]; u < G[A.split1_off + s + 1]; ++u) {
      const int par = G[A.split1_ent + 2 * u], s1 = G[A.split1_ent + 2 * u + 1];
      const double term = x.out.o(j, par) + (x.in.at(ST_P, d, i, s1) + m.lam(par) * t);
      const double z = term + in_c - x.Z;
      if (z == NEG) continue;
      stat_energy<MODE>(x, par, t, z);
      a.add(term);
    }
  }
  x.out.o(i, s) = a.value();
}

// ---- heavy sums of the outside pass (gathers over strictly larger cells) -----------------------
ELEMDP_HD double o1_term(const TableView& in, const TableView& out, int i, int j, int jj, int par, int s2) {
  return out.at(ST_B, jj - i, i, par) + in.at(ST_2, jj - j, j, s2);
}
ELEMDP_HD double o2_term(const TableView& in, const TableView& out, int i, int j, int ii, int par, int s1) {
  return out.at(ST_B, j - ii, ii, par) + in.at(ST_1, i - ii, ii, s1);
}
// inner pair P(i,j) of the interior loop `it` (outer cell E(it.i, it.j))
ELEMDP_HD double oP_term(const TableView& in, const TableView& out, int i, int j, const LoopItem& it, int par, int s2,
                         int s3, double lt) {
  return out.at(ST_E, it.j - it.i, it.i, par) + (in.at(ST_L, i - it.i, it.i, s2) + (in.at(ST_L, it.j - j, j, s3) + lt));
}
// left loop L(it.i, it.k) / right loop L(it.l, it.j) of the interior loop `it`
ELEMDP_HD double oLl_term(const TableView& in, const TableView& out, const LoopItem& it, int par, int s1, int s3, double lt) {
  return out.at(ST_E, it.j - it.i, it.i, par) + (in.at(ST_P, it.l - it.k, it.k, s1) + (in.at(ST_L, it.j - it.l, it.l, s3) + lt));
}
ELEMDP_HD double oLr_term(const TableView& in, const TableView& out, const LoopItem& it, int par, int s1, int s2, double lt) {
  return out.at(ST_E, it.j - it.i, it.i, par) + (in.at(ST_P, it.l - it.k, it.k, s1) + (in.at(ST_L, it.k - it.i, it.i, s2) + lt));
}
// first admissible end jj of a parent B(i,jj) of the child 1(i,j): jj >= j + dmin[j]
ELEMDP_HD bool o2_valid(const SeqView& q, int i, int ii) {  // is_parsable<ST_1>(ii, i)
  const int di = q.dmin[ii];
  return di != 0 && i - ii >= di;
}

// H1: 1(i,j,s) as the "1" child of B(i,jj,par), jj > j (rule 2)
template <class Sink> ELEMDP_HD double heavy_o1(OutCtx<Sink>& x, int d, int i, int s) {
  const ModelView& m = x.m; const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay; const int32_t* G = m.big;
  const int j = i + d;
  LseAcc a;
  const int dj = q.dmin[j];
  if (j < q.L && dj > 0) {
    const int jmax = (i + q.W < q.L) ? i + q.W : q.L;
    const int u0 = G[A.split1_off + s];
    if (G[A.split1_off + s + 1] - u0 == 1) {
      const int par = G[A.split1_ent + 2 * u0], s2 = G[A.split1_ent + 2 * u0 + 1], j_lo = j + dj;
      lse_run4(a, jmax - j_lo + 1, [&](int) { return true; }, [&](int n) { return o1_term(x.in, x.out, i, j, j_lo + n, par, s2); });
      return a.value();
    }
    for (int jj = j + dj; jj <= jmax; ++jj)
      for (int u = G[A.split1_off + s]; u < G[A.split1_off + s + 1]; ++u)
        a.add(o1_term(x.in, x.out, i, j, jj, G[A.split1_ent + 2 * u], G[A.split1_ent + 2 * u + 1]));
  }
  return a.value();
}
// H2: 2(i,j,s) as the "2" child of B(ii,j,par), ii < i (rule 2)
template <class Sink> ELEMDP_HD double heavy_o2(OutCtx<Sink>& x, int d, int i, int s) {
  const ModelView& m = x.m; const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay; const int32_t* G = m.big;
  const int j = i + d;
  LseAcc a;
  const int imin = (j - q.W > 0) ? j - q.W : 0;
  const int u0 = G[A.split2_off + s];
  if (G[A.split2_off + s + 1] - u0 == 1) {
    const int par = G[A.split2_ent + 2 * u0], s1 = G[A.split2_ent + 2 * u0 + 1];
    lse_run4(a, i - imin, [&](int n) { return o2_valid(q, i, i - 1 - n); },
             [&](int n) { return o2_term(x.in, x.out, i, j, i - 1 - n, par, s1); });
    return a.value();
  }
  for (int ii = i - 1; ii >= imin; --ii) {
    if (!o2_valid(q, i, ii)) continue;
    for (int u = G[A.split2_off + s]; u < G[A.split2_off + s + 1]; ++u)
      a.add(o2_term(x.in, x.out, i, j, ii, G[A.split2_ent + 2 * u], G[A.split2_ent + 2 * u + 1]));
  }
  return a.value();
}
// HP: P(i,j,s) as inner pair of the interior loops around it (rule 6c) + their energy statistic
template <int MODE, class Sink> ELEMDP_HD double heavy_oP(OutCtx<Sink>& x, int d, int i, int s) {
  const ModelView& m = x.m; const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay; const int32_t* G = m.big;
  const int j = i + d;
  const double in_c = x.in.at(ST_P, d, i, s);
  LseAcc a;
  if (in_c == ELEMDP_NEG_INF) return ELEMDP_NEG_INF;
  const int pc = q.cell(i, d);
  const int u0 = G[A.quad1_off + s];
  if (MODE == OUT_NONE && G[A.quad1_off + s + 1] - u0 == 1) {   // (no statistics: the BPP filter)
    const int par = G[A.quad1_ent + 3 * u0], s2 = G[A.quad1_ent + 3 * u0 + 1], s3 = G[A.quad1_ent + 3 * u0 + 2];
    const int n0 = q.by_inner_off[pc];
    const double lam = m.lam(par);
    lse_run4(a, q.by_inner_off[pc + 1] - n0, [&](int) { return true; }, [&](int n) {
      const LoopItem it = q.items[q.by_inner_idx[n0 + n]];
      return oP_term(x.in, x.out, i, j, it, par, s2, s3, lam * it.tsc);
    });
    return a.value();
  }
  for (int n = q.by_inner_off[pc]; n < q.by_inner_off[pc + 1]; ++n) {
    const LoopItem it = q.items[q.by_inner_idx[n]];
    for (int u = G[A.quad1_off + s]; u < G[A.quad1_off + s + 1]; ++u) {
      const int par = G[A.quad1_ent + 3 * u];
      const double term = oP_term(x.in, x.out, i, j, it, par, G[A.quad1_ent + 3 * u + 1], G[A.quad1_ent + 3 * u + 2],
                                  m.lam(par) * it.tsc);
      const double z = term + in_c - x.Z;
      if (z == ELEMDP_NEG_INF) continue;
      stat_energy<MODE>(x, par, it.tsc, z);
      a.add(term);
    }
  }
  return a.value();
}
// HL: L(i,j,s) as left loop (cell (it.i,it.k) = (i,j)) or right loop (cell (it.l,it.j) = (i,j)) of interior loops
template <class Sink> ELEMDP_HD double heavy_oL(OutCtx<Sink>& x, int d, int i, int s) {
  const ModelView& m = x.m; const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay; const int32_t* G = m.big;
  LseAcc a;
  const int lc = q.cell(i, d);
  for (int n = q.by_left_off[lc]; n < q.by_left_off[lc + 1]; ++n) {
    const LoopItem it = q.items[q.by_left_idx[n]];
    for (int u = G[A.quad2_off + s]; u < G[A.quad2_off + s + 1]; ++u) {
      const int par = G[A.quad2_ent + 3 * u];
      a.add(oLl_term(x.in, x.out, it, par, G[A.quad2_ent + 3 * u + 1], G[A.quad2_ent + 3 * u + 2], m.lam(par) * it.tsc));
    }
  }
  for (int n = q.by_right_off[lc]; n < q.by_right_off[lc + 1]; ++n) {
    const LoopItem it = q.items[q.by_right_idx[n]];
    for (int u = G[A.quad3_off + s]; u < G[A.quad3_off + s + 1]; ++u) {
      const int par = G[A.quad3_ent + 3 * u];
      a.add(oLr_term(x.in, x.out, it, par, G[A.quad3_ent + 3 * u + 1], G[A.quad3_ent + 3 * u + 2], m.lam(par) * it.tsc));
    }
  }
  return a.value();
}

// (ext_in_hp: HP already holds the rule-7 term of the P child -- the scaled-linear kernels gather it with the other heavy sums)
struct HeavyOut { double H1, H2, HP, HL; bool ext_in_hp = false; };

// band target (i,d,s), outside direction, given the heavy sums.  Requires every larger diagonal and the
// whole exterior chain outside_o to be final.
template <int MODE, class Sink> ELEMDP_HD void outside_target_u(OutCtx<Sink>& x, int d, int i, int s, const HeavyOut& H) {
  const ModelView& m = x.m;
  const SeqView& q = x.q;
  const TableView& in = x.in;
  const TableView& out = x.out;
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int32_t* G = m.big;
  const double NEG = ELEMDP_NEG_INF;
  const double Z = x.Z;
  const int j = i + d;
  const bool isloop = I[A.st_is_loop + s] != 0;
  const bool pok = q.pair_ok(i, d);
  const bool lok = q.left_ok(i, d);
  const bool mok = m_ok(m, q, i, d);
  const bool eok = q.e_ok(i, d);
  const bool up_ok = q.pair_ok(i - 1, d + 2);  // enclosing pair cell (i-1, j+1)
  const bool doM = mok && m_ok(m, q, i - 1, d + 1) && q.unp[i - 1];
  const bool do2 = lok && q.left_ok(i, d + 1) && q.unp[j];
  const bool doL = isloop && j < q.L && d + 1 <= q.W;

  // ---- operand fetch: inside values of this target and the parents' outside values (diagonals d+1, d+2)
  const double rE = in.at(ST_E, d, i, s), rM = in.at(ST_M, d, i, s), r1 = in.at(ST_1, d, i, s), rB = in.at(ST_B, d, i, s);
  const double r2 = in.at(ST_2, d, i, s), rP = in.at(ST_P, d, i, s), rL = in.at(ST_L, d, i, s);
  const double inE = eok ? rE : NEG;
  const double inM = mok ? rM : NEG;
  const double in1 = lok ? r1 : NEG;
  const double inB = lok ? rB : NEG;
  const double in2 = lok ? r2 : NEG;
  const double inP = pok ? rP : NEG;
  const double inL = isloop ? rL : NEG;
  double ecl = NEG, ehp = NEG, est_up = NEG, eml = NEG, eext = NEG;
  if (eok) { const int pc = q.cell(i - 1, d + 2); ecl = q.e_close[pc]; ehp = q.e_hp[pc]; }
  if (up_ok && pok) est_up = q.e_stack[q.cell(i - 1, d + 2)];
  if (pok) { eml = q.e_ml[q.cell(i, d)]; eext = q.e_ext[q.cell(i, d)]; }
  const int rp0 = I[A.rpair_off + s], nRP = I[A.rpair_off + s + 1] - rp0;
  const int rl0 = I[A.rleft_off + s], nRL = I[A.rleft_off + s + 1] - rl0;
  const int rr0 = I[A.rright_off + s], nRR = I[A.rright_off + s + 1] - rr0;
  double yP[kUnary], yM[kUnary], y2[kUnary], yL[kUnary];
  const bool needP = (up_ok && (inE != NEG || (inP != NEG && est_up != NEG)));
  const int dp1 = d + 1 <= q.W ? d + 1 : q.W, dp2 = d + 2 <= q.W ? d + 2 : q.W, im1 = i > 0 ? i - 1 : 0;
#pragma unroll
  for (int u = 0; u < kUnary; ++u) {  // unconditional loads from clamped cells, masked afterwards (see inside_target_u)
    const int par_p = (u < nRP) ? I[A.rpair_ent + 2 * (rp0 + u)] : 0;
    const int par_l = (u < nRL) ? I[A.rleft_ent + 2 * (rl0 + u)] : 0;
    const int par_r = (u < nRR) ? I[A.rright_ent + 2 * (rr0 + u)] : 0;
    const double tP = out.at(ST_P, dp2, im1, par_p), tM = out.at(ST_M, dp1, im1, par_l);
    const double t2 = out.at(ST_2, dp1, i, par_r), tL = out.at(ST_L, dp1, i, par_r);
    yP[u] = (needP && u < nRP) ? tP : NEG;
    yM[u] = (doM && inM != NEG && u < nRL) ? tM : NEG;
    y2[u] = (do2 && in2 != NEG && u < nRR) ? t2 : NEG;
    yL[u] = (doL && inL != NEG && u < nRR && I[A.st_is_loop + par_r]) ? tL : NEG;
  }

  // NB every transition's posterior z contains inside(child); the reference drops transitions with
  // z == log 0 *before* touching the outside table (motif_trainer.hpp:378), so a child whose inside
  // value is log 0 keeps outside = log 0 -- hence the `in != NEG` guards below.

  // ---- E(i,j,s) as child of P(i-1,j+1,par): rule 1a
  double oE = NEG;
  if (inE != NEG) {
    LseAcc a;
    for (int u = 0; u < nRP; ++u) {
      const int par = I[A.rpair_ent + 2 * (rp0 + u)], tf = I[A.rpair_ent + 2 * (rp0 + u) + 1];
      const double op = (u < kUnary) ? yP[u < kUnary ? u : 0] : out.at(ST_P, d + 2, i - 1, par);
      const double term = op + w_pair(m, q, par, s, tf, i - 1, j);
      const double z = term + inE - Z;
      if (z == NEG) continue;
      if (!stat_pair<MODE>(x, i, j, par, s, z)) continue;
      a.add(term);
    }
    oE = a.value();
  }
  out.at(ST_E, d, i, s) = oE;

  // ---- M(i,j,s): child of E(i,j,s) (6a) and of M(i-1,j,par) (5a)
  double oM = NEG;
  if (inM != NEG) {
    LseAcc a;
    if (eok && ecl != NEG) {
      const double term = oE + m.lam(s) * ecl;
      const double z = term + inM - Z;
      if (z != NEG) { stat_energy<MODE>(x, s, ecl, z); a.add(term); }
    }
    if (doM) {
      for (int u = 0; u < nRL; ++u) {
        const int par = I[A.rleft_ent + 2 * (rl0 + u)], tf = I[A.rleft_ent + 2 * (rl0 + u) + 1];
        const double op = (u < kUnary) ? yM[u < kUnary ? u : 0] : out.at(ST_M, d + 1, i - 1, par);
        const double term = op + w_left(m, q, s, tf, i - 1);
        const double z = term + inM - Z;
        if (z == NEG) continue;
        if (!stat_left<MODE>(x, i, par, s, z)) continue;
        a.add(term);
      }
    }
    oM = a.value();
  }
  out.at(ST_M, d, i, s) = oM;

  // ---- 1(i,j,s): heavy sum H1 ; B(i,j,s): child of M (5b) and of 1 (4b)
  double o1 = NEG, oB = NEG, o2 = NEG;
  if (lok) {
    if (in1 != NEG) o1 = H.H1;
    if (inB != NEG) {
      LseAcc a;
      if (mok) a.add(oM);
      a.add(o1);
      oB = a.value();
    }
    // ---- 2(i,j,s): child of 1 (4a), of 2(i,j+1,par) (3a), and heavy sum H2
    if (in2 != NEG) {
      LseAcc a;
      a.add(o1);
      if (do2) {
        for (int u = 0; u < nRR; ++u) {
          const int par = I[A.rright_ent + 2 * (rr0 + u)], tf = I[A.rright_ent + 2 * (rr0 + u) + 1];
          const double op = (u < kUnary) ? y2[u < kUnary ? u : 0] : out.at(ST_2, d + 1, i, par);
          const double term = op + w_right(m, q, par, tf, j);
          const double z = term + in2 - Z;
          if (z == NEG) continue;
          if (!stat_right<MODE>(x, j, par, s, z)) continue;
          a.add(term);
        }
      }
      a.add(H.H2);
      o2 = a.value();
    }
  }
  out.at(ST_1, d, i, s) = o1;
  out.at(ST_B, d, i, s) = oB;
  out.at(ST_2, d, i, s) = o2;

  // ---- P(i,j,s): child of O (7), of P(i-1,j+1,par) (1b), of 2 (3b), inner pair of E(i',j') (6c: heavy sum HP)
  double oP = NEG;
  if (inP != NEG) {
    LseAcc a;
    if (eext != NEG)  // rule 7: parent O(j, par), sibling prefix O(i, s2); this = pair part (h,r)
      for (int u = G[A.split2_off + s]; u < G[A.split2_off + s + 1]; ++u) {
        const int par = G[A.split2_ent + 2 * u], s2 = G[A.split2_ent + 2 * u + 1];
        a.add(out.o(j, par) + (in.o(i, s2) + m.lam(par) * eext));
      }
    if (up_ok && est_up != NEG)  // rule 1b
      for (int u = 0; u < nRP; ++u) {
        const int par = I[A.rpair_ent + 2 * (rp0 + u)], tf = I[A.rpair_ent + 2 * (rp0 + u) + 1];
        const double op = (u < kUnary) ? yP[u < kUnary ? u : 0] : out.at(ST_P, d + 2, i - 1, par);
        const double term = op + (w_pair(m, q, par, s, tf, i - 1, j) + m.lam(par) * est_up);
        const double z = term + inP - Z;
        if (z == NEG) continue;
        if (!stat_pair<MODE>(x, i, j, par, s, z)) continue;
        stat_energy<MODE>(x, par, est_up, z);
        a.add(term);
      }
    if (eml != NEG) {  // rule 3b
      const double term = o2 + m.lam(s) * eml;
      const double z = term + inP - Z;
      if (z != NEG) { stat_energy<MODE>(x, s, eml, z); a.add(term); }
    }
    a.add(H.HP);
    oP = a.value();
  }
  out.at(ST_P, d, i, s) = oP;

  // ---- L(i,j,s): child of E (6b), of L(i,j+1,par), loops of interior loops (6c: heavy sum HL)
  double oL = NEG;
  if (inL != NEG) {
    LseAcc a;
    if (eok && ehp != NEG) {
      const double term = oE + m.lam(s) * ehp;
      const double z = term + inL - Z;
      if (z != NEG) { stat_energy<MODE>(x, s, ehp, z); a.add(term); }
    }
    if (doL) {
      for (int u = 0; u < nRR; ++u) {
        const int par = I[A.rright_ent + 2 * (rr0 + u)], tf = I[A.rright_ent + 2 * (rr0 + u) + 1];
        if (!I[A.st_is_loop + par]) continue;
        const double op = (u < kUnary) ? yL[u < kUnary ? u : 0] : out.at(ST_L, d + 1, i, par);
        const double term = op + w_right(m, q, par, tf, j);
        const double z = term + inL - Z;
        if (z == NEG) continue;
        if (!stat_right<MODE>(x, j, par, s, z)) continue;
        a.add(term);
      }
    }
    a.add(H.HL);
    oL = a.value();
  }
  out.at(ST_L, d, i, s) = oL;
}

// which heavy sums a target needs (the GPU skips the gathers otherwise)
ELEMDP_HD bool needs_o1(const SeqView& q, const TableView& in, int d, int i, int s) { return q.left_ok(i, d) && in.at(ST_1, d, i, s) != ELEMDP_NEG_INF; }

// serial form: heavy sums evaluated in place (CPU emulation, S = 1 paths)
template <int MODE, class Sink> ELEMDP_HD void outside_target(OutCtx<Sink>& x, int d, int i, int s) {
  const SeqView& q = x.q;
  HeavyOut H;
  const bool lok = q.left_ok(i, d);
  H.H1 = (lok && x.in.at(ST_1, d, i, s) != ELEMDP_NEG_INF) ? heavy_o1(x, d, i, s) : ELEMDP_NEG_INF;
  H.H2 = (lok && x.in.at(ST_2, d, i, s) != ELEMDP_NEG_INF) ? heavy_o2(x, d, i, s) : ELEMDP_NEG_INF;
  H.HP = q.pair_ok(i, d) ? heavy_oP<MODE>(x, d, i, s) : ELEMDP_NEG_INF;
  // (OUT_NONE = BPP filter: nothing reads the outside value of a loop cell, and its plan has no by_left / by_right order)
  H.HL = (MODE != OUT_NONE && x.m.ints[x.m.lay.st_is_loop + s] && x.in.at(ST_L, d, i, s) != ELEMDP_NEG_INF) ? heavy_oL(x, d, i, s) : ELEMDP_NEG_INF;
  outside_target_u<MODE>(x, d, i, s, H);
}

}  // namespace elemdp
