// lin_rules.h -- the train evaluation in the SCALED LINEAR semiring (host/device-agnostic rule code).
//
// Same gather formulation as dp_rules.h (rules of SURVEY.md Appendix A; reference
// RNAelem/motif_trainer.hpp:124-272 over energy_model.hpp:340-547 x motif_model.hpp:230-613), but the
// tables hold Boltzmann weights instead of their logarithms:  (+) is a plain add, (x) a multiply, so a
// term of the two O(L W^2) rules costs one FMA instead of one exp.  What keeps the numbers in range is
// a similarity transform: every emission of the base at position p carries the factor psb[base(p)], an
// exact power of two chosen per evaluation so that the background emission has weight ~1.  Every parse
// emits every position exactly once, hence
//     inside(i,j,.)  = true value * prod_{p in [i,j)} psb      outside(i,j,.) = true * prod_{p notin [i,j)} psb
//     O(j,.)         = true * prod_{p < j} psb                 Z = true * prod_{p < L} psb
// and every posterior  in * out * w / Z  -- all the train evaluation needs -- is scale free.  Powers of
// two make the transform exact.  A sequence whose partition functions leave the double range (0, inf or
// NaN) is flagged and re-evaluated by the log-space pipeline (train_kernels.hip), which also decides the
// reference's skip rule (motif_trainer.hpp:211-215) for it.
//
// Guards mirror dp_rules.h: "x == log 0" becomes "x == 0".
//
// Tables are COMPACT (TableView::ld / st / lda, dp_rules.h): a plane keeps columns only for the interval states that are
// useful in it, and nothing is stored for a cell that is not parsable in the plane (is_parsable<e>, energy_model.hpp:289-338:
// P, E at kept pairs; B, 1, 2 where left_ok; M where m_ok; pair tables where 0 < dmin[i] < d).  Every load therefore
// names the liveness of its operand (`live`), known from the pair mask and dmin alone; a dead operand reads as 0 without
// touching the table.
#pragma once
#include <cmath>
#include <cstdint>

#include "dp_rules.h"
#include "lin_params.h"

namespace elemdp {

// planes of the exponentiated structural terms: xwc[(k*5 + term) * stride + cell], k = lambda class of the rule's parent
enum LinTerm : int { XT_STACK = 0, XT_EXT = 1, XT_ML = 2, XT_CLOSE = 3, XT_HP = 4 };

ELEMDP_HD double lin_weight(double lam, double e) { return (e == ELEMDP_NEG_INF) ? 0. : exp(lam * e); }

// ---- emission weights (linear forms of w_right / w_left / w_pair) --------------------------------
ELEMDP_HD double lw_right(const ModelView& m, const SeqView& q, int par, int tau_flag, int pos) {
  const int b = q.seq[pos];
  double w = (m.no_prf || b == 0) ? 1. : m.lin[kLinEth + m.param_index(m.ints[m.lay.st_row_r + par], b - 1)];
  const double ws = m.ints[m.lay.st_w_r + par] ? q.ews[pos] : 1.;
  const double t = tau_flag ? m.lin[kLinTau] : 1.;
  return w * (t * ws);
}
ELEMDP_HD double lw_left(const ModelView& m, const SeqView& q, int child, int tau_flag, int pos) {
  const int b = q.seq[pos];
  double w = (m.no_prf || b == 0) ? 1. : m.lin[kLinEth + m.param_index(m.ints[m.lay.st_row_l + child], b - 1)];
  const double ws = m.ints[m.lay.st_w_l + child] ? q.ews[pos] : 1.;
  const double t = tau_flag ? m.lin[kLinTau] : 1.;
  return w * (t * ws);
}
ELEMDP_HD double lw_pair(const ModelView& m, const SeqView& q, int par, int child, int tau_flag, int pi, int pj) {
  const int bi = q.seq[pi], bj = q.seq[pj];
  double w = 1.;
  if (!m.no_prf) {
    if (m.ints[m.lay.st_pair_r + par]) {
      const int t = bp_type(bi, bj);
      w = t ? m.lin[kLinEth + m.param_index(m.ints[m.lay.st_row_r + par], t - 1)] : m.lin[kLinPsb + bi] * m.lin[kLinPsb + bj];
    } else {
      w = (bi ? m.lin[kLinEth + m.param_index(m.ints[m.lay.st_row_l + child], bi - 1)] : 1.) *
          (bj ? m.lin[kLinEth + m.param_index(m.ints[m.lay.st_row_r + par], bj - 1)] : 1.);
    }
  }
  const double ws = (m.ints[m.lay.st_w_l + child] ? q.ews[pi] : 1.) * (m.ints[m.lay.st_w_r + par] ? q.ews[pj] : 1.);
  const double t = tau_flag ? m.lin[kLinTau] : 1.;
  return w * (t * ws);
}
ELEMDP_HD int lamk(const ModelView& m, int s) { return m.ints[m.lay.st_lam + s]; }
ELEMDP_HD double xw_cell(const SeqView& q, int k, int term, int cell) { return q.xwc[(size_t)(k * 5 + term) * q.xwc_stride + cell]; }
ELEMDP_HD double xw_item(const SeqView& q, int k, int it) { return q.xwi[(size_t)k * q.xwi_stride + it]; }

// ---- rule 2 through the pair-sparse factorisation ---------------------------------------------------
// B(i,j,s) = sum_k sum_{(s1,t) in split(s)} 1(i,k,s1) * 2(k,j,t) costs O(W) terms per cell and tuple.  But 2(k,j,.) is a stem
// P(k,l,.) followed by a tail of unpaired bases (rules 3a, 3b), and the pairs that survive the BPP filter are few.  With
//     A(i,j,(s1,t)) := sum_{i<k<j} 1(i,k,s1) * 2(k,j,t)
// rules 3a / 3b give a recurrence along the row (the tail grows by one right emission, or a stem ends at j):
//     A(i,j,(s1,t)) = [unp(j-1)] sum_{t' in right(t)} A(i,j-1,(s1,t')) w_right(t,t',j-1)
//                   + sum_{k: (k,j) kept, i<k} 1(i,k,s1) P(k,j,t) xml(k,j)
// and B(i,j,s) = sum of A(i,j,p) over the pairs p = (s1,t) with (s; s1,t) a split.  (The guards of rule 3a hold by
// themselves: 2(k,j-1,t') != 0 implies left_ok(k,.) for both spans, and j-k < d <= W.)  Same sums as the reference
// (energy_model.hpp:358-365 x motif_model.hpp:368-381) in another association; one O(pairs ending at j) loop per cell
// instead of O(W) operand rows.  The table T.ap holds A per (diagonal, cell, pair).
template <bool CON = false>
ELEMDP_HD double lin_inside_apair(const ModelView& m, const SeqView& q, const TableView& T, int d, int i, int p,
                                  const Constraint& con = Constraint{-1, -1, 0}) {
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int j = i + d;
  const int s1 = I[A.ap_s1 + p], t = I[A.ap_t + p];
  const int dmi = q.dmin[i];
  if (!(dmi > 0 && dmi < d)) return 0.;    // no k with 1(i,k,.) != 0 in front of j: the entry does not exist
  double a = 0.;
  if (d - 1 > dmi && q.unp[j - 1])         // (the entry of (i, j-1) exists iff dmin[i] < d - 1)
    for (int e = I[A.ap_chain_off + p]; e < I[A.ap_chain_off + p + 1]; ++e) {
      const int pc = I[A.ap_chain_ent + 2 * e], tf = I[A.ap_chain_ent + 2 * e + 1];
      if (CON && !allow_right(m, con, q.L, j, t, I[A.ap_t + pc])) continue;
      a = fma(T.a(d - 1, i, pc), lw_right(m, q, t, tf, j - 1), a);
    }
  const int kl = lamk(m, t);
  // stems (k, j) that end at j and start at k >= i + dmin[i] (1(i,k,.) is parsable): spans 1 .. d-dmin[i] of row j of the
  // end-indexed pair mask
  for_mask_bits(q.okbits_end, j * (q.W + 1), 1, d - dmi, [&](int sp) {
    const int k = j - sp;
    a = fma(T.ld(ST_1, k - i, i, s1), T.ld(ST_P, sp, k, t) * xw_cell(q, kl, XT_ML, q.cell(k, sp)), a);
  });
  T.a(d, i, p) = a;
  return a;
}
// all pairs of one cell (serial form: CPU emulation)
template <bool CON = false>
ELEMDP_HD void lin_inside_cell_pairs(const ModelView& m, const SeqView& q, const TableView& T, int d, int i,
                                     const Constraint& con = Constraint{-1, -1, 0}) {
  for (int p = 0; p < m.lay.n_ap; ++p) lin_inside_apair<CON>(m, q, T, d, i, p, con);
}
// B(i,j,s) from the pair table of the same cell
ELEMDP_HD double lheavy_bif(const ModelView& m, const SeqView& q, const TableView& T, int d, int i, int s) {
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  double a = 0.;
  const int dmi = q.dmin[i];
  if (!(dmi > 0 && dmi < d)) return 0.;
  for (int p = 0; p < A.n_ap; ++p)
    if (I[A.ap_tgt + p] == s) a += T.a(d, i, p);
  return a;
}
ELEMDP_HD double lloop_term(const TableView& T, int i, int j, const LoopItem& x, int s1, int s2, int s3) {
  return T.ld(ST_P, x.l - x.k, x.k, s1) * (T.ld(ST_L, x.k - i, i, s2) * T.ld(ST_L, j - x.l, x.l, s3));
}
ELEMDP_HD double lheavy_loop(const ModelView& m, const SeqView& q, const TableView& T, int d, int i, int s) {
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.big;
  const int j = i + d, k = lamk(m, s);
  double a = 0.;
  const int c0 = q.by_outer_off[q.cell(i, d)], c1 = q.by_outer_off[q.cell(i, d) + 1];
  for (int it = c0; it < c1; ++it) {
    if (!q.item_in[it]) continue;
    const LoopItem x = q.items[it];
    const double xw = xw_item(q, k, it);
    for (int t = G[A.quad_off + s]; t < G[A.quad_off + s + 1]; ++t)
      a = fma(lloop_term(T, i, j, x, G[A.quad_ent + 3 * t], G[A.quad_ent + 3 * t + 1], G[A.quad_ent + 3 * t + 2]), xw, a);
  }
  return a;
}

// does the compact table keep a row for plane e at cell (i, d)?  (is_parsable<e>, energy_model.hpp:289-338)
ELEMDP_HD bool lin_cell_live(const ModelView& m, const SeqView& q, int e, int d, int i) {
  if (d < 0 || d > q.W || i < 0 || i + d > q.L) return false;
  switch (e) {
    case ST_P: return q.pair_ok(i, d);
    case ST_E: return q.e_ok(i, d);
    case ST_M: return m_ok(m, q, i, d);
    case ST_B: case ST_1: case ST_2: return q.left_ok(i, d);
    default: return true;
  }
}
// value of an entry for exports and tests: 0 where nothing is stored
ELEMDP_HD double lin_get(const ModelView& m, const SeqView& q, const TableView& T, int e, int d, int i, int s) {
  return T.ld(e, d, i, s, lin_cell_live(m, q, e, d, i));
}

struct Cell7 { double vP, vE, vM, vB, v1, v2, vL; };

// P,E,M,B,1,2,L of target (i,d,s) from the heavy sums HB (rule 2) and HE (rule 6c); stores and returns them.
// CON: the start constraint of the scan's second pass (c.ys) is applied to the emitting rules (motif_scanner.hpp:594-622)
template <bool CON = false>
ELEMDP_HD Cell7 lin_inside_target_u(const ModelView& m, const SeqView& q, const TableView& T, int d, int i, int s, double HB,
                                    double HE, const Constraint& con = Constraint{-1, -1, 0}) {
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int j = i + d;
  const int kl = lamk(m, s);
  const bool isloop = I[A.st_is_loop + s] != 0;
  const bool pok = q.pair_ok(i, d);
  const bool lok = q.left_ok(i, d);
  const bool mok = m_ok(m, q, i, d);
  const bool eok = q.e_ok(i, d);
  const bool doL = isloop && d > 0;
  const bool do2 = lok && q.left_ok(i, d - 1) && q.unp[j - 1];
  const bool doM = mok && m_ok(m, q, i + 1, d - 1) && q.unp[i];

  const int r0 = I[A.right_off + s], nR = I[A.right_off + s + 1] - r0;
  const int p0 = I[A.pair_off + s], nP = I[A.pair_off + s + 1] - p0;
  const int l0 = I[A.left_off + s], nL = I[A.left_off + s + 1] - l0;
  // exponentiated structural terms (0 where the log term is log 0); loads are unconditional from valid cells
  const int c_here = q.cell(i, d), c_up = (i > 0 && d + 2 <= q.W) ? q.cell(i - 1, d + 2) : c_here;
  const double l_st = xw_cell(q, kl, XT_STACK, c_here), l_ml = xw_cell(q, kl, XT_ML, c_here);
  const double l_cl = xw_cell(q, kl, XT_CLOSE, c_up), l_hp = xw_cell(q, kl, XT_HP, c_up);
  const double xst = pok ? l_st : 0., xml = pok ? l_ml : 0., xcl = eok ? l_cl : 0., xhp = eok ? l_hp : 0.;
  const int d1 = d > 0 ? d - 1 : 0, d2 = d > 1 ? d - 2 : 0, i1 = i < q.L ? i + 1 : i;
  const int pr = j > 0 ? j - 1 : 0;   // position emitted on the right
  // liveness of the operand cells: L(i,j-1) exists for d > 0; 2(i,j-1) where left_ok; E(i+1,j-1) where its closing pair (i,j)
  // is kept; P(i+1,j-1) where that inner pair is kept; M(i+1,j) where m_ok
  const bool inner_ok = d >= 2 && q.pair_ok(i + 1, d - 2);
  const bool cE = pok && d >= 2, cP = pok && inner_ok;
  double sL = 0., s2 = 0., sP = 0., sM = 0.;
#pragma unroll
  for (int u = 0; u < kUnaryR; ++u) {
    const bool vr = u < nR;
    const int s1 = vr ? I[A.right_ent + 2 * (r0 + u)] : 0, tfr = vr ? I[A.right_ent + 2 * (r0 + u) + 1] : 0;
    const double tL = T.ld(ST_L, d1, i, s1, vr && doL), t2 = T.ld(ST_2, d1, i, s1, vr && do2);
    const bool okr = vr && (!CON || allow_right(m, con, q.L, j, s, s1));
    const double wr = (okr && d > 0) ? lw_right(m, q, s, tfr, pr) : 0.;
    sL += (doL && okr) ? tL * wr : 0.;
    s2 += (do2 && okr) ? t2 * wr : 0.;
    if (u < kUnary) {
      const bool vp = u < nP;
      const int sp = vp ? I[A.pair_ent + 2 * (p0 + u)] : 0, tfp = vp ? I[A.pair_ent + 2 * (p0 + u) + 1] : 0;
      const bool vl = u < nL;
      const int sl = vl ? I[A.left_ent + 2 * (l0 + u)] : 0, tfl = vl ? I[A.left_ent + 2 * (l0 + u) + 1] : 0;
      const double tE = T.ld(ST_E, d2, i1, sp, vp && cE), tP = T.ld(ST_P, d2, i1, sp, vp && cP);
      const double tM = T.ld(ST_M, d1, i1, sl, vl && doM);
      const bool okp = vp && pok && (!CON || allow_pair(m, con, q.L, i, j, s, sp));
      const bool okl = vl && doM && (!CON || allow_left(m, con, i, s, sl));
      const double wp = okp ? lw_pair(m, q, s, sp, tfp, i, pr) : 0.;
      const double wl = okl ? lw_left(m, q, sl, tfl, i) : 0.;
      sP += okp ? wp * fma(tP, xst, tE) : 0.;
      sM += okl ? tM * wl : 0.;
    }
  }
  for (int u = kUnaryR; u < nR; ++u) {
    const int s1 = I[A.right_ent + 2 * (r0 + u)], tf = I[A.right_ent + 2 * (r0 + u) + 1];
    if (CON && !allow_right(m, con, q.L, j, s, s1)) continue;
    const double wr = lw_right(m, q, s, tf, pr);
    if (doL) sL += T.ld(ST_L, d - 1, i, s1) * wr;
    if (do2) s2 += T.ld(ST_2, d - 1, i, s1) * wr;
  }
  for (int u = kUnary; u < nP; ++u) {
    if (!pok) break;
    const int s1 = I[A.pair_ent + 2 * (p0 + u)], tf = I[A.pair_ent + 2 * (p0 + u) + 1];
    if (CON && !allow_pair(m, con, q.L, i, j, s, s1)) continue;
    sP += lw_pair(m, q, s, s1, tf, i, j - 1) * fma(T.ld(ST_P, d - 2, i + 1, s1, cP), xst, T.ld(ST_E, d - 2, i + 1, s1, cE));
  }
  for (int u = kUnary; u < nL; ++u) {
    if (!doM) break;
    const int s1 = I[A.left_ent + 2 * (l0 + u)], tf = I[A.left_ent + 2 * (l0 + u) + 1];
    if (CON && !allow_left(m, con, i, s, s1)) continue;
    sM += T.ld(ST_M, d - 1, i + 1, s1) * lw_left(m, q, s1, tf, i);
  }
  Cell7 c;
  c.vL = isloop ? (d == 0 ? ((m.st_l(s) == m.st_r(s)) ? 1. : 0.) : sL) : 0.;   // motif_trainer.hpp:89-95
  c.vP = pok ? sP : 0.;                                                          // rules 1a, 1b
  c.vB = lok ? HB : 0.;                                                          // rule 2
  c.v2 = lok ? fma(c.vP, xml, s2) : 0.;                                          // rules 3a, 3b
  c.v1 = lok ? c.v2 + c.vB : 0.;                                                 // rules 4a, 4b
  c.vM = mok ? sM + c.vB : 0.;                                                   // rules 5a, 5b
  c.vE = eok ? fma(c.vM, xcl, fma(c.vL, xhp, HE)) : 0.;                          // rules 6a, 6b, 6c
  T.st(ST_L, d, i, s, c.vL);
  T.st(ST_P, d, i, s, c.vP, pok);
  T.st(ST_B, d, i, s, c.vB, lok);
  T.st(ST_2, d, i, s, c.v2, lok);
  T.st(ST_1, d, i, s, c.v1, lok);
  T.st(ST_M, d, i, s, c.vM, mok);
  T.st(ST_E, d, i, s, c.vE, eok);
  return c;
}
template <bool CON = false>
ELEMDP_HD Cell7 lin_inside_target(const ModelView& m, const SeqView& q, const TableView& T, int d, int i, int s,
                                  const Constraint& c = Constraint{-1, -1, 0}) {
  const double HB = q.left_ok(i, d) ? lheavy_bif(m, q, T, d, i, s) : 0.;
  const double HE = q.e_ok(i, d) ? lheavy_loop(m, q, T, d, i, s) : 0.;
  return lin_inside_target_u<CON>(m, q, T, d, i, s, HB, HE, c);
}

// exterior chain, one step (rules 7, 8), j >= 1.  `part` of `nparts`: the pairs (i, j) are dealt to nparts lanes (i = j-1-part,
// j-1-part-nparts, ..); part 0 also takes rule 8.  The value of the step is the sum of the parts.
template <bool CON = false>
ELEMDP_HD double lin_inside_ext_part(const ModelView& m, const SeqView& q, const TableView& T, int j, int s, const Constraint& c,
                                    int part, int nparts) {
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int32_t* G = m.big;
  const int kl = lamk(m, s);
  double a = 0.;
  const int i0 = (j - q.W > 0) ? j - q.W : 0;
  for (int i = j - 1 - part; i >= i0; i -= nparts) {
    const int d = j - i;
    if (!q.pair_ok(i, d)) continue;
    const double xe = xw_cell(q, kl, XT_EXT, q.cell(i, d));   // (0 where the term is log 0; travels with the rows below)
    double b = 0.;
    for (int u = G[A.split_off + s]; u < G[A.split_off + s + 1]; ++u)
      b = fma(T.o(i, G[A.split_ent + 2 * u]), T.ld(ST_P, d, i, G[A.split_ent + 2 * u + 1]), b);
    a = fma(b, xe, a);
  }
  if (part == 0 && q.unp[j - 1])
    for (int t = I[A.right_off + s]; t < I[A.right_off + s + 1]; ++t) {
      if (CON && !allow_right(m, c, q.L, j, s, I[A.right_ent + 2 * t])) continue;
      a = fma(T.o(j - 1, I[A.right_ent + 2 * t]), lw_right(m, q, s, I[A.right_ent + 2 * t + 1], j - 1), a);
    }
  return a;
}
template <bool CON = false>
ELEMDP_HD void lin_inside_ext_target(const ModelView& m, const SeqView& q, const TableView& T, int j, int s,
                                     const Constraint& c = Constraint{-1, -1, 0}) {
  T.o(j, s) = lin_inside_ext_part<CON>(m, q, T, j, s, c, 0, 1);
}
ELEMDP_HD double lin_part(const ModelView& m, const TableView& T, bool ari, bool nasi) {
  return (nasi ? T.o(T.L, m.lay.s00) : 0.) + (ari ? T.o(T.L, m.lay.s0m2) + T.o(T.L, m.lay.s0m1) : 0.);
}

// ---- outside ------------------------------------------------------------------------------------
template <class Sink> struct LinOutCtx {
  const ModelView& m;
  const SeqView& q;
  const TableView& in;
  const TableView& out;
  double invZ;
  Sink& sink;
  Constraint c = Constraint{-1, -1, 0};   // OUT_END: the motif starts at c.ys
};

// Statistics of one emitting transition; z = linear posterior.  OUT_TRAIN / OUT_SCAN: expected emission counts
// (motif_trainer.hpp:384-388 / profile_hmm.hpp:144-179); OUT_SCAN: + start / inner position posteriors
// (motif_scanner.hpp:546-573); OUT_END: end posteriors given the start (:715-747) -- returns false when the start
// constraint excludes the transition (the caller then drops the term from the outside sums as well).
// Sink: en(idx, w), eh(k, w), pos(which, p, w) with which 0 = start, 1 = inner, 2 = end (linear accumulators).
template <int MODE, class Sink> ELEMDP_HD bool lstat_pair(LinOutCtx<Sink>& x, int i, int j, int par, int ch, double z) {
  if (MODE == OUT_NONE) return true;
  const ModelView& m = x.m;
  const int k = i - 1;
  if (MODE == OUT_SCAN || MODE == OUT_END) {
    const int pl = m.st_l(par), pr = m.st_r(par), cl = m.st_l(ch), cr = m.st_r(ch), M = m.lay.M;
    if (MODE == OUT_END) {
      if (x.c.ys == k && !(pl == 0 && cl == 1)) return false;
      if (x.c.ys == j && !(cr == 0 && pr == 1)) return false;
      if (z != 0.) {
        if (pl == M - 2 && cl == M - 1) x.sink.pos(2, k, z);
        if (cr == M - 2 && pr == M - 1) x.sink.pos(2, j, z);
        if (pr == M - 2 && x.q.L == j + 1) x.sink.pos(2, x.q.L, z);
      }
      return true;
    }
    if (z != 0.) {
      if (pl == 0 && cl == 1) x.sink.pos(0, k, z);
      if (cr == 0 && pr == 1) x.sink.pos(0, j, z);
      if (cl != 0 && cl != M - 1) x.sink.pos(1, k, z);
      if (pr != 0 && pr != M - 1) x.sink.pos(1, j, z);
    }
  }
  if (m.no_prf || z == 0.) return true;
  const int bi = x.q.seq[k], bj = x.q.seq[j];
  if (m.ints[m.lay.st_pair_r + par]) {
    const int t = bp_type(bi, bj);
    if (t) x.sink.en(m.param_index(m.ints[m.lay.st_row_r + par], t - 1), z);
  } else {
    if (bi) x.sink.en(m.param_index(m.ints[m.lay.st_row_l + ch], bi - 1), z);
    if (bj) x.sink.en(m.param_index(m.ints[m.lay.st_row_r + par], bj - 1), z);
  }
  return true;
}
// right emission: parent (., j+1, par), child (., j, ch): position j
template <int MODE, class Sink> ELEMDP_HD bool lstat_right(LinOutCtx<Sink>& x, int j, int par, int ch, double z) {
  if (MODE == OUT_NONE) return true;
  const ModelView& m = x.m;
  if (MODE == OUT_SCAN || MODE == OUT_END) {
    const int pr = m.st_r(par), cr = m.st_r(ch), M = m.lay.M;
    if (MODE == OUT_END) {
      if (x.c.ys == j && !(cr == 0 && pr == 1)) return false;
      if (z != 0.) {
        if (cr == M - 2 && pr == M - 1) x.sink.pos(2, j, z);
        if (pr == M - 2 && x.q.L == j + 1) x.sink.pos(2, x.q.L, z);
      }
      return true;
    }
    if (z != 0.) {
      if (cr == 0 && pr == 1) x.sink.pos(0, j, z);
      if (pr != 0 && pr != M - 1) x.sink.pos(1, j, z);
    }
  }
  if (m.no_prf || z == 0.) return true;
  const int b = x.q.seq[j];
  if (b) x.sink.en(m.param_index(m.ints[m.lay.st_row_r + par], b - 1), z);
  return true;
}
// left emission: parent M(i-1, j, par), child M(i, j, ch): position i-1
template <int MODE, class Sink> ELEMDP_HD bool lstat_left(LinOutCtx<Sink>& x, int i, int par, int ch, double z) {
  if (MODE == OUT_NONE) return true;
  const ModelView& m = x.m;
  const int k = i - 1;
  if (MODE == OUT_SCAN || MODE == OUT_END) {
    const int pl = m.st_l(par), cl = m.st_l(ch), M = m.lay.M;
    if (MODE == OUT_END) {
      if (x.c.ys == k && !(pl == 0 && cl == 1)) return false;
      if (z != 0. && pl == M - 2 && cl == M - 1) x.sink.pos(2, k, z);
      return true;
    }
    if (z != 0.) {
      if (pl == 0 && cl == 1) x.sink.pos(0, k, z);
      if (cl != 0 && cl != M - 1) x.sink.pos(1, k, z);
    }
  }
  if (m.no_prf || z == 0.) return true;
  const int b = x.q.seq[k];
  if (b) x.sink.en(m.param_index(m.ints[m.lay.st_row_l + ch], b - 1), z);
  return true;
}
// EH[idx(parent)] += tsc * posterior (motif_trainer.hpp:380-381); tsc may be log 0 where the posterior is 0
template <int MODE, class Sink> ELEMDP_HD void lstat_energy(LinOutCtx<Sink>& x, int par, double tsc, double z) {
  if (MODE == OUT_TRAIN && z != 0.) x.sink.eh(x.m.eh_index(par), tsc * z);
}

// exterior chain backwards, one step (rules 8, 7 reversed) with the statistics of those transitions; `part` of `nparts`
// as in lin_inside_ext_part (the parents (i, j) dealt by j; part 0 also takes rule 8)
template <int MODE, class Sink> ELEMDP_HD double lin_outside_ext_part(LinOutCtx<Sink>& x, int i, int s, int part, int nparts) {
  const ModelView& m = x.m;
  const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int32_t* G = m.big;
  const double in_c = x.in.o(i, s);
  if (in_c == 0.) return 0.;
  const double inz = in_c * x.invZ;
  double a = 0.;
  if (part == 0 && q.unp[i])
    for (int t = I[A.rright_off + s]; t < I[A.rright_off + s + 1]; ++t) {
      const int par = I[A.rright_ent + 2 * t], tf = I[A.rright_ent + 2 * t + 1];
      const double term = x.out.o(i + 1, par) * lw_right(m, q, par, tf, i);
      if (!lstat_right<MODE>(x, i, par, s, term * inz)) continue;
      a += term;
    }
  const int jmax = (i + q.W < q.L) ? i + q.W : q.L;
  for (int j = i + 1 + part; j <= jmax; j += nparts) {
    const int d = j - i;
    if (!q.pair_ok(i, d)) continue;
    const int c = q.cell(i, d);
    const double t = q.e_ext[c];      // (log 0: both weights below are 0, the terms vanish and take no statistic)
    const double x0 = xw_cell(q, 0, XT_EXT, c), x1 = xw_cell(q, 1, XT_EXT, c);
    for (int u = G[A.split1_off + s]; u < G[A.split1_off + s + 1]; ++u) {
      const int par = G[A.split1_ent + 2 * u], s1 = G[A.split1_ent + 2 * u + 1];
      const double term = x.out.o(j, par) * (x.in.ld(ST_P, d, i, s1) * (lamk(m, par) ? x1 : x0));
      lstat_energy<MODE>(x, par, t, term * inz);
      a += term;
    }
  }
  return a;
}
// The two halves of lin_outside_ext_part for the blocked chain (k4_out_ext with 512 threads): the pair terms of a step read chain
// rows at least five steps back, so four steps' worth is summed side by side; rule 8 (the row of the previous step) stays
// sequential.  (The value of a step is then rule 8 + the parts' pair sums instead of (rule 8 + part 0's pairs) + the other parts:
// the same terms, associated differently.)
template <int MODE, class Sink> ELEMDP_HD double lin_outside_ext_rule8(LinOutCtx<Sink>& x, int i, int s) {
  const ModelView& m = x.m;
  const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const double in_c = x.in.o(i, s);
  if (in_c == 0.) return 0.;
  const double inz = in_c * x.invZ;
  double a = 0.;
  if (q.unp[i])
    for (int t = I[A.rright_off + s]; t < I[A.rright_off + s + 1]; ++t) {
      const int par = I[A.rright_ent + 2 * t], tf = I[A.rright_ent + 2 * t + 1];
      const double term = x.out.o(i + 1, par) * lw_right(m, q, par, tf, i);
      if (!lstat_right<MODE>(x, i, par, s, term * inz)) continue;
      a += term;
    }
  return a;
}
template <int MODE, class Sink> ELEMDP_HD double lin_outside_ext_pairs(LinOutCtx<Sink>& x, int i, int s, int part, int nparts) {
  const ModelView& m = x.m;
  const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.big;
  const double in_c = x.in.o(i, s);
  if (in_c == 0.) return 0.;
  const double inz = in_c * x.invZ;
  double a = 0.;
  const int jmax = (i + q.W < q.L) ? i + q.W : q.L;
  for (int j = i + 1 + part; j <= jmax; j += nparts) {
    const int d = j - i;
    if (!q.pair_ok(i, d)) continue;
    const int c = q.cell(i, d);
    const double t = q.e_ext[c];
    const double x0 = xw_cell(q, 0, XT_EXT, c), x1 = xw_cell(q, 1, XT_EXT, c);
    for (int u = G[A.split1_off + s]; u < G[A.split1_off + s + 1]; ++u) {
      const int par = G[A.split1_ent + 2 * u], s1 = G[A.split1_ent + 2 * u + 1];
      const double term = x.out.o(j, par) * (x.in.ld(ST_P, d, i, s1) * (lamk(m, par) ? x1 : x0));
      lstat_energy<MODE>(x, par, t, term * inz);
      a += term;
    }
  }
  return a;
}
template <int MODE, class Sink> ELEMDP_HD void lin_outside_ext_target(LinOutCtx<Sink>& x, int i, int s) {
  x.out.o(i, s) = lin_outside_ext_part<MODE>(x, i, s, 0, 1);
}

// ---- rule 2 in the outside direction, factorised (see lin_inside_apair) ----------------------------
// With outA(i,l,(s1,t)) := d Z / d A(i,l,(s1,t)):
//     outA(i,j,p)   = outB(i,j,tgt(p)) + [unp(j)] sum_{p'' : p in chain(p'')} outA(i,j+1,p'') w_right(t'',t,j)     (+ statistics)
//     out1(i,k,s1)  = sum_{stems (k,l)} sum_{p=(s1,t)} outA(i,l,p) P(k,l,t) xml(k,l)                                  ("H1")
//     out2(k,l,t)   = [direct part: rules 4a, 3a over the plane-2 table] + HA(k,l,t),
//     HA(k,l,t)     = sum_{i<k} sum_{p=(s1,t)} outA(i,l,p) 1(i,k,s1)       -- needed only where a stem P(k,l) takes it (rule 3b)
// The plane-2 OUTSIDE table therefore holds the direct part only (debug_tables adds HA for the export).
// H1: cell (i,d) in the role 1(i,k,s), k = i + d
template <class Sink> ELEMDP_HD double lheavy_o1(LinOutCtx<Sink>& x, int d, int i, int s) {
  const ModelView& m = x.m; const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay; const int32_t* I = m.ints;
  const int k = i + d;
  const int hi = (q.W - d < q.L - k) ? q.W - d : q.L - k;   // stems (k, k+sp): the parent B(i, k+sp) stays in the band
  double a = 0.;
  for_mask_bits(q.okbits, k * (q.W + 1), 1, hi, [&](int sp) {
    const int c = q.cell(k, sp);
    for (int e = I[A.ap_by_s1_off + s]; e < I[A.ap_by_s1_off + s + 1]; ++e) {
      const int p = I[A.ap_by_s1_ent + e], t = I[A.ap_t + p];
      a = fma(x.out.a(d + sp, i, p), x.in.ld(ST_P, sp, k, t) * xw_cell(q, lamk(m, t), XT_ML, c), a);
    }
  });
  return a;
}
// HA: cell (i,d) in the role of the stem cell (k,l) = (i, i+d), state t of plane 2 / P
template <class Sink> ELEMDP_HD double lheavy_o2(LinOutCtx<Sink>& x, int d, int i, int t) {
  const ModelView& m = x.m; const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay; const int32_t* I = m.ints;
  const int j = i + d;
  const int imin = (j - q.W > 0) ? j - q.W : 0;
  double a = 0.;
  for (int ii = i - 1; ii >= imin; --ii) {
    if (!o2_valid(q, i, ii)) continue;          // 1(ii, i, .) is log 0
    for (int e = I[A.ap_by_t_off + t]; e < I[A.ap_by_t_off + t + 1]; ++e) {
      const int p = I[A.ap_by_t_ent + e];
      a = fma(x.out.a(j - ii, ii, p), x.in.ld(ST_1, i - ii, ii, I[A.ap_s1 + p]), a);
    }
  }
  return a;
}
// outside value of the pair entry (i,d,p), after the unary phase of the cell wrote out B; statistics of the tail emissions
// (oB_tgt = the outside value of B(i,d,tgt(p)) just computed by the unary phase; ignored when the pair has no target)
template <int MODE, class Sink> ELEMDP_HD void lin_outside_apair(LinOutCtx<Sink>& x, int d, int i, int p, double oB_tgt) {
  const ModelView& m = x.m; const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay; const int32_t* I = m.ints;
  const int j = i + d;
  const int t = I[A.ap_t + p], tgt = I[A.ap_tgt + p];
  const int dmi = q.dmin[i];
  if (!(dmi > 0 && dmi < d)) return;        // the entry does not exist (lin_inside_apair)
  const bool step = d + 1 <= q.W && j < q.L && q.unp[j];
  const int e0 = I[A.ap_rchain_off + p], ne = step ? I[A.ap_rchain_off + p + 1] - e0 : 0;
  // (the parents' values are fetched together with the inside value: one round trip; lists longer than kUnary follow)
  const double a_in = x.in.a(d, i, p);
  double op[kUnary];
#pragma unroll
  for (int u = 0; u < kUnary; ++u) op[u] = x.out.lda(d + 1, i, u < ne ? I[A.ap_rchain_ent + 2 * (e0 + u)] : p, step);
  double a = (tgt >= 0) ? oB_tgt : 0.;
  if (a_in != 0.) {
    const double inz = a_in * x.invZ;
    for (int u = 0; u < ne; ++u) {
      const int pp = I[A.ap_rchain_ent + 2 * (e0 + u)], tf = I[A.ap_rchain_ent + 2 * (e0 + u) + 1];
      const int par = I[A.ap_t + pp];
      double o = 0.;
      if (u < kUnary) {
#pragma unroll
        for (int k = 0; k < kUnary; ++k) o = (k == u) ? op[k] : o;     // (a select chain: no dynamic register indexing)
      } else {
        o = x.out.a(d + 1, i, pp);
      }
      const double term = o * lw_right(m, q, par, tf, j);
      if (!lstat_right<MODE>(x, j, par, t, term * inz)) continue;
      a += term;
    }
  } else {
    a = 0.;
  }
  x.out.a(d, i, p) = a;
}
template <int MODE, class Sink> ELEMDP_HD void lin_outside_cell_pairs(LinOutCtx<Sink>& x, int d, int i) {
  for (int p = 0; p < x.m.lay.n_ap; ++p) {
    const int tgt = x.m.ints[x.m.lay.ap_tgt + p];
    lin_outside_apair<MODE>(x, d, i, p, tgt >= 0 ? x.out.ld(ST_B, d, i, tgt, x.q.left_ok(i, d)) : 0.);
  }
}
// HP and the energy statistic of rule 6c: the posterior of (item, tuple) is term * inside P(i,j,s) / Z
template <int MODE, class Sink> ELEMDP_HD double lheavy_oP(LinOutCtx<Sink>& x, int d, int i, int s) {
  const ModelView& m = x.m; const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay; const int32_t* G = m.big;
  const int j = i + d;
  const double in_c = x.in.ld(ST_P, d, i, s);     // (the caller has checked pair_ok(i, d))
  if (in_c == 0.) return 0.;
  const double inz = in_c * x.invZ;
  double a = 0.;
  const int pc = q.cell(i, d);
  for (int n = q.by_inner_off[pc]; n < q.by_inner_off[pc + 1]; ++n) {
    const int idx = q.by_inner_idx[n];
    const LoopItem it = q.items[idx];
    const double x0 = xw_item(q, 0, idx), x1 = xw_item(q, 1, idx);
    for (int u = G[A.quad1_off + s]; u < G[A.quad1_off + s + 1]; ++u) {
      const int par = G[A.quad1_ent + 3 * u];
      const double term = x.out.ld(ST_E, it.j - it.i, it.i, par) *
                          (x.in.ld(ST_L, i - it.i, it.i, G[A.quad1_ent + 3 * u + 1]) *
                           (x.in.ld(ST_L, it.j - j, j, G[A.quad1_ent + 3 * u + 2]) * (lamk(m, par) ? x1 : x0)));
      lstat_energy<MODE>(x, par, it.tsc, term * inz);
      a += term;
    }
  }
  return a;
}
template <class Sink> ELEMDP_HD double lheavy_oL(LinOutCtx<Sink>& x, int d, int i, int s) {
  const ModelView& m = x.m; const SeqView& q = x.q;
  const AutomatonLayout& A = m.lay; const int32_t* G = m.big;
  double a = 0.;
  const int lc = q.cell(i, d);
  for (int n = q.by_left_off[lc]; n < q.by_left_off[lc + 1]; ++n) {
    const int idx = q.by_left_idx[n];
    const LoopItem it = q.items[idx];
    const double x0 = xw_item(q, 0, idx), x1 = xw_item(q, 1, idx);
    for (int u = G[A.quad2_off + s]; u < G[A.quad2_off + s + 1]; ++u) {
      const int par = G[A.quad2_ent + 3 * u];
      a = fma(x.out.ld(ST_E, it.j - it.i, it.i, par),
              x.in.ld(ST_P, it.l - it.k, it.k, G[A.quad2_ent + 3 * u + 1]) *
                  (x.in.ld(ST_L, it.j - it.l, it.l, G[A.quad2_ent + 3 * u + 2]) * (lamk(m, par) ? x1 : x0)), a);
    }
  }
  for (int n = q.by_right_off[lc]; n < q.by_right_off[lc + 1]; ++n) {
    const int idx = q.by_right_idx[n];
    const LoopItem it = q.items[idx];
    const double x0 = xw_item(q, 0, idx), x1 = xw_item(q, 1, idx);
    for (int u = G[A.quad3_off + s]; u < G[A.quad3_off + s + 1]; ++u) {
      const int par = G[A.quad3_ent + 3 * u];
      a = fma(x.out.ld(ST_E, it.j - it.i, it.i, par),
              x.in.ld(ST_P, it.l - it.k, it.k, G[A.quad3_ent + 3 * u + 1]) *
                  (x.in.ld(ST_L, it.k - it.i, it.i, G[A.quad3_ent + 3 * u + 2]) * (lamk(m, par) ? x1 : x0)), a);
    }
  }
  return a;
}

// band target (i,d,s), outside direction, given the heavy sums (H1 -> state 1, H2 -> 2, HP -> P, HL -> L)
template <int MODE, class Sink>
ELEMDP_HD double lin_outside_target_u(LinOutCtx<Sink>& x, int d, int i, int s, const HeavyOut& H) {   // returns out B(i,d,s)
  const ModelView& m = x.m;
  const SeqView& q = x.q;
  const TableView& in = x.in;
  const TableView& out = x.out;
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int32_t* G = m.big;
  const double invZ = x.invZ;
  const int j = i + d;
  const int kl = lamk(m, s);
  const bool isloop = I[A.st_is_loop + s] != 0;
  const bool pok = q.pair_ok(i, d);
  const bool lok = q.left_ok(i, d);
  const bool mok = m_ok(m, q, i, d);
  const bool eok = q.e_ok(i, d);
  const bool up_ok = q.pair_ok(i - 1, d + 2);
  const bool doM = mok && m_ok(m, q, i - 1, d + 1) && q.unp[i > 0 ? i - 1 : 0];
  const bool do2 = lok && q.left_ok(i, d + 1) && q.unp[j];
  const bool doL = isloop && j < q.L && d + 1 <= q.W;

  const double inE = in.ld(ST_E, d, i, s, eok), inM = in.ld(ST_M, d, i, s, mok), in1 = in.ld(ST_1, d, i, s, lok);
  const double inB = in.ld(ST_B, d, i, s, lok), in2 = in.ld(ST_2, d, i, s, lok), inP = in.ld(ST_P, d, i, s, pok);
  const double inL = in.ld(ST_L, d, i, s, isloop);
  const int c_here = q.cell(i, d), c_up = (i > 0 && d + 2 <= q.W && i + d < q.L) ? q.cell(i - 1, d + 2) : c_here;
  // raw terms (for the energy statistic) and their exponentials for both lambda classes
  const double e_cl = q.e_close[c_up], e_hp = q.e_hp[c_up], e_su = q.e_stack[c_up], e_ml = q.e_ml[c_here];
  const double l_cl = xw_cell(q, kl, XT_CLOSE, c_up), l_hp = xw_cell(q, kl, XT_HP, c_up), l_ml = xw_cell(q, kl, XT_ML, c_here);
  const double l_su0 = xw_cell(q, 0, XT_STACK, c_up), l_su1 = xw_cell(q, 1, XT_STACK, c_up);
  const double l_ex0 = xw_cell(q, 0, XT_EXT, c_here), l_ex1 = xw_cell(q, 1, XT_EXT, c_here);
  const double xcl = eok ? l_cl : 0., xhp = eok ? l_hp : 0., xml = pok ? l_ml : 0.;
  const bool su_ok = up_ok && pok;
  const double xsu0 = su_ok ? l_su0 : 0., xsu1 = su_ok ? l_su1 : 0.;
  const double xex0 = pok ? l_ex0 : 0., xex1 = pok ? l_ex1 : 0.;

  const int rp0 = I[A.rpair_off + s], nRP = I[A.rpair_off + s + 1] - rp0;
  const int rl0 = I[A.rleft_off + s], nRL = I[A.rleft_off + s + 1] - rl0;
  const int rr0 = I[A.rright_off + s], nRR = I[A.rright_off + s + 1] - rr0;
  const int dp1 = d + 1 <= q.W ? d + 1 : q.W, dp2 = d + 2 <= q.W ? d + 2 : q.W, im1 = i > 0 ? i - 1 : 0;
  const int jr = j < q.L ? j : (q.L > 0 ? q.L - 1 : 0);   // clamped position for the weight of a masked term

  // E as child of P(i-1,j+1,par) (rule 1a) and P as child of P(i-1,j+1,par) (rule 1b) share parents and weights;
  // so do 2 and L as children of 2 / L (i,j+1,par).  Parent values of the first kUnary list entries are fetched
  // unconditionally (clamped cells) so that the loads of one target are in flight together.
  const double inEz = inE * invZ, inPz = inP * invZ, inMz = inM * invZ, in2z = in2 * invZ, inLz = inL * invZ;
  const bool aE = up_ok && inE != 0., aP = su_ok && inP != 0., aM = doM && inM != 0., a2 = do2 && in2 != 0.,
             aL = doL && inL != 0.;
  double oE = 0., oP1b = 0., sM = 0., s2 = 0., sL = 0.;
  auto step_pair = [&](int par, int tf, double op) {
    const double w = lw_pair(m, q, par, s, tf, im1, jr);
    const double tE = aE ? op * w : 0.;
    const double tP = aP ? op * (w * (lamk(m, par) ? xsu1 : xsu0)) : 0.;
    if (!lstat_pair<MODE>(x, i, j, par, s, fma(tE, inEz, tP * inPz))) return;
    lstat_energy<MODE>(x, par, e_su, tP * inPz);
    oE += tE;
    oP1b += tP;
  };
  auto step_left = [&](int par, int tf, double op) {
    const double term = op * lw_left(m, q, s, tf, im1);
    if (!lstat_left<MODE>(x, i, par, s, term * inMz)) return;
    sM += term;
  };
  auto step_right = [&](int par, int tf, double op2, double opL) {
    const double w = lw_right(m, q, par, tf, jr);
    const double t2 = a2 ? op2 * w : 0.;
    const double tL = (aL && I[A.st_is_loop + par]) ? opL * w : 0.;
    if (!lstat_right<MODE>(x, j, par, s, fma(t2, in2z, tL * inLz))) return;
    s2 += t2;
    sL += tL;
  };
#pragma unroll
  for (int u = 0; u < kUnaryR; ++u) {
    const bool vr = u < nRR;
    const int par_r = vr ? I[A.rright_ent + 2 * (rr0 + u)] : 0, tf_r = vr ? I[A.rright_ent + 2 * (rr0 + u) + 1] : 0;
    const double op2 = out.ld(ST_2, dp1, i, par_r, vr && do2), opL = out.ld(ST_L, dp1, i, par_r, vr && doL);
    if (u < kUnary) {
      const bool vp = u < nRP, vl = u < nRL;
      const int par_p = vp ? I[A.rpair_ent + 2 * (rp0 + u)] : 0, tf_p = vp ? I[A.rpair_ent + 2 * (rp0 + u) + 1] : 0;
      const int par_l = vl ? I[A.rleft_ent + 2 * (rl0 + u)] : 0, tf_l = vl ? I[A.rleft_ent + 2 * (rl0 + u) + 1] : 0;
      const double opP = out.ld(ST_P, dp2, im1, par_p, vp && up_ok), opM = out.ld(ST_M, dp1, im1, par_l, vl && doM);
      if (vp && (aE || aP)) step_pair(par_p, tf_p, opP);
      if (vl && aM) step_left(par_l, tf_l, opM);
    }
    if (vr && (a2 || aL)) step_right(par_r, tf_r, op2, opL);
  }
  if (aE || aP)
    for (int u = kUnary; u < nRP; ++u) {
      const int par = I[A.rpair_ent + 2 * (rp0 + u)];
      step_pair(par, I[A.rpair_ent + 2 * (rp0 + u) + 1], out.ld(ST_P, d + 2, i - 1, par));
    }
  if (aM)
    for (int u = kUnary; u < nRL; ++u) {
      const int par = I[A.rleft_ent + 2 * (rl0 + u)];
      step_left(par, I[A.rleft_ent + 2 * (rl0 + u) + 1], out.ld(ST_M, d + 1, i - 1, par));
    }
  if (a2 || aL)
    for (int u = kUnaryR; u < nRR; ++u) {
      const int par = I[A.rright_ent + 2 * (rr0 + u)];
      step_right(par, I[A.rright_ent + 2 * (rr0 + u) + 1], out.ld(ST_2, d + 1, i, par, a2), out.ld(ST_L, d + 1, i, par, aL));
    }
  out.st(ST_E, d, i, s, oE, eok);

  // M: child of E (6a) and of M(i-1,j,par) (5a)
  double oM = 0.;
  if (inM != 0.) {
    const double t6a = oE * xcl;
    lstat_energy<MODE>(x, s, e_cl, t6a * inMz);
    oM = t6a + sM;
  }
  out.st(ST_M, d, i, s, oM, mok);

  // 1: heavy sum H1 ; B: child of M (5b) and 1 (4b) ; 2: child of 1 (4a), 2(i,j+1,par) (3a), heavy sum H2
  const double o1 = (in1 != 0.) ? H.H1 : 0.;
  const double oB = (inB != 0.) ? (mok ? oM : 0.) + o1 : 0.;
  const double o2 = (in2 != 0.) ? o1 + s2 : 0.;   // direct part (rules 4a, 3a); the rule-2 part reaches P as H.H2 = HA
  out.st(ST_1, d, i, s, o1, lok);
  out.st(ST_B, d, i, s, oB, lok);
  out.st(ST_2, d, i, s, o2, lok);

  // P: child of O (7), of P(i-1,j+1,par) (1b), of 2 (3b), inner pair of interior loops (6c: HP)
  double oP = 0.;
  if (inP != 0.) {
    double a = 0.;
    if ((xex0 != 0. || xex1 != 0.) && !H.ext_in_hp)
      for (int u = G[A.split2_off + s]; u < G[A.split2_off + s + 1]; ++u) {
        const int par = G[A.split2_ent + 2 * u], s2i = G[A.split2_ent + 2 * u + 1];
        a = fma(out.o(j, par), in.o(i, s2i) * (lamk(m, par) ? xex1 : xex0), a);
      }
    const double t3b = (o2 + H.H2) * xml;
    lstat_energy<MODE>(x, s, e_ml, t3b * inPz);
    oP = a + oP1b + t3b + H.HP;
  }
  out.st(ST_P, d, i, s, oP, pok);

  // L: child of E (6b), of L(i,j+1,par), loops of interior loops (6c: HL)
  double oL = 0.;
  if (inL != 0.) {
    const double t6b = oE * xhp;
    lstat_energy<MODE>(x, s, e_hp, t6b * inLz);
    oL = t6b + sL + H.HL;
  }
  out.st(ST_L, d, i, s, oL);
  return oB;
}

template <int MODE, class Sink> ELEMDP_HD void lin_outside_target(LinOutCtx<Sink>& x, int d, int i, int s) {
  const SeqView& q = x.q;
  HeavyOut H;
  const bool lok = q.left_ok(i, d);
  const bool pok = q.pair_ok(i, d);
  H.H1 = (lok && x.in.ld(ST_1, d, i, s) != 0.) ? lheavy_o1(x, d, i, s) : 0.;
  H.H2 = (pok && x.in.ld(ST_P, d, i, s) != 0.) ? lheavy_o2(x, d, i, s) : 0.;
  H.HP = pok ? lheavy_oP<MODE>(x, d, i, s) : 0.;
  H.HL = (x.m.ints[x.m.lay.st_is_loop + s] && x.in.ld(ST_L, d, i, s) != 0.) ? lheavy_oL(x, d, i, s) : 0.;
  lin_outside_target_u<MODE>(x, d, i, s, H);
}

}  // namespace elemdp
