// scan_rules.h -- CYK (max-product) targets with traceback records, and the traceback itself.
//
// Per-target form of RNAelemScanDP::CYKFun (RNAelem/motif_scanner.hpp:802-913) and trace_back
// (:262-362).  The candidate order inside one target follows the reference's sweep order
// (energy_model.hpp:346-437 x motif_model.hpp:262-421) because `compare` keeps the first strictly
// greatest candidate (:821): ties -- frequent with uniform theta -- must resolve identically.
#pragma once
#include "dp_rules.h"

namespace elemdp {

// products that enter a candidate are rounded before the sum, as in the reference's CPU code (no fma contraction):
// equal candidates must compare equal whatever form the sum has
#if defined(__HIP_DEVICE_COMPILE__)
#define ELEMDP_MUL_RN(a, b) __dmul_rn((a), (b))
#else
#define ELEMDP_MUL_RN(a, b) ((a) * (b))
#endif

// Trace records.  Only the exterior chain keeps them ([j][s], one row per position).  The band targets -- seven planes x S
// states per cell, as much memory as the table itself -- do not: trace_back re-derives the record of each target it visits
// (a few hundred per sequence) from the finished table with the sweep's own candidate order (cyk_retrace below).
struct TraceView {
  TraceRec* ext;
  TraceRec* one = nullptr;   // retrace: the seven records of the one target being re-derived (no table stores)
};

struct MaxAcc {
  double best;
  TraceRec tr;
  ELEMDP_HD MaxAcc() : best(ELEMDP_NEG_INF) { tr.k = tr.l = -1; tr.t = -1; tr.e1 = -1; tr.s1 = -1; }
  ELEMDP_HD void offer(double y, int k, int l, int t, int e1, int s1) {
    if (best < y) {
      best = y;
      tr.k = (int16_t)k; tr.l = (int16_t)l; tr.t = (int8_t)t; tr.e1 = (int8_t)e1; tr.s1 = (int16_t)s1;
    }
  }
};

// last index of the maximum (max_index, util.hpp:232-241)
ELEMDP_HD int last_argmax(const double* v, int n) {
  int s = 0;
  double m = -1.7976931348623157e308;
  for (int i = 0; i < n; ++i)
    if (m <= v[i]) { s = i; m = v[i]; }
  return s;
}

// The two candidate lists of a target that grow with the span (the "heavy" part): kept apart so that the batch kernel
// (k5_cyk in lin_kernels.hip) can evaluate them with lanes over (split point | item, state tuple) and hand the winner to
// cyk_target_u.  Candidate order = reference order: rule 2 by split point, then by tuple of s; rule 6c by item (by_outer
// order), then by tuple of s.
ELEMDP_HD MaxAcc cyk_split_best(const ModelView& m, const SeqView& q, const TableView& T, int d, int i, int s) {
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.big;
  const int j = i + d;
  MaxAcc aB;
  for (int k = i + q.dmin[i]; k < j; ++k) {
    const int dk = q.dmin[k];
    if (dk == 0 || j - k < dk) continue;
    for (int t = G[A.split_off + s]; t < G[A.split_off + s + 1]; ++t) {
      const int s1 = G[A.split_ent + 2 * t], s2 = G[A.split_ent + 2 * t + 1];
      aB.offer(T.ldm(ST_1, k - i, i, s1) + T.ldm(ST_2, j - k, k, s2), i, k, TT_B_12, ST_1, s1);
    }
  }
  return aB;
}
ELEMDP_HD MaxAcc cyk_item_best(const ModelView& m, const SeqView& q, const TableView& T, int d, int i, int s) {
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.big;
  const int j = i + d;
  const double lam = m.lam(s);
  MaxAcc aE;
  const int c0 = q.by_outer_off[q.cell(i, d)], c1 = q.by_outer_off[q.cell(i, d) + 1];
  for (int it = c0; it < c1; ++it) {
    if (!q.item_in[it]) continue;
    const LoopItem x = q.items[it];
    const double lt = ELEMDP_MUL_RN(lam, x.tsc);
    for (int t = G[A.quad_off + s]; t < G[A.quad_off + s + 1]; ++t) {
      const int s1 = G[A.quad_ent + 3 * t], s2 = G[A.quad_ent + 3 * t + 1], s3 = G[A.quad_ent + 3 * t + 2];
      aE.offer(T.ldm(ST_P, x.l - x.k, x.k, s1) + (T.ldm(ST_L, x.k - i, i, s2) + (T.ldm(ST_L, j - x.l, x.l, s3) + lt)), x.k,
               x.l, TT_E_P, ST_P, s1);
    }
  }
  return aE;
}

ELEMDP_HD void cyk_put(const TableView& T, const TraceView& R, int e, int d, int i, int s, const MaxAcc& a) {
  if (R.one) R.one[e] = a.tr;
  else T.stm(e, d, i, s, a.best);
}

// everything else of the target; hB = winner of rule 2 (used when left_ok), hE = winner of rule 6c (used when e_ok).
// only >= 0 (the traceback's re-derivation): just that plane -- the values of the target's other planes it builds on are read
// from the finished table instead of being computed again (they are what the sweep stored).
ELEMDP_HD void cyk_target_u(const ModelView& m, const SeqView& q, const TableView& T, const TraceView& R,
                            const Constraint& c, int d, int i, int s, const MaxAcc& hB, const MaxAcc& hE, int only = -1) {
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int j = i + d;
  const double NEG = ELEMDP_NEG_INF;
  const double lam = m.lam(s);
  const bool isloop = I[A.st_is_loop + s] != 0;
  const bool all = only < 0;
  auto want = [&](int e) { return all || only == e; };

  MaxAcc aL;
  if (want(ST_L)) {
    if (isloop) {
      if (d == 0) { if (m.st_l(s) == m.st_r(s)) aL.best = 0.; }
      else
        for (int t = I[A.right_off + s]; t < I[A.right_off + s + 1]; ++t) {
          const int s1 = I[A.right_ent + 2 * t], tf = I[A.right_ent + 2 * t + 1];
          if (!allow_right(m, c, q.L, j, s, s1)) continue;
          aL.offer(T.ldm(ST_L, d - 1, i, s1) + w_right(m, q, s, tf, j - 1), i, j - 1, TT_L_L, ST_L, s1);
        }
    }
    cyk_put(T, R, ST_L, d, i, s, aL);
  }

  const bool pok = q.pair_ok(i, d);
  MaxAcc aP;
  if (want(ST_P)) {
    if (pok) {
      const double est = q.e_stack[q.cell(i, d)];
      for (int t = I[A.pair_off + s]; t < I[A.pair_off + s + 1]; ++t) {  // all 1a candidates first ...
        const int s1 = I[A.pair_ent + 2 * t], tf = I[A.pair_ent + 2 * t + 1];
        if (!allow_pair(m, c, q.L, i, j, s, s1)) continue;
        aP.offer(T.ldm(ST_E, d - 2, i + 1, s1) + w_pair(m, q, s, s1, tf, i, j - 1), i + 1, j - 1, TT_P_E, ST_E, s1);
      }
      if (est != NEG)
        for (int t = I[A.pair_off + s]; t < I[A.pair_off + s + 1]; ++t) {  // ... then 1b
          const int s1 = I[A.pair_ent + 2 * t], tf = I[A.pair_ent + 2 * t + 1];
          if (!allow_pair(m, c, q.L, i, j, s, s1)) continue;
          aP.offer(T.ldm(ST_P, d - 2, i + 1, s1) + (w_pair(m, q, s, s1, tf, i, j - 1) + ELEMDP_MUL_RN(lam, est)), i + 1, j - 1, TT_P_P,
                   ST_P, s1);
        }
    }
    cyk_put(T, R, ST_P, d, i, s, aP);
  }

  const bool lok = q.left_ok(i, d);
  MaxAcc aB;
  if (lok) aB = hB;
  if (want(ST_B)) cyk_put(T, R, ST_B, d, i, s, aB);

  MaxAcc a2, a1;
  if (all || only == ST_2 || only == ST_1) {
    if (lok) {
      if (want(ST_2)) {
        if (q.left_ok(i, d - 1) && q.unp[j - 1])
          for (int t = I[A.right_off + s]; t < I[A.right_off + s + 1]; ++t) {
            const int s1 = I[A.right_ent + 2 * t], tf = I[A.right_ent + 2 * t + 1];
            if (!allow_right(m, c, q.L, j, s, s1)) continue;
            a2.offer(T.ldm(ST_2, d - 1, i, s1) + w_right(m, q, s, tf, j - 1), i, j - 1, TT_2_2, ST_2, s1);
          }
        if (pok) {
          const double eml = q.e_ml[q.cell(i, d)];
          if (eml != NEG) a2.offer((all ? aP.best : T.ldm(ST_P, d, i, s)) + ELEMDP_MUL_RN(lam, eml), i, j, TT_2_P, ST_P, s);
        }
      }
      if (want(ST_1)) {
        a1.offer(all ? a2.best : T.ldm(ST_2, d, i, s), i, j, TT_1_2, ST_2, s);
        a1.offer(aB.best, i, j, TT_1_B, ST_B, s);
      }
    }
    if (want(ST_2)) cyk_put(T, R, ST_2, d, i, s, a2);
    if (want(ST_1)) cyk_put(T, R, ST_1, d, i, s, a1);
  }

  const bool mok = m_ok(m, q, i, d);
  MaxAcc aM;
  if (want(ST_M)) {
    if (mok) {
      if (m_ok(m, q, i + 1, d - 1) && q.unp[i])
        for (int t = I[A.left_off + s]; t < I[A.left_off + s + 1]; ++t) {
          const int s1 = I[A.left_ent + 2 * t], tf = I[A.left_ent + 2 * t + 1];
          if (!allow_left(m, c, i, s, s1)) continue;
          aM.offer(T.ldm(ST_M, d - 1, i + 1, s1) + w_left(m, q, s1, tf, i), i + 1, j, TT_M_M, ST_M, s1);
        }
      if (lok) aM.offer(aB.best, i, j, TT_M_B, ST_B, s);
    }
    cyk_put(T, R, ST_M, d, i, s, aM);
  }

  if (want(ST_E)) {
    MaxAcc aE;
    if (q.e_ok(i, d)) {
      const int pc = q.cell(i - 1, d + 2);
      if (mok) { const double t = q.e_close[pc]; if (t != NEG) aE.offer((all ? aM.best : T.ldm(ST_M, d, i, s)) + ELEMDP_MUL_RN(lam, t), i, j, TT_E_M, ST_M, s); }
      if (isloop) { const double t = q.e_hp[pc]; if (t != NEG) aE.offer((all ? aL.best : T.ldm(ST_L, d, i, s)) + ELEMDP_MUL_RN(lam, t), i, j, TT_E_H, ST_L, s); }
      if (aE.best < hE.best) aE = hE;   // (the item candidates come last; the first strictly greatest one wins)
    }
    cyk_put(T, R, ST_E, d, i, s, aE);
  }
}

ELEMDP_HD void cyk_target(const ModelView& m, const SeqView& q, const TableView& T, const TraceView& R,
                          const Constraint& c, int d, int i, int s) {
  const MaxAcc hB = q.left_ok(i, d) ? cyk_split_best(m, q, T, d, i, s) : MaxAcc();
  const MaxAcc hE = q.e_ok(i, d) ? cyk_item_best(m, q, T, d, i, s) : MaxAcc();
  cyk_target_u(m, q, T, R, c, d, i, s, hB, hE);
}

// The record of band target (e, d, i, s) from the finished table: the candidates of the sweep in the sweep's order, hence the
// same winner.  Only a B or an E target walks its span-long list again (rule 2, rule 6c); the other planes take B's value from
// the table.
ELEMDP_HD TraceRec cyk_retrace(const ModelView& m, const SeqView& q, const TableView& T, const Constraint& c, int e, int d, int i,
                               int s) {
  MaxAcc hB, hE;
  if (q.left_ok(i, d)) {
    if (e == ST_B) hB = cyk_split_best(m, q, T, d, i, s);
    else hB.best = T.ldm(ST_B, d, i, s);
  }
  if (e == ST_E && q.e_ok(i, d)) hE = cyk_item_best(m, q, T, d, i, s);
  TraceRec rec[kNumBandStates];
  TraceView R1;
  R1.ext = nullptr;
  R1.one = rec;
  cyk_target_u(m, q, T, R1, c, d, i, s, hB, hE, e);
  return rec[e];
}

// exterior target O(j, s) in two parts, so that the batch kernel (k5_cyk_ext) can find the pairs (i, j) of a step with all
// its lanes and then walk only those: rule 7 for ONE pair cell (i, d = j - i) with exterior term t != log 0 ...
ELEMDP_HD void cyk_ext_pair(const ModelView& m, const TableView& T, MaxAcc& a, int j, int s, int i, double t) {
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.big;
  const int d = j - i;
  const double lt = ELEMDP_MUL_RN(m.lam(s), t);
  for (int u = G[A.split_off + s]; u < G[A.split_off + s + 1]; ++u) {
    const int s2 = G[A.split_ent + 2 * u], s1 = G[A.split_ent + 2 * u + 1];
    a.offer(T.o(i, s2) + (T.ldm(ST_P, d, i, s1) + lt), i, j, TT_O_OP, ST_P, s1);
  }
}
// ... and rule 8 (the candidates behind those of rule 7), the value and the record
ELEMDP_HD void cyk_ext_finish(const ModelView& m, const SeqView& q, const TableView& T, const TraceView& R, const Constraint& c,
                              int j, int s, MaxAcc& a) {
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  if (q.unp[j - 1])
    for (int t = I[A.right_off + s]; t < I[A.right_off + s + 1]; ++t) {
      const int s1 = I[A.right_ent + 2 * t], tf = I[A.right_ent + 2 * t + 1];
      if (!allow_right(m, c, q.L, j, s, s1)) continue;
      a.offer(T.o(j - 1, s1) + w_right(m, q, s, tf, j - 1), 0, j - 1, TT_O_O, ST_O, s1);
    }
  T.o(j, s) = a.best;
  R.ext[(size_t)j * T.S + s] = a.tr;
}
ELEMDP_HD void cyk_ext_target(const ModelView& m, const SeqView& q, const TableView& T, const TraceView& R,
                              const Constraint& c, int j, int s) {
  const double NEG = ELEMDP_NEG_INF;
  MaxAcc a;
  const int i0 = (j - q.W > 0) ? j - q.W : 0;
  for (int i = j - 1; i >= i0; --i) {   // (candidate order: pairs by descending i, the first strictly greatest wins)
    const int d = j - i;
    if (!q.pair_ok(i, d)) continue;
    const double t = q.e_ext[q.cell(i, d)];
    if (t == NEG) continue;
    cyk_ext_pair(m, T, a, j, s, i, t);
  }
  cyk_ext_finish(m, q, T, R, c, j, s, a);
}

struct TraceFrame { int16_t i, j; int8_t e; int16_t s; };

// structure letters by code: 0 'O', 1 'L', 2 'R', 3 'H', 4 'B', 5 'I', 6 'M'
ELEMDP_HD char rss_letter(int code) {
  return code == 0 ? 'O' : code == 1 ? 'L' : code == 2 ? 'R' : code == 3 ? 'H' : code == 4 ? 'B' : code == 5 ? 'I' : 'M';
}
ELEMDP_HD void fill_letters(char* rss, int from, int n, int code) {
  const char ch = rss_letter(code);
  for (int p = from; p < from + n; ++p) rss[p] = ch;
}

// Walks the trace from O(L, s0) and writes the motif node per position (`path`, psihat) and the
// structure letters (`rss`: O L R H B I M, blank where nothing was written).  `stack` is caller
// scratch; returns false on overflow.
ELEMDP_HD bool trace_back(const ModelView& m, const SeqView& q, const TableView& T, const TraceView& R, const Constraint& c, int L,
                          int s0, int32_t* path, char* rss, TraceFrame* stack, int cap) {
  const AutomatonLayout& A = m.lay;
  const int32_t* I = m.ints;
  const int32_t* G = m.big;
  const int M = A.M;
  // (l, r) -> state id by search (only used a few times per sequence)
  auto find_state = [&](int l, int r) {
    for (int s = 0; s < A.S; ++s) if (I[A.st_l + s] == l && I[A.st_r + s] == r) return s;
    return -1;
  };
  (void)M;
  int top = 0;
  stack[top++] = TraceFrame{0, (int16_t)L, (int8_t)ST_O, (int16_t)s0};
  auto fill = [&](int from, int n, int code) { fill_letters(rss, from, n, code); };
  while (top > 0) {
    const TraceFrame f = stack[--top];
    const TraceRec t = (f.e == ST_O) ? R.ext[(size_t)f.j * T.S + f.s] : cyk_retrace(m, q, T, c, f.e, f.j - f.i, f.i, f.s);
    if (t.t < 0) continue;  // leaf
    if (top + 3 > cap) return false;
    const int s1 = t.s1;
    const int fr = I[A.st_r + f.s], fl = I[A.st_l + f.s];
    const int s1l = I[A.st_l + s1], s1r = I[A.st_r + s1];
    switch (t.t) {
      case TT_L_L: path[t.l] = fr; stack[top++] = TraceFrame{t.k, t.l, t.e1, (int16_t)s1}; break;
      case TT_O_O: path[t.l] = fr; rss[t.l] = 'O'; stack[top++] = TraceFrame{t.k, t.l, t.e1, (int16_t)s1}; break;
      case TT_2_2: path[t.l] = fr; rss[t.l] = 'M'; stack[top++] = TraceFrame{t.k, t.l, t.e1, (int16_t)s1}; break;
      case TT_E_H: fill(f.i, f.j - f.i, 3); stack[top++] = TraceFrame{t.k, t.l, t.e1, f.s}; break;
      case TT_E_M: case TT_M_B: case TT_2_P: case TT_1_2: case TT_1_B:
        stack[top++] = TraceFrame{t.k, t.l, t.e1, f.s};
        break;
      case TT_P_E: case TT_P_P:
        path[f.i] = s1l; rss[f.i] = 'L'; path[t.l] = fr; rss[t.l] = 'R';
        stack[top++] = TraceFrame{t.k, t.l, t.e1, (int16_t)s1};
        break;
      case TT_O_OP: {
        const int s2 = find_state(fl, s1l);
        stack[top++] = TraceFrame{t.k, t.l, t.e1, (int16_t)s1};
        stack[top++] = TraceFrame{(int16_t)fl, t.k, (int8_t)ST_O, (int16_t)s2};
        break;
      }
      case TT_E_P: {
        const int s2 = find_state(fl, s1l), s3 = find_state(s1r, fr);
        const int n1 = f.j - t.l, n2 = t.k - f.i;
        if (0 == n1) fill(f.i, n2, 4);
        else if (0 == n2) fill(t.l, n1, 4);
        else { fill(f.i, n2, 5); fill(t.l, n1, 5); }
        stack[top++] = TraceFrame{t.l, f.j, (int8_t)ST_L, (int16_t)s3};
        stack[top++] = TraceFrame{f.i, t.k, (int8_t)ST_L, (int16_t)s2};
        stack[top++] = TraceFrame{t.k, t.l, t.e1, (int16_t)s1};
        break;
      }
      case TT_B_12: {
        const int s2 = find_state(s1r, fr);
        stack[top++] = TraceFrame{t.l, f.j, (int8_t)ST_2, (int16_t)s2};
        stack[top++] = TraceFrame{t.k, t.l, t.e1, (int16_t)s1};
        break;
      }
      case TT_M_M: path[f.i] = s1l; rss[f.i] = 'M'; stack[top++] = TraceFrame{t.k, t.l, (int8_t)ST_M, (int16_t)s1}; break;
      default: break;
    }
  }
  return true;
}

}  // namespace elemdp
