// lin_fast.h -- table-driven unary phases of the train kernels k4_in / k4_out (device code, gfx950).
//
// Same rules as lin_inside_target_u / lin_outside_target_u (lin_rules.h; reference: the unary rules 1a, 1b, 3a, 3b, 4, 5, 6a,
// 6b and L <- L of RNAelem/energy_model.hpp:366-430 x motif_model.hpp:243-421, statistics motif_trainer.hpp:374-458), but
// everything a (cell, state) lane needs beyond the table values is looked up instead of derived:
//   * the PROGRAM of the state (AutomatonLayout::fp_in / fp_out, built by Automaton::flatten): its own columns, and per
//     unary transition the operand columns and the id of the transition;
//   * the WEIGHT TABLES of the evaluation behind the linear parameter block (lin_params.h): emission weight of transition
//     `id` for base b (right, left) or pair type t (pair), tau included;
//   * the CELL RECORD of the workgroup's cell (built once per cell by a few lanes while the heavy sums run): parsability
//     flags, the two bases a transition into / out of the cell emits, their position weights, and the exponentiated
//     structural terms of both lambda classes, masked by the flags.
// The generic rule code spends ~95 % of its instructions on deriving those per lane (instruction issue, not arithmetic,
// bounds the band kernels); it stays the path of automata with longer lists than kF* and of the CPU emulation that pins the
// rules against the oracle (tests/emul).
// The scan passes run here too: what the scanner functors test on the nodes of an emitting transition
// (motif_scanner.hpp:546-573, 594-622, 715-747; lstat_* and allow_* of the generic code) is one word of ScanFlag bits per
// transition (AutomatonLayout::fs_in / fs_out), the comparisons of the emitted positions with the chosen start are cell flags.
#pragma once
#include "lin_rules.h"

namespace elemdp {

#ifndef ELEMDP_UNARY2
#define ELEMDP_UNARY2 1
#endif
// kFR / kFP / kFL (template parameters below): unary transitions per list the code unrolls -- the longest lists of the automaton
// (AutomatonLayout::fp_max), at most kFastR / kFastP / kFastL.  Every unrolled slot is a load instruction, used or not.

// ---- cell records ---------------------------------------------------------------------------------------------------------
constexpr int kCellInD = 10;    // doubles per cell, inside:  ews(i), ews(j-1), xst[2], xml[2], xcl[2], xhp[2]
constexpr int kCellOutD = 14;   // doubles per cell, outside: ews(i-1), ews(j), xcl[2], xhp[2], xml[2], xsu[2], e_cl, e_hp, e_su, e_ml
// flag bits of a cell record
enum : int { CF_POK = 1, CF_LOK = 2, CF_MOK = 4, CF_EOK = 8, CF_DO2 = 16, CF_DOM = 32, CF_CE = 64, CF_CP = 128,   // inside
             CF_UP = 64, CF_DOL = 128,                                                                              // outside reuse
             // scan passes under the start constraint Ys: the position the cell's left / right emission covers is Ys (inside: i,
             // j - 1; outside: i - 1, j); outside: j is the last position of the sequence
             CF_YL = 1 << 17, CF_YR = 1 << 18, CF_JLAST = 1 << 19 };
ELEMDP_HD int fcol(int packed, int byte) { const int c = (packed >> (8 * byte)) & 0xff; return c == 0xff ? -1 : c; }

// which global value lane k (0..7) of a cell fetches for the inside record, and from which cell
ELEMDP_HD double cell_in_fetch(const SeqView& q, int d, int i, int k) {
  const int c_here = q.cell(i, d), c_up = (i > 0 && d + 2 <= q.W) ? q.cell(i - 1, d + 2) : c_here;
  const int term = k < 2 ? XT_STACK : k < 4 ? XT_ML : k < 6 ? XT_CLOSE : XT_HP;
  return xw_cell(q, k & 1, term, k < 4 ? c_here : c_up);
}
// flags of the inside record (reads the staged context: pair mask, dmin, unpaired flags, bases)
ELEMDP_HD int cell_in_flags(const ModelView& m, const SeqView& q, int d, int i) {
  const int j = i + d;
  const bool pok = q.pair_ok(i, d), lok = q.left_ok(i, d), mok = m_ok(m, q, i, d), eok = q.e_ok(i, d);
  const bool do2 = lok && d > 0 && q.left_ok(i, d - 1) && q.unp[j - 1];
  const bool doM = mok && m_ok(m, q, i + 1, d - 1) && q.unp[i];
  const bool inner = d >= 2 && q.pair_ok(i + 1, d - 2);
  const int bi = i < q.L ? q.seq[i] : 0, bj = j > 0 ? q.seq[j - 1] : 0;
  return (pok ? CF_POK : 0) | (lok ? CF_LOK : 0) | (mok ? CF_MOK : 0) | (eok ? CF_EOK : 0) | (do2 ? CF_DO2 : 0) | (doM ? CF_DOM : 0) |
         ((pok && d >= 2) ? CF_CE : 0) | ((pok && inner) ? CF_CP : 0) | (bi << 8) | (bj << 11) | (bp_type(bi, bj) << 14);
}

// P,E,M,B,1,2,L of target (i, d, state of program P) from the heavy sums *pHB (rule 2) and *pHE (rule 6c, both in LDS: read
// where they are used); stores them
// CON: the start constraint of the scan's second pass (FS = the ScanFlag words; CF_YL / CF_YR set in fl)
template <int kFR, int kFP, int kFL, bool CON = false>
ELEMDP_HD void fast_inside_unary(const AutomatonLayout& A, const int32_t* P, const double* lin, const TableView& T,
                                                  const double* cr, int fl, int d, int i, const double* pHB, const double* pHE, int nrep,
                                                  int rstride, const int32_t* FS = nullptr) {
  const int w0 = P[0], w1 = P[1], w2 = P[2];
  const bool isloop = w0 & 1, wr_pos = w0 & 8;
  const int kl = (w0 >> 2) & 1, nR = (w0 >> 8) & 15, nP = (w0 >> 12) & 15, nL = (w0 >> 16) & 15;
  const bool pok = fl & CF_POK, lok = fl & CF_LOK, mok = fl & CF_MOK, eok = fl & CF_EOK, do2 = fl & CF_DO2, doM = fl & CF_DOM;
  const bool cE = fl & CF_CE, cP = fl & CF_CP;
  const bool doL = isloop && d > 0;
  const bool yl = CON && (fl & CF_YL), yr = CON && (fl & CF_YR);
  const int bi = (fl >> 8) & 7, bj = (fl >> 11) & 7, ty = (fl >> 14) & 7;
  const int d1 = d > 0 ? d - 1 : 0, d2 = d > 1 ? d - 2 : 0, i1 = i + 1;
  // operands of all transitions first: one round of loads
  double tL[kFR], t2[kFR], tE[kFP], tP[kFP], tM[kFL];
  int eR[kFR], eP[kFP], eL[kFL];
#pragma unroll
  for (int u = 0; u < kFR; ++u) {
    eR[u] = P[4 + u];
    tL[u] = T.ldc(ST_L, d1, i, fcol(eR[u], 0), u < nR && doL);
    t2[u] = T.ldc(ST_2, d1, i, fcol(eR[u], 1), u < nR && do2);
  }
#pragma unroll
  for (int u = 0; u < kFP; ++u) {
    eP[u] = P[8 + u];
    tE[u] = T.ldc(ST_E, d2, i1, fcol(eP[u], 0), u < nP && cE);
    tP[u] = T.ldc(ST_P, d2, i1, fcol(eP[u], 1), u < nP && cP);
  }
#pragma unroll
  for (int u = 0; u < kFL; ++u) {
    eL[u] = P[12 + u];
    tM[u] = T.ldc(ST_M, d1, i1, fcol(eL[u], 0), u < nL && doM);
  }
  const double ews_i = cr[0], ews_j = cr[1];
  const double xst = cr[2 + kl], xml = cr[4 + kl], xcl = cr[6 + kl], xhp = cr[8 + kl];   // (0 where pok / eok is not set)
  const double pj = wr_pos ? ews_j : 1.;
  double sL = 0., s2 = 0., sP = 0., sM = 0.;
#pragma unroll
  for (int u = 0; u < kFR; ++u)
    if (u < nR) {
      const int id = (eR[u] >> 16) & 0x7fff;
      double w = lin[A.lin_wr + 5 * id + bj] * pj;
      if (CON && yr && !(FS[id] & SF_SR)) w = 0.;   // allow_right
      sL = fma(tL[u], w, sL);
      s2 = fma(t2[u], w, s2);
    }
#pragma unroll
  for (int u = 0; u < kFP; ++u)
    if (u < nP) {
      const int id = (eP[u] >> 16) & 0x7fff;
      double w = lin[A.lin_wp + 8 * id + ty] * ((eP[u] < 0 ? ews_i : 1.) * pj);
      if (CON && (yl || yr)) {   // allow_pair
        const int sf = FS[A.n_wr + A.n_wl + id];
        if ((yl && !(sf & SF_SL)) || (yr && !(sf & SF_SR))) w = 0.;
      }
      sP = fma(w, fma(tP[u], xst, tE[u]), sP);
    }
#pragma unroll
  for (int u = 0; u < kFL; ++u)
    if (u < nL) {
      const int id = (eL[u] >> 16) & 0x7fff;
      double w = lin[A.lin_wl + 5 * id + bi] * (eL[u] < 0 ? ews_i : 1.);
      if (CON && yl && !(FS[A.n_wr + id] & SF_SL)) w = 0.;   // allow_left
      sM = fma(tM[u], w, sM);
    }
  double HB = *pHB, HE = *pHE;
  for (int r = 1; r < nrep; ++r) { HB += pHB[r * rstride]; HE += pHE[r * rstride]; }   // (deterministic mode: one copy per wave)
  const double vL = isloop ? (d == 0 ? ((w0 & 2) ? 1. : 0.) : sL) : 0.;   // motif_trainer.hpp:89-95
  const double vP = pok ? sP : 0.;                                          // rules 1a, 1b
  const double vB = lok ? HB : 0.;                                          // rule 2
  const double v2 = lok ? fma(vP, xml, s2) : 0.;                            // rules 3a, 3b
  const double v1 = lok ? v2 + vB : 0.;                                     // rules 4a, 4b
  const double vM = mok ? sM + vB : 0.;                                     // rules 5a, 5b
  const double vE = eok ? fma(vM, xcl, fma(vL, xhp, HE)) : 0.;              // rules 6a, 6b, 6c
  // (the B plane is not stored: nothing reads it -- the outside pass of the train kernels decides liveness from the pair entries)
  const int cLo = fcol(w2, 2), cPo = fcol(w1, 0), c2o = fcol(w2, 1), c1o = fcol(w2, 0), cMo = fcol(w1, 2), cEo = fcol(w1, 1);
  if (cLo >= 0) T.band[T.cidx(ST_L, d, i, cLo)] = vL;
  if (pok && cPo >= 0) {
    T.band[T.cidx(ST_P, d, i, cPo)] = vP;
    // X = P * exp(lambda e_ml): what rule 3b hands to the stems of the factorised rule 2 (pair phases of k4_in / k4_out).  It takes
    // the rows of the B plane, which the train kernels leave unused, under P's columns (AutomatonLayout::fp_ok checks the fit).
    T.band[T.cidx(ST_B, d, i, cPo)] = vP * xml;
  }
  if (lok && c2o >= 0) T.band[T.cidx(ST_2, d, i, c2o)] = v2;
  if (lok && c1o >= 0) T.band[T.cidx(ST_1, d, i, c1o)] = v1;
  if (mok && cMo >= 0) T.band[T.cidx(ST_M, d, i, cMo)] = vM;
  if (eok && cEo >= 0) T.band[T.cidx(ST_E, d, i, cEo)] = vE;
}

// ---- outside --------------------------------------------------------------------------------------------------------------
// value k (0..11) of the outside record of cell (i, d): exponentiated terms (masked later) and the raw terms of the statistics
ELEMDP_HD double cell_out_fetch(const SeqView& q, int d, int i, int k) {
  const int c_here = q.cell(i, d), c_up = (i > 0 && d + 2 <= q.W && i + d < q.L) ? q.cell(i - 1, d + 2) : c_here;
  switch (k) {
    case 0: case 1: return xw_cell(q, k & 1, XT_CLOSE, c_up);
    case 2: case 3: return xw_cell(q, k & 1, XT_HP, c_up);
    case 4: case 5: return xw_cell(q, k & 1, XT_ML, c_here);
    case 6: case 7: return xw_cell(q, k & 1, XT_STACK, c_up);
    case 8: return q.e_close[c_up];
    case 9: return q.e_hp[c_up];
    case 10: return q.e_stack[c_up];
    default: return q.e_ml[c_here];
  }
}
ELEMDP_HD int cell_out_flags(const ModelView& m, const SeqView& q, int d, int i) {
  const int j = i + d;
  const bool pok = q.pair_ok(i, d), lok = q.left_ok(i, d), mok = m_ok(m, q, i, d), eok = q.e_ok(i, d);
  const bool up_ok = q.pair_ok(i - 1, d + 2);
  const bool doM = mok && m_ok(m, q, i - 1, d + 1) && q.unp[i > 0 ? i - 1 : 0];
  const bool do2 = lok && q.left_ok(i, d + 1) && q.unp[j];
  const bool doLc = j < q.L && d + 1 <= q.W;
  const int bl = i > 0 ? q.seq[i - 1] : 0, br = j < q.L ? q.seq[j] : 0;
  return (pok ? CF_POK : 0) | (lok ? CF_LOK : 0) | (mok ? CF_MOK : 0) | (eok ? CF_EOK : 0) | (do2 ? CF_DO2 : 0) | (doM ? CF_DOM : 0) |
         (up_ok ? CF_UP : 0) | (doLc ? CF_DOL : 0) | (bl << 8) | (br << 11) | (bp_type(bl, br) << 14);
}
// which flag masks value k of the outside record (a value that is not masked is only used under a non-zero posterior)
ELEMDP_HD bool cell_out_mask(int fl, int k) {
  if (k < 4) return fl & CF_EOK;
  if (k < 6) return fl & CF_POK;
  if (k < 8) return (fl & CF_UP) && (fl & CF_POK);
  return true;
}

// Position posteriors of one emitting transition in the scan passes (lstat_pair / lstat_left / lstat_right of lin_rules.h with
// the node tests as ScanFlag bits sf): OUT_SCAN adds the posterior z to the start / inner accumulators, OUT_END to the end
// accumulator; returns false when the start constraint of OUT_END excludes the transition.  kl / kr: the positions of the
// left / right emission (LEFT / RIGHT: which of them the transition has); yl / yr: that position is the chosen start.
template <int MODE, bool LEFT, bool RIGHT, class Sink>
ELEMDP_HD bool fast_scan_stat(Sink& sink, int sf, int kl, int kr, bool yl, bool yr, bool jlast, double z) {
  if (MODE == OUT_END) {
    if ((LEFT && yl && !(sf & SF_SL)) || (RIGHT && yr && !(sf & SF_SR))) return false;
    if (z != 0.) {
      if (LEFT && (sf & SF_EL)) sink.pos(2, kl, z);
      if (RIGHT && (sf & SF_ER)) sink.pos(2, kr, z);
      if (RIGHT && (sf & SF_PM2) && jlast) sink.pos(2, kr + 1, z);
    }
  } else if (MODE == OUT_SCAN && z != 0.) {
    if (LEFT && (sf & SF_SL)) sink.pos(0, kl, z);
    if (RIGHT && (sf & SF_SR)) sink.pos(0, kr, z);
    if (LEFT && (sf & SF_IL)) sink.pos(1, kl, z);
    if (RIGHT && (sf & SF_IR)) sink.pos(1, kr, z);
  }
  return true;
}

// band target (i, d, state of program P) of an outside sweep; returns out B.  invZ: of the lane's world.
// (heavy sums H1, H2, HP, HL of the target at ph[0], ph[CS], ph[2 CS], ph[3 CS] in LDS: read where they are used)
// MODE: OUT_TRAIN (expected counts + energy statistics), OUT_SCAN (counts + start / inner posteriors), OUT_END (end posteriors
// under the start constraint: CF_YL / CF_YR / CF_JLAST in fl); FS = the ScanFlag words of the scan modes.
template <int kFR, int kFP, int kFL, int MODE, class Sink>
ELEMDP_HD double fast_outside_unary(const AutomatonLayout& A, const int32_t* P, const int32_t* G, const double* lin,
                                                     const TableView& in, const TableView& out, const double* cr, int fl, int d, int i,
                                                     double invZ, bool lam_same, bool no_prf, Sink& sink, const double* ph, int CS, int nrep,
                                                     int rstride, const int32_t* FS = nullptr) {
  constexpr bool SCANM = MODE == OUT_SCAN || MODE == OUT_END;
  const bool yl = fl & CF_YL, yr = fl & CF_YR, jlast = fl & CF_JLAST;
  const int j = i + d;
  const int w0 = P[0], w1 = P[1], w2 = P[2], enl = P[3];
  const bool isloop = w0 & 1, wl_s = w0 & 16;
  const int kl = (w0 >> 2) & 1, nRR = (w0 >> 8) & 15, nRP = (w0 >> 12) & 15, nRL = (w0 >> 16) & 15;
  const bool pok = fl & CF_POK, lok = fl & CF_LOK, mok = fl & CF_MOK, eok = fl & CF_EOK, do2 = fl & CF_DO2, doM = fl & CF_DOM;
  const bool up_ok = fl & CF_UP, doL = isloop && (fl & CF_DOL);
  const int bl = (fl >> 8) & 7, br = (fl >> 11) & 7, ty = (fl >> 14) & 7;
  const int cLo = fcol(w2, 2), cPo = fcol(w1, 0), c2o = fcol(w2, 1), c1o = fcol(w2, 0), cMo = fcol(w1, 2), cEo = fcol(w1, 1);
  const int dp1 = d + 1, dp2 = d + 2, im1 = i - 1;   // (only dereferenced where the parent cell exists: do2 / doL / up_ok / doM)
  // Two batches of table operands: (A) what the pair and left parents need -- E, P, M of the cell, the parents' P and M --, then
  // (B) the rest.  One batch (everything in flight together) costs ~30 more vector registers than the rest of the kernel and
  // so a workgroup per CU; ELEMDP_UNARY2 = 0 keeps it for comparison.  The B planes and the outside plane 1 are not stored
  // here: nothing reads them (the generic kernels, which debug_tables uses, store every plane).
  const double ews_l = cr[0], ews_r = cr[1];
  const int ehs = lam_same ? 0 : kl;
  const double inE = in.ldc(ST_E, d, i, cEo, eok), inM = in.ldc(ST_M, d, i, cMo, mok), inP = in.ldc(ST_P, d, i, cPo, pok);
  double opP[kFP], opM[kFL];
  int eP[kFP], eL[kFL];
#pragma unroll
  for (int u = 0; u < kFP; ++u) {
    eP[u] = P[8 + u];
    opP[u] = out.ldc(ST_P, dp2, im1, fcol(eP[u], 0), u < nRP && up_ok);
  }
#pragma unroll
  for (int u = 0; u < kFL; ++u) {
    eL[u] = P[12 + u];
    opM[u] = out.ldc(ST_M, dp1, im1, fcol(eL[u], 0), u < nRL && doM);
  }
#if !ELEMDP_UNARY2
  const double in1 = in.ldc(ST_1, d, i, c1o, lok), in2 = in.ldc(ST_2, d, i, c2o, lok), inL = in.ldc(ST_L, d, i, cLo, isloop);
  const double r7 = out.ldc(ST_P, d, i, cPo, pok);
  double op2[kFR], opL[kFR];
  int eR[kFR];
#pragma unroll
  for (int u = 0; u < kFR; ++u) {
    eR[u] = P[4 + u];
    op2[u] = out.ldc(ST_2, dp1, i, fcol(eR[u], 0), u < nRR && do2);
    opL[u] = out.ldc(ST_L, dp1, i, fcol(eR[u], 1), u < nRR && doL && eR[u] < 0);   // (sign bit: the parent is a loop state)
  }
#endif
  const double inEz = inE * invZ, inPz = inP * invZ, inMz = inM * invZ;
  const bool aE = up_ok && inE != 0., aP = up_ok && pok && inP != 0., aM = doM && inM != 0.;
  double oE = 0., oP1b = 0., sM = 0., s2 = 0., sL = 0.;
  {
    const double xsu0 = cr[8], xsu1 = cr[9], e_su = cr[12];
    // E as child of P(i-1,j+1,par) (rule 1a) and P as child of it (rule 1b): same parents, same emission
#pragma unroll
    for (int u = 0; u < kFP; ++u)
      if (u < nRP && (aE || aP)) {
        const int id = (eP[u] >> 16) & 0x7fff;
        const int f = G[A.fe_p + 3 * id], offR = G[A.fe_p + 3 * id + 1], offL = G[A.fe_p + 3 * id + 2];
        const double w = lin[A.lin_wp + 8 * id + ty] * (((f & 2) ? ews_l : 1.) * ((f & 4) ? ews_r : 1.));
        const double tE = aE ? opP[u] * w : 0.;
        const double tP = aP ? opP[u] * (w * ((f & 8) ? xsu1 : xsu0)) : 0.;
        const double zP = tP * inPz, z = fma(tE, inEz, zP);
        if (SCANM && !fast_scan_stat<MODE, true, true>(sink, FS[A.n_wr + A.n_wl + id], i - 1, j, yl, yr, jlast, z)) continue;
        if (MODE != OUT_END && !no_prf && z != 0.) {   // expected emission counts (motif_trainer.hpp:384-388 / profile_hmm.hpp:144-179)
          if (f & 1) { if (ty) sink.en(offR + ty, z); }
          else { if (bl) sink.en(offL + bl, z); if (br) sink.en(offR + br, z); }
        }
        if (MODE == OUT_TRAIN && zP != 0.) sink.eh(lam_same ? 0 : ((f >> 3) & 1), e_su * zP);   // motif_trainer.hpp:380-381
        oE += tE;
        oP1b += tP;
      }
  }
  // M as child of M(i-1,j,par) (rule 5a): left emission by the l-node of this state
#pragma unroll
  for (int u = 0; u < kFL; ++u)
    if (u < nRL && aM) {
      const int id = (eL[u] >> 16) & 0x7fff;
      const double term = opM[u] * (lin[A.lin_wl + 5 * id + bl] * (wl_s ? ews_l : 1.));
      const double z = term * inMz;
      if (SCANM && !fast_scan_stat<MODE, true, false>(sink, FS[A.n_wr + id], i - 1, j, yl, yr, jlast, z)) continue;
      if (MODE != OUT_END && !no_prf && z != 0. && bl) sink.en(enl + bl, z);
      sM += term;
    }
  if (eok && cEo >= 0) out.band[out.cidx(ST_E, d, i, cEo)] = oE;
  double oM = 0.;
  if (inM != 0.) {   // child of E (6a) and of M(i-1,j,par) (5a)
    const double t6a = oE * cr[2 + kl], z = t6a * inMz;
    if (MODE == OUT_TRAIN && z != 0.) sink.eh(ehs, cr[10] * z);
    oM = t6a + sM;
  }
  if (mok && cMo >= 0) out.band[out.cidx(ST_M, d, i, cMo)] = oM;
#if ELEMDP_UNARY2
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
  const double in1 = in.ldc(ST_1, d, i, c1o, lok), in2 = in.ldc(ST_2, d, i, c2o, lok), inL = in.ldc(ST_L, d, i, cLo, isloop);
  const double r7 = out.ldc(ST_P, d, i, cPo, pok);
  double op2[kFR], opL[kFR];
  int eR[kFR];
#pragma unroll
  for (int u = 0; u < kFR; ++u) {
    eR[u] = P[4 + u];
    op2[u] = out.ldc(ST_2, dp1, i, fcol(eR[u], 0), u < nRR && do2);
    opL[u] = out.ldc(ST_L, dp1, i, fcol(eR[u], 1), u < nRR && doL && eR[u] < 0);   // (sign bit: the parent is a loop state)
  }
#endif
  const double in2z = in2 * invZ, inLz = inL * invZ;
  const bool a2 = do2 && in2 != 0., aL = doL && inL != 0.;
  // 2 and L as children of 2 / L (i,j+1,par) (rules 3a, L <- L): right emission by the parent's r-node
#pragma unroll
  for (int u = 0; u < kFR; ++u)
    if (u < nRR && (a2 || aL)) {
      const int id = (eR[u] >> 16) & 0x7fff;
      const int enr = G[A.fe_r + 2 * id], fr = G[A.fe_r + 2 * id + 1];
      const double w = lin[A.lin_wr + 5 * id + br] * ((fr & 1) ? ews_r : 1.);
      const double t2 = a2 ? op2[u] * w : 0., tL = aL ? opL[u] * w : 0.;
      const double z = fma(t2, in2z, tL * inLz);
      if (SCANM && !fast_scan_stat<MODE, false, true>(sink, FS[id], i - 1, j, yl, yr, jlast, z)) continue;
      if (MODE != OUT_END && !no_prf && z != 0. && br) sink.en(enr + br, z);
      s2 += t2;
      sL += tL;
    }
  // 1 (heavy sum H1), B (child of M (5b) and of 1 (4b): its inside value is non-zero wherever a pair entry that takes it is),
  // 2 (child of 1 (4a), of 2(i,j+1,par) (3a); the rule-2 part reaches P as H2 = HA)
  double H1 = ph[0], H2 = ph[CS], HP = ph[2 * CS], HL = ph[3 * CS];
  for (int r = 1; r < nrep; ++r) {   // (deterministic mode: one copy per wave)
    H1 += ph[r * rstride]; H2 += ph[r * rstride + CS]; HP += ph[r * rstride + 2 * CS]; HL += ph[r * rstride + 3 * CS];
  }
  const double o1 = (in1 != 0.) ? H1 : 0.;
  const double oB = lok ? (mok ? oM : 0.) + o1 : 0.;
  const double o2 = (in2 != 0.) ? o1 + s2 : 0.;
  if (lok && c2o >= 0) out.band[out.cidx(ST_2, d, i, c2o)] = o2;
  double oP = 0.;
  if (inP != 0.) {   // child of O (7: r7), of P(i-1,j+1,par) (1b), of 2 (3b), inner pair of interior loops (6c: HP)
    const double t3b = (o2 + H2) * cr[6 + kl], z = t3b * inPz;
    if (MODE == OUT_TRAIN && z != 0.) sink.eh(ehs, cr[13] * z);
    oP = oP1b + t3b + (HP + r7);
  }
  if (pok && cPo >= 0) out.band[out.cidx(ST_P, d, i, cPo)] = oP;
  double oL = 0.;
  if (inL != 0.) {   // child of E (6b), of L(i,j+1,par), loops of interior loops (6c: HL)
    const double t6b = oE * cr[4 + kl], z = t6b * inLz;
    if (MODE == OUT_TRAIN && z != 0.) sink.eh(ehs, cr[11] * z);
    oL = t6b + sL + HL;
  }
  if (cLo >= 0) out.band[out.cidx(ST_L, d, i, cLo)] = oL;
  return oB;
}

}  // namespace elemdp
