// plan_rules.h -- parameter-independent per-sequence preparation ("plan"), host/device agnostic.
//
// Everything the reference recomputes inside its sweep on every optimizer step although it does
// not depend on theta / lambda is computed once per batch and kept resident in HBM:
//   * which cells may hold a base pair      EnergyModel::fill_bpp_tables  energy_model.hpp:211-266
//   * the structural term of every rule     the tsc arguments at          energy_model.hpp:346-437
//   * the interior-loop candidates (k,l)    the double loop at            energy_model.hpp:413-426
//     (inside enumeration) and :527-541 (outside enumeration, a superset when C < 30)
#pragma once
#include "dp_rules.h"
#include "energy_rules.h"

namespace elemdp {

struct PlanCfg {
  int32_t no_ene;    // --no-energy: every structural term is 0 (energy_model.hpp:351,372,399,...)
  int32_t min_span;  // smallest span of a base-pair cell: turn+2 = 5, or 1 with NO_TURN (:217)
  int32_t fix_rss;   // structure given per sequence
};

// ndot[p] = number of non-'.' characters in fix[0..p); segment [a,b) is all dots iff equal counts
ELEMDP_HD bool all_dots(const int32_t* ndot, int a, int b) { return ndot == nullptr || ndot[b] == ndot[a]; }

// canonical candidate: cell (i,d) may pair by sequence alone (energy_model.hpp:216-219)
ELEMDP_HD bool canonical_pair(const uint8_t* seq, int L, int W, int min_span, int i, int d) {
  return d >= min_span && d <= W && i + d <= L && bp_type(seq[i], seq[i + d - 1]) > 0;
}

struct PairTerms { double stack, ext, ml, close, hp; };

// structural terms of a kept pair cell (i,d); `inner_ok` = cell (i+1,d-2) is kept too
ELEMDP_HD PairTerms pair_terms(const EnergyTables& e, const PlanCfg& cfg, const uint8_t* seq, int L, const int32_t* ndot,
                               int i, int d, bool inner_ok) {
  const int j = i + d;
  PairTerms t;
  const double NEG = ELEMDP_NEG_INF;
  t.stack = inner_ok ? (cfg.no_ene ? 0. : loop_energy(e, seq, i, j - 1, i + 1, j - 2)) : NEG;     // :351-352
  t.ext = cfg.no_ene ? 0. : sum_ext_m(e, seq, L, i, j - 1, true);                                  // :429-430
  t.ml = cfg.no_ene ? 0. : sum_ext_m(e, seq, L, i, j - 1, false) + e.ml_intern;                    // :372-373
  t.close = cfg.no_ene ? 0. : sum_ext_m(e, seq, L, j - 1, i, false) + (e.ml_closing + e.ml_intern);  // :399-402
  t.hp = cfg.no_ene ? 0. : hairpin_energy(e, seq, i, j - 1);                                       // :407-408
  if (!all_dots(ndot, i + 1, j - 1)) t.hp = NEG;                                                   // :409-410
  return t;
}

// Enumerates the interior-loop candidates of the E cell (i,d) (closing pair = cell (i-1,d+2)) in the
// reference's inside order (l descending, k ascending) over the OUTSIDE set
//   i <= k <= min(j-2, i+C),  k+2 <= l <= j,  (k,l) != (i,j),  P(k,l) kept,  tsc finite
// and calls f(k, l, tsc, in_inside_set) for each; in_inside_set <=> (k-i)+(j-l) <= C.
template <class OkFn, class F>
ELEMDP_HD void enum_interior(const EnergyTables& e, const PlanCfg& cfg, const uint8_t* seq, int L, int W, int C,
                             const int32_t* ndot, const OkFn& ok, int i, int d, F&& f) {
  const int j = i + d;
  for (int l = j; l >= i + 2; --l) {
    const int kmax = (l - 2 < i + C) ? l - 2 : i + C;
    for (int k = i; k <= kmax; ++k) {
      if (k == i && l == j) continue;
      if (l - k > W || !ok(k, l - k)) continue;
      const double tsc = cfg.no_ene ? 0. : loop_energy(e, seq, i - 1, j, k, l - 1);
      if (tsc == ELEMDP_NEG_INF) continue;
      if (!all_dots(ndot, i, k) || !all_dots(ndot, l, j)) continue;  // :419-421
      f(k, l, tsc, (k - i) + (j - l) <= C);
    }
  }
}

// The same enumeration, in the same order, from the END-major pair mask (bit l * (W+1) + (l - k) <=> pair cell (k, l-k)):
// for a fixed end l the candidates k = i .. kmax are one run of bits (k ascending = span descending), so only kept pairs
// are visited -- ~(d-1) short bit runs per E cell instead of up to C * d single tests.  `word(n)` returns the n-th 32-bit
// word of the mask (0 past the end).
template <class WordFn, class F>
ELEMDP_HD void enum_interior_by_end(const EnergyTables& e, const PlanCfg& cfg, const uint8_t* seq, int L, int W, int C,
                                    const int32_t* ndot, const WordFn& word, int i, int d, F&& f) {
  const int j = i + d;
  for (int l = j; l >= i + 2; --l) {
    const int kmax = (l - 2 < i + C) ? l - 2 : i + C;
    const int dhi = (l - i < W) ? l - i : W, dlo = l - kmax;   // spans l - k of the candidates k = i .. kmax
    for (int top = dhi; top >= dlo; top -= 32) {
      const int lo = (top - 31 > dlo) ? top - 31 : dlo;
      const int len = top - lo + 1;
      const long long b0 = (long long)l * (W + 1) + lo;
      const int w = (int)(b0 >> 5), sh = (int)(b0 & 31);
      const unsigned long long two = ((unsigned long long)word(w + 1) << 32) | (unsigned long long)word(w);
      uint32_t m = (uint32_t)(two >> sh);
      if (len < 32) m &= (1u << len) - 1u;
      while (m) {
        const int b = 31 - __builtin_clz(m);
        m &= ~(1u << b);
        const int k = l - (lo + b);
        if (k == i && l == j) continue;
        const double tsc = cfg.no_ene ? 0. : loop_energy(e, seq, i - 1, j, k, l - 1);
        if (tsc == ELEMDP_NEG_INF) continue;
        if (!all_dots(ndot, i, k) || !all_dots(ndot, l, j)) continue;
        f(k, l, tsc, (k - i) + (j - l) <= C);
      }
    }
  }
}

// The NUMBER of candidates enum_interior_by_end visits, from popcounts of the same bit runs -- valid when no loop term is log 0
// except by size (loop_tables_finite, energy_tables.h; a sequence without N) and no structure is imposed (ndot == nullptr): then a kept pair (k, l) is
// dropped only where the loop would hold more than kMaxLoop unpaired bases (loop_energy, energy_rules.h:77).  The count pass of
// the plan builder (k_plan_cells) evaluated the energy of every candidate just to test it against log 0.
template <class WordFn>
ELEMDP_HD int count_interior_by_end(bool no_ene, int L, int W, int C, const WordFn& word, int i, int d) {
  const int j = i + d;
  int n = 0;
  for (int l = j; l >= i + 2; --l) {
    int kmax = (l - 2 < i + C) ? l - 2 : i + C;
    if (!no_ene) {
      const int kcap = i + kMaxLoop - (j - l);
      if (kcap < kmax) kmax = kcap;
      if (kmax < i) break;            // (the right side alone exceeds kMaxLoop from here on)
    }
    const int dhi = (l - i < W) ? l - i : W, dlo = l - kmax;
    for (int top = dhi; top >= dlo; top -= 32) {
      const int lo = (top - 31 > dlo) ? top - 31 : dlo;
      const int len = top - lo + 1;
      const long long b0 = (long long)l * (W + 1) + lo;
      const int w = (int)(b0 >> 5), sh = (int)(b0 & 31);
      const unsigned long long two = ((unsigned long long)word(w + 1) << 32) | (unsigned long long)word(w);
      uint32_t m = (uint32_t)(two >> sh);
      if (len < 32) m &= (1u << len) - 1u;
      n += __builtin_popcount(m);
    }
  }
  if (d <= W && d >= 2) {             // (k, l) = (i, j): the closing pair's stack (rule 1b), not a loop
    const long long b = (long long)j * (W + 1) + d;
    if ((word((int)(b >> 5)) >> (b & 31)) & 1u) --n;
  }
  return n;
}

}  // namespace elemdp
