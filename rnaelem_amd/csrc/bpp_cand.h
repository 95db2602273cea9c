// bpp_cand.h -- the candidate loops of rule 6c of the BPP filter as a table (host / device agnostic: the GPU kernels of
// bpp_kernels.hip stage it, the CPU driver tests/emul checks it against loop_weight shape by shape).
#pragma once
#include <cstdint>
#include <cstring>

#include "energy_tables.h"

namespace elemdp {

// Candidate loops of rule 6c as a table (round 4).  A loop with u1 unpaired bases on the left and u2 on the right (T = u1 + u2 <=
// kMaxLoop) weighs, in all but eight small shapes, g(u1, u2) * f(closing pair) * f(inner pair) (energy_param.hpp:775-792): the
// factor of the pair that is NOT the cell being summed is folded into an extra plane of the band tables when that pair's value is
// written (classes below), so a candidate costs one table load and one fma with a coefficient from this table; the lanes of a
// workgroup walk the entries in the same order whatever the sequence (no mask walk, no branch per candidate).
enum { BC_I = 0, BC_N, BC_B, BC_CLASSES };   // generic loops (mismatch_i); 1 x n loops (mismatch_1ni); bulges of two and more bases (term_au)
struct BppCand { double coef; int32_t u1, T; };
constexpr int kBppCandMax = 496;
constexpr int kBppRunMin = 6;                  // from T = 6 on every (u1, T - u1), u1 = 2 .. T - 2, is a generic loop
constexpr int kBppRunMax = 416;
struct BppCandTable {
  int32_t base[BC_CLASSES];                    // first entry of a class in e[]
  int32_t upto[BC_CLASSES][kMaxLoop + 1];      // entries of the class with u1 + u2 <= T (entries are sorted by T)
  // the generic class once more as RUNS: the entries u1 = 2 .. T - 2 of one T are neighbours in a plane row, so the per-sequence
  // kernels take them with 16-byte loads: run_coef[run_off[T] + m] = g(2 + m, T - 2 - m) (symmetric in its arguments: the outside
  // sweep walks the row the other way with the same array), each run padded with zeros to a multiple of four
  int32_t run_off[kMaxLoop + 2];
  BppCand e[kBppCandMax];
  double run_coef[kBppRunMax];                 // (the last four are zeros: the quad a lane takes past the end of its list)
  uint8_t quad_T[kBppRunMax / 4], quad_m[kBppRunMax / 4];   // quad q = run_coef[4q .. 4q+3] belongs to T = quad_T[q], starts at m = quad_m[q]
};
// (the special shapes in the order the kernels deal them to the parts of a cell)
constexpr int kBppSpecialU1[8] = {0, 1, 1, 1, 2, 2, 2, 3};
constexpr int kBppSpecialU2[8] = {1, 0, 1, 2, 1, 2, 3, 2};
inline void build_bpp_cand(const EnergyTables& x, BppCandTable* t) {
  std::memset(t, 0, sizeof(*t));
  int n = 0;
  for (int c = 0; c < BC_CLASSES; ++c) {
    t->base[c] = n;
    for (int T = 0; T <= kMaxLoop; ++T) {
      for (int u1 = 0; u1 <= T; ++u1) {
        const int u2 = T - u1, u = u1 > u2 ? u1 : u2;
        int cls = -1;
        double coef = 0.;
        if (0 == u1 || 0 == u2) {
          if (u >= 2) { cls = BC_B; coef = x.bulge[u]; }                                  // (u = 0: rule 1b; u = 1: special)
        } else if (u > 2 && !(5 == T && (2 == u1 || 2 == u2))) {
          cls = (1 == u1 || 1 == u2) ? BC_N : BC_I;
          coef = x.interior[T] * x.ninio[u1 > u2 ? u1 - u2 : u2 - u1];
        }
        if (cls == c) { t->e[n].coef = coef; t->e[n].u1 = u1; t->e[n].T = T; ++n; }
      }
      t->upto[c][T] = n - t->base[c];
    }
  }
  int r = 0;
  for (int T = 0; T <= kMaxLoop + 1; ++T) {
    t->run_off[T] = r;
    if (T < kBppRunMin || T > kMaxLoop) continue;
    for (int u1 = 2; u1 <= T - 2; ++u1) t->run_coef[r++] = x.interior[T] * x.ninio[u1 > T - u1 ? 2 * u1 - T : T - 2 * u1];
    while (r & 3) t->run_coef[r++] = 0.;
    for (int q = t->run_off[T] / 4; q < r / 4; ++q) { t->quad_T[q] = (uint8_t)T; t->quad_m[q] = (uint8_t)(4 * q - t->run_off[T]); }
  }
  t->quad_T[kBppRunMax / 4 - 1] = kBppRunMin; t->quad_m[kBppRunMax / 4 - 1] = 0;      // the zero quad: any valid address
}
// the eight shapes that do not factorise (stacked bulge, 1x1, 1x2, 2x1, 2x2, 2x3, 3x2): evaluated by loop_weight
constexpr int kBppSpecial = 8;

}  // namespace elemdp
