// lin_params.h -- host side of the scaled-linear train pipeline: the per-evaluation parameter block
// (see lin_rules.h for what the kernels do with it).  No DP code here.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "device_layout.h"

namespace elemdp {

// layout of the per-evaluation linear parameter block (doubles): tau, psb[5], log2 psb[5], eth[n_theta], then the weight tables
// of the table-driven unary phases: WR[n_wr][5], WL[n_wl][5], WP[n_wp][8] (offsets in AutomatonLayout::lin_w*)
constexpr int kLinTau = 0, kLinPsb = 1, kLinPl2 = 6, kLinEth = 11;

// bases (codes A,C,G,U = 1..4) of base-pair type t = 1..6 = CG,GC,GU,UG,AU,UA (bio_sequence.hpp:20-26)
inline int bp_left(int t) { return t == 1 ? 2 : t == 2 ? 3 : t == 3 ? 3 : t == 4 ? 4 : t == 5 ? 1 : 4; }
inline int bp_right(int t) { return t == 1 ? 3 : t == 2 ? 2 : t == 3 ? 4 : t == 4 ? 3 : t == 5 ? 4 : 1; }

// Linear parameter block from the log-space theta (after the softmax, if any):
//   psb[b]  = 2^-round(log2 of the background emission of base b): the per-position scale of the similarity
//             transform (exact powers of two; 1 for N and under --no-profile)
//   eth[..] = exp(theta) * psb of the emitted base(s): 4-column rows emit one base, 6-column rows a base pair
inline void make_lin_params(const AutomatonLayout& lay, const int32_t* ints, const double* theta, double tau, bool no_prf,
                            std::vector<double>* out) {
  out->assign(kLinEth + lay.n_theta, 1.);
  double* p = out->data();
  p[kLinTau] = tau;
  const int32_t* row_off = ints + lay.row_off;
  const int bg_row = ints[lay.st_row_r + lay.s00];
  for (int b = 0; b < 5; ++b) {
    double e2 = 0.;
    if (!no_prf && b > 0 && bg_row >= 0) {
      e2 = -std::rint(theta[row_off[bg_row] + b - 1] * 1.4426950408889634);
      if (!(e2 > -1000.)) e2 = -1000.;   // (also catches NaN)
      if (e2 > 1000.) e2 = 1000.;
    }
    p[kLinPl2 + b] = e2;
    p[kLinPsb + b] = std::ldexp(1., (int)e2);
  }
  for (int r = 0; r < lay.n_rows; ++r) {
    const int w = row_off[r + 1] - row_off[r];
    for (int c = 0; c < w; ++c) {
      const double sc = (w == 6) ? p[kLinPsb + bp_left(c + 1)] * p[kLinPsb + bp_right(c + 1)] : p[kLinPsb + c + 1];
      p[kLinEth + row_off[r] + c] = no_prf ? 1. : std::exp(theta[row_off[r] + c]) * sc;
    }
  }
  // Weight tables of the table-driven unary phases (lin_fast.h; AutomatonLayout::lin_w*): the emission weight of every unary
  // transition for every base (right, left: lw_right / lw_left of lin_rules.h without the position weight) or pair type
  // (pair: lw_pair; type 0 = not a canonical pair never occurs under a canonical pair mask).
  if (lay.lin_total <= kLinEth + lay.n_theta) return;
  out->resize(lay.lin_total, 1.);
  p = out->data();
  auto eth = [&](int row, int col) { return (no_prf || row < 0) ? 1. : p[kLinEth + row_off[row] + col]; };
  for (int s = 0; s < lay.S; ++s) {
    for (int e = ints[lay.right_off + s]; e < ints[lay.right_off + s + 1]; ++e) {
      const double t = ints[lay.right_ent + 2 * e + 1] ? tau : 1.;
      for (int b = 0; b < 5; ++b) p[lay.lin_wr + 5 * e + b] = (b ? eth(ints[lay.st_row_r + s], b - 1) : 1.) * t;
    }
    for (int e = ints[lay.left_off + s]; e < ints[lay.left_off + s + 1]; ++e) {
      const int ch = ints[lay.left_ent + 2 * e];
      const double t = ints[lay.left_ent + 2 * e + 1] ? tau : 1.;
      for (int b = 0; b < 5; ++b) p[lay.lin_wl + 5 * e + b] = (b ? eth(ints[lay.st_row_l + ch], b - 1) : 1.) * t;
    }
    for (int e = ints[lay.pair_off + s]; e < ints[lay.pair_off + s + 1]; ++e) {
      const int ch = ints[lay.pair_ent + 2 * e];
      const double t = ints[lay.pair_ent + 2 * e + 1] ? tau : 1.;
      for (int ty = 0; ty < 8; ++ty) {
        double w = 1.;
        if (ty >= 1 && ty <= 6 && !no_prf)
          w = ints[lay.st_pair_r + s] ? eth(ints[lay.st_row_r + s], ty - 1)
                                      : eth(ints[lay.st_row_l + ch], bp_left(ty) - 1) * eth(ints[lay.st_row_r + s], bp_right(ty) - 1);
        p[lay.lin_wp + 8 * e + ty] = w * t;
      }
    }
  }
}

}  // namespace elemdp
