// host_prep.h -- small host-side pieces of the batch loader shared by the engine and the CPU
// emulation used in tests (pure functions, no device code).
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace elemdp {

// Position weights from pseudo-qualities (RNAelem::set_ws, RNAelem/motif_model.hpp:62-70):
// ws[i] = log((0.01+q_i)/(0.01+mode(q))) for i < L, ws[L] = (q_L == 0 ? -inf : 0).
// mode = the most frequent quality value, the LAST one on ties (max_index, util.hpp:232-241).
// (the weight of a position depends on its quality value and the mode only: a 94 x 94 table of the logarithm and of its
// exponential, the form the linear pipeline reads, replaces a log and an exp per position -- 36 + 15 ms per 10 000 x L=300)
struct WsTables {
  double ws[127 - 33][127 - 33], ews[127 - 33][127 - 33];    // [quality][mode]
  WsTables() {
    for (int q = 0; q < 127 - 33; ++q)
      for (int m = 0; m < 127 - 33; ++m) {
        ws[q][m] = std::log((0.01 + double(q)) / (0.01 + m));
        ews[q][m] = std::exp(ws[q][m]);
      }
  }
};
inline const WsTables& ws_tables() { static const WsTables t; return t; }
inline void position_weights(const uint8_t* qual, int n_qual, double* ws, double* ews = nullptr) {
  int cnt[127 - 33] = {0};
  for (int i = 0; i < n_qual; ++i) {
    int q = qual[i];
    if (q >= 127 - 33) q = 127 - 33 - 1;  // the reference would throw on such input (cnt.at)
    cnt[q] += 1;
  }
  int mode = 0, best = std::numeric_limits<int>::lowest();
  for (int v = 0; v < 127 - 33; ++v)
    if (best <= cnt[v]) { mode = v; best = cnt[v]; }
  const WsTables& t = ws_tables();
  for (int i = 0; i + 1 < n_qual; ++i) {
    const int q = qual[i];
    if (q < 127 - 33) { ws[i] = t.ws[q][mode]; if (ews) ews[i] = t.ews[q][mode]; }
    else { ws[i] = std::log((0.01 + double(q)) / (0.01 + mode)); if (ews) ews[i] = std::exp(ws[i]); }
  }
  ws[n_qual - 1] = (qual[n_qual - 1] == 0) ? -std::numeric_limits<double>::infinity() : 0.;
  if (ews) ews[n_qual - 1] = (qual[n_qual - 1] == 0) ? 0. : 1.;
}

// prefix counts of non-'.' characters of a dot-bracket string (FIX_RSS support)
inline void nondot_prefix(const char* fix, int L, int32_t* ndot) {
  ndot[0] = 0;
  for (int p = 0; p < L; ++p) ndot[p + 1] = ndot[p] + (fix[p] != '.');
}

}  // namespace elemdp
