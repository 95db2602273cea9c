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
inline void position_weights(const uint8_t* qual, int n_qual, double* ws) {
  int cnt[127 - 33] = {0};
  for (int i = 0; i < n_qual; ++i) {
    int q = qual[i];
    if (q >= 127 - 33) q = 127 - 33 - 1;  // the reference would throw on such input (cnt.at)
    cnt[q] += 1;
  }
  int mode = 0, best = std::numeric_limits<int>::lowest();
  for (int v = 0; v < 127 - 33; ++v)
    if (best <= cnt[v]) { mode = v; best = cnt[v]; }
  for (int i = 0; i + 1 < n_qual; ++i) ws[i] = std::log((0.01 + double(qual[i])) / (0.01 + mode));
  ws[n_qual - 1] = (qual[n_qual - 1] == 0) ? -std::numeric_limits<double>::infinity() : 0.;
}

// prefix counts of non-'.' characters of a dot-bracket string (FIX_RSS support)
inline void nondot_prefix(const char* fix, int L, int32_t* ndot) {
  ndot[0] = 0;
  for (int p = 0; p < L; ++p) ndot[p + 1] = ndot[p] + (fix[p] != '.');
}

}  // namespace elemdp
