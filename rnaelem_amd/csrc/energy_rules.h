// energy_rules.h -- evaluation of the Turner loop terms on flat tables (host/device agnostic).
//
// Same functions as the reference's EnergyParam::sum_ext_m / hairpin_energy / loop_energy
// (RNAelem/energy_param.hpp:686-795): arguments are 0-based INCLUSIVE base positions, values are
// log Boltzmann weights (log 0 = -inf means "not allowed").  Special hairpins are matched on
// 2-bit packed keys instead of substring search in a text list (:722-738).
#pragma once
#include "dp_rules.h"
#include "energy_tables.h"

namespace elemdp {

ELEMDP_HD bool is_au(int type) { return 2 < type; }

// exterior / multiloop stem term (energy_param.hpp:686-708)
ELEMDP_HD double sum_ext_m(const EnergyTables& e, const uint8_t* s, int n, int i, int j, bool ext) {
  const int type = bp_type(s[i], s[j]);
  double z = 0.;
  if (0 <= i - 1 && j + 1 < n) {
    const int five = s[i - 1], three = s[j + 1];
    z = z + (ext ? e.mismatch_ext[type * 25 + five * 5 + three] : e.mismatch_m[type * 25 + five * 5 + three]);
    if (is_au(type)) z = z + e.term_au;
  } else {
    if (0 <= i - 1) z = z + e.dangle5[type * 5 + s[i - 1]];
    if (j + 1 < n) z = z + e.dangle3[type * 5 + s[j + 1]];
    if (is_au(type)) z = z + e.term_au;
  }
  return z;
}

// packed 2-bit key of s[i..j] (inclusive); returns false if an N occurs
ELEMDP_HD bool pack_bases(const uint8_t* s, int i, int j, uint32_t* key) {
  uint32_t k = 0;
  for (int p = i; p <= j; ++p) {
    if (s[p] == 0) return false;
    k = (k << 2) | (uint32_t)(s[p] - 1);
  }
  *key = k;
  return true;
}
ELEMDP_HD int find_key(const uint32_t* keys, int n, uint32_t key) {
  for (int t = 0; t < n; ++t) if (keys[t] == key) return t;
  return -1;
}

// hairpin closed by pair (i,j) (energy_param.hpp:710-742)
ELEMDP_HD double hairpin_energy(const EnergyTables& e, const uint8_t* s, int i, int j) {
  const int d = j - i - 1;
  if (d < 1) return ELEMDP_NEG_INF;
  const int type = bp_type(s[i], s[j]);
  double z = (d <= kMaxLoop) ? e.hairpin[d]
                             : e.hairpin[kMaxLoop] - (e.lxc37 * log(double(d) * (1. / kMaxLoop)) * 10. * (1. / ((37 + 273.15) * 1.98717)));
  uint32_t key;
  if (d < 3) {
  } else if (3 == d) {
    int t = pack_bases(s, i, j, &key) ? find_key(e.tri_key, e.n_tri, key) : -1;
    if (t >= 0) return e.triloop[t];
    else if (is_au(type)) z = z + e.term_au;
  } else if (4 == d) {
    int t = pack_bases(s, i, j, &key) ? find_key(e.tetra_key, e.n_tetra, key) : -1;
    if (t >= 0) return e.tetraloop[t];  // (pair type 7 never occurs for real pairs)
  } else if (6 == d) {
    int t = pack_bases(s, i, j, &key) ? find_key(e.hexa_key, e.n_hexa, key) : -1;
    if (t >= 0) return e.hexaloop[t];
  }
  if (3 < d) z = z + e.mismatch_h[type * 25 + s[i + 1] * 5 + s[j - 1]];
  return z;
}

// interior loop / bulge / stack closed by (i,j) with inner pair (p,q) (energy_param.hpp:744-795)
ELEMDP_HD double loop_energy(const EnergyTables& e, const uint8_t* s, int i, int j, int p, int q) {
  const int type = bp_type(s[i], s[j]);
  const int type2 = bp_type(s[q], s[p]);
  const int u1 = p - i - 1, u2 = j - q - 1;
  const int u = u1 > u2 ? u1 : u2;
  double z;
  if (u1 < 0 || u2 < 0 || kMaxLoop < u1 + u2) {
    z = ELEMDP_NEG_INF;
  } else if (0 == u1 && 0 == u2) {
    z = e.stack[type * 7 + type2];
  } else if (0 == u1 || 0 == u2) {
    z = e.bulge[u];
    if (1 == u) z = z + e.stack[type * 7 + type2];
    else {
      if (is_au(type)) z = z + e.term_au;
      if (is_au(type2)) z = z + e.term_au;
    }
  } else if (u <= 2) {
    if (2 == u1 + u2) z = e.int11[((type * 8 + type2) * 5 + s[i + 1]) * 5 + s[j - 1]];
    else if (1 == u1 && 2 == u2) z = e.int21[(((type * 8 + type2) * 5 + s[i + 1]) * 5 + s[q + 1]) * 5 + s[j - 1]];
    else if (2 == u1 && 1 == u2) z = e.int21[(((type2 * 8 + type) * 5 + s[q + 1]) * 5 + s[i + 1]) * 5 + s[p - 1]];
    else z = e.int22[((((type * 8 + type2) * 5 + s[i + 1]) * 5 + s[p - 1]) * 5 + s[q + 1]) * 5 + s[j - 1]];
  } else {
    z = e.interior[u1 + u2] + e.ninio[u1 > u2 ? u1 - u2 : u2 - u1];
    const double* mm = (1 == u1 || 1 == u2) ? e.mismatch_1ni : (5 == u1 + u2) ? e.mismatch_23i : e.mismatch_i;
    z = z + (mm[type * 25 + s[i + 1] * 5 + s[j - 1]] + mm[type2 * 25 + s[q + 1] * 5 + s[p - 1]]);
  }
  return z;
}

// exp(loop_energy) from the EXPONENTIATED tables x (exp_tables below): products instead of a sum and its exponential, 0 instead of
// log 0.  The BPP filter, which needs every candidate loop of the canonical mask once per direction, takes this form
// (bpp_kernels.hip); the plan keeps the logarithm (loop_energy) that the scan's max-product pass and lambda need.
ELEMDP_HD double loop_weight(const EnergyTables& x, const uint8_t* s, int i, int j, int p, int q) {
  const int u1 = p - i - 1, u2 = j - q - 1;
  if (u1 < 0 || u2 < 0 || kMaxLoop < u1 + u2) return 0.;
  const int type = bp_type(s[i], s[j]);
  const int type2 = bp_type(s[q], s[p]);
  const int u = u1 > u2 ? u1 : u2;
  if (0 == u) return x.stack[type * 7 + type2];
  if (0 == u1 || 0 == u2) {
    if (1 == u) return x.bulge[1] * x.stack[type * 7 + type2];
    double z = x.bulge[u];
    if (is_au(type)) z *= x.term_au;
    if (is_au(type2)) z *= x.term_au;
    return z;
  }
  if (u <= 2) {
    if (2 == u1 + u2) return x.int11[((type * 8 + type2) * 5 + s[i + 1]) * 5 + s[j - 1]];
    if (1 == u1 && 2 == u2) return x.int21[(((type * 8 + type2) * 5 + s[i + 1]) * 5 + s[q + 1]) * 5 + s[j - 1]];
    if (2 == u1 && 1 == u2) return x.int21[(((type2 * 8 + type) * 5 + s[q + 1]) * 5 + s[i + 1]) * 5 + s[p - 1]];
    return x.int22[((((type * 8 + type2) * 5 + s[i + 1]) * 5 + s[p - 1]) * 5 + s[q + 1]) * 5 + s[j - 1]];
  }
  const double* mm = (1 == u1 || 1 == u2) ? x.mismatch_1ni : (5 == u1 + u2) ? x.mismatch_23i : x.mismatch_i;
  return (x.interior[u1 + u2] * x.ninio[u1 > u2 ? u1 - u2 : u2 - u1]) *
         (mm[type * 25 + s[i + 1] * 5 + s[j - 1]] * mm[type2 * 25 + s[q + 1] * 5 + s[p - 1]]);
}

}  // namespace elemdp
