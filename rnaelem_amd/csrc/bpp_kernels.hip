// bpp_kernels.hip -- the BPP filter K1 (EnergyModel::set_seq .. fill_bpp_tables .. lnBPP, RNAelem/energy_model.hpp:188-276) in
// the LINEAR semiring, without a plan of the unfiltered pair mask.
//
// The filter is the plain McCaskill partition function (the one-state automaton, lambda = 1, no emissions) followed by
// ln BPP(i,j) = ln(inside P * outside P / Z) >= ln(min_bpp).  Round 1 ran it in log space over an interior-loop item list of
// the UNFILTERED canonical mask (~10x the items of the final plan: building that list was more than half of
// elemdp_load_batch).  Here:
//   * band tables hold Boltzmann weights (a span is at most W bases, so the values stay far inside the double range
//     whatever the sequence length); only the two exterior chains, whose values grow with L, are kept as logarithms,
//     and the outside band values are divided by Z from the start (rule 7 enters as exp(ln O(i) + ln outO(j) - ln Z));
//   * rule 2 is factorised as in lin_rules.h (A(i,j) = sum_k 1(i,k) 2(k,j) by a recurrence along the row + the stems that
//     end at j);
//   * the interior loops of rule 6c are enumerated in place from the pair mask, one lane per (cell, left end), loop_energy evaluated on
//     the fly (the same energy_rules.h functions the plan builder calls) -- each candidate is needed once per direction, so
//     materialising the list only cost bandwidth.
// Same sums as k3_bpp_* (train_kernels.hip), which stay for spans beyond the linear range (W > kBppLinMaxSpan).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "kernels.h"
#include "plan_rules.h"

namespace elemdp {
namespace {

enum { BP_P = 0, BP_E, BP_M, BP_B, BP_1, BP_2, BP_A, BP_X, BP_PM = BP_X + BC_CLASSES, BP_IN_PLANES };   // inside planes; BP_X + c: P times the inner
                                                             // pair's factor of class c; BP_PM: P times the multiloop term of its pair (rule 2's stems)
enum { BO_P = 0, BO_E, BO_M, BO_2, BO_A, BO_X, BO_OUT_PLANES = BO_X + BC_CLASSES };   // outside planes (divided by Z); BO_X + c: E times the closing pair's factor of class c
static_assert(BP_IN_PLANES == kBppInPlanes && BO_OUT_PLANES == kBppOutPlanes, "plane counts of kernels.h");
enum { XW_STACK = 0, XW_EXT, XW_ML, XW_CLOSE, XW_HP };

struct Seq {
  int L, W, C;
  const uint8_t* seq;
  const uint32_t* ok;     // canonical pair mask, bit i * (W+1) + d
  const int16_t* dmin;
  const double* xw;       // + term * xw_stride + d * (L+1) + i
  size_t xw_stride;
  double* tin;            // + plane * t_stride + d * (L+1) + i
  double* tout;
  size_t t_stride;
  double* lo_in;          // ln O(j), j = 0 .. L
  double* lo_out;         // ln outO(j) - ln Z
  __device__ __forceinline__ bool pair_ok(int i, int d) const {
    if (i < 0 || d < 0 || d > W || i + d > L) return false;
    const int c = i * (W + 1) + d;
    return (ok[c >> 5] >> (c & 31)) & 1u;
  }
  __device__ __forceinline__ int cell(int i, int d) const { return d * (L + 1) + i; }   // (terms are stored by diagonal, like the tables)
  __device__ __forceinline__ double x(int term, int c) const { return xw[(size_t)term * xw_stride + c]; }
  __device__ __forceinline__ double& in(int plane, int d, int i) const { return tin[(size_t)plane * t_stride + (size_t)d * (L + 1) + i]; }
  __device__ __forceinline__ double& out(int plane, int d, int i) const { return tout[(size_t)plane * t_stride + (size_t)d * (L + 1) + i]; }
  __device__ __forceinline__ bool left_ok(int i, int d) const {
    if (d > W || d < 0 || i + d > L) return false;
    const int dm = dmin[i];
    return dm > 0 && d >= dm;
  }
  __device__ __forceinline__ bool m_ok(int i, int d, int m_min) const { return 0 < i && i + d < L && d <= W && m_min <= d; }
};

__device__ __forceinline__ Seq make_seq(const BppLinArgs& a, int n) {
  const SeqPlan p = a.plans[n];
  Seq q;
  q.L = p.L; q.W = p.W; q.C = p.C;
  q.seq = a.seq + p.seq_base;
  q.ok = a.okbits + p.bits_base;
  q.dmin = a.dmin + p.dmin_base;
  q.xw = a.xw + p.cell_base; q.xw_stride = a.xw_stride;
  q.tin = a.tin + p.cell_base; q.tout = a.tout + p.cell_base; q.t_stride = a.t_stride;
  q.lo_in = a.lo_in + p.dmin_base; q.lo_out = a.lo_out + p.dmin_base;
  return q;
}

// set bits bit0 + n, n in [lo, hi], of a mask: calls f(n) (ascending)
template <class F> __device__ __forceinline__ void for_bits(const uint32_t* m, int bit0, int lo, int hi, F f) {
  if (hi < lo) return;
  const int b = bit0 + lo, e = bit0 + hi;
  int w = b >> 5;
  uint32_t word = m[w] & (~0u << (b & 31));
  for (;;) {
    while (word) {
      const int bit = (w << 5) + __builtin_ctz(word);
      if (bit > e) return;
      f(bit - bit0);
      word &= word - 1;
    }
    if (((w + 1) << 5) > e) return;
    word = m[++w];
  }
}

__device__ void build_pair_lists(const Seq& q, int16_t* plist, int32_t* poff, int* cnt);
// ---- exp of the structural terms of every canonical pair, and dmin (and the pairs of every diagonal: build_pair_lists)
__global__ __launch_bounds__(kThreads) void k6_terms(BppLinArgs a) {
  const SeqPlan p = a.plans[blockIdx.y];
  const int L = p.L, W = p.W;
  const int ncell = (L + 1) * (W + 1);
  if ((int)(blockIdx.x * kThreads) >= ncell) return;
  const Seq q = make_seq(a, blockIdx.y);
  if (blockIdx.x == 0) {
    int16_t* dmin = a.dmin + p.dmin_base;
    for (int i = threadIdx.x; i <= L; i += kThreads) {
      int dm = 0;
      for (int d = 1; d <= W && i + d <= L; ++d)
        if (q.pair_ok(i, d)) { dm = d; break; }
      dmin[i] = (int16_t)dm;
    }
    if (a.plist) {
      __shared__ int cnt[kBppLinMaxSpan + 2];
      build_pair_lists(q, a.plist + p.cell_base, a.poff + (size_t)blockIdx.y * a.poff_stride, cnt);
    }
  }
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= ncell) return;
  const int d = c / (L + 1), i = c - d * (L + 1);      // (terms are stored by diagonal: the sweeps read them along i)
  double v[5] = {0., 0., 0., 0., 0.};
  if (q.pair_ok(i, d)) {
    const PlanCfg cfg{a.no_ene, a.min_span, 0};
    const PairTerms t = pair_terms(*a.et, cfg, q.seq, L, nullptr, i, d, q.pair_ok(i + 1, d - 2));
    const double e[5] = {t.stack, t.ext, t.ml, t.close, t.hp};
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = (e[k] == ELEMDP_NEG_INF) ? 0. : exp(e[k]);
  }
  double* xw = a.xw + p.cell_base;
#pragma unroll
  for (int k = 0; k < 5; ++k) xw[(size_t)k * a.xw_stride + c] = v[k];
}

// ---- the diagonal kernels: kC consecutive cells per workgroup.  The heavy sums are spread over the lanes as (cell, slot)
// work items -- interior loops only for the cells that have them (a third of the cells; the list is compacted with a
// ballot), slot = left end of the inner / outer pair -- and every partial sum has its own LDS word, added up in a fixed
// order by the cell's lane: the filter is reproducible bit for bit from run to run.  The bases, the first-pair spans and the
// rows of the pair mask the workgroup walks are staged in LDS with one round of loads (a pair test is then an LDS read, not
// a dependent global load), and candidates are taken four at a time so that their table loads are in flight together.
constexpr int kC = 32;        // cells per workgroup
constexpr int kStem = 8;      // lanes per cell for the stem sums
constexpr int kSlots = 16;    // partial sums per cell for the interior loops: slot s takes the left ends at distance s and 31 - s
                              // (+ 32, ..): the candidates of a left end fall off linearly with its distance, so a slot's two
                              // ends together are the same work for every slot -- the lanes of a wave finish together
constexpr int kWin = 384;     // staged bases
constexpr int kB = 4;         // candidates per batch of loads

struct BppLds {
  double stem[kC][kStem + 1];
  double stem2[kC][kStem + 1];
  double loop[kC][kSlots + 1];
  int list[kC];
  int n_list;
  int dmin[kC];
  uint8_t seq[kWin];
};
extern __shared__ uint32_t s_bits[];   // mask rows of the workgroup (bpp_mask_words(W) words)

__host__ __device__ inline int bpp_mask_rows(int W) { return kC + W + kMaxLoop + 6; }
__host__ __device__ inline int bpp_mask_words(int W) { return (bpp_mask_rows(W) * (W + 1) + 31) / 32 + 3; }

// stage s[lo..hi] and return a pointer p with p[x] = s[x] for lo <= x <= hi (the plain pointer if the window is too long)
__device__ __forceinline__ const uint8_t* stage_seq(const Seq& q, uint8_t* buf, int lo, int hi) {
  if (lo < 0) lo = 0;
  if (hi > q.L - 1) hi = q.L - 1;
  if (hi - lo + 1 > kWin) return q.seq;
  for (int x = lo + (int)threadIdx.x; x <= hi; x += kThreads) buf[x - lo] = q.seq[x];
  return buf - lo;
}
// stage the mask rows [r_lo, r_hi] (clipped to the sequence) and return a pointer m with m[w] = ok[w] for their words
__device__ __forceinline__ const uint32_t* stage_bits(const Seq& q, int r_lo, int r_hi) {
  if (r_lo < 0) r_lo = 0;
  if (r_hi > q.L) r_hi = q.L;
  const int W1 = q.W + 1;
  const int w0 = (r_lo * W1) >> 5, w1 = (((r_hi + 1) * W1 + 31) >> 5) + 1;     // [w0, w1)
  const int wend = (int)((((long long)(q.L + 1) * W1) + 31) >> 5);
  for (int t = threadIdx.x; t < w1 - w0; t += kThreads) s_bits[t] = (w0 + t < wend) ? q.ok[w0 + t] : 0u;
  return s_bits - w0;
}
struct Mask {
  const uint32_t* m;
  int L, W;
  __device__ __forceinline__ bool ok(int i, int d) const {
    if (i < 0 || d < 0 || d > W || i + d > L) return false;
    const int c = i * (W + 1) + d;
    return (m[c >> 5] >> (c & 31)) & 1u;
  }
};
// walks the set bits bit0 + n, n in [lo, hi], ascending: next() returns n, or -1 at the end
struct Bits {
  const uint32_t* m;
  int bit0, e, w;
  uint32_t word;
  __device__ __forceinline__ void init(const uint32_t* mask, int b0, int lo, int hi) {
    m = mask; bit0 = b0; e = b0 + hi;
    const int b = b0 + (lo > 0 ? lo : 0);
    w = b >> 5;
    word = (hi >= lo && hi >= 0) ? (mask[w] & (~0u << (b & 31))) : 0u;
    if (hi < lo || hi < 0) e = -1;
  }
  __device__ __forceinline__ int next() {
    for (;;) {
      if (word) {
        const int bit = (w << 5) + __builtin_ctz(word);
        word &= word - 1;
        if (bit > e) { word = 0u; e = -1; return -1; }
        return bit - bit0;
      }
      if (((w + 1) << 5) > e) return -1;
      word = m[++w];
    }
  }
};

__global__ __launch_bounds__(kThreads) void k6_in(BppLinArgs a) {
  __shared__ BppLds sh;
  const Seq q = make_seq(a, blockIdx.y);
  const int d = a.d, tid = threadIdx.x;
  if (d > q.W) return;
  const int i0 = blockIdx.x * kC;
  if (i0 > q.L - d) return;
  const int W = q.W, L = q.L;
  const int nc = (kC < L - d - i0 + 1) ? kC : L - d - i0 + 1;
  const uint8_t* sq = stage_seq(q, sh.seq, i0 - 1, i0 + nc - 1 + d);
  const Mask mk{stage_bits(q, i0 - 1, i0 + nc - 1 + d), L, W};
  if (tid < kC) sh.dmin[tid] = (tid < nc) ? q.dmin[i0 + tid] : 0;
  __syncthreads();
  if (tid < 64) {      // the cells with interior loops (E cells whose closing pair is allowed)
    const int i = i0 + tid;
    const bool e = tid < nc && i > 0 && d + 2 <= W && mk.ok(i - 1, d + 2);
    const unsigned long long m = __ballot(e);
    if (e) sh.list[__popcll(m & ((1ull << tid) - 1ull))] = tid;
    if (tid == 0) sh.n_list = __popcll(m);
  }
  // rule 2, factorised: the stems (k, j) that end at j and start behind i + dmin[i]
  {
    const int ci = tid / kStem, ln = tid % kStem;
    double A = 0.;
    if (ci < nc) {
      const int i = i0 + ci, j = i + d, dmi = sh.dmin[ci];
      if (dmi > 0 && dmi < d) {
        const int smax = d - dmi;
        for (int sp0 = 1 + ln; sp0 <= smax; sp0 += kB * kStem) {
          double x1[kB], xp[kB], xw[kB];
          bool on[kB];
#pragma unroll
          for (int u = 0; u < kB; ++u) {
            const int sp = sp0 + u * kStem;
            on[u] = sp <= smax && mk.ok(j - sp, sp);
            x1[u] = xp[u] = xw[u] = 0.;
            if (on[u]) { x1[u] = q.in(BP_1, d - sp, i); xp[u] = q.in(BP_P, sp, j - sp); xw[u] = q.x(XW_ML, q.cell(j - sp, sp)); }
          }
#pragma unroll
          for (int u = 0; u < kB; ++u)
            if (on[u]) A = fma(x1[u], xp[u] * xw[u], A);
        }
      }
      sh.stem[ci][ln] = A;
    }
  }
  __syncthreads();
  // rule 6c, inside set: inner pairs (k, l), i <= k, l <= j, (k-i) + (j-l) <= C, (k,l) != (i,j); slots take the left ends k
  {
    const EnergyTables& xet = *a.xet;
    const int nE = sh.n_list, amax = (q.C < d - 2) ? q.C : d - 2;
    for (int w = tid; w < nE * kSlots; w += kThreads) {
      const int ci = sh.list[w / kSlots], slot = w % kSlots;
      const int i = i0 + ci, j = i + d;
      double HE = 0.;
      for (int e2 = 0; e2 <= 2 * (amax / (2 * kSlots)) + 1; ++e2) {
        const int da = (e2 >> 1) * 2 * kSlots + ((e2 & 1) ? 2 * kSlots - 1 - slot : slot);
        if (da > amax) continue;
        const int k = i + da;
        const int lmin = (k + 2 > j - (q.C - da)) ? k + 2 : j - (q.C - da);
        Bits it;
        it.init(mk.m, k * (W + 1), lmin - k, (j - k < W) ? j - k : W);
        for (;;) {
          int sp[kB];
          sp[0] = it.next();
          if (sp[0] < 0) break;
#pragma unroll
          for (int u = 1; u < kB; ++u) sp[u] = (sp[u - 1] < 0) ? -1 : it.next();
          double xw[kB], pv[kB];
#pragma unroll
          for (int u = 0; u < kB; ++u) {
            const bool on = sp[u] >= 0 && !(da == 0 && k + sp[u] == j);
            xw[u] = 0.; pv[u] = 0.;
            if (on) {
              xw[u] = a.no_ene ? 1. : loop_weight(xet, sq, i - 1, j, k, k + sp[u] - 1);
              pv[u] = q.in(BP_P, sp[u], k);
            }
          }
#pragma unroll
          for (int u = 0; u < kB; ++u) HE = fma(pv[u], xw[u], HE);
          if (sp[kB - 1] < 0) break;
        }
      }
      sh.loop[ci][slot] = HE;
    }
  }
  __syncthreads();
  if (tid >= nc) return;
  const int i = i0 + tid, j = i + d;
  const int dmi = sh.dmin[tid];
  auto left_ok = [&](int dd) { return dd <= W && dd >= 0 && i + dd <= L && dmi > 0 && dd >= dmi; };
  const bool pok = mk.ok(i, d), lok = left_ok(d), mok = q.m_ok(i, d, a.m_min);
  const bool eok = i > 0 && d + 2 <= W && mk.ok(i - 1, d + 2);
  double A = 0., HE = 0.;
#pragma unroll
  for (int k = 0; k < kStem; ++k) A += sh.stem[tid][k];
  if (eok)
#pragma unroll
    for (int k = 0; k < kSlots; ++k) HE += sh.loop[tid][k];
  if (dmi > 0 && dmi < d) A += q.in(BP_A, d - 1, i);      // the tail grows by the unpaired base j-1
  const int c = q.cell(i, d), c_up = eok ? q.cell(i - 1, d + 2) : c;
  double vP = 0.;
  if (pok && d >= 2) vP = fma(q.in(BP_P, d - 2, i + 1), q.x(XW_STACK, c), q.in(BP_E, d - 2, i + 1));   // rules 1b, 1a
  const double vB = lok ? A : 0.;
  const double s2 = (lok && left_ok(d - 1)) ? q.in(BP_2, d - 1, i) : 0.;                                  // rule 3a
  const double v2 = lok ? fma(vP, pok ? q.x(XW_ML, c) : 0., s2) : 0.;                                     // rule 3b
  const double v1 = lok ? v2 + vB : 0.;                                                                   // rules 4a, 4b
  const double sM = (mok && q.m_ok(i + 1, d - 1, a.m_min)) ? q.in(BP_M, d - 1, i + 1) : 0.;               // rule 5a
  const double vM = mok ? sM + vB : 0.;                                                                   // rule 5b
  const double vE = eok ? fma(vM, q.x(XW_CLOSE, c_up), q.x(XW_HP, c_up) + HE) : 0.;                       // rules 6a, 6b (L = 1), 6c
  q.in(BP_P, d, i) = vP; q.in(BP_E, d, i) = vE; q.in(BP_M, d, i) = vM; q.in(BP_B, d, i) = vB;
  q.in(BP_1, d, i) = v1; q.in(BP_2, d, i) = v2; q.in(BP_A, d, i) = A;
  (void)j;
}

// ---- exterior chains in log space, one wave per sequence: ln O(j) (rules 7, 8) and ln outO(i) - ln Z
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o, 64); v = (w > v) ? w : v; }
  return v;
}
__device__ __forceinline__ double wave_add(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_lse(double t) {   // log sum exp over the lanes (t = -inf: no term)
  const double m = wave_max(t);
  if (m == ELEMDP_NEG_INF) return m;
  const double s = wave_add(t == ELEMDP_NEG_INF ? 0. : exp(t - m));
  return m + log(s);
}
__global__ __launch_bounds__(64) void k6_in_ext(BppLinArgs a) {
  const Seq q = make_seq(a, blockIdx.x);
  const int lane = threadIdx.x, L = q.L, W = q.W;
  if (lane == 0) q.lo_in[0] = 0.;
  __syncthreads();
  for (int j = 1; j <= L; ++j) {
    double t = (lane == 0) ? q.lo_in[j - 1] : ELEMDP_NEG_INF;    // rule 8
    for (int sp = 1 + lane; sp <= W && sp <= j; sp += 64) {      // rule 7: pairs (j - sp, j)
      const int i = j - sp;
      if (!q.pair_ok(i, sp)) continue;
      const double w = q.in(BP_P, sp, i) * q.x(XW_EXT, q.cell(i, sp));
      if (w > 0.) {
        const double u = q.lo_in[i] + log(w);
        t = (t == ELEMDP_NEG_INF) ? u : (t > u ? t + log1p(exp(u - t)) : u + log1p(exp(t - u)));
      }
    }
    const double r = wave_lse(t);
    if (lane == 0) q.lo_in[j] = r;
    __syncthreads();
  }
}
__global__ __launch_bounds__(64) void k6_out_ext(BppLinArgs a) {
  const Seq q = make_seq(a, blockIdx.x);
  const int lane = threadIdx.x, L = q.L, W = q.W;
  const double lz = q.lo_in[L];
  if (lane == 0) q.lo_out[L] = -lz;          // terminal O(L): outside 1, divided by Z
  __syncthreads();
  for (int i = L - 1; i >= 0; --i) {
    double t = (lane == 0) ? q.lo_out[i + 1] : ELEMDP_NEG_INF;
    for (int sp = 1 + lane; sp <= W && i + sp <= L; sp += 64) {
      if (!q.pair_ok(i, sp)) continue;
      const double w = q.in(BP_P, sp, i) * q.x(XW_EXT, q.cell(i, sp));
      if (w > 0.) {
        const double u = q.lo_out[i + sp] + log(w);
        t = (t == ELEMDP_NEG_INF) ? u : (t > u ? t + log1p(exp(u - t)) : u + log1p(exp(t - u)));
      }
    }
    const double r = wave_lse(t);
    if (lane == 0) q.lo_out[i] = r;
    __syncthreads();
  }
}

// ---- outside, diagonal d (values divided by Z)
__global__ __launch_bounds__(kThreads) void k6_out(BppLinArgs a) {
  __shared__ BppLds sh;
  const Seq q = make_seq(a, blockIdx.y);
  const int d = a.d, tid = threadIdx.x;
  if (d > q.W) return;
  const int i0 = blockIdx.x * kC;
  if (i0 > q.L - d) return;
  const int W = q.W, L = q.L;
  const int nc = (kC < L - d - i0 + 1) ? kC : L - d - i0 + 1;
  const uint8_t* sq = stage_seq(q, sh.seq, i0 - q.C - 2, i0 + nc + W);
  const int Cc = (q.C < kMaxLoop) ? q.C : kMaxLoop;
  const Mask mk{stage_bits(q, i0 - Cc - 2, i0 + nc - 1 + d), L, W};
  if (tid < kC) sh.dmin[tid] = (tid < nc) ? q.dmin[i0 + tid] : 0;
  __syncthreads();
  if (tid < 64) {      // the stems P(i,j) that occur: they collect the interior loops around them
    const int i = i0 + tid;
    const bool e = tid < nc && mk.ok(i, d) && q.in(BP_P, d, i) != 0.;
    const unsigned long long m = __ballot(e);
    if (e) sh.list[__popcll(m & ((1ull << tid) - 1ull))] = tid;
    if (tid == 0) sh.n_list = __popcll(m);
  }
  {
    const int ci = tid / kStem, ln = tid % kStem;
    if (ci < nc) {
      const int i = i0 + ci, j = i + d, dmi = sh.dmin[ci];
      const bool lok = d <= W && i + d <= L && dmi > 0 && d >= dmi;
      const bool pok = mk.ok(i, d);
      const double in1 = lok ? q.in(BP_1, d, i) : 0.;
      // H1: 1(i,j) under B(i,l) through a stem (j, l) that starts at j;  HA: what reaches 2(i,j) through rule 2 (taken by
      // the stem P(i,j) only) -- one round of loads for four candidates of each
      double H1 = 0., HA = 0.;
      const int hi = (in1 != 0.) ? ((W - d < L - j) ? W - d : L - j) : 0;
      const int bmax = pok ? ((W - d < i) ? W - d : i) : 0;
      const int nmax = (hi > bmax) ? hi : bmax;
      for (int n0 = 1 + ln; n0 <= nmax; n0 += kB * kStem) {
        double oa[kB], xp[kB], xw[kB], ob[kB], x1[kB];
        bool on[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const int n = n0 + u * kStem;
          on[u] = n <= hi && mk.ok(j, n);
          oa[u] = xp[u] = xw[u] = ob[u] = x1[u] = 0.;
          if (on[u]) { oa[u] = q.out(BO_A, d + n, i); xp[u] = q.in(BP_P, n, j); xw[u] = q.x(XW_ML, q.cell(j, n)); }
          if (n <= bmax) { ob[u] = q.out(BO_A, d + n, i - n); x1[u] = q.in(BP_1, n, i - n); }
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          if (on[u]) H1 = fma(oa[u], xp[u] * xw[u], H1);
          if (n0 + u * kStem <= bmax) HA = fma(ob[u], x1[u], HA);
        }
      }
      sh.stem[ci][ln] = H1;
      sh.stem2[ci][ln] = HA;
    }
  }
  __syncthreads();
  // HP: the interior loops around the stem; slots take the left ends of the outer cells
  {
    const EnergyTables& xet = *a.xet;
    const int nP = sh.n_list;
    for (int w = tid; w < nP * kSlots; w += kThreads) {
      const int ci = sh.list[w / kSlots], slot = w % kSlots;
      const int i = i0 + ci, j = i + d;
      const int amax = (Cc < i - 1) ? Cc : i - 1;        // outside set: k - i' <= C (no loop beyond kMaxLoop); closing pair starts at i' - 1 >= 0
      double HP = 0.;
      for (int e2 = 0; e2 <= 2 * (amax / (2 * kSlots)) + 1; ++e2) {
        const int da = (e2 >> 1) * 2 * kSlots + ((e2 & 1) ? 2 * kSlots - 1 - slot : slot);
        if (da > amax || da < 0) continue;
        const int io = i - da;                            // outer E cell (io, jo), closing pair cell (io - 1, jo - io + 2)
        // (the reference's outside set does not bound jo - j; a loop of more than kMaxLoop unpaired bases weighs
        // log 0, so the walk stops at jo - j = kMaxLoop - da)
        int hi = (W < L - io + 1) ? W : L - io + 1;
        if (!a.no_ene && hi > j - io + 2 + kMaxLoop - da) hi = j - io + 2 + kMaxLoop - da;
        Bits it;
        it.init(mk.m, (io - 1) * (W + 1), j - io + 2, hi);
        for (;;) {
          int sp[kB];
          sp[0] = it.next();
          if (sp[0] < 0) break;
#pragma unroll
          for (int u = 1; u < kB; ++u) sp[u] = (sp[u - 1] < 0) ? -1 : it.next();
          double xw[kB], ov[kB];
#pragma unroll
          for (int u = 0; u < kB; ++u) {
            const int jo = io + sp[u] - 2;
            const bool on = sp[u] >= 0 && !(da == 0 && jo == j);
            xw[u] = 0.; ov[u] = 0.;
            if (on) {
              xw[u] = a.no_ene ? 1. : loop_weight(xet, sq, io - 1, jo, i, j - 1);
              ov[u] = q.out(BO_E, jo - io, io);
            }
          }
#pragma unroll
          for (int u = 0; u < kB; ++u) HP = fma(ov[u], xw[u], HP);
          if (sp[kB - 1] < 0) break;
        }
      }
      sh.loop[ci][slot] = HP;
    }
  }
  __syncthreads();
  if (tid >= nc) return;
  const int i = i0 + tid, j = i + d;
  const int dmi = sh.dmin[tid];
  auto left_ok = [&](int dd) { return dd <= W && dd >= 0 && i + dd <= L && dmi > 0 && dd >= dmi; };
  const bool pok = mk.ok(i, d), lok = left_ok(d), mok = q.m_ok(i, d, a.m_min);
  const bool up_ok = i > 0 && d + 2 <= W && mk.ok(i - 1, d + 2);
  const double inP = q.in(BP_P, d, i), inA = q.in(BP_A, d, i), in1 = q.in(BP_1, d, i);
  double H1 = 0., HA = 0., HP = 0.;
  if (lok && in1 != 0.)
#pragma unroll
    for (int k = 0; k < kStem; ++k) H1 += sh.stem[tid][k];
  if (pok && inP != 0.) {
#pragma unroll
    for (int k = 0; k < kStem; ++k) HA += sh.stem2[tid][k];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) HP += sh.loop[tid][k];
  }
  const double inE = q.in(BP_E, d, i), inM = q.in(BP_M, d, i), inB = q.in(BP_B, d, i), in2 = q.in(BP_2, d, i);
  const int c = q.cell(i, d), c_up = up_ok ? q.cell(i - 1, d + 2) : c;
  const double opP = up_ok ? q.out(BO_P, d + 2, i - 1) : 0.;
  const double oE = (up_ok && inE != 0.) ? opP : 0.;                                                        // rule 1a
  const double oP1b = (up_ok && pok && inP != 0.) ? opP * q.x(XW_STACK, c_up) : 0.;                          // rule 1b
  const bool doM = mok && q.m_ok(i - 1, d + 1, a.m_min);
  const double sM = (doM && inM != 0.) ? q.out(BO_M, d + 1, i - 1) : 0.;                                     // rule 5a
  const double oM = (inM != 0.) ? fma(oE, up_ok ? q.x(XW_CLOSE, c_up) : 0., sM) : 0.;                        // rule 6a
  const double o1 = (in1 != 0.) ? H1 : 0.;
  const double oB = (inB != 0.) ? (mok ? oM : 0.) + o1 : 0.;                                                 // rules 5b, 4b
  const bool do2 = lok && left_ok(d + 1) && j < L;
  const double s2 = (do2 && in2 != 0.) ? q.out(BO_2, d + 1, i) : 0.;                                         // rule 3a
  const double o2 = (in2 != 0.) ? o1 + s2 : 0.;                                                              // rule 4a (direct part)
  double oP = 0.;
  if (inP != 0.) {
    const double xe = pok ? q.x(XW_EXT, c) : 0.;
    const double r7 = (xe != 0.) ? exp(q.lo_in[i] + q.lo_out[j]) * xe : 0.;                                  // rule 7 (lo_out holds - ln Z)
    oP = r7 + oP1b + (o2 + HA) * (pok ? q.x(XW_ML, c) : 0.) + HP;                                            // rules 3b, 6c
  }
  double oA = 0.;
  if (inA != 0.) oA = (lok ? oB : 0.) + ((d + 1 <= W && j < L) ? q.out(BO_A, d + 1, i) : 0.);
  q.out(BO_P, d, i) = oP; q.out(BO_E, d, i) = oE; q.out(BO_M, d, i) = oM; q.out(BO_2, d, i) = o2; q.out(BO_A, d, i) = oA;
}

// ---- round 4: the interior loops of rule 6c from the candidate table (kernels.h: BppCandTable) instead of a walk over the
// pair mask.  Lane = (cell that has loops, part): the parts of a cell deal the entries of a class among themselves, the cells
// of a part are neighbours on the diagonal, so a wave's loads are row segments and every lane runs the same instructions; a
// candidate that is no pair reads a zero from the plane (every cell of a plane is written).  The eight shapes that do not
// factorise go through loop_weight, one per part.  Same sums as k6_in / k6_out in another (fixed) order.
constexpr int kParts = 32;    // most parts per cell (parts = 256 / cells with loops, at least 8)
struct BppLdsT {
  double stem[kC][kStem + 1];
  double stem2[kC][kStem + 1];
  double loop[kC][kParts + 1];
  BppCand cand[kBppCandMax];
  int list[kC];
  int n_list;
  int dmin[kC];
  uint8_t seq[kWin];
};
__constant__ int8_t kSpecialU1[kBppSpecial] = {0, 1, 1, 1, 2, 2, 2, 3};
__constant__ int8_t kSpecialU2[kBppSpecial] = {1, 0, 1, 2, 1, 2, 3, 2};
static_assert(kBppSpecialU1[0] == 0 && kBppSpecialU1[1] == 1 && kBppSpecialU1[2] == 1 && kBppSpecialU1[3] == 1 && kBppSpecialU1[4] == 2 &&
              kBppSpecialU1[5] == 2 && kBppSpecialU1[6] == 2 && kBppSpecialU1[7] == 3 && kBppSpecialU2[0] == 1 && kBppSpecialU2[1] == 0 &&
              kBppSpecialU2[2] == 1 && kBppSpecialU2[3] == 2 && kBppSpecialU2[4] == 1 && kBppSpecialU2[5] == 2 && kBppSpecialU2[6] == 3 &&
              kBppSpecialU2[7] == 2, "the special shapes of bpp_cand.h (which the CPU check walks)");

// stages the entries with u1 + u2 <= tmax of every class; n[c] = their number, off[c] = where the class starts in sh.cand
__device__ __forceinline__ void stage_cand(const BppCandTable* __restrict__ tab, BppLdsT& sh, int tmax, int n[BC_CLASSES], int off[BC_CLASSES]) {
  int tot = 0;
#pragma unroll
  for (int c = 0; c < BC_CLASSES; ++c) { n[c] = (tmax >= 0) ? tab->upto[c][tmax] : 0; off[c] = tot; tot += n[c]; }
  for (int t = threadIdx.x; t < tot; t += kThreads) {
    const int c = (t < off[1]) ? 0 : (t < off[2]) ? 1 : 2;
    sh.cand[t] = tab->e[tab->base[c] + (t - off[c])];
  }
}

__global__ __launch_bounds__(kThreads) void k6_in_tab(BppLinArgs a) {
  __shared__ BppLdsT sh;
  const Seq q = make_seq(a, blockIdx.y);
  const int d = a.d, tid = threadIdx.x;
  if (d > q.W) return;
  const int i0 = blockIdx.x * kC;
  if (i0 > q.L - d) return;
  const int W = q.W, L = q.L;
  const int nc = (kC < L - d - i0 + 1) ? kC : L - d - i0 + 1;
  const uint8_t* sq = stage_seq(q, sh.seq, i0 - 1, i0 + nc - 1 + d);
  const Mask mk{stage_bits(q, i0 - 1, i0 + nc - 1 + d), L, W};
  if (tid < kC) sh.dmin[tid] = (tid < nc) ? q.dmin[i0 + tid] : 0;
  const int Cc = (q.C < kMaxLoop) ? q.C : kMaxLoop;
  const int tmax = (Cc < d - 2) ? Cc : d - 2;          // inside set: (k - i) + (j - l) <= C, inner span >= 2
  int cn[BC_CLASSES], co[BC_CLASSES];
  stage_cand(a.cand, sh, tmax, cn, co);
  __syncthreads();
  if (tid < 64) {      // the cells with interior loops (E cells whose closing pair is allowed)
    const int i = i0 + tid;
    const bool e = tid < nc && i > 0 && d + 2 <= W && mk.ok(i - 1, d + 2);
    const unsigned long long m = __ballot(e);
    if (e) sh.list[__popcll(m & ((1ull << tid) - 1ull))] = tid;
    if (tid == 0) sh.n_list = __popcll(m);
  }
  // rule 2, factorised: the stems (k, j) that end at j and start behind i + dmin[i]
  {
    const int ci = tid / kStem, ln = tid % kStem;
    double A = 0.;
    if (ci < nc) {
      const int i = i0 + ci, j = i + d, dmi = sh.dmin[ci];
      if (dmi > 0 && dmi < d) {
        const int smax = d - dmi;
        for (int sp0 = 1 + ln; sp0 <= smax; sp0 += kB * kStem) {
          double x1[kB], xp[kB];
          bool on[kB];
#pragma unroll
          for (int u = 0; u < kB; ++u) {
            const int sp = sp0 + u * kStem;
            on[u] = sp <= smax && mk.ok(j - sp, sp);
            x1[u] = xp[u] = 0.;
            if (on[u]) { x1[u] = q.in(BP_1, d - sp, i); xp[u] = q.in(BP_PM, sp, j - sp); }
          }
#pragma unroll
          for (int u = 0; u < kB; ++u)
            if (on[u]) A = fma(x1[u], xp[u], A);
        }
      }
      sh.stem[ci][ln] = A;
    }
  }
  __syncthreads();
  const int nE = sh.n_list;
  const int parts = (nE > 0) ? ((kThreads / nE < kParts) ? kThreads / nE : kParts) : 1;
  if (tmax >= 1 && tid < nE * parts) {
    const EnergyTables& xet = *a.xet;
    const int part = tid / nE, ci = sh.list[tid - part * nE];
    const int i = i0 + ci, j = i + d;
    // factors of the closing pair (i-1, j)
    const int type = bp_type(sq[i - 1], sq[j]);
    const int mi = type * 25 + sq[i] * 5 + sq[j - 1];
    const double fac[BC_CLASSES] = {xet.mismatch_i[mi], xet.mismatch_1ni[mi], is_au(type) ? xet.term_au : 1.};
    const size_t row = (size_t)(L + 1);
    double HE = 0.;
#pragma unroll
    for (int c = 0; c < BC_CLASSES; ++c) {
      const double* __restrict__ pl = q.tin + (size_t)(BP_X + c) * q.t_stride + (size_t)d * row + i;
      const int end = co[c] + cn[c];
      double acc = 0.;
      for (int t0 = co[c] + part; t0 < end; t0 += kB * parts) {
        double cf[kB], pv[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const int t = t0 + u * parts;
          cf[u] = 0.; pv[u] = 0.;
          if (t < end) {
            const BppCand e = sh.cand[t];
            cf[u] = e.coef;
            pv[u] = pl[(ptrdiff_t)e.u1 - (ptrdiff_t)e.T * (ptrdiff_t)row];        // plane[(d - T) * (L+1) + i + u1]
          }
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) acc = fma(cf[u], pv[u], acc);
      }
      HE = fma(fac[c], acc, HE);
    }
    if (part < kBppSpecial) {
      const int u1 = kSpecialU1[part], u2 = kSpecialU2[part];
      const int k = i + u1, sp = d - u1 - u2;
      if (u1 + u2 <= tmax && mk.ok(k, sp)) HE = fma(q.in(BP_P, sp, k), loop_weight(xet, sq, i - 1, j, k, k + sp - 1), HE);
    }
    sh.loop[ci][part] = HE;
  }
  __syncthreads();
  if (tid >= nc) return;
  const int i = i0 + tid, j = i + d;
  const int dmi = sh.dmin[tid];
  auto left_ok = [&](int dd) { return dd <= W && dd >= 0 && i + dd <= L && dmi > 0 && dd >= dmi; };
  const bool pok = mk.ok(i, d), lok = left_ok(d), mok = q.m_ok(i, d, a.m_min);
  const bool eok = i > 0 && d + 2 <= W && mk.ok(i - 1, d + 2);
  double A = 0., HE = 0.;
#pragma unroll
  for (int k = 0; k < kStem; ++k) A += sh.stem[tid][k];
  if (eok && tmax >= 1)
    for (int k = 0; k < parts; ++k) HE += sh.loop[tid][k];
  if (dmi > 0 && dmi < d) A += q.in(BP_A, d - 1, i);      // the tail grows by the unpaired base j-1
  const int c = q.cell(i, d), c_up = eok ? q.cell(i - 1, d + 2) : c;
  double vP = 0.;
  if (pok && d >= 2) vP = fma(q.in(BP_P, d - 2, i + 1), q.x(XW_STACK, c), q.in(BP_E, d - 2, i + 1));   // rules 1b, 1a
  const double vB = lok ? A : 0.;
  const double s2 = (lok && left_ok(d - 1)) ? q.in(BP_2, d - 1, i) : 0.;                                  // rule 3a
  const double v2 = lok ? fma(vP, pok ? q.x(XW_ML, c) : 0., s2) : 0.;                                     // rule 3b
  const double v1 = lok ? v2 + vB : 0.;                                                                   // rules 4a, 4b
  const double sM = (mok && q.m_ok(i + 1, d - 1, a.m_min)) ? q.in(BP_M, d - 1, i + 1) : 0.;               // rule 5a
  const double vM = mok ? sM + vB : 0.;                                                                   // rule 5b
  const double vE = eok ? fma(vM, q.x(XW_CLOSE, c_up), q.x(XW_HP, c_up) + HE) : 0.;                       // rules 6a, 6b (L = 1), 6c
  q.in(BP_P, d, i) = vP; q.in(BP_E, d, i) = vE; q.in(BP_M, d, i) = vM; q.in(BP_B, d, i) = vB;
  q.in(BP_1, d, i) = v1; q.in(BP_2, d, i) = v2; q.in(BP_A, d, i) = A;
  // P(i,j) as the INNER pair (i, j-1) of a loop: times its factor of every class (bases i-1 and j lie inside that loop)
  double xI = 0., xN = 0., xB = 0.;
  if (vP != 0.) {
    const EnergyTables& xet = *a.xet;
    const int type2 = bp_type(sq[j - 1], sq[i]);
    xB = is_au(type2) ? vP * xet.term_au : vP;
    if (i > 0 && j < L) {
      const int mi = type2 * 25 + sq[j] * 5 + sq[i - 1];
      xI = vP * xet.mismatch_i[mi];
      xN = vP * xet.mismatch_1ni[mi];
    }
  }
  q.in(BP_X + BC_I, d, i) = xI; q.in(BP_X + BC_N, d, i) = xN; q.in(BP_X + BC_B, d, i) = xB;
  q.in(BP_PM, d, i) = (pok && vP != 0.) ? vP * q.x(XW_ML, c) : 0.;
}

__global__ __launch_bounds__(kThreads) void k6_out_tab(BppLinArgs a) {
  __shared__ BppLdsT sh;
  const Seq q = make_seq(a, blockIdx.y);
  const int d = a.d, tid = threadIdx.x;
  if (d > q.W) return;
  const int i0 = blockIdx.x * kC;
  if (i0 > q.L - d) return;
  const int W = q.W, L = q.L;
  const int nc = (kC < L - d - i0 + 1) ? kC : L - d - i0 + 1;
  const uint8_t* sq = stage_seq(q, sh.seq, i0 - q.C - 2, i0 + nc + W);
  const int Cc = (q.C < kMaxLoop) ? q.C : kMaxLoop;
  const Mask mk{stage_bits(q, i0 - Cc - 2, i0 + nc - 1 + d), L, W};
  if (tid < kC) sh.dmin[tid] = (tid < nc) ? q.dmin[i0 + tid] : 0;
  const int tmax = (kMaxLoop < W - 2 - d) ? kMaxLoop : W - 2 - d;     // outside set: the closing pair spans at most W
  int cn[BC_CLASSES], co[BC_CLASSES];
  stage_cand(a.cand, sh, tmax, cn, co);
  __syncthreads();
  if (tid < 64) {      // the stems P(i,j) that occur: they collect the interior loops around them
    const int i = i0 + tid;
    const bool e = tid < nc && mk.ok(i, d) && q.in(BP_P, d, i) != 0.;
    const unsigned long long m = __ballot(e);
    if (e) sh.list[__popcll(m & ((1ull << tid) - 1ull))] = tid;
    if (tid == 0) sh.n_list = __popcll(m);
  }
  {
    const int ci = tid / kStem, ln = tid % kStem;
    if (ci < nc) {
      const int i = i0 + ci, j = i + d, dmi = sh.dmin[ci];
      const bool lok = d <= W && i + d <= L && dmi > 0 && d >= dmi;
      const bool pok = mk.ok(i, d);
      const double in1 = lok ? q.in(BP_1, d, i) : 0.;
      double H1 = 0., HA = 0.;
      const int hi = (in1 != 0.) ? ((W - d < L - j) ? W - d : L - j) : 0;
      const int bmax = pok ? ((W - d < i) ? W - d : i) : 0;
      const int nmax = (hi > bmax) ? hi : bmax;
      for (int n0 = 1 + ln; n0 <= nmax; n0 += kB * kStem) {
        double oa[kB], xp[kB], ob[kB], x1[kB];
        bool on[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const int n = n0 + u * kStem;
          on[u] = n <= hi && mk.ok(j, n);
          oa[u] = xp[u] = ob[u] = x1[u] = 0.;
          if (on[u]) { oa[u] = q.out(BO_A, d + n, i); xp[u] = q.in(BP_PM, n, j); }
          if (n <= bmax) { ob[u] = q.out(BO_A, d + n, i - n); x1[u] = q.in(BP_1, n, i - n); }
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          if (on[u]) H1 = fma(oa[u], xp[u], H1);
          if (n0 + u * kStem <= bmax) HA = fma(ob[u], x1[u], HA);
        }
      }
      sh.stem[ci][ln] = H1;
      sh.stem2[ci][ln] = HA;
    }
  }
  __syncthreads();
  // HP: the interior loops around the stem (i, j-1): outer E cells (i - u1, d + T), closing pair (i - u1 - 1, j + u2)
  const int nP = sh.n_list;
  const int parts = (nP > 0) ? ((kThreads / nP < kParts) ? kThreads / nP : kParts) : 1;
  if (tmax >= 1 && tid < nP * parts) {
    const EnergyTables& xet = *a.xet;
    const int part = tid / nP, ci = sh.list[tid - part * nP];
    const int i = i0 + ci, j = i + d;
    const int amax = (Cc < i - 1) ? Cc : i - 1;          // k - i' <= C; the closing pair starts at i' - 1 >= 0
    const int rmax = L - 1 - j;                          // .. and ends at j + u2 <= L - 1
    // factors of the inner pair (i, j-1)
    double fac[BC_CLASSES] = {0., 0., 0.};
    {
      const int type2 = bp_type(sq[j - 1], sq[i]);
      fac[BC_B] = is_au(type2) ? xet.term_au : 1.;
      if (i > 0 && j < L) {
        const int mi = type2 * 25 + sq[j] * 5 + sq[i - 1];
        fac[BC_I] = xet.mismatch_i[mi];
        fac[BC_N] = xet.mismatch_1ni[mi];
      }
    }
    const size_t row = (size_t)(L + 1);
    double HP = 0.;
#pragma unroll
    for (int c = 0; c < BC_CLASSES; ++c) {
      const double* __restrict__ pl = q.tout + (size_t)(BO_X + c) * q.t_stride + (size_t)d * row + i;
      const int end = co[c] + cn[c];
      double acc = 0.;
      for (int t0 = co[c] + part; t0 < end; t0 += kB * parts) {
        double cf[kB], ov[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const int t = t0 + u * parts;
          cf[u] = 0.; ov[u] = 0.;
          if (t < end) {
            const BppCand e = sh.cand[t];
            if (e.u1 <= amax && e.T - e.u1 <= rmax) {
              cf[u] = e.coef;
              ov[u] = pl[(ptrdiff_t)e.T * (ptrdiff_t)row - (ptrdiff_t)e.u1];      // plane[(d + T) * (L+1) + i - u1]
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) acc = fma(cf[u], ov[u], acc);
      }
      HP = fma(fac[c], acc, HP);
    }
    if (part < kBppSpecial) {
      const int u1 = kSpecialU1[part], u2 = kSpecialU2[part];
      const int io = i - u1, jo = j + u2;
      if (u1 + u2 <= tmax && u1 <= amax && u2 <= rmax && mk.ok(io - 1, jo - io + 2))
        HP = fma(q.out(BO_E, jo - io, io), loop_weight(xet, sq, io - 1, jo, i, j - 1), HP);
    }
    sh.loop[ci][part] = HP;
  }
  __syncthreads();
  if (tid >= nc) return;
  const int i = i0 + tid, j = i + d;
  const int dmi = sh.dmin[tid];
  auto left_ok = [&](int dd) { return dd <= W && dd >= 0 && i + dd <= L && dmi > 0 && dd >= dmi; };
  const bool pok = mk.ok(i, d), lok = left_ok(d), mok = q.m_ok(i, d, a.m_min);
  const bool up_ok = i > 0 && d + 2 <= W && mk.ok(i - 1, d + 2);
  const double inP = q.in(BP_P, d, i), inA = q.in(BP_A, d, i), in1 = q.in(BP_1, d, i);
  double H1 = 0., HA = 0., HP = 0.;
  if (lok && in1 != 0.)
#pragma unroll
    for (int k = 0; k < kStem; ++k) H1 += sh.stem[tid][k];
  if (pok && inP != 0.) {
#pragma unroll
    for (int k = 0; k < kStem; ++k) HA += sh.stem2[tid][k];
    if (tmax >= 1)
      for (int k = 0; k < parts; ++k) HP += sh.loop[tid][k];
  }
  const double inE = q.in(BP_E, d, i), inM = q.in(BP_M, d, i), inB = q.in(BP_B, d, i), in2 = q.in(BP_2, d, i);
  const int c = q.cell(i, d), c_up = up_ok ? q.cell(i - 1, d + 2) : c;
  const double opP = up_ok ? q.out(BO_P, d + 2, i - 1) : 0.;
  const double oE = (up_ok && inE != 0.) ? opP : 0.;                                                        // rule 1a
  const double oP1b = (up_ok && pok && inP != 0.) ? opP * q.x(XW_STACK, c_up) : 0.;                          // rule 1b
  const bool doM = mok && q.m_ok(i - 1, d + 1, a.m_min);
  const double sM = (doM && inM != 0.) ? q.out(BO_M, d + 1, i - 1) : 0.;                                     // rule 5a
  const double oM = (inM != 0.) ? fma(oE, up_ok ? q.x(XW_CLOSE, c_up) : 0., sM) : 0.;                        // rule 6a
  const double o1 = (in1 != 0.) ? H1 : 0.;
  const double oB = (inB != 0.) ? (mok ? oM : 0.) + o1 : 0.;                                                 // rules 5b, 4b
  const bool do2 = lok && left_ok(d + 1) && j < L;
  const double s2 = (do2 && in2 != 0.) ? q.out(BO_2, d + 1, i) : 0.;                                         // rule 3a
  const double o2 = (in2 != 0.) ? o1 + s2 : 0.;                                                              // rule 4a (direct part)
  double oP = 0.;
  if (inP != 0.) {
    const double xe = pok ? q.x(XW_EXT, c) : 0.;
    const double r7 = (xe != 0.) ? exp(q.lo_in[i] + q.lo_out[j]) * xe : 0.;                                  // rule 7 (lo_out holds - ln Z)
    oP = r7 + oP1b + (o2 + HA) * (pok ? q.x(XW_ML, c) : 0.) + HP;                                            // rules 3b, 6c
  }
  double oA = 0.;
  if (inA != 0.) oA = (lok ? oB : 0.) + ((d + 1 <= W && j < L) ? q.out(BO_A, d + 1, i) : 0.);
  q.out(BO_P, d, i) = oP; q.out(BO_E, d, i) = oE; q.out(BO_M, d, i) = oM; q.out(BO_2, d, i) = o2; q.out(BO_A, d, i) = oA;
  // E(i,j) as the region under the CLOSING pair (i-1, j) of a loop: times that pair's factor of every class
  double xI = 0., xN = 0., xB = 0.;
  if (oE != 0.) {
    const EnergyTables& xet = *a.xet;
    const int type = bp_type(sq[i - 1], sq[j]);
    const int mi = type * 25 + sq[i] * 5 + sq[(d > 0) ? j - 1 : j];
    xI = oE * xet.mismatch_i[mi];
    xN = oE * xet.mismatch_1ni[mi];
    xB = is_au(type) ? oE * xet.term_au : oE;
  }
  q.out(BO_X + BC_I, d, i) = xI; q.out(BO_X + BC_N, d, i) = xN; q.out(BO_X + BC_B, d, i) = xB;
}

// ---- round 4: one workgroup per SEQUENCE sweeps all diagonals of a direction in one launch (k6_in_seq / k6_out_seq).  The
// diagonal kernels above pay, per block of 32 cells and diagonal, a plan record, the staging of mask rows / bases / candidate
// table and three barriers for a few hundred loads -- 13 dependent round trips per workgroup, 102 launches per chunk; here the
// mask, the bases, the first-pair spans, the candidate table and the pairs of every diagonal (build_pair_lists) are staged ONCE, the
// 512 threads take all cells of a diagonal together (stems: (cell, lane) items; loops: (pair of the diagonal, part) items), and
// a diagonal costs two barriers.  A diagonal's values reach the next one through global memory: written and read by the same
// workgroup, whose waves share the CU's vector L1 (write-through), with the barrier's workgroup-scope fence in between.
// Same sums as the _tab kernels in the same order per cell (parts differ: the partial sums are added in another fixed order).
#ifndef ELEMDP_SEQ_THREADS
#define ELEMDP_SEQ_THREADS 512
#endif
constexpr int kSeqThreads = ELEMDP_SEQ_THREADS;
#ifndef ELEMDP_SEQ_WAVES_IN
#define ELEMDP_SEQ_WAVES_IN 6   // waves per SIMD the per-sequence kernels are compiled for (6: three workgroups of 512 per CU, 80 registers)
#endif
#ifndef ELEMDP_SEQ_WAVES_OUT
#define ELEMDP_SEQ_WAVES_OUT 4
#endif
#ifndef ELEMDP_SEQ_BATCH
#define ELEMDP_SEQ_BATCH 8
#endif
constexpr int kBL = ELEMDP_SEQ_BATCH;   // candidates per batch of loads in the loop sums
#ifndef ELEMDP_SEQ_QUADS
#define ELEMDP_SEQ_QUADS 2
#endif
constexpr int kBQ = ELEMDP_SEQ_QUADS;   // quads of the generic class per batch of loads
#ifndef ELEMDP_SEQ_KO
#define ELEMDP_SEQ_KO 0         // timing experiments: 1 no class loops, 2 no special shapes, 4 no stems (inside sweep)
#endif
#ifndef ELEMDP_SEQ_STEM_BATCH
#define ELEMDP_SEQ_STEM_BATCH 6
#endif
constexpr int kBS = ELEMDP_SEQ_STEM_BATCH;   // stems per batch of loads (inside sweep)
constexpr int kSeqStem = 2;      // most lanes per cell for the stem sums
// a candidate as the per-sequence kernels stage it: the offset of its plane entry from the lane's base pointer is formed once per
// sequence (inside sweep: (kMaxLoop - T) * (L+1) + u1 from the row kMaxLoop diagonals below the cell's; outside: T * (L+1) - u1),
// so a candidate costs an LDS read, one 64-bit shift-add, the load and the fma
struct SeqCand { double coef; uint32_t off; int16_t u1, u2; };
static_assert(sizeof(SeqCand) == 16, "one ds_read_b128 per candidate");
struct __attribute__((packed, aligned(8))) Dbl2 { double x, y; };     // two neighbours of a plane row: one 16-byte load
struct SeqLdsLayout { int cand, runc, runo, qoff, part, stem, stem2, dmin, slot, plist, poff, seq, bits, total; };
__host__ __device__ inline SeqLdsLayout seq_lds_layout(int lmax, int wmax, int pmax, bool out) {
  SeqLdsLayout y;
  int o = 0;
  auto take = [&](int bytes) { const int at = o; o += (bytes + 15) & ~15; return at; };
  const int ncm = lmax + 1;
  y.cand = take((int)sizeof(SeqCand) * kBppCandMax);
  y.runc = take(8 * kBppRunMax);
  y.runo = take(4 * (kMaxLoop + 2));
  y.qoff = take(4 * (kBppRunMax / 4));
  y.part = take(8 * ((kSeqThreads > ncm ? kSeqThreads : ncm) + kParts));
  y.stem = take(8 * kSeqStem * ncm);
  y.stem2 = out ? take(8 * kSeqStem * ncm) : y.stem;
  y.dmin = take(2 * ncm);
  y.slot = take(2 * ncm);
  y.plist = take(2 * (pmax + 1));
  y.poff = take(4 * (wmax + 3));
  y.seq = take(lmax + 1);
  y.bits = take(4 * (((ncm * (wmax + 1) + 31) >> 5) + 2));
  y.total = o;
  return y;
}

// pairs of every diagonal, ascending in i: plist[poff[d] .. poff[d+1]) = the i with (i, d) in the mask (block 0 of k6_terms)
__device__ void build_pair_lists(const Seq& q, int16_t* plist, int32_t* poff, int* cnt /* LDS: W + 2 ints */) {
  const int L = q.L, W = q.W, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = kThreads / 64;
  for (int d = wv; d <= W; d += nw) {
    int n = 0;
    for (int i0 = 0; i0 + d <= L; i0 += 64) n += __popcll(__ballot(q.pair_ok(i0 + lane, d)));
    if (lane == 0) cnt[d] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int d = 0; d <= W; ++d) { const int c = cnt[d]; cnt[d] = run; poff[d] = run; run += c; }
    poff[W + 1] = run;
  }
  __syncthreads();
  for (int d = wv; d <= W; d += nw) {
    int at = cnt[d];
    for (int i0 = 0; i0 + d <= L; i0 += 64) {
      const bool e = q.pair_ok(i0 + lane, d);
      const unsigned long long m = __ballot(e);
      if (e) plist[at + __popcll(m & ((1ull << lane) - 1ull))] = (int16_t)(i0 + lane);
      at += __popcll(m);
    }
  }
}

struct SeqCtx {
  SeqCand* cand; double* runc; int32_t* runo; uint32_t* qoff; double* part; double* stem; double* stem2; int16_t* dmin; int16_t* slot; int16_t* plist; int32_t* poff;
  uint8_t* seq; uint32_t* bits;
};
template <bool OUT>
__device__ __forceinline__ SeqCtx seq_stage(const BppLinArgs& a, const Seq& q, int n, unsigned char* base, const SeqLdsLayout& y) {
  SeqCtx c;
  c.cand = reinterpret_cast<SeqCand*>(base + y.cand);
  c.runc = reinterpret_cast<double*>(base + y.runc); c.runo = reinterpret_cast<int32_t*>(base + y.runo);
  c.qoff = reinterpret_cast<uint32_t*>(base + y.qoff); c.part = reinterpret_cast<double*>(base + y.part);
  c.stem = reinterpret_cast<double*>(base + y.stem); c.stem2 = reinterpret_cast<double*>(base + y.stem2);
  c.dmin = reinterpret_cast<int16_t*>(base + y.dmin); c.slot = reinterpret_cast<int16_t*>(base + y.slot);
  c.plist = reinterpret_cast<int16_t*>(base + y.plist); c.poff = reinterpret_cast<int32_t*>(base + y.poff);
  c.seq = base + y.seq; c.bits = reinterpret_cast<uint32_t*>(base + y.bits);
  const int tid = threadIdx.x, L = q.L, W = q.W;
  const SeqPlan p = a.plans[n];
  for (int t = tid; t < kBppCandMax; t += kSeqThreads) {
    const BppCand e = a.cand->e[t];
    SeqCand s;
    s.coef = e.coef; s.u1 = (int16_t)e.u1; s.u2 = (int16_t)(e.T - e.u1);
    s.off = OUT ? (uint32_t)(e.T * (L + 1) - e.u1) : (uint32_t)((kMaxLoop - e.T) * (L + 1) + e.u1);
    c.cand[t] = s;
  }
  for (int t = tid; t < kBppRunMax; t += kSeqThreads) c.runc[t] = a.cand->run_coef[t];
  for (int t = tid; t < kMaxLoop + 2; t += kSeqThreads) c.runo[t] = a.cand->run_off[t];
  for (int t = tid; t < kBppRunMax / 4; t += kSeqThreads) {     // where a quad starts, from the lane's base (low 20 bits) | T << 20 | m << 25
    const uint32_t T = a.cand->quad_T[t], m = a.cand->quad_m[t];
    const uint32_t off = OUT ? T * (uint32_t)(L + 1) - T + 2u + m : (kMaxLoop - T) * (uint32_t)(L + 1) + 2u + m;
    c.qoff[t] = off | (T << 20) | (m << 25);
  }
  const int nword = (int)((((long long)(L + 1) * (W + 1)) + 31) >> 5);
  for (int t = tid; t < nword + 2; t += kSeqThreads) c.bits[t] = (t < nword) ? q.ok[t] : 0u;
  for (int t = tid; t < L; t += kSeqThreads) c.seq[t] = q.seq[t];
  for (int t = tid; t <= L; t += kSeqThreads) { c.dmin[t] = q.dmin[t]; c.slot[t] = -1; }
  const int32_t* poff = a.poff + (size_t)n * a.poff_stride;
  for (int t = tid; t <= W + 1; t += kSeqThreads) c.poff[t] = poff[t];
  const int np = poff[W + 1];
  const int16_t* pl = a.plist + p.cell_base;
  for (int t = tid; t < np; t += kSeqThreads) c.plist[t] = pl[t];
  return c;
}

// debug clock of the per-sequence kernels (BppLinArgs::prof): thread 0 adds the cycles since its last mark to a phase
struct SeqClock {
  unsigned long long* row; unsigned long long t0;
  __device__ __forceinline__ void start(unsigned long long* r) { row = (threadIdx.x == 0) ? r : nullptr; if (row) t0 = __builtin_readcyclecounter(); }
  __device__ __forceinline__ void mark(int k) {
    if (row) { const unsigned long long t = __builtin_readcyclecounter(); atomicAdd(row + k, t - t0); t0 = t; }
  }
};
__global__ __launch_bounds__(kSeqThreads, ELEMDP_SEQ_WAVES_IN) void k6_in_seq(BppLinArgs a) {
  extern __shared__ __align__(16) unsigned char s_seq[];
  const Seq q = make_seq(a, blockIdx.x);
  const int tid = threadIdx.x, W = q.W, L = q.L;
  const SeqLdsLayout y = seq_lds_layout(a.lmax, a.wmax, a.pmax, false);
  const SeqCtx cx = seq_stage<false>(a, q, blockIdx.x, s_seq, y);
  const uint8_t* sq = cx.seq;
  const Mask mk{cx.bits, L, W};
  const EnergyTables& xet = *a.xet;
  const int Cc = (q.C < kMaxLoop) ? q.C : kMaxLoop;
  const size_t row = (size_t)(L + 1);
  const int ncm = a.lmax + 1;
  SeqClock clk; clk.start(a.prof ? a.prof + 0 : nullptr);
  __syncthreads();
  clk.mark(0);
  for (int d = 0; d <= W && d <= L; ++d) {
    const int nc = L - d + 1;
    const int tmax = (Cc < d - 2) ? Cc : d - 2;          // inside set: (k - i) + (j - l) <= C, inner span >= 2
    // rule 2, factorised: the stems (k, j) that end at j and start behind i + dmin[i]; lanes of a cell take every ns-th
    const int ns = (kSeqThreads / nc < 1) ? 1 : (kSeqThreads / nc < kSeqStem ? kSeqThreads / nc : kSeqStem);
    for (int w = tid; w < nc * ns; w += kSeqThreads) {
      const int ln = w / nc, i = w - ln * nc, j = i + d, dmi = cx.dmin[i];
      double A = 0.;
      if (!(ELEMDP_SEQ_KO & 4) && dmi > 0 && dmi < d) {
        const int smax = d - dmi;
        // (no mask test: BP_PM of a cell that is no pair is 0, and a test per stem is an LDS round trip in front of its loads)
        for (int sp0 = 1 + ln; sp0 <= smax; sp0 += kBS * ns) {
          double x1[kBS], xp[kBS];
#pragma unroll
          for (int u = 0; u < kBS; ++u) {
            const int sp = sp0 + u * ns, spc = (sp <= smax) ? sp : smax;
            x1[u] = q.in(BP_1, d - spc, i);
            const double pm = q.in(BP_PM, spc, j - spc);
            xp[u] = (sp <= smax) ? pm : 0.;
          }
#pragma unroll
          for (int u = 0; u < kBS; ++u) A = fma(x1[u], xp[u], A);
        }
      }
      cx.stem[ln * ncm + i] = A;
    }
    clk.mark(1);
    // rule 6c: the E cells (i, d) whose closing pair (i-1, j) is allowed = the pairs of diagonal d + 2
    const int lb = (tmax >= 1 && d + 2 <= W) ? cx.poff[d + 2] : 0;
    const int nE = (tmax >= 1 && d + 2 <= W) ? cx.poff[d + 3] - lb : 0;
    const int parts = (nE > 0) ? ((kSeqThreads / nE < 1) ? 1 : (kSeqThreads / nE < kParts ? kSeqThreads / nE : kParts)) : 1;
    if (nE > 0) {
      int cn[BC_CLASSES], co[BC_CLASSES];
#pragma unroll
      for (int c = 0; c < BC_CLASSES; ++c) { cn[c] = a.cand->upto[c][tmax]; co[c] = a.cand->base[c]; }
      for (int w = tid; w < nE * parts; w += kSeqThreads) {
        const int part = w / nE, ix = w - part * nE;
        const int i = cx.plist[lb + ix] + 1, j = i + d;
        const int type = bp_type(sq[i - 1], sq[j]);
        const int mi = type * 25 + sq[i] * 5 + sq[j - 1];
        const double fac[BC_CLASSES] = {xet.mismatch_i[mi], xet.mismatch_1ni[mi], is_au(type) ? xet.term_au : 1.};
        double HE = 0.;
        if (!(ELEMDP_SEQ_KO & 1) && tmax >= kBppRunMin) {
          // generic loops: the entries of one T are a run of the row d - T: quads of four neighbours (two 16-byte loads), the parts
          // of a cell deal the quads, four quads' loads in flight together; past the end of its list a lane takes the zero quad
          const double* __restrict__ pb = reinterpret_cast<const double*>(reinterpret_cast<uintptr_t>(q.tin) +
              8 * ((size_t)(BP_X + BC_I) * q.t_stride + (size_t)i) + 8 * (ptrdiff_t)row * ((ptrdiff_t)d - kMaxLoop));
          const int nq = cx.runo[tmax + 1] >> 2;
          double acc = 0., acc2 = 0.;
          for (int q0 = part; q0 < nq; q0 += kBQ * parts) {
            Dbl2 v0[kBQ], v1[kBQ];
            int qq[kBQ];
#pragma unroll
            for (int u = 0; u < kBQ; ++u) {
              const int qi = q0 + u * parts;
              qq[u] = (qi < nq) ? qi : kBppRunMax / 4 - 1;
              const double* src = pb + (cx.qoff[qq[u]] & 0xFFFFFu);
              v0[u] = *reinterpret_cast<const Dbl2*>(src); v1[u] = *reinterpret_cast<const Dbl2*>(src + 2);
            }
#pragma unroll
            for (int u = 0; u < kBQ; ++u) {
              const double* cf = cx.runc + 4 * qq[u];
              const double2 c0 = *reinterpret_cast<const double2*>(cf), c1 = *reinterpret_cast<const double2*>(cf + 2);
              acc = fma(c0.x, v0[u].x, acc); acc2 = fma(c0.y, v0[u].y, acc2);
              acc = fma(c1.x, v1[u].x, acc); acc2 = fma(c1.y, v1[u].y, acc2);
            }
          }
          HE = fac[BC_I] * (acc + acc2);
        }
        clk.mark(2);
#pragma unroll
        for (int c = BC_I + 1; c < ((ELEMDP_SEQ_KO & 1) ? 0 : BC_CLASSES); ++c) {
          // (base kMaxLoop rows below the cell's: the offsets of the entries are unsigned; only those of valid entries are added)
          const double* __restrict__ pl = reinterpret_cast<const double*>(reinterpret_cast<uintptr_t>(q.tin) +
              8 * ((size_t)(BP_X + c) * q.t_stride + (size_t)i) + 8 * (ptrdiff_t)row * ((ptrdiff_t)d - kMaxLoop));
          const int end = co[c] + cn[c];
          double acc = 0.;
          for (int t0 = co[c] + part; t0 < end; t0 += kBL * parts) {
            double cf[kBL], pv[kBL];
#pragma unroll
            for (int u = 0; u < kBL; ++u) {      // (past the end: the last entry again, with coefficient 0 -- no branch per candidate)
              const int t = t0 + u * parts;
              const SeqCand e = cx.cand[(t < end) ? t : end - 1];
              cf[u] = (t < end) ? e.coef : 0.;
              pv[u] = pl[e.off];
            }
#pragma unroll
            for (int u = 0; u < kBL; ++u) acc = fma(cf[u], pv[u], acc);
          }
          HE = fma(fac[c], acc, HE);
        }
        clk.mark(3);
        for (int sx = part; sx < ((ELEMDP_SEQ_KO & 2) ? 0 : kBppSpecial); sx += parts) {
          const int u1 = kSpecialU1[sx], u2 = kSpecialU2[sx];
          const int k = i + u1, sp = d - u1 - u2;
          if (u1 + u2 <= tmax && mk.ok(k, sp)) HE = fma(q.in(BP_P, sp, k), loop_weight(xet, sq, i - 1, j, k, k + sp - 1), HE);
        }
        cx.part[w] = HE;
        if (part == 0) cx.slot[i] = (int16_t)ix;
      }
    }
    clk.mark(4);
    __syncthreads();
    clk.mark(5);
    for (int i = tid; i < nc; i += kSeqThreads) {
      const int j = i + d;
      const int dmi = cx.dmin[i];
      auto left_ok = [&](int dd) { return dd <= W && dd >= 0 && i + dd <= L && dmi > 0 && dd >= dmi; };
      const bool pok = mk.ok(i, d), lok = left_ok(d), mok = q.m_ok(i, d, a.m_min);
      const bool eok = i > 0 && d + 2 <= W && mk.ok(i - 1, d + 2);
      // every operand is fetched unconditionally (valid addresses: the conditions below only select): one round of loads per cell
      // instead of a dependent one per rule
      const int c = q.cell(i, d), c_up = eok ? q.cell(i - 1, d + 2) : c;
      const int d1 = (d >= 1) ? d - 1 : 0, d2 = (d >= 2) ? d - 2 : 0, i1 = (d >= 1) ? i + 1 : i;
      const double lA = q.in(BP_A, d1, i), l2 = q.in(BP_2, d1, i), lM = q.in(BP_M, d1, i1);
      const double lP = q.in(BP_P, d2, i1), lE = q.in(BP_E, d2, i1);
      const double xS = q.x(XW_STACK, c), xM = q.x(XW_ML, c), xC = q.x(XW_CLOSE, c_up), xH = q.x(XW_HP, c_up);
      const int type2 = bp_type(sq[(d >= 1) ? j - 1 : i], sq[i]);
      const int mi = type2 * 25 + sq[(j < L) ? j : L - 1] * 5 + sq[(i > 0) ? i - 1 : 0];
      const double fI = xet.mismatch_i[mi], fN = xet.mismatch_1ni[mi];
      double A = 0., HE = 0.;
      for (int k = 0; k < ns; ++k) A += cx.stem[k * ncm + i];
      const int sl = cx.slot[i];
      if (sl >= 0) {
        for (int k = 0; k < parts; ++k) HE += cx.part[k * nE + sl];
        cx.slot[i] = -1;
      }
      if (dmi > 0 && dmi < d) A += lA;      // the tail grows by the unpaired base j-1
      double vP = 0.;
      if (pok && d >= 2) vP = fma(lP, xS, lE);                                                                // rules 1b, 1a
      const double vB = lok ? A : 0.;
      const double s2 = (lok && left_ok(d - 1)) ? l2 : 0.;                                                    // rule 3a
      const double v2 = lok ? fma(vP, pok ? xM : 0., s2) : 0.;                                                // rule 3b
      const double v1 = lok ? v2 + vB : 0.;                                                                   // rules 4a, 4b
      const double sM = (mok && q.m_ok(i + 1, d - 1, a.m_min)) ? lM : 0.;                                     // rule 5a
      const double vM = mok ? sM + vB : 0.;                                                                   // rule 5b
      const double vE = eok ? fma(vM, xC, xH + HE) : 0.;                                                      // rules 6a, 6b (L = 1), 6c
      q.in(BP_P, d, i) = vP; q.in(BP_E, d, i) = vE; q.in(BP_M, d, i) = vM; q.in(BP_B, d, i) = vB;
      q.in(BP_1, d, i) = v1; q.in(BP_2, d, i) = v2; q.in(BP_A, d, i) = A;
      double xI = 0., xN = 0., xB = 0.;
      if (vP != 0.) {
        xB = is_au(type2) ? vP * xet.term_au : vP;
        if (i > 0 && j < L) { xI = vP * fI; xN = vP * fN; }
      }
      q.in(BP_X + BC_I, d, i) = xI; q.in(BP_X + BC_N, d, i) = xN; q.in(BP_X + BC_B, d, i) = xB;
      q.in(BP_PM, d, i) = (pok && vP != 0.) ? vP * xM : 0.;
    }
    clk.mark(6);
    __syncthreads();
    clk.mark(7);
  }
}

__global__ __launch_bounds__(kSeqThreads, ELEMDP_SEQ_WAVES_OUT) void k6_out_seq(BppLinArgs a) {
  extern __shared__ __align__(16) unsigned char s_seq[];
  const Seq q = make_seq(a, blockIdx.x);
  const int tid = threadIdx.x, W = q.W, L = q.L;
  const SeqLdsLayout y = seq_lds_layout(a.lmax, a.wmax, a.pmax, true);
  const SeqCtx cx = seq_stage<true>(a, q, blockIdx.x, s_seq, y);
  const uint8_t* sq = cx.seq;
  const Mask mk{cx.bits, L, W};
  const EnergyTables& xet = *a.xet;
  const int Cc = (q.C < kMaxLoop) ? q.C : kMaxLoop;
  const size_t row = (size_t)(L + 1);
  const int ncm = a.lmax + 1;
  SeqClock clk; clk.start(a.prof ? a.prof + 8 : nullptr);
  __syncthreads();
  clk.mark(0);
  for (int d = (W < L) ? W : L; d >= 0; --d) {
    const int nc = L - d + 1;
    const int tmax = (kMaxLoop < W - 2 - d) ? kMaxLoop : W - 2 - d;     // outside set: the closing pair spans at most W
    const int ns = (kSeqThreads / nc < 1) ? 1 : (kSeqThreads / nc < kSeqStem ? kSeqThreads / nc : kSeqStem);
    // H1: 1(i,j) under B(i,l) through a stem (j, l) that starts at j;  HA: what reaches 2(i,j) through rule 2
    for (int w = tid; w < nc * ns; w += kSeqThreads) {
      const int ln = w / nc, i = w - ln * nc, j = i + d, dmi = cx.dmin[i];
      const bool lok = dmi > 0 && d >= dmi;
      const bool pok = mk.ok(i, d);
      double H1 = 0., HA = 0.;
      const int hi = lok ? ((W - d < L - j) ? W - d : L - j) : 0;
      const int bmax = pok ? ((W - d < i) ? W - d : i) : 0;
      // (two loops: five operands per candidate in one cost 40 registers; no mask test: BP_PM of a cell that is no pair is 0)
      for (int n0 = 1 + ln; n0 <= hi; n0 += kBS * ns) {
        double oa[kBS], xp[kBS];
#pragma unroll
        for (int u = 0; u < kBS; ++u) {
          const int n = n0 + u * ns, nn = (n <= hi) ? n : hi;
          oa[u] = q.out(BO_A, d + nn, i);
          const double pm = q.in(BP_PM, nn, j);
          xp[u] = (n <= hi) ? pm : 0.;
        }
#pragma unroll
        for (int u = 0; u < kBS; ++u) H1 = fma(oa[u], xp[u], H1);
      }
      for (int n0 = 1 + ln; n0 <= bmax; n0 += kBS * ns) {
        double ob[kBS], x1[kBS];
#pragma unroll
        for (int u = 0; u < kBS; ++u) {
          const int n = n0 + u * ns, nn = (n <= bmax) ? n : bmax;
          ob[u] = q.out(BO_A, d + nn, i - nn);
          const double v = q.in(BP_1, nn, i - nn);
          x1[u] = (n <= bmax) ? v : 0.;
        }
#pragma unroll
        for (int u = 0; u < kBS; ++u) HA = fma(ob[u], x1[u], HA);
      }
      cx.stem[ln * ncm + i] = H1;
      cx.stem2[ln * ncm + i] = HA;
    }
    clk.mark(1);
    // HP: the interior loops around the stems (i, j-1) of this diagonal: outer E cells (i - u1, d + T)
    const int lb = (tmax >= 1) ? cx.poff[d] : 0;
    const int nP = (tmax >= 1) ? cx.poff[d + 1] - lb : 0;
    const int parts = (nP > 0) ? ((kSeqThreads / nP < 1) ? 1 : (kSeqThreads / nP < kParts ? kSeqThreads / nP : kParts)) : 1;
    if (nP > 0) {
      int cn[BC_CLASSES], co[BC_CLASSES];
#pragma unroll
      for (int c = 0; c < BC_CLASSES; ++c) { cn[c] = a.cand->upto[c][tmax]; co[c] = a.cand->base[c]; }
      for (int w = tid; w < nP * parts; w += kSeqThreads) {
        const int part = w / nP, ix = w - part * nP;
        const int i = cx.plist[lb + ix], j = i + d;
        const int amax = (Cc < i - 1) ? Cc : i - 1;          // k - i' <= C; the closing pair starts at i' - 1 >= 0
        const int rmax = L - 1 - j;                          // .. and ends at j + u2 <= L - 1
        double fac[BC_CLASSES] = {0., 0., 0.};
        {
          const int type2 = bp_type(sq[j - 1], sq[i]);
          fac[BC_B] = is_au(type2) ? xet.term_au : 1.;
          if (i > 0 && j < L) {
            const int mi = type2 * 25 + sq[j] * 5 + sq[i - 1];
            fac[BC_I] = xet.mismatch_i[mi];
            fac[BC_N] = xet.mismatch_1ni[mi];
          }
        }
        double HP = 0.;
        if (tmax >= kBppRunMin) {
          // generic loops: the entries of one T are a run of the row d + T, walked from u1 = T - 2 down (the coefficients are symmetric):
          // element m of the run is (u1, u2) = (T - 2 - m, 2 + m)
          const double* __restrict__ pb = q.tout + (size_t)(BO_X + BC_I) * q.t_stride + (size_t)d * row + i;
          const int nq = cx.runo[tmax + 1] >> 2;
          const int mhi = rmax - 2;
          double acc = 0., acc2 = 0.;
          for (int q0 = part; q0 < nq; q0 += kBQ * parts) {
            Dbl2 v0[kBQ], v1[kBQ];
            int qq[kBQ];
            uint32_t qo[kBQ];
#pragma unroll
            for (int u = 0; u < kBQ; ++u) {
              const int qi = q0 + u * parts;
              qq[u] = (qi < nq) ? qi : kBppRunMax / 4 - 1;
              qo[u] = cx.qoff[qq[u]];
              const double* src = pb + (qo[u] & 0xFFFFFu);
              v0[u] = *reinterpret_cast<const Dbl2*>(src); v1[u] = *reinterpret_cast<const Dbl2*>(src + 2);
            }
#pragma unroll
            for (int u = 0; u < kBQ; ++u) {
              const double* cf = cx.runc + 4 * qq[u];
              const double2 c0 = *reinterpret_cast<const double2*>(cf), c1 = *reinterpret_cast<const double2*>(cf + 2);
              const int T = (int)((qo[u] >> 20) & 31u), m = (int)(qo[u] >> 25);
              const int mlo = T - 2 - amax;
              acc = fma((m >= mlo && m <= mhi) ? c0.x : 0., v0[u].x, acc);
              acc2 = fma((m + 1 >= mlo && m + 1 <= mhi) ? c0.y : 0., v0[u].y, acc2);
              acc = fma((m + 2 >= mlo && m + 2 <= mhi) ? c1.x : 0., v1[u].x, acc);
              acc2 = fma((m + 3 >= mlo && m + 3 <= mhi) ? c1.y : 0., v1[u].y, acc2);
            }
          }
          HP = fac[BC_I] * (acc + acc2);
        }
        clk.mark(2);
#pragma unroll
        for (int c = BC_I + 1; c < BC_CLASSES; ++c) {
          const double* __restrict__ pl = q.tout + (size_t)(BO_X + c) * q.t_stride + (size_t)d * row + i;
          const int end = co[c] + cn[c];
          double acc = 0.;
          for (int t0 = co[c] + part; t0 < end; t0 += kBL * parts) {
            double cf[kBL], ov[kBL];
#pragma unroll
            for (int u = 0; u < kBL; ++u) {      // (an entry outside the sequence reads some finite entry of the plane -- the
              const int t = t0 + u * parts;      //  planes are cleared when they are allocated -- with coefficient 0)
              const SeqCand e = cx.cand[(t < end) ? t : end - 1];
              cf[u] = (t < end && e.u1 <= amax && e.u2 <= rmax) ? e.coef : 0.;
              ov[u] = pl[e.off];
            }
#pragma unroll
            for (int u = 0; u < kBL; ++u) acc = fma(cf[u], ov[u], acc);
          }
          HP = fma(fac[c], acc, HP);
        }
        clk.mark(3);
        for (int sx = part; sx < kBppSpecial; sx += parts) {
          const int u1 = kSpecialU1[sx], u2 = kSpecialU2[sx];
          const int io = i - u1, jo = j + u2;
          if (u1 + u2 <= tmax && u1 <= amax && u2 <= rmax && mk.ok(io - 1, jo - io + 2))
            HP = fma(q.out(BO_E, jo - io, io), loop_weight(xet, sq, io - 1, jo, i, j - 1), HP);
        }
        cx.part[w] = HP;
        if (part == 0) cx.slot[i] = (int16_t)ix;
      }
    }
    clk.mark(4);
    __syncthreads();
    clk.mark(5);
    for (int i = tid; i < nc; i += kSeqThreads) {
      const int j = i + d;
      const int dmi = cx.dmin[i];
      auto left_ok = [&](int dd) { return dd <= W && dd >= 0 && i + dd <= L && dmi > 0 && dd >= dmi; };
      const bool pok = mk.ok(i, d), lok = left_ok(d), mok = q.m_ok(i, d, a.m_min);
      const bool up_ok = i > 0 && d + 2 <= W && mk.ok(i - 1, d + 2);
      // every operand is fetched unconditionally (valid addresses: the conditions below only select)
      const double inP = q.in(BP_P, d, i), inA = q.in(BP_A, d, i), in1 = q.in(BP_1, d, i);
      const double inE = q.in(BP_E, d, i), inM = q.in(BP_M, d, i), inB = q.in(BP_B, d, i), in2 = q.in(BP_2, d, i);
      const int c = q.cell(i, d), c_up = up_ok ? q.cell(i - 1, d + 2) : c;
      const bool v2u = i > 0 && d + 2 <= W && j + 1 <= L, v1l = i > 0 && d + 1 <= W, v1r = d + 1 <= W && j + 1 <= L;
      const double lP2 = q.out(BO_P, v2u ? d + 2 : d, v2u ? i - 1 : i);
      const double lM1 = q.out(BO_M, v1l ? d + 1 : d, v1l ? i - 1 : i);
      const double l21 = q.out(BO_2, v1r ? d + 1 : d, i), lA1 = q.out(BO_A, v1r ? d + 1 : d, i);
      const double xS = q.x(XW_STACK, c_up), xC = q.x(XW_CLOSE, c_up), xE = q.x(XW_EXT, c), xM = q.x(XW_ML, c);
      const double r7l = q.lo_in[i] + q.lo_out[j];
      const int typeo = bp_type(sq[(i > 0) ? i - 1 : 0], sq[(j < L) ? j : L - 1]);
      const int mio = typeo * 25 + sq[i < L ? i : L - 1] * 5 + sq[(d > 0) ? j - 1 : ((j < L) ? j : L - 1)];
      const double fI = xet.mismatch_i[mio], fN = xet.mismatch_1ni[mio];
      double H1 = 0., HA = 0., HP = 0.;
      if (lok && in1 != 0.)
        for (int k = 0; k < ns; ++k) H1 += cx.stem[k * ncm + i];
      if (pok && inP != 0.)
        for (int k = 0; k < ns; ++k) HA += cx.stem2[k * ncm + i];
      const int sl = cx.slot[i];
      if (sl >= 0) {
        if (inP != 0.)
          for (int k = 0; k < parts; ++k) HP += cx.part[k * nP + sl];
        cx.slot[i] = -1;
      }
      const double opP = up_ok ? lP2 : 0.;
      const double oE = (up_ok && inE != 0.) ? opP : 0.;                                                        // rule 1a
      const double oP1b = (up_ok && pok && inP != 0.) ? opP * xS : 0.;                                          // rule 1b
      const bool doM = mok && q.m_ok(i - 1, d + 1, a.m_min);
      const double sM = (doM && inM != 0.) ? lM1 : 0.;                                                           // rule 5a
      const double oM = (inM != 0.) ? fma(oE, up_ok ? xC : 0., sM) : 0.;                                         // rule 6a
      const double o1 = (in1 != 0.) ? H1 : 0.;
      const double oB = (inB != 0.) ? (mok ? oM : 0.) + o1 : 0.;                                                 // rules 5b, 4b
      const bool do2 = lok && left_ok(d + 1) && j < L;
      const double s2 = (do2 && in2 != 0.) ? l21 : 0.;                                                           // rule 3a
      const double o2 = (in2 != 0.) ? o1 + s2 : 0.;                                                              // rule 4a (direct part)
      double oP = 0.;
      if (inP != 0.) {
        const double xe = pok ? xE : 0.;
        const double r7 = (xe != 0.) ? exp(r7l) * xe : 0.;                                                       // rule 7 (lo_out holds - ln Z)
        oP = r7 + oP1b + (o2 + HA) * (pok ? xM : 0.) + HP;                                                       // rules 3b, 6c
      }
      double oA = 0.;
      if (inA != 0.) oA = (lok ? oB : 0.) + ((d + 1 <= W && j < L) ? lA1 : 0.);
      q.out(BO_P, d, i) = oP; q.out(BO_E, d, i) = oE; q.out(BO_M, d, i) = oM; q.out(BO_2, d, i) = o2; q.out(BO_A, d, i) = oA;
      double xI = 0., xN = 0., xB = 0.;
      if (oE != 0.) {
        xI = oE * fI;
        xN = oE * fN;
        xB = is_au(typeo) ? oE * xet.term_au : oE;
      }
      q.out(BO_X + BC_I, d, i) = xI; q.out(BO_X + BC_N, d, i) = xN; q.out(BO_X + BC_B, d, i) = xB;
    }
    clk.mark(6);
    __syncthreads();
    clk.mark(7);
  }
}

// ---- ln BPP >= ln min_bpp: the filtered mask, the number of kept pairs, optionally ln BPP of every candidate
__global__ __launch_bounds__(kThreads) void k6_threshold(BppLinArgs a) {
  __shared__ int cnt[kThreads / 64];
  const SeqPlan p = a.plans[blockIdx.x];
  const Seq q = make_seq(a, blockIdx.x);
  const int L = p.L, W = p.W;
  const int ncell = (L + 1) * (W + 1), nword = (ncell + 31) / 32;
  int kept = 0;
  for (int wd = threadIdx.x; wd < nword; wd += kThreads) {
    const uint32_t in_bits = q.ok[wd];
    uint32_t out_bits = 0;
    for (int k = 0; k < 32; ++k) {
      if (!((in_bits >> k) & 1u)) continue;
      const int cc = wd * 32 + k;
      const int i = cc / (W + 1), d = cc - i * (W + 1);
      const double pr = q.in(BP_P, d, i) * q.out(BO_P, d, i);     // (the outside value is already divided by Z)
      const double ln = (pr > 0.) ? log(pr) : ELEMDP_NEG_INF;
      if (a.lnbpp) a.lnbpp[p.cell_base + cc] = ln;
      if (a.log_min_bpp <= ln) { out_bits |= 1u << k; ++kept; }
    }
    a.okbits_out[p.bits_base + wd] = out_bits;
  }
  for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off, 64);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = kept;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < kThreads / 64; ++w) tot += cnt[w];
    a.kept[blockIdx.x] = tot;
  }
}

}  // namespace

hipError_t launch_bpp_lin(const BppLinArgs& base, int G, int Lmax, int Wmax, hipStream_t st) {
  if (G <= 0) return hipSuccess;
  BppLinArgs a = base;
  // the candidate-table kernels need the loop energies (without them every loop weighs 1, whatever its size: the mask walk)
  const bool tab = a.cand && !a.no_ene && !getenv("ELEMDP_BPP_WALK");
  // one workgroup per sequence for a whole direction where its LDS image fits a CU (two or three per CU up to L ~ 600)
  a.lmax = Lmax; a.wmax = Wmax;
  const SeqLdsLayout yi = seq_lds_layout(Lmax, Wmax, a.pmax, false), yo = seq_lds_layout(Lmax, Wmax, a.pmax, true);
  const char* cap_env = getenv("ELEMDP_BPP_SEQ_KB");      // (experiments: the largest LDS image the per-sequence kernels take)
  const int cap_kb = cap_env ? atoi(cap_env) : 150;        // (one workgroup per CU still beats the launches per diagonal: 2 000 x L=1000 load 0.140 -> 0.108 s)
  const bool per_seq = tab && a.plist && a.poff && Lmax < 32767 && yo.total <= cap_kb * 1024 && !getenv("ELEMDP_BPP_DIAG");
  if (!per_seq) a.plist = nullptr;
  const int ncell_max = (Lmax + 1) * (Wmax + 1);
  const size_t lds_bits = sizeof(uint32_t) * (size_t)bpp_mask_words(Wmax);
  hipLaunchKernelGGL(k6_terms, dim3((ncell_max + kThreads - 1) / kThreads, G), dim3(kThreads), 0, st, a);
  if (per_seq) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k6_in_seq), hipFuncAttributeMaxDynamicSharedMemorySize, yi.total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k6_in_seq, dim3(G), dim3(kSeqThreads), (size_t)yi.total, st, a);
  } else
  for (int d = 0; d <= Wmax; ++d) {
    const int ncell = Lmax - d + 1;
    if (ncell <= 0) break;
    a.d = d;
    if (tab) hipLaunchKernelGGL(k6_in_tab, dim3((ncell + kC - 1) / kC, G), dim3(kThreads), lds_bits, st, a);
    else hipLaunchKernelGGL(k6_in, dim3((ncell + kC - 1) / kC, G), dim3(kThreads), lds_bits, st, a);
  }
  hipLaunchKernelGGL(k6_in_ext, dim3(G), dim3(64), 0, st, a);
  hipLaunchKernelGGL(k6_out_ext, dim3(G), dim3(64), 0, st, a);
  if (per_seq) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k6_out_seq), hipFuncAttributeMaxDynamicSharedMemorySize, yo.total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k6_out_seq, dim3(G), dim3(kSeqThreads), (size_t)yo.total, st, a);
  } else
  for (int d = Wmax; d >= 0; --d) {
    const int ncell = Lmax - d + 1;
    if (ncell <= 0) continue;
    a.d = d;
    if (tab) hipLaunchKernelGGL(k6_out_tab, dim3((ncell + kC - 1) / kC, G), dim3(kThreads), lds_bits, st, a);
    else hipLaunchKernelGGL(k6_out, dim3((ncell + kC - 1) / kC, G), dim3(kThreads), lds_bits, st, a);
  }
  hipLaunchKernelGGL(k6_threshold, dim3(G), dim3(kThreads), 0, st, a);
  return hipGetLastError();
}

}  // namespace elemdp
