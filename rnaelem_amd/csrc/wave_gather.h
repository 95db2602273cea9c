// wave_gather.h -- device-only (HIP) building blocks shared by the DP kernels: statistics sink,
// wave-level reductions and the wave-per-cell / lane-per-tuple evaluation of the heavy sums.
#pragma once
#include <hip/hip_runtime.h>

#include "dp_rules.h"
#include "kernels.h"

namespace elemdp {

__device__ __forceinline__ void lse_atomic(double* addr, double z) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = *p, assumed;
  do {
    assumed = old;
    const double nv = lse2(__longlong_as_double((long long)assumed), z);
    old = atomicCAS(p, assumed, (unsigned long long)__double_as_longlong(nv));
  } while (old != assumed);
}

// ---------------------------------------------------------------------------------------------
// statistics sinks
// ---------------------------------------------------------------------------------------------
struct GpuSink {
  double* en_;       // LDS: expected emission counts of the running pass
  double* post_[3];  // LDS: log-space position posteriors (start, inner, end) or null
  double eh0, eh1;   // lane-private energy statistics
  __device__ __forceinline__ void en(int idx, double w) { atomicAdd(&en_[idx], w); }
  __device__ __forceinline__ void eh(int k, double w) { eh1 += k ? w : 0.; eh0 += k ? 0. : w; }   // (selects keep the sink in registers)
  __device__ __forceinline__ void pos(int which, int p, double z) { if (post_[which]) lse_atomic(&post_[which][p], z); }
};

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Heavy phase: one WAVE per cell, one LANE per state tuple of the rule, the k / item loop unrolled
// four-fold so that eight (twelve) independent table loads are in flight per lane before the first
// exp.  Each lane keeps a streaming log-sum-exp of its tuple; the partial sums of the tuples that
// feed the same interval state (contiguous in the grouped lists) are merged through a 1 KiB LDS
// scratch per wave by the lanes that own the states.  Supports S <= 2 * 64 states.
// ---------------------------------------------------------------------------------------------
constexpr int kWaves = kThreads / 64;
constexpr int kStateChunks = 2;  // S <= 128

struct WaveCtx {
  double* scr_m;  // LDS, 64 doubles of this wave
  double* scr_s;
  int lane;
};

// merge the lanes' (m,s) partials of tuples [t0, t0+64) into the state lanes' accumulators
__device__ __forceinline__ void merge_tuples(const WaveCtx& w, const int32_t* G, int off_list, int S, int t0, int n_tuple,
                                             const LseAcc& lane_acc, LseAcc (&st)[kStateChunks]) {
  w.scr_m[w.lane] = lane_acc.m;
  w.scr_s[w.lane] = lane_acc.s;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int c = 0; c < kStateChunks; ++c) {
    const int s = c * 64 + w.lane;
    if (s < S) {
      int a = G[off_list + s], b = G[off_list + s + 1];
      a = a > t0 ? a : t0;
      b = b < t0 + 64 ? b : t0 + 64;
      b = b < n_tuple ? b : n_tuple;
      for (int t = a; t < b; ++t) st[c].merge(w.scr_m[t - t0], w.scr_s[t - t0]);
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// ---- inside: B(i,j,.) (rule 2) and the interior-loop part of E(i,j,.) (rule 6c) of ONE cell
__device__ inline void heavy_inside_cell(const ModelView& m, const SeqView& q, const TableView& T, const WaveCtx& w, int d, int i,
                                  double* he_tmp) {
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.big;
  const int S = A.S, j = i + d, lane = w.lane;
  const double NEG = ELEMDP_NEG_INF;
  if (q.left_ok(i, d)) {
    LseAcc st[kStateChunks];
    const int k_lo = i + q.dmin[i];
    for (int t0 = 0; t0 < A.n_split; t0 += 64) {
      const int t = t0 + lane;
      const bool act = t < A.n_split;
      const int s1 = act ? G[A.split_ent + 2 * t] : 0, s2 = act ? G[A.split_ent + 2 * t + 1] : 0;
      LseAcc acc;
      for (int kb = k_lo; kb < j; kb += 64) {
        const int kc = kb + lane;
        unsigned long long mask = __ballot(kc < j && bif_valid(q, j, kc));
        while (mask) {
          int k[4];
          bool v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            v[u] = mask != 0;
            k[u] = v[u] ? kb + __ffsll((long long)mask) - 1 : k_lo;
            mask &= mask - 1;
          }
          double x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = bif_term(T, i, j, k[u], s1, s2);  // unconditional: loads stay batched
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = (act && v[u]) ? x[u] : NEG;
#pragma unroll
          for (int u = 0; u < 4; ++u) acc.add(x[u]);
        }
      }
      merge_tuples(w, G, A.split_off, S, t0, A.n_split, acc, st);
    }
#pragma unroll
    for (int c = 0; c < kStateChunks; ++c) {
      const int s = c * 64 + lane;
      if (s < S) T.at(ST_B, d, i, s) = st[c].value();
    }
  }
  if (q.e_ok(i, d)) {
    LseAcc st[kStateChunks];
    const int c0 = q.by_outer_off[q.cell(i, d)], c1 = q.by_outer_off[q.cell(i, d) + 1];
    if (c1 > c0) {
      for (int t0 = 0; t0 < A.n_quad; t0 += 64) {
        const int t = t0 + lane;
        const bool act = t < A.n_quad;
        const int s1 = act ? G[A.quad_ent + 3 * t] : 0, s2 = act ? G[A.quad_ent + 3 * t + 1] : 0,
                  s3 = act ? G[A.quad_ent + 3 * t + 2] : 0;
        const double lam = act ? m.lam(G[A.quad_tgt + t]) : 0.;
        LseAcc acc;
        for (int it = c0; it < c1; it += 2) {
          const bool v1 = it + 1 < c1;
          const LoopItem xa = q.items[it];
          const LoopItem xb = q.items[v1 ? it + 1 : it];
          const bool ia = q.item_in[it] != 0, ib = v1 && q.item_in[v1 ? it + 1 : it] != 0;
          const double ra = loop_term(T, i, j, xa, s1, s2, s3, lam * xa.tsc);
          const double rb = loop_term(T, i, j, xb, s1, s2, s3, lam * xb.tsc);
          acc.add((act && ia) ? ra : NEG);
          acc.add((act && ib) ? rb : NEG);
        }
        merge_tuples(w, G, A.quad_off, S, t0, A.n_quad, acc, st);
      }
    }
#pragma unroll
    for (int c = 0; c < kStateChunks; ++c) {
      const int s = c * 64 + lane;
      if (s < S) he_tmp[(size_t)i * S + s] = st[c].value();
    }
  }
}

// ---- outside: the four gathers of ONE cell.  H1 goes straight into the table of state 1,
// H2 / HP / HL into the per-slot temporaries tmp[0..2][i][s].
template <int MODE>
__device__ void heavy_outside_cell(OutCtx<GpuSink>& x, const WaveCtx& w, int d, int i, double* tmp, size_t tmp_stride) {
  const ModelView& m = x.m;
  const SeqView& q = x.q;
  const TableView& in = x.in;
  const TableView& out = x.out;
  const AutomatonLayout& A = m.lay;
  const int32_t* G = m.big;
  const int S = A.S, j = i + d, lane = w.lane;
  const double NEG = ELEMDP_NEG_INF;
  const bool lok = q.left_ok(i, d);
  if (lok) {
    // H1: child 1(i,j,s1) of B(i,jj,par) with sibling 2(j,jj,s2)
    {
      LseAcc st[kStateChunks];
      const int dj = q.dmin[j];
      const int jmax = (i + q.W < q.L) ? i + q.W : q.L;
      if (j < q.L && dj > 0 && j + dj <= jmax) {
        for (int t0 = 0; t0 < A.n_split; t0 += 64) {
          const int t = t0 + lane;
          const bool act = t < A.n_split;
          const int par = act ? G[A.split1_ent + 2 * t] : 0, s2 = act ? G[A.split1_ent + 2 * t + 1] : 0;
          LseAcc acc;
          for (int jb = j + dj; jb <= jmax; jb += 4) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = o1_term(in, out, i, j, (jb + u <= jmax) ? jb + u : jmax, par, s2);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (act && jb + u <= jmax) ? v[u] : NEG;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc.add(v[u]);
          }
          merge_tuples(w, G, A.split1_off, S, t0, A.n_split, acc, st);
        }
      }
#pragma unroll
      for (int c = 0; c < kStateChunks; ++c) {
        const int s = c * 64 + lane;
        if (s < S) out.at(ST_1, d, i, s) = (in.at(ST_1, d, i, s) != NEG) ? st[c].value() : NEG;
      }
    }
    // H2: child 2(i,j,s2) of B(ii,j,par) with sibling 1(ii,i,s1)
    {
      LseAcc st[kStateChunks];
      const int imin = (j - q.W > 0) ? j - q.W : 0;
      for (int t0 = 0; t0 < A.n_split; t0 += 64) {
        const int t = t0 + lane;
        const bool act = t < A.n_split;
        const int par = act ? G[A.split2_ent + 2 * t] : 0, s1 = act ? G[A.split2_ent + 2 * t + 1] : 0;
        LseAcc acc;
        for (int ib = imin; ib < i; ib += 64) {
          const int ic = ib + lane;
          unsigned long long mask = __ballot(ic < i && o2_valid(q, i, ic));
          while (mask) {
            int ii[4];
            bool v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              v[u] = mask != 0;
              ii[u] = v[u] ? ib + __ffsll((long long)mask) - 1 : imin;
              mask &= mask - 1;
            }
            double y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) y[u] = o2_term(in, out, i, j, ii[u], par, s1);
#pragma unroll
            for (int u = 0; u < 4; ++u) y[u] = (act && v[u]) ? y[u] : NEG;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc.add(y[u]);
          }
        }
        merge_tuples(w, G, A.split2_off, S, t0, A.n_split, acc, st);
      }
#pragma unroll
      for (int c = 0; c < kStateChunks; ++c) {
        const int s = c * 64 + lane;
        if (s < S) tmp[0 * tmp_stride + (size_t)i * S + s] = st[c].value();
      }
    }
  }
  const int cellid = q.cell(i, d);
  if (q.pair_ok(i, d)) {
    // HP: inner pair P(i,j,s1) of interior loops E(i',j',par) + energy statistic
    LseAcc st[kStateChunks];
    const int n0 = q.by_inner_off[cellid], n1 = q.by_inner_off[cellid + 1];
    if (n1 > n0) {
      for (int t0 = 0; t0 < A.n_quad; t0 += 64) {
        const int t = t0 + lane;
        const bool act = t < A.n_quad;
        const int par = act ? G[A.quad1_ent + 3 * t] : 0, s2 = act ? G[A.quad1_ent + 3 * t + 1] : 0,
                  s3 = act ? G[A.quad1_ent + 3 * t + 2] : 0;
        const double lam = act ? m.lam(par) : 0.;
        const double in_c = act ? in.at(ST_P, d, i, G[A.quad1_tgt + t]) : NEG;
        const bool live = act && in_c != NEG;
        LseAcc acc;
        for (int n = n0; n < n1; n += 2) {
          const bool v1 = n + 1 < n1;
          const LoopItem xa = q.items[q.by_inner_idx[n]];
          const LoopItem xb = q.items[q.by_inner_idx[v1 ? n + 1 : n]];
          const double ra = oP_term(in, out, i, j, xa, par, s2, s3, lam * xa.tsc);
          const double rb = oP_term(in, out, i, j, xb, par, s2, s3, lam * xb.tsc);
          const double ta = live ? ra : NEG;
          const double tb = (live && v1) ? rb : NEG;
          if (MODE == OUT_TRAIN) {
            const double za = ta + in_c - x.Z, zb = tb + in_c - x.Z;
            if (live && za != NEG) x.sink.eh(m.eh_index(par), xa.tsc * exp(za));
            if (live && v1 && zb != NEG) x.sink.eh(m.eh_index(par), xb.tsc * exp(zb));
          }
          acc.add(ta);
          acc.add(tb);
        }
        merge_tuples(w, G, A.quad1_off, S, t0, A.n_quad, acc, st);
      }
    }
#pragma unroll
    for (int c = 0; c < kStateChunks; ++c) {
      const int s = c * 64 + lane;
      if (s < S) tmp[1 * tmp_stride + (size_t)i * S + s] = st[c].value();
    }
  }
  {
    // HL: left loop (cell = (it.i, it.k)) and right loop (cell = (it.l, it.j)) of interior loops
    LseAcc st[kStateChunks];
    // (OUT_NONE = BPP filter: the outside value of a loop cell is never read, its plan has no by_left / by_right order)
    const int l0 = (MODE == OUT_NONE) ? 0 : q.by_left_off[cellid], l1 = (MODE == OUT_NONE) ? 0 : q.by_left_off[cellid + 1];
    const int r0 = (MODE == OUT_NONE) ? 0 : q.by_right_off[cellid], r1 = (MODE == OUT_NONE) ? 0 : q.by_right_off[cellid + 1];
    if (l1 > l0) {
      for (int t0 = 0; t0 < A.n_quad; t0 += 64) {
        const int t = t0 + lane;
        const bool act = t < A.n_quad;
        const int par = act ? G[A.quad2_ent + 3 * t] : 0, s1 = act ? G[A.quad2_ent + 3 * t + 1] : 0,
                  s3 = act ? G[A.quad2_ent + 3 * t + 2] : 0;
        const double lam = act ? m.lam(par) : 0.;
        LseAcc acc;
        for (int n = l0; n < l1; n += 2) {
          const bool v1 = n + 1 < l1;
          const LoopItem xa = q.items[q.by_left_idx[n]];
          const LoopItem xb = q.items[q.by_left_idx[v1 ? n + 1 : n]];
          const double ra = oLl_term(in, out, xa, par, s1, s3, lam * xa.tsc);
          const double rb = oLl_term(in, out, xb, par, s1, s3, lam * xb.tsc);
          acc.add(act ? ra : NEG);
          acc.add((act && v1) ? rb : NEG);
        }
        merge_tuples(w, G, A.quad2_off, S, t0, A.n_quad, acc, st);
      }
    }
    if (r1 > r0) {
      for (int t0 = 0; t0 < A.n_quad; t0 += 64) {
        const int t = t0 + lane;
        const bool act = t < A.n_quad;
        const int par = act ? G[A.quad3_ent + 3 * t] : 0, s1 = act ? G[A.quad3_ent + 3 * t + 1] : 0,
                  s2 = act ? G[A.quad3_ent + 3 * t + 2] : 0;
        const double lam = act ? m.lam(par) : 0.;
        LseAcc acc;
        for (int n = r0; n < r1; n += 2) {
          const bool v1 = n + 1 < r1;
          const LoopItem xa = q.items[q.by_right_idx[n]];
          const LoopItem xb = q.items[q.by_right_idx[v1 ? n + 1 : n]];
          const double ra = oLr_term(in, out, xa, par, s1, s2, lam * xa.tsc);
          const double rb = oLr_term(in, out, xb, par, s1, s2, lam * xb.tsc);
          acc.add(act ? ra : NEG);
          acc.add((act && v1) ? rb : NEG);
        }
        merge_tuples(w, G, A.quad3_off, S, t0, A.n_quad, acc, st);
      }
    }
#pragma unroll
    for (int c = 0; c < kStateChunks; ++c) {
      const int s = c * 64 + lane;
      if (s < S) tmp[2 * tmp_stride + (size_t)i * S + s] = st[c].value();
    }
  }
}



}  // namespace elemdp
