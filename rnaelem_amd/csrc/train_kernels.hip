// train_kernels.hip -- the train evaluation as a diagonal-synchronous batch pipeline (gfx950).
//
// A group of G sequences is swept in lockstep: for every anti-diagonal d ONE launch evaluates the
// heavy sums of all cells (d) of all G sequences (one wave per cell, one lane per state tuple) and
// ONE launch finishes the cells (one lane per (cell, interval state), operands of diagonals d-1 /
// d-2 fetched up front).  Dependencies between diagonals are the kernel boundaries of the stream,
// so there is no intra-kernel barrier and every launch exposes G x (L+1-d) x S independent lanes:
// memory latency is hidden by occupancy (small kernels, many resident waves), not by per-lane
// pipelining.  Tables stay resident in HBM in the [e][d][i][s] layout (coalesced along (i,s)).
// The exterior chain O(j) is inherently sequential in j: one workgroup per sequence.
//
// Schedule per group (RNAelemTrainDP::operator(), RNAelem/motif_trainer.hpp:204-227):
//   inside: for d = 0..W { in_heavy(d); in_u(d) }; in_ext  -> Z(ari,nasi), Z(ari), Z(nasi), skip flag
//   outside pass p in {full terminals, masked}: out_ext(p); for d = W..0 { out_heavy(d,p); out_u(d,p) }
// Expected counts are accumulated per workgroup in LDS and flushed with fp64 global atomics into
// the per-sequence result row; k_reduce (kernels.hip) sums the rows in input order.
#include <hip/hip_runtime.h>

#include "dp_rules.h"
#include "kernels.h"
#include "wave_gather.h"

namespace elemdp {
namespace {

struct Views {
  ModelView m;
  __device__ explicit Views(const TrArgs& a) : m(a.restricted ? *a.layp_r : *a.layp) {}
  SeqView q;
  TableView in, out;
  int n;         // sequence index in the batch
  double* row;   // per-sequence result row
  double* tmp;   // heavy-sum temporaries of this slot
};

__device__ __forceinline__ bool make_views(const TrArgs& a, int g, Views& v) {
  const int n = a.grp[g];
  v.n = n;
  const SeqPlan p = a.plans[n];
  const ParamBlock* pb = reinterpret_cast<const ParamBlock*>(a.params);
  v.m.ints = a.restricted ? a.ints_r : a.ints;
  v.m.big = v.m.ints;
  v.m.theta = a.params + sizeof(ParamBlock) / sizeof(double);
  v.m.lambda[0] = pb->lambda[0];
  v.m.lambda[1] = pb->lambda[1];
  v.m.log_tau = pb->log_tau;
  v.m.lam_same = pb->lam_same;
  v.m.no_prf = a.no_prf;
  v.m.m_min = a.m_min;
  SeqView& q = v.q;
  q.L = p.L; q.W = p.W; q.C = p.C;
  q.seq = a.b.seq + p.seq_base;
  q.ws = a.b.ws + p.pos_base;
  q.unp = a.b.unp + p.pos_base;
  q.okbits = a.okbits + p.bits_base;
  q.dmin = a.p.dmin + p.dmin_base;
  q.e_stack = a.p.e_stack + p.cell_base; q.e_ext = a.p.e_ext + p.cell_base; q.e_ml = a.p.e_ml + p.cell_base;
  q.e_close = a.p.e_close + p.cell_base; q.e_hp = a.p.e_hp + p.cell_base;
  q.items = a.p.items + p.item_base; q.item_in = a.p.item_in + p.item_base;
  q.by_outer_off = a.p.by_outer_off + p.off_base;
  q.by_inner_off = a.p.by_inner_off + p.off_base; q.by_inner_idx = a.p.by_inner_idx + p.item_base;
  q.by_left_off = a.p.by_left_off + p.off_base; q.by_left_idx = a.p.by_left_idx + p.item_base;
  q.by_right_off = a.p.by_right_off + p.off_base; q.by_right_idx = a.p.by_right_idx + p.item_base;
  v.in.band = a.band_in + (size_t)g * a.band_stride;
  v.in.ext = a.ext_in + (size_t)g * a.ext_stride;
  v.out.band = a.band_out + (size_t)g * a.band_stride;
  v.out.ext = a.ext_out + (size_t)g * a.ext_stride;
  v.in.L = v.out.L = p.L; v.in.W = v.out.W = p.W; v.in.S = v.out.S = a.lay.S;
  v.row = a.seq_out + (size_t)n * a.out_stride;
  v.tmp = a.tmp + (size_t)g * 3 * a.tmp_stride;
  return true;
}

// ---- inside, heavy sums of diagonal d: grid (ceil(ncell / kWaves), G), one wave per cell
__global__ __launch_bounds__(kThreads) void k3_in_heavy(TrArgs a) {
  __shared__ double scr[kWaves * 128];
  Views v(a);
  make_views(a, blockIdx.y, v);
  const int wave = threadIdx.x >> 6;
  const int i = blockIdx.x * kWaves + wave;
  if (a.d > v.q.W || i > v.q.L - a.d) return;
  WaveCtx w;
  w.lane = threadIdx.x & 63;
  w.scr_m = scr + wave * 128;
  w.scr_s = w.scr_m + 64;
  heavy_inside_cell(v.m, v.q, v.in, w, a.d, i, v.tmp);
}

// ---- inside, finish diagonal d: grid (ceil(ncell * S / kThreads), G), one lane per (cell, state)
// SERIAL: the heavy sums are evaluated by the lane itself (used when only one state is swept, where a wave per
// cell would leave 63 lanes idle: the BPP filter and the no-motif pass)
template <bool SERIAL>
__global__ __launch_bounds__(kThreads) void k3_in_u(TrArgs a) {
  Views v(a);
  make_views(a, blockIdx.y, v);
  const int S = v.m.lay.S, NA = v.m.lay.n_active, d = a.d;
  if (d > v.q.W) return;
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= (v.q.L - d + 1) * NA) return;
  const int i = t / NA, s = t - i * NA;
  const Constraint c{-1, -1, 0};
  if (SERIAL) { inside_target<false>(v.m, v.q, v.in, c, d, i, s); return; }
  const double HB = v.q.left_ok(i, d) ? v.in.at(ST_B, d, i, s) : ELEMDP_NEG_INF;
  const double HE = v.q.e_ok(i, d) ? v.tmp[(size_t)i * S + s] : ELEMDP_NEG_INF;
  inside_target_u<false>(v.m, v.q, v.in, c, d, i, s, HB, HE);
}

// ---- one-state automaton (the BPP filter): an exterior step has up to W candidate pairs and ONE state -- instead of one
// lane walking them with a chain of dependent loads, the lanes of the first wave take one candidate each and the partial
// log-sums meet in a butterfly (fixed order: deterministic).
template <int LANES = 64>
__device__ __forceinline__ void wave_lse_merge(LseAcc& a) {
  // log of the sum over a group of LANES lanes: group maximum, ONE exponential per lane, group sum (butterflies: every
  // lane gets the result, in a fixed order)
  double mx = a.m;
  for (int off = LANES / 2; off > 0; off >>= 1) { const double o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
  double sc = (a.m == ELEMDP_NEG_INF) ? 0. : a.s * exp_neg(a.m - mx);
  for (int off = LANES / 2; off > 0; off >>= 1) sc += __shfl_xor(sc, off, 64);
  a.m = mx;
  a.s = (mx == ELEMDP_NEG_INF) ? 0. : sc;
}

// ---- exterior chain of the inside pass + partition functions: one workgroup (128 lanes) per sequence
__global__ __launch_bounds__(128) void k3_in_ext(TrArgs a) {
  Views v(a);
  make_views(a, blockIdx.x, v);
  const int S = v.m.lay.n_active, tid = threadIdx.x;
  const Constraint c{-1, -1, 0};
  for (int s = tid; s < S; s += 128) v.in.o(0, s) = (s == a.lay.s00) ? 0. : ELEMDP_NEG_INF;
  __syncthreads();
  if (a.lay.S == 1) {   // (one state, one split tuple (0,0), one right transition 0 -> 0)
    const ModelView& m = v.m;
    const SeqView& q = v.q;
    const double lam = m.lam(0);
    const int tf = m.ints[m.lay.right_ent + 1];
    for (int j = 1; j <= q.L; ++j) {
      if (tid < 64) {
        LseAcc acc;
        const int i0 = (j - q.W > 0) ? j - q.W : 0;
        for (int i = j - 1 - tid; i >= i0; i -= 64) {   // rule 7
          const int d = j - i;
          const bool ok = q.pair_ok(i, d);
          const double t = q.e_ext[q.cell(i, ok ? d : 0)];
          const double term = v.in.o(i, 0) + (v.in.at(ST_P, d, i, 0) + lam * t);
          if (ok && t != ELEMDP_NEG_INF) acc.add(term);
        }
        if (tid == 0 && q.unp[j - 1]) acc.add(v.in.o(j - 1, 0) + w_right(m, q, 0, tf, j - 1));   // rule 8
        wave_lse_merge(acc);
        if (tid == 0) v.in.o(j, 0) = acc.value();
      }
      __syncthreads();
    }
  } else {
    for (int j = 1; j <= v.q.L; ++j) {
      for (int s = tid; s < S; s += 128) inside_ext_target<false>(v.m, v.q, v.in, c, j, s);
      __syncthreads();
    }
  }
  if (tid == 0) {
    const double Zo = part_func(v.m, v.in, true, true), Za = part_func(v.m, v.in, true, false),
                 Zn = part_func(v.m, v.in, false, true);
    const bool skip = !(isfinite(Zo) && isfinite(Za));   // motif_trainer.hpp:211-215
    const SeqPlan p = a.plans[v.n];
    v.row[0] = Zo; v.row[1] = Za; v.row[2] = Zn;
    // f_n = Z(ari,nasi) - Z(label); with --lik-ratio a sequence without motif contributes Z(ari) - Z(ari,nasi)
    v.row[3] = skip ? 0. : (p.positive ? Zo - Za : (a.lik_ratio ? Za - Zo : Zo - Zn));
    v.row[4] = skip ? 1. : 0.;
    v.row[5] = skip ? 0. : p.bpp_eff;
  }
}

struct PassInfo { double Z; bool ari, nasi, skip; int en_off, eh_off; };
// schedule 0 (reference, motif_trainer.hpp:209-225): pass 0 = terminals (ari,nasi), Z = Z(ari,nasi);
//                                                     pass 1 = the label's mask, Z = Z(ari) or Z(nasi)
// schedule 1 (linear): pass 0 = terminals ari only, Z = Z(ari); pass 1 = nasi only on the one-state automaton,
//                      Z = Z(nasi); k3_combine turns the two statistics into those of schedule 0.
__device__ __forceinline__ PassInfo pass_info(const TrArgs& a, const Views& v) {
  PassInfo pi;
  const int nt = a.lay.n_theta;
  const bool positive = a.plans[v.n].positive != 0;
  pi.skip = v.row[4] != 0.;
  if (a.schedule == 0) {
    if (a.pass == 0) { pi.Z = v.row[0]; pi.ari = true; pi.nasi = true; }
    else {   // the label's mask; --lik-ratio uses the "has motif" terminals for both labels (the roles are swapped in k3_combine)
      const bool use_ari = positive || a.lik_ratio;
      pi.Z = use_ari ? v.row[1] : v.row[2]; pi.ari = use_ari; pi.nasi = !use_ari;
    }
  } else {
    if (a.pass == 0) { pi.Z = v.row[1]; pi.ari = true; pi.nasi = false; }
    else { pi.Z = v.row[2]; pi.ari = false; pi.nasi = true; }
    if (pi.Z == ELEMDP_NEG_INF) pi.skip = true;   // that component carries no probability mass
  }
  pi.en_off = 6 + a.pass * nt;
  pi.eh_off = 6 + 2 * nt + 2 * a.pass;
  return pi;
}

// flush the workgroup's LDS statistics into the sequence's result row
__device__ __forceinline__ void flush_stats(const TrArgs& a, const Views& v, const PassInfo& pi, GpuSink& sink, double* l_en,
                                            double* l_eh) {
  const double e0 = wave_sum(sink.eh0), e1 = wave_sum(sink.eh1);
  if ((threadIdx.x & 63) == 0) { atomicAdd(&l_eh[0], e0); atomicAdd(&l_eh[1], e1); }
  __syncthreads();
  const int nt = a.lay.n_theta;
  for (int t = threadIdx.x; t < nt; t += blockDim.x) {
    const double val = l_en[t];
    if (val != 0.) atomicAdd(&v.row[pi.en_off + t], val);
  }
  if (threadIdx.x < 2 && l_eh[threadIdx.x] != 0.) atomicAdd(&v.row[pi.eh_off + threadIdx.x], l_eh[threadIdx.x]);
}

// ---- exterior chain of an outside pass (runs before the band of that pass)
template <int MODE>
__global__ __launch_bounds__(128) void k3_out_ext(TrArgs a) {
  extern __shared__ double l_stat[];   // n_theta + 2
  Views v(a);
  make_views(a, blockIdx.x, v);
  const PassInfo pi = pass_info(a, v);
  if (pi.skip) return;
  const int S = v.m.lay.n_active, tid = threadIdx.x, nt = a.lay.n_theta;
  double* l_en = l_stat;
  double* l_eh = l_stat + nt;
  for (int t = tid; t < nt + 2; t += 128) l_stat[t] = 0.;
  GpuSink sink;
  sink.en_ = l_en;
  sink.post_[0] = sink.post_[1] = sink.post_[2] = nullptr;
  sink.eh0 = sink.eh1 = 0.;
  OutCtx<GpuSink> x{v.m, v.q, v.in, v.out, pi.Z, Constraint{-1, -1, 0}, sink};
  for (int s = tid; s < S; s += 128) {
    double t = ELEMDP_NEG_INF;
    if (pi.nasi && s == a.lay.s00) t = 0.;
    if (pi.ari && (s == a.lay.s0m1 || s == a.lay.s0m2)) t = 0.;
    v.out.o(v.q.L, s) = t;
  }
  __syncthreads();
  if (MODE == OUT_NONE && a.lay.S == 1) {   // (BPP filter: no statistics, one state, tuple (0,0))
    const ModelView& m = v.m;
    const SeqView& q = v.q;
    const double lam = m.lam(0);
    const int tf = m.ints[m.lay.rright_ent + 1];
    for (int i = q.L - 1; i >= 0; --i) {
      if (tid < 64) {
        const double in_c = v.in.o(i, 0);
        LseAcc acc;
        const int jmax = (i + q.W < q.L) ? i + q.W : q.L;
        for (int j = i + 1 + tid; j <= jmax; j += 64) {   // rule 7
          const int d = j - i;
          const bool ok = q.pair_ok(i, d);
          const double t = q.e_ext[q.cell(i, ok ? d : 0)];
          const double term = v.out.o(j, 0) + (v.in.at(ST_P, d, i, 0) + lam * t);
          if (ok && t != ELEMDP_NEG_INF) acc.add(term);
        }
        if (tid == 0 && q.unp[i]) acc.add(v.out.o(i + 1, 0) + w_right(m, q, 0, tf, i));   // rule 8
        wave_lse_merge(acc);
        if (tid == 0) v.out.o(i, 0) = (in_c == ELEMDP_NEG_INF) ? ELEMDP_NEG_INF : acc.value();
      }
      __syncthreads();
    }
  } else {
    for (int i = v.q.L - 1; i >= 0; --i) {
      for (int s = tid; s < S; s += 128) outside_ext_target<MODE>(x, i, s);
      __syncthreads();
    }
  }
  if (MODE == OUT_TRAIN) flush_stats(a, v, pi, sink, l_en, l_eh);
}

// ---- BPP filter on small batches (the mini-batch training mode loads 64-128 sequences per evaluation): with one lane
// per cell a launch lasts as long as its longest chain of dependent loads (~100 split points / items); here a group of
// kBppLanes lanes takes a cell, its lanes the split points / items, and the partial log-sums meet in a butterfly.  Same sums, another (fixed)
// order.  One state, one tuple per list.
constexpr int kBppLanes = 16;   // lanes per cell (a wave takes 64 / kBppLanes cells)
__global__ __launch_bounds__(kThreads) void k3_bpp_in_wave(TrArgs a) {
  Views v(a);
  make_views(a, blockIdx.y, v);
  const ModelView& m = v.m;
  const SeqView& q = v.q;
  const int d = a.d, lane = threadIdx.x % kBppLanes;
  if (d > q.W) return;
  const int i = blockIdx.x * (kThreads / kBppLanes) + threadIdx.x / kBppLanes;
  if (i > q.L - d) return;
  const int j = i + d;
  double HB = ELEMDP_NEG_INF, HE = ELEMDP_NEG_INF;
  if (q.left_ok(i, d)) {   // rule 2 (uniform per wave)
    LseAcc acc;
    for (int k = i + q.dmin[i] + lane; k < j; k += kBppLanes)
      if (bif_valid(q, j, k)) acc.add(bif_term(v.in, i, j, k, 0, 0));
    wave_lse_merge<kBppLanes>(acc);
    HB = acc.value();
  }
  if (q.e_ok(i, d)) {      // rule 6c
    LseAcc acc;
    const double lam = m.lam(0);
    const int c0 = q.by_outer_off[q.cell(i, d)], c1 = q.by_outer_off[q.cell(i, d) + 1];
    for (int it = c0 + lane; it < c1; it += kBppLanes) {
      const LoopItem x = q.items[it];
      const double term = loop_term(v.in, i, j, x, 0, 0, 0, lam * x.tsc);
      if (q.item_in[it]) acc.add(term);
    }
    wave_lse_merge<kBppLanes>(acc);
    HE = acc.value();
  }
  if (lane == 0) inside_target_u<false>(m, q, v.in, Constraint{-1, -1, 0}, d, i, 0, HB, HE);
}

__global__ __launch_bounds__(kThreads) void k3_bpp_out_wave(TrArgs a) {
  Views v(a);
  make_views(a, blockIdx.y, v);
  const PassInfo pi = pass_info(a, v);
  const ModelView& m = v.m;
  const SeqView& q = v.q;
  const int d = a.d, lane = threadIdx.x % kBppLanes;
  if (pi.skip || d > q.W) return;
  const int i = blockIdx.x * (kThreads / kBppLanes) + threadIdx.x / kBppLanes;
  if (i > q.L - d) return;
  const int j = i + d;
  GpuSink sink;
  sink.en_ = nullptr;
  sink.post_[0] = sink.post_[1] = sink.post_[2] = nullptr;
  sink.eh0 = sink.eh1 = 0.;
  OutCtx<GpuSink> x{m, q, v.in, v.out, pi.Z, Constraint{-1, -1, 0}, sink};
  HeavyOut H;
  H.H1 = H.H2 = H.HP = H.HL = ELEMDP_NEG_INF;
  if (q.left_ok(i, d)) {
    if (v.in.at(ST_1, d, i, 0) != ELEMDP_NEG_INF) {   // H1: parents B(i, jj), jj > j
      LseAcc acc;
      const int dj = q.dmin[j];
      if (j < q.L && dj > 0) {
        const int jmax = (i + q.W < q.L) ? i + q.W : q.L;
        for (int jj = j + dj + lane; jj <= jmax; jj += kBppLanes) acc.add(o1_term(v.in, v.out, i, j, jj, 0, 0));
      }
      wave_lse_merge<kBppLanes>(acc);
      H.H1 = acc.value();
    }
    if (v.in.at(ST_2, d, i, 0) != ELEMDP_NEG_INF) {   // H2: parents B(ii, j), ii < i
      LseAcc acc;
      const int imin = (j - q.W > 0) ? j - q.W : 0;
      for (int ii = i - 1 - lane; ii >= imin; ii -= kBppLanes)
        if (o2_valid(q, i, ii)) acc.add(o2_term(v.in, v.out, i, j, ii, 0, 0));
      wave_lse_merge<kBppLanes>(acc);
      H.H2 = acc.value();
    }
  }
  if (q.pair_ok(i, d) && v.in.at(ST_P, d, i, 0) != ELEMDP_NEG_INF) {   // HP: the interior loops around the pair
    LseAcc acc;
    const double lam = m.lam(0);
    const int pc = q.cell(i, d);
    for (int n = q.by_inner_off[pc] + lane; n < q.by_inner_off[pc + 1]; n += kBppLanes) {
      const LoopItem it = q.items[q.by_inner_idx[n]];
      acc.add(oP_term(v.in, v.out, i, j, it, 0, 0, 0, lam * it.tsc));
    }
    wave_lse_merge<kBppLanes>(acc);
    H.HP = acc.value();
  }
  if (lane == 0) outside_target_u<OUT_NONE>(x, d, i, 0, H);
}

// ---- outside, heavy sums of diagonal d
__global__ __launch_bounds__(kThreads) void k3_out_heavy(TrArgs a) {
  __shared__ double scr[kWaves * 128];
  __shared__ double l_eh[2];
  Views v(a);
  make_views(a, blockIdx.y, v);
  const PassInfo pi = pass_info(a, v);
  if (pi.skip || a.d > v.q.W) return;
  const int wave = threadIdx.x >> 6;
  const int i = blockIdx.x * kWaves + wave;
  if (threadIdx.x < 2) l_eh[threadIdx.x] = 0.;
  __syncthreads();
  GpuSink sink;
  sink.en_ = nullptr;
  sink.post_[0] = sink.post_[1] = sink.post_[2] = nullptr;
  sink.eh0 = sink.eh1 = 0.;
  if (i <= v.q.L - a.d) {
    OutCtx<GpuSink> x{v.m, v.q, v.in, v.out, pi.Z, Constraint{-1, -1, 0}, sink};
    WaveCtx w;
    w.lane = threadIdx.x & 63;
    w.scr_m = scr + wave * 128;
    w.scr_s = w.scr_m + 64;
    heavy_outside_cell<OUT_TRAIN>(x, w, a.d, i, v.tmp, a.tmp_stride);
  }
  const double e0 = wave_sum(sink.eh0), e1 = wave_sum(sink.eh1);
  if ((threadIdx.x & 63) == 0) {
    if (e0 != 0.) atomicAdd(&v.row[pi.eh_off], e0);
    if (e1 != 0.) atomicAdd(&v.row[pi.eh_off + 1], e1);
  }
}

// ---- outside, finish diagonal d
template <bool SERIAL, int MODE>
__global__ __launch_bounds__(kThreads) void k3_out_u(TrArgs a) {
  extern __shared__ double l_stat[];   // n_theta + 2
  Views v(a);
  make_views(a, blockIdx.y, v);
  const PassInfo pi = pass_info(a, v);
  const int S = v.m.lay.S, NA = v.m.lay.n_active, d = a.d, nt = a.lay.n_theta;
  if (pi.skip || d > v.q.W) return;
  if ((int)(blockIdx.x * kThreads) >= (v.q.L - d + 1) * NA) return;
  double* l_en = l_stat;
  double* l_eh = l_stat + nt;
  for (int t = threadIdx.x; t < nt + 2; t += kThreads) l_stat[t] = 0.;
  __syncthreads();
  GpuSink sink;
  sink.en_ = l_en;
  sink.post_[0] = sink.post_[1] = sink.post_[2] = nullptr;
  sink.eh0 = sink.eh1 = 0.;
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t < (v.q.L - d + 1) * NA) {
    const int i = t / NA, s = t - i * NA;
    OutCtx<GpuSink> x{v.m, v.q, v.in, v.out, pi.Z, Constraint{-1, -1, 0}, sink};
    if (SERIAL) outside_target<MODE>(x, d, i, s);
    HeavyOut H;
    const bool lok = v.q.left_ok(i, d);
    H.H1 = lok ? v.out.at(ST_1, d, i, s) : ELEMDP_NEG_INF;
    H.H2 = lok ? v.tmp[0 * a.tmp_stride + (size_t)i * S + s] : ELEMDP_NEG_INF;
    H.HP = v.q.pair_ok(i, d) ? v.tmp[1 * a.tmp_stride + (size_t)i * S + s] : ELEMDP_NEG_INF;
    H.HL = v.tmp[2 * a.tmp_stride + (size_t)i * S + s];
    if (!SERIAL) outside_target_u<MODE>(x, d, i, s, H);
  }
  __syncthreads();
  if (MODE == OUT_TRAIN) flush_stats(a, v, pi, sink, l_en, l_eh);
}

// Final statistics of a sequence from those of the two outside passes (A in the "o" slot, B in the "x" slot).
// schedule 1: A = ari-only, B = nasi-only.  Outside values are linear in the terminal vector, so with p_a = Z(ari)/Z,
//   p_n = Z(nasi)/Z the full-terminal statistics are F = p_a A + p_n B; the reference's pair (o, x) is (F, A) for a
//   sequence with motif, (F, B) without -- and (A, F) without motif under --lik-ratio (motif_trainer.hpp:163-171).
// schedule 0 (the reference's own two passes): only --lik-ratio needs work here: (o, x) were computed as (F, A) and are
//   swapped for a sequence without motif.
__global__ __launch_bounds__(kThreads) void k3_combine(TrArgs a, int G) {
  const int g = blockIdx.x;
  if (g >= G) return;
  const int n = a.grp[g];
  double* row = a.seq_out + (size_t)n * a.out_stride;
  if (row[4] != 0.) return;
  const int nt = a.lay.n_theta;
  const bool positive = a.plans[n].positive != 0;
  const double pa = (row[1] == ELEMDP_NEG_INF) ? 0. : exp(row[1] - row[0]);
  const double pn = (row[2] == ELEMDP_NEG_INF) ? 0. : exp(row[2] - row[0]);
  for (int t = threadIdx.x; t < nt + 2; t += kThreads) {
    double* A = (t < nt) ? &row[6 + t] : &row[6 + 2 * nt + (t - nt)];
    double* B = (t < nt) ? &row[6 + nt + t] : &row[6 + 2 * nt + 2 + (t - nt)];
    const double va = *A, vb = *B;
    if (a.schedule == 1) {
      const double full = pa * va + pn * vb;
      if (a.lik_ratio && !positive) { *A = va; *B = full; }
      else { *A = full; *B = positive ? va : vb; }
    } else if (a.lik_ratio && !positive) {
      *A = vb; *B = va;
    }
  }
}

// ---- K1 tail: ln BPP of every candidate pair and the min_bpp threshold (energy_model.hpp:195-201, 257-261)
__global__ __launch_bounds__(kThreads) void k3_bpp_threshold(TrArgs a, BppOut o) {
  __shared__ int cnt[kWaves];
  Views v(a);
  make_views(a, blockIdx.x, v);
  const SeqPlan p = a.plans[v.n];
  const int L = p.L, W = p.W;
  const int ncell = (L + 1) * (W + 1), nword = (ncell + 31) / 32;
  const double Z = v.in.o(L, 0);
  int kept = 0;
  for (int wd = threadIdx.x; wd < nword; wd += kThreads) {
    const uint32_t in_bits = v.q.okbits[wd];
    uint32_t out_bits = 0;
    for (int k = 0; k < 32; ++k) {
      if (!((in_bits >> k) & 1u)) continue;
      const int cc = wd * 32 + k;
      const int i = cc / (W + 1), d = cc - i * (W + 1);
      const double ln = (v.in.at(ST_P, d, i, 0) + v.out.at(ST_P, d, i, 0)) - Z;
      if (o.lnbpp) o.lnbpp[p.cell_base + cc] = ln;
      if (o.log_min_bpp <= ln) { out_bits |= 1u << k; ++kept; }
    }
    o.okbits_out[p.bits_base + wd] = out_bits;
  }
  for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off, 64);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = kept;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < kWaves; ++w) tot += cnt[w];
    o.kept[v.n] = tot;
  }
}

}  // namespace

// K1 for one group: plain McCaskill inside / outside through the one-state automaton, then the threshold.
hipError_t launch_bpp_group(const TrArgs& base, const BppOut& o, int G, int Lmax, int Wmax, hipStream_t st) {
  if (G <= 0) return hipSuccess;
  TrArgs a = base;
  a.restricted = 0;
  // the one-state automaton of the filter: kBppLanes lanes per cell (measured faster than a lane per cell at 64 and at
  // 5 000 sequences: 34 + 56 ms instead of 80 + 60 ms per 2 000 x L=300)
  const bool wave_cells = a.lay.S == 1;
  const int cpbw = kThreads / kBppLanes;
  for (int d = 0; d <= Wmax; ++d) {
    const int ncell = Lmax - d + 1;
    if (ncell <= 0) break;
    a.d = d;
    if (wave_cells) hipLaunchKernelGGL(k3_bpp_in_wave, dim3((ncell + cpbw - 1) / cpbw, G), dim3(kThreads), 0, st, a);
    else hipLaunchKernelGGL(k3_in_u<true>, dim3((ncell + kThreads - 1) / kThreads, G), dim3(kThreads), 0, st, a);
  }
  hipLaunchKernelGGL(k3_in_ext, dim3(G), dim3(128), 0, st, a);
  a.schedule = 1;   // "nasi only": terminal O(L, 0), Z = O(L, 0)
  a.pass = 1;
  hipLaunchKernelGGL(k3_out_ext<OUT_NONE>, dim3(G), dim3(128), sizeof(double) * 2, st, a);
  for (int d = Wmax; d >= 0; --d) {
    const int ncell = Lmax - d + 1;
    if (ncell <= 0) continue;
    a.d = d;
    if (wave_cells) hipLaunchKernelGGL(k3_bpp_out_wave, dim3((ncell + cpbw - 1) / cpbw, G), dim3(kThreads), 0, st, a);
    else hipLaunchKernelGGL((k3_out_u<true, OUT_NONE>), dim3((ncell + kThreads - 1) / kThreads, G), dim3(kThreads), sizeof(double) * 2, st, a);
  }
  hipLaunchKernelGGL(k3_bpp_threshold, dim3(G), dim3(kThreads), 0, st, a, o);
  return hipGetLastError();
}

// Enqueues one whole train evaluation of the group on `st`.
hipError_t launch_train_group(const TrArgs& base, int G, int Lmax, int Wmax, hipStream_t st) {
  if (G <= 0) return hipSuccess;
  TrArgs a = base;
  a.restricted = 0;
  const int S = a.lay.S;
  const size_t stat_lds = sizeof(double) * (a.lay.n_theta + 2);
  const bool wave_heavy = S > 1;   // one state: the lane-serial form wastes nothing
  if (!a.no_rss) {
    for (int d = 0; d <= Wmax; ++d) {
      const int ncell = Lmax - d + 1;
      if (ncell <= 0) break;
      a.d = d;
      if (wave_heavy) {
        hipLaunchKernelGGL(k3_in_heavy, dim3((ncell + kWaves - 1) / kWaves, G), dim3(kThreads), 0, st, a);
        hipLaunchKernelGGL(k3_in_u<false>, dim3((ncell * S + kThreads - 1) / kThreads, G), dim3(kThreads), 0, st, a);
      } else {
        hipLaunchKernelGGL(k3_in_u<true>, dim3((ncell + kThreads - 1) / kThreads, G), dim3(kThreads), 0, st, a);
      }
    }
  }
  hipLaunchKernelGGL(k3_in_ext, dim3(G), dim3(128), 0, st, a);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && a.first_pass_only) break;
    a.pass = pass;
    a.restricted = (a.schedule == 1 && pass == 1) ? 1 : 0;
    const bool serial = a.restricted || !wave_heavy;
    const int na = a.restricted ? 1 : S;
    hipLaunchKernelGGL(k3_out_ext<OUT_TRAIN>, dim3(G), dim3(128), stat_lds, st, a);
    if (!a.no_rss) {
      for (int d = Wmax; d >= 0; --d) {
        const int ncell = Lmax - d + 1;
        if (ncell <= 0) continue;
        a.d = d;
        if (serial) {
          hipLaunchKernelGGL((k3_out_u<true, OUT_TRAIN>), dim3((ncell * na + kThreads - 1) / kThreads, G), dim3(kThreads), stat_lds, st, a);
        } else {
          hipLaunchKernelGGL(k3_out_heavy, dim3((ncell + kWaves - 1) / kWaves, G), dim3(kThreads), 0, st, a);
          hipLaunchKernelGGL((k3_out_u<false, OUT_TRAIN>), dim3((ncell * S + kThreads - 1) / kThreads, G), dim3(kThreads), stat_lds, st, a);
        }
      }
    }
  }
  if ((a.schedule == 1 || a.lik_ratio) && !a.first_pass_only) hipLaunchKernelGGL(k3_combine, dim3(G), dim3(kThreads), 0, st, a, G);
  return hipGetLastError();
}

}  // namespace elemdp
